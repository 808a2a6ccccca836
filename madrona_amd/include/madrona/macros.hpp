// madrona_amd header overlay: compiler / backend macros.
// API contract: reference include/madrona/macros.hpp (names only).
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MADRONA_HD __host__ __device__
#define MADRONA_DEVICE __device__
#define MADRONA_HOST_LAMBDA __host__
#else
#define MADRONA_HD
#define MADRONA_DEVICE
#define MADRONA_HOST_LAMBDA
#endif

// Host-API functions (registry, task-graph builder) are called from the
// simulator's registerTypes / setupTasks, which the build wraps in
// `#pragma clang force_cuda_host_device` and which therefore also get emitted
// for the device.  Such functions are declared MADRONA_HOST_API (host+device)
// and keep their real body inside MADRONA_HOST_ONLY; the device copy traps.
#define MADRONA_HOST_API MADRONA_HD
#if defined(__HIP_DEVICE_COMPILE__)
#define MADRONA_ON_HOST 0
#define MADRONA_DEVICE_STUB() do { __builtin_trap(); __builtin_unreachable(); } while (0)
#else
#define MADRONA_ON_HOST 1
#define MADRONA_DEVICE_STUB() do {} while (0)
#endif

#define MADRONA_ALWAYS_INLINE __attribute__((always_inline))
#define MADRONA_NO_INLINE __attribute__((noinline))
#define MADRONA_UNREACHABLE() __builtin_unreachable()
#define MADRONA_UNROLL _Pragma("unroll")
#define MADRONA_CACHE_LINE 64
#define MADRONA_EXPORT __attribute__((visibility("default")))
#define MADRONA_IMPORT

#define MADRONA_STRINGIFY_HELPER(m) #m
#define MADRONA_STRINGIFY(m) MADRONA_STRINGIFY_HELPER(m)
#define MADRONA_LOC_APPEND(m) m ": " __FILE__ " @ " MADRONA_STRINGIFY(__LINE__)

// This backend always runs many worlds behind the task graph executor.
#ifndef MADRONA_MW_MODE
#define MADRONA_MW_MODE 1
#endif
#ifndef MADRONA_USE_TASK_GRAPH
#define MADRONA_USE_TASK_GRAPH 1
#endif
#define MADRONA_MW_COND(...) __VA_ARGS__

// Wavefront width on gfx950 (the reference hard-codes 32 in mw_gpu/cu_utils.hpp)
#define MADRONA_WAVE_SIZE 64

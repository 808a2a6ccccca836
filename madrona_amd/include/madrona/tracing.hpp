// Host-side event tracing with the reference's interface (include/madrona/
// tracing.hpp, src/common/tracing.cpp): managers bracket their phases with
// HostEventLogging(HostEvent::...), FinalizeLogging(dir) writes
// <dir><pid or $MADRONA_MWGPU_TRACE_NAME>_madrona_host_tracing.bin holding the N
// event codes followed by the N time stamps (int64 each).  Header-only here;
// events are only recorded when MADRONA_TRACING is defined, as in the reference.
// Per-kernel device timing is what mwhip_profile reports (the reference's
// DeviceTracing splits its megakernel by node, this backend has one kernel per
// node to begin with).
#pragma once

#include <madrona/macros.hpp>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include <unistd.h>

namespace madrona {

enum class HostEvent : uint32_t {
    initStart = 0,
    initEnd = 1,
    megaKernelStart = 2,    // here: a replay of the step graph is queued
    megaKernelEnd = 3,
    renderStart = 4,
    renderEnd = 5,
};

struct HostTracing {
    std::vector<HostEvent> events;
    std::vector<uint64_t> time_stamps;
};

// (one log per thread, like the reference's thread_local HOST_TRACING)
inline HostTracing &hostTracing()
{
    static thread_local HostTracing tracing;
    return tracing;
}
#define HOST_TRACING (::madrona::hostTracing())

inline uint64_t GetTimeStamp()
{
#if defined(__x86_64__)
    return __builtin_ia32_rdtsc();
#else
    return (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
#endif
}

inline void HostEventLogging([[maybe_unused]] HostEvent event)
{
#ifdef MADRONA_TRACING
    HOST_TRACING.events.push_back(event);
    HOST_TRACING.time_stamps.push_back(GetTimeStamp());
#endif
}

inline void WriteToFile(void *data, size_t num_bytes, const std::string &file_path,
                        const std::string &name)
{
    std::string file_name = file_path;
    if (const char *trace_name = getenv("MADRONA_MWGPU_TRACE_NAME")) {
        file_name += std::string(trace_name) + name + ".bin";
    } else {
        file_name += std::to_string((uint32_t)getpid()) + name + ".bin";
    }
    if (FILE *f = fopen(file_name.c_str(), "wb")) {
        fwrite(data, 1, num_bytes, f);
        fclose(f);
    }
}

template <typename T>
inline void WriteToFile(T *events, size_t size, const std::string &file_path,
                        const std::string &name)
{
    WriteToFile((void *)events, size * sizeof(T), file_path, name);
}

inline void FinalizeLogging(const std::string file_path)
{
    HostTracing &t = HOST_TRACING;
    const size_t n = t.events.size();
    std::vector<int64_t> concat(n * 2);
    for (size_t i = 0; i < n; i++) {
        concat[i] = (int64_t)t.events[i];
        concat[i + n] = (int64_t)t.time_stamps[i];
    }
    WriteToFile<int64_t>(concat.data(), n * 2, file_path, "_madrona_host_tracing");
}

}

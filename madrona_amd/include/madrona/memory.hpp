// Allocation entry points simulators / engine modules may name.
// API contract: reference src/mw/device/include/madrona/memory.hpp:17-180
// (rawAlloc / rawDealloc, DefaultAlloc, mwGPU::HostAllocator,
// mwGPU::TmpAllocator) and include/madrona/memory.hpp (DefaultAlloc on the
// host).
//
// Where the memory comes from on the MI355X backend:
//  * TmpAllocator  -- the executor's per-step bump region (ecs_state::tmpBase,
//    reset by ResetTmpAllocNode), 256-byte granules like the reference's.
//  * rawAlloc / HostAllocator on the device -- the executor's persistent bump
//    region (ecs_state::persistBase): sized by the executor from a dry run of
//    the world constructors, lives as long as the executor, is never handed
//    back (rawDealloc / *Free are no-ops on the device).  The reference's device
//    code asks a host thread to reserve / map / allocate on demand through a
//    mailbox (src/mw/device/memory.cpp:27-121); here growth is the executor's
//    job between replays (table growth), so reserveMemory returns
//    the full reservation up front and mapMemory has nothing left to do.
#pragma once

#include <madrona/taskgraph.hpp>
#include <madrona/sync.hpp>

#include <cstdlib>

namespace madrona {

MADRONA_HD inline void *rawAlloc(size_t num_bytes)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return mwhip::persistAlloc(mwGPU::getStateManager(), num_bytes);
#else
    return malloc(num_bytes);
#endif
}

MADRONA_HD inline void rawDealloc(void *ptr)
{
#if defined(__HIP_DEVICE_COMPILE__)
    (void)ptr;
#else
    free(ptr);
#endif
}

class DefaultAlloc {
public:
    MADRONA_HD inline void *alloc(size_t num_bytes) { return rawAlloc(num_bytes); }
    MADRONA_HD inline void dealloc(void *ptr) { rawDealloc(ptr); }
};

// what the reference calls AllocContext / InitAlloc on the host side of some
// module constructors; the default allocator serves both roles here
using InitAlloc = DefaultAlloc;

namespace mwGPU {

class HostAllocator {
public:
    MADRONA_HD inline void *reserveMemory(uint64_t max_bytes,
                                          uint64_t init_num_bytes)
    {
        (void)init_num_bytes;
        return rawAlloc(roundUpReservation(max_bytes));
    }

    MADRONA_HD inline void *allocMemory(uint64_t num_bytes)
    {
        return rawAlloc(roundUpAlloc(num_bytes));
    }

    MADRONA_HD inline void mapMemory(void *addr, uint64_t num_bytes)
    {
        (void)addr;
        (void)num_bytes;
    }

    MADRONA_HD inline void reserveFree(void *addr, uint64_t num_bytes,
                                       uint64_t num_reserve_bytes)
    {
        (void)addr;
        (void)num_bytes;
        (void)num_reserve_bytes;
    }

    MADRONA_HD inline void allocFree(void *addr) { (void)addr; }

    // 2 MiB: the granule the executor maps table memory in
    MADRONA_HD inline uint64_t roundUpReservation(uint64_t num_bytes)
    {
        return (num_bytes + (2ull << 20) - 1) & ~((2ull << 20) - 1);
    }

    MADRONA_HD inline uint64_t roundUpAlloc(uint64_t num_bytes)
    {
        return (num_bytes + 255ull) & ~255ull;
    }
};

// (both allocators are stateless façades over the device-resident ecs_state:
// the state's address stands in for "the" allocator object)
MADRONA_HD inline HostAllocator *getHostAllocator()
{
    return (HostAllocator *)getStateManager();
}

class TmpAllocator {
public:
    MADRONA_HD inline void *alloc(uint64_t num_bytes)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        return getStateManager()->tmpAlloc(num_bytes);
#else
        (void)num_bytes;
        return nullptr;
#endif
    }

    MADRONA_HD inline void reset()
    {
#if defined(__HIP_DEVICE_COMPILE__)
        __hip_atomic_store(&getStateManager()->tmpOffset, 0ull,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    }

    MADRONA_HD static inline TmpAllocator &get()
    {
        return *(TmpAllocator *)getStateManager();
    }
};

}

}

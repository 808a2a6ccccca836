// Raw allocation entry points simulators / engine modules may call.
// API contract: reference src/mw/device/include/madrona/memory.hpp:17-29
// (rawAlloc / rawDealloc on the device).  Device-side allocations come from
// the executor's persistent bump region and live as long as the executor.
#pragma once

#include <madrona/taskgraph.hpp>

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

}

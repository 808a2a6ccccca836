// mwGPU::GPUImplConsts -- the handful of device-side globals simulators and
// engine modules reach for directly (API contract: reference
// src/mw/device/include/madrona/mw_gpu/const.hpp:14-58; used e.g. by
// src/render/ecs_system.cpp:145-157, 325-345).
//
// The reference patches these into a __constant__ struct when it loads its
// NVRTC module.  Here everything of that kind already lives in the
// device-resident ecs_state (mwhip/ecs_state.hpp), so get() assembles the
// struct from it on the fly and returns it BY VALUE (the reference returns a
// reference: `GPUImplConsts::get().numWorlds` reads the same either way).
// Members with no counterpart in this backend (the job system, the megakernel's
// task graph object, device tracing) are null.
#pragma once

#include <madrona/taskgraph.hpp>

#include <cstdint>

namespace madrona {
namespace mwGPU {

struct GPUImplConsts {
    void *jobSystemAddr;
    void *taskGraph;
    void *stateManagerAddr;
    void *worldDataAddr;
    void *hostAllocatorAddr;
    void *hostPrintAddr;
    void *tmpAllocatorAddr;
    void *deviceTracingAddr;
    void *meshBVHsAddr;
    void *bvhInternalData;
    uint32_t numWorldDataBytes;
    uint32_t numWorlds;
    uint32_t jobGridsOffset;
    uint32_t jobListOffset;
    uint32_t maxJobsPerGrid;
    uint32_t sharedJobTrackerOffset;
    uint32_t userJobTrackerOffset;
    uint32_t numMeshBVHs;
    uint32_t raycastOutputResolution;
    uint32_t raycastRGBD;

    MADRONA_HD static inline GPUImplConsts get()
    {
        GPUImplConsts c {};
#if defined(__HIP_DEVICE_COMPILE__)
        StateManager *mgr = getStateManager();
        c.stateManagerAddr = mgr;
        c.worldDataAddr = mgr->worldData;
        // (allocators and the print channel are stateless façades over the
        // ecs_state: their "address" is the state itself)
        c.hostAllocatorAddr = mgr;
        c.hostPrintAddr = mgr->hostPrintRing;
        c.tmpAllocatorAddr = mgr;
        c.numWorldDataBytes = mgr->worldDataStride;
        c.numWorlds = (uint32_t)mgr->numWorlds;
        // (until the ray caster's BLAS exists: one object-space root AABB per
        // object id, madrona/render/ecs.hpp)
        c.meshBVHsAddr = mgr->moduleData[2];
        c.bvhInternalData = mgr->moduleData[3];
        c.raycastOutputResolution = mgr->raycastOutputResolution;
        c.raycastRGBD = mgr->raycastRGBD;
#endif
        return c;
    }
};

}
}

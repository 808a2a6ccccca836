// Oracle builds only (SIM_BACKEND_REF_CPU): what the reference's Vulkan renderer
// would own -- the buffers its CPU-mode render-prep systems append instance /
// view records to (reference src/render/ecs_interop.hpp, filled by
// src/render/ecs_system.cpp:100-159, 211-314) -- owned by the simulator's
// manager instead, plus the C entry the tests read them through.
#pragma once

#ifdef SIM_BACKEND_REF_CPU
// the reference's private bridge struct, where it lies
#include "ecs_interop.hpp"
#include <madrona/sync.hpp>

#include <cstring>
#include <vector>

namespace simmgr {

struct CpuRenderBridge {
    madrona::render::RenderECSBridge bridge {};
    std::vector<madrona::render::PerspectiveCameraData> views;
    std::vector<madrona::render::InstanceData> instances;
    std::vector<uint64_t> instanceKeys, viewKeys;
    uint32_t totalViews = 0, totalInstances = 0;
    madrona::AtomicU32 viewCounter { 0 };
    madrona::AtomicU32 instanceCounter { 0 };

    CpuRenderBridge(uint32_t num_worlds, uint32_t max_records_per_world,
                    uint32_t resolution, uint32_t max_views_per_world,
                    uint32_t max_instances_per_world)
    {
        const size_t cap = (size_t)num_worlds * max_records_per_world;
        views.resize(cap);
        instances.resize(cap);
        instanceKeys.resize(cap);
        viewKeys.resize(cap);
        bridge.views = views.data();
        bridge.instances = instances.data();
        bridge.totalNumViews = &totalViews;
        bridge.totalNumInstances = &totalInstances;
        bridge.totalNumViewsCPUInc = &viewCounter;
        bridge.totalNumInstancesCPUInc = &instanceCounter;
        bridge.instancesWorldIDs = instanceKeys.data();
        bridge.viewsWorldIDs = viewKeys.data();
        bridge.renderWidth = (int32_t)resolution;
        bridge.renderHeight = (int32_t)resolution;
        bridge.maxViewsPerworld = max_views_per_world;
        bridge.maxInstancesPerWorld = max_instances_per_world;
        bridge.isGPUBackend = false;
    }

    // the renderer zeroes the append counters before every step
    void beginStep()
    {
        viewCounter.store_relaxed(0);
        instanceCounter.store_relaxed(0);
    }

    // kind 0 = instances (64 B each), 1 = views (48 B each), in arrival order,
    // + the (world << 32 | entity id) key of each.  Returns the count.
    int64_t records(int32_t kind, void *dst, uint64_t *keys_dst,
                    uint64_t max_records)
    {
        const uint64_t n = kind == 0 ? instanceCounter.load_relaxed() :
                                       viewCounter.load_relaxed();
        if (n > max_records) return -2;
        if (kind == 0) {
            memcpy(dst, instances.data(), n * 64);
            memcpy(keys_dst, instanceKeys.data(), n * 8);
        } else {
            memcpy(dst, views.data(), n * 48);
            memcpy(keys_dst, viewKeys.data(), n * 8);
        }
        return (int64_t)n;
    }
};

}
#endif

// Device entry points of a simulator.
// API contract: reference src/mw/device/include/madrona/mw_gpu_entry.hpp:12-91
// -- the simulator's .cpp ends with
//     MADRONA_BUILD_MWGPU_ENTRY(ContextT, WorldT, ConfigT, InitT);
// In the reference that instantiates three NVRTC-compiled kernels (initECS,
// initWorlds, initTasks).  Here it emits ONE exported C symbol,
// madronaMWHipUserEntry(), returning the mwhip_user_entry table the executor
// needs: host thunks for registerTypes / setupTasks (both run on the host in
// this backend) and the host stub of the world-constructor kernel.
#pragma once

#include <madrona/taskgraph.hpp>
#include <madrona/registry.hpp>

#include <mwhip.h>

namespace madrona {
namespace mwGPU {

#if defined(__HIPCC__)

namespace entryKernels {

// One thread per world: placement-new the simulator's per-world object, which
// creates the world's initial entities through the device Context
// (== entryKernels::initWorlds, reference mw_gpu_entry.hpp:37-56).
template <typename ContextT, typename WorldT, typename ConfigT, typename InitT>
__global__ void __launch_bounds__(64)
initWorlds(mwhip::EcsState *S, const void *cfg, const void *user_inits,
           int32_t num_worlds)
{
    int32_t world_idx = (int32_t)(threadIdx.x + blockDim.x * blockIdx.x);
    if (world_idx >= num_worlds) {
        return;
    }

    StateManager *state_mgr = static_cast<StateManager *>(S);
    WorldBase *world = TaskGraph::getWorld(state_mgr, world_idx);
    ContextT ctx = TaskGraph::makeContext<ContextT>(
        state_mgr, WorldID { world_idx }, /* exclusive_world = */ true);

    new (world) WorldT(ctx, *(const ConfigT *)cfg,
                       ((const InitT *)user_inits)[world_idx]);
}

}

template <typename ContextT, typename WorldT, typename ConfigT, typename InitT>
struct MWHipEntry {
    static void registerTypes(mwhip_exec *exec, const void *cfg)
    {
        StateManager host_mgr {};
        host_mgr.hostExec = exec;

        // Entity is component 0 and WorldID component 1 in every table
        // (reference src/mw/device/state.cpp:150-151)
        host_mgr.registerComponent<Entity>();
        host_mgr.registerComponent<WorldID>();

        ECSRegistry registry(&host_mgr, nullptr);
        WorldT::registerTypes(registry, *(const ConfigT *)cfg);
    }

    static void setupTasks(mwhip_exec *exec, const void *cfg)
    {
        StateManager host_mgr {};
        host_mgr.hostExec = exec;

        TaskGraphManager mgr(exec, &host_mgr, mwhip_num_task_graphs(exec));
        WorldT::setupTasks(mgr, *(const ConfigT *)cfg);
        mgr.constructGraphs();
    }

    static void bindDeviceState(void *state_dev)
    {
        hipError_t res = hipMemcpyToSymbol(
            HIP_SYMBOL(mwGPU::deviceStateManager), &state_dev, sizeof(void *));
        if (res != hipSuccess) {
            fprintf(stderr, "madrona_amd: binding device state failed: %s\n",
                    hipGetErrorString(res));
            abort();
        }
    }

    static const mwhip_user_entry *get()
    {
        static const mwhip_user_entry entry {
            MWHIP_ABI_VERSION,
            &registerTypes,
            &setupTasks,
            (const void *)&entryKernels::initWorlds<
                ContextT, WorldT, ConfigT, InitT>,
            &bindDeviceState,
        };
        return &entry;
    }
};

#endif

}
}

// The function below may itself be compiled host+device (the build wraps the
// simulator's sources in `#pragma clang force_cuda_host_device`), so the
// table is fetched through a __host__ lambda: that is legal in both passes and
// makes the device pass instantiate (and emit) the initWorlds kernel.
#define MADRONA_BUILD_MWGPU_ENTRY(ContextT, WorldT, ConfigT, InitT) \
    extern "C" MADRONA_EXPORT const mwhip_user_entry *madronaMWHipUserEntry() \
    { \
        [[maybe_unused]] auto get_entry = [] MADRONA_HOST_LAMBDA () { \
            return ::madrona::mwGPU::MWHipEntry< \
                ContextT, WorldT, ConfigT, InitT>::get(); \
        }; \
        MADRONA_ENTRY_RETURN(get_entry) \
    }

#if defined(__HIP_DEVICE_COMPILE__)
#define MADRONA_ENTRY_RETURN(fn) return nullptr;
#else
#define MADRONA_ENTRY_RETURN(fn) return fn();
#endif

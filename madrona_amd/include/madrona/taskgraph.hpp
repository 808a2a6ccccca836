// Task graph construction API + the per-node HIP kernels it instantiates.
//
// API contract: reference src/mw/device/include/madrona/taskgraph.hpp:28-420
// (the GPU-mode flavour: TaskGraph::Builder, NodeID{int32_t}, addNodeFn /
// addOneOffNode / addDynamicCountNode, CustomParallelForNode, ParallelForNode,
// ClearTmpNode, RecycleEntitiesNode, ResetTmpAllocNode, SortArchetypeNode,
// CompactArchetypeNode) -- simulators' setupTasks compile unchanged.
//
// MI355X design: there is no persistent megakernel
// (reference src/mw/device/taskgraph.cpp:142-317).  setupTasks runs on the
// HOST; every node becomes one specialised __global__ kernel (instantiated
// right here, in the simulator's own HIP translation unit, so the user's
// system function is inlined into it) or one runtime-provided kernel chain
// (sort).  The builder hands kernel stubs + node data to libmadrona_hip.so
// through the C ABI, which orders them with the reference's topo-sort rule and
// captures them into a hipGraph.  Kernels use a fixed grid and read their trip
// count (table row counts) from device memory with grid-stride loops, so the
// graph never needs a host round trip (reference: numDynamicInvocations).
#pragma once

#include <madrona/context.hpp>
#include <madrona/optional.hpp>
#include <madrona/query.hpp>
#include <madrona/span.hpp>
#include <madrona/state.hpp>
#include <madrona/mw_gpu/tracing.hpp>

#include <mwhip.h>

#include <cstring>
#include <functional>
#include <memory>
#include <new>
#include <string>
#include <tuple>
#include <type_traits>
#include <vector>

namespace madrona {

class TaskGraph;
class TaskGraphManager;

struct NodeBase {
    // Kept for source compatibility with custom nodes: in the reference the
    // predecessor node writes the successor's invocation count here
    // (device taskgraph.hpp:28-32).  Kernels in this backend read counts
    // directly from the state, see CustomNodeKernel below.
    uint32_t numDynamicInvocations;
};

namespace mwGPU {

#if defined(__HIPCC__)
// == GPUImplConsts::get().stateManagerAddr (reference mw_gpu/const.hpp)
inline __device__ StateManager *deviceStateManager = nullptr;
#endif

MADRONA_HD inline StateManager *getStateManager()
{
#if defined(__HIP_DEVICE_COMPILE__)
    return deviceStateManager;
#else
    return nullptr;
#endif
}

#if defined(__HIPCC__)
// mw_gpu/tracing.hpp: no-ops unless MADRONA_TRACING, as in the reference
MADRONA_DEVICE inline void DeviceTracing::resetIndex()
{
#ifdef MADRONA_TRACING
    DeviceTracing *t = get(getStateManager());
    if (t != nullptr) {
        __hip_atomic_store(&t->cur_index_, 0, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
#endif
}

MADRONA_DEVICE inline void DeviceTracing::Log([[maybe_unused]] DeviceEvent event,
                                              [[maybe_unused]] uint32_t func_id,
                                              [[maybe_unused]] uint32_t num_invocations,
                                              [[maybe_unused]] uint32_t node_id)
{
#ifdef MADRONA_TRACING
    LogTo(getStateManager(), event, func_id, num_invocations, node_id,
          threadIdx.x == 0);
#endif
}

MADRONA_DEVICE inline void DeviceTracing::Log([[maybe_unused]] DeviceEvent event,
                                              [[maybe_unused]] uint32_t func_id,
                                              [[maybe_unused]] uint32_t num_invocations,
                                              [[maybe_unused]] uint32_t node_id,
                                              [[maybe_unused]] bool is_leader)
{
#ifdef MADRONA_TRACING
    LogTo(getStateManager(), event, func_id, num_invocations, node_id, is_leader);
#endif
}
#endif

}

class TaskGraph {
public:
    // (reference: 256.  Node data is what a kernel reaches with ONE load from
    // its arguments; the physics step keeps a table of addresses there)
    static inline constexpr uint32_t maxNodeDataBytes = 2048;

    struct alignas(64) NodeData {
        char userData[maxNodeDataBytes];
    };

    struct NodeID {
        int32_t id;
    };

    struct DataID {
        int32_t id;
    };

    template <typename NodeT>
    struct TypedDataID : DataID {};

    class Builder {
    public:
        inline Builder(mwhip_exec *exec, StateManager *state_mgr,
                       uint32_t taskgraph_id);

        template <typename NodeT, typename... Args>
        MADRONA_HOST_API TypedDataID<NodeT> constructNodeData(Args &&...args);

        template <auto fn, typename NodeT>
        MADRONA_HOST_API NodeID addNodeFn(TypedDataID<NodeT> data,
                         Span<const NodeID> dependencies,
                         Optional<NodeID> parent_node =
                             Optional<NodeID>::none(),
                         uint32_t fixed_num_invocations = 0,
                         uint32_t num_threads_per_invocation = 1);

        template <typename NodeT, int32_t count = 1, typename... Args>
        MADRONA_HOST_API NodeID addOneOffNode(Span<const NodeID> dependencies,
                                              Args &&...args);

        template <typename NodeT, typename... Args>
        MADRONA_HOST_API NodeID addDynamicCountNode(Span<const NodeID> dependencies,
                                   uint32_t num_threads_per_invocation,
                                   Args &&...args);

        template <typename NodeT>
        MADRONA_HOST_API inline NodeID addToGraph(
            Span<const NodeID> dependencies);

        template <typename NodeT>
        MADRONA_HOST_API NodeT &getDataRef(TypedDataID<NodeT> data_id);

        MADRONA_HD inline uint32_t getTaskgraphID() const { return taskgraph_id_; }
        MADRONA_HD inline StateManager &stateManager() { return *state_mgr_; }
        MADRONA_HD inline mwhip_exec *exec() const { return exec_; }

        // Backend-facing: stage a fully described node.
        inline NodeID addRuntimeNode(const mwhip_node_desc &desc,
                                     int32_t staged_data_idx,
                                     Span<const NodeID> dependencies);

        // Uploads node data blocks and registers all staged nodes.
        inline void flush();

    private:
        struct StagedNode {
            mwhip_node_desc desc;
            std::string name;
            int32_t dataIdx;
            std::vector<int32_t> deps;
        };

        mwhip_exec *exec_;
        StateManager *state_mgr_;
        uint32_t taskgraph_id_;
        std::vector<std::unique_ptr<NodeData>> node_datas_;
        std::vector<uint32_t> node_data_bytes_;
        std::vector<StagedNode> staged_;
    };

    MADRONA_HD static inline WorldBase *getWorld(StateManager *state_mgr,
                                                     int32_t world_idx)
    {
        return (WorldBase *)(mwhip::loadInvariant(&state_mgr->worldData) +
            (uint64_t)world_idx *
                mwhip::loadInvariant(&state_mgr->worldDataStride));
    }

    template <typename ContextT>
    MADRONA_HD static inline ContextT makeContext(StateManager *state_mgr,
                                                  WorldID world_id,
                                                  bool exclusive_world = false)
    {
        using WorldDataT = typename WorldTypeExtract<ContextT>::type;
        return ContextT((WorldDataT *)getWorld(state_mgr, world_id.idx),
                        WorkerInit { world_id, state_mgr, exclusive_world });
    }

    template <typename ContextT>
    MADRONA_HD static inline ContextT makeContext(WorldID world_id)
    {
        return makeContext<ContextT>(mwGPU::getStateManager(), world_id);
    }

private:
    template <typename ContextT, bool = false>
    struct WorldTypeExtract {
        using type = typename ContextT::WorldDataT;
    };

    template <bool ignore>
    struct WorldTypeExtract<Context, ignore> {
        using type = WorldBase;
    };
};

using TaskGraphNodeID = TaskGraph::NodeID;
using TaskGraphBuilder = TaskGraph::Builder;

class TaskGraphManager {
public:
    inline TaskGraphManager(mwhip_exec *exec, StateManager *state_mgr,
                            uint32_t num_taskgraphs);

    template <EnumType EnumT>
    MADRONA_HOST_API TaskGraphBuilder &init(EnumT taskgraph_id)
    {
        return init((uint32_t)taskgraph_id);
    }

    MADRONA_HOST_API inline TaskGraphBuilder &init(uint32_t taskgraph_id);

    // Called by the backend after the user's setupTasks returns.
    inline void constructGraphs();

private:
    // (only read in the host bodies of the MADRONA_HOST_API members)
    [[maybe_unused]] mwhip_exec *exec_;
    [[maybe_unused]] StateManager *state_mgr_;
    std::vector<std::unique_ptr<TaskGraphBuilder>> builders_;
};

// ---------------------------------------------------------------------------
// Built-in nodes
// ---------------------------------------------------------------------------

// Calls Fn(ctx, components...) once per entity that has all ComponentTs.
// threads_per_invocation / items_per_invocation other than 1/1 select the
// reference's "N items per invocation" calling convention
// (Fn(WorldID *, ComponentTs *..., int32_t n), device taskgraph.inl:229-266).
template <typename ContextT, auto Fn,
          int32_t threads_per_invocation,
          int32_t items_per_invocation,
          typename... ComponentTs>
class CustomParallelForNode : public NodeBase {
public:
    MADRONA_HOST_API static TaskGraph::NodeID addToGraph(
        TaskGraph::Builder &builder,
        Span<const TaskGraph::NodeID> dependencies);
};

template <typename ContextT, auto Fn, typename... ComponentTs>
using ParallelForNode =
    CustomParallelForNode<ContextT, Fn, 1, 1, ComponentTs...>;

struct ClearTmpNodeBase : NodeBase {
    MADRONA_HOST_API static inline TaskGraph::NodeID addToGraph(
        TaskGraph::Builder &builder,
        Span<const TaskGraph::NodeID> dependencies,
        uint32_t archetype_id);
};

template <typename ArchetypeT>
struct ClearTmpNode : ClearTmpNodeBase {
    MADRONA_HOST_API static TaskGraph::NodeID addToGraph(
        TaskGraph::Builder &builder,
        Span<const TaskGraph::NodeID> dependencies)
    {
        return ClearTmpNodeBase::addToGraph(builder, dependencies,
            TypeTracker::typeID<ArchetypeT>());
    }
};

// Entity ids return to their world's cache at destroy time in this backend
// (see mwhip/ecs_state.hpp), so there is nothing left to recycle; the node is
// accepted and ordered like any other but launches no kernel.
struct RecycleEntitiesNode : NodeBase {
    MADRONA_HOST_API static inline TaskGraph::NodeID addToGraph(
        TaskGraph::Builder &builder,
        Span<const TaskGraph::NodeID> dependencies);
};

struct ResetTmpAllocNode : NodeBase {
    MADRONA_HOST_API static inline TaskGraph::NodeID addToGraph(
        TaskGraph::Builder &builder,
        Span<const TaskGraph::NodeID> dependencies);
};

struct SortArchetypeNodeBase : NodeBase {
    MADRONA_HOST_API static inline TaskGraph::NodeID addToGraph(
        TaskGraph::Builder &builder,
        Span<const TaskGraph::NodeID> dependencies,
        uint32_t archetype_id,
        int32_t component_id);
};

template <typename ArchetypeT, typename ComponentT>
struct SortArchetypeNode : SortArchetypeNodeBase {
    MADRONA_HOST_API static TaskGraph::NodeID addToGraph(
        TaskGraph::Builder &builder,
        Span<const TaskGraph::NodeID> dependencies)
    {
        return SortArchetypeNodeBase::addToGraph(builder, dependencies,
            TypeTracker::typeID<ArchetypeT>(),
            (int32_t)TypeTracker::typeID<ComponentT>());
    }
};

// == SortArchetypeNode<ArchetypeT, WorldID> + ResetTmpAllocNode
// (reference device taskgraph.inl:321-332)
template <typename ArchetypeT>
struct CompactArchetypeNode {
    MADRONA_HOST_API static TaskGraph::NodeID addToGraph(
        TaskGraph::Builder &builder,
        Span<const TaskGraph::NodeID> dependencies)
    {
        auto sort_sys = builder.addToGraph<
            SortArchetypeNode<ArchetypeT, WorldID>>(dependencies);
        return builder.addToGraph<ResetTmpAllocNode>({sort_sys});
    }
};

}

#include "taskgraph.inl"

// mwGPU::DeviceTracing -- the device-side event log of the reference
// (src/mw/device/include/madrona/mw_gpu/tracing.hpp:14-130; host side
// src/mw/cuda_exec.cpp:204-257, parser scripts/parse_device_tracing.py).
//
// Same record (40-byte DeviceLog), same event codes, same on/off switch
// (compiled in only with -DMADRONA_TRACING=1, on the simulator's device TU and
// on the runtime: `make -C madrona_amd TRACING=1 OUT=_build_tracing`), same file
// (<dir><pid or $MADRONA_MWGPU_TRACE_NAME>_madrona_device_tracing.bin, written
// when the executor is destroyed, the first 100 steps).  User code may log its
// own events with DeviceTracing::Log(...) as in the reference.
//
// What the events mean on this backend (one kernel per node instead of one
// persistent megakernel whose blocks pull chunks of nodes):
//   calibration  first record of a step: funcID = waves per 256-thread
//                workgroup, numInvocations = 0, nodeID = compute units
//   nodeStart    a kernel of the step graph is about to start (logged by a
//                one-thread marker launch in front of it): funcID = index into
//                the kernel-name table (written next to the log, ..._device_
//                tracing_nodes.txt), numInvocations = threads launched, nodeID =
//                position of the kernel in the graph
//   blockStart / blockWait   first / last instruction of every workgroup of that
//                kernel (thread 0): numInvocations = linear workgroup index,
//                smID = (XCC << 8 | SE << 5 | SH << 4 | CU), the unit it ran on;
//                written to slots the marker reserved (no atomics)
//   nodeFinish   synthesised on the host when the file is written: the latest
//                blockWait of the node (a "last workgroup" counter would
//                serialise the workgroups of a big grid on one atomic)
//   blockExit    end of the step (the health kernel)
// cycleCount is the 100 MHz constant clock (s_memrealtime) in ns.
#pragma once

#include <madrona/macros.hpp>
#include <madrona/mwhip/ecs_state.hpp>

#include <cstdint>

namespace madrona {
namespace mwGPU {

class DeviceTracingManager;

enum class DeviceEvent : uint32_t {
    calibration = 0,
    nodeStart = 1,
    nodeFinish = 2,
    blockStart = 3,
    blockWait = 4,
    blockExit = 5,
};

class DeviceTracing {
public:
    // 1M events per step; a step that overflows is dropped (index = -1)
    static constexpr inline uint64_t maxLogSize = 1000000;

    struct DeviceLog {
        DeviceEvent event;
        uint32_t funcID;
        uint32_t numInvocations;
        uint32_t nodeID;
        uint32_t warpID;
        uint32_t blockID;
        uint32_t smID;
        uint32_t padding;       // the record's index in its step
        uint64_t cycleCount;
    };

    int32_t cur_index_;
    DeviceLog device_logs_[maxLogSize];

    inline int32_t getIndex() const { return cur_index_; }

#if defined(__HIPCC__)
    // The kernel being traced: its position in the graph, its kernel-name
    // index, and the log slots the marker launch in front of it reserved for
    // its workgroups -- two each, so that no workgroup touches the log's one
    // index (an atomic per workgroup on one address costs 11 ns, serialised:
    // 20 K records turned a 150 us step into 630 us).
    struct alignas(16) Cursor {
        uint32_t nodeID;
        uint32_t funcID;
        uint32_t firstSlot;
        uint32_t numWorkgroups;
    };
    static constexpr uint32_t unusedSlot = 0xFFFFFFFFu;     // DeviceLog::event

    static MADRONA_DEVICE inline DeviceTracing *get(mwhip::EcsState *S)
    {
        return (DeviceTracing *)S->deviceTracing;
    }

    // the reference's entry points for user code; defined in
    // madrona/taskgraph.hpp (they find the state through getStateManager())
    static MADRONA_DEVICE inline void resetIndex();
    static MADRONA_DEVICE inline void Log(DeviceEvent event, uint32_t func_id,
                                          uint32_t num_invocations, uint32_t node_id);
    static MADRONA_DEVICE inline void Log(DeviceEvent event, uint32_t func_id,
                                          uint32_t num_invocations, uint32_t node_id,
                                          bool is_leader);

    // (engine kernels pass the state they were launched with)
    static MADRONA_DEVICE inline void LogTo(mwhip::EcsState *S, DeviceEvent event,
                                            uint32_t func_id,
                                            uint32_t num_invocations,
                                            uint32_t node_id, bool is_leader)
    {
        DeviceTracing *t = S != nullptr ? get(S) : nullptr;
        if (!is_leader || t == nullptr) {
            return;
        }
        if (__hip_atomic_load(&t->cur_index_, __ATOMIC_RELAXED,
                              __HIP_MEMORY_SCOPE_AGENT) < 0) {
            return;
        }
        const uint32_t log_index = (uint32_t)__hip_atomic_fetch_add(
            &t->cur_index_, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (log_index >= maxLogSize) {
            // this step's trace is incomplete: dropped by the host
            __hip_atomic_store(&t->cur_index_, -1, __ATOMIC_RELEASE,
                               __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        t->device_logs_[log_index] = DeviceLog {
            event, func_id, num_invocations, node_id,
            (uint32_t)(threadIdx.x / 64u),
            (uint32_t)(blockIdx.x + blockIdx.y * gridDim.x),
            computeUnitID(), log_index, globalTimer(),
        };
    }

    static MADRONA_DEVICE inline uint64_t globalTimer()
    {
        return (uint64_t)wall_clock64() * 10ull;    // 100 MHz -> ns
    }

    // HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID: [3:0]
    static MADRONA_DEVICE inline uint32_t computeUnitID()
    {
        const uint32_t hw = __builtin_amdgcn_s_getreg(4 | (8 << 6) | (7 << 11));
        const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
        return (xcc << 8) | hw;
    }

#endif

    friend class DeviceTracingManager;
};

static_assert(sizeof(DeviceTracing::DeviceLog) == 40);

}

namespace mwhip {

// First and last thing a traced kernel does: `TraceScope trace(S);` (the
// destructor also runs on early returns).  Empty unless MADRONA_TRACING.
struct TraceScope {
#if defined(MADRONA_TRACING) && defined(__HIPCC__)
    EcsState *S;

    MADRONA_DEVICE inline explicit TraceScope(EcsState *s)
        : S(s)
    {
        log(mwGPU::DeviceEvent::blockStart);
    }

    MADRONA_DEVICE inline ~TraceScope()
    {
        log(mwGPU::DeviceEvent::blockWait);
    }

    TraceScope(const TraceScope &) = delete;

private:
    MADRONA_DEVICE inline void log(mwGPU::DeviceEvent event)
    {
        using mwGPU::DeviceTracing;
        if (threadIdx.x != 0 || threadIdx.y != 0 || S->deviceTracing == nullptr) {
            return;
        }
        // (plain loads: the marker kernel finished before this one started)
        const DeviceTracing::Cursor cur = *(const DeviceTracing::Cursor *)S->traceCursor;
        const uint32_t wg = blockIdx.x + blockIdx.y * gridDim.x +
            blockIdx.z * gridDim.x * gridDim.y;
        if (wg >= cur.numWorkgroups) {
            return;
        }
        const uint32_t slot = cur.firstSlot + 2u * wg +
            (event == mwGPU::DeviceEvent::blockWait ? 1u : 0u);
        DeviceTracing::get(S)->device_logs_[slot] = DeviceTracing::DeviceLog {
            event, cur.funcID, wg, cur.nodeID, 0u, wg,
            DeviceTracing::computeUnitID(), slot, DeviceTracing::globalTimer(),
        };
    }
#else
    MADRONA_HD inline explicit TraceScope(EcsState *) {}
#endif
};

}
}

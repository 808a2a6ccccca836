// libmadrona_hip.so -- host runtime of the MI355X many-world ECS backend.
// Implements the C ABI declared in include/mwhip.h.
//
// Replaces, for the hot path only, the reference's GPU executor host side
// (src/mw/cuda_exec.cpp: state allocation :1721-1948, graph build :2174-2292,
// run :2756-2794) and the device-side registry / task-graph builder
// (src/mw/device/state.cpp:154-440, taskgraph_utils.cpp:30-146).  There is no
// runtime compiler, no megakernel and no host<->device mailbox: registration
// and graph construction are host code, every node is its own kernel, and a
// step is one hipGraph replay on the executor's private stream.
#include "exec_internal.hpp"

// ---- this translation unit: the small device kernels the runtime owns ----

namespace {

// ClearTmpNode / ResetTmpAllocNode (reference taskgraph_utils.cpp:171-230):
// a handful of scalar stores; consecutive ones share one launch -- or ride on
// the last kernel of the sort chain they follow (sort_archetype.hip).
__global__ void miscOpsKernel(EcsState *S, const MiscOp *ops, uint32_t num_ops)
{
    TraceScope trace_scope(S);
    applyMiscOps(S, ops, num_ops, threadIdx.x);
}

// ---- exclusive scan over a few device arrays (MWHIP_NODE_EXCLUSIVE_SCAN) ----
// Single pass, chained through 8-byte {epoch tag | status | value} granules
// like the sort's look-back (relaxed agent-scope atomics, ticketed tiles).
__global__ void __launch_bounds__(kScanThreads)
exclusiveScanKernel(EcsState *S, const ScanNode *node_ptr)
{
    TraceScope trace_scope(S);
    const ScanNode &node = *node_ptr;
    const mwhip_scan_params &p = node.params;

    __shared__ uint32_t lds_tile;
    __shared__ uint32_t lds_wave[kScanThreads / 64];
    __shared__ uint32_t lds_prefix;

    // segment layout: tiles never straddle segments
    int32_t seg_len[MWHIP_SCAN_MAX_SEGMENTS];
    uint32_t seg_tile_start[MWHIP_SCAN_MAX_SEGMENTS + 1];
    uint32_t total_tiles = 0;
    for (uint32_t s = 0; s < MWHIP_SCAN_MAX_SEGMENTS; s++) {
        int32_t len = s < p.num_segments ? *p.lengths[s] : 0;
        seg_len[s] = len > 0 ? len : 0;
        seg_tile_start[s] = total_tiles;
        total_tiles += (uint32_t)((seg_len[s] + kScanTile - 1) / kScanTile);
    }
    seg_tile_start[MWHIP_SCAN_MAX_SEGMENTS] = total_tiles;

    if (threadIdx.x == 0) {
        lds_tile = atomicAdd(&node.state->ticket, 1u);
    }
    __syncthreads();
    const uint32_t tile = lds_tile;
    const uint32_t epoch = node.state->epoch;
    const uint32_t tag = epoch + 1u;

    if (tile < total_tiles) {
        uint32_t seg = 0;
        while (seg + 1 < MWHIP_SCAN_MAX_SEGMENTS && tile >= seg_tile_start[seg + 1]) {
            seg++;
        }
        const int32_t base = (int32_t)(tile - seg_tile_start[seg]) * kScanTile;
        uint32_t *data = p.data[seg];
        const int32_t len = seg_len[seg];

        // blocked arrangement: thread t owns items [t*8, t*8+8) of the tile
        uint32_t v[kScanItems];
        uint32_t thread_sum = 0;
#pragma unroll
        for (int j = 0; j < kScanItems; j++) {
            int32_t i = base + (int32_t)threadIdx.x * kScanItems + j;
            v[j] = i < len ? data[i] : 0u;
            thread_sum += v[j];
        }

        // block exclusive scan of thread sums
        const uint32_t lane = threadIdx.x & 63u;
        const uint32_t wave = threadIdx.x >> 6;
        uint32_t incl = thread_sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t up = __shfl_up(incl, d, 64);
            if ((int)lane >= d) incl += up;
        }
        if (lane == 63) lds_wave[wave] = incl;
        __syncthreads();
        uint32_t wave_base = 0, tile_total = 0;
#pragma unroll
        for (int w = 0; w < kScanThreads / 64; w++) {
            uint32_t ws = lds_wave[w];
            if (w < (int)wave) wave_base += ws;
            tile_total += ws;
        }

        // look back for the sum of all earlier tiles
        if (threadIdx.x == 0) {
            unsigned long long *g = node.granules;
            uint32_t exclusive = 0;
            if (tile == 0) {
                __hip_atomic_store(&g[0], ((unsigned long long)tag << 32) |
                    (2ull << 30) | tile_total, __ATOMIC_RELAXED,
                    __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __hip_atomic_store(&g[tile], ((unsigned long long)tag << 32) |
                    (1ull << 30) | tile_total, __ATOMIC_RELAXED,
                    __HIP_MEMORY_SCOPE_AGENT);
                int32_t look = (int32_t)tile - 1;
                uint32_t spins = 0;
                while (true) {
                    unsigned long long x = __hip_atomic_load(&g[look],
                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((uint32_t)(x >> 32) != tag) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > (1u << 26)) {
                            raiseError(S, kErrSortLookback);
                            break;
                        }
                        continue;
                    }
                    exclusive += (uint32_t)(x & ((1ull << 30) - 1ull));
                    if ((x >> 30) & 2ull) break;
                    look -= 1;
                }
                __hip_atomic_store(&g[tile], ((unsigned long long)tag << 32) |
                    (2ull << 30) | (unsigned long long)(exclusive + tile_total),
                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            lds_prefix = exclusive;

            if (tile == total_tiles - 1) {
                uint32_t total = exclusive + tile_total;
                if (total > p.capacity) {
                    raiseError(S, kErrTableOverflow);
                    total = p.capacity;
                }
                *p.total_out = (int32_t)total;
                if (p.needs_sort_out != nullptr && total > 0) {
                    *p.needs_sort_out = 1u;
                }
            }
        }
        __syncthreads();

        uint32_t running = lds_prefix + wave_base + incl - thread_sum;
#pragma unroll
        for (int j = 0; j < kScanItems; j++) {
            int32_t i = base + (int32_t)threadIdx.x * kScanItems + j;
            if (i < len) {
                data[i] = running;
            }
            running += v[j];
        }
    } else if (total_tiles == 0 && tile == 0 && threadIdx.x == 0) {
        *p.total_out = 0;
    }

    // last block resets the ticket and advances the epoch for the next launch
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        uint32_t done = atomicAdd(&node.state->arrivals, 1u);
        if (done == gridDim.x - 1) {
            node.state->arrivals = 0;
            node.state->ticket = 0;
            node.state->epoch = epoch + 1u;
        }
    }
}

// Holds the stream until the host flips a flag in pinned memory, so that a
// whole step's kernels + timing events can be queued behind it and then run
// back to back on the device (per-kernel event deltas would otherwise mostly
// measure the host's launch rate).
__global__ void gateKernel(int32_t *host_flag)
{
    if (threadIdx.x != 0) return;
    for (uint32_t spins = 0; spins < (1u << 22); spins++) {
        if (__hip_atomic_load(host_flag, __ATOMIC_RELAXED,
                              __HIP_MEMORY_SCOPE_SYSTEM) != 0) {
            break;
        }
        __builtin_amdgcn_s_sleep(32);
    }
}

// Brackets a measurement window in a kernel trace: profiles/summarize_rocprof.py
// keeps the dispatches between the first and the last launch of this kernel
// (mwhip_mark_window), so the committed rocprofv3 averages cover exactly the
// steps bench.py timed.
__global__ void benchWindowMarker(uint32_t *signal, uint32_t id)
{
    if (threadIdx.x == 0 && signal != nullptr && id == 0xFFFFFFFFu) {
        *signal = id;   // never taken: keeps the arguments alive
    }
}

// End-of-graph health record written straight into pinned host memory.
// One thread per output word (consecutive lanes -> consecutive words of a
// record: coalesced stores; a column's words of one row are contiguous loads).
__global__ void __launch_bounds__(256)
packRowsKernel(PackArgs args, uint32_t *dst)
{
    const uint64_t total = (uint64_t)args.numRows * args.recordWords;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
         i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t row = (uint32_t)(i / args.recordWords);
        const uint32_t w = (uint32_t)(i % args.recordWords);
        uint32_t c = 0;
#pragma unroll
        for (uint32_t k = 1; k < MWHIP_PACK_MAX_COLUMNS; k++) {
            if (k < args.numColumns && w >= args.firstWord[k]) {
                c = k;
            }
        }
        dst[i] = args.src[c][(uint64_t)row * args.words[c] +
                             (w - args.firstWord[c])];
    }
}

// First kernel of a step graph with an input ring (mwhip_set_input_ring): slot
// (step-graph replays since the ring was set) % num_slots of a device-resident
// ring -> an exported column, i.e. a new set of actions every step without the
// host touching the executor's stream between two graph launches.  Render
// graphs neither read nor advance the rings (they bump word 0 only), so a
// render replay between two steps does not skip a slot.
__global__ void __launch_bounds__(256)
inputRingKernel(EcsState *S, uint32_t *dst, const uint32_t *ring,
                uint32_t slot_words, uint32_t num_slots, uint32_t first_replay)
{
    TraceScope trace_scope(S);
    const uint32_t replay = __hip_atomic_load(S->replayCounter + kStepReplayWord,
                                              __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t *src =
        ring + (size_t)((replay - first_replay) % num_slots) * slot_words;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < slot_words;
         i += gridDim.x * blockDim.x) {
        dst[i] = src[i];
    }
}

#ifdef MADRONA_TRACING
// One thread in front of every kernel of a traced graph (mw_gpu/tracing.hpp):
// calibration starts a step's log, nodeStart names the kernel whose
// workgroups log next, blockExit ends the step.
__global__ void __launch_bounds__(256)
traceMarkKernel(EcsState *S, uint32_t event, uint32_t node_id, uint32_t func_id,
                uint32_t invocations, uint32_t workgroups)
{
    using mwGPU::DeviceEvent;
    using mwGPU::DeviceTracing;
    DeviceTracing *t = (DeviceTracing *)S->deviceTracing;
    if (t == nullptr) return;
    __shared__ uint32_t first_slot;
    if (threadIdx.x == 0) {
        first_slot = DeviceTracing::unusedSlot;
        if ((DeviceEvent)event == DeviceEvent::calibration) {
            __hip_atomic_store(&t->cur_index_, 0, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        // this record + two per workgroup of the kernel that follows
        const uint32_t want = 1u + 2u * workgroups;
        int32_t base = __hip_atomic_load(&t->cur_index_, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT);
        if (base >= 0) {
            base = __hip_atomic_fetch_add(&t->cur_index_, (int32_t)want,
                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((uint64_t)base + want > DeviceTracing::maxLogSize) {
                // this step's trace is incomplete: dropped by the host
                __hip_atomic_store(&t->cur_index_, -1, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                base = -1;
            }
        }
        DeviceTracing::Cursor cur { node_id, func_id, 0u, 0u };
        if (base >= 0) {
            t->device_logs_[base] = DeviceTracing::DeviceLog {
                (DeviceEvent)event, func_id, invocations, node_id, 0u, 0u,
                DeviceTracing::computeUnitID(), (uint32_t)base,
                DeviceTracing::globalTimer(),
            };
            cur.firstSlot = (uint32_t)base + 1u;
            cur.numWorkgroups = workgroups;
            first_slot = cur.firstSlot;
        }
        *(DeviceTracing::Cursor *)S->traceCursor = cur;
    }
    __syncthreads();
    // (a kernel without a TraceScope leaves its slots like this)
    if (first_slot != DeviceTracing::unusedSlot) {
        for (uint32_t i = threadIdx.x; i < 2u * workgroups; i += blockDim.x) {
            t->device_logs_[first_slot + i].event = (DeviceEvent)DeviceTracing::unusedSlot;
        }
    }
}

#endif
// report_rows == 0 (render pass): error flags and the replay counter only -- the
// step's row statistics and high-water marks stay as its own health kernel
// reported them
__global__ void statsKernel(EcsState *S, int32_t *host_out,
                            uint32_t *replay_signal, uint32_t report_rows)
{
    TraceScope trace_scope(S);
    uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a < S->numArchetypeSlots && report_rows != 0u) {
        TableHdr &tbl = S->tables[a];
        host_out[kStatsRows + a] = tbl.registered ? tbl.numRows : -1;
        // the step's high-water mark (what growth is sized by), then reset
        host_out[kStatsPeaks + a] = tbl.registered ?
            (tbl.peakRows > tbl.numRows ? tbl.peakRows : tbl.numRows) : -1;
        tbl.peakRows = 0;
        // longest appended tail a compaction sort of the step met
        host_out[kStatsTails + a] = tbl.registered ? tbl.tailRows : 0;
        tbl.tailRows = 0;
    }
    if (a == 0) {
        host_out[0] = (int32_t)S->errorFlags;
        host_out[1] = S->numIds;
        // this replay is complete (mwhip_stream_wait_replays polls this; the
        // host reads the copy in pinned memory without waiting)
        if (report_rows != 0u) {
            // a step graph (not a render graph): the input rings move on
            __hip_atomic_fetch_add(replay_signal + kStepReplayWord, 1u,
                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        uint32_t done = __hip_atomic_fetch_add(replay_signal, 1u,
            __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM) + 1u;
        __hip_atomic_store((uint32_t *)&host_out[kStatsReplays], done,
                           __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}

namespace madrona {
namespace mwhip {

const void *miscOpsKernelFn() { return (const void *)&miscOpsKernel; }
const void *exclusiveScanKernelFn() { return (const void *)&exclusiveScanKernel; }
const void *gateKernelFn() { return (const void *)&gateKernel; }
const void *benchWindowMarkerFn() { return (const void *)&benchWindowMarker; }
const void *packRowsKernelFn() { return (const void *)&packRowsKernel; }
const void *inputRingKernelFn() { return (const void *)&inputRingKernel; }
#ifdef MADRONA_TRACING
const void *traceMarkKernelFn() { return (const void *)&traceMarkKernel; }
#endif
const void *statsKernelFn() { return (const void *)&statsKernel; }

}
}

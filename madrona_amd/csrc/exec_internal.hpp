// Host-side internals shared by the translation units of libmadrona_hip.so
// (runtime.hip: the C ABI; runtime_state.hip: device state, world construction,
// table growth; runtime_launch.hip: launch lists and step graphs;
// runtime_kernels.hip: the runtime's own small kernels).  Not installed.
#pragma once
#include "runtime_internal.hpp"
#include <madrona/tracing.hpp>
#include <cstddef>
#include "render_internal.hpp"

#include <hip/hip_ext.h>

#include <atomic>
#include <chrono>
#include <cinttypes>
#include <cstdarg>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <unordered_map>


// (shared by the translation units, not exported from the library)
#define MWHIP_RT __attribute__((visibility("hidden")))

// ---- errors (runtime.hip) -------------------------------------------------------
MWHIP_RT int fail(int code, const char *fmt, ...);

#define HIPCHK(expr) \
    do { \
        hipError_t hipchk_res_ = (expr); \
        if (hipchk_res_ != hipSuccess) { \
            return fail(-10, "%s -> %s (%s:%d)", #expr, \
                hipGetErrorString(hipchk_res_), __FILE__, __LINE__); \
        } \
    } while (0)

MWHIP_RT uint32_t envU32(const char *name, uint32_t fallback);

namespace madrona {
namespace mwhip {

// ---- what the runtime's own kernels share with the host (runtime_kernels.hip) ----
// exclusive scan over a few device arrays (MWHIP_NODE_EXCLUSIVE_SCAN)
constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;

struct ScanState {
    uint32_t ticket;
    uint32_t epoch;
    uint32_t arrivals;
    uint32_t pad;
};

struct ScanNode {
    mwhip_scan_params params;
    ScanState *state;
    unsigned long long *granules;
    uint32_t maxTiles;
};

struct PackArgs {
    const uint32_t *src[MWHIP_PACK_MAX_COLUMNS];
    uint32_t words[MWHIP_PACK_MAX_COLUMNS];     // per row, per column
    uint32_t firstWord[MWHIP_PACK_MAX_COLUMNS]; // of the column inside a record
    uint32_t numColumns;
    uint32_t recordWords;
    uint32_t numRows;
};

// layout of the pinned health record (int32 words)
constexpr uint32_t kStatsRows = 2;                          // [kMaxArchetypes]
constexpr uint32_t kStatsGate = 2 + kMaxArchetypes;         // profiling gate flag
constexpr uint32_t kStatsPeaks = 3 + kMaxArchetypes;        // [kMaxArchetypes]
constexpr uint32_t kStatsReplays = 3 + 2 * kMaxArchetypes;  // replays completed
constexpr uint32_t kStatsTails = 4 + 2 * kMaxArchetypes;    // [kMaxArchetypes]
constexpr uint32_t kStatsWords = 4 + 3 * kMaxArchetypes;

// word of the replay-signal block that counts completed replays of STEP graphs
// only (the ones that start with the input rings); word 0 counts every replay
constexpr uint32_t kStepReplayWord = 16;

// host stubs of those kernels (for KernelLaunch::fn / hipLaunchKernel)
MWHIP_RT const void *miscOpsKernelFn();
MWHIP_RT const void *exclusiveScanKernelFn();
MWHIP_RT const void *gateKernelFn();
MWHIP_RT const void *benchWindowMarkerFn();
MWHIP_RT const void *packRowsKernelFn();
MWHIP_RT const void *inputRingKernelFn();
#ifdef MADRONA_TRACING
MWHIP_RT const void *traceMarkKernelFn();
#endif
MWHIP_RT const void *statsKernelFn();

}
}

using namespace madrona;
using namespace madrona::mwhip;

// ---------------------------------------------------------------------------
// executor
// ---------------------------------------------------------------------------
struct ComponentRec {
    bool registered = false;
    uint32_t alignment = 0;
    uint32_t bytes = 0;
};

// Growable device memory: address space reserved up front
// (hipMemAddressReserve), backed 2 MiB at a time (hipMemCreate / hipMemMap), so
// a table can grow without any pointer into it changing -- the reference's GPU
// backend does the same through its host allocator thread (memory.cpp:20-178,
// cuda_exec.cpp:1603-1719).  One handle per 2 MiB chunk: the granule on which
// map + set-access behaved on this ROCm (larger or mixed chunk sizes returned
// hipErrorInvalidValue from hipMemSetAccess).
struct VmRange {
    char *base = nullptr;
    size_t reserved = 0;
    size_t mapped = 0;
    std::vector<hipMemGenericAllocationHandle_t> chunks;
};
static constexpr size_t kVmChunk = (size_t)2 << 20;

struct ArchetypeRec {
    bool registered = false;
    uint32_t id = 0;
    std::vector<uint32_t> comps;        // flattened user components
    uint32_t flags = 0;
    uint32_t maxPerWorld = 0;
    bool singleton = false;
    bool bigSort = false;
    uint32_t smallBusy = 0;     // consecutive reports of a busy one-launch sort           // outgrew the single-launch sort once
    // world sorts of this table take the compaction chain unless something
    // other than world sorts reorders / truncates it (a sort by another key,
    // ClearTmp, a scan node writing its row count: scrambled), or its appended
    // tails keep outgrowing what one workgroup sorts (noCompact)
    bool scrambled = false;
    bool noCompact = false;
    uint32_t longTails = 0;         // steps whose tail exceeded the limit
    int64_t peakSeen = 0;           // largest per-step peak reported so far
    uint32_t fillingUntil = 0;      // replay count until which the queue is kept short
    int32_t singletonOrdinal = -1;
    uint32_t capacity = 0;              // rows backed by memory right now
    uint32_t reservedCapacity = 0;      // rows the address space allows
    uint32_t numColumns = 0;
    uint32_t rowBytes = 0;
    std::vector<void *> primary;
    std::vector<void *> alt;
    // growable archetypes: the ranges behind primary / alt / sort buffers
    std::vector<VmRange *> primaryVm, altVm;
    VmRange *sortVm[5] = { nullptr, nullptr, nullptr, nullptr, nullptr };
    std::vector<uint32_t> colBytes;
    std::vector<uint32_t> colFlags;
    std::vector<uint32_t> colComponent;
    int32_t *worldOffsets = nullptr;
    int32_t *worldCounts = nullptr;
    // sort scratch (allocated on first use)
    SortState *sortState = nullptr;
    uint32_t *keysA = nullptr, *keysB = nullptr;
    int32_t *idxA = nullptr, *idxB = nullptr;
    unsigned long long *lookback = nullptr;
    int32_t *tileCounts = nullptr;      // compaction chain, per prefix tile
    int32_t *tileTailStart = nullptr;
    int32_t *tailLand = nullptr;        // [capacity] where each sorted tail row lands
};

struct QueryRec {
    std::vector<uint32_t> comps;
    uint32_t offset;
    uint32_t numMatching;
    uint32_t flags;
};

struct NodeRec {
    mwhip_node_desc desc;
    std::string name;
    std::vector<int32_t> deps;
};

struct TaskGraphRec {
    std::vector<NodeRec> nodes;
    std::vector<void *> dataDev;
    std::vector<int32_t> sorted;
    bool built = false;
};

struct LaunchGraph {
    // device memory that belongs to THIS graph (sort batches' site / column /
    // slice tables, scan state, row-snapshot granules ...): released when the
    // graph is rebuilt or freed, not at mwhip_destroy (every rebuild
    // -- growth, a table outgrowing the one-launch sort, set_input_ring -- used
    // to leave the previous graph's buffers allocated)
    std::vector<void *> ownedAllocations;
    std::vector<KernelLaunch> launches;
    std::vector<std::unique_ptr<SortBatch>> sortBatches;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graphExec = nullptr;
    std::string statName;
    std::vector<std::string> statNames;     // backing store for mwhip_kernel_stat::name
    std::vector<uint32_t> taskGraphIds;     // to rebuild after a table grew
    // optional last node: pack exported columns into a send buffer
    bool hasPack = false;
    PackArgs pack {};
    void *packDst = nullptr;
    // the batch ray caster's pass instead of task graphs
    bool isRender = false;

    LaunchGraph() = default;
    LaunchGraph(const LaunchGraph &) = delete;
    LaunchGraph &operator=(const LaunchGraph &) = delete;
    // (a graph that dies on an error path of instantiateLaunchGraph or of a
    // rebuild loop gives back what it had allocated so far; the caller has
    // made sure nothing of it is in flight)
    ~LaunchGraph()
    {
        if (graphExec) (void)hipGraphExecDestroy(graphExec);
        if (graph) (void)hipGraphDestroy(graph);
        for (void *p : ownedAllocations) {
            (void)hipFree(p);
        }
    }
};

// Where devAlloc records what it hands out while THIS thread builds a launch
// graph (the graph then owns the memory); another thread's allocations stay with
// their executor.
// (__thread: no dynamic initialisation, so no TLS init function to call from the
// other translation units)
extern MWHIP_RT __thread std::vector<void *> *t_allocScope;   // (runtime_state.hip)

struct mwhip_exec {
    mwhip_state_config cfg {};
    mwhip_user_entry entry {};
    hipStream_t stream = nullptr;

    std::vector<ComponentRec> components;
    std::vector<ArchetypeRec> archetypes;
    std::unordered_map<uint32_t, std::vector<uint32_t>> bundles;
    std::vector<QueryRec> queries;
    std::vector<uint32_t> queryDataHost;
    uint32_t queryCapacity = 1u << 16;      // reference state.hpp:207
    uint32_t numSingletons = 0;
    bool registrationOpen = false;
    bool stateBuilt = false;

    std::vector<void *> exported;

    EcsState hostState {};
    EcsState *stateDev = nullptr;
    std::vector<TableHdr> tablesHost;
    uint32_t singletonIdEnd = 0;            // rounded up to a block of 64

    void *userCfgDev = nullptr;
    void *worldInitsDev = nullptr;

    std::vector<TaskGraphRec> taskGraphs;
    std::unordered_map<uint64_t, std::unique_ptr<LaunchGraph>> launchGraphs;
    // where devAlloc records what it hands out: the graph being built, or
    // (nullptr) the executor's own list, freed at mwhip_destroy
    uint64_t nextGraphHandle = 1;

    // mwGPU::HostPrint: ring in pinned host memory + the thread that drains it
    // while replays are in flight (replaces the reference's HostPrintCPU
    // thread, src/mw/cuda_exec.cpp)
    // device -> host requests for table memory (mwhip::GrowMailbox) and the
    // lock that orders the service thread against growth between replays
    hipStream_t serviceStream = nullptr;    // fills of freshly mapped memory
    VmRange *entityVm = nullptr;            // entity slots (growable)
    VmRange *tmpVm = nullptr;               // Context::tmpAlloc region (growable)
    GrowMailbox *growMailbox = nullptr;
    std::mutex growMutex;
    bool headersStale = false;      // device headers / graphs lag the mapped rows
    HostPrintRing *printRing = nullptr;
    std::mutex printMutex;
    std::thread printThread;
    std::atomic<bool> printStop { false };
    int32_t *statsHost = nullptr;           // pinned, device-visible
    std::vector<void *> allocations;
    std::vector<std::unique_ptr<VmRange>> vmRanges;
    std::vector<uint32_t> rowsAtGraphBuild; // per archetype, see queryCapacityRows
    uint32_t *replaySignal = nullptr;       // device: replays completed
    uint32_t replaysLaunched = 0;           // host: replays queued
    uint32_t tableGrowth = 1;               // reserved / initial rows
    uint32_t numGrowths = 0;
    bool checkAfterRun = true;
    bool sortCarriesMisc = true;    // MADRONA_MWHIP_SORT_CARRIES_MISC

    // mwhip_set_input_ring
    struct InputRing {
        uint32_t *dst;
        const uint32_t *ring;
        uint32_t slotWords;
        uint32_t numSlots;
        uint32_t firstReplay;   // replays completed when the ring was set
    };
    std::vector<InputRing> inputRings;

    // MADRONA_TRACING builds: the device event log (mw_gpu/tracing.hpp), the
    // records of the first steps, the names funcID indexes
    void *deviceTracing = nullptr;
    void *traceCursor = nullptr;
    std::vector<mwGPU::DeviceTracing::DeviceLog> traceLogs;
    std::vector<std::string> traceNames;
    uint32_t traceSteps = 0;
    // MADRONA_MWHIP_SORT_COMPACT: 0 never, 1 world sorts of tables nothing else
    // reorders, 2 every world sort (tests: the chain is correct on any table,
    // its one-workgroup tail sort is just slow when the whole table is "tail")
    uint32_t sortCompaction = 1;
    bool eagerReplay = false;       // MADRONA_MWHIP_EAGER (measurement, replayGraph)
    const void *pforGroupKernel = nullptr;  // mwhip_set_pfor_group_kernel
    void *pforBodyScratch = nullptr;        // 8 bytes: where report mode writes
    // MADRONA_MWHIP_EXEC_CONFIG_FILE (the reference's
    // MADRONA_MWGPU_EXEC_CONFIG_FILE, cuda_exec.cpp:2115-2172): per task-graph
    // node (index in execution order) the workgroups per CU its kernel may
    // occupy -- the reference's "blocks per SM" of the megakernel that runs the
    // node --, written by madrona_amd/scripts/profile.py.  0 / absent: default.
    std::vector<uint32_t> nodeWorkgroupsPerCU;
    uint32_t numCUs = 256;

    // batch ray caster: geometry (bottom-level BVHs) + where the ECS keeps what
    // it reads and writes
    bool haveRenderGeometry = false;
    bool haveRenderLayout = false;
    RenderGeometryHost renderGeometry;
    RenderGeometryDev renderGeometryDev {};
    mwhip_render_layout renderLayout {};
    BvhNode *tlasNodes = nullptr;
    PreparedInstance *preparedInstances = nullptr;
};

// ---- functions one translation unit defines and another calls ----------------------
MWHIP_RT int devAlloc(mwhip_exec *exec, void **out, size_t bytes, bool zero = true);
      // (runtime_state.hip)
MWHIP_RT int vmAlloc(mwhip_exec *exec, void **out, VmRange **range_out,
                   size_t reserve_bytes, size_t map_bytes, bool zero);
      // (runtime_state.hip)
MWHIP_RT void vmFreeAll(mwhip_exec *exec);
      // (runtime_state.hip)
MWHIP_RT int findColumn(const ArchetypeRec &arch, uint32_t component_id);
      // (runtime.hip)
MWHIP_RT int buildDeviceState(mwhip_exec *exec);
      // (runtime_state.hip)
MWHIP_RT void drainHostPrints(mwhip_exec *exec, bool in_flight);
      // (runtime_state.hip)
MWHIP_RT const char *describeError(uint32_t flags);
      // (runtime_state.hip)
MWHIP_RT int launchOne(mwhip_exec *exec, KernelLaunch &k, hipStream_t stream);
      // (runtime_launch.hip)
MWHIP_RT bool compactionEligible(const mwhip_exec *exec, uint32_t archetype_id,
                               uint32_t component_id);
      // (runtime_launch.hip)
MWHIP_RT int sortAllArchetypes(mwhip_exec *exec);
      // (runtime_launch.hip)
MWHIP_RT int constructWorlds(mwhip_exec *exec);
      // (runtime_state.hip)
MWHIP_RT int loadExecConfigFile(mwhip_exec *exec);
      // (runtime_state.hip)
MWHIP_RT void releaseLaunchGraph(LaunchGraph &lg);
      // (runtime_launch.hip)
MWHIP_RT int instantiateLaunchGraph(mwhip_exec *exec,
                                  const std::vector<uint32_t> &ids,
                                  const std::string &stat_name,
                                  std::unique_ptr<LaunchGraph> &out,
                                  const LaunchGraph *pack_from = nullptr);
      // (runtime_launch.hip)
MWHIP_RT void serviceGrowRequests(mwhip_exec *exec);
      // (runtime_state.hip)
MWHIP_RT int growTablesFromDevice(mwhip_exec *exec);
      // (runtime_state.hip)
MWHIP_RT int sortsOutgrown(mwhip_exec *exec);
      // (runtime_state.hip)
MWHIP_RT int growTablesAfterReplay(mwhip_exec *exec);
      // (runtime_state.hip)
MWHIP_RT int collectDeviceTrace(mwhip_exec *exec);
      // (runtime_state.hip)
MWHIP_RT void writeDeviceTrace(mwhip_exec *exec);
      // (runtime_state.hip)

template <typename T>
inline int devAllocT(mwhip_exec *exec, T **out, size_t count, bool zero = true)
{
    return devAlloc(exec, (void **)out, count * sizeof(T), zero);
}


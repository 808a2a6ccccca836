// Internal declarations shared by the translation units of libmadrona_hip.so.
#pragma once

#include <hip/hip_runtime.h>

#include <madrona/ecs.hpp>
#include <madrona/mwhip/ecs_state.hpp>
#include <madrona/mw_gpu/tracing.hpp>
#include <mwhip.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace madrona {
namespace mwhip {

// ---- sort ----------------------------------------------------------------------

// Device-resident, per sort site (one per sorted archetype+key).
// Life cycle of one run of the chain (no kernel needs to know when another
// kernel's LAST workgroup is done -- one atomic per workgroup on one address
// serialises at 11 ns unloaded, ~70 ns under memory load on MI355X,
// profiles/tools/atomic_microbench.hip):
//   histogram   workgroup 0 of a site records active / rowsIn / keyColumn
//   last pass   workgroup 0 PUBLISHES the table (pointer swap, new row count):
//               nothing reads the table until the chain ends, and the pass
//               itself works from the key / index buffers
//   gather      old buffers (now columnsAlt) -> current buffers; one workgroup
//               cleans the state the passes have finished with
// Compaction chain (world sorts of tables that are still sorted from the last
// time, see sort_archetype.hip):
//   prepare     workgroup 0 records active / rowsIn / prefixRows / keyColumn,
//               sorts the appended tail and assigns its rows to prefix tiles;
//               the others count the prefix's survivors per tile
//   scatter     workgroup 0 publishes; every tile writes the permutation and
//               the sorted keys of its surviving rows and of the tail rows
//               that land in it
//   gather      as above
struct SortState {
    uint32_t bins[4 * 256];         // digit histograms of up to 4 passes
    uint32_t numValid;              // rows whose key != 0xFFFFFFFF
    uint32_t epoch;                 // tags look-back granules; never reset
    uint32_t active;                // this run sorts (world sort: table was dirty)
    int32_t rowsIn;                 // rows when the chain started
    int32_t rowsOut;                // rows the sorted table has
    uint32_t tailByLands;           // compaction chain: 1 = tailLand[] holds where every
                                    // RAW tail row lands (the scatter tiles pick and order
                                    // their own), 0 = prepare sorted the tail by world
    const uint32_t *keyColumn;      // the key column when the chain started
    unsigned long long statRowsIn;  // cumulative, for measurement
    unsigned long long statRowsOut;
    unsigned long long statRuns;
    // compaction chain
    int32_t prefixRows;             // rows of the sorted prefix this run starts from
    int32_t tailLive;               // live rows behind it (sorted by world by prepare)
    unsigned long long statTailRows;// cumulative rows behind the prefix
    uint32_t landsBlocked;          // runs that still take the sorted-tail path after a
                                    // scatter tile owned more tail rows than it orders in LDS
    uint32_t reserved_;
};

struct SortSite {
    uint32_t archetype;
    uint32_t keyColumn;
    int32_t numPasses;
    uint32_t worldSort;
    uint32_t numGatherColumns;      // gather grid rows (blockIdx.y) of this site
    uint32_t firstGatherColumn;     // ... and where they start in the batch's column list
    uint32_t hasPinned;             // some column must keep its address
    uint32_t *keysA;
    uint32_t *keysB;
    int32_t *idxA;
    int32_t *idxB;
    unsigned long long *lookback;   // [numTiles][256] granules
    SortState *state;
    // compaction chain: survivors of every 2048-row tile of the sorted prefix,
    // and for every tile the first row of the sorted tail that lands in it
    int32_t *tileCounts;            // [numTiles]
    int32_t *tileTailStart;         // [numTiles + 1]
    int32_t *tailLand;              // [rows] prefix position each tail row lands at (sorted
                                    // tail order, or raw tail order: SortState::tailByLands)
};

// pseudo-column of a world-sort site: rebuild worldOffsets / worldCounts
inline constexpr uint32_t kWorldRangesColumn = 0xFFFFFFFFu;

struct GatherColumn {
    uint32_t site;
    uint32_t column;                // or kWorldRangesColumn
    uint32_t wordBytes;             // 16 / 8 / 4 / 1
    uint32_t wordsPerRow;
    unsigned long long invMagic;    // floor(2^64 / wordsPerRow) + 1
    // rows of whole dwords move in 16-byte chunks of the destination
    // (gatherRowsWide); 0: word by word as described above
    uint32_t rowDwords;
    uint32_t pad_;
    unsigned long long invMagicDwords;  // floor(2^64 / rowDwords) + 1
};

// One workgroup of the gather: slice `slice` of `numSlices` of a column.  The
// workgroups of a launch are shared out over the columns by the bytes each has
// to move (a 28-byte solver-state column gets seven times the workgroups of a
// 4-byte id column), about as many in total as the chip keeps resident at once.
struct GatherSlice {
    uint32_t column;                // index into the batch's GatherColumn list
    uint32_t slice;
    uint32_t numSlices;
    uint32_t blocked;               // 1: a contiguous run of rows, 0: strided over the column
};

struct SortSiteHost {
    uint32_t archetype;
    uint32_t keyColumn;
    int numPasses;
    bool worldSort;
    uint32_t capacity;
    uint32_t rowBytes;
    SortState *stateDev;
};

enum class SortRole : uint32_t { None, Histogram, Onesweep, Gather, Finalize, Small,
                                 CompactPrepare, CompactScatter };

struct SortBatch {
    std::vector<SortSiteHost> sites;
    EcsState *stateDev = nullptr;
    SortSite *sitesDev = nullptr;
    GatherColumn *gatherColumnsDev = nullptr;
    uint32_t numGatherColumns = 0;
    GatherSlice *gatherSlicesDev = nullptr;
    uint32_t numGatherSlices = 0;
    bool hasPinned = false;         // a sorted table has exported columns
    bool gatherWide = false;        // some column moves in 16-byte chunks (sortGather<true>)
    uint32_t maxCapacity = 0;
    // every table of the batch holds few rows: one launch (sortSmall) instead
    // of the chain.  Decided from the rows the tables hold when the graph is
    // built; the executor rebuilds its graphs when one outgrows it.
    bool small = false;
    // every site is a world sort of a table that nothing but world sorts
    // reorders: prepare + scatter instead of histogram + key passes
    bool compact = false;
};

// ---- ClearTmp / ResetTmpAlloc ---------------------------------------------------
enum MiscOpKind : uint32_t { kOpClearTmp = 0, kOpResetTmpAlloc = 1 };

struct MiscOp {
    uint32_t kind;
    uint32_t archetype;
};

#if defined(__HIPCC__)
// thread `tid` of the caller applies op `tid`
__device__ inline void applyMiscOps(EcsState *S, const MiscOp *ops,
                                    uint32_t num_ops, uint32_t tid)
{
    if (tid >= num_ops) return;
    const MiscOp op = ops[tid];
    if (op.kind == kOpClearTmp) {
        TableHdr &tbl = S->tables[op.archetype];
        if (tbl.numRows != 0) {
            tbl.needsSort = 1u;
        }
        if (tbl.numRows > tbl.peakRows) {
            tbl.peakRows = tbl.numRows;
        }
        tbl.numRows = 0;
        tbl.sortedRows = 0;
    } else if (op.kind == kOpResetTmpAlloc) {
        S->tmpOffset = 0ull;
    }
}
#endif

// ---- launches ------------------------------------------------------------------

struct KernelLaunch {
    const void *fn = nullptr;
    dim3 grid { 1, 1, 1 };
    dim3 block { 1, 1, 1 };
    alignas(16) unsigned char argStorage[320] {};
    uint32_t argOffsets[9] {};
    uint32_t numArgs = 0;
    uint32_t dynamicLds = 0;        // bytes of dynamic LDS

    std::string name;               // node name
    const char *role = "";          // kernel role inside the node
    uint32_t kind = MWHIP_NODE_KERNEL;
    uint32_t archetype = 0xFFFFFFFFu;
    uint32_t bytesPerRow = 0;
    uint32_t ioDeclared = 0;        // bytesPerRow from a declared read / write set
    uint32_t nodeIndex = 0xFFFFFFFFu;   // position of the node in its task graph's order
    uint32_t countMode = 0;
    uint32_t fixedCount = 0;
    uint32_t queryOffset = 0;
    uint32_t numMatching = 0;
    const SortBatch *sortBatch = nullptr;
    SortRole sortRole = SortRole::None;
    bool carriesMisc = false;       // trailing (ops, count) arguments were pushed

    // ---- the launch's place in the step's DAG (buildLaunchDeps) ----
    // A kernel node of a task graph (ParallelFor / custom kernel) waits for the
    // launches of the nodes IT NAMED as dependencies, like the reference's
    // builder orders them -- nodes that name the same dependency may run side
    // by side --; every other launch (sorts, misc ops, the runtime's own
    // kernels) waits for everything before it and everything after waits for it.
    bool dagKernel = false;
    uint32_t tgId = 0;              // task graph the node belongs to
    int32_t tgNode = -1;            // node id inside it (mwhip_tg_add_node's return)
    std::vector<int32_t> deps;      // indices into LaunchGraph::launches

    // ---- side-by-side ParallelFor nodes in one launch (groupLaunches) ----
    const void *pforBody = nullptr;         // mwhip_node_desc::pfor_body
    const void *pforGroupKernel = nullptr;  // ... ::pfor_group_kernel (or the executor's)
    uint32_t pforArg1 = 0;                  // num_matching | exclusive-world flag
    bool rowSnapshot = false;               // the node can append rows: never grouped
    uint32_t pforVgprs = 0;                 // of the node's own kernel (groupLaunches)
    uint32_t pforWriteMask = 0xFFFFFFFFu;   // query components the system may write
    mwhip_pfor_args pforArgs {};
    struct Member {                         // of a grouped launch (profiles)
        std::string name;
        uint32_t queryOffset, bytesPerRow, ioDeclared;
    };
    std::vector<Member> members;

    template <typename T>
    void pushArg(const T &v)
    {
        uint32_t off = numArgs == 0 ? 0 : argOffsets[numArgs];
        off = (off + (uint32_t)alignof(T) - 1) & ~((uint32_t)alignof(T) - 1);
        memcpy(argStorage + off, &v, sizeof(T));
        argOffsets[numArgs] = off;
        numArgs += 1;
        argOffsets[numArgs] = off + (uint32_t)sizeof(T);
    }

    template <typename... Ts>
    void setArgs(const Ts &...vs)
    {
        numArgs = 0;
        argOffsets[0] = 0;
        (pushArg(vs), ...);
    }

    void argPointers(void **out)
    {
        for (uint32_t i = 0; i < numArgs; i++) {
            out[i] = argStorage + argOffsets[i];
        }
    }
};

int sortNumPasses(bool world_sort, uint32_t num_worlds);
uint32_t sortTileSize();
uint32_t sortSmallRowLimit();
uint32_t sortSmallBusyRows();
uint32_t sortCompactTailLimit();
void buildSortLaunches(const SortBatch &batch, std::vector<KernelLaunch> &out);

}
}

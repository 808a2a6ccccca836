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

// ---- this translation unit: from task graphs to launches -- sort batches, the launch
// list, side-by-side ParallelFor nodes, the step as a hipGraph ----

// ---------------------------------------------------------------------------
// launches
// ---------------------------------------------------------------------------
MWHIP_RT int launchOne(mwhip_exec *exec, KernelLaunch &k, hipStream_t stream)
{
    void *args[8];
    k.argPointers(args);
    HIPCHK(hipLaunchKernel(k.fn, k.grid, k.block, args, k.dynamicLds, stream));
    (void)exec;
    return 0;
}

static int ensureSortScratch(mwhip_exec *exec, ArchetypeRec &arch)
{
    if (arch.sortState != nullptr) return 0;
    // (shared by every graph that sorts this table: not the building graph's)
    struct ScopeOff {
        mwhip_exec *e; std::vector<void *> *saved;
        explicit ScopeOff(mwhip_exec *x) : e(x), saved(t_allocScope) { t_allocScope = nullptr; }
        ~ScopeOff() { (void)e; t_allocScope = saved; }
    } scope_off(exec);

    int rc = devAllocT(exec, &arch.sortState, 1);
    if (rc != 0) return rc;
    if (arch.reservedCapacity > arch.capacity) {
        void **bufs[5] = { (void **)&arch.keysA, (void **)&arch.keysB,
                           (void **)&arch.idxA, (void **)&arch.idxB,
                           (void **)&arch.tailLand };
        for (int i = 0; i < 5; i++) {
            rc = vmAlloc(exec, bufs[i], &arch.sortVm[i],
                         (size_t)arch.reservedCapacity * 4,
                         (size_t)arch.capacity * 4, false);
            if (rc != 0) return rc;
        }
    } else {
        rc = devAllocT(exec, &arch.keysA, arch.capacity, false);
        if (rc != 0) return rc;
        rc = devAllocT(exec, &arch.keysB, arch.capacity, false);
        if (rc != 0) return rc;
        rc = devAllocT(exec, &arch.idxA, arch.capacity, false);
        if (rc != 0) return rc;
        rc = devAllocT(exec, &arch.idxB, arch.capacity, false);
        if (rc != 0) return rc;
        rc = devAllocT(exec, &arch.tailLand, arch.capacity, false);
        if (rc != 0) return rc;
    }
    // (look-back slots for every tile the table can ever have)
    size_t tiles =
        (arch.reservedCapacity + sortTileSize() - 1) / sortTileSize();
    rc = devAllocT(exec, &arch.lookback, tiles * 256);
    if (rc != 0) return rc;
    rc = devAllocT(exec, &arch.tileCounts, tiles + 2);
    if (rc != 0) return rc;
    rc = devAllocT(exec, &arch.tileTailStart, tiles + 2);
    return rc;
}

// Can the world sort of this archetype start from what the last one left?
MWHIP_RT bool compactionEligible(const mwhip_exec *exec, uint32_t archetype_id,
                               uint32_t component_id)
{
    const ArchetypeRec &arch = exec->archetypes[archetype_id];
    if (component_id != 1 || exec->sortCompaction == 0) return false;
    return exec->sortCompaction == 2 || (!arch.scrambled && !arch.noCompact);
}

// batches that move at least this much take the gather's blocked assignment
static constexpr double kGatherBlockedBytes = 256.0 * 1024.0 * 1024.0;
static int makeSortBatch(mwhip_exec *exec,
                         const std::vector<std::pair<uint32_t, uint32_t>> &specs,
                         std::unique_ptr<SortBatch> &out, bool compact = false)
{
    out.reset(new SortBatch {});
    out->stateDev = exec->stateDev;
    out->compact = compact;

    std::vector<SortSite> sites;
    std::vector<GatherColumn> cols;
    bool all_small = envU32("MADRONA_MWHIP_SORT_SMALL", 1) != 0;

    for (auto [archetype_id, component_id] : specs) {
        if (archetype_id >= exec->archetypes.size() ||
                !exec->archetypes[archetype_id].registered) {
            return fail(-3, "sort node on unregistered archetype %u",
                        archetype_id);
        }
        ArchetypeRec &arch = exec->archetypes[archetype_id];
        int key_col = findColumn(arch, component_id);
        if (key_col < 0) {
            return fail(-3, "sort node: archetype %u has no component %u",
                        archetype_id, component_id);
        }
        if (arch.colBytes[key_col] != 4) {
            return fail(-3, "sort key component %u is not 4 bytes",
                        component_id);
        }

        int rc = ensureSortScratch(exec, arch);
        if (rc != 0) return rc;

        const bool world_sort = component_id == 1;

        SortSiteHost host_site {};
        host_site.archetype = archetype_id;
        host_site.keyColumn = (uint32_t)key_col;
        host_site.worldSort = world_sort;
        host_site.numPasses = sortNumPasses(world_sort, exec->cfg.num_worlds);
        host_site.capacity = arch.capacity;
        host_site.rowBytes = arch.rowBytes;
        host_site.stateDev = arch.sortState;
        out->sites.push_back(host_site);

        SortSite site {};
        site.archetype = archetype_id;
        site.keyColumn = (uint32_t)key_col;
        site.numPasses = host_site.numPasses;
        site.worldSort = world_sort ? 1u : 0u;
        site.keysA = arch.keysA;
        site.keysB = arch.keysB;
        site.idxA = arch.idxA;
        site.idxB = arch.idxB;
        site.lookback = arch.lookback;
        site.state = arch.sortState;
        site.tileCounts = arch.tileCounts;
        site.tileTailStart = arch.tileTailStart;
        site.tailLand = arch.tailLand;
        sites.push_back(site);

        out->maxCapacity = std::max(out->maxCapacity, arch.capacity);
        {
            // small-table path while the table holds at most a quarter of the
            // limit (checked again after every replay, sortsOutgrown())
            uint64_t rows_now = archetype_id < exec->rowsAtGraphBuild.size() ?
                exec->rowsAtGraphBuild[archetype_id] : arch.capacity;
            // (a table that can never hold more than that quarter stays small)
            if ((uint64_t)arch.capacity * 4 <= sortSmallRowLimit()) rows_now = 0;
            if (arch.bigSort || rows_now * 4 > sortSmallRowLimit()) {
                all_small = false;
            }
        }
        uint32_t site_columns = 0;
        sites.back().firstGatherColumn = (uint32_t)cols.size();
        if (world_sort) {
            // first in the list: its two binary-search chains per world overlap
            // with the column traffic of the workgroups scheduled after it
            GatherColumn ranges {};
            ranges.site = (uint32_t)sites.size() - 1;
            ranges.column = kWorldRangesColumn;
            cols.push_back(ranges);
            site_columns++;
        }
        for (uint32_t c = 0; c < arch.numColumns; c++) {
            if ((arch.colFlags[c] & kColumnPinned) != 0u) {
                out->hasPinned = true;
                sites.back().hasPinned = 1u;
            }
            site_columns++;
            uint32_t bytes = arch.colBytes[c];
            GatherColumn gc {};
            gc.site = (uint32_t)sites.size() - 1;
            gc.column = c;
            gc.wordBytes = bytes % 16 == 0 ? 16 : (bytes % 8 == 0 ? 8 :
                (bytes % 4 == 0 ? 4 : 1));
            gc.wordsPerRow = bytes / gc.wordBytes;
            gc.invMagic = gc.wordsPerRow <= 1 ? 0ull :
                (~0ull / gc.wordsPerRow) + 1ull;
            // MADRONA_MWHIP_GATHER_WIDE=1: rows of whole dwords in 16-byte chunks
            // of the destination (round 3's default).  Re-measured in round 4
            // next to the blocked assignment (profiles/r04_sort_variants.jsonl):
            // word by word is as fast or faster at every size -- 15.0 against
            // 15.6 us at 4096 Escape-Room worlds, 28.1 against 30.7 at 8192 with
            // physics, 168 against 204 at 65536 -- so that is the default again.
            const bool wide = envU32("MADRONA_MWHIP_GATHER_WIDE", 0) != 0;
            if (wide && bytes % 4 == 0 && bytes != 0) {
                gc.rowDwords = bytes / 4;
                out->gatherWide = true;
                gc.invMagicDwords = gc.rowDwords <= 1 ? 0ull :
                    (~0ull / gc.rowDwords) + 1ull;
            }
            cols.push_back(gc);
        }
        sites.back().numGatherColumns = site_columns;
    }
    out->small = all_small;

    int rc = devAllocT(exec, &out->sitesDev, sites.size());
    if (rc != 0) return rc;
    HIPCHK(hipMemcpy(out->sitesDev, sites.data(),
        sites.size() * sizeof(SortSite), hipMemcpyHostToDevice));
    rc = devAllocT(exec, &out->gatherColumnsDev, cols.size());
    if (rc != 0) return rc;
    HIPCHK(hipMemcpy(out->gatherColumnsDev, cols.data(),
        cols.size() * sizeof(GatherColumn), hipMemcpyHostToDevice));
    out->numGatherColumns = (uint32_t)cols.size();

    // ---- the gather's workgroups, shared out by bytes to move ----
    {
        // about one workgroup per 32 KB, between one and four resident rounds of
        // the chip (measured, profiles/r03_sort_variants.jsonl: 2048 is best at
        // 38 MB, 4096 at 94 MB, 8192 at 610 MB)
        std::vector<double> weight(cols.size(), 0.0);
        double total = 0.0;
        for (size_t c = 0; c < cols.size(); c++) {
            const SortSiteHost &site = out->sites[cols[c].site];
            const ArchetypeRec &arch = exec->archetypes[site.archetype];
            uint64_t rows = site.archetype < exec->rowsAtGraphBuild.size() ?
                exec->rowsAtGraphBuild[site.archetype] : 0;
            if (rows == 0) rows = arch.capacity;
            if (cols[c].column == kWorldRangesColumn) {
                // two binary searches per world: latency, not bytes
                weight[c] = 64.0 * exec->cfg.num_worlds;
            } else {
                // (+ the 4-byte permutation entry every row of a column reads)
                weight[c] = (double)rows * (arch.colBytes[cols[c].column] + 4.0);
            }
            total += weight[c];
        }
        const uint32_t target =
            (uint32_t)std::min(std::max(total / 32768.0, 2048.0), 8192.0);
        std::vector<GatherSlice> slices;
        // big batches: a contiguous run of rows per workgroup instead of a
        // stride over the whole column (measured at 65536 / 16384 Escape-Room
        // worlds on one box, profiles/r04_sort_variants.jsonl: strided 300 /
        // 53 us, contiguous 252 / 52 us)
        const uint32_t blocked = total >= kGatherBlockedBytes ? 1u : 0u;
        for (size_t c = 0; c < cols.size(); c++) {
            uint32_t n = (uint32_t)(target * weight[c] / std::max(total, 1.0) + 0.5);
            // at least 4 KB of work per workgroup, at least one workgroup
            n = std::min<uint32_t>(n, (uint32_t)(weight[c] / 4096.0) + 1u);
            n = std::max<uint32_t>(n, 1u);
            for (uint32_t i = 0; i < n; i++) {
                slices.push_back(GatherSlice { (uint32_t)c, i, n, blocked });
            }
        }
        rc = devAllocT(exec, &out->gatherSlicesDev, slices.size());
        if (rc != 0) return rc;
        HIPCHK(hipMemcpy(out->gatherSlicesDev, slices.data(),
            slices.size() * sizeof(GatherSlice), hipMemcpyHostToDevice));
        out->numGatherSlices = (uint32_t)slices.size();
    }
    return 0;
}

static void pickGrid(mwhip_exec *exec, KernelLaunch &k, uint64_t max_invocations,
                     uint32_t threads_per_invocation)
{
    (void)exec;
    if (threads_per_invocation >= 64) {
        // wave- (or workgroup-) per-invocation kernels: one workgroup of that
        // size per invocation, the hardware scheduler balances the rest
        k.block = dim3(threads_per_invocation, 1, 1);
        k.grid = dim3((uint32_t)std::min<uint64_t>(
            std::max<uint64_t>(max_invocations, 1), 1u << 20), 1, 1);
        return;
    }
    uint64_t threads = max_invocations * std::max(threads_per_invocation, 1u);
    // small tables: 64-thread workgroups so the work spreads over more CUs;
    // big tables: 256-thread workgroups, capped, with grid-stride loops
    uint32_t block = threads >= 256ull * 512ull ? 256u : 64u;
    uint64_t blocks = (threads + block - 1) / block;
    blocks = std::min<uint64_t>(std::max<uint64_t>(blocks, 1), 2048);
    k.block = dim3(block, 1, 1);
    k.grid = dim3((uint32_t)blocks, 1, 1);
}

static uint64_t queryCapacityRows(mwhip_exec *exec, uint32_t offset,
                                  uint32_t num_matching, uint32_t num_components_hint)
{
    (void)num_components_hint;
    // find the query record to learn its component count
    for (const QueryRec &q : exec->queries) {
        if (q.offset == offset) {
            uint64_t rows = 0;
            const uint32_t *p = exec->queryDataHost.data() + offset;
            for (uint32_t i = 0; i < num_matching; i++) {
                // Grids follow what the tables hold when the graph is built
                // (x2 head room, at least 4096), not their capacity: the
                // kernels stride over the device-resident row count, so a
                // fuller table is still covered, and a table declared with 64
                // rows per world but holding 4 does not launch 16x the
                // workgroups it needs (a trivial system: 4.3 -> ~3 us).  Graphs
                // are rebuilt when a table grows.
                const ArchetypeRec &arch = exec->archetypes[p[0]];
                uint64_t live = p[0] < exec->rowsAtGraphBuild.size() ?
                    exec->rowsAtGraphBuild[p[0]] : arch.capacity;
                // (tables that are empty at build time -- temporaries, joints
                // -- say nothing about their steady state: capacity)
                rows += live == 0 ? arch.capacity :
                    std::min<uint64_t>(arch.capacity,
                                       std::max<uint64_t>(2 * live, 4096));
                p += 1 + q.comps.size();
            }
            return rows;
        }
    }
    return 0;
}

// Which tables does something other than a world sort reorder or truncate?
// (every task graph of the executor counts, not only the ones being built: a
// launch graph over another task graph may run in between)
static int findScrambledTables(mwhip_exec *exec)
{
    for (ArchetypeRec &arch : exec->archetypes) {
        arch.scrambled = false;
    }
    for (TaskGraphRec &tg : exec->taskGraphs) {
        for (const NodeRec &node : tg.nodes) {
            const mwhip_node_desc &d = node.desc;
            if (d.kind == MWHIP_NODE_CLEAR_TMP ||
                    (d.kind == MWHIP_NODE_SORT_ARCHETYPE && d.component_id != 1)) {
                if (d.archetype_id < exec->archetypes.size()) {
                    exec->archetypes[d.archetype_id].scrambled = true;
                }
            } else if (d.kind == MWHIP_NODE_EXCLUSIVE_SCAN && d.node_data_id >= 0) {
                // a scan that writes its total into a table's row count
                mwhip_scan_params params;
                HIPCHK(hipMemcpy(&params, tg.dataDev[d.node_data_id],
                                 sizeof(params), hipMemcpyDeviceToHost));
                const char *first = (const char *)exec->hostState.tables;
                const char *at = (const char *)params.total_out;
                if (at >= first && at < first + exec->tablesHost.size() *
                                                 sizeof(TableHdr)) {
                    exec->archetypes[(size_t)(at - first) / sizeof(TableHdr)]
                        .scrambled = true;
                }
            }
        }
    }
    return 0;
}

static int buildLaunchList(mwhip_exec *exec, const std::vector<uint32_t> &tg_ids,
                           LaunchGraph &lg)
{
    {
        int rc = findScrambledTables(exec);
        if (rc != 0) return rc;
    }
    if (!lg.isRender) {
        for (const mwhip_exec::InputRing &ring : exec->inputRings) {
            KernelLaunch k;
            k.fn = inputRingKernelFn();
            k.grid = dim3(std::min<uint32_t>((ring.slotWords + 255u) / 256u, 1024u), 1, 1);
            k.block = dim3(256, 1, 1);
            k.setArgs(exec->stateDev, ring.dst, ring.ring, ring.slotWords, ring.numSlots,
                      ring.firstReplay);
            k.name = "input";
            k.role = "ring";
            k.kind = MWHIP_NODE_RECYCLE;
            lg.launches.push_back(k);
        }
    }
    for (uint32_t tg_id : tg_ids) {
        if (tg_id >= exec->taskGraphs.size()) {
            return fail(-3, "task graph %u does not exist", tg_id);
        }
        TaskGraphRec &tg = exec->taskGraphs[tg_id];
        const std::vector<int32_t> &order = tg.sorted;

        std::vector<MiscOp> pending_misc;
        auto flushMisc = [&]() -> int {
            if (pending_misc.empty()) return 0;
            MiscOp *ops_dev;
            int rc = devAllocT(exec, &ops_dev, pending_misc.size());
            if (rc != 0) return rc;
            HIPCHK(hipMemcpy(ops_dev, pending_misc.data(),
                pending_misc.size() * sizeof(MiscOp), hipMemcpyHostToDevice));
            KernelLaunch k;
            k.fn = miscOpsKernelFn();
            k.grid = dim3(1, 1, 1);
            k.block = dim3(64, 1, 1);
            k.setArgs(exec->stateDev, (const MiscOp *)ops_dev,
                      (uint32_t)pending_misc.size());
            k.name = "misc";
            k.role = "clear/reset";
            k.kind = MWHIP_NODE_CLEAR_TMP;
            lg.launches.push_back(k);
            pending_misc.clear();
            return 0;
        };

        for (size_t oi = 0; oi < order.size(); oi++) {
            NodeRec &node = tg.nodes[order[oi]];
            const mwhip_node_desc &d = node.desc;

            switch (d.kind) {
            case MWHIP_NODE_KERNEL: {
                int rc = flushMisc();
                if (rc != 0) return rc;

                KernelLaunch k;
                k.fn = d.kernel;
                void *data_dev = d.node_data_id >= 0 ?
                    tg.dataDev[d.node_data_id] : nullptr;
                k.setArgs(exec->stateDev, data_dev, d.arg0, d.arg1);
                if (d.wants_pfor_args != 0 &&
                        d.count_mode == MWHIP_COUNT_QUERY_ROWS) {
                    mwhip_pfor_args pa {};
                    pa.num_matching = d.num_matching;
                    pa.num_inline = 0;
                    for (const QueryRec &q : exec->queries) {
                        if (q.offset != d.query_offset) continue;
                        if (q.numMatching <= MWHIP_PFOR_MAX_INLINE &&
                                q.comps.size() <= MWHIP_PFOR_MAX_COMPONENTS) {
                            const uint32_t *p =
                                exec->queryDataHost.data() + q.offset;
                            for (uint32_t m = 0; m < q.numMatching; m++) {
                                pa.tables[m] = exec->hostState.tables + p[0];
                                for (size_t c = 0; c < q.comps.size(); c++) {
                                    pa.columns[m][c] = (uint16_t)p[1 + c];
                                }
                                p += 1 + q.comps.size();
                            }
                            pa.num_inline = q.numMatching;
                        }
                        break;
                    }
                    // A system that can append rows (its kernel carries the
                    // static LDS marker of appendRowIssue; any other static
                    // LDS errs on the safe side) must not visit rows created
                    // during its own node: such nodes fix their row counts
                    // once per launch.
                    hipFuncAttributes attr {};
                    HIPCHK(hipFuncGetAttributes(&attr, d.kernel));
                    if (d.num_matching > 0 && attr.sharedSizeBytes != 0) {
                        void *sync_dev = nullptr;
                        int src = devAlloc(exec, &sync_dev,
                            sizeof(PforRowSync) + 8ull * d.num_matching);
                        if (src != 0) return src;
                        pa.row_sync = sync_dev;
                        k.dynamicLds = 4u * d.num_matching;
                    }
                    k.pushArg(pa);
                    k.pforArgs = pa;
                    k.rowSnapshot = pa.row_sync != nullptr;
                    k.pforBody = d.pfor_body;
                    k.pforGroupKernel = d.pfor_group_kernel != nullptr ?
                        d.pfor_group_kernel : exec->pforGroupKernel;
                    k.pforArg1 = d.arg1;
                    k.pforVgprs = (uint32_t)std::max(attr.numRegs, 0);
                    k.pforWriteMask = d.write_mask;
                }
                k.name = node.name;
                k.role = "";
                k.kind = d.kind;
                k.bytesPerRow = d.bytes_per_row;
                k.ioDeclared = d.io_declared;
                k.countMode = d.count_mode;
                k.fixedCount = d.fixed_count;
                k.queryOffset = d.query_offset;
                k.numMatching = d.num_matching;

                uint64_t max_inv = 0;
                if (d.count_mode == MWHIP_COUNT_QUERY_ROWS) {
                    max_inv = queryCapacityRows(exec, d.query_offset,
                                                d.num_matching, 0);
                    if (d.num_matching == 0) max_inv = 1;
                } else if (d.count_mode == MWHIP_COUNT_PER_WORLD) {
                    max_inv = exec->cfg.num_worlds;
                } else {
                    max_inv = d.fixed_count == 0xFFFFFFFFu ?
                        (256ull * 1024ull) : std::max(d.fixed_count, 1u);
                }
                pickGrid(exec, k, max_inv, d.threads_per_invocation);
                // exec config: workgroups per CU of this node's kernel (only
                // ParallelFor kernels: they stride over their rows with the grid)
                if (d.count_mode == MWHIP_COUNT_QUERY_ROWS &&
                        oi < exec->nodeWorkgroupsPerCU.size() &&
                        exec->nodeWorkgroupsPerCU[oi] != 0u) {
                    k.grid.x = std::min<uint32_t>(k.grid.x,
                        exec->nodeWorkgroupsPerCU[oi] * exec->numCUs);
                }
                k.nodeIndex = (uint32_t)oi;
                k.dagKernel = true;
                k.tgId = tg_id;
                k.tgNode = order[oi];
                lg.launches.push_back(k);
            } break;
            case MWHIP_NODE_SORT_ARCHETYPE: {
                int rc = flushMisc();
                if (rc != 0) return rc;

                // Batch this sort with the sort nodes that directly follow it.
                // ResetTmpAlloc / Recycle nodes in between commute with the
                // sort (it uses neither) and are replayed after the batch.
                std::vector<std::pair<uint32_t, uint32_t>> specs;
                specs.emplace_back(d.archetype_id, d.component_id);
                std::string name = node.name;
                size_t oj = oi + 1;
                {
                    for (; oj < order.size(); oj++) {
                        const mwhip_node_desc &nd = tg.nodes[order[oj]].desc;
                        if (nd.kind == MWHIP_NODE_RESET_TMP_ALLOC) {
                            pending_misc.push_back({ kOpResetTmpAlloc, 0 });
                            continue;
                        }
                        if (nd.kind == MWHIP_NODE_RECYCLE) {
                            continue;
                        }
                        if (nd.kind != MWHIP_NODE_SORT_ARCHETYPE) {
                            break;
                        }
                        bool dup = false;
                        for (auto &s : specs) {
                            if (s.first == nd.archetype_id) dup = true;
                        }
                        if (dup) break;
                        specs.emplace_back(nd.archetype_id, nd.component_id);
                    }
                    // collapse duplicate deferred resets
                    if (pending_misc.size() > 1) pending_misc.resize(1);
                }

                // World sorts of tables that nothing else reorders start from
                // what the last sort left (compaction chain); the rest of the
                // batch takes the radix chain.  Sites of one batch sort
                // different tables, so the two chains commute.
                // (Batches of small tables stay whole: one launch either way.)
                std::vector<std::pair<uint32_t, uint32_t>> by_chain[2];
                bool any_compact = false;
                for (auto &spec : specs) {
                    any_compact = any_compact ||
                        compactionEligible(exec, spec.first, spec.second);
                }
                std::unique_ptr<SortBatch> whole;
                rc = makeSortBatch(exec, specs, whole, false);
                if (rc != 0) return rc;
                const bool split = any_compact && !whole->small;
                for (auto &spec : specs) {
                    by_chain[split && compactionEligible(exec, spec.first,
                                                         spec.second) ?
                             1 : 0].push_back(spec);
                }
                for (int chain = 0; chain < 2; chain++) {
                    if (by_chain[chain].empty()) continue;
                    std::unique_ptr<SortBatch> batch;
                    if (!split) {
                        batch = std::move(whole);
                    } else {
                        rc = makeSortBatch(exec, by_chain[chain], batch, chain == 1);
                        if (rc != 0) return rc;
                    }

                    size_t first = lg.launches.size();
                    buildSortLaunches(*batch, lg.launches);
                    for (size_t i = first; i < lg.launches.size(); i++) {
                        lg.launches[i].name = name;
                        lg.launches[i].archetype = by_chain[chain][0].first;
                    }
                    lg.sortBatches.push_back(std::move(batch));
                }
                // The ResetTmpAlloc nodes deferred behind the batch ride on its
                // last kernel instead of taking a launch of their own (~4 us):
                // the sort uses neither the scratch allocator nor their result.
                if (!pending_misc.empty() && exec->sortCarriesMisc) {
                    MiscOp *ops_dev;
                    rc = devAllocT(exec, &ops_dev, pending_misc.size());
                    if (rc != 0) return rc;
                    HIPCHK(hipMemcpy(ops_dev, pending_misc.data(),
                        pending_misc.size() * sizeof(MiscOp), hipMemcpyHostToDevice));
                    // (the last kernel of a chain is sortSmall, sortGather or
                    // sortFinalize: each ends in (ops, count), empty by default)
                    KernelLaunch &last = lg.launches.back();
                    last.numArgs -= 2;
                    last.pushArg((const MiscOp *)ops_dev);
                    last.pushArg((uint32_t)pending_misc.size());
                    last.carriesMisc = true;
                    pending_misc.clear();
                }
                oi = oj - 1;
            } break;
            case MWHIP_NODE_EXCLUSIVE_SCAN: {
                int rc = flushMisc();
                if (rc != 0) return rc;
                if (d.node_data_id < 0) {
                    return fail(-3, "scan node '%s' without parameters",
                                node.name.c_str());
                }

                // node data holds mwhip_scan_params; wrap it with the scan's
                // own state (ticket / epoch / granules)
                mwhip_scan_params params;
                HIPCHK(hipMemcpy(&params, tg.dataDev[d.node_data_id],
                                 sizeof(params), hipMemcpyDeviceToHost));
                uint32_t max_tiles = d.fixed_count == 0 ? 1u :
                    (d.fixed_count + kScanTile - 1) / kScanTile +
                    MWHIP_SCAN_MAX_SEGMENTS;

                ScanNode host_node {};
                host_node.params = params;
                host_node.maxTiles = max_tiles;
                rc = devAllocT(exec, &host_node.state, 1);
                if (rc != 0) return rc;
                rc = devAllocT(exec, &host_node.granules, max_tiles);
                if (rc != 0) return rc;
                ScanNode *node_dev;
                rc = devAllocT(exec, &node_dev, 1);
                if (rc != 0) return rc;
                HIPCHK(hipMemcpy(node_dev, &host_node, sizeof(ScanNode),
                                 hipMemcpyHostToDevice));

                KernelLaunch k;
                k.fn = exclusiveScanKernelFn();
                k.grid = dim3(max_tiles, 1, 1);
                k.block = dim3(kScanThreads, 1, 1);
                k.setArgs(exec->stateDev, (const ScanNode *)node_dev);
                k.name = node.name;
                k.role = "scan";
                k.kind = d.kind;
                lg.launches.push_back(k);
            } break;
            case MWHIP_NODE_CLEAR_TMP:
                pending_misc.push_back({ kOpClearTmp, d.archetype_id });
                break;
            case MWHIP_NODE_RESET_TMP_ALLOC:
                pending_misc.push_back({ kOpResetTmpAlloc, 0 });
                break;
            case MWHIP_NODE_RECYCLE:
                break;
            default:
                return fail(-3, "unknown node kind %u", d.kind);
            }

            if (pending_misc.size() >= 64) {
                int rc = flushMisc();
                if (rc != 0) return rc;
            }
        }

        int rc = flushMisc();
        if (rc != 0) return rc;
    }

    // every replay ends with the health kernel: error flags and row counts to
    // pinned host memory (table growth reads them), replay counter bumped
    {
        KernelLaunch k;
        k.fn = statsKernelFn();
        k.grid = dim3(1, 1, 1);
        k.block = dim3(256, 1, 1);
        int32_t *host_out = nullptr;
        HIPCHK(hipHostGetDevicePointer((void **)&host_out, exec->statsHost, 0));
        k.setArgs(exec->stateDev, host_out, exec->replaySignal,
                  lg.isRender ? 0u : 1u);
        k.name = "stats";
        k.role = "health";
        k.kind = MWHIP_NODE_RECYCLE;
        lg.launches.push_back(k);
    }

    return 0;
}

MWHIP_RT int sortAllArchetypes(mwhip_exec *exec)
{
    // World-sort every non-singleton table once after world construction so
    // rows are world-major (world constructors run in parallel and append in
    // arrival order; the stable sort keeps each world's creation order).
    std::vector<std::pair<uint32_t, uint32_t>> specs;
    for (uint32_t a = 0; a < exec->archetypes.size(); a++) {
        const ArchetypeRec &arch = exec->archetypes[a];
        if (arch.registered && !arch.singleton) {
            specs.emplace_back(a, 1u);
        }
    }
    if (specs.empty()) return 0;

    std::unique_ptr<SortBatch> batch;
    int rc = makeSortBatch(exec, specs, batch);
    if (rc != 0) return rc;

    std::vector<KernelLaunch> launches;
    buildSortLaunches(*batch, launches);
    for (KernelLaunch &k : launches) {
        rc = launchOne(exec, k, exec->stream);
        if (rc != 0) return rc;
    }
    HIPCHK(hipStreamSynchronize(exec->stream));
    return 0;
}

// ---------------------------------------------------------------------------
// launch graphs
// ---------------------------------------------------------------------------
// Launch list (grids sized from the tables' current capacities) + hipGraph.
// The ray caster's launches for the tables as they are now (grids follow the
// camera table's capacity: the kernels loop over the views that exist).
static int renderLaunches(mwhip_exec *exec, std::vector<KernelLaunch> &out)
{
    const mwhip_render_layout &lay = exec->renderLayout;
    auto column_of = [exec](uint32_t archetype, uint32_t component,
                            uint32_t *out_col) -> int {
        if (archetype >= exec->archetypes.size() ||
                !exec->archetypes[archetype].registered) {
            return fail(-3, "render layout: archetype %u is not registered",
                        archetype);
        }
        const ArchetypeRec &arch = exec->archetypes[archetype];
        for (uint32_t c = 2; c < arch.numColumns; c++) {
            if (arch.colComponent[c] == component) {
                *out_col = c;
                return 0;
            }
        }
        return fail(-3, "render layout: archetype %u has no component %u",
                    archetype, component);
    };

    RenderParams params {};
    params.layout = lay;
    int rc = column_of(lay.renderable_archetype, lay.instance_component,
                       &params.instanceColumn);
    if (rc != 0) return rc;
    rc = column_of(lay.renderable_archetype, lay.morton_component,
                   &params.mortonColumn);
    if (rc != 0) return rc;
    rc = column_of(lay.renderable_archetype, lay.tlbvh_component,
                   &params.tlbvhColumn);
    if (rc != 0) return rc;
    rc = column_of(lay.camera_archetype, lay.camera_component,
                   &params.cameraColumn);
    if (rc != 0) return rc;
    rc = column_of(lay.light_archetype, lay.light_component, &params.lightColumn);
    if (rc != 0) return rc;
    rc = column_of(lay.output_archetype, lay.rgb_component, &params.rgbColumn);
    if (rc != 0) return rc;
    rc = column_of(lay.output_archetype, lay.depth_component,
                   &params.depthColumn);
    if (rc != 0) return rc;

    const ArchetypeRec &inst = exec->archetypes[lay.renderable_archetype];
    const ArchetypeRec &cams = exec->archetypes[lay.camera_archetype];
    const ArchetypeRec &outs = exec->archetypes[lay.output_archetype];
    const uint32_t res = exec->cfg.raycast_output_resolution;
    if (inst.colBytes[params.instanceColumn] != 64 ||
            inst.colBytes[params.tlbvhColumn] != 32 ||
            cams.colBytes[params.cameraColumn] != 48 ||
            exec->archetypes[lay.light_archetype].colBytes[params.lightColumn] != 40 ||
            outs.colBytes[params.depthColumn] != res * res * 4u) {
        return fail(-3, "render layout: component sizes are not those of "
                    "madrona/render/ecs.hpp");
    }

    if (exec->tlasNodes == nullptr) {
        // one node slot per instance row the table can ever hold (a world of n
        // instances uses n - 1 of its n slots).  (The executor's, not the
        // building graph's: every later render graph reuses them.)
        std::vector<void *> *const saved_scope = t_allocScope;
        t_allocScope = nullptr;
        struct Restore {
            mwhip_exec *e; std::vector<void *> *s;
            ~Restore() { (void)e; t_allocScope = s; }
        } restore { exec, saved_scope };
        rc = devAllocT(exec, &exec->tlasNodes, inst.reservedCapacity, false);
        if (rc != 0) return rc;
        rc = devAllocT(exec, &exec->preparedInstances, inst.reservedCapacity,
                       false);
        if (rc != 0) return rc;
    }

    params.resolution = res;
    params.rgbd = exec->cfg.raycast_rgbd;
    params.tlasNodes = exec->tlasNodes;
    params.prepared = exec->preparedInstances;
    params.geometry = exec->renderGeometryDev;
    // (any grid is correct: workgroups stride over the tiles of the views that
    // exist; sized for the views the table held when the graph was built)
    uint32_t views = cams.capacity;
    if (lay.camera_archetype < exec->rowsAtGraphBuild.size()) {
        views = std::max(exec->rowsAtGraphBuild[lay.camera_archetype], 16u);
    }
    // (the grid: one workgroup per view, at least six per CU -- see
    // buildRenderLaunches)
    int num_cus = 256;
    (void)hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount,
                                exec->cfg.gpu_id);
    // (per executor: its own device's CU count)
    const uint32_t max_wgs = (uint32_t)std::max(num_cus, 1) * 6u;
    buildRenderLaunches(exec->stateDev, params, exec->cfg.num_worlds, views,
                        std::max(max_wgs, 1u), out);
    return 0;
}

#ifdef MADRONA_TRACING
// Marker launches (traceMarkKernel) around the kernels of a graph: called last,
// when render / pack kernels have been spliced in.
static int addTraceMarkers(mwhip_exec *exec, LaunchGraph &lg)
{
    {
        using mwGPU::DeviceEvent;
        auto mark = [&](DeviceEvent event, uint32_t node_id, uint32_t func_id,
                        uint32_t invocations, uint32_t workgroups) {
            KernelLaunch k;
            k.fn = traceMarkKernelFn();
            k.grid = dim3(1, 1, 1);
            k.block = dim3(256, 1, 1);
            k.setArgs(exec->stateDev, (uint32_t)event, node_id, func_id, invocations,
                      workgroups);
            k.name = "trace";
            k.role = "mark";
            k.kind = MWHIP_NODE_RECYCLE;
            return k;
        };
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, exec->cfg.gpu_id));
        std::vector<KernelLaunch> traced;
        traced.push_back(mark(DeviceEvent::calibration,
                              (uint32_t)prop.multiProcessorCount, 4u, 0u, 0u));
        uint32_t node_id = 0;
        for (const KernelLaunch &k : lg.launches) {
            const std::string label =
                k.role[0] != '\0' ? k.name + ":" + k.role : k.name;
            uint32_t func_id = 0;
            while (func_id < exec->traceNames.size() &&
                   exec->traceNames[func_id] != label) {
                func_id++;
            }
            if (func_id == exec->traceNames.size()) {
                exec->traceNames.push_back(label);
            }
            const uint32_t workgroups = k.grid.x * k.grid.y * k.grid.z;
            traced.push_back(mark(DeviceEvent::nodeStart, node_id, func_id,
                workgroups * k.block.x * k.block.y * k.block.z, workgroups));
            traced.push_back(k);
            node_id++;
        }
        traced.push_back(mark(DeviceEvent::blockExit, node_id, 0u, 0u, 0u));
        lg.launches = std::move(traced);
    }
    return 0;
}

#endif
// the graph objects and the device memory the graph owns (the stream has been
// waited for, or nothing of this graph is in flight)
MWHIP_RT void releaseLaunchGraph(LaunchGraph &lg)
{
    if (lg.graphExec) (void)hipGraphExecDestroy(lg.graphExec);
    if (lg.graph) (void)hipGraphDestroy(lg.graph);
    lg.graphExec = nullptr;
    lg.graph = nullptr;
    for (void *p : lg.ownedAllocations) {
        (void)hipFree(p);
    }
    lg.ownedAllocations.clear();
}

// ParallelFor nodes that named the same dependencies -- the simulator's statement
// that they do not depend on one another -- become ONE launch of the simulator's
// group kernel (taskgraph.inl pforGroupKernel: blockIdx.y = the node): a launch
// is ~4 us on this stack whatever it does, a group of k nodes saves k - 1 of
// them.  Only nodes that cannot append rows (no row snapshot) and carry a body;
// consecutive in the builder's order.  MADRONA_MWHIP_GROUP=0: every node its own
// launch (rounds 1-4).
static int groupLaunches(mwhip_exec *exec, LaunchGraph &lg)
{
    if (envU32("MADRONA_MWHIP_GROUP", 1) == 0 || exec->eagerReplay) {
        return 0;
    }
    // A member is reached through a function pointer: the shared kernel is
    // compiled for the registers of the heaviest body in the module and a body
    // runs out of line.  Row functions of a few loads and stores do not notice;
    // a long one does (the portable lidar system, 121 VGPRs, next to
    // collectObservations: step + 38 us, profiles/r05_group_variants.jsonl), so
    // a node whose own kernel needs more than this many VGPRs keeps its launch.
    const uint32_t max_vgprs = envU32("MADRONA_MWHIP_GROUP_MAX_VGPRS", 96);
    auto groupable = [&](const KernelLaunch &k) {
        return k.dagKernel && k.pforBody != nullptr &&
            k.pforGroupKernel != nullptr && !k.rowSnapshot &&
            k.pforVgprs <= max_vgprs &&
            k.countMode == MWHIP_COUNT_QUERY_ROWS && k.tgNode >= 0 &&
            k.tgId < exec->taskGraphs.size();
    };
    auto depsOf = [&](const KernelLaunch &k) {
        std::vector<int32_t> d = exec->taskGraphs[k.tgId].nodes[(size_t)k.tgNode].deps;
        std::sort(d.begin(), d.end());
        d.erase(std::unique(d.begin(), d.end()), d.end());
        return d;
    };
    // The reference runs same-dependency siblings one after the other, so a
    // simulator may lean on registration order without saying so.  What the
    // signatures show is checked here: two nodes do not share a launch when one
    // may write (non-const reference) a component the other names, on a table
    // both queries match.  (What a system reaches through ctx.get() is not in
    // its signature: INTEGRATION.md section 3, MADRONA_MWHIP_GROUP=0.)
    auto queryOf = [&](const KernelLaunch &k) -> const QueryRec * {
        for (const QueryRec &q : exec->queries) {
            if (q.offset == k.queryOffset) return &q;
        }
        return nullptr;
    };
    auto tablesOfQuery = [&](const QueryRec &q) {
        std::vector<uint32_t> tables;
        const uint32_t *p = exec->queryDataHost.data() + q.offset;
        for (uint32_t m = 0; m < q.numMatching; m++) {
            tables.push_back(p[0]);
            p += 1 + q.comps.size();
        }
        return tables;
    };
    auto conflicts = [&](const KernelLaunch &a, const KernelLaunch &b) {
        const QueryRec *qa = queryOf(a), *qb = queryOf(b);
        if (qa == nullptr || qb == nullptr) return true;
        bool shared_table = false;
        const std::vector<uint32_t> tb = tablesOfQuery(*qb);
        for (uint32_t t : tablesOfQuery(*qa)) {
            shared_table = shared_table ||
                std::find(tb.begin(), tb.end(), t) != tb.end();
        }
        if (!shared_table) return false;
        for (size_t ia = 0; ia < qa->comps.size(); ia++) {
            for (size_t ib = 0; ib < qb->comps.size(); ib++) {
                if (qa->comps[ia] != qb->comps[ib]) continue;
                const bool wa = ia >= 32 || ((a.pforWriteMask >> ia) & 1u) != 0u;
                const bool wb = ib >= 32 || ((b.pforWriteMask >> ib) & 1u) != 0u;
                if (wa || wb) return true;
            }
        }
        return false;
    };
    std::vector<KernelLaunch> out;
    for (size_t i = 0; i < lg.launches.size(); ) {
        size_t j = i + 1;
        if (groupable(lg.launches[i])) {
            const std::vector<int32_t> deps = depsOf(lg.launches[i]);
            while (j < lg.launches.size() && j - i < MWHIP_PFOR_GROUP_MAX &&
                   groupable(lg.launches[j]) &&
                   lg.launches[j].tgId == lg.launches[i].tgId &&
                   // (a body is only called from the group kernel of the code
                   // object that defines it)
                   lg.launches[j].pforGroupKernel == lg.launches[i].pforGroupKernel &&
                   depsOf(lg.launches[j]) == deps) {
                bool clash = false;
                for (size_t m = i; m < j; m++) {
                    clash = clash || conflicts(lg.launches[m], lg.launches[j]);
                }
                if (clash) break;
                j++;
            }
        }
        if (j - i < 2) {
            out.push_back(lg.launches[i]);
            i = j;
            continue;
        }
        mwhip_pfor_group group {};
        group.count = (uint32_t)(j - i);
        KernelLaunch g;
        g.fn = lg.launches[i].pforGroupKernel;
        uint32_t threads = 0;
        std::string name = "group[";
        for (size_t m = i; m < j; m++) {
            const KernelLaunch &k = lg.launches[m];
            group.body[m - i] = k.pforBody;
            group.query_offset[m - i] = k.queryOffset;
            group.num_matching_and_flags[m - i] = k.pforArg1;
            group.query[m - i] = k.pforArgs;
            threads = std::max(threads, k.grid.x * k.block.x);
            name += (m == i ? "" : " | ") + k.name;
            g.members.push_back({ k.name, k.queryOffset, k.bytesPerRow, k.ioDeclared });
        }
        name += "]";
        mwhip_pfor_group *group_dev = nullptr;
        int rc = devAllocT(exec, &group_dev, 1);
        if (rc != 0) return rc;
        HIPCHK(hipMemcpy(group_dev, &group, sizeof(group), hipMemcpyHostToDevice));
        g.grid = dim3(std::max((threads + 255u) / 256u, 1u), group.count, 1);
        g.block = dim3(256, 1, 1);
        g.setArgs(exec->stateDev, (const mwhip_pfor_group *)group_dev);
        g.name = name;
        g.role = "";
        g.kind = MWHIP_NODE_KERNEL;
        g.countMode = MWHIP_COUNT_QUERY_ROWS;
        g.ioDeclared = 1;
        for (const auto &mem : g.members) {
            g.ioDeclared = g.ioDeclared && mem.ioDeclared != 0u ? 1u : 0u;
        }
        g.nodeIndex = lg.launches[i].nodeIndex;
        // (waits for what its members named; everything that named a member
        // waits for it: the group stands in the chain where its first member
        // stood, the others were right behind it)
        g.dagKernel = false;
        out.push_back(g);
        i = j;
    }
    lg.launches.swap(out);
    return 0;
}

// Edges of the step's graph (KernelLaunch::deps): a chain in the builder's
// topological order.  (Round 5 also built the task graph's real edges as
// branches of the hipGraph and measured them slower than the chain on this
// runtime -- configs[1] 141 -> 170 us per step, configs[2] 1.110 -> 1.190 ms,
// profiles/r05_dag_variants.jsonl: a fork / join costs more than the launch
// floors it overlaps.  Nodes that may run side by side share ONE launch instead,
// groupLaunches.  The switch is gone since round 6.)
static void buildLaunchDeps(LaunchGraph &lg)
{
    for (size_t i = 0; i < lg.launches.size(); i++) {
        KernelLaunch &k = lg.launches[i];
        k.deps.clear();
        if (i > 0) {
            k.deps.push_back((int32_t)i - 1);
        }
    }
}

MWHIP_RT int instantiateLaunchGraph(mwhip_exec *exec,
                                  const std::vector<uint32_t> &ids,
                                  const std::string &stat_name,
                                  std::unique_ptr<LaunchGraph> &out,
                                  const LaunchGraph *pack_from)
{
    std::unique_ptr<LaunchGraph> lg(new LaunchGraph {});
    lg->statName = stat_name;
    lg->taskGraphIds = ids;
    struct ScopeOn {
        mwhip_exec *e;
        ScopeOn(mwhip_exec *x, std::vector<void *> *v) : e(x) { t_allocScope = v; }
        ~ScopeOn() { (void)e; t_allocScope = nullptr; }
    } scope_on(exec, &lg->ownedAllocations);

    // (ParallelFor grids are sized from the rows the tables hold now, not from
    // their capacities)
    {
        HIPCHK(hipStreamSynchronize(exec->stream));
        std::vector<TableHdr> hdrs(exec->tablesHost.size());
        HIPCHK(hipMemcpy(hdrs.data(), exec->hostState.tables,
                         hdrs.size() * sizeof(TableHdr), hipMemcpyDeviceToHost));
        exec->rowsAtGraphBuild.assign(hdrs.size(), 0);
        for (size_t a = 0; a < hdrs.size(); a++) {
            exec->rowsAtGraphBuild[a] =
                (uint32_t)std::max(hdrs[a].numRows, 0);
        }
    }

    lg->isRender = pack_from != nullptr && pack_from->isRender;
    int rc = buildLaunchList(exec, ids, *lg);
    if (rc != 0) return rc;

    if (lg->isRender) {
        // TLAS build + ray caster, before the health kernel
        std::vector<KernelLaunch> render;
        rc = renderLaunches(exec, render);
        if (rc != 0) return rc;
        lg->launches.insert(lg->launches.end() - 1, render.begin(), render.end());
    }

    if (pack_from != nullptr && pack_from->hasPack) {
        lg->hasPack = true;
        lg->pack = pack_from->pack;
        lg->packDst = pack_from->packDst;

        KernelLaunch k;
        k.fn = packRowsKernelFn();
        const uint64_t total =
            (uint64_t)lg->pack.numRows * lg->pack.recordWords;
        k.grid = dim3((uint32_t)std::min<uint64_t>(
            std::max<uint64_t>((total + 255) / 256, 1), 4096), 1, 1);
        k.block = dim3(256, 1, 1);
        k.setArgs(lg->pack, (uint32_t *)lg->packDst);
        k.name = "pack";
        k.role = "pack.rows";
        k.kind = MWHIP_NODE_RECYCLE;
        // before the health kernel that closes every replay
        lg->launches.insert(lg->launches.end() - 1, k);
    }

#ifdef MADRONA_TRACING
    rc = addTraceMarkers(exec, *lg);
    if (rc != 0) return rc;
    // (a traced step logs kernel after kernel: no side-by-side launches)
    for (KernelLaunch &k : lg->launches) {
        k.dagKernel = false;
    }
#endif

    rc = groupLaunches(exec, *lg);
    if (rc != 0) return rc;
    buildLaunchDeps(*lg);

    // The step as an explicit hipGraph: one kernel node per launch, edges from
    // buildLaunchDeps.  (Rounds 1-4 captured the launches from the stream: a
    // chain, whatever the task graph said.)
    HIPCHK(hipGraphCreate(&lg->graph, 0));
    std::vector<hipGraphNode_t> nodes(lg->launches.size(), nullptr);
    for (size_t i = 0; i < lg->launches.size(); i++) {
        KernelLaunch &k = lg->launches[i];
        void *args[8];
        k.argPointers(args);
        hipKernelNodeParams params {};
        params.func = const_cast<void *>(k.fn);
        params.gridDim = k.grid;
        params.blockDim = k.block;
        params.sharedMemBytes = k.dynamicLds;
        params.kernelParams = args;
        params.extra = nullptr;
        std::vector<hipGraphNode_t> deps;
        for (int32_t d : k.deps) {
            deps.push_back(nodes[(size_t)d]);
        }
        HIPCHK(hipGraphAddKernelNode(&nodes[i], lg->graph, deps.data(),
                                     deps.size(), &params));
    }
    HIPCHK(hipGraphInstantiate(&lg->graphExec, lg->graph, nullptr, nullptr, 0));
    out = std::move(lg);
    return 0;
}


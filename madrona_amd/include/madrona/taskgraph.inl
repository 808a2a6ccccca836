#pragma once

namespace madrona {

namespace mwhip {

// ---- algorithmic bytes of a system from its signature -----
// const T & / T by value = read once, T & = read + write; + 4 B WorldID/row.
template <typename ArgT>
constexpr uint32_t systemArgBytes()
{
    using BaseT = std::remove_cv_t<std::remove_reference_t<ArgT>>;
    if constexpr (std::is_lvalue_reference_v<ArgT> &&
                  !std::is_const_v<std::remove_reference_t<ArgT>>) {
        return 2u * (uint32_t)sizeof(BaseT);
    } else {
        return (uint32_t)sizeof(BaseT);
    }
}

template <typename FnT> struct SystemTraits {
    static constexpr uint32_t bytesPerRow = 0;
};

template <typename CtxT, typename... ArgTs>
struct SystemTraits<void (*)(CtxT &, ArgTs...)> {
    static constexpr uint32_t bytesPerRow =
        4u + (0u + ... + systemArgBytes<ArgTs>());
};

// ---- which of its query's components a system may write ---------------------
// bit i = argument i behind the context is a non-const lvalue reference.  Only
// meaningful when the signature has one argument per query component (in the
// query's order); anything else -- functors, batch systems -- is "all of them".
template <typename FnT, size_t NumComponents> struct SystemWriteMask {
    static constexpr uint32_t value = 0xFFFFFFFFu;
};

template <typename CtxT, typename... ArgTs, size_t NumComponents>
struct SystemWriteMask<void (*)(CtxT &, ArgTs...), NumComponents> {
    static constexpr uint32_t compute()
    {
        if constexpr (sizeof...(ArgTs) != NumComponents || NumComponents > 32) {
            return 0xFFFFFFFFu;
        } else {
            uint32_t mask = 0, bit = 0;
            ((mask |= (std::is_lvalue_reference_v<ArgTs> &&
                       !std::is_const_v<std::remove_reference_t<ArgTs>>) ?
                          (1u << bit) : 0u, bit++), ...);
            return mask;
        }
    }
    static constexpr uint32_t value = compute();
};

// ---- what a system ACTUALLY reads and writes per row -------
// "Each node declares its read/write set next to the kernel; when unknown,
// fall back to the signature rule."  A simulator declares it next to the
// system, outside its own namespace (as for systemWavesPerSIMD):
//
//   template <> inline constexpr madrona::mwhip::SystemIOBytes
//       madrona::mwhip::systemIO<escape::movementSystem> =
//           madrona::mwhip::declareIO<
//               madrona::mwhip::Reads<escape::Action, madrona::base::Rotation>,
//               madrona::mwhip::Writes<madrona::phys::ExternalForce,
//                                      madrona::phys::ExternalTorque>>();
//
// Reads: every component value the function loads for a row -- the query's
// columns it really looks at, and what it reaches through ctx.get(Entity) /
// getDirect(Loc) (size per access, not per cache line; a type named twice is
// counted twice, e.g. the Position of both agents); Entity if the signature
// takes it.  Writes: every component it stores (a T & that is only written
// appears in Writes alone; read and written: in both).  Shared read-only
// tables (ObjectManager, BVH nodes) are excluded, as in SURVEY §8d.  The node's
// algorithmic bytes are rows x (4 [WorldID] + reads + writes).
template <typename... Ts> struct Reads {
    static constexpr uint32_t bytes = (0u + ... + (uint32_t)sizeof(Ts));
};
template <typename... Ts> struct Writes {
    static constexpr uint32_t bytes = (0u + ... + (uint32_t)sizeof(Ts));
};
// n copies of T (an observation of n other entities, a lidar of n samples)
template <typename T, uint32_t n> struct Times { char bytes_[sizeof(T) * n]; };

struct SystemIOBytes {
    int32_t read;       // < 0: not declared, the signature rule applies
    int32_t write;
};

template <auto Fn>
inline constexpr SystemIOBytes systemIO = { -1, -1 };

template <typename ReadsT, typename WritesT>
constexpr SystemIOBytes declareIO()
{
    return SystemIOBytes { (int32_t)ReadsT::bytes, (int32_t)WritesT::bytes };
}

// ---- two systems over the same rows in ONE node ------
// rowChain<fnA, fnB, Ctx, Cs...> is a system over the components Cs... that runs
// fnA and then fnB on the row: each takes the components of Cs... its signature
// names.  For a pair the task graph would chain (fnB waits for fnA) where both
// only touch THEIR OWN row's components of Cs... -- whatever else they reach
// through ctx.get() must not be written by the other: the caller's claim, the
// read / write sets do not carry it.  One launch instead of two, and what fnA
// wrote for the row is still in registers when fnB reads it.
//
//   builder.addToGraph<ParallelForNode<Engine,
//       mwhip::rowChain<doorOpenSystem, setDoorPositionSystem,
//                       Engine, OpenState, DoorProperties, Position>,
//       OpenState, DoorProperties, Position>>({button_sys});
template <typename WantT, typename FirstT, typename... RestTs>
MADRONA_HD constexpr WantT & rowChainArg(FirstT &first, RestTs &...rest)
{
    if constexpr (std::is_same_v<WantT, FirstT>) {
        return first;
    } else {
        static_assert(sizeof...(RestTs) > 0,
            "rowChain: a system takes a component the node does not list");
        return rowChainArg<WantT>(rest...);
    }
}

template <typename FnT> struct RowChainCall;
template <typename CtxT, typename... ArgTs>
struct RowChainCall<void (*)(CtxT &, ArgTs...)> {
    template <auto Fn, typename... ComponentTs>
    MADRONA_HD static inline void call(CtxT &ctx, ComponentTs &...row)
    {
        Fn(ctx, rowChainArg<std::remove_cv_t<std::remove_reference_t<ArgTs>>>(
                    row...)...);
    }
};

// (a class, so that the systems' names are part of the function's: profiles
// label the node "chain[ns::fnA > ns::fnB]")
template <auto FnA, auto FnB, typename CtxT, typename... ComponentTs>
struct RowChain {
    MADRONA_HD static inline void run(CtxT &ctx, ComponentTs &...row)
    {
        RowChainCall<decltype(FnA)>::template call<FnA>(ctx, row...);
        RowChainCall<decltype(FnB)>::template call<FnB>(ctx, row...);
    }
};

template <auto FnA, auto FnB, typename CtxT, typename... ComponentTs>
inline constexpr auto rowChain = &RowChain<FnA, FnB, CtxT, ComponentTs...>::run;

// Pulls "ns::fnName" out of __PRETTY_FUNCTION__ of a function templated on
// <auto Fn> (host only, used to label kernels in profiles); a rowChain is
// "chain[ns::fnA > ns::fnB]".
template <auto Fn>
inline std::string systemName()
{
    std::string pretty = __PRETTY_FUNCTION__;
    size_t start = pretty.find("Fn = ");
    if (start == std::string::npos) {
        return "system";
    }
    start += 5;
    if (pretty[start] == '&') start++;
    auto plain = [&](size_t from) {
        if (pretty[from] == '&') from++;
        size_t end = pretty.find_first_of("];,<>", from);
        return pretty.substr(from, end - from);
    };
    const size_t chain = pretty.find("RowChain<", start);
    if (chain != std::string::npos) {
        const size_t first = chain + 9;
        const size_t comma = pretty.find(", ", first);
        if (comma != std::string::npos) {
            return "chain[" + plain(first) + " > " + plain(comma + 2) + "]";
        }
    }
    size_t end = pretty.find_first_of("];,", start);
    return pretty.substr(start, end - start);
}

#if defined(__HIPCC__)

template <typename ContextT, auto Fn, typename... ComponentTs, size_t... Is>
MADRONA_DEVICE inline void invokeSystemRow(ContextT &ctx, void *const *cols,
                                           int32_t row,
                                           std::index_sequence<Is...>)
{
    Fn(ctx, ((ComponentTs *)cols[Is])[row]...);
}

// One thread per matching row; rows of each matched archetype are walked with
// a grid-stride loop whose bound is the table's device-resident row count.
// Adjacent lanes touch adjacent rows of every SoA column (coalesced).
//
// The query resolution (table headers + column indices) arrives by value in
// the kernel-argument segment, so the dependent chain at kernel start is
// [row count + column pointers] -> [component data]: two round trips, not the
// five of ecs_state -> query table -> table -> column -> data.  These kernels
// are launch/latency bound at Escape-Room sizes (a few 1e4 rows).
//
// threads_per_invocation > 1 (CustomParallelForNode<..., T, 1, ...>): T
// consecutive lanes make the same call for one row and tell themselves apart
// with threadIdx.x % T, the reference's convention (device taskgraph.inl:
// 190-226) -- e.g. one ray per lane in a lidar system.
template <typename ContextT, auto Fn, int32_t threads_per_invocation,
          typename... ComponentTs>
MADRONA_DEVICE inline void parallelForTable(StateManager *state_mgr,
                                            TableHdr &tbl,
                                            const uint16_t *col_indices,
                                            bool exclusive_world,
                                            int32_t num_rows,
                                            int32_t rows_before)
{
    constexpr size_t N = sizeof...(ComponentTs);

    const int32_t tid = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x) /
        threads_per_invocation;
    const int32_t stride = (int32_t)(gridDim.x * blockDim.x) /
        threads_per_invocation;

    // The matched tables form ONE index space (rows_before = rows of the tables
    // ahead of this one): a thread continues in this table where the previous
    // one ended, so with fewer rows than threads every thread runs one row in
    // total -- one dependent chain (row -> world -> components -> ...), not
    // one per matched archetype.
#ifdef MADRONA_PFOR_PER_TABLE
    (void)rows_before;
    int32_t first_row = tid;
#else
    int32_t first_row = tid - rows_before % stride;
    if (first_row < 0) {
        first_row += stride;
    }
#endif
    if (first_row >= num_rows) {
        return;
    }

    const WorldID *world_col = (const WorldID *)tbl.columns[1];

    void *cols[N > 0 ? N : 1];
MADRONA_UNROLL
    for (size_t c = 0; c < N; c++) {
        cols[c] = tbl.columns[col_indices[c]];
    }

    for (int32_t row = first_row; row < num_rows; row += stride) {
        WorldID world_id = world_col[row];
        // destroyed but not yet compacted away
        if (world_id.idx == -1) {
            continue;
        }

        ContextT ctx = TaskGraph::makeContext<ContextT>(
            state_mgr, world_id, exclusive_world);
        invokeSystemRow<ContextT, Fn, ComponentTs...>(
            ctx, cols, row, std::make_index_sequence<N>());
    }
}

// Row counts of the matched tables, fixed once per launch (nodes whose system
// can append rows; mwhip_pfor_args::row_sync).  Workgroup 0 reads the counts
// and publishes them as {tag, rows} granules; every other workgroup waits for
// granules carrying this launch's tag -- replays completed so far + 1: a node
// runs once per replay.  A workgroup only starts running the system -- only
// then can it append rows -- after it has seen the published counts, so they
// are the counts from before the node.  No atomics: a ticket per workgroup on
// one address serialises (11 ns each unloaded on MI355X, profiles/tools/
// atomic_microbench.hip); workgroups are dispatched in index order, so
// workgroup 0 is never behind the ones waiting for it.
// out: dynamic LDS, one word per matched table.
template <typename TableOfFn>
MADRONA_DEVICE inline void pforRowSnapshot(EcsState *S, PforRowSync *sync,
                                           uint32_t num_tables,
                                           TableOfFn &&table_of,
                                           int32_t *out)
{
    if (threadIdx.x == 0) {
        const unsigned long long tag = (unsigned long long)(
            __hip_atomic_load(S->replayCounter, __ATOMIC_RELAXED,
                              __HIP_MEMORY_SCOPE_AGENT) + 1u);
        if (blockIdx.x == 0) {
            for (uint32_t a = 0; a < num_tables; a++) {
                int32_t n = __hip_atomic_load(&table_of(a)->numRows,
                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&sync->rows[a],
                    (tag << 32) | (unsigned long long)(uint32_t)n,
                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                out[a] = n;
            }
        } else {
            for (uint32_t a = 0; a < num_tables; a++) {
                unsigned long long g;
                // (bounded: the tag is the executor's replay count, so this
                // relies on the launch graphs of one executor running one at a
                // time -- mwhip.h, mwhip_run_async -- ; two of them in flight
                // on different streams would wait for a tag nobody publishes.
                // Then: the live row count and an error flag instead of a hang)
                uint32_t spins = 0;
                while (true) {
                    g = __hip_atomic_load(&sync->rows[a], __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_AGENT);
                    if ((g >> 32) == tag) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 24)) {
                        mwhip::raiseError(S, mwhip::kErrSortLookback);
                        g = (unsigned long long)(uint32_t)__hip_atomic_load(
                            &table_of(a)->numRows, __ATOMIC_RELAXED,
                            __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
                out[a] = (int32_t)(uint32_t)g;
            }
        }
    }
    __syncthreads();
}

// Waves per SIMD a system's kernel is compiled for, i.e. its register budget
// (512 / waves VGPRs per lane); 0 = the compiler's choice.  A system that is a
// chain of dependent loads with one wavefront per world (a BVH query, entity
// creation) and inlines enough code to take 150+ registers runs a quarter of
// the worlds at a time; a simulator caps it with
//   template <> inline constexpr unsigned madrona::mwhip::systemWavesPerSIMD<fn> = 4;
// before its setupTasks (sims/hideseek: the lock system, 109 -> 99 us).  Measured
// both ways: the lidar systems -- arithmetic heavy -- lose from more waves
// (profiles/r03_lidar_occupancy_variants.jsonl).
template <auto Fn>
inline constexpr unsigned systemWavesPerSIMD = 0;

// ---- a node's body as a __device__ function ----------------------------------
// Nodes that named the same dependencies and cannot append rows are run side by
// side in ONE launch (pforGroupKernel below, blockIdx.y = the node): a launch is
// ~4 us on this stack whatever it does (an empty kernel event-times at 4.0 us in
// the step graph), and branches of a hipGraph cost more than
// they save (15.7).  The group kernel reaches a node through the address of this
// function (mwhip_node_desc::pfor_body, handed out by the kernel's report mode).
using PforBodyFn = void (*)(EcsState *, uint32_t, uint32_t, const mwhip_pfor_args *);

template <typename ContextT, auto Fn, int32_t threads_per_invocation,
          typename... ComponentTs>
__device__ __attribute__((noinline)) void parallelForBody(
    EcsState *S, uint32_t query_offset, uint32_t num_matching_and_flags,
    const mwhip_pfor_args *query_ptr)
{
    constexpr size_t N = sizeof...(ComponentTs);
    const uint32_t num_matching = num_matching_and_flags & 0x7FFFFFFFu;
    const bool exclusive_world = (num_matching_and_flags >> 31) != 0u;
    StateManager *state_mgr = static_cast<StateManager *>(S);
    const mwhip_pfor_args &query = *query_ptr;

    // (never with a row snapshot: nodes that can append rows are not grouped)
    if (query.num_inline == num_matching) {
        int32_t num_rows[MWHIP_PFOR_MAX_INLINE];
MADRONA_UNROLL
        for (uint32_t a = 0; a < MWHIP_PFOR_MAX_INLINE; a++) {
            num_rows[a] = a >= num_matching ? 0 :
                ((const TableHdr *)query.tables[a])->numRows;
        }
        int32_t rows_before = 0;
MADRONA_UNROLL
        for (uint32_t a = 0; a < MWHIP_PFOR_MAX_INLINE; a++) {
            if (a < num_matching) {
                parallelForTable<ContextT, Fn, threads_per_invocation,
                                 ComponentTs...>(
                    state_mgr, *(TableHdr *)query.tables[a], query.columns[a],
                    exclusive_world, num_rows[a], rows_before);
                rows_before += num_rows[a];
            }
        }
        return;
    }
    const uint32_t *query_values = S->queryData + query_offset;
    for (uint32_t a = 0; a < num_matching; a++) {
        uint16_t col_indices[N > 0 ? N : 1];
MADRONA_UNROLL
        for (size_t c = 0; c < N; c++) {
            col_indices[c] = (uint16_t)query_values[1 + c];
        }
        TableHdr &tbl = S->tables[query_values[0]];
        parallelForTable<ContextT, Fn, threads_per_invocation,
                         ComponentTs...>(
            state_mgr, tbl, col_indices, exclusive_world, tbl.numRows, 0);
        query_values += 1 + N;
    }
}

// blockIdx.y = member of the group; every member strides over its rows with the
// launch's x grid
template <int Unused = 0>     // (a template: one definition however often included)
__global__ void __launch_bounds__(256)
pforGroupKernel(EcsState *S, const mwhip_pfor_group *group)
{
    TraceScope trace_scope(S);
    const uint32_t m = blockIdx.y;
    const PforBodyFn body = (PforBodyFn)loadInvariant(&group->body[m]);
    body(S, loadInvariant(&group->query_offset[m]),
         loadInvariant(&group->num_matching_and_flags[m]), &group->query[m]);
}

template <typename ContextT, auto Fn, int32_t threads_per_invocation,
          typename... ComponentTs>
__global__ void __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(systemWavesPerSIMD<Fn>)))
parallelForKernel(EcsState *S, void *report_to, uint32_t query_offset,
                  uint32_t num_matching_and_flags, mwhip_pfor_args query)
{
    if (num_matching_and_flags == 0xFFFFFFFFu) {
        // report mode (mwhip_pfor_body): where this node's body lives
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            *(PforBodyFn *)report_to = &parallelForBody<
                ContextT, Fn, threads_per_invocation, ComponentTs...>;
        }
        return;
    }
    TraceScope trace_scope(S);
    constexpr size_t N = sizeof...(ComponentTs);
    extern __shared__ int32_t pfor_snapshot_rows[];

    // bit 31: every matched archetype is a singleton => one thread per world
    const uint32_t num_matching = num_matching_and_flags & 0x7FFFFFFFu;
    const bool exclusive_world = (num_matching_and_flags >> 31) != 0u;

    StateManager *state_mgr = static_cast<StateManager *>(S);

    if (query.num_inline == num_matching) {
        int32_t num_rows[MWHIP_PFOR_MAX_INLINE];
        if (query.row_sync != nullptr) {
            pforRowSnapshot(S, (PforRowSync *)query.row_sync, num_matching,
                [&](uint32_t a) { return (TableHdr *)query.tables[a]; },
                pfor_snapshot_rows);
        }
MADRONA_UNROLL
        for (uint32_t a = 0; a < MWHIP_PFOR_MAX_INLINE; a++) {
            num_rows[a] = a >= num_matching ? 0 :
                (query.row_sync != nullptr ? pfor_snapshot_rows[a] :
                 ((const TableHdr *)query.tables[a])->numRows);
        }

        int32_t rows_before = 0;
MADRONA_UNROLL
        for (uint32_t a = 0; a < MWHIP_PFOR_MAX_INLINE; a++) {
            if (a < num_matching) {
                parallelForTable<ContextT, Fn, threads_per_invocation,
                                 ComponentTs...>(
                    state_mgr, *(TableHdr *)query.tables[a], query.columns[a],
                    exclusive_world, num_rows[a], rows_before);
                rows_before += num_rows[a];
            }
        }
        return;
    }

    // general path: walk the query table
    const uint32_t *query_values = S->queryData + query_offset;
    if (query.row_sync != nullptr) {
        pforRowSnapshot(S, (PforRowSync *)query.row_sync, num_matching,
            [&](uint32_t a) {
                return &S->tables[query_values[a * (1 + N)]];
            }, pfor_snapshot_rows);
    }
    for (uint32_t a = 0; a < num_matching; a++) {
        uint16_t col_indices[N > 0 ? N : 1];
MADRONA_UNROLL
        for (size_t c = 0; c < N; c++) {
            col_indices[c] = (uint16_t)query_values[1 + c];
        }
        TableHdr &tbl = S->tables[query_values[0]];
        parallelForTable<ContextT, Fn, threads_per_invocation,
                         ComponentTs...>(
            state_mgr, tbl, col_indices, exclusive_world,
            query.row_sync != nullptr ? pfor_snapshot_rows[a] : tbl.numRows, 0);
        query_values += 1 + N;
    }
}

// N-items-per-invocation convention of CustomParallelForNode
// (reference device taskgraph.inl:229-266): Fn(WorldID *, Cs *..., count).
template <auto Fn, int32_t items_per_invocation, typename... ComponentTs,
          size_t... Is>
MADRONA_DEVICE inline void invokeSystemBatch(WorldID *worlds, void *const *cols,
                                             int32_t row, int32_t count,
                                             std::index_sequence<Is...>)
{
    Fn(worlds + row, ((ComponentTs *)cols[Is] + row)..., count);
}

template <auto Fn, int32_t threads_per_invocation,
          int32_t items_per_invocation, typename... ComponentTs>
__global__ void __launch_bounds__(256)
parallelForBatchKernel(EcsState *S, void *, uint32_t query_offset,
                       uint32_t num_matching_and_flags, mwhip_pfor_args query)
{
    TraceScope trace_scope(S);
    constexpr size_t N = sizeof...(ComponentTs);
    extern __shared__ int32_t pfor_snapshot_rows[];
    const uint32_t num_matching = num_matching_and_flags & 0x7FFFFFFFu;
    const uint32_t *query_values = S->queryData + query_offset;

    if (query.row_sync != nullptr) {
        pforRowSnapshot(S, (PforRowSync *)query.row_sync, num_matching,
            [&](uint32_t a) {
                return &S->tables[query_values[a * (1 + N)]];
            }, pfor_snapshot_rows);
    }

    // every lane of a threads_per_invocation group makes the same call; the
    // user function differentiates lanes itself (threadIdx.x % group size)
    const int32_t tid = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    const int32_t group = tid / threads_per_invocation;
    const int32_t num_groups =
        (int32_t)(gridDim.x * blockDim.x) / threads_per_invocation;

    for (uint32_t a = 0; a < num_matching; a++) {
        TableHdr &tbl = S->tables[query_values[0]];
        const int32_t num_rows = query.row_sync != nullptr ?
            pfor_snapshot_rows[a] : tbl.numRows;

        void *cols[N > 0 ? N : 1];
MADRONA_UNROLL
        for (size_t c = 0; c < N; c++) {
            cols[c] = tbl.columns[query_values[1 + c]];
        }

        for (int32_t base = group * items_per_invocation; base < num_rows;
             base += num_groups * items_per_invocation) {
            int32_t count = num_rows - base;
            if (count > items_per_invocation) count = items_per_invocation;
            invokeSystemBatch<Fn, items_per_invocation, ComponentTs...>(
                (WorldID *)tbl.columns[1], cols, base, count,
                std::make_index_sequence<N>());
        }

        query_values += 1 + N;
    }
}

// Custom node: (node->*fn)(invocation_idx) for invocation_idx < count, where
// count is fixed or NodeT::numInvocations() evaluated on the device.
template <typename NodeT, auto fn, bool dynamic_count>
__global__ void __launch_bounds__(256)
customNodeKernel(EcsState *S, void *node_data, uint32_t fixed_count, uint32_t)
{
    TraceScope trace_scope(S);
    NodeT *node = (NodeT *)node_data;

    uint32_t count = fixed_count;
    if constexpr (dynamic_count) {
        count = node->numInvocations();
    }

    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = tid; i < count; i += stride) {
        std::invoke(fn, node, (int32_t)i);
    }
}

#endif // __HIPCC__

}

// ---------------------------------------------------------------------------
// Builder (host API: device copies are stubs, see MADRONA_HOST_API)
// ---------------------------------------------------------------------------

TaskGraph::Builder::Builder(mwhip_exec *exec, StateManager *state_mgr,
                            uint32_t taskgraph_id)
    : exec_(exec), state_mgr_(state_mgr), taskgraph_id_(taskgraph_id)
{}

template <typename NodeT, typename... Args>
MADRONA_HOST_API TaskGraph::TypedDataID<NodeT>
TaskGraph::Builder::constructNodeData(Args &&...args)
{
    static_assert(sizeof(NodeT) <= maxNodeDataBytes);
    static_assert(alignof(NodeT) <= alignof(NodeData));
    static_assert(std::is_trivially_copyable_v<NodeT>,
        "node data is copied to the device byte for byte");

#if MADRONA_ON_HOST
    int32_t data_idx = (int32_t)node_datas_.size();
    node_datas_.emplace_back(new NodeData {});
    node_data_bytes_.push_back((uint32_t)sizeof(NodeT));
    new (node_datas_.back()->userData) NodeT(std::forward<Args>(args)...);

    return TypedDataID<NodeT> { DataID { data_idx } };
#else
    MADRONA_DEVICE_STUB();
#endif
}

template <typename NodeT>
MADRONA_HOST_API NodeT &TaskGraph::Builder::getDataRef(
    TypedDataID<NodeT> data_id)
{
#if MADRONA_ON_HOST
    return *(NodeT *)node_datas_[data_id.id]->userData;
#else
    MADRONA_DEVICE_STUB();
#endif
}

TaskGraph::NodeID TaskGraph::Builder::addRuntimeNode(
    const mwhip_node_desc &desc, int32_t staged_data_idx,
    Span<const NodeID> dependencies)
{
    StagedNode staged;
    staged.desc = desc;
    staged.name = desc.name != nullptr ? desc.name : "node";
    staged.dataIdx = staged_data_idx;
    for (NodeID dep : dependencies) {
        staged.deps.push_back(dep.id);
    }

    staged_.push_back(std::move(staged));
    return NodeID { (int32_t)staged_.size() - 1 };
}

template <auto fn, typename NodeT>
MADRONA_HOST_API TaskGraph::NodeID TaskGraph::Builder::addNodeFn(
    TypedDataID<NodeT> data,
    Span<const NodeID> dependencies,
    Optional<NodeID>,
    uint32_t fixed_num_invocations,
    uint32_t num_threads_per_invocation)
{
    // a __host__ lambda may name a __global__ function; defining it here also
    // instantiates the kernel in the device compilation pass
#if defined(__HIPCC__)
    [[maybe_unused]] auto kernel_stub = [] __host__ () -> const void * {
        return (const void *)&mwhip::customNodeKernel<NodeT, fn, false>;
    };
#else
    auto kernel_stub = []() -> const void * { return nullptr; };
#endif

#if MADRONA_ON_HOST
    mwhip_node_desc desc {};
    desc.kind = MWHIP_NODE_KERNEL;
    std::string name = "custom:" + mwhip::systemName<fn>();
    desc.name = name.c_str();
    desc.kernel = kernel_stub();
    desc.count_mode = MWHIP_COUNT_FIXED;
    desc.fixed_count = fixed_num_invocations;
    desc.arg0 = fixed_num_invocations;
    desc.threads_per_invocation = num_threads_per_invocation;

    return addRuntimeNode(desc, data.id, dependencies);
#else
    MADRONA_DEVICE_STUB();
#endif
}

template <typename NodeT, int32_t count, typename... Args>
MADRONA_HOST_API TaskGraph::NodeID TaskGraph::Builder::addOneOffNode(
    Span<const NodeID> dependencies, Args &&...args)
{
    auto data_id = constructNodeData<NodeT>(std::forward<Args>(args)...);
    return addNodeFn<&NodeT::run>(data_id, dependencies,
                                  Optional<NodeID>::none(), count);
}

template <typename NodeT, typename... Args>
MADRONA_HOST_API TaskGraph::NodeID TaskGraph::Builder::addDynamicCountNode(
    Span<const NodeID> dependencies,
    uint32_t num_threads_per_invocation,
    Args &&...args)
{
#if defined(__HIPCC__)
    [[maybe_unused]] auto kernel_stub = [] __host__ () -> const void * {
        return (const void *)&mwhip::customNodeKernel<NodeT, &NodeT::run, true>;
    };
#else
    auto kernel_stub = []() -> const void * { return nullptr; };
#endif

#if MADRONA_ON_HOST
    auto data_id = constructNodeData<NodeT>(std::forward<Args>(args)...);

    mwhip_node_desc desc {};
    desc.kind = MWHIP_NODE_KERNEL;
    std::string name = "dynamic:" + mwhip::systemName<&NodeT::run>();
    desc.name = name.c_str();
    desc.kernel = kernel_stub();
    // the count is only known on the device: launch a full grid
    desc.count_mode = MWHIP_COUNT_FIXED;
    desc.fixed_count = 0xFFFFFFFFu;
    desc.threads_per_invocation = num_threads_per_invocation;

    return addRuntimeNode(desc, data_id.id, dependencies);
#else
    MADRONA_DEVICE_STUB();
#endif
}

template <typename NodeT>
MADRONA_HOST_API TaskGraph::NodeID TaskGraph::Builder::addToGraph(
    Span<const NodeID> dependencies)
{
    return NodeT::addToGraph(*this, dependencies);
}

void TaskGraph::Builder::flush()
{
    std::vector<int32_t> data_ids(node_datas_.size(), -1);
    for (size_t i = 0; i < node_datas_.size(); i++) {
        data_ids[i] = mwhip_tg_add_node_data(exec_, taskgraph_id_,
            node_datas_[i]->userData, node_data_bytes_[i]);
        mwhip::check(data_ids[i], "tg_add_node_data");
    }

    for (StagedNode &staged : staged_) {
        staged.desc.name = staged.name.c_str();
        staged.desc.node_data_id =
            staged.dataIdx >= 0 ? data_ids[staged.dataIdx] : -1;
        int32_t id = mwhip_tg_add_node(exec_, taskgraph_id_, &staged.desc,
            staged.deps.data(), (uint32_t)staged.deps.size());
        mwhip::check(id, "tg_add_node");
    }

    staged_.clear();
}

// ---------------------------------------------------------------------------
// TaskGraphManager
// ---------------------------------------------------------------------------

TaskGraphManager::TaskGraphManager(mwhip_exec *exec, StateManager *state_mgr,
                                   uint32_t num_taskgraphs)
    : exec_(exec), state_mgr_(state_mgr), builders_(num_taskgraphs)
{}

MADRONA_HOST_API TaskGraphBuilder &TaskGraphManager::init(uint32_t taskgraph_id)
{
#if MADRONA_ON_HOST
    builders_[taskgraph_id].reset(
        new TaskGraphBuilder(exec_, state_mgr_, taskgraph_id));
    return *builders_[taskgraph_id];
#else
    MADRONA_DEVICE_STUB();
#endif
}

void TaskGraphManager::constructGraphs()
{
    for (auto &builder : builders_) {
        if (builder) {
            builder->flush();
        }
    }
}

// ---------------------------------------------------------------------------
// Built-in nodes
// ---------------------------------------------------------------------------

template <typename ContextT, auto Fn,
          int32_t threads_per_invocation,
          int32_t items_per_invocation,
          typename... ComponentTs>
MADRONA_HOST_API TaskGraph::NodeID
CustomParallelForNode<ContextT, Fn, threads_per_invocation,
                      items_per_invocation, ComponentTs...>::addToGraph(
    TaskGraph::Builder &builder,
    Span<const TaskGraph::NodeID> dependencies)
{
#if defined(__HIPCC__)
    [[maybe_unused]] auto kernel_stub = [] __host__ () -> const void * {
        if constexpr (items_per_invocation == 1) {
            static_assert(threads_per_invocation >= 1 &&
                64 % threads_per_invocation == 0,
                "threads per invocation must divide the wavefront");
            return (const void *)&mwhip::parallelForKernel<
                ContextT, Fn, threads_per_invocation, ComponentTs...>;
        } else {
            return (const void *)&mwhip::parallelForBatchKernel<
                Fn, threads_per_invocation, items_per_invocation,
                ComponentTs...>;
        }
    };
    // (this module's group kernel: side-by-side nodes in one launch)
    [[maybe_unused]] auto group_stub = [] __host__ () -> const void * {
        return (const void *)&mwhip::pforGroupKernel<0>;
    };
#else
    auto kernel_stub = []() -> const void * { return nullptr; };
    [[maybe_unused]] auto group_stub = []() -> const void * { return nullptr; };
#endif

#if MADRONA_ON_HOST
    auto query = builder.stateManager().template query<ComponentTs...>();
    const QueryRef *ref = query.getSharedRef();

    mwhip_node_desc desc {};
    desc.kind = MWHIP_NODE_KERNEL;
    std::string name = mwhip::systemName<Fn>();
    desc.name = name.c_str();
    desc.kernel = kernel_stub();
    desc.node_data_id = -1;
    desc.arg0 = ref->offset;
    // exclusive_world (lock-free id cache): one lane per world, i.e. only
    // singleton archetypes AND one thread per invocation -- with several
    // lanes per world create / destroy must take the world's lock
    desc.arg1 = ref->numMatchingArchetypes |
        (((ref->flags & MWHIP_QUERY_ALL_SINGLETON) != 0u &&
          threads_per_invocation == 1) ? 0x80000000u : 0u);
    desc.count_mode = MWHIP_COUNT_QUERY_ROWS;
    desc.wants_pfor_args = 1u;
    desc.query_offset = ref->offset;
    desc.num_matching = ref->numMatchingArchetypes;
    desc.threads_per_invocation = (uint32_t)threads_per_invocation;
#if defined(__HIPCC__)
    if constexpr (items_per_invocation == 1) {
        // the node's body for grouped launches, and this module's group kernel
        desc.pfor_body = mwhip_pfor_body(builder.exec(), desc.kernel);
        desc.pfor_group_kernel = group_stub();
        (void)mwhip_set_pfor_group_kernel(builder.exec(), group_stub());
    }
#endif

    // (what the runtime checks before it lets two nodes share a launch)
    desc.write_mask = mwhip::SystemWriteMask<
        std::decay_t<decltype(Fn)>, sizeof...(ComponentTs)>::value;

    if constexpr (mwhip::systemIO<Fn>.read >= 0) {
        // declared next to the system
        desc.bytes_per_row = 4u + (uint32_t)mwhip::systemIO<Fn>.read +
            (uint32_t)mwhip::systemIO<Fn>.write;
        desc.io_declared = 1u;
    } else if constexpr (items_per_invocation == 1) {
        desc.bytes_per_row = mwhip::SystemTraits<decltype(Fn)>::bytesPerRow;
    } else {
        desc.bytes_per_row = 4u + (0u + ... + (uint32_t)sizeof(ComponentTs));
    }

    return builder.addRuntimeNode(desc, -1, dependencies);
#else
    MADRONA_DEVICE_STUB();
#endif
}

namespace mwhip {

MADRONA_HOST_API inline TaskGraph::NodeID addSimpleNode(
    TaskGraph::Builder &builder, Span<const TaskGraph::NodeID> dependencies,
    uint32_t kind, const char *name, uint32_t archetype_id,
    uint32_t component_id)
{
#if MADRONA_ON_HOST
    mwhip_node_desc desc {};
    desc.kind = kind;
    desc.name = name;
    desc.node_data_id = -1;
    desc.archetype_id = archetype_id;
    desc.component_id = component_id;
    return builder.addRuntimeNode(desc, -1, dependencies);
#else
    MADRONA_DEVICE_STUB();
#endif
}

}

MADRONA_HOST_API TaskGraph::NodeID ClearTmpNodeBase::addToGraph(
    TaskGraph::Builder &builder,
    Span<const TaskGraph::NodeID> dependencies,
    uint32_t archetype_id)
{
    return mwhip::addSimpleNode(builder, dependencies, MWHIP_NODE_CLEAR_TMP,
                                "ClearTmp", archetype_id, 0);
}

MADRONA_HOST_API TaskGraph::NodeID RecycleEntitiesNode::addToGraph(
    TaskGraph::Builder &builder,
    Span<const TaskGraph::NodeID> dependencies)
{
    return mwhip::addSimpleNode(builder, dependencies, MWHIP_NODE_RECYCLE,
                                "RecycleEntities", 0, 0);
}

MADRONA_HOST_API TaskGraph::NodeID ResetTmpAllocNode::addToGraph(
    TaskGraph::Builder &builder,
    Span<const TaskGraph::NodeID> dependencies)
{
    return mwhip::addSimpleNode(builder, dependencies,
                                MWHIP_NODE_RESET_TMP_ALLOC, "ResetTmpAlloc",
                                0, 0);
}

MADRONA_HOST_API TaskGraph::NodeID SortArchetypeNodeBase::addToGraph(
    TaskGraph::Builder &builder,
    Span<const TaskGraph::NodeID> dependencies,
    uint32_t archetype_id,
    int32_t component_id)
{
    return mwhip::addSimpleNode(builder, dependencies,
                                MWHIP_NODE_SORT_ARCHETYPE, "SortArchetype",
                                archetype_id, (uint32_t)component_id);
}

}

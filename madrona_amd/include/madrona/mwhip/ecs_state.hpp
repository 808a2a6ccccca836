// Device-resident ECS state of the MI355X backend + the device-side entity /
// row allocation primitives.  Shared between libmadrona_hip.so (runtime, sort
// kernels) and the header overlay compiled into the simulator's HIP TU.
//
// Layout decisions:
//  * ONE global SoA table per archetype for all worlds, column 0 = Entity,
//    column 1 = WorldID, then user components -- the contract of the
//    reference GPU backend (src/mw/device/state.cpp:163-341,
//    device/include/madrona/table.hpp:18-40).
//  * every column has a ping-pong twin (columnsAlt): the sort node gathers
//    out-of-place and swaps pointers instead of staging + copying back
//    (SURVEY.md Appendix C / D4).
//  * entity ids are handed out from per-world caches of 64-id blocks that
//    mirror the reference CPU backend's IDMap (include/madrona/impl/
//    id_map_impl.inl:69-260) so that entity ids are bit-identical to the CPU
//    oracle, instead of the reference GPU backend's global fetch_add
//    (device/state.cpp:442-527) which makes ids scheduling dependent.
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MWHIP_HD __host__ __device__
#define MWHIP_DEV __device__
#else
#define MWHIP_HD
#define MWHIP_DEV
#endif

namespace madrona {

struct Entity;
struct Loc;

namespace mwhip {

inline constexpr uint32_t kMaxColumns = 128;       // reference table.hpp:21
inline constexpr uint32_t kMaxArchetypes = 256;    // reference state.hpp:204
inline constexpr uint32_t kMaxComponents = 1024;   // reference state.hpp:203
inline constexpr int32_t kIdsPerBlock = 64;        // reference id_map.hpp:134
inline constexpr int32_t kIdSentinel = -1;         // 0xFFFF'FFFF_i32
inline constexpr uint16_t kNoColumn = 0xFFFF;
inline constexpr uint32_t kBundleMask = 0x80000000u;

enum ColumnFlags : uint32_t {
    kColumnPinned = 1u << 0,    // exported: address must stay fixed across sorts
};

// Device-side error flags (EcsState::errorFlags, sticky, reported by the health
// kernel at the end of every replay: mwhip_run / mwhip_synchronize return -10
// and mwhip_last_error() names the flag).  The reference aborts the process in
// these situations (FATAL / assert); here the replay COMPLETES and what it left
// behind is partly stale -- after an error the exported tensors must not be
// fed to a trainer:
//   kErrTableOverflow   an append found its table full and no memory could be
//                       mapped behind it in time: the entity was not created
//                       (makeEntity returned Entity::none()); every tensor of
//                       the worlds that asked is short of those rows.  In the
//                       physics step: a world's candidate / contact budget ran
//                       out -- that WORLD skipped its whole physics step.
//   kErrPhysics         a rigid-body table was not grouped by world, a world has
//                       more bodies than the step kernel was built for, or a
//                       primitive pair is unsupported: the world skipped its
//                       physics step for this replay.  Its Position, Rotation,
//                       Velocity and solver-state columns hold the values of
//                       the replay before; everything computed after the
//                       physics nodes (observations, lidar, rewards, render
//                       instance records) is computed from those stale poses.
//                       Other worlds are unaffected.
//   kErrRender          a world has more than 1024 instances / the instance
//                       table was unsorted / a traversal stack overflowed: the
//                       rgb / depth outputs of that world's views are stale or
//                       partial.
//   the others          (entity store, tmpAlloc, constructor blocks, sort
//                       look-back) leave the tables in an undefined state:
//                       discard the executor.
enum ErrorFlags : uint32_t {
    kErrTableOverflow = 1u << 0,
    kErrEntityOverflow = 1u << 1,
    kErrTmpOverflow = 1u << 2,
    kErrInitBlocks = 1u << 3,
    kErrSortLookback = 1u << 4,
    kErrPersistOverflow = 1u << 5,
    kErrPhysics = 1u << 6,
    kErrRender = 1u << 7,
};

struct TableHdr {
    void *columns[kMaxColumns];
    void *columnsAlt[kMaxColumns];
    uint32_t columnBytes[kMaxColumns];
    uint32_t columnFlags[kMaxColumns];
    uint16_t columnComponent[kMaxColumns];  // component id stored in each column
    int32_t numColumns;
    int32_t numRows;            // appended to with agent-scope atomics
    int32_t capacity;           // rows mapped in every column
    uint32_t needsSort;
    int32_t *worldOffsets;      // [numWorlds]
    int32_t *worldCounts;       // [numWorlds]
    uint32_t maxPerWorld;
    uint32_t registered;
    uint32_t rowBytes;          // sum of columnBytes
    // most rows the table held at any point of the running step (rows destroyed
    // and re-created in one step coexist until its compaction): what table
    // growth is sized by.  Raised where rows are dropped (sort, ClearTmp), read
    // and reset by the end-of-replay health kernel.
    int32_t peakRows;
    // Rows [0, sortedRows) are what the last world sort left: grouped by world
    // in world order, except that some have been destroyed in place since
    // (WorldID -1); everything appended since sits behind them.  Set where a
    // world sort publishes the table, zeroed by whatever else reorders or
    // truncates it (a sort by another key, ClearTmp).  The compaction path of
    // the sort node (csrc/sort_archetype.hip) starts from it.
    int32_t sortedRows;
    // rows behind the sorted prefix at the largest world sort of the running
    // step (health kernel: read and reset)
    int32_t tailRows;
};

// == IDMap::Node with V = Loc (reference impl/id_map.hpp:41-53): a live slot
// stores the entity's Loc, a free slot stores the free-list links.
struct EntitySlot {
    union {
        struct { uint32_t archetype; int32_t row; } loc;
        struct { int32_t subNext; int32_t globalNext; } freeNode;
    };
    uint32_t gen;
};

// == IDMap::Cache (reference impl/id_map.hpp:24-36), one per world, plus a
// lock for the (rare) case of several threads of one world creating /
// destroying entities inside the same ParallelFor node.
struct IdCache {
    int32_t freeHead;
    int32_t numFree;
    int32_t overflowHead;
    int32_t numOverflow;
    uint32_t lock;
    int32_t initBlocksUsed;   // blocks taken from the global store during world init
    int32_t runtimeBlocksUsed;// blocks taken after init (static per-world partition)
    uint32_t pad_;
};

// Device -> host message ring of mwGPU::HostPrint (madrona/mw_gpu/
// host_print.hpp), in pinned host memory mapped into the device.
struct HostPrintRecord {
    static constexpr int32_t maxArgs = 12;
    static constexpr int32_t maxChars = 120;
    enum Type : uint8_t { I32, U32, I64, U64, Float, Ptr };

    unsigned long long seq;         // 1 + ticket once the record is complete
    unsigned long long args[maxArgs];
    uint32_t numArgs;
    uint8_t types[maxArgs];
    char fmt[maxChars];
};

struct HostPrintRing {
    static constexpr uint32_t numRecords = 1024;

    unsigned long long head;        // next ticket (device, system-scope atomics)
    unsigned long long tail;        // tickets below this have been printed (host)
    unsigned long long dropped;
    unsigned long long pad_;
    HostPrintRecord records[numRecords];
};

// Device -> host requests for table memory, in pinned host memory.  A thread
// that appends past what is mapped posts the rows it needs and waits for the
// executor's service thread to map more behind the table's columns (addresses
// do not change); the reference's device code asks its host thread the same
// way (src/mw/device/memory.cpp:27-121, src/mw/cuda_exec.cpp:1603-1719).
// (two more slots after the archetypes': the entity store, in ids, and the
// per-step scratch region of Context::tmpAlloc, in KiB)
inline constexpr uint32_t kGrowSlotEntities = kMaxArchetypes;
inline constexpr uint32_t kGrowSlotTmp = kMaxArchetypes + 1;
inline constexpr uint32_t kGrowSlots = kMaxArchetypes + 2;

struct GrowMailbox {
    int32_t capacity[kGrowSlots];       // rows mapped; written by the host
    int32_t requested[kGrowSlots];      // rows wanted; raised by the device
    uint32_t serviceEnabled;            // a host thread is answering
    // diagnostics of the first append that gave up: archetype, row, rows mapped
    int32_t failedArchetype;
    int32_t failedRow;
    int32_t failedCapacity;
};

struct EcsState {
    TableHdr *tables;               // [numArchetypeSlots]
    uint16_t *colLookup;            // [numArchetypeSlots * numComponentSlots]
    // [numArchetypeSlots * numComponentSlots] current base address of the
    // column holding that component (nullptr if absent): Context::get needs
    // one dependent load after the entity slot instead of two.  Kept in sync
    // by sortFinalize when it swaps ping-pong buffers.
    void **colPtr;
    uint32_t *queryData;
    EntitySlot *entities;
    IdCache *worldCaches;           // [numWorlds]
    int32_t *initBlockBase;         // [numWorlds] first id of world's init blocks (pass 2)
    char *worldData;
    char *tmpBase;
    // never-freed allocations made by world constructors (e.g. the physics
    // BVH arrays; reference: device malloc heap, src/mw/cuda_exec.cpp:267-285)
    char *persistBase;

    unsigned long long tmpCapacity;
    unsigned long long tmpOffset;   // bump pointer
    unsigned long long persistCapacity;
    unsigned long long persistOffset;
    unsigned long long idFreeHead;  // {gen:32 | head:32} global list of returned blocks

    uint32_t numArchetypeSlots;
    uint32_t numComponentSlots;
    uint32_t worldDataStride;
    int32_t numWorlds;
    int32_t entityCapacity;
    int32_t numIds;                 // end of the ids handed out during init (multiple of 64)
    uint32_t initMode;              // 0 run, 1 init pass (count), 2 init pass (assign)
    uint32_t errorFlags;

    void *hostExec;                 // host mirror only: owning mwhip_exec*
    int32_t runtimeIdBase;          // first id of the post-init block partition
    int32_t pad_;
    void *moduleData[4];            // module-private device pointers (physics scratch, ...)
    HostPrintRing *hostPrintRing;   // pinned host memory, or nullptr
    // replays completed so far (bumped by the last kernel of every replay):
    // with a node's position in the graph, a unique tag per launch
    const uint32_t *replayCounter;
    // batch ray caster (CudaBatchRenderConfig): 0 = off
    uint32_t raycastOutputResolution;
    uint32_t raycastRGBD;
    GrowMailbox *growMailbox;       // pinned host memory, or nullptr
    // MADRONA_TRACING builds of the runtime (mw_gpu/tracing.hpp): the device
    // event log, and which kernel of the graph is running (written by a
    // marker launch in front of every kernel); nullptr otherwise
    void *deviceTracing;
    void *traceCursor;
};

// Load through the constant address space: for data no kernel of the *user*
// code object ever writes (the ecs_state header fields, the column-pointer
// table -- rewritten only by the sort's finalize kernel, between user
// kernels).  The compiler may then hoist / batch these loads across the
// stores of a system instead of re-reading them after every store: a
// per-world system chains dozens of `ctx.get<T>(e) = ...`, and each
// dependent reload is a ~0.5 us round trip with one lane per wave.
template <typename T>
MWHIP_HD inline T loadInvariant(const T *p)
{
    static_assert(sizeof(T) == 4 || sizeof(T) == 8);
#if !defined(__HIP_DEVICE_COMPILE__)
    return *p;
#else
    if constexpr (sizeof(T) == 8) {
        unsigned long long v = *(const __attribute__((address_space(4)))
            unsigned long long *)(unsigned long long)p;
        return __builtin_bit_cast(T, v);
    } else {
        unsigned int v = *(const __attribute__((address_space(4)))
            unsigned int *)(unsigned long long)p;
        return __builtin_bit_cast(T, v);
    }
#endif
}

// Load through the GLOBAL address space (whole dwords of a trivially copyable
// T).  Through a generic pointer the same load is a flat_load: it counts against
// the LDS counter as well, so that the next wait for an LDS access also waits
// for every row still on its way from HBM.  For loads whose whole point is to
// be in flight together (phys_impl/world_step.inl, loadWorldFramed).
template <typename T>
MWHIP_HD inline T loadGlobal(const T *p)
{
#if !defined(__HIP_DEVICE_COMPILE__)
    return *p;
#else
    static_assert(sizeof(T) % 4 == 0);
    struct Words { unsigned int w[sizeof(T) / 4]; };
    const __attribute__((address_space(1))) unsigned int *g =
        (const __attribute__((address_space(1))) unsigned int *)
            (unsigned long long)p;
    Words words;
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; i++) {
        words.w[i] = g[i];
    }
    return __builtin_bit_cast(T, words);
#endif
}

MWHIP_HD inline bool loadGlobalBool(const bool *p)
{
#if !defined(__HIP_DEVICE_COMPILE__)
    return *p;
#else
    return *(const __attribute__((address_space(1))) unsigned char *)
        (unsigned long long)p != 0;
#endif
}

// ... and the store that goes with it (a flat_store of a value that came out of
// LDS is held back by nothing, but the wait in front of the next flat access
// counts it)
template <typename T>
MWHIP_HD inline void storeGlobal(T *p, const T &v)
{
#if !defined(__HIP_DEVICE_COMPILE__)
    *p = v;
#else
    static_assert(sizeof(T) % 4 == 0);
    struct Words { unsigned int w[sizeof(T) / 4]; };
    const Words words = __builtin_bit_cast(Words, v);
    __attribute__((address_space(1))) unsigned int *g =
        (__attribute__((address_space(1))) unsigned int *)(unsigned long long)p;
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; i++) {
        g[i] = words.w[i];
    }
#endif
}

// header fields that never change once the executor is built
MWHIP_HD inline TableHdr *tablesOf(const EcsState *S)
{
    return loadInvariant(&S->tables);
}

MWHIP_HD inline EntitySlot *entitiesOf(const EcsState *S)
{
    return loadInvariant(&S->entities);
}

MWHIP_HD inline IdCache *worldCachesOf(const EcsState *S)
{
    return loadInvariant(&S->worldCaches);
}

// a table's column base (swapped only by the sort's finalize kernel)
MWHIP_HD inline void *columnOf(const TableHdr &tbl, int32_t column_idx)
{
    return loadInvariant(&tbl.columns[column_idx]);
}

// Per-node state of the row-count snapshot (mwhip_pfor_args::row_sync): one
// {launch tag : 32 | rows : 32} granule per matched table, written by
// workgroup 0 of the launch and polled by the others.
struct PforRowSync {
    unsigned long long reserved;
    unsigned long long rows[1];     // [num_matching]
};

#if defined(__HIPCC__)

MWHIP_DEV inline int32_t atomicAddI32(int32_t *p, int32_t v)
{
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_AGENT);
}

MWHIP_DEV inline void raiseError(EcsState *S, uint32_t flag)
{
    __hip_atomic_fetch_or(&S->errorFlags, flag, __ATOMIC_RELAXED,
                          __HIP_MEMORY_SCOPE_AGENT);
}

// ---- per-world id cache ----------------------------------------------------
// The lock protects the cache when several lanes of the same world allocate
// concurrently.  Written in the "try, do, release inside one loop iteration"
// form so divergent lanes of one wavefront cannot dead-lock each other.
template <typename Fn>
MWHIP_DEV inline void withWorldCache(EcsState *S, int32_t world, Fn &&fn)
{
    IdCache *cache = &worldCachesOf(S)[world];
    bool done = false;
    while (!done) {
        if (__hip_atomic_exchange(&cache->lock, 1u, __ATOMIC_ACQUIRE,
                                  __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            fn(*cache);
            __hip_atomic_store(&cache->lock, 0u, __ATOMIC_RELEASE,
                               __HIP_MEMORY_SCOPE_AGENT);
            done = true;
        } else {
            __builtin_amdgcn_s_sleep(1);
        }
    }
}

// Pops the head of a cached free list; mirrors `assignCachedID`
// (reference id_map_impl.inl:73-101) including the contiguous-run encoding in
// freeNode.globalNext.
// head_value: the current *head when the caller has already read it
MWHIP_DEV inline int32_t popCachedId(EcsState *S, int32_t *head, uint32_t *gen_out,
                                    int32_t head_value)
{
    int32_t new_id = head_value;
    EntitySlot &node = entitiesOf(S)[new_id];
    // the whole 12-byte slot in one round trip
    const int32_t sub_next = node.freeNode.subNext;
    const int32_t num_contiguous = node.freeNode.globalNext;
    const uint32_t gen = node.gen;

    if (num_contiguous == 1) {
        *head = sub_next;
    } else {
        int32_t next_free = new_id + 1;
        EntitySlot &next_node = entitiesOf(S)[next_free];
        next_node.freeNode.subNext = sub_next;
        next_node.freeNode.globalNext = num_contiguous - 1;
        next_node.gen = 0;
        *head = next_free;
    }

    *gen_out = gen;
    return new_id;
}

// Takes a fresh block of 64 ids from the end of the id store
// (== store_.expand(ids_per_cache_), reference id_map_impl.inl:158-182).
// During world construction the block order is made deterministic (world
// major, like the CPU backend's sequential constructor loop): pass 1 counts
// blocks per world, pass 2 replays with prefix-summed bases.
// After init a world's k-th new block comes from a static partition,
// runtimeIdBase + (k * numWorlds + world) * 64: independent of thread
// scheduling, so runs are reproducible.  (The CPU backend hands such blocks out
// in world-major order *within a step*, which a node-major GPU schedule cannot
// reproduce without serialising worlds; ids therefore match the CPU backend
// bit for bit as long as worlds get their blocks during construction -- the
// case for Escape-Room / Hide-and-Seek style simulators -- and are otherwise a
// per-world renaming of them.  DESIGN.md §5.)
MWHIP_DEV inline bool waitForMailbox(EcsState *S, uint32_t arch, int32_t row);

MWHIP_DEV inline int32_t expandIdStore(EcsState *S, int32_t world, IdCache &cache)
{
    int32_t block_start;
    if (S->initMode == 2u) {
        block_start = S->initBlockBase[world] +
            cache.initBlocksUsed * kIdsPerBlock;
        cache.initBlocksUsed += 1;
    } else if (S->initMode == 1u) {
        cache.initBlocksUsed += 1;
        block_start = atomicAddI32(&S->numIds, kIdsPerBlock);
    } else {
        block_start = S->runtimeIdBase +
            (cache.runtimeBlocksUsed * S->numWorlds + world) * kIdsPerBlock;
        cache.runtimeBlocksUsed += 1;
    }

    // past what the header says is mapped: the store lives in reserved address
    // space like the tables, the service thread maps more behind it
    if (block_start + kIdsPerBlock > S->entityCapacity &&
            !waitForMailbox(S, kGrowSlotEntities,
                            block_start + kIdsPerBlock - 1)) {
        raiseError(S, kErrEntityOverflow);
        block_start = 0;
    }
    return block_start;
}

MWHIP_DEV inline int32_t acquireIdLocked(EcsState *S, int32_t world, IdCache &cache,
                                        uint32_t *gen_out)
{
    // counters and list heads in one round trip (they share the cache
    // struct's line); the popped id's slot is then the only dependent read
    const int32_t num_overflow = cache.numOverflow;
    const int32_t num_free = cache.numFree;
    const int32_t overflow_head = cache.overflowHead;
    const int32_t free_head = cache.freeHead;

    if (num_overflow > 0) {
        cache.numOverflow = num_overflow - 1;
        return popCachedId(S, &cache.overflowHead, gen_out, overflow_head);
    }

    if (num_free > 0) {
        cache.numFree = num_free - 1;
        return popCachedId(S, &cache.freeHead, gen_out, free_head);
    }

    // refill from the global list of returned blocks (id_map_impl.inl:118-156)
    unsigned long long cur = __hip_atomic_load(&S->idFreeHead, __ATOMIC_ACQUIRE,
                                               __HIP_MEMORY_SCOPE_AGENT);
    int32_t free_ids = kIdSentinel;
    while (true) {
        int32_t head = (int32_t)(uint32_t)(cur & 0xFFFFFFFFull);
        if (head == kIdSentinel) {
            break;
        }
        uint32_t gen = (uint32_t)(cur >> 32);
        int32_t next = __hip_atomic_load(&entitiesOf(S)[head].freeNode.globalNext,
            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long desired =
            ((unsigned long long)(gen + 1u) << 32) | (uint32_t)next;
        if (__hip_atomic_compare_exchange_strong(&S->idFreeHead, &cur, desired,
                __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) {
            free_ids = head;
            break;
        }
    }

    if (free_ids != kIdSentinel) {
        entitiesOf(S)[free_ids].freeNode.globalNext = 1;
        cache.freeHead = free_ids;
        cache.numFree = kIdsPerBlock - 1;
        return popCachedId(S, &cache.freeHead, gen_out, free_ids);
    }

    int32_t first_id = expandIdStore(S, world, cache);
    entitiesOf(S)[first_id].gen = 0;

    int32_t free_start = first_id + 1;
    EntitySlot &next_free = entitiesOf(S)[free_start];
    next_free.freeNode.subNext = kIdSentinel;
    next_free.freeNode.globalNext = kIdsPerBlock - 1;
    next_free.gen = 0;

    cache.freeHead = free_start;
    cache.numFree = kIdsPerBlock - 1;

    *gen_out = 0;
    return first_id;
}

// mirrors IDMap::releaseID (reference id_map_impl.inl:186-224)
// cur_gen: the slot's generation (the caller has just validated the handle
// against it, so it is not read again)
MWHIP_DEV inline void releaseIdLocked(EcsState *S, IdCache &cache, int32_t id,
                                      uint32_t cur_gen)
{
    EntitySlot &node = entitiesOf(S)[id];
    node.gen = cur_gen + 1;
    node.freeNode.globalNext = 1;

    // the cache's four words in one round trip
    const int32_t num_free = cache.numFree;
    const int32_t free_head = cache.freeHead;
    const int32_t num_overflow = cache.numOverflow;
    const int32_t overflow_head = cache.overflowHead;

    if (num_free < kIdsPerBlock) {
        node.freeNode.subNext = free_head;
        cache.freeHead = id;
        cache.numFree = num_free + 1;
        return;
    }

    if (num_overflow < kIdsPerBlock) {
        node.freeNode.subNext = overflow_head;
        cache.overflowHead = id;
        cache.numOverflow = num_overflow + 1;
    }

    if (cache.numOverflow == kIdsPerBlock) {
        int32_t new_head = cache.overflowHead;
        unsigned long long cur = __hip_atomic_load(&S->idFreeHead,
            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (true) {
            __hip_atomic_store(&entitiesOf(S)[new_head].freeNode.globalNext,
                (int32_t)(uint32_t)(cur & 0xFFFFFFFFull), __ATOMIC_RELAXED,
                __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long desired =
                ((unsigned long long)((uint32_t)(cur >> 32) + 1u) << 32) |
                (uint32_t)new_head;
            if (__hip_atomic_compare_exchange_strong(&S->idFreeHead, &cur,
                    desired, __ATOMIC_RELEASE, __ATOMIC_RELAXED,
                    __HIP_MEMORY_SCOPE_AGENT)) {
                break;
            }
        }
        cache.overflowHead = kIdSentinel;
        cache.numOverflow = 0;
    }
}

// 16-byte aligned persistent allocation (world constructors only).
MWHIP_DEV inline void *persistAlloc(EcsState *S, unsigned long long num_bytes)
{
    num_bytes = (num_bytes + 15ull) & ~15ull;
    unsigned long long off = __hip_atomic_fetch_add(&S->persistOffset, num_bytes,
        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (off + num_bytes > S->persistCapacity) {
        raiseError(S, kErrPersistOverflow);
        return S->persistBase;
    }
    return S->persistBase + off;
}

// ---- rows --------------------------------------------------------------------
// Appends one row to the archetype's global table.  Rows of one world that are
// created by one thread keep their creation order under the stable world sort,
// which is what makes per-world row order equal to the CPU backend's.
// Split in two so that a caller can put independent memory work between the
// atomic and the first use of its result (one lane per world: every dependent
// round trip is exposed).
// Any kernel that can reach a row append carries this word of static LDS.
// The host reads the kernel's static LDS size (hipFuncGetAttributes) when it
// builds the launch list: a ParallelFor kernel WITHOUT static LDS provably
// cannot append rows and reads the live row counts directly; every other one
// snapshots them once per launch (pforRowSnapshot, taskgraph.inl).  The test
// errs on the safe side: LDS used for anything else also selects the snapshot.
MWHIP_DEV inline void markRowAppender()
{
    __shared__ uint32_t mwhip_row_appender_marker;
    *(volatile uint32_t *)&mwhip_row_appender_marker = 1u;
}

// One atomic per wavefront and table, not per lane: an atomic on one address
// costs 11 ns serialised on an idle chip and several times that under load
// (profiles/r02_atomic_microbench.txt) -- a system in which a few thousand
// lanes each append one row spent its time queueing on the row counter.  The
// lanes that are in this call together and append to the same table take
// consecutive rows in lane order (deterministic, unlike arrival order).
MWHIP_DEV inline int32_t appendRowIssue(TableHdr &tbl)
{
    markRowAppender();
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const unsigned long long mine = (unsigned long long)&tbl;
    unsigned long long todo = __ballot(1);
    int32_t row = 0;
    while (todo != 0ull) {
        const int leader = __builtin_ctzll(todo);
        const unsigned long long lead_tbl =
            ((unsigned long long)(uint32_t)__shfl((int32_t)(mine >> 32), leader, 64) << 32) |
            (unsigned long long)(uint32_t)__shfl((int32_t)mine, leader, 64);
        const bool same = mine == lead_tbl;
        const unsigned long long group = __ballot(same) & todo;
        if (same) {
            int32_t base = 0;
            if ((int)lane == leader) {
                tbl.needsSort = 1u;
                base = atomicAddI32(&tbl.numRows, (int32_t)__builtin_popcountll(group));
            }
            base = __shfl(base, leader, 64);
            row = base + (int32_t)__builtin_popcountll(group & ((1ull << lane) - 1ull));
        }
        todo &= ~group;
    }
    return row;
}

// Slow path of appendRowCheck: the row lies past what the table's header says
// is mapped.  The header may be behind (it is refreshed between replays): ask
// the mailbox, and if the row really is unmapped, request it and wait for the
// service thread.  Bounded: without an answer in ~0.2 s the append fails like
// on a fixed-capacity table.
// Waits until slot `arch` of the mailbox covers unit `row` (a table row, an
// entity id, a KiB of scratch), asking the executor's service thread for it.
MWHIP_DEV inline bool waitForMailbox(EcsState *S, uint32_t arch, int32_t row)
{
    GrowMailbox *mb = S->growMailbox;
    if (mb == nullptr) {
        return false;
    }
    if (row < __hip_atomic_load(&mb->capacity[arch], __ATOMIC_ACQUIRE,
                                __HIP_MEMORY_SCOPE_SYSTEM)) {
        return true;
    }
    if (__hip_atomic_load(&mb->serviceEnabled, __ATOMIC_RELAXED,
                          __HIP_MEMORY_SCOPE_SYSTEM) == 0u) {
        return false;
    }
    // (the mailbox is host memory: only loads, stores, add, swap and CAS are
    // atomic across the bus -- no fetch_max.  Waiters re-post instead: the
    // largest request wins within a few rounds.)
    for (uint32_t spins = 0; spins < 50000u; spins++) {
        if ((spins & 15u) == 0u &&
                __hip_atomic_load(&mb->requested[arch], __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_SYSTEM) < row + 1) {
            __hip_atomic_store(&mb->requested[arch], row + 1, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __builtin_amdgcn_s_sleep(127);
        if (row < __hip_atomic_load(&mb->capacity[arch], __ATOMIC_ACQUIRE,
                                    __HIP_MEMORY_SCOPE_SYSTEM)) {
            return true;
        }
    }
    return false;
}

MWHIP_DEV inline bool waitForTableMemory(EcsState *S, TableHdr &tbl, int32_t row)
{
    return waitForMailbox(S, (uint32_t)(&tbl - tablesOf(S)), row);
}

MWHIP_DEV inline int32_t appendRowCheck(EcsState *S, TableHdr &tbl, int32_t row)
{
    const int32_t capacity = loadInvariant(&tbl.capacity);
    if (row >= capacity && !waitForTableMemory(S, tbl, row)) {
        raiseError(S, kErrTableOverflow);
        if (GrowMailbox *mb = S->growMailbox; mb != nullptr &&
                __hip_atomic_load(&mb->failedRow, __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
            const int32_t arch = (int32_t)(&tbl - tablesOf(S));
            __hip_atomic_store(&mb->failedArchetype, arch, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&mb->failedCapacity,
                __hip_atomic_load(&mb->capacity[arch], __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_SYSTEM),
                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&mb->failedRow, row, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        }
        // keep writes in bounds; the host aborts after the step
        atomicAddI32(&tbl.numRows, -1);
        row = capacity - 1;
    }
    return row;
}

MWHIP_DEV inline int32_t appendRow(EcsState *S, TableHdr &tbl)
{
    return appendRowCheck(S, tbl, appendRowIssue(tbl));
}

#endif // __HIPCC__

}
}

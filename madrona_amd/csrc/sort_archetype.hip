// SortArchetypeNode / CompactArchetypeNode for gfx950 (MI355X, CDNA4).
//
// Contract (reference src/mw/device/sort_archetype.cpp):
// stable ascending sort of an archetype's global table by a 4-byte key column
// (the WorldID column for the "compaction" sort), every other column permuted
// with the same permutation, rows whose WorldID is -1 (destroyed) dropped,
// entity_store[e.id].loc.row updated, worldOffsets[]/worldCounts[] rebuilt.
//
// This is NOT the reference's CUB-derived pipeline (~20 megakernel nodes, 512
// key tiles, every payload column moved twice).  Three ways through the node,
// all ending in the same fused gather:
//
// (1) radix chain -- any key, any state of the table:
//   sortHistogram    histograms of all P digits at once (few fat workgroups:
//                    one atomic per non-empty bin per workgroup), survivors
//   sortOnesweep x P one LSD pass each: 2048-key tiles (8 keys per lane, 4
//                    wave64s), wave-ballot digit matching + mbcnt ranking,
//                    keys / indices staged through LDS so the global scatter is
//                    digit-contiguous, decoupled look-back through 8-byte
//                    {epoch tag, status | count} granules stored and polled
//                    with relaxed agent-scope atomics (the data is the flag;
//                    tile = workgroup index, no tickets or arrival counters).
//                    Workgroup 0 of the LAST pass publishes the table (column
//                    pointer swap, new row count): nothing reads the table
//                    until the chain ends.
//   sortGather       ONE fused out-of-place gather of every column (grid.y =
//                    column) from the old buffers (now the twins) into the
//                    current ones, in 16/8/4-byte words; entity Loc remap, the
//                    new WorldID column and the world ranges (two binary
//                    searches per world over the sorted keys) ride along; one
//                    workgroup cleans the state the passes are done with
//   sortFinalize     only for batches with exported (pinned) columns: those
//                    were gathered into their twin and are copied back
//
// (2) compaction chain -- world sorts of tables that nothing but world sorts
//     reorders.  After a step such a table is what the last world sort left
//     (rows grouped by world, TableHdr::sortedRows of them), minus the rows
//     destroyed in place since, plus a short appended tail: radix-sorting it
//     again moves 40 bytes per row to find out what is already known.
//   sortCompactPrepare  workgroup 0 sorts the TAIL by world (stable, one
//                    workgroup) and hands its rows to the prefix tiles they
//                    land in; the others count the survivors of every
//                    2048-row tile of the prefix
//   sortCompactScatter  per tile: survivor flags + tail rows landing in the
//                    tile -> one block scan -> permutation and sorted keys.
//                    dest(prefix row i of world w) = survivors before i + tail
//                    rows of worlds < w; dest(tail row j of world w) =
//                    survivors before the end of w's old range + j.  No
//                    inter-workgroup dependency (the tile counts come from the
//                    kernel before).  Workgroup 0 publishes.
//   sortGather / sortFinalize  as above
//
// (3) sortSmall -- tables that hold few rows: the whole node in one launch.
//
// Every kernel takes an array of sort "sites" and picks sites[blockIdx.y /
// column map], so consecutive sort nodes of a task graph run as ONE chain.
//
// Algorithmic HBM bytes per sort: 4N (histogram) + P*16N
// (key+index read & write per pass, first pass reads keys only) + 4N' (index
// read) + 2*B_row*N' (every column read once, written once).  The compaction
// chain moves 8N + 8N' instead of the first two terms.
#include "runtime_internal.hpp"

#include <cstdlib>

namespace madrona {
namespace mwhip {

namespace {

// tables whose capacity is at most this take the single-launch path (sortSmall)
constexpr uint32_t kSmallSortRows = 32768;

constexpr int kSortThreads = 256;
constexpr int kSortWaves = kSortThreads / 64;
constexpr int kSortItems = 8;
constexpr int kSortTile = kSortThreads * kSortItems;   // 2048 keys
constexpr int kRadixBits = 8;
constexpr int kRadixDigits = 1 << kRadixBits;

constexpr unsigned long long kStatusAggregate = 1ull << 30;
constexpr unsigned long long kStatusInclusive = 2ull << 30;
constexpr unsigned long long kCountMask = (1ull << 30) - 1ull;

__device__ inline uint32_t sortKey(const SortSite &site, uint32_t raw)
{
    // World sort: keys are world ids in [0, W) or 0xFFFFFFFF for destroyed
    // rows; only the low 8*P bits are sorted, and the all-ones pattern still
    // sorts after every live world because W <= 2^(8P) - 1.
    (void)site;
    return raw;
}

__device__ inline unsigned long long ballot64(bool pred)
{
    return __ballot(pred);
}

__device__ inline uint32_t laneId()
{
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// Exclusive scan of one value per thread across a 256-thread block.
// scratch: kSortWaves uint32 in LDS.
__device__ inline uint32_t blockExclusiveScan256(uint32_t v, uint32_t *scratch,
                                                 uint32_t *total_out)
{
    const uint32_t lane = laneId();
    const uint32_t wave = threadIdx.x >> 6;

    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t up = __shfl_up(incl, d, 64);
        if ((int)lane >= d) incl += up;
    }

    if (lane == 63) {
        scratch[wave] = incl;
    }
    __syncthreads();

    uint32_t wave_base = 0;
    uint32_t total = 0;
#pragma unroll
    for (int w = 0; w < kSortWaves; w++) {
        uint32_t s = scratch[w];
        if (w < (int)wave) wave_base += s;
        total += s;
    }
    __syncthreads();

    if (total_out != nullptr) *total_out = total;
    return wave_base + incl - v;
}

// ---------------------------------------------------------------------------
// kernel 1: histogram of every pass's digit + survivors + clear offsets
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kSortThreads)
sortHistogram(EcsState *S, const SortSite *sites)
{
    TraceScope trace_scope(S);
    const SortSite &site = sites[blockIdx.y];
    TableHdr &tbl = S->tables[site.archetype];

    SortState *state = site.state;
    const int32_t n = tbl.numRows;
    const uint32_t *keys = (const uint32_t *)tbl.columns[site.keyColumn];

    // nothing below this kernel looks at the table header again to decide
    // whether (or what) to sort
    const bool active = site.worldSort == 0u || tbl.needsSort != 0u;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        state->active = active ? 1u : 0u;
        state->rowsIn = n;
        state->keyColumn = keys;
    }
    if (!active) {
        return;
    }

    const int32_t num_passes = site.numPasses;

    __shared__ uint32_t lds_hist[4][kRadixDigits];
    __shared__ uint32_t lds_valid;

    for (int i = threadIdx.x; i < 4 * kRadixDigits; i += kSortThreads) {
        (&lds_hist[0][0])[i] = 0;
    }
    if (threadIdx.x == 0) lds_valid = 0;
    __syncthreads();

    uint32_t my_valid = 0;
    const int32_t stride = (int32_t)(gridDim.x * kSortThreads);
    for (int32_t i = (int32_t)(blockIdx.x * kSortThreads + threadIdx.x); i < n;
         i += stride) {
        uint32_t key = sortKey(site, keys[i]);
        my_valid += (key != 0xFFFFFFFFu) ? 1u : 0u;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            if (p < num_passes) {
                atomicAdd(&lds_hist[p][(key >> (p * kRadixBits)) & 0xFFu], 1u);
            }
        }
    }

    // wave-reduce the survivor count, one LDS atomic per wave
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        my_valid += __shfl_down(my_valid, d, 64);
    }
    if (laneId() == 0 && my_valid != 0) {
        atomicAdd(&lds_valid, my_valid);
    }
    __syncthreads();

    for (int i = threadIdx.x; i < num_passes * kRadixDigits; i += kSortThreads) {
        uint32_t c = (&lds_hist[0][0])[i];
        if (c != 0) {
            atomicAdd(&state->bins[i], c);
        }
    }
    if (threadIdx.x == 0 && lds_valid != 0) {
        atomicAdd(&state->numValid, lds_valid);
    }

}

// Makes the sorted table current: ping-pong swap of every column that is not
// pinned, new row count.  Executed by ONE workgroup, BEFORE the gather: between
// the last key pass and the end of the chain nothing reads the table, and the
// gather moves the rows from the old buffers (now columnsAlt) into the current
// ones.  Pinned (exported) columns keep their address: they are gathered into
// their twin and copied back by sortFinalize.
__device__ inline void publishSite(EcsState *S, const SortSite &site,
                                   TableHdr &tbl, int32_t n)
{
    SortState *state = site.state;
    const int32_t n_out = site.worldSort ? (int32_t)state->numValid : n;
    for (int32_t c = threadIdx.x; c < tbl.numColumns; c += blockDim.x) {
        if ((tbl.columnFlags[c] & kColumnPinned) == 0u) {
            void *old_buf = tbl.columns[c];
            void *new_buf = tbl.columnsAlt[c];
            tbl.columns[c] = new_buf;
            tbl.columnsAlt[c] = old_buf;
            S->colPtr[site.archetype * S->numComponentSlots +
                      tbl.columnComponent[c]] = new_buf;
        }
    }
    if (threadIdx.x == 0) {
        state->rowsOut = n_out;
        state->statRowsIn += (unsigned long long)n;
        state->statRowsOut += (unsigned long long)n_out;
        state->statRuns += 1ull;

        if (n > tbl.peakRows) {
            tbl.peakRows = n;
        }
        tbl.numRows = n_out;
        // A sort by any other key scrambles the rows across worlds: the next
        // world sort / compaction must not early-out, worldOffsets / worldCounts
        // are stale until then (reference sort_archetype.cpp:1001-1007).
        tbl.needsSort = site.worldSort ? 0u : 1u;
        tbl.sortedRows = site.worldSort ? n_out : 0;
    }
}

// State the key passes are done with, cleaned for the next run (by one
// workgroup of the gather, or at the end of sortSmall).
__device__ inline void cleanSortState(SortState *state)
{
    for (int i = threadIdx.x; i < 4 * kRadixDigits; i += blockDim.x) {
        state->bins[i] = 0;
    }
    if (threadIdx.x == 0) {
        state->numValid = 0;
        state->epoch += 1u;
    }
}

// ---------------------------------------------------------------------------
// kernel 2: one LSD radix pass (one-sweep, decoupled look-back)
// ---------------------------------------------------------------------------
struct alignas(16) OnesweepLDS {
    uint32_t waveHist[kSortWaves][kRadixDigits]; // per-wave digit counts -> offsets
    uint32_t digitStart[kRadixDigits];           // tile-local exclusive digit offsets
    int32_t globalBase[kRadixDigits];            // dst index of local position 0 of digit
    uint32_t stageKeys[kSortTile];
    int32_t stageIdx[kSortTile];
    uint32_t scanScratch[kSortWaves];
};

__global__ void __launch_bounds__(kSortThreads)
sortOnesweep(EcsState *S, const SortSite *sites, uint32_t pass)
{
    TraceScope trace_scope(S);
    const SortSite &site = sites[blockIdx.y];
    TableHdr &tbl = S->tables[site.archetype];

    if ((int32_t)pass >= site.numPasses) {
        return;
    }
    SortState *state = site.state;
    if (state->active == 0u) {
        return;
    }

    const int32_t n = state->rowsIn;

    // The last pass no longer touches the table (it reads the key / index
    // buffers of the pass before it, or the key column through the pointer the
    // histogram kernel saved): its first workgroup publishes the sorted table
    // -- ping-pong swap, new row count -- while the others still scatter.
    if ((int32_t)pass == site.numPasses - 1 && blockIdx.x == 0) {
        publishSite(S, site, tbl, n);
    }

    __shared__ OnesweepLDS lds;

    // Tile = workgroup index + round * grid.  Workgroups are dispatched in index
    // order and the grid never exceeds what the chip keeps resident at once
    // (buildSortLaunches caps it), so the predecessors a tile waits for in the
    // look-back below are running or done: in round 0 they have a lower
    // workgroup index, in later rounds every workgroup is resident.  No ticket
    // counter: one atomic per tile on one address costs more than the rest of
    // a small pass.  (Rounds beyond the first: tables of more tiles than the
    // cap, or a table that grew during the replay -- on-demand growth.)
    for (uint32_t tile = blockIdx.x;
         (int32_t)(tile * (uint32_t)kSortTile) < n; tile += gridDim.x) {
    const int32_t tile_base = (int32_t)(tile * (uint32_t)kSortTile);
    __syncthreads();
    for (int i = threadIdx.x; i < kSortWaves * kRadixDigits; i += kSortThreads) {
        (&lds.waveHist[0][0])[i] = 0;
    }
    __syncthreads();
    const int32_t tile_count = min(kSortTile, n - tile_base);

    const uint32_t *keys_in;
    const int32_t *idx_in;
    if (pass == 0) {
        keys_in = state->keyColumn;
        idx_in = nullptr;
    } else {
        keys_in = (pass & 1u) ? site.keysA : site.keysB;
        idx_in = (pass & 1u) ? site.idxA : site.idxB;
    }
    uint32_t *keys_out = (pass & 1u) ? site.keysB : site.keysA;
    int32_t *idx_out = (pass & 1u) ? site.idxB : site.idxA;

    const uint32_t shift = pass * kRadixBits;
    const uint32_t lane = laneId();
    const uint32_t wave = threadIdx.x >> 6;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;

    // ---- load (wave-striped: wave w owns a contiguous 512-key run, item j of
    // lane l is key j*64+l of the run, which keeps the sort stable) ----------
    uint32_t key[kSortItems];
    int32_t idx[kSortItems];
    uint32_t rank[kSortItems];

    const int32_t wave_base = (int32_t)(wave * 64u * kSortItems);
#pragma unroll
    for (int j = 0; j < kSortItems; j++) {
        int32_t local = wave_base + j * 64 + (int32_t)lane;
        bool valid = local < tile_count;
        int32_t gi = tile_base + local;
        key[j] = valid ? sortKey(site, keys_in[gi]) : 0xFFFFFFFFu;
        idx[j] = valid ? (idx_in != nullptr ? idx_in[gi] : gi) : -1;
    }

    // ---- rank within the wave: ballot-match the 8 digit bits ---------------
#pragma unroll
    for (int j = 0; j < kSortItems; j++) {
        int32_t local = wave_base + j * 64 + (int32_t)lane;
        bool valid = local < tile_count;
        uint32_t digit = (key[j] >> shift) & 0xFFu;

        unsigned long long match = ballot64(valid);
#pragma unroll
        for (int b = 0; b < kRadixBits; b++) {
            bool bit = ((digit >> b) & 1u) != 0u;
            unsigned long long vote = ballot64(bit && valid);
            match &= bit ? vote : ~vote;
        }

        uint32_t before = (uint32_t)__popcll(match & lane_lt);
        uint32_t count = (uint32_t)__popcll(match);

        uint32_t prev = 0;
        if (valid) {
            prev = lds.waveHist[wave][digit];
        }
        __builtin_amdgcn_wave_barrier();
        if (valid && before == 0) {
            lds.waveHist[wave][digit] = prev + count;
        }
        __builtin_amdgcn_wave_barrier();

        rank[j] = prev + before;
    }
    __syncthreads();

    // ---- per-digit: wave offsets, tile totals, tile-local digit starts -----
    const uint32_t d = threadIdx.x;     // 256 threads <-> 256 digits
    uint32_t digit_total = 0;
#pragma unroll
    for (int w = 0; w < kSortWaves; w++) {
        uint32_t c = lds.waveHist[w][d];
        lds.waveHist[w][d] = digit_total;
        digit_total += c;
    }

    // global start of this digit = exclusive scan of the pass's histogram
    uint32_t bin_count = state->bins[pass * kRadixDigits + d];
    uint32_t bin_start = blockExclusiveScan256(bin_count, lds.scanScratch, nullptr);
    uint32_t digit_start =
        blockExclusiveScan256(digit_total, lds.scanScratch, nullptr);
    lds.digitStart[d] = digit_start;

    // ---- decoupled look-back over predecessor tiles for digit d ------------
    const uint32_t tag = state->epoch * 8u + pass + 1u;
    unsigned long long *granules = site.lookback;
    const size_t my_slot = (size_t)tile * kRadixDigits + d;

    uint32_t exclusive = 0;
    if (tile == 0) {
        __hip_atomic_store(&granules[my_slot],
            ((unsigned long long)tag << 32) | kStatusInclusive | digit_total,
            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        __hip_atomic_store(&granules[my_slot],
            ((unsigned long long)tag << 32) | kStatusAggregate | digit_total,
            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

        int32_t look = (int32_t)tile - 1;
        uint32_t spins = 0;
        while (true) {
            unsigned long long g = __hip_atomic_load(
                &granules[(size_t)look * kRadixDigits + d],
                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((uint32_t)(g >> 32) != tag) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 26)) {
                    raiseError(S, kErrSortLookback);
                    break;
                }
                continue;
            }
            exclusive += (uint32_t)(g & kCountMask);
            if ((g & kStatusInclusive) != 0ull) {
                break;
            }
            look -= 1;
        }

        __hip_atomic_store(&granules[my_slot],
            ((unsigned long long)tag << 32) | kStatusInclusive |
                (unsigned long long)(exclusive + digit_total),
            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    lds.globalBase[d] =
        (int32_t)(bin_start + exclusive) - (int32_t)digit_start;
    __syncthreads();

    // ---- stage keys + indices in LDS at their tile-local sorted position ---
#pragma unroll
    for (int j = 0; j < kSortItems; j++) {
        int32_t local = wave_base + j * 64 + (int32_t)lane;
        if (local < tile_count) {
            uint32_t digit = (key[j] >> shift) & 0xFFu;
            uint32_t pos = lds.digitStart[digit] + lds.waveHist[wave][digit] +
                rank[j];
            lds.stageKeys[pos] = key[j];
            lds.stageIdx[pos] = idx[j];
        }
    }
    __syncthreads();

    // ---- digit-contiguous global scatter -----------------------------------
#pragma unroll
    for (int j = 0; j < kSortItems; j++) {
        int32_t pos = j * kSortThreads + (int32_t)threadIdx.x;
        if (pos < tile_count) {
            uint32_t k = lds.stageKeys[pos];
            uint32_t digit = (k >> shift) & 0xFFu;
            int32_t dst = lds.globalBase[digit] + pos;
            keys_out[dst] = k;
            idx_out[dst] = lds.stageIdx[pos];
        }
    }
    }   // tile rounds
}

// ---------------------------------------------------------------------------
// kernel 3: fused gather of every column + entity remap + world boundaries
// ---------------------------------------------------------------------------
template <typename WordT>
__device__ inline void gatherWords(const WordT *__restrict__ src,
                                   WordT *__restrict__ dst,
                                   const int32_t *__restrict__ perm,
                                   int32_t num_rows, uint32_t words_per_row,
                                   unsigned long long inv_magic,
                                   int32_t first, int32_t step)
{
    const long long total = (long long)num_rows * words_per_row;
    const long long stride = step;
    long long j = first;

    // Four independent (index -> word) chains in flight per thread: at
    // Escape-Room sizes a thread only has a handful of words, and without the
    // explicit batching each one costs two dependent round trips in sequence.
    constexpr int kBatch = 4;

    if (words_per_row == 1) {
        for (; j + (kBatch - 1) * stride < total; j += kBatch * stride) {
            int32_t p[kBatch];
            WordT w[kBatch];
#pragma unroll
            for (int u = 0; u < kBatch; u++) p[u] = perm[j + u * stride];
#pragma unroll
            for (int u = 0; u < kBatch; u++) w[u] = src[p[u]];
#pragma unroll
            for (int u = 0; u < kBatch; u++) dst[j + u * stride] = w[u];
        }
        for (; j < total; j += stride) {
            dst[j] = src[perm[j]];
        }
    } else {
        // row = j / words_per_row via a 64-bit reciprocal (exact for j < 2^32)
        for (; j + (kBatch - 1) * stride < total; j += kBatch * stride) {
            uint32_t off[kBatch];
            int32_t p[kBatch];
            WordT w[kBatch];
#pragma unroll
            for (int u = 0; u < kBatch; u++) {
                long long ju = j + u * stride;
                uint32_t row =
                    (uint32_t)__umul64hi((unsigned long long)ju, inv_magic);
                off[u] = (uint32_t)(ju - (long long)row * words_per_row);
                p[u] = perm[row];
            }
#pragma unroll
            for (int u = 0; u < kBatch; u++) {
                w[u] = src[(long long)p[u] * words_per_row + off[u]];
            }
#pragma unroll
            for (int u = 0; u < kBatch; u++) dst[j + u * stride] = w[u];
        }
        for (; j < total; j += stride) {
            uint32_t row = (uint32_t)__umul64hi((unsigned long long)j, inv_magic);
            uint32_t off = (uint32_t)(j - (long long)row * words_per_row);
            dst[j] = src[(long long)perm[row] * words_per_row + off];
        }
    }
}

// Columns whose rows are whole dwords (all but 1- / 2-byte components), moved in
// 16-byte chunks of the DESTINATION: dst is a dense array of rows, so chunk k
// = dwords [4k, 4k + 4) of it is one aligned 16-byte store.  Its source dwords
// belong to at most two rows when a row has >= 2 dwords, to four rows of a
// 4-byte column; they are contiguous in the source too whenever those rows are
// consecutive there (perm[r + 1] == perm[r] + 1) -- almost always: a world
// sort moves rows in long runs -- and then they are one (dword-aligned) 16-byte
// load.  12- and 28-byte components (positions, solver state: most of a rigid
// body's bytes) would otherwise move 4 bytes per lane per instruction.
struct __attribute__((packed, aligned(4))) Dwords4 { uint32_t x, y, z, w; };

__device__ inline void gatherRowsWide(const uint32_t *__restrict__ src,
                                      uint32_t *__restrict__ dst,
                                      const int32_t *__restrict__ perm,
                                      int32_t num_rows, uint32_t row_dwords,
                                      unsigned long long inv_magic,
                                      int32_t first, int32_t step)
{
    const long long total = (long long)num_rows * row_dwords;
    const long long chunks = total >> 2;
    constexpr int kBatch = 4;

    auto move = [&](long long k0, int count) {
        // (count <= kBatch chunks k0, k0 + step, ...: independent chains)
        long long src_at[kBatch];   // source dword of the chunk's first dword
        int32_t row_b[kBatch];      // source row of the dwords behind `split`
        uint32_t split[kBatch];     // dwords of the chunk that come from the first row
        bool contiguous[kBatch];
        uint4 rows4[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; u++) {
            if (u >= count) continue;
            const long long d0 = (k0 + (long long)u * step) << 2;
            if (row_dwords == 1u) {
                const uint4 p = *(const uint4 *)(perm + d0);
                rows4[u] = p;
                src_at[u] = (long long)(int32_t)p.x;
                contiguous[u] = p.y == p.x + 1u && p.z == p.x + 2u && p.w == p.x + 3u;
                split[u] = 1u;
                row_b[u] = 0;
            } else {
                const uint32_t r0 =
                    (uint32_t)__umul64hi((unsigned long long)d0, inv_magic);
                const uint32_t o0 = (uint32_t)(d0 - (long long)r0 * row_dwords);
                const int32_t p0 = perm[r0];
                // (the chunk ends inside row r0, or in row r0 + 1 which exists:
                // d0 + 3 < total)
                const bool crosses = o0 + 4u > row_dwords;
                const int32_t p1 = crosses ? perm[r0 + 1u] : p0 + 1;
                src_at[u] = (long long)p0 * row_dwords + o0;
                split[u] = crosses ? row_dwords - o0 : 4u;
                row_b[u] = p1;
                contiguous[u] = p1 == p0 + 1;
            }
        }
        uint4 v[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; u++) {
            if (u >= count) continue;
            if (contiguous[u]) {
                const Dwords4 w = *(const Dwords4 *)(src + src_at[u]);
                v[u] = make_uint4(w.x, w.y, w.z, w.w);
            } else if (row_dwords == 1u) {
                v[u] = make_uint4(src[(int32_t)rows4[u].x], src[(int32_t)rows4[u].y],
                                  src[(int32_t)rows4[u].z], src[(int32_t)rows4[u].w]);
            } else {
                const long long b = (long long)row_b[u] * row_dwords -
                    (long long)split[u];
                uint32_t w[4];
#pragma unroll
                for (uint32_t e = 0; e < 4u; e++) {
                    w[e] = e < split[u] ? src[src_at[u] + e] : src[b + e];
                }
                v[u] = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
#pragma unroll
        for (int u = 0; u < kBatch; u++) {
            if (u >= count) continue;
            *(uint4 *)(dst + ((k0 + (long long)u * step) << 2)) = v[u];
        }
    };

    long long k = first;
    for (; k + (long long)(kBatch - 1) * step < chunks; k += (long long)kBatch * step) {
        move(k, kBatch);
    }
    for (; k < chunks; k += step) {
        move(k, 1);
    }
    // the last, partial chunk
    if (first == 0) {
        for (long long d = chunks << 2; d < total; d++) {
            const uint32_t row = row_dwords == 1u ? (uint32_t)d :
                (uint32_t)__umul64hi((unsigned long long)d, inv_magic);
            const uint32_t off = (uint32_t)(d - (long long)row * row_dwords);
            dst[d] = src[(long long)perm[row] * row_dwords + off];
        }
    }
}

struct alignas(16) Word16 { uint32_t v[4]; };
struct alignas(8) Word8 { uint32_t v[2]; };

// first index in sorted[0, n) whose key is >= key
__device__ inline int32_t lowerBound(const uint32_t *__restrict__ sorted,
                                     int32_t n, uint32_t key)
{
    int32_t lo = 0, len = n;
    while (len > 0) {
        int32_t half = len >> 1;
        bool right = sorted[lo + half] < key;
        lo = right ? lo + half + 1 : lo;
        len = right ? len - half - 1 : half;
    }
    return lo;
}

// worldOffsets[w] / worldCounts[w] straight from the sorted keys: two binary
// searches per world (independent chains, ~log2 n cached loads each).  No
// sentinel pass before and no fix-up pass after the gather, and worlds without
// rows come out right by construction (count 0 at the position they would
// occupy).  tid / stride enumerate the worlds.
__device__ inline void writeWorldRanges(TableHdr &tbl,
                                        const uint32_t *__restrict__ sorted,
                                        int32_t n_out, int32_t num_worlds,
                                        int32_t tid, int32_t stride)
{
    for (int32_t w = tid; w < num_worlds; w += stride) {
        int32_t first = lowerBound(sorted, n_out, (uint32_t)w);
        int32_t end = lowerBound(sorted, n_out, (uint32_t)w + 1u);
        tbl.worldOffsets[w] = first;
        tbl.worldCounts[w] = end - first;
    }
}

__device__ inline int32_t compactNumTiles(int32_t prefix);

// One column (blockIdx.y) of one site: rows move from the buffers the table
// had when the chain started into its current ones.
template <bool Wide = true>
__device__ inline void gatherColumn(EcsState *S, const SortSite &site,
                                    const GatherColumn &gc, TableHdr &tbl,
                                    int32_t row_begin, int32_t row_end,
                                    int32_t tid, int32_t stride,
                                    const int32_t *perm_rows = nullptr,
                                    const uint32_t *sorted_keys = nullptr,
                                    bool published = true);

// Wide: some column of the batch moves in 16-byte destination chunks
// (MADRONA_MWHIP_GATHER_WIDE=1).  The word-by-word kernel is a stream of
// (index -> word -> store) chains: what it wants is wavefronts in flight, and
// without the wide routine it fits EIGHT per SIMD (64 registers) -- measured
// with the register cap alone, spilling the unused wide path
// (profiles/r04_sort_variants.jsonl): 15.1 -> 12.3 us at 4096 Escape-Room worlds,
// 28.4 -> 25.9 at 8192 with physics, 177 -> 159 at 65536.
#ifndef MADRONA_SORT_GATHER_WAVES
#define MADRONA_SORT_GATHER_WAVES 8
#endif
template <bool Wide>
__global__ void __launch_bounds__(kSortThreads)
__attribute__((amdgpu_waves_per_eu(Wide ? 0 : MADRONA_SORT_GATHER_WAVES)))
sortGather(EcsState *S, const SortSite *sites, const GatherColumn *columns,
           const GatherSlice *slices, const MiscOp *trailing_ops,
           uint32_t num_trailing_ops)
{
    TraceScope trace_scope(S);
    // (ResetTmpAlloc nodes that followed the chain in the task graph)
    if (blockIdx.x == 0) {
        applyMiscOps(S, trailing_ops, num_trailing_ops, threadIdx.x);
    }
    const GatherSlice slice = slices[blockIdx.x];
    const GatherColumn gc = columns[slice.column];
    const SortSite &site = sites[gc.site];
    TableHdr &tbl = S->tables[site.archetype];

    SortState *state = site.state;
    if (state->active == 0u) {
        return;
    }
    const int32_t n_out = state->rowsOut;

    // (Row tiles -- a workgroup per 1024 rows of a site moving them in EVERY
    // column with the permutation staged once in LDS -- were built and measured
    // in round 4: slower than contiguous rows per column slice, 291 against
    // 252 us at 65536 Escape-Room worlds on one box, and its three inlined
    // copies of the column routine cost the other modes 30 % as well.  Removed.)
    if (slice.blocked != 0u && gc.column != kWorldRangesColumn) {
        // a contiguous run of rows per workgroup (big tables: every workgroup
        // stays inside a few pages of every buffer it touches instead of
        // striding over all of them)
        int32_t per = (n_out + (int32_t)slice.numSlices - 1) / (int32_t)slice.numSlices;
        per = (per + 3) & ~3;       // (16-byte destination chunks stay aligned)
        const int32_t row_begin = (int32_t)slice.slice * per;
        const int32_t row_end = row_begin + per < n_out ? row_begin + per : n_out;
        if (row_begin < row_end) {
            gatherColumn<Wide>(S, site, gc, tbl, row_begin, row_end,
                               (int32_t)threadIdx.x, (int32_t)kSortThreads);
        }
    } else {
        gatherColumn<Wide>(S, site, gc, tbl, 0, n_out,
                           (int32_t)(slice.slice * kSortThreads + threadIdx.x),
                           (int32_t)(slice.numSlices * kSortThreads));
    }

    // (the passes are over: their histograms and counters are dead)
    if (gc.column == 0u && slice.slice == 0u) {
        cleanSortState(state);
        // the per-tile landing counters of the compaction chain start every
        // run at zero (whichever way the tail went this time)
        const int32_t tiles = compactNumTiles(state->prefixRows) + 1;
        for (int32_t t = (int32_t)threadIdx.x; t < tiles; t += (int32_t)blockDim.x) {
            site.tileTailStart[t] = 0;
        }
    }
}

// Rows [row_begin, row_end) of the sorted table.  tid / stride: this thread's
// index among, and the number of, the threads working on them.
// perm_rows: the permutation entries of rows [row_begin, row_end) if the caller
// has them closer than the index buffer.  Wide = false leaves the 16-byte-chunk
// routine out (and its registers: the word-by-word kernel then fits eight
// wavefronts per SIMD).
// sorted_keys: ALL sorted keys if the caller has them closer than the key
// buffer.  published = false: publishSite has not run yet, rows move from the
// current buffers into the twins (no caller left since the multi-workgroup
// small sort was removed in round 5).
template <bool Wide>
__device__ inline void gatherColumn(EcsState *S, const SortSite &site,
                                    const GatherColumn &gc, TableHdr &tbl,
                                    int32_t row_begin, int32_t row_end,
                                    int32_t tid, int32_t stride,
                                    const int32_t *perm_rows,
                                    const uint32_t *sorted_keys,
                                    bool published)
{
    const bool final_in_b = ((site.numPasses - 1) & 1) != 0;
    const int32_t *perm = perm_rows != nullptr ? perm_rows :
        (final_in_b ? site.idxB : site.idxA) + row_begin;
    const int32_t n_rows = row_end - row_begin;

    const uint32_t *sorted_all = sorted_keys != nullptr ? sorted_keys :
        (final_in_b ? site.keysB : site.keysA);
    const uint32_t col = gc.column;
    if (col == kWorldRangesColumn) {
        writeWorldRanges(tbl, sorted_all, row_end, S->numWorlds, tid, stride);
        return;
    }

    // swapped already (publishSite): the rows still sit in what is now the
    // twin; pinned columns were not swapped and go the other way
    const bool pinned = !published || (tbl.columnFlags[col] & kColumnPinned) != 0u;
    const void *src = pinned ? tbl.columns[col] : tbl.columnsAlt[col];
    void *dst = pinned ? tbl.columnsAlt[col] : tbl.columns[col];

    if (col == 0) {
        // Entity column: 8-byte handles; update the entity store's row
        const Entity *esrc = (const Entity *)src;
        Entity *edst = (Entity *)dst;
        for (int32_t i = row_begin + tid; i < row_end; i += stride) {
            Entity e = esrc[perm[i - row_begin]];
            edst[i] = e;
            if (e.id >= 0) {
                S->entities[e.id].loc.row = i;
            }
        }
        return;
    }

    if (col == 1 && site.worldSort) {
        // the sorted keys ARE the new WorldID column: contiguous copy
        const uint32_t *sorted = sorted_all;
        int32_t *wdst = (int32_t *)dst;
        for (int32_t i = row_begin + tid; i < row_end; i += stride) {
            wdst[i] = (int32_t)sorted[i];
        }
        return;
    }

    const size_t row_bytes = tbl.columnBytes[col];
    char *dst_rows = (char *)dst + (size_t)row_begin * row_bytes;
    if constexpr (Wide) {
        if (gc.rowDwords != 0u) {
            gatherRowsWide((const uint32_t *)src, (uint32_t *)dst_rows, perm, n_rows,
                           gc.rowDwords, gc.invMagicDwords, tid, stride);
            return;
        }
    }

    switch (gc.wordBytes) {
    case 16:
        gatherWords<Word16>((const Word16 *)src, (Word16 *)dst_rows, perm, n_rows,
                            gc.wordsPerRow, gc.invMagic, tid, stride);
        break;
    case 8:
        gatherWords<Word8>((const Word8 *)src, (Word8 *)dst_rows, perm, n_rows,
                           gc.wordsPerRow, gc.invMagic, tid, stride);
        break;
    case 4:
        gatherWords<uint32_t>((const uint32_t *)src, (uint32_t *)dst_rows, perm,
                              n_rows, gc.wordsPerRow, gc.invMagic, tid, stride);
        break;
    default:
        gatherWords<uint8_t>((const uint8_t *)src, (uint8_t *)dst_rows, perm,
                             n_rows, gc.wordsPerRow, gc.invMagic, tid, stride);
        break;
    }
}

// ---------------------------------------------------------------------------
// kernel 4 (only for batches with exported columns): copy-back
// ---------------------------------------------------------------------------
// An exported column must keep its address (PyTorch holds it): it was gathered
// into its twin; the rows come back here.
__global__ void __launch_bounds__(kSortThreads)
sortFinalize(EcsState *S, const SortSite *sites, const MiscOp *trailing_ops,
             uint32_t num_trailing_ops)
{
    TraceScope trace_scope(S);
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        applyMiscOps(S, trailing_ops, num_trailing_ops, threadIdx.x);
    }
    const SortSite &site = sites[blockIdx.y];
    TableHdr &tbl = S->tables[site.archetype];
    SortState *state = site.state;
    if (state->active == 0u) {
        return;
    }

    const int32_t n_out = state->rowsOut;
    const int32_t stride = (int32_t)(gridDim.x * kSortThreads);
    const int32_t tid = (int32_t)(blockIdx.x * kSortThreads + threadIdx.x);

    for (int32_t c = 0; c < tbl.numColumns; c++) {
        if ((tbl.columnFlags[c] & kColumnPinned) == 0u) continue;
        const uint32_t *src = (const uint32_t *)tbl.columnsAlt[c];
        uint32_t *dst = (uint32_t *)tbl.columns[c];
        // column allocations are padded to 16 B, copy whole dwords
        long long words = ((long long)n_out * tbl.columnBytes[c] + 3) / 4;
        for (long long j = tid; j < words; j += stride) {
            dst[j] = src[j];
        }
    }
}

// ---------------------------------------------------------------------------
// small tables: the whole node in ONE launch
// ---------------------------------------------------------------------------
// A chain of histogram / passes / gather (/ finalize) costs its ~4 us launch
// floor per kernel even when the table is empty or unchanged.  Tables that can
// never hold more than kSmallSortRows rows (a physics joint table, ...) are
// sorted by one 1024-thread workgroup per site instead: per pass an LDS
// histogram, then 1024-key tiles ranked with the same wave-ballot matching as
// the one-sweep kernel and scattered in tile order (stable); then the column
// gather, the world ranges and the publication, separated by workgroup
// barriers only.
constexpr int kSmallThreads = 1024;
constexpr int kSmallWaves = kSmallThreads / 64;

struct SmallSortLDS {
    uint32_t hist[kRadixDigits];
    uint32_t binBase[kRadixDigits];
    uint32_t waveBase[kSmallWaves][kRadixDigits];
    uint32_t scan[4];
    uint32_t valid;
};

// ping-pong buffers of the one-workgroup passes (global memory or LDS)
struct RadixBuffers {
    uint32_t *keys[2];
    int32_t *rows[2];
};

// Stable LSD radix passes over n keys by ONE kSmallThreads-wide workgroup.
// Pass 0 reads keys0[i] (source row first_row + i); pass p writes keys / rows to
// buffer (first_out + p) & 1 and reads what pass p - 1 wrote.
// Leaves the number of keys != 0xFFFFFFFF in lds.valid (they sort last).
__device__ inline void oneGroupRadixPasses(SmallSortLDS &lds, int32_t num_passes,
                                           const RadixBuffers &buf,
                                           const uint32_t *keys0, int32_t first_row,
                                           int32_t n, int32_t first_out)
{
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = laneId();
    const uint32_t wave = tid >> 6;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;

    if (tid == 0) lds.valid = 0;

    for (int32_t pass = 0; pass < num_passes; pass++) {
        const int out = (first_out + pass) & 1;
        const uint32_t *keys_in = pass == 0 ? keys0 : buf.keys[out ^ 1];
        const int32_t *idx_in = pass == 0 ? nullptr : buf.rows[out ^ 1];
        uint32_t *keys_out = buf.keys[out];
        int32_t *idx_out = buf.rows[out];
        const uint32_t shift = (uint32_t)pass * kRadixBits;

        if (tid < (uint32_t)kRadixDigits) lds.hist[tid] = 0;
        __syncthreads();

        uint32_t my_valid = 0;
        for (int32_t i = (int32_t)tid; i < n; i += kSmallThreads) {
            uint32_t key = keys_in[i];
            my_valid += key != 0xFFFFFFFFu ? 1u : 0u;
            atomicAdd(&lds.hist[(key >> shift) & 0xFFu], 1u);
        }
        if (pass == 0) {
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
                my_valid += __shfl_down(my_valid, d, 64);
            }
            if (lane == 0 && my_valid != 0) atomicAdd(&lds.valid, my_valid);
        }
        __syncthreads();

        // exclusive scan of the 256 bins by the first four waves
        {
            uint32_t v = tid < (uint32_t)kRadixDigits ? lds.hist[tid] : 0u;
            uint32_t incl = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                uint32_t up = __shfl_up(incl, d, 64);
                if ((int)lane >= d) incl += up;
            }
            if (wave < 4 && lane == 63) lds.scan[wave] = incl;
            __syncthreads();
            if (tid < (uint32_t)kRadixDigits) {
                uint32_t base = 0;
                for (uint32_t w = 0; w < wave; w++) base += lds.scan[w];
                lds.binBase[tid] = base + incl - v;
            }
            __syncthreads();
        }

        // One key per thread and tile.  (Eight keys per thread, wave-striped as in
        // the one-sweep kernel -- a quarter of the barriers -- was measured
        // slower: 13.3 -> 16.5 us for the 1100-row tails of 8192 Escape-Room
        // worlds, three busy waves with eight dependent ranking rounds each.)
        for (int32_t tile = 0; tile < n; tile += kSmallThreads) {
            for (uint32_t i = tid; i < (uint32_t)(kSmallWaves * kRadixDigits);
                 i += kSmallThreads) {
                (&lds.waveBase[0][0])[i] = 0;
            }
            __syncthreads();

            const int32_t i = tile + (int32_t)tid;
            const bool valid = i < n;
            const uint32_t key = valid ? keys_in[i] : 0xFFFFFFFFu;
            const int32_t src = valid ?
                (idx_in != nullptr ? idx_in[i] : first_row + i) : -1;
            const uint32_t digit = (key >> shift) & 0xFFu;

            unsigned long long match = ballot64(valid);
#pragma unroll
            for (int b = 0; b < kRadixBits; b++) {
                bool bit = ((digit >> b) & 1u) != 0u;
                unsigned long long vote = ballot64(bit && valid);
                match &= bit ? vote : ~vote;
            }
            const uint32_t before = (uint32_t)__popcll(match & lane_lt);
            if (valid && before == 0) {
                lds.waveBase[wave][digit] = (uint32_t)__popcll(match);
            }
            __syncthreads();

            // digit d: running base over the waves of this tile, then over tiles
            if (tid < (uint32_t)kRadixDigits) {
                uint32_t run = lds.binBase[tid];
#pragma unroll
                for (int w = 0; w < kSmallWaves; w++) {
                    uint32_t c = lds.waveBase[w][tid];
                    lds.waveBase[w][tid] = run;
                    run += c;
                }
                lds.binBase[tid] = run;
            }
            __syncthreads();

            if (valid) {
                const uint32_t dst = lds.waveBase[wave][digit] + before;
                keys_out[dst] = key;
                idx_out[dst] = src;
            }
            __syncthreads();
        }
    }
}

// worldOffsets / worldCounts of a small table by ONE kSmallThreads-wide
// workgroup: rows per world counted in LDS, exclusive scan, eight consecutive
// worlds per thread.
constexpr int32_t kSmallRangeWorlds = 8192;

template <int THREADS>
__device__ inline unsigned long long blockExclusiveScanU64(
    unsigned long long v, unsigned long long *scratch);

__device__ inline void smallWorldRanges(TableHdr &tbl, const uint32_t *sorted,
                                        int32_t n_out, int32_t num_worlds,
                                        uint32_t *world_rows,
                                        unsigned long long *scan_scratch)
{
    const int32_t tid = (int32_t)threadIdx.x;
    for (int32_t w = tid; w < kSmallRangeWorlds; w += kSmallThreads) {
        world_rows[w] = 0;
    }
    __syncthreads();
    for (int32_t i = tid; i < n_out; i += kSmallThreads) {
        const uint32_t w = sorted[i];
        if (w < (uint32_t)num_worlds) {
            atomicAdd(&world_rows[w], 1u);
        }
    }
    __syncthreads();
    constexpr int32_t per_thread = kSmallRangeWorlds / kSmallThreads;
    uint32_t mine[per_thread];
    uint32_t sum = 0;
#pragma unroll
    for (int32_t j = 0; j < per_thread; j++) {
        mine[j] = world_rows[tid * per_thread + j];
        sum += mine[j];
    }
    uint32_t run = (uint32_t)blockExclusiveScanU64<kSmallThreads>(sum, scan_scratch);
#pragma unroll
    for (int32_t j = 0; j < per_thread; j++) {
        const int32_t w = tid * per_thread + j;
        if (w < num_worlds) {
            tbl.worldOffsets[w] = (int32_t)run;
            tbl.worldCounts[w] = (int32_t)mine[j];
        }
        run += mine[j];
    }
    __syncthreads();
}

// The whole sort of one small table by ONE workgroup (global ping-pong buffers).
__device__ inline void sortSmallOneGroup(EcsState *S, const SortSite &site,
                                         uint32_t site_index, TableHdr &tbl,
                                         const GatherColumn *columns,
                                         uint32_t num_columns, SmallSortLDS &lds,
                                         uint32_t *world_rows,
                                         unsigned long long *range_scan)
{
    const int32_t n = tbl.numRows;
    const uint32_t tid = threadIdx.x;
    const uint32_t *final_keys =
        ((site.numPasses - 1) & 1) != 0 ? site.keysB : site.keysA;

    // (pass p writes buffer p & 1: the last one lands where the gather looks)
    const RadixBuffers buffers { { site.keysA, site.keysB },
                                 { site.idxA, site.idxB } };
    oneGroupRadixPasses(lds, site.numPasses, buffers,
                        (const uint32_t *)tbl.columns[site.keyColumn], 0, n, 0);
    __syncthreads();

    // same order as the chain: publish (swap), then gather old -> current
    SortState *state = site.state;
    if (tid == 0) {
        state->numValid = lds.valid;
    }
    __syncthreads();
    publishSite(S, site, tbl, n);
    __syncthreads();
    const int32_t n_out = site.worldSort ? (int32_t)lds.valid : n;

    // gather: this site's columns one after the other
    for (uint32_t c = 0; c < num_columns; c++) {
        const GatherColumn gc = columns[c];
        if (gc.site != site_index) continue;
        if (gc.column == kWorldRangesColumn && S->numWorlds <= kSmallRangeWorlds) {
            // Two binary searches per world are ~24 dependent loads each, and
            // ONE workgroup doing them for 8192 worlds was 55 of this kernel's
            // 70 us (device trace, joint table of 8192 Escape-Room worlds):
            // count the rows of every world in LDS and scan instead.
            smallWorldRanges(tbl, final_keys, n_out, S->numWorlds, world_rows, range_scan);
            continue;
        }
        gatherColumn(S, site, gc, tbl, 0, n_out, (int32_t)tid, kSmallThreads);
    }
    __syncthreads();

    // pinned (exported) columns keep their address: copy the gathered rows back
    for (int32_t c = 0; c < tbl.numColumns; c++) {
        if ((tbl.columnFlags[c] & kColumnPinned) == 0u) continue;
        const uint32_t *src = (const uint32_t *)tbl.columnsAlt[c];
        uint32_t *dst = (uint32_t *)tbl.columns[c];
        long long words = ((long long)n_out * tbl.columnBytes[c] + 3) / 4;
        for (long long j = tid; j < words; j += kSmallThreads) {
            dst[j] = src[j];
        }
    }
    cleanSortState(state);
}

__global__ void __launch_bounds__(kSmallThreads)
sortSmall(EcsState *S, const SortSite *sites, const GatherColumn *columns,
          uint32_t num_columns, const MiscOp *trailing_ops,
          uint32_t num_trailing_ops)
{
    TraceScope trace_scope(S);
    if (blockIdx.x == 0) {
        applyMiscOps(S, trailing_ops, num_trailing_ops, threadIdx.x);
    }
    const SortSite &site = sites[blockIdx.x];
    TableHdr &tbl = S->tables[site.archetype];
    if (site.worldSort && tbl.needsSort == 0u) {
        return;
    }

    __shared__ SmallSortLDS lds;
    __shared__ uint32_t world_rows[kSmallRangeWorlds];
    __shared__ unsigned long long range_scan[kSmallWaves];
    sortSmallOneGroup(S, site, blockIdx.x, tbl, columns, num_columns, lds,
                      world_rows, range_scan);
}


// ---------------------------------------------------------------------------
// compaction chain (world sorts of tables that are still sorted from last time)
// ---------------------------------------------------------------------------
// State of such a table when its next world sort runs:
//   rows [0, P)   P = TableHdr::sortedRows: what the last world sort left, in
//                 world order, some of them destroyed in place since (key
//                 0xFFFFFFFF); worldOffsets / worldCounts still describe them
//   rows [P, N)   appended since, any world order, some destroyed again
// The stable sort of that is: per world, its surviving prefix rows in order,
// then its tail rows in tail order.  With s(i) = survivors among prefix rows
// [0, i), end(w) = worldOffsets[w] + worldCounts[w], and the live tail rows
// sorted (stably) by world, j = index in that order:
//   prefix row i, world w:  dest = s(i) + #{ tail rows of worlds < w }
//   tail row j, world w:    dest = s(end(w)) + j
// A tail row "lands" at end(w); tile t of the prefix (2048 rows) owns the tail
// rows landing in [2048 t, 2048 (t + 1)), the last tile also those landing at P.
// Landing points are monotone in w, so every tile owns a contiguous range of
// the sorted tail, and #{ tail rows of worlds < w } for a prefix row i of a tile
// = tail rows owned by earlier tiles + owned ones landing at or before i.
constexpr int kCompactTile = kSortTile;            // prefix rows per scatter tile
constexpr int kCompactTileShift = 11;
static_assert((1 << kCompactTileShift) == kCompactTile);
constexpr int kPrepTilesPerGroup = 4;              // count: 16 waves x 512 rows
constexpr int kLdsTailRows = 2048;                 // tails sorted in LDS

__device__ inline int32_t compactNumTiles(int32_t prefix)
{
    const int32_t t = (prefix + kCompactTile - 1) >> kCompactTileShift;
    return t > 0 ? t : 1;
}

// The tail by landing points (round 4).  Sorting the tail by world on ONE
// workgroup (compactSortTail) is the chain's serial step: 10-13 us for the
// ~1 K-row tails of 8192 worlds, 88 us for the ~10-30 K rows of 65536.  It is
// not needed: a tail row of world w lands at end(w), a position of the sorted
// prefix; the scatter tile that owns that position only has to know (1) how
// many live tail rows land in earlier tiles and (2) its own rows in (world,
// tail index) order -- both of which it gets from one scan of the landing
// points.  prepare leaves, per tail row, its landing point (tailLand) and,
// per tile, how many land in it (tileTailStart doubles as that counter; one
// atomic per tail row on one of hundreds of addresses); a tile that owns rows
// picks them out of the landing points and orders them in LDS.  Used whenever
// there is a sorted prefix to land in and the tail is short (else -- a cold
// start, a synchronised reset of every world -- the old path or, after three
// long tails, the radix chain).
constexpr int32_t kLandsMaxTail = 1 << 14;
constexpr int32_t kOwnCap = 2048;       // own rows a scatter tile orders in LDS
// (a tile that owns more -- worlds that multiplied their rows in one step --
// ranks them the slow way once and sends the next runs down the old path)
constexpr uint32_t kLandsBlockedRuns = 16;
__device__ inline bool compactTailByLands(int32_t prefix, int32_t tail,
                                          uint32_t blocked_runs)
{
    // (a SHORT tail: on average at most a quarter of a tile's threads get a
    // row to place.  A table whose tail is comparable to its prefix -- the
    // joint table of the physics step: ~2 K rows, a thousand of them new every
    // step, all landing in its one tile -- is better off with the one-workgroup
    // LDS radix sort of the old path)
    return prefix > 0 && tail <= kLandsMaxTail && (long long)tail * 8 <= prefix &&
        blocked_runs == 0u;
}

// a short tail (the steady state) is sorted without leaving the CU
struct TailSortLDS {
    uint32_t keys[2][kLdsTailRows];
    int32_t rows[2][kLdsTailRows];
};

// One kSmallThreads-wide workgroup: rows [prefix, n) sorted by world (stable,
// destroyed rows dropped) into the key / index buffers the gather does NOT
// read, where each of them lands in the prefix (tailLand), and for every prefix
// tile the first sorted tail row landing in it or later (tileTailStart).
// Returns the live tail rows (also left in state->tailLive).
__device__ inline int32_t compactSortTail(const SortSite &site, TableHdr &tbl,
                                          SortState *state, const uint32_t *keys,
                                          int32_t n, int32_t prefix,
                                          SmallSortLDS &lds, TailSortLDS &tail_lds)
{
    const uint32_t tid = threadIdx.x;
    auto &tail_keys_lds = tail_lds.keys;
    auto &tail_rows_lds = tail_lds.rows;

    const int32_t tail = n - prefix;
    const bool in_lds = tail <= kLdsTailRows;
    // The last pass lands in the buffer the gather does NOT read (first_out = 1
    // with { A, B }): the scatter fills the other one while tiles still read
    // the tail.
    const bool final_in_b = ((site.numPasses - 1) & 1) != 0;
    uint32_t *tail_keys = final_in_b ? site.keysA : site.keysB;
    int32_t *tail_rows = final_in_b ? site.idxA : site.idxB;
    const int final_buf = (1 + site.numPasses - 1) & 1;
    {
        // (assigned, not brace-initialised: LDS addresses are no constant
        // expressions for a static initialiser)
        RadixBuffers buffers;
        buffers.keys[0] = in_lds ? tail_keys_lds[0] : site.keysA;
        buffers.keys[1] = in_lds ? tail_keys_lds[1] : site.keysB;
        buffers.rows[0] = in_lds ? tail_rows_lds[0] : site.idxA;
        buffers.rows[1] = in_lds ? tail_rows_lds[1] : site.idxB;
        oneGroupRadixPasses(lds, site.numPasses, buffers, keys + prefix, prefix,
                            tail, 1);
    }
    __syncthreads();
    const int32_t tail_live = (int32_t)lds.valid;

    if (tid == 0) {
        state->tailLive = tail_live;
        state->statTailRows += (unsigned long long)tail;
        if (tail > tbl.tailRows) {
            tbl.tailRows = tail;
        }
    }

    // where every sorted tail row lands in the prefix: the end of its world's
    // old range
    const int32_t num_tiles = compactNumTiles(prefix);
    const int32_t *offs = tbl.worldOffsets;
    const int32_t *cnts = tbl.worldCounts;
    int32_t *land_lds = (int32_t *)tail_keys_lds[final_buf ^ 1];    // (free now)
    for (int32_t j = (int32_t)tid; j < tail_live; j += kSmallThreads) {
        const uint32_t w = in_lds ? tail_keys_lds[final_buf][j] : tail_keys[j];
        const int32_t land = offs[w] + cnts[w];
        site.tailLand[j] = land;
        if (in_lds) {
            land_lds[j] = land;
            tail_keys[j] = w;
            tail_rows[j] = tail_rows_lds[final_buf][j];
        }
    }
    __syncthreads();

    // tileTailStart[t] = first sorted tail row landing in tile t or later
    auto tile_of = [&](int32_t j) {
        const int32_t land = in_lds ? land_lds[j] : site.tailLand[j];
        const int32_t t = land >> kCompactTileShift;
        return t < num_tiles - 1 ? t : num_tiles - 1;
    };
    if (tail_live == 0) {
        for (int32_t t = (int32_t)tid; t <= num_tiles; t += kSmallThreads) {
            site.tileTailStart[t] = 0;
        }
        return 0;
    }
    for (int32_t j = (int32_t)tid; j < tail_live; j += kSmallThreads) {
        const int32_t mine = tile_of(j);
        const int32_t before = j > 0 ? tile_of(j - 1) : -1;
        for (int32_t t = before + 1; t <= mine; t++) {
            site.tileTailStart[t] = j;
        }
        if (j == tail_live - 1) {
            for (int32_t t = mine + 1; t <= num_tiles; t++) {
                site.tileTailStart[t] = tail_live;
            }
        }
    }
    return tail_live;
}

__global__ void __launch_bounds__(kSmallThreads)
sortCompactPrepare(EcsState *S, const SortSite *sites)
{
    TraceScope trace_scope(S);
    const SortSite &site = sites[blockIdx.y];
    TableHdr &tbl = S->tables[site.archetype];
    SortState *state = site.state;

    // (the header does not change before the scatter kernel publishes)
    const bool active = tbl.needsSort != 0u;
    const int32_t n = tbl.numRows;
    int32_t prefix = tbl.sortedRows;
    if (prefix < 0 || prefix > n) {
        prefix = 0;         // whatever truncated the table: everything is "tail"
    }
    const uint32_t *keys = (const uint32_t *)tbl.columns[site.keyColumn];
    const uint32_t tid = threadIdx.x;

    // how the tail is handled: every workgroup of both kernels derives the
    // same answer (header values + a counter only the NEXT kernel changes)
    const int32_t tail = n - prefix;
    const uint32_t blocked = state->landsBlocked;
    const bool by_lands = compactTailByLands(prefix, tail, blocked);
    if (blockIdx.x == 0 && tid == 0) {
        state->active = active ? 1u : 0u;
        state->rowsIn = n;
        state->keyColumn = keys;
        state->prefixRows = prefix;
        state->tailByLands = (active && by_lands) ? 1u : 0u;
    }
    if (!active) {
        return;
    }

    if (by_lands) {
        // where every (raw) tail row lands in the prefix -- the end of its
        // world's old range --, or -1 for a row destroyed again, and how many
        // land in each tile: all workgroups, a row per thread.  No sort.
        const int32_t *offs = tbl.worldOffsets;
        const int32_t *cnts = tbl.worldCounts;
        const int32_t num_tiles = compactNumTiles(prefix);
        const uint32_t lane = laneId();
        for (int32_t j0 = (int32_t)(blockIdx.x * kSmallThreads + (tid & ~63u));
             j0 < tail; j0 += (int32_t)(gridDim.x * kSmallThreads)) {
            const int32_t j = j0 + (int32_t)lane;
            int32_t land = -1, t = -1;
            if (j < tail) {
                const uint32_t w = keys[prefix + j];
                if (w != 0xFFFFFFFFu) {
                    land = offs[w] + cnts[w];
                    t = land >> kCompactTileShift;
                    t = t < num_tiles - 1 ? t : num_tiles - 1;
                }
                site.tailLand[j] = land;
            }
            // one atomic per RUN of lanes with the same tile (the rows a world
            // appended in a step are neighbours in the tail: a wavefront
            // usually holds a handful of runs, not 64 different counters)
            const int32_t t_prev = __shfl_up(t, 1, 64);
            const unsigned long long heads = ballot64(lane == 0u || t != t_prev);
            if (t >= 0 && (lane == 0u || t != t_prev)) {
                const unsigned long long later = lane == 63u ? 0ull : heads >> (lane + 1u);
                const int32_t run = later != 0ull ? (int32_t)__builtin_ctzll(later) + 1 :
                                                   64 - (int32_t)lane;
                atomicAdd(&site.tileTailStart[t], run);
            }
        }
        if (blockIdx.x == 0 && tid == 0) {
            state->statTailRows += (unsigned long long)tail;
            if (tail > tbl.tailRows) {
                tbl.tailRows = tail;
            }
        }
    }

    if (blockIdx.x != 0) {
        // ---- survivors of the prefix, per scatter tile ----
        __shared__ uint32_t tile_live[kPrepTilesPerGroup];
        const uint32_t lane = laneId();
        const uint32_t wave = tid >> 6;
        const int32_t group_rows = kPrepTilesPerGroup * kCompactTile;
        for (int32_t group = (int32_t)blockIdx.x - 1; group * group_rows < prefix;
             group += (int32_t)gridDim.x - 1) {
            if (tid < (uint32_t)kPrepTilesPerGroup) tile_live[tid] = 0;
            __syncthreads();
            // wave w: rows [512 w, 512 w + 512) of the group, lane-strided
            const int32_t base = group * group_rows + (int32_t)wave * 512;
            uint32_t live = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int32_t i = base + j * 64 + (int32_t)lane;
                live += (i < prefix && keys[i] != 0xFFFFFFFFu) ? 1u : 0u;
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
                live += __shfl_down(live, d, 64);
            }
            if (lane == 0 && live != 0) {
                atomicAdd(&tile_live[wave >> 2], live);
            }
            __syncthreads();
            if (tid < (uint32_t)kPrepTilesPerGroup) {
                const int32_t tile = group * kPrepTilesPerGroup + (int32_t)tid;
                if ((tile << kCompactTileShift) < prefix) {
                    site.tileCounts[tile] = (int32_t)tile_live[tid];
                }
            }
            __syncthreads();
        }
        return;
    }

    if (by_lands) {
        return;
    }

    // ---- workgroup 0: the tail, sorted by world ----
    __shared__ SmallSortLDS lds;
    __shared__ TailSortLDS tail_lds;
    compactSortTail(site, tbl, state, keys, n, prefix, lds, tail_lds);
}

template <int THREADS>
struct alignas(16) CompactLDS {
    uint32_t landing[kCompactTile + 4];     // tail rows landing at each position
    uint32_t liveBefore[kCompactTile + 4];  // survivors of the tile before each position
    unsigned long long scan[THREADS / 64];
    int32_t reduce[THREADS / 64];
    // tail by landing points: (world << 32 | tail index) of the tail rows that
    // land in this tile, ordered here
    unsigned long long ownKey[kOwnCap];
    uint32_t ownCount;
};

template <int THREADS>
__device__ inline int32_t blockSum(int32_t v, int32_t *scratch)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        v += __shfl_down(v, d, 64);
    }
    __syncthreads();
    if (laneId() == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    int32_t total = 0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; w++) total += scratch[w];
    return total;
}

// exclusive scan of one 64-bit value per thread across the block
template <int THREADS>
__device__ inline unsigned long long blockExclusiveScanU64(
    unsigned long long v, unsigned long long *scratch)
{
    const uint32_t lane = laneId();
    const uint32_t wave = threadIdx.x >> 6;
    unsigned long long incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        unsigned long long up = __shfl_up(incl, d, 64);
        if ((int)lane >= d) incl += up;
    }
    __syncthreads();
    if (lane == 63) scratch[wave] = incl;
    __syncthreads();
    unsigned long long wave_base = 0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; w++) {
        if (w < (int)wave) wave_base += scratch[w];
    }
    return wave_base + incl - v;
}

// this thread's ITEMS consecutive prefix rows of the tile
template <int ITEMS>
__device__ inline void loadTileKeys(const uint32_t *keys, int32_t mine, int32_t last,
                                    uint32_t (&key)[ITEMS])
{
    if (mine + ITEMS <= last) {
        if constexpr (ITEMS == 8) {
            const uint4 lo = *(const uint4 *)(keys + mine);
            const uint4 hi = *(const uint4 *)(keys + mine + 4);
            key[0] = lo.x; key[1] = lo.y; key[2] = lo.z; key[3] = lo.w;
            key[4] = hi.x; key[5] = hi.y; key[6] = hi.z; key[7] = hi.w;
        } else {
            static_assert(ITEMS == 2);
            const uint2 v = *(const uint2 *)(keys + mine);
            key[0] = v.x; key[1] = v.y;
        }
    } else {
#pragma unroll
        for (int j = 0; j < ITEMS; j++) {
            key[j] = mine + j < last ? keys[mine + j] : 0xFFFFFFFFu;
        }
    }
}

// One prefix tile [first, last) whose keys the block holds (ITEMS per thread,
// thread-major): the surviving rows and the sorted tail rows landing in the
// tile go to their place in the sorted order.  live_before_tile = survivors of
// the tiles before this one.  Ends with a barrier.
template <int THREADS, int ITEMS>
__device__ inline void compactScatterTile(const SortSite &site, CompactLDS<THREADS> &lds,
                                          int32_t tile, int32_t first,
                                          const uint32_t (&key)[ITEMS],
                                          int32_t live_before_tile,
                                          uint32_t *out_keys, int32_t *out_rows,
                                          const uint32_t *tail_keys,
                                          const int32_t *tail_rows)
{
    static_assert(THREADS * ITEMS == kCompactTile);
    const int32_t tid = (int32_t)threadIdx.x;
    const int32_t mine = first + tid * ITEMS;

    // the sorted tail rows landing in this tile (typically a handful: the
    // first one per thread stays in registers)
    const int32_t tail_first = site.tileTailStart[tile];
    const int32_t tail_end = site.tileTailStart[tile + 1];
    auto landing_at = [&](int32_t j) {
        int32_t at = site.tailLand[j] - first;
        return at < 0 ? 0 : (at > kCompactTile ? kCompactTile : at);
    };
    const int32_t my_tail = tail_first + tid;
    int32_t my_at = 0, my_row = 0;
    uint32_t my_world = 0;
    if (my_tail < tail_end) {
        my_at = landing_at(my_tail);
        my_row = tail_rows[my_tail];
        my_world = tail_keys[my_tail];
    }

    // (lds.landing was cleared before a barrier by the caller)
    if (my_tail < tail_end) {
        atomicAdd(&lds.landing[my_at], 1u);
    }
    for (int32_t j = my_tail + THREADS; j < tail_end; j += THREADS) {
        atomicAdd(&lds.landing[landing_at(j)], 1u);
    }
    __syncthreads();

    uint32_t land[ITEMS];
    if constexpr (ITEMS == 8) {
        const uint4 lo = *(const uint4 *)&lds.landing[tid * ITEMS];
        const uint4 hi = *(const uint4 *)&lds.landing[tid * ITEMS + 4];
        land[0] = lo.x; land[1] = lo.y; land[2] = lo.z; land[3] = lo.w;
        land[4] = hi.x; land[5] = hi.y; land[6] = hi.z; land[7] = hi.w;
    } else {
        const uint2 v = *(const uint2 *)&lds.landing[tid * ITEMS];
        land[0] = v.x; land[1] = v.y;
    }
    uint32_t my_live = 0, my_land = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; j++) {
        my_live += key[j] != 0xFFFFFFFFu ? 1u : 0u;
        my_land += land[j];
    }
    const unsigned long long excl = blockExclusiveScanU64<THREADS>(
        (unsigned long long)my_live | ((unsigned long long)my_land << 32),
        lds.scan);
    uint32_t live = (uint32_t)excl;             // survivors of the tile before my rows
    uint32_t landed = (uint32_t)(excl >> 32);   // owned tail rows landing before them

#pragma unroll
    for (int j = 0; j < ITEMS; j++) {
        landed += land[j];
        lds.liveBefore[tid * ITEMS + j] = live;
        if (key[j] != 0xFFFFFFFFu) {
            const int32_t dest = live_before_tile + (int32_t)live +
                tail_first + (int32_t)landed;
            out_rows[dest] = mine + j;
            out_keys[dest] = key[j];
            live += 1u;
        }
    }
    if (tid == THREADS - 1) {
        lds.liveBefore[kCompactTile] = live;    // the tile's survivors
    }
    __syncthreads();

    if (my_tail < tail_end) {
        const int32_t dest = live_before_tile +
            (int32_t)lds.liveBefore[my_at] + my_tail;
        out_rows[dest] = my_row;
        out_keys[dest] = my_world;
    }
    for (int32_t j = my_tail + THREADS; j < tail_end; j += THREADS) {
        const int32_t dest = live_before_tile +
            (int32_t)lds.liveBefore[landing_at(j)] + j;
        out_rows[dest] = tail_rows[j];
        out_keys[dest] = tail_keys[j];
    }
    __syncthreads();
}

// The same for a tail that was NOT sorted (SortState::tailByLands).  `before` =
// live tail rows landing in earlier tiles (= where this tile's start in the
// sorted tail), `own_total` = the ones landing here (both from prepare's
// per-tile counters).  The tile picks its rows out of the landing points
// (site.tailLand, -1 = destroyed), orders them by (world, tail index) -- the
// stable sort by world -- and writes them and its surviving prefix rows to
// their places.  A tile that owns more than it can order in LDS ranks them by
// counting (quadratic: not the steady state) and blocks the path for a while.
template <int THREADS, int ITEMS>
__device__ inline void compactScatterTileLands(
    const SortSite &site, SortState *state, CompactLDS<THREADS> &lds, int32_t tile,
    int32_t num_tiles, int32_t first, const uint32_t (&key)[ITEMS],
    int32_t live_before_tile, int32_t before, int32_t own_total,
    uint32_t *out_keys, int32_t *out_rows, const uint32_t *table_keys,
    int32_t prefix, int32_t tail)
{
    static_assert(THREADS * ITEMS == kCompactTile);
    const int32_t tid = (int32_t)threadIdx.x;
    const int32_t mine = first + tid * ITEMS;

    auto is_own = [&](int32_t land) {
        if (land < 0) return false;
        const int32_t t = land >> kCompactTileShift;
        return (t < num_tiles - 1 ? t : num_tiles - 1) == tile;
    };
    auto at_of = [&](int32_t land) {
        const int32_t at = land - first;
        return at < 0 ? 0 : (at > kCompactTile ? kCompactTile : at);
    };
    const bool in_lds = own_total <= kOwnCap;

    // ---- my tail rows: picked out of the landing points ----
    // (lds.landing and lds.ownCount were cleared before a barrier by the caller)
    if (own_total != 0) {
        for (int32_t j0 = tid * 4; j0 < tail; j0 += THREADS * 4) {
            int32_t land[4] = { -1, -1, -1, -1 };
            if (j0 + 4 <= tail) {
                const int4 v = *(const int4 *)(site.tailLand + j0);
                land[0] = v.x; land[1] = v.y; land[2] = v.z; land[3] = v.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    land[e] = j0 + e < tail ? site.tailLand[j0 + e] : -1;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; e++) {
                if (is_own(land[e])) {
                    atomicAdd(&lds.landing[at_of(land[e])], 1u);
                    if (in_lds) {
                        const uint32_t slot = atomicAdd(&lds.ownCount, 1u);
                        lds.ownKey[slot] =
                            ((unsigned long long)table_keys[prefix + j0 + e] << 32) |
                            (unsigned long long)(uint32_t)(j0 + e);
                    }
                }
            }
        }
    }
    __syncthreads();

    // ---- ... ordered by (world, tail index) ----
    // a handful of rows (the steady state: what one or two worlds appended):
    // every thread ranks its row by counting; more: bitonic, in LDS
    constexpr int32_t rank_by_counting = THREADS;
    if (own_total > rank_by_counting && in_lds) {
        int32_t n_pad = 2;
        while (n_pad < own_total) n_pad <<= 1;
        for (int32_t i = own_total + tid; i < n_pad; i += THREADS) {
            lds.ownKey[i] = ~0ull;
        }
        for (int32_t k = 2; k <= n_pad; k <<= 1) {
            for (int32_t jj = k >> 1; jj > 0; jj >>= 1) {
                __syncthreads();
                for (int32_t i = tid; i < n_pad; i += THREADS) {
                    const int32_t partner = i ^ jj;
                    if (partner > i) {
                        const unsigned long long a = lds.ownKey[i];
                        const unsigned long long b = lds.ownKey[partner];
                        const bool ascending = (i & k) == 0;
                        if ((a > b) == ascending) {
                            lds.ownKey[i] = b;
                            lds.ownKey[partner] = a;
                        }
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- the prefix rows of the tile (as compactScatterTile) ----
    uint32_t land[ITEMS];
    if constexpr (ITEMS == 8) {
        const uint4 lo = *(const uint4 *)&lds.landing[tid * ITEMS];
        const uint4 hi = *(const uint4 *)&lds.landing[tid * ITEMS + 4];
        land[0] = lo.x; land[1] = lo.y; land[2] = lo.z; land[3] = lo.w;
        land[4] = hi.x; land[5] = hi.y; land[6] = hi.z; land[7] = hi.w;
    } else {
        const uint2 v = *(const uint2 *)&lds.landing[tid * ITEMS];
        land[0] = v.x; land[1] = v.y;
    }
    uint32_t my_live = 0, my_land = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; j++) {
        my_live += key[j] != 0xFFFFFFFFu ? 1u : 0u;
        my_land += land[j];
    }
    const unsigned long long excl = blockExclusiveScanU64<THREADS>(
        (unsigned long long)my_live | ((unsigned long long)my_land << 32),
        lds.scan);
    uint32_t live = (uint32_t)excl;
    uint32_t landed = (uint32_t)(excl >> 32);
#pragma unroll
    for (int j = 0; j < ITEMS; j++) {
        landed += land[j];
        lds.liveBefore[tid * ITEMS + j] = live;
        if (key[j] != 0xFFFFFFFFu) {
            const int32_t dest = live_before_tile + (int32_t)live + before +
                (int32_t)landed;
            out_rows[dest] = mine + j;
            out_keys[dest] = key[j];
            live += 1u;
        }
    }
    if (tid == THREADS - 1) {
        lds.liveBefore[kCompactTile] = live;
    }
    __syncthreads();

    // ---- my tail rows to their places ----
    if (in_lds && own_total <= rank_by_counting) {
        if (tid < own_total) {
            const unsigned long long k = lds.ownKey[tid];
            int32_t r = 0;
            for (int32_t o = 0; o < own_total; o++) {
                r += lds.ownKey[o] < k ? 1 : 0;
            }
            const int32_t j = (int32_t)(uint32_t)k;
            const int32_t dest = live_before_tile +
                (int32_t)lds.liveBefore[at_of(site.tailLand[j])] + before + r;
            out_rows[dest] = prefix + j;
            out_keys[dest] = (uint32_t)(k >> 32);
        }
    } else if (in_lds) {
        for (int32_t r = tid; r < own_total; r += THREADS) {
            const unsigned long long k = lds.ownKey[r];
            const int32_t j = (int32_t)(uint32_t)k;
            const int32_t dest = live_before_tile +
                (int32_t)lds.liveBefore[at_of(site.tailLand[j])] + before + r;
            out_rows[dest] = prefix + j;
            out_keys[dest] = (uint32_t)(k >> 32);
        }
    } else {
        // more rows than the tile orders in LDS: rank each by counting the own
        // rows that sort before it
        if (tid == 0) {
            state->landsBlocked = kLandsBlockedRuns;
        }
        for (int32_t j = tid; j < tail; j += THREADS) {
            const int32_t land_j = site.tailLand[j];
            if (!is_own(land_j)) continue;
            const uint32_t w = table_keys[prefix + j];
            int32_t rank = 0;
            for (int32_t k = 0; k < tail; k++) {
                if (!is_own(site.tailLand[k])) continue;
                const uint32_t other = table_keys[prefix + k];
                rank += (other < w || (other == w && k < j)) ? 1 : 0;
            }
            const int32_t dest = live_before_tile +
                (int32_t)lds.liveBefore[at_of(land_j)] + before + rank;
            out_rows[dest] = prefix + j;
            out_keys[dest] = w;
        }
    }
    __syncthreads();
}

__global__ void __launch_bounds__(kSortThreads)
sortCompactScatter(EcsState *S, const SortSite *sites)
{
    TraceScope trace_scope(S);
    const SortSite &site = sites[blockIdx.y];
    TableHdr &tbl = S->tables[site.archetype];
    SortState *state = site.state;
    if (state->active == 0u) {
        return;
    }

    __shared__ CompactLDS<kSortThreads> lds;

    const int32_t n = state->rowsIn;
    const int32_t prefix = state->prefixRows;
    const bool by_lands = state->tailByLands != 0u;
    const int32_t tail = n - prefix;
    int32_t tail_live = by_lands ? 0 : state->tailLive;
    const int32_t num_tiles = compactNumTiles(prefix);
    const uint32_t *keys = state->keyColumn;
    const int32_t tid = (int32_t)threadIdx.x;

    const bool final_in_b = ((site.numPasses - 1) & 1) != 0;
    uint32_t *out_keys = final_in_b ? site.keysB : site.keysA;
    int32_t *out_rows = final_in_b ? site.idxB : site.idxA;
    const uint32_t *tail_keys = final_in_b ? site.keysA : site.keysB;
    const int32_t *tail_rows = final_in_b ? site.idxA : site.idxB;
    auto tile_count = [&](int32_t t) {
        return (t << kCompactTileShift) < prefix ? site.tileCounts[t] : 0;
    };

    auto tile_lands = [&](int32_t t) {
        return by_lands ? site.tileTailStart[t] : 0;
    };

    if (blockIdx.x == 0) {
        // rows the sorted table has = survivors of the prefix + live tail rows
        int32_t part = 0, lands = 0;
        for (int32_t t = tid; t < num_tiles; t += kSortThreads) {
            part += tile_count(t);
            lands += tile_lands(t);
        }
        const int32_t survivors = blockSum<kSortThreads>(part, lds.reduce);
        if (by_lands) {
            tail_live = blockSum<kSortThreads>(lands, lds.reduce);
            if (tid == 0) {
                state->tailLive = tail_live;
            }
        } else if (tid == 0 && state->landsBlocked != 0u) {
            state->landsBlocked -= 1u;
        }
        if (tid == 0) {
            state->numValid = (uint32_t)(survivors + tail_live);
        }
        __syncthreads();
        publishSite(S, site, tbl, n);
        __syncthreads();
    }

    for (int32_t tile = (int32_t)blockIdx.x; tile < num_tiles;
         tile += (int32_t)gridDim.x) {
        const int32_t first = tile << kCompactTileShift;
        const int32_t last = min(first + kCompactTile, prefix);

        for (int32_t i = tid; i < kCompactTile + 4; i += kSortThreads) {
            lds.landing[i] = 0;
        }
        uint32_t key[kSortItems];
        loadTileKeys<kSortItems>(keys, first + tid * kSortItems, last, key);

        if (tid == 0) {
            lds.ownCount = 0;
        }
        // survivors of the tiles before this one, and the live tail rows that
        // land in them (the sums' barriers also order the clearing of
        // lds.landing before the atomics of the tile)
        int32_t part = 0, lands = 0;
        for (int32_t t = tid; t < tile; t += kSortThreads) {
            part += tile_count(t);
            lands += tile_lands(t);
        }
        const int32_t live_before_tile = blockSum<kSortThreads>(part, lds.reduce);

        if (by_lands) {
            const int32_t before = blockSum<kSortThreads>(lands, lds.reduce);
            compactScatterTileLands<kSortThreads, kSortItems>(site, state, lds, tile,
                num_tiles, first, key, live_before_tile, before, tile_lands(tile),
                out_keys, out_rows, keys, prefix, tail);
        } else {
            compactScatterTile<kSortThreads, kSortItems>(site, lds, tile, first, key,
                live_before_tile, out_keys, out_rows, tail_keys, tail_rows);
        }
    }
}

// (Round 3 also built prepare + scatter as ONE launch -- workgroup 0 handing the
// sorted tail to the tile workgroups through an agent-scope release / acquire
// pair and tagged granules, tiles exchanging their survivor counts the same
// way -- and measured it slower: 34 us against 13 + 10.5 at 264 K rows, 236
// against 85 + 18 at 1.8 M (profiles/r03_sort_variants_plan_kernel.jsonl).
// Every tile polling its predecessors' counts puts T^2 / 2 uncached loads on a
// handful of cache lines, and with that fixed the hand-off itself -- header
// broadcast, fence, flag, poll -- costs about the launch floor it removes.
// Not kept.)
}

// ---------------------------------------------------------------------------
// Host side: expand a batch of sort sites into kernel launches
// ---------------------------------------------------------------------------
int sortNumPasses(bool world_sort, uint32_t num_worlds)
{
    if (!world_sort) {
        return 4;
    }
    // bits needed so that W live world ids and the all-ones "destroyed" key
    // stay distinct after truncation (reference sort_archetype.cpp:1432-1438)
    uint32_t bits = 32u - (uint32_t)__builtin_clz(num_worlds + 1u);
    return (int)((bits + kRadixBits - 1) / kRadixBits);
}

uint32_t sortTileSize() { return (uint32_t)kSortTile; }


uint32_t sortSmallRowLimit() { return kSmallSortRows; }

// rows from which a table that keeps being re-sorted is better off with the
// compaction chain than with the one-launch sort (runtime.hip, sortsOutgrown)
uint32_t sortSmallBusyRows() { return 768u; }

// rows behind the sorted prefix one workgroup sorts about as fast as the radix
// chain would take for the whole table; tables that keep exceeding it go back
// to the radix chain (runtime.hip, sortsOutgrown)
uint32_t sortCompactTailLimit() { return 16384u; }

// Key-pass workgroups spin on their predecessor tiles: all of a grid must be
// able to be resident at once (4 workgroups of 22.5 KB LDS / 4 waves per CU on
// 256 CUs; the kernels fit 6), larger tables take rounds.
// MADRONA_MWHIP_SORT_MAX_GRID overrides it (tests: grids smaller than the tile
// count).
static uint32_t sortMaxGrid()
{
    // (read when a graph is built, not cached: tests switch it per executor)
    const char *e = getenv("MADRONA_MWHIP_SORT_MAX_GRID");
    const uint32_t n = e != nullptr ? (uint32_t)strtoul(e, nullptr, 10) : 0u;
    return n == 0 ? 1024u : n;
}

void buildSortLaunches(const SortBatch &batch, std::vector<KernelLaunch> &out)
{
    const uint32_t num_sites = (uint32_t)batch.sites.size();

    // every table of the batch is small: the whole node in one launch
    if (batch.small) {
        KernelLaunch k;
        k.fn = (const void *)&sortSmall;
        k.grid = dim3(num_sites, 1, 1);
        k.block = dim3(kSmallThreads, 1, 1);
        k.setArgs(batch.stateDev, batch.sitesDev, batch.gatherColumnsDev,
                  batch.numGatherColumns, (const MiscOp *)nullptr, 0u);
        k.role = "sort.small";
        k.kind = MWHIP_NODE_SORT_ARCHETYPE;
        k.sortBatch = &batch;
        k.sortRole = SortRole::Small;
        out.push_back(k);
        return;
    }

    uint32_t max_capacity = 0;
    int max_passes = 0;
    for (const SortSiteHost &s : batch.sites) {
        max_capacity = std::max(max_capacity, s.capacity);
        max_passes = std::max(max_passes, s.numPasses);
    }

    const uint32_t tiles = std::min<uint32_t>(
        (max_capacity + kSortTile - 1) / kSortTile, sortMaxGrid());
    const uint32_t stream_blocks = std::min<uint32_t>(
        std::max<uint32_t>((max_capacity + kSortThreads * 4 - 1) /
                           (kSortThreads * 4), 1u), 1024u);

    if (batch.compact) {
        {
            KernelLaunch k;
            k.fn = (const void *)&sortCompactPrepare;
            const uint32_t group_rows = kPrepTilesPerGroup * kCompactTile;
            k.grid = dim3(1u + std::min<uint32_t>(std::max<uint32_t>(
                (max_capacity + group_rows - 1) / group_rows, 1u), sortMaxGrid()),
                num_sites, 1);
            k.block = dim3(kSmallThreads, 1, 1);
            k.setArgs(batch.stateDev, batch.sitesDev);
            k.role = "sort.compact.prepare";
            k.kind = MWHIP_NODE_SORT_ARCHETYPE;
            k.sortBatch = &batch;
            k.sortRole = SortRole::CompactPrepare;
            out.push_back(k);
        }
        {
            KernelLaunch k;
            k.fn = (const void *)&sortCompactScatter;
            k.grid = dim3(std::max(tiles, 1u), num_sites, 1);
            k.block = dim3(kSortThreads, 1, 1);
            k.setArgs(batch.stateDev, batch.sitesDev);
            k.role = "sort.compact.scatter";
            k.kind = MWHIP_NODE_SORT_ARCHETYPE;
            k.sortBatch = &batch;
            k.sortRole = SortRole::CompactScatter;
            out.push_back(k);
        }
    } else {
    {
        KernelLaunch k;
        k.fn = (const void *)&sortHistogram;
        // few, fat workgroups: every workgroup ends with one atomic per
        // non-empty bin on the site's global histogram, and atomics on one
        // address serialise (profiles/tools/atomic_microbench.hip)
        k.grid = dim3(std::min<uint32_t>(stream_blocks, 64u), num_sites, 1);
        k.block = dim3(kSortThreads, 1, 1);
        k.setArgs(batch.stateDev, batch.sitesDev);
        k.role = "sort.histogram";
        k.kind = MWHIP_NODE_SORT_ARCHETYPE;
        k.sortBatch = &batch;
        k.sortRole = SortRole::Histogram;
        out.push_back(k);
    }

    for (int p = 0; p < max_passes; p++) {
        KernelLaunch k;
        k.fn = (const void *)&sortOnesweep;
        k.grid = dim3(std::max(tiles, 1u), num_sites, 1);
        k.block = dim3(kSortThreads, 1, 1);
        k.setArgs(batch.stateDev, batch.sitesDev, (uint32_t)p);
        k.role = "sort.onesweep";
        k.kind = MWHIP_NODE_SORT_ARCHETYPE;
        k.sortBatch = &batch;
        k.sortRole = SortRole::Onesweep;
        out.push_back(k);
    }
    }   // radix chain

    {
        KernelLaunch k;
        k.fn = batch.gatherWide ? (const void *)&sortGather<true> :
                                  (const void *)&sortGather<false>;
        // one workgroup per slice of a column (runtime.hip, makeSortBatch: the
        // workgroups are shared out over the columns by the bytes they move)
        k.grid = dim3(std::max(batch.numGatherSlices, 1u), 1, 1);
        k.block = dim3(kSortThreads, 1, 1);
        k.setArgs(batch.stateDev, batch.sitesDev, batch.gatherColumnsDev,
                  batch.gatherSlicesDev, (const MiscOp *)nullptr, 0u);
        k.role = "sort.gather";
        k.kind = MWHIP_NODE_SORT_ARCHETYPE;
        k.sortBatch = &batch;
        k.sortRole = SortRole::Gather;
        out.push_back(k);
    }

    if (batch.hasPinned) {
        KernelLaunch k;
        k.fn = (const void *)&sortFinalize;
        // W-sized fix-up + (rare) pinned-column copy-back: keep the grid small,
        // every block pays a device-scope fence for the last-block hand-off
        k.grid = dim3(std::min<uint32_t>(stream_blocks, 16u), num_sites, 1);
        k.block = dim3(kSortThreads, 1, 1);
        k.setArgs(batch.stateDev, batch.sitesDev, (const MiscOp *)nullptr, 0u);
        k.role = "sort.finalize";
        k.kind = MWHIP_NODE_SORT_ARCHETYPE;
        k.sortBatch = &batch;
        k.sortRole = SortRole::Finalize;
        out.push_back(k);
    }
}

}
}

// Part of the fused rigid-body step (phys_impl/world_step.inl includes the parts
// in order, inside namespace madrona::phys::kernels): wavefront primitives of the fused step (groups of LPW lanes).

namespace wave {

__device__ inline uint32_t laneID()
{
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// phase boundary inside a wave: earlier global writes of any lane become
// visible to later reads of every lane (same CU), and the compiler may not
// move memory operations across it
__device__ inline void phaseFence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// The helpers below work on groups of LPW consecutive lanes (LPW = 64: the whole
// wavefront; 32: two worlds share a wavefront, one per half).  `lane` is the
// index inside the group; the lanes of a group are always converged when they
// get here, the other group of the wavefront need not be.
template <int LPW = 64>
__device__ inline uint32_t exclusiveScan(uint32_t v, uint32_t lane,
                                         uint32_t *total)
{
    uint32_t incl = v;
#pragma unroll
    for (uint32_t d = 1; d < (uint32_t)LPW; d <<= 1) {
        uint32_t up = __shfl_up(incl, d, LPW);
        if (lane >= d) incl += up;
    }
    *total = __shfl(incl, LPW - 1, LPW);
    return incl - v;
}

// A lane's partner in the four steps that reduce a row of 16 lanes: the other
// lane of its pair, the other pair of its quad, the mirror lane of its eight,
// the mirror lane of its row -- DPP moves (a VALU operand modifier: no LDS
// crossbar round trip, where __shfl_xor is a ds_bpermute of ~100 cycles a step).
// After the four every lane holds the row's result; groups wider than a row
// finish with __shfl_xor.  The reductions below are commutative and associative
// (ties included), so the order of the steps does not show in the result.
template <int Ctrl>
__device__ inline uint32_t dppMove(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, Ctrl, 0xF, 0xF, true);
}
template <int Ctrl>
__device__ inline float dppMove(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(
        __builtin_bit_cast(int, v), Ctrl, 0xF, 0xF, true));
}
inline constexpr int dppSwapPairs = 0xB1;       // quad_perm:[1,0,3,2]
inline constexpr int dppSwapHalfQuads = 0x4E;   // quad_perm:[2,3,0,1]
inline constexpr int dppHalfMirror = 0x141;     // row_half_mirror
inline constexpr int dppMirror = 0x140;         // row_mirror

template <int LPW = 64>
__device__ inline uint32_t maxReduce(uint32_t v)
{
    static_assert(LPW >= 16);
    auto with = [&](uint32_t o) { v = o > v ? o : v; };
    with(dppMove<dppSwapPairs>(v));
    with(dppMove<dppSwapHalfQuads>(v));
    with(dppMove<dppHalfMirror>(v));
    with(dppMove<dppMirror>(v));
#pragma unroll
    for (uint32_t d = 16; d < (uint32_t)LPW; d <<= 1) {
        with((uint32_t)__shfl_xor(v, d, LPW));
    }
    return v;
}

// ballot over the lanes of my group: bit i = lane i of the group
template <int LPW = 64>
__device__ inline uint64_t groupBallot(bool pred)
{
    const uint64_t all = __builtin_amdgcn_ballot_w64(pred);
    if constexpr (LPW == 64) {
        return all;
    } else {
        const uint32_t first = laneID() & ~(uint32_t)(LPW - 1);
        return (all >> first) & ((1ull << LPW) - 1ull);
    }
}

// set bits of a group ballot below my lane
__device__ inline uint32_t rankInGroup(uint64_t mask, uint32_t lane)
{
    return (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
}

// position of the n-th set bit of a group ballot (n < its population count)
template <int LPW = 64>
__device__ inline uint32_t nthSetBit(uint64_t mask, uint32_t n)
{
    uint32_t pos = 0;
    if constexpr (LPW == 64) {
        const uint32_t low = (uint32_t)__builtin_popcount((uint32_t)mask);
        const bool high = n >= low;
        n = high ? n - low : n;
        pos = high ? 32u : 0u;
        mask = high ? mask >> 32 : mask;
    }
    uint32_t m = (uint32_t)mask;
#pragma unroll
    for (uint32_t width = 16; width != 0; width >>= 1) {
        const uint32_t below = (uint32_t)__builtin_popcount(m & ((1u << width) - 1u));
        const bool up = n >= below;
        n = up ? n - below : n;
        pos += up ? width : 0u;
        m = up ? m >> width : m;
    }
    return pos;
}

// arg-max over the wave where the LOWEST index wins among equal values -- the
// result of a sequential "if (v > best)" scan in index order.  Every lane
// returns the winner.  (Lane-local values are never NaN: they start at
// -FLT_MAX and are only replaced through a strict >.)
template <int LPW = 64>
__device__ inline void argMaxFirst(float &v, uint32_t &idx)
{
    static_assert(LPW >= 16);
    auto with = [&](float ov, uint32_t oi) {
        if (ov > v || (ov == v && oi < idx)) {
            v = ov;
            idx = oi;
        }
    };
    with(dppMove<dppSwapPairs>(v), dppMove<dppSwapPairs>(idx));
    with(dppMove<dppSwapHalfQuads>(v), dppMove<dppSwapHalfQuads>(idx));
    with(dppMove<dppHalfMirror>(v), dppMove<dppHalfMirror>(idx));
    with(dppMove<dppMirror>(v), dppMove<dppMirror>(idx));
#pragma unroll
    for (uint32_t d = 16; d < (uint32_t)LPW; d <<= 1) {
        with(__shfl_xor(v, d, LPW), (uint32_t)__shfl_xor(idx, d, LPW));
    }
}

__device__ inline uint32_t rankInBallot(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
        __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

}

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

template <int LPW = 64>
__device__ inline uint32_t maxReduce(uint32_t v)
{
#pragma unroll
    for (uint32_t d = LPW / 2; d > 0; d >>= 1) {
        uint32_t o = __shfl_xor(v, d, LPW);
        v = o > v ? o : v;
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
#pragma unroll
    for (uint32_t d = LPW / 2; d > 0; d >>= 1) {
        float ov = __shfl_xor(v, d, LPW);
        uint32_t oi = __shfl_xor(idx, d, LPW);
        if (ov > v || (ov == v && oi < idx)) {
            v = ov;
            idx = oi;
        }
    }
}

__device__ inline uint32_t rankInBallot(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
        __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

}

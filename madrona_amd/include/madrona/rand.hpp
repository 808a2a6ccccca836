// Counter-based RNG (Threefry2x32, 20 rounds, JAX-style key splitting).
// API contract: reference include/madrona/rand.hpp:9-60; bit-exact with
// rand.inl (known answers from the reference's tests/rand.cpp are pinned in
// tests/test_oracle_golden.py through the C restatement in oracle/restate).
#pragma once

#include <madrona/macros.hpp>
#include <madrona/math.hpp>
#include <madrona/types.hpp>
#include <madrona/utils.hpp>

namespace madrona {

struct RandKey {
    uint32_t a;
    uint32_t b;
};

namespace rand {

namespace detail {

MADRONA_HD constexpr inline uint32_t rotl32(uint32_t v, uint32_t d)
{
    return (v << d) | (v >> (32u - d));
}

// four Threefry mix rounds with the given rotation constants
MADRONA_HD constexpr inline void mix4(uint32_t &x0, uint32_t &x1,
                                      uint32_t r0, uint32_t r1,
                                      uint32_t r2, uint32_t r3)
{
    x0 += x1; x1 = rotl32(x1, r0) ^ x0;
    x0 += x1; x1 = rotl32(x1, r1) ^ x0;
    x0 += x1; x1 = rotl32(x1, r2) ^ x0;
    x0 += x1; x1 = rotl32(x1, r3) ^ x0;
}

}

// Threefry2x32-20 keyed by src, applied to the counter (idx, idx_upper)
MADRONA_HD constexpr inline RandKey split_i(RandKey src, uint32_t idx,
                                            uint32_t idx_upper = 0)
{
    const uint32_t k0 = src.a;
    const uint32_t k1 = src.b;
    const uint32_t k2 = 0x1BD11BDAu ^ k0 ^ k1;

    uint32_t x0 = idx + k0;
    uint32_t x1 = idx_upper + k1;

    detail::mix4(x0, x1, 13, 15, 26, 6);
    x0 += k1; x1 += k2 + 1u;
    detail::mix4(x0, x1, 17, 29, 16, 24);
    x0 += k2; x1 += k0 + 2u;
    detail::mix4(x0, x1, 13, 15, 26, 6);
    x0 += k0; x1 += k1 + 3u;
    detail::mix4(x0, x1, 17, 29, 16, 24);
    x0 += k1; x1 += k2 + 4u;
    detail::mix4(x0, x1, 13, 15, 26, 6);
    x0 += k2; x1 += k0 + 5u;

    return RandKey { x0, x1 };
}

MADRONA_HD constexpr inline RandKey initKey(uint32_t seed,
                                            uint32_t seed_upper = 0)
{
    return split_i(RandKey { seed, seed_upper }, 0);
}

MADRONA_HD constexpr inline uint32_t bits32(RandKey k) { return k.a ^ k.b; }

MADRONA_HD constexpr inline uint64_t bits64(RandKey k)
{
    return ((uint64_t)k.b << 32) | (uint64_t)k.a;
}

// [0, 1): 24 random mantissa bits
MADRONA_HD constexpr inline float bitsToFloat01(uint32_t rand_bits)
{
    return (float)(rand_bits >> 8) * 0x1p-24f;
}

// Unbiased integer in [a, b) (Lemire's multiply-shift with rejection)
MADRONA_HD constexpr inline int32_t sampleI32(RandKey k, int32_t a, int32_t b)
{
    const uint32_t s = (uint32_t)(b - a);

    uint64_t prod = (uint64_t)bits32(k) * (uint64_t)s;
    uint32_t lo = (uint32_t)prod;

    if (lo < s) {
        const uint32_t threshold = (0u - s) % s;
        while (lo < threshold) {
            k = split_i(k, 0);
            prod = (uint64_t)bits32(k) * (uint64_t)s;
            lo = (uint32_t)prod;
        }
    }

    return (int32_t)(uint32_t)(prod >> 32) + a;
}

MADRONA_HD constexpr inline int32_t sampleI32Biased(RandKey k, int32_t a,
                                                    int32_t b)
{
    return (int32_t)utils::u32mulhi(bits32(k), (uint32_t)(b - a));
}

MADRONA_HD constexpr inline float sampleUniform(RandKey k)
{
    return bitsToFloat01(bits32(k));
}

MADRONA_HD constexpr inline bool sampleBool(RandKey k)
{
    return (__builtin_popcount(bits32(k)) & 1) == 0;
}

MADRONA_HD constexpr inline math::Vector2 sample2xUniform(RandKey k)
{
    return math::Vector2 { bitsToFloat01(k.a), bitsToFloat01(k.b) };
}

}

// Stateful convenience wrapper: sample i uses split_i(key, i)
class RNG {
public:
    MADRONA_HD inline RNG() : k_ { 0, 0 }, count_(0) {}
    MADRONA_HD inline RNG(RandKey k) : k_(k), count_(0) {}
    MADRONA_HD inline RNG(uint32_t seed) : RNG(rand::initKey(seed)) {}

    MADRONA_HD inline int32_t sampleI32(int32_t a, int32_t b) { return rand::sampleI32(advance(), a, b); }
    MADRONA_HD inline int32_t sampleI32Biased(int32_t a, int32_t b) { return rand::sampleI32Biased(advance(), a, b); }
    MADRONA_HD inline float sampleUniform() { return rand::sampleUniform(advance()); }
    MADRONA_HD inline bool sampleBool() { return rand::sampleBool(advance()); }
    MADRONA_HD inline RandKey randKey() { return advance(); }

private:
    MADRONA_HD inline RandKey advance() { return rand::split_i(k_, count_++); }

    RandKey k_;
    uint32_t count_;
};

}

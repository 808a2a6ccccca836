/* TEST INFRASTRUCTURE ONLY.  Restates reference include/madrona/rand.inl. */
#include "oracle_restate.h"

static uint32_t rotl(uint32_t v, uint32_t d) { return (v << d) | (v >> (32u - d)); }

/* rand.inl:16-108 (split_i): Threefry2x32, 20 rounds, key = src, counter =
 * (idx, idx_upper) */
void oracle_split_i(uint32_t key_a, uint32_t key_b, uint32_t idx,
                    uint32_t idx_upper, uint32_t *out_a, uint32_t *out_b)
{
    static const uint32_t rot[8] = { 13, 15, 26, 6, 17, 29, 16, 24 };
    uint32_t ks[3] = { key_a, key_b, 0x1BD11BDAu ^ key_a ^ key_b };
    uint32_t x0 = idx + ks[0];
    uint32_t x1 = idx_upper + ks[1];

    for (int group = 0; group < 5; group++) {
        const uint32_t *r = (group & 1) ? rot + 4 : rot;
        for (int i = 0; i < 4; i++) {
            x0 += x1;
            x1 = rotl(x1, r[i]);
            x1 ^= x0;
        }
        x0 += ks[(group + 1) % 3];
        x1 += ks[(group + 2) % 3] + (uint32_t)(group + 1);
    }

    *out_a = x0;
    *out_b = x1;
}

/* rand.inl:110-113 */
uint32_t oracle_bits32(uint32_t a, uint32_t b) { return a ^ b; }

/* rand.inl:120-165 (Lemire's unbiased bounded integer with rejection) */
int32_t oracle_sample_i32(uint32_t a, uint32_t b, int32_t lo, int32_t hi)
{
    uint32_t s = (uint32_t)(hi - lo);
    uint32_t x = oracle_bits32(a, b);
    uint64_t m = (uint64_t)x * (uint64_t)s;
    uint32_t l = (uint32_t)m;

    if (l < s) {
        uint32_t t = (0u - s) % s;
        while (l < t) {
            oracle_split_i(a, b, 0, 0, &a, &b);
            x = oracle_bits32(a, b);
            m = (uint64_t)x * (uint64_t)s;
            l = (uint32_t)m;
        }
    }
    return (int32_t)(uint32_t)(m >> 32) + lo;
}

/* rand.inl:167-173 */
int32_t oracle_sample_i32_biased(uint32_t a, uint32_t b, int32_t lo, int32_t hi)
{
    uint32_t s = (uint32_t)(hi - lo);
    return (int32_t)(uint32_t)(((uint64_t)oracle_bits32(a, b) * s) >> 32);
}

/* rand.inl:200-214 */
float oracle_bits_to_float01(uint32_t bits)
{
    return (float)(bits >> 8) * 0x1p-24f;
}

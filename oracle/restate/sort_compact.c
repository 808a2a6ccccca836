/* TEST INFRASTRUCTURE ONLY.  Restates what one SortArchetypeNode /
 * CompactArchetypeNode invocation must produce:
 * reference GPU pipeline src/mw/device/sort_archetype.cpp:977-1405 (stable LSD
 * radix sort of (key, row index), rows with key 0xFFFFFFFF dropped by
 * resizeTable :1246-1257, world ranges :1269-1337, column moves :1339-1405);
 * on the CPU backend the same table state results from compactArchetype per
 * world (src/core/state.cpp:724-769). */
#include "oracle_restate.h"

#include <stdlib.h>
#include <string.h>

int32_t oracle_sort_perm(const uint32_t *keys, int32_t n, int32_t drop_invalid,
                         int32_t *perm_out, int32_t num_worlds,
                         int32_t *world_offsets, int32_t *world_counts)
{
    /* stable LSD radix sort, 8-bit digits, 4 passes over the full key */
    int32_t *a = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    int32_t *b = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    for (int32_t i = 0; i < n; i++) a[i] = i;

    for (int pass = 0; pass < 4; pass++) {
        int32_t bins[257] = { 0 };
        for (int32_t i = 0; i < n; i++) {
            bins[((keys[a[i]] >> (8 * pass)) & 0xFFu) + 1]++;
        }
        for (int d = 0; d < 256; d++) bins[d + 1] += bins[d];
        for (int32_t i = 0; i < n; i++) {
            b[bins[(keys[a[i]] >> (8 * pass)) & 0xFFu]++] = a[i];
        }
        int32_t *t = a; a = b; b = t;
    }

    int32_t n_out = n;
    if (drop_invalid) {
        n_out = 0;
        while (n_out < n && keys[a[n_out]] != 0xFFFFFFFFu) n_out++;
    }
    memcpy(perm_out, a, sizeof(int32_t) * (size_t)n_out);

    if (drop_invalid && world_offsets && world_counts) {
        /* clearWorldOffsetsAndCounts + computeWorldCounts + correctWorldCounts */
        for (int32_t w = 0; w < num_worlds; w++) {
            world_offsets[w] = n_out;
            world_counts[w] = 0;
        }
        for (int32_t i = 0; i < n_out; i++) {
            uint32_t k = keys[a[i]];
            if (i == 0 || keys[a[i - 1]] != k) world_offsets[k] = i;
            world_counts[k] += 1;
        }
    }

    free(a);
    free(b);
    return n_out;
}

void oracle_gather_column(const void *src, void *dst, const int32_t *perm,
                          int32_t n_out, uint32_t elem_bytes)
{
    for (int32_t i = 0; i < n_out; i++) {
        memcpy((char *)dst + (size_t)i * elem_bytes,
               (const char *)src + (size_t)perm[i] * elem_bytes, elem_bytes);
    }
}

/* TEST INFRASTRUCTURE ONLY -- never linked into or called by the product.
 *
 * Plain-C restatement of the kernel-level contracts of the hot path, each
 * function citing the reference code it follows.  Pinned (tests/
 * test_oracle_golden.py) against
 *   - the known answers in the reference's own tests (tests/rand.cpp:131-141),
 *   - the published Threefry2x32-20 known-answer vectors (Random123 / JAX),
 *   - the reference's real EntityStore/IDMap and CPU compactArchetype, driven
 *     through oracle/ref_shims (compiled from /root/reference into oracle/_ref).
 */
#ifndef ORACLE_RESTATE_H
#define ORACLE_RESTATE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- rand (reference include/madrona/rand.inl) ---- */
void oracle_split_i(uint32_t key_a, uint32_t key_b, uint32_t idx,
                    uint32_t idx_upper, uint32_t *out_a, uint32_t *out_b);
uint32_t oracle_bits32(uint32_t a, uint32_t b);
int32_t oracle_sample_i32(uint32_t a, uint32_t b, int32_t lo, int32_t hi);
int32_t oracle_sample_i32_biased(uint32_t a, uint32_t b, int32_t lo, int32_t hi);
float oracle_bits_to_float01(uint32_t bits);

/* ---- entity id allocation (reference include/madrona/impl/id_map_impl.inl) ---- */
typedef struct oracle_idmap oracle_idmap;
oracle_idmap *oracle_idmap_create(uint32_t num_caches, uint32_t capacity);
void oracle_idmap_destroy(oracle_idmap *m);
/* returns id, writes generation */
int32_t oracle_idmap_acquire(oracle_idmap *m, uint32_t cache, uint32_t *gen_out);
void oracle_idmap_release(oracle_idmap *m, uint32_t cache, int32_t id);

/* ---- sort / compact node contract (reference
 * src/mw/device/sort_archetype.cpp, CPU src/core/state.cpp:724-769) ----
 * keys[n]: u32 sort keys (0xFFFFFFFF = destroyed row when drop_invalid).
 * perm_out[n_out]: source row of each output row (stable ascending order).
 * world_offsets/world_counts[num_worlds]: only when drop_invalid (world sort).
 * Returns n_out. */
int32_t oracle_sort_perm(const uint32_t *keys, int32_t n, int32_t drop_invalid,
                         int32_t *perm_out, int32_t num_worlds,
                         int32_t *world_offsets, int32_t *world_counts);
/* gathers one column: dst[i] = src[perm[i]] (elem_bytes each) */
void oracle_gather_column(const void *src, void *dst, const int32_t *perm,
                          int32_t n_out, uint32_t elem_bytes);

#ifdef __cplusplus
}
#endif
#endif

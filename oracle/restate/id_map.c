/* TEST INFRASTRUCTURE ONLY.  Restates the reference's entity id allocator:
 * IDMap<Entity, Loc, ...> (include/madrona/impl/id_map_impl.inl) as used by
 * EntityStore (src/core/state.cpp:62-84) with one Cache per world
 * (src/mw/cpu_exec.cpp:110-116).  Single threaded (the oracle runs with
 * numWorkers=1), so the lock-free list heads are plain variables. */
#include "oracle_restate.h"

#include <stdlib.h>

#define IDS_PER_CACHE 64
#define SENTINEL (-1)

typedef struct {
    int32_t sub_next;     /* FreeNode::subNext  */
    int32_t global_next;  /* FreeNode::globalNext (also run length in a cache) */
    uint32_t gen;
} node_t;

typedef struct {
    int32_t free_head, num_free, overflow_head, num_overflow;
} cache_t;

struct oracle_idmap {
    node_t *nodes;
    uint32_t capacity;
    uint32_t num_ids;       /* LockedMapStore::numIDs */
    int32_t global_head;    /* free_head_.head */
    cache_t *caches;
    uint32_t num_caches;
};

oracle_idmap *oracle_idmap_create(uint32_t num_caches, uint32_t capacity)
{
    oracle_idmap *m = (oracle_idmap *)calloc(1, sizeof(*m));
    m->nodes = (node_t *)calloc(capacity, sizeof(node_t));
    m->capacity = capacity;
    m->num_ids = 0;                 /* EntityStore(): map_(0) */
    m->global_head = SENTINEL;
    m->caches = (cache_t *)calloc(num_caches, sizeof(cache_t));
    m->num_caches = num_caches;
    for (uint32_t i = 0; i < num_caches; i++) {
        /* IDMap::Cache::Cache(), id_map_impl.inl:16-22 */
        m->caches[i].free_head = SENTINEL;
        m->caches[i].overflow_head = SENTINEL;
    }
    return m;
}

void oracle_idmap_destroy(oracle_idmap *m)
{
    if (!m) return;
    free(m->nodes);
    free(m->caches);
    free(m);
}

/* assignCachedID, id_map_impl.inl:73-101 */
static int32_t assign_cached(oracle_idmap *m, int32_t *head, uint32_t *gen_out)
{
    int32_t new_id = *head;
    node_t *node = &m->nodes[new_id];
    int32_t num_contiguous = node->global_next;

    if (num_contiguous == 1) {
        *head = node->sub_next;
    } else {
        int32_t next_free = new_id + 1;
        node_t *next = &m->nodes[next_free];
        next->sub_next = node->sub_next;
        next->global_next = num_contiguous - 1;
        next->gen = 0;
        *head = next_free;
    }

    *gen_out = node->gen;
    return new_id;
}

/* IDMap::acquireID, id_map_impl.inl:69-184 */
int32_t oracle_idmap_acquire(oracle_idmap *m, uint32_t cache_idx, uint32_t *gen_out)
{
    cache_t *cache = &m->caches[cache_idx];

    if (cache->num_overflow > 0) {
        cache->num_overflow -= 1;
        return assign_cached(m, &cache->overflow_head, gen_out);
    }

    if (cache->num_free > 0) {
        cache->num_free -= 1;
        return assign_cached(m, &cache->free_head, gen_out);
    }

    if (m->global_head != SENTINEL) {
        int32_t free_ids = m->global_head;
        m->global_head = m->nodes[free_ids].global_next;
        m->nodes[free_ids].global_next = 1;
        cache->free_head = free_ids;
        cache->num_free = IDS_PER_CACHE - 1;
        return assign_cached(m, &cache->free_head, gen_out);
    }

    /* expand the store by one block */
    uint32_t block_start = m->num_ids;
    m->num_ids += IDS_PER_CACHE;
    if (m->num_ids > m->capacity) abort();

    m->nodes[block_start].gen = 0;
    node_t *next_free = &m->nodes[block_start + 1];
    next_free->sub_next = SENTINEL;
    next_free->global_next = IDS_PER_CACHE - 1;
    next_free->gen = 0;

    cache->free_head = (int32_t)block_start + 1;
    cache->num_free = IDS_PER_CACHE - 1;

    *gen_out = 0;
    return (int32_t)block_start;
}

/* IDMap::releaseID, id_map_impl.inl:186-224 */
void oracle_idmap_release(oracle_idmap *m, uint32_t cache_idx, int32_t id)
{
    cache_t *cache = &m->caches[cache_idx];
    node_t *node = &m->nodes[id];
    node->gen += 1;
    node->global_next = 1;

    if (cache->num_free < IDS_PER_CACHE) {
        node->sub_next = cache->free_head;
        cache->free_head = id;
        cache->num_free += 1;
        return;
    }

    if (cache->num_overflow < IDS_PER_CACHE) {
        node->sub_next = cache->overflow_head;
        cache->overflow_head = id;
        cache->num_overflow += 1;
    }

    if (cache->num_overflow == IDS_PER_CACHE) {
        m->nodes[cache->overflow_head].global_next = m->global_head;
        m->global_head = cache->overflow_head;
        cache->overflow_head = SENTINEL;
        cache->num_overflow = 0;
    }
}

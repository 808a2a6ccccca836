// TEST INFRASTRUCTURE ONLY.  Thin C handle over the REFERENCE's real entity id
// allocator (madrona::EntityStore = IDMap<Entity, Loc, LockedMapStore>,
// reference include/madrona/state.hpp + src/core/state.cpp:62-84), compiled
// from /root/reference into oracle/_ref/libidmap_ref.so.  Used to pin
// oracle/restate/id_map.c.  Also exposes rand:: for pinning threefry.c.
#include <madrona/state.hpp>
#include <madrona/rand.hpp>
#include <madrona/impl/id_map_impl.inl>

#include <vector>
#include <memory>

using namespace madrona;

struct RefIdMap {
    EntityStore store;
    std::vector<std::unique_ptr<EntityStore::Cache>> caches;
};

extern "C" {

__attribute__((visibility("default")))
RefIdMap *ref_idmap_create(uint32_t num_caches)
{
    RefIdMap *m = new RefIdMap {};
    for (uint32_t i = 0; i < num_caches; i++) {
        m->caches.emplace_back(new EntityStore::Cache());
    }
    return m;
}

__attribute__((visibility("default")))
void ref_idmap_destroy(RefIdMap *m) { delete m; }

__attribute__((visibility("default")))
int32_t ref_idmap_acquire(RefIdMap *m, uint32_t cache, uint32_t *gen_out)
{
    Entity e = m->store.newEntity(*m->caches[cache]);
    // a live slot must hold a Loc for freeEntity's bookkeeping to be valid
    m->store.setLoc(e, Loc { 0, 0 });
    *gen_out = e.gen;
    return e.id;
}

__attribute__((visibility("default")))
void ref_idmap_release(RefIdMap *m, uint32_t cache, int32_t id, uint32_t gen)
{
    m->store.freeEntity(*m->caches[cache], Entity { gen, id });
}

__attribute__((visibility("default")))
void ref_split_i(uint32_t a, uint32_t b, uint32_t idx, uint32_t idx_upper,
                 uint32_t *out_a, uint32_t *out_b)
{
    RandKey k = rand::split_i(RandKey { a, b }, idx, idx_upper);
    *out_a = k.a;
    *out_b = k.b;
}

__attribute__((visibility("default")))
int32_t ref_sample_i32(uint32_t a, uint32_t b, int32_t lo, int32_t hi)
{
    return rand::sampleI32(RandKey { a, b }, lo, hi);
}

__attribute__((visibility("default")))
int32_t ref_sample_i32_biased(uint32_t a, uint32_t b, int32_t lo, int32_t hi)
{
    return rand::sampleI32Biased(RandKey { a, b }, lo, hi);
}

__attribute__((visibility("default")))
float ref_sample_uniform(uint32_t a, uint32_t b)
{
    return rand::sampleUniform(RandKey { a, b });
}

}

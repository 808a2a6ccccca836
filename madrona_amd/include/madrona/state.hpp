// StateManager of the MI355X backend.
//
// API contract: reference src/mw/device/include/madrona/state.hpp:58-226 (the
// device-side StateManager every system and the physics module talk to) and
// include/madrona/registry.hpp (the registration surface).  One class serves
// both roles here:
//   * on the HOST it is a thin typed front end over the C ABI (mwhip.h):
//     registerComponent<T>() -> mwhip_register_component(...), etc.  It is
//     what ECSRegistry wraps while the simulator's registerTypes runs.
//   * on the DEVICE the same type is a view over the device-resident
//     mwhip::EcsState (tables, column lookup, entity store) with the
//     accessor names of the reference's device StateManager.
#pragma once

#include <madrona/ecs.hpp>
#include <madrona/ecs_flags.hpp>
#include <madrona/optional.hpp>
#include <madrona/query.hpp>
#include <madrona/type_tracker.hpp>
#include <madrona/utils.hpp>
#include <madrona/mwhip/ecs_state.hpp>

#include <mwhip.h>

#include <array>
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <utility>

namespace madrona {

namespace mwhip {

// Device-only ECS operations reached from host code (e.g. a world constructor
// called on the host): always a programming error in this backend.
inline void hostOnlyAbort(const char *what)
{
    fprintf(stderr, "madrona_amd: Context::%s is only available in device code\n",
            what);
    abort();
}

inline void check(int rc, const char *what)
{
    if (rc < 0) {
        fprintf(stderr, "madrona_amd: %s failed (%d): %s\n", what, rc,
                mwhip_last_error());
        abort();
    }
}

}

class StateManager : public mwhip::EcsState {
public:
    // ---- host: registration (reference device state.inl:7-158) -------------
    template <typename ComponentT>
    MADRONA_HOST_API ComponentID registerComponent(uint32_t num_bytes = 0);
    // host: which = 0 ray caster resolution, 1 RGBD flag, 2 max views per world
    MADRONA_HOST_API inline uint32_t renderConfig(int which) const;
    // host: the tables / components of the batch ray caster, by type id
    // (archetypes: renderable, camera, light, output; components: instance,
    // Morton code, TLBVH leaf, camera, light, RGB, depth)
    MADRONA_HOST_API inline void setRenderLayout(const uint32_t (&archetypes)[4],
                                                 const uint32_t (&components)[7]);

    template <typename ArchetypeT, typename... MetadataComponentTs>
    MADRONA_HOST_API ArchetypeID registerArchetype(
        ComponentMetadataSelector<MetadataComponentTs...> component_metadatas,
        ArchetypeFlags archetype_flags,
        CountT max_num_entities_per_world);

    template <typename SingletonT>
    MADRONA_HOST_API void registerSingleton();

    template <typename BundleT>
    MADRONA_HOST_API void registerBundle();

    template <typename AliasT, typename BundleT>
    MADRONA_HOST_API void registerBundleAlias();

    template <typename ArchetypeT, typename ComponentT>
    MADRONA_HOST_API ComponentT *exportColumn(int32_t slot);

    template <typename SingletonT>
    MADRONA_HOST_API SingletonT *exportSingleton(int32_t slot);

    // host: mwhip_make_query; (device-side query creation is not needed by
    // the hot path: every ParallelFor node's query is built with the graph)
    template <typename... ComponentTs>
    MADRONA_HOST_API Query<ComponentTs...> query();

    // ---- device: accessors (reference device state.inl:160-530) -------------
    // Declared host+device because simulator code that uses them is compiled
    // for both sides; the host copies of the mutating ones abort.
    MADRONA_HD inline Loc getLoc(Entity e) const;

    template <typename ComponentT>
    MADRONA_HD inline ComponentT &getUnsafe(Entity e);
    template <typename ComponentT>
    MADRONA_HD inline ComponentT &getUnsafe(Loc loc);

    template <typename ComponentT>
    MADRONA_HD inline ResultRef<ComponentT> get(Entity e);
    template <typename ComponentT>
    MADRONA_HD inline ResultRef<ComponentT> get(Loc loc);

    template <typename ComponentT>
    MADRONA_HD inline ComponentT &getDirect(int32_t column_idx, Loc loc);

    template <typename SingletonT>
    MADRONA_HD inline SingletonT &getSingleton(WorldID world_id);
    template <typename SingletonT>
    MADRONA_HD inline SingletonT *getSingletonColumn();

    // exclusive: the caller is the only thread acting for this world in the
    // running node (per-world systems, world constructors) -- skips the
    // per-world id-cache lock, whose agent-scope acquire/release costs an L2
    // write-back + L1 invalidate (~3 us) per call on MI355X
    MADRONA_HD Entity makeEntityNow(WorldID world_id, uint32_t archetype_id,
                                    bool exclusive = false,
                                    Loc *loc_out = nullptr);
    MADRONA_HD Loc makeTemporary(WorldID world_id, uint32_t archetype_id);
    MADRONA_HD void destroyEntityNow(WorldID caller_world, Entity e,
                                     bool exclusive = false);

    // ---- wave-cooperative creation / destruction (this backend, device only;
    // CustomParallelForNode<..., 64, 1> systems) ------------------------------
    // All 64 lanes of the invocation's wavefront call together.  The effect is
    // that of the lanes with `want` set calling makeEntityNow / destroyEntityNow
    // one after the other in lane order -- same entity ids and generations (ids
    // are popped from / pushed onto the world's id cache in that order), rows in
    // lane order -- at the latency of one such call: rows come from one atomic,
    // a single lane walks the id free list, everything else is per lane.
    // destroy: the wanted entities must be distinct.
    MADRONA_HD Entity makeEntityOrdered(WorldID world_id, uint32_t archetype_id,
                                        bool want, bool exclusive, Loc *loc_out);
    MADRONA_HD void destroyEntityOrdered(WorldID caller_world, Entity e, bool want,
                                         bool exclusive);
    MADRONA_HD void clearTemporaries(uint32_t archetype_id);

    template <typename ArchetypeT, typename ComponentT>
    MADRONA_HD inline ComponentT *getArchetypeComponent();
    MADRONA_HD inline void *getArchetypeComponent(uint32_t archetype_id,
                                                      uint32_t component_id);
    MADRONA_HD inline int32_t getArchetypeColumnIndex(uint32_t archetype_id,
                                                          uint32_t component_id);
    MADRONA_HD inline void *getArchetypeColumn(uint32_t archetype_id,
                                                   int32_t column_idx);

    template <typename ArchetypeT, typename ComponentT>
    MADRONA_HD inline std::pair<ComponentT *, uint32_t>
    getWorldComponentsAndCount(uint32_t world_id);
    template <typename ArchetypeT>
    MADRONA_HD inline Entity *getWorldEntities(uint32_t world_id);

    template <typename ArchetypeT>
    MADRONA_HD inline int32_t *getArchetypeWorldOffsets();
    MADRONA_HD inline int32_t *getArchetypeWorldOffsets(uint32_t archetype_id);
    template <typename ArchetypeT>
    MADRONA_HD inline int32_t *getArchetypeWorldCounts();
    MADRONA_HD inline int32_t *getArchetypeWorldCounts(uint32_t archetype_id);

    template <typename ArchetypeT>
    MADRONA_HD inline uint32_t getArchetypeNumRows();
    MADRONA_HD inline int32_t numArchetypeRows(uint32_t archetype_id) const;
    MADRONA_HD inline int32_t getArchetypeNumColumns(uint32_t archetype_id);
    MADRONA_HD inline uint32_t getArchetypeColumnBytesPerRow(
        uint32_t archetype_id, int32_t column_idx);

    MADRONA_HD inline bool archetypeNeedsSort(uint32_t archetype_id) const;
    MADRONA_HD inline void archetypeSetNeedsSort(uint32_t archetype_id);

    // Walks the world's row range of every archetype matched by the query;
    // requires the tables to be sorted by world (reference state.inl:214-252).
    template <int32_t num_components, typename Fn>
    MADRONA_HD inline void iterateQuery(uint32_t world_id,
                                            const QueryRef *query_ref, Fn &&fn);

    MADRONA_HD inline void *tmpAlloc(uint64_t num_bytes);

    template <typename SingletonT>
    struct SingletonArchetype : public madrona::Archetype<SingletonT> {};

    static constexpr uint32_t bundle_typeid_mask_ = mwhip::kBundleMask;
    static constexpr uint32_t user_component_offset_ = 2;
    static constexpr uint32_t max_archetype_components_ = mwhip::kMaxColumns - 2;

private:
    inline mwhip_exec *exec() const { return (mwhip_exec *)hostExec; }

    static inline uint32_t next_component_id_ = 0;
    static inline uint32_t next_archetype_id_ = 0;
    static inline uint32_t next_bundle_id_ = mwhip::kBundleMask;
};

}

#include "state.inl"

#pragma once

namespace madrona {

// ===========================================================================
// Host: registration -> C ABI
// ===========================================================================

template <typename ComponentT>
MADRONA_HOST_API ComponentID StateManager::registerComponent(uint32_t num_bytes)
{
    TypeTracker::touchDeviceSymbol<ComponentT>();
#if MADRONA_ON_HOST
    TypeTracker::registerType<ComponentT>(&next_component_id_);
    uint32_t id = TypeTracker::typeID<ComponentT>();

    uint32_t size = num_bytes == 0 ? (uint32_t)sizeof(ComponentT) : num_bytes;
    mwhip::check(mwhip_register_component(exec(), id,
        (uint32_t)alignof(ComponentT), size), "registerComponent");

    return ComponentID { id };
#else
    MADRONA_DEVICE_STUB();
#endif
}

namespace mwhip {

template <typename ComponentT, typename... FlagComponentTs>
inline ComponentFlags lookupComponentFlags(
    const ComponentMetadataSelector<FlagComponentTs...> &sel)
{
    ComponentFlags result = ComponentFlags::None;
    if constexpr (sizeof...(FlagComponentTs) > 0) {
        const bool matches[] = { std::is_same_v<ComponentT, FlagComponentTs>... };
        for (size_t i = 0; i < sizeof...(FlagComponentTs); i++) {
            if (matches[i]) {
                result = sel.flags[i];
                break;
            }
        }
    }
    return result;
}

template <typename> struct ArchetypeUnpack;
template <typename... Cs>
struct ArchetypeUnpack<Archetype<Cs...>> {
    static constexpr uint32_t count = sizeof...(Cs);

    static void ids(uint32_t *out)
    {
        uint32_t tmp[] = { TypeTracker::typeID<Cs>()..., 0u };
        for (uint32_t i = 0; i < count; i++) out[i] = tmp[i];
    }

    template <typename... Ms>
    static void flags(const ComponentMetadataSelector<Ms...> &sel, uint32_t *out)
    {
        uint32_t tmp[] = { (uint32_t)lookupComponentFlags<Cs>(sel)..., 0u };
        for (uint32_t i = 0; i < count; i++) out[i] = tmp[i];
    }
};

template <typename> struct BundleUnpack;
template <typename... Cs>
struct BundleUnpack<Bundle<Cs...>> {
    static constexpr uint32_t count = sizeof...(Cs);

    static void ids(uint32_t *out)
    {
        uint32_t tmp[] = { TypeTracker::typeID<Cs>()..., 0u };
        for (uint32_t i = 0; i < count; i++) out[i] = tmp[i];
    }
};

}

template <typename ArchetypeT, typename... MetadataComponentTs>
MADRONA_HOST_API ArchetypeID StateManager::registerArchetype(
    ComponentMetadataSelector<MetadataComponentTs...> component_metadatas,
    ArchetypeFlags archetype_flags,
    CountT max_num_entities_per_world)
{
    TypeTracker::touchDeviceSymbol<ArchetypeT>();
#if MADRONA_ON_HOST
    TypeTracker::registerType<ArchetypeT>(&next_archetype_id_);
    uint32_t id = TypeTracker::typeID<ArchetypeT>();

    using Unpack = mwhip::ArchetypeUnpack<typename ArchetypeT::Base>;
    uint32_t component_ids[Unpack::count + 1];
    uint32_t component_flags[Unpack::count + 1];
    Unpack::ids(component_ids);
    Unpack::flags(component_metadatas, component_flags);

    for (uint32_t i = 0; i < Unpack::count; i++) {
        if (component_ids[i] == TypeTracker::unassignedTypeID) {
            fprintf(stderr, "madrona_amd: archetype %u uses an unregistered "
                    "component (position %u)\n", id, i);
            abort();
        }
    }

    mwhip::check(mwhip_register_archetype(exec(), id, component_ids,
        component_flags, Unpack::count, (uint32_t)archetype_flags,
        (uint32_t)max_num_entities_per_world), "registerArchetype");

    return ArchetypeID { id };
#else
    MADRONA_DEVICE_STUB();
#endif
}

MADRONA_HOST_API inline uint32_t StateManager::renderConfig(int which) const
{
#if MADRONA_ON_HOST
    uint32_t res = 0, rgbd = 0;
    mwhip_render_config((const mwhip_exec *)hostExec, &res, &rgbd);
    if (which == 2) {
        return mwhip_render_max_views((const mwhip_exec *)hostExec);
    }
    return which == 0 ? res : rgbd;
#else
    (void)which;
    MADRONA_DEVICE_STUB();
#endif
}

MADRONA_HOST_API inline void StateManager::setRenderLayout(
    const uint32_t (&archetypes)[4], const uint32_t (&components)[7])
{
#if MADRONA_ON_HOST
    mwhip_render_layout layout {};
    layout.renderable_archetype = archetypes[0];
    layout.camera_archetype = archetypes[1];
    layout.light_archetype = archetypes[2];
    layout.output_archetype = archetypes[3];
    layout.instance_component = components[0];
    layout.morton_component = components[1];
    layout.tlbvh_component = components[2];
    layout.camera_component = components[3];
    layout.light_component = components[4];
    layout.rgb_component = components[5];
    layout.depth_component = components[6];
    mwhip::check(mwhip_set_render_layout(exec(), &layout), "setRenderLayout");
#else
    (void)archetypes;
    (void)components;
    MADRONA_DEVICE_STUB();
#endif
}

template <typename SingletonT>
MADRONA_HOST_API void StateManager::registerSingleton()
{
    TypeTracker::touchDeviceSymbol<SingletonT>();
    TypeTracker::touchDeviceSymbol<SingletonArchetype<SingletonT>>();
#if MADRONA_ON_HOST
    using ArchetypeT = SingletonArchetype<SingletonT>;

    registerComponent<SingletonT>();
    registerArchetype<ArchetypeT>(
        ComponentMetadataSelector<> {},
        (ArchetypeFlags)MWHIP_ARCHETYPE_SINGLETON, 1);

    mwhip::check(mwhip_register_singleton(exec(),
        TypeTracker::typeID<ArchetypeT>(),
        TypeTracker::typeID<SingletonT>()), "registerSingleton");
#else
    MADRONA_DEVICE_STUB();
#endif
}

template <typename BundleT>
MADRONA_HOST_API void StateManager::registerBundle()
{
    TypeTracker::touchDeviceSymbol<BundleT>();
#if MADRONA_ON_HOST
    TypeTracker::registerType<BundleT>(&next_bundle_id_);
    uint32_t id = TypeTracker::typeID<BundleT>();

    using Unpack = mwhip::BundleUnpack<typename BundleT::Base>;
    uint32_t component_ids[Unpack::count + 1];
    Unpack::ids(component_ids);

    mwhip::check(mwhip_register_bundle(exec(), id, component_ids,
        Unpack::count), "registerBundle");
#else
    MADRONA_DEVICE_STUB();
#endif
}

template <typename AliasT, typename BundleT>
MADRONA_HOST_API void StateManager::registerBundleAlias()
{
    TypeTracker::touchDeviceSymbol<AliasT>();
#if MADRONA_ON_HOST
    uint32_t bundle_id = TypeTracker::typeID<BundleT>();
    assert(bundle_id != TypeTracker::unassignedTypeID);

    // the alias shares the bundle's id (reference device state.inl:124-131)
    TypeTracker::registerType<AliasT>(&bundle_id);
#else
    MADRONA_DEVICE_STUB();
#endif
}

template <typename ArchetypeT, typename ComponentT>
MADRONA_HOST_API ComponentT *StateManager::exportColumn(int32_t slot)
{
#if MADRONA_ON_HOST
    return (ComponentT *)mwhip_export_column(exec(),
        TypeTracker::typeID<ArchetypeT>(), TypeTracker::typeID<ComponentT>(),
        slot);
#else
    MADRONA_DEVICE_STUB();
#endif
}

template <typename SingletonT>
MADRONA_HOST_API SingletonT *StateManager::exportSingleton(int32_t slot)
{
#if MADRONA_ON_HOST
    return exportColumn<SingletonArchetype<SingletonT>, SingletonT>(slot);
#else
    MADRONA_DEVICE_STUB();
#endif
}

template <typename... ComponentTs>
MADRONA_HOST_API Query<ComponentTs...> StateManager::query()
{
#if MADRONA_ON_HOST
    uint32_t component_ids[] = {
        TypeTracker::typeID<std::remove_const_t<ComponentTs>>()...
    };

    QueryRef ref {};
    mwhip::check(mwhip_make_query(exec(), component_ids,
        (uint32_t)sizeof...(ComponentTs), &ref.offset,
        &ref.numMatchingArchetypes, &ref.flags), "makeQuery");
    ref.numComponents = (uint32_t)sizeof...(ComponentTs);
    ref.numReferences = 1;

    return Query<ComponentTs...>(ref);
#else
    MADRONA_DEVICE_STUB();
#endif
}

// ===========================================================================
// Device: accessors
// ===========================================================================

MADRONA_HD Loc StateManager::getLoc(Entity e) const
{
    if (e.id < 0) {
        return Loc::none();
    }

    const mwhip::EntitySlot &slot = mwhip::entitiesOf(this)[e.id];
    if (slot.gen != e.gen) {
        return Loc::none();
    }

    return Loc { slot.loc.archetype, slot.loc.row };
}

template <typename ComponentT>
MADRONA_HD ComponentT &StateManager::getUnsafe(Loc loc)
{
    uint32_t component_id = TypeTracker::typeID<ComponentT>();
#if defined(__HIP_DEVICE_COMPILE__)
    void *const *col_ptrs = mwhip::loadInvariant(&colPtr);
    uint32_t num_slots = mwhip::loadInvariant(&numComponentSlots);
    return ((ComponentT *)mwhip::loadInvariant(
        &col_ptrs[loc.archetype * num_slots + component_id]))[loc.row];
#else
    return ((ComponentT *)colPtr[loc.archetype * numComponentSlots +
                                 component_id])[loc.row];
#endif
}

template <typename ComponentT>
MADRONA_HD ComponentT &StateManager::getUnsafe(Entity e)
{
    const mwhip::EntitySlot &slot = mwhip::entitiesOf(this)[e.id];
    return getUnsafe<ComponentT>(Loc { slot.loc.archetype, slot.loc.row });
}

template <typename ComponentT>
MADRONA_HD ResultRef<ComponentT> StateManager::get(Loc loc)
{
    uint32_t component_id = TypeTracker::typeID<ComponentT>();
#if defined(__HIP_DEVICE_COMPILE__)
    void *const *col_ptrs = mwhip::loadInvariant(&colPtr);
    uint32_t num_slots = mwhip::loadInvariant(&numComponentSlots);
    ComponentT *col = (ComponentT *)mwhip::loadInvariant(
        &col_ptrs[loc.archetype * num_slots + component_id]);
#else
    ComponentT *col = (ComponentT *)colPtr[
        loc.archetype * numComponentSlots + component_id];
#endif
    if (col == nullptr) {
        return ResultRef<ComponentT>(nullptr);
    }

    return ResultRef<ComponentT>(col + loc.row);
}

template <typename ComponentT>
MADRONA_HD ResultRef<ComponentT> StateManager::get(Entity e)
{
    Loc loc = getLoc(e);
    if (!loc.valid()) {
        return ResultRef<ComponentT>(nullptr);
    }

    return get<ComponentT>(loc);
}

template <typename ComponentT>
MADRONA_HD ComponentT &StateManager::getDirect(int32_t column_idx, Loc loc)
{
    return ((ComponentT *)mwhip::columnOf(
        mwhip::tablesOf(this)[loc.archetype], column_idx))[loc.row];
}

template <typename SingletonT>
MADRONA_HD SingletonT *StateManager::getSingletonColumn()
{
    // a singleton archetype has exactly one user column, column 2
    uint32_t archetype_id =
        TypeTracker::typeID<SingletonArchetype<SingletonT>>();
    return (SingletonT *)mwhip::columnOf(
        mwhip::tablesOf(this)[archetype_id], user_component_offset_);
}

template <typename SingletonT>
MADRONA_HD SingletonT &StateManager::getSingleton(WorldID world_id)
{
    return getSingletonColumn<SingletonT>()[world_id.idx];
}

MADRONA_HD inline Entity StateManager::makeEntityNow(WorldID world_id, uint32_t archetype_id, bool exclusive, Loc *loc_out)
{
#if defined(__HIP_DEVICE_COMPILE__)
    mwhip::TableHdr &tbl = mwhip::tablesOf(this)[archetype_id];
    // the row counter's atomic and the id cache reads are independent: issue
    // the atomic, take the id, then look at the row
    int32_t row = mwhip::appendRowIssue(tbl);

    uint32_t gen = 0;
    int32_t id = 0;
    if (exclusive) {
        // one thread per world in this node: nobody else touches the cache
        id = mwhip::acquireIdLocked(this, world_id.idx,
                                    mwhip::worldCachesOf(this)[world_id.idx],
                                    &gen);
    } else {
        mwhip::withWorldCache(this, world_id.idx, [&](mwhip::IdCache &cache) {
            id = mwhip::acquireIdLocked(this, world_id.idx, cache, &gen);
        });
    }

    row = mwhip::appendRowCheck(this, tbl, row);

    mwhip::EntitySlot &slot = mwhip::entitiesOf(this)[id];
    slot.loc.archetype = archetype_id;
    slot.loc.row = row;

    Entity e { gen, id };
    ((Entity *)mwhip::columnOf(tbl, 0))[row] = e;
    ((WorldID *)mwhip::columnOf(tbl, 1))[row] = world_id;

    if (loc_out != nullptr) {
        *loc_out = Loc { archetype_id, row };
    }
    return e;
#else
    (void)0;
    mwhip::hostOnlyAbort("makeEntity"); return Entity::none();
#endif
}

MADRONA_HD inline Loc StateManager::makeTemporary(WorldID world_id, uint32_t archetype_id)
{
#if defined(__HIP_DEVICE_COMPILE__)
    mwhip::TableHdr &tbl = tables[archetype_id];
    int32_t row = mwhip::appendRow(this, tbl);

    // sentinel so the sort node does not try to remap an entity slot
    ((Entity *)tbl.columns[0])[row] = Entity::none();
    ((WorldID *)tbl.columns[1])[row] = world_id;

    return Loc { archetype_id, row };
#else
    (void)0;
    mwhip::hostOnlyAbort("makeTemporary"); return Loc::none();
#endif
}

// Mirrors the CPU backend (src/core/state.cpp:188-203): stale handles are
// ignored, the row keeps its slot until the next compaction, the id goes back
// to the *calling* world's cache.  The row is tagged the way both backends'
// ParallelFor skip tests expect: Entity::none() (CPU state.inl:484) and
// WorldID -1 (GPU taskgraph.inl:208), which is also the sort key that drops it.
MADRONA_HD inline void StateManager::destroyEntityNow(WorldID caller_world, Entity e, bool exclusive)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (e.id < 0) {
        return;
    }

    // generation and location in one round trip
    const mwhip::EntitySlot &slot = mwhip::entitiesOf(this)[e.id];
    const uint32_t slot_gen = slot.gen;
    Loc loc { slot.loc.archetype, slot.loc.row };
    if (slot_gen != e.gen) {
        return;
    }

    mwhip::TableHdr &tbl = mwhip::tablesOf(this)[loc.archetype];
    ((Entity *)mwhip::columnOf(tbl, 0))[loc.row] = Entity::none();
    ((WorldID *)mwhip::columnOf(tbl, 1))[loc.row] = WorldID { -1 };
    tbl.needsSort = 1u;

    if (exclusive) {
        mwhip::releaseIdLocked(this,
            mwhip::worldCachesOf(this)[caller_world.idx], e.id, slot_gen);
    } else {
        mwhip::withWorldCache(this, caller_world.idx,
                              [&](mwhip::IdCache &cache) {
            mwhip::releaseIdLocked(this, cache, e.id, slot_gen);
        });
    }
#else
    (void)0;
    mwhip::hostOnlyAbort("destroyEntity");
#endif
}

MADRONA_HD inline Entity StateManager::makeEntityOrdered(
    WorldID world_id, uint32_t archetype_id, bool want, bool exclusive,
    Loc *loc_out)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ int32_t ordered_ids[64];
    __shared__ uint32_t ordered_gens[64];

    const uint64_t mask = __builtin_amdgcn_ballot_w64(want);
    if (mask == 0) {
        return Entity::none();
    }
    const uint32_t lane =
        __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const uint32_t count = (uint32_t)__builtin_popcountll(mask);
    const uint32_t rank =
        (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
    const uint32_t leader = (uint32_t)__builtin_ctzll(mask);

    if (lane == leader) {
        // the ids the sequential calls would have taken, in their order; the
        // cache lives in registers while the list is walked (one dependent
        // read per id: the popped slot names the next head)
        auto take = [&](mwhip::IdCache &cache) {
            mwhip::IdCache local = cache;
            for (uint32_t k = 0; k < count; k++) {
                uint32_t gen = 0;
                ordered_ids[k] =
                    mwhip::acquireIdLocked(this, world_id.idx, local, &gen);
                ordered_gens[k] = gen;
            }
            local.lock = cache.lock;
            cache = local;
        };
        if (exclusive) {
            take(mwhip::worldCachesOf(this)[world_id.idx]);
        } else {
            mwhip::withWorldCache(this, world_id.idx, take);
        }
    }

    // rows: one atomic per archetype among the wanted lanes (they may ask for
    // different ones), a table's rows in lane order
    int32_t my_row = 0;
    for (uint64_t remaining = mask; remaining != 0; ) {
        const uint32_t first_lane = (uint32_t)__builtin_ctzll(remaining);
        const uint32_t arch = (uint32_t)__shfl((int)archetype_id, (int)first_lane, 64);
        const uint64_t same =
            __builtin_amdgcn_ballot_w64(want && archetype_id == arch);
        int32_t first_row = 0;
        if (lane == first_lane) {
            mwhip::TableHdr &shared_tbl = mwhip::tablesOf(this)[arch];
            mwhip::markRowAppender();
            shared_tbl.needsSort = 1u;
            first_row = mwhip::atomicAddI32(&shared_tbl.numRows,
                                            (int32_t)__builtin_popcountll(same));
        }
        first_row = __shfl(first_row, (int)first_lane, 64);
        if (want && archetype_id == arch) {
            my_row = first_row +
                (int32_t)__builtin_popcountll(same & ((1ull << lane) - 1ull));
        }
        remaining &= ~same;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    Entity e = Entity::none();
    if (want) {
        mwhip::TableHdr &tbl = mwhip::tablesOf(this)[archetype_id];
        const int32_t row = mwhip::appendRowCheck(this, tbl, my_row);
        const int32_t id = ordered_ids[rank];
        const uint32_t gen = ordered_gens[rank];

        mwhip::EntitySlot &slot = mwhip::entitiesOf(this)[id];
        slot.loc.archetype = archetype_id;
        slot.loc.row = row;

        e = Entity { gen, id };
        ((Entity *)mwhip::columnOf(tbl, 0))[row] = e;
        ((WorldID *)mwhip::columnOf(tbl, 1))[row] = world_id;
        if (loc_out != nullptr) {
            *loc_out = Loc { archetype_id, row };
        }
    }
    // (the next ordered call reuses the two arrays)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return e;
#else
    (void)world_id; (void)archetype_id; (void)want; (void)exclusive; (void)loc_out;
    mwhip::hostOnlyAbort("makeEntityOrdered"); return Entity::none();
#endif
}

MADRONA_HD inline void StateManager::destroyEntityOrdered(
    WorldID caller_world, Entity e, bool want, bool exclusive)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ int32_t ordered_ids[64];
    __shared__ uint32_t ordered_gens[64];

    // every lane looks at its own entity (stale handles are ignored, as in
    // destroyEntityNow) and tags its row -- before any id goes back: a free
    // slot's links overwrite the location
    bool valid = false;
    uint32_t slot_gen = 0;
    if (want && e.id >= 0) {
        const mwhip::EntitySlot &slot = mwhip::entitiesOf(this)[e.id];
        slot_gen = slot.gen;
        const Loc loc { slot.loc.archetype, slot.loc.row };
        if (slot_gen == e.gen) {
            valid = true;
            mwhip::TableHdr &tbl = mwhip::tablesOf(this)[loc.archetype];
            ((Entity *)mwhip::columnOf(tbl, 0))[loc.row] = Entity::none();
            ((WorldID *)mwhip::columnOf(tbl, 1))[loc.row] = WorldID { -1 };
            tbl.needsSort = 1u;
        }
    }

    const uint64_t mask = __builtin_amdgcn_ballot_w64(valid);
    if (mask == 0) {
        return;
    }
    const uint32_t lane =
        __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const uint32_t count = (uint32_t)__builtin_popcountll(mask);
    const uint32_t rank =
        (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
    if (valid) {
        ordered_ids[rank] = e.id;
        ordered_gens[rank] = slot_gen;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    if (lane == (uint32_t)__builtin_ctzll(mask)) {
        // back onto the world's free list in lane order (stores only)
        auto give = [&](mwhip::IdCache &cache) {
            mwhip::IdCache local = cache;
            for (uint32_t k = 0; k < count; k++) {
                mwhip::releaseIdLocked(this, local, ordered_ids[k], ordered_gens[k]);
            }
            local.lock = cache.lock;
            cache = local;
        };
        if (exclusive) {
            give(mwhip::worldCachesOf(this)[caller_world.idx]);
        } else {
            mwhip::withWorldCache(this, caller_world.idx, give);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#else
    (void)caller_world; (void)e; (void)want; (void)exclusive;
    mwhip::hostOnlyAbort("destroyEntityOrdered");
#endif
}

MADRONA_HD inline void StateManager::clearTemporaries(uint32_t archetype_id)
{
    mwhip::TableHdr &tbl = tables[archetype_id];
    if (tbl.numRows != 0) {
        tbl.needsSort = 1u;
    }
    tbl.numRows = 0;
    tbl.sortedRows = 0;
}

MADRONA_HD int32_t StateManager::getArchetypeColumnIndex(uint32_t archetype_id,
                                              uint32_t component_id)
{
    return (int32_t)colLookup[archetype_id * numComponentSlots + component_id];
}

MADRONA_HD void *StateManager::getArchetypeColumn(uint32_t archetype_id, int32_t column_idx)
{
    return tables[archetype_id].columns[column_idx];
}

MADRONA_HD void *StateManager::getArchetypeComponent(uint32_t archetype_id,
                                          uint32_t component_id)
{
    return getArchetypeColumn(archetype_id,
        getArchetypeColumnIndex(archetype_id, component_id));
}

template <typename ArchetypeT, typename ComponentT>
MADRONA_HD ComponentT *StateManager::getArchetypeComponent()
{
    return (ComponentT *)getArchetypeComponent(
        TypeTracker::typeID<ArchetypeT>(), TypeTracker::typeID<ComponentT>());
}

MADRONA_HD int32_t *StateManager::getArchetypeWorldOffsets(uint32_t archetype_id)
{
    return tables[archetype_id].worldOffsets;
}

MADRONA_HD int32_t *StateManager::getArchetypeWorldCounts(uint32_t archetype_id)
{
    return tables[archetype_id].worldCounts;
}

template <typename ArchetypeT>
MADRONA_HD int32_t *StateManager::getArchetypeWorldOffsets()
{
    return getArchetypeWorldOffsets(TypeTracker::typeID<ArchetypeT>());
}

template <typename ArchetypeT>
MADRONA_HD int32_t *StateManager::getArchetypeWorldCounts()
{
    return getArchetypeWorldCounts(TypeTracker::typeID<ArchetypeT>());
}

template <typename ArchetypeT, typename ComponentT>
MADRONA_HD std::pair<ComponentT *, uint32_t> StateManager::getWorldComponentsAndCount(
    uint32_t world_id)
{
    uint32_t archetype_id = TypeTracker::typeID<ArchetypeT>();
    ComponentT *col = getArchetypeComponent<ArchetypeT, ComponentT>();
    mwhip::TableHdr &tbl = tables[archetype_id];
    return { col + tbl.worldOffsets[world_id],
             (uint32_t)tbl.worldCounts[world_id] };
}

template <typename ArchetypeT>
MADRONA_HD Entity *StateManager::getWorldEntities(uint32_t world_id)
{
    mwhip::TableHdr &tbl = tables[TypeTracker::typeID<ArchetypeT>()];
    return (Entity *)tbl.columns[0] + tbl.worldOffsets[world_id];
}

template <typename ArchetypeT>
MADRONA_HD uint32_t StateManager::getArchetypeNumRows()
{
    return (uint32_t)tables[TypeTracker::typeID<ArchetypeT>()].numRows;
}

MADRONA_HD int32_t StateManager::numArchetypeRows(uint32_t archetype_id) const
{
    return tables[archetype_id].numRows;
}

MADRONA_HD int32_t StateManager::getArchetypeNumColumns(uint32_t archetype_id)
{
    return tables[archetype_id].numColumns;
}

MADRONA_HD uint32_t StateManager::getArchetypeColumnBytesPerRow(uint32_t archetype_id,
                                                     int32_t column_idx)
{
    return tables[archetype_id].columnBytes[column_idx];
}

MADRONA_HD bool StateManager::archetypeNeedsSort(uint32_t archetype_id) const
{
    return tables[archetype_id].needsSort != 0;
}

MADRONA_HD void StateManager::archetypeSetNeedsSort(uint32_t archetype_id)
{
    // A caller that asks for a re-sort may have rewritten the WorldID keys of
    // rows that are already in the table: nothing of it counts as "still in
    // world order" any more, so the next world sort takes the radix chain
    // (the compaction chain trusts rows [0, sortedRows)).
    tables[archetype_id].needsSort = 1u;
    tables[archetype_id].sortedRows = 0;
}

namespace mwhip {

template <typename Fn, int32_t... Indices>
MADRONA_HD inline void iterateQueryImpl(
    EcsState *S, int32_t world_id, const QueryRef *query_ref, Fn &&fn,
    std::integer_sequence<int32_t, Indices...>)
{
    const uint32_t *query_values = &S->queryData[query_ref->offset];
    const int32_t num_archetypes = (int32_t)query_ref->numMatchingArchetypes;

    for (int32_t i = 0; i < num_archetypes; i++) {
        uint32_t archetype_idx = query_values[0];
        query_values += 1;

        TableHdr &tbl = S->tables[archetype_idx];
        int32_t world_offset = tbl.worldOffsets[world_id];
        int32_t world_count = tbl.worldCounts[world_id];

        for (int32_t r = 0; r < world_count; r++) {
            fn(world_offset + r, tbl.columns[query_values[Indices]]...);
        }

        query_values += sizeof...(Indices);
    }
}

}

template <int32_t num_components, typename Fn>
MADRONA_HD void StateManager::iterateQuery(uint32_t world_id, const QueryRef *query_ref,
                                Fn &&fn)
{
    mwhip::iterateQueryImpl(this, (int32_t)world_id, query_ref,
        std::forward<Fn>(fn),
        std::make_integer_sequence<int32_t, num_components>());
}

// 256-byte aligned bump allocation from the executor's scratch region
// (reference mwGPU::TmpAllocator, device/memory.cpp:123-178).
MADRONA_HD void *StateManager::tmpAlloc(uint64_t num_bytes)
{
#if defined(__HIP_DEVICE_COMPILE__)
    num_bytes = utils::roundUpPow2(num_bytes, (uint64_t)256);
    unsigned long long off = __hip_atomic_fetch_add(&tmpOffset,
        (unsigned long long)num_bytes, __ATOMIC_RELAXED,
        __HIP_MEMORY_SCOPE_AGENT);
    // (the region grows like a table: reserved address space, more of it mapped
    // on request)
    if (off + num_bytes > tmpCapacity &&
            !mwhip::waitForMailbox(this, mwhip::kGrowSlotTmp,
                (int32_t)((off + num_bytes + 1023ull) >> 10) - 1)) {
        mwhip::raiseError(this, mwhip::kErrTmpOverflow);
        return tmpBase;
    }
    return tmpBase + off;
#else
    (void)0;
    mwhip::hostOnlyAbort("tmpAlloc"); return nullptr;
#endif
}

}

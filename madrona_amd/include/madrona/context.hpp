// Context: the per-invocation handle every system receives.
// API contract: reference src/mw/device/include/madrona/context.hpp:17-82
// (the GPU-mode Context) -- same method names and semantics.  The object is
// built per row by the ParallelFor kernel and lives in registers: a pointer to
// the world's user data, a pointer to the device ECS state and the world id.
#pragma once

#include <madrona/fwd.hpp>
#include <madrona/ecs.hpp>
#include <madrona/state.hpp>
#include <madrona/registry.hpp>

namespace madrona {

struct WorkerInit {
    WorldID worldID;
    StateManager *stateMgr;
    // true when the running node has exactly one thread per world
    bool exclusiveWorld = false;
};

class Context {
public:
    MADRONA_HD inline Context(WorldBase *world_data, const WorkerInit &init)
        : data_(world_data), state_mgr_(init.stateMgr), world_id_(init.worldID),
          exclusive_world_(init.exclusiveWorld),
          cached_id_(-1), cached_gen_(0), cached_loc_ { 0, 0 }
    {}

    template <typename ArchetypeT>
    MADRONA_HD inline Entity makeEntity()
    {
        return makeEntity(TypeTracker::typeID<ArchetypeT>());
    }

    MADRONA_HD inline Entity makeEntity(uint32_t archetype_id)
    {
        Loc loc;
        Entity e = state_mgr_->makeEntityNow(world_id_, archetype_id,
                                             exclusive_world_, &loc);
        // the usual next step is a run of ctx.get<T>(e) = ...: remember where
        // the entity lives so those skip the entity-store round trip
        cached_id_ = e.id;
        cached_gen_ = e.gen;
        cached_loc_ = loc;
        return e;
    }

    template <typename ArchetypeT>
    MADRONA_HD inline Loc makeTemporary()
    {
        return makeTemporary(TypeTracker::typeID<ArchetypeT>());
    }

    MADRONA_HD inline Loc makeTemporary(uint32_t archetype_id)
    {
        return state_mgr_->makeTemporary(world_id_, archetype_id);
    }

    MADRONA_HD inline void destroyEntity(Entity e)
    {
        if (e.id == cached_id_) {
            cached_id_ = -1;
        }
        state_mgr_->destroyEntityNow(world_id_, e, exclusive_world_);
    }

    // ---- this backend, systems that run 64 lanes per invocation
    // (CustomParallelForNode<..., 64, 1, ...> under MADRONA_GPU_MODE): all 64
    // lanes call together; the effect -- entity ids, generations, row order --
    // is that of the lanes with `want` calling makeEntity / destroyEntity one
    // after the other in lane order (state.hpp) ----
    template <typename ArchetypeT>
    MADRONA_HD inline Entity makeEntityOrdered(bool want)
    {
        return makeEntityOrdered(TypeTracker::typeID<ArchetypeT>(), want);
    }

    // (lanes may ask for different archetypes: ids still go out in lane order)
    MADRONA_HD inline Entity makeEntityOrdered(uint32_t archetype_id, bool want)
    {
        Loc loc { 0, 0 };
        Entity e = state_mgr_->makeEntityOrdered(world_id_, archetype_id, want,
                                                 exclusive_world_, &loc);
        if (want) {
            cached_id_ = e.id;
            cached_gen_ = e.gen;
            cached_loc_ = loc;
        }
        return e;
    }

    MADRONA_HD inline void destroyEntityOrdered(Entity e, bool want)
    {
        if (want && e.id == cached_id_) {
            cached_id_ = -1;
        }
        state_mgr_->destroyEntityOrdered(world_id_, e, want, exclusive_world_);
    }

    MADRONA_HD inline Loc loc(Entity e) const
    {
        return state_mgr_->getLoc(e);
    }

    template <typename ComponentT>
    MADRONA_HD inline ComponentT &get(Entity e)
    {
        return state_mgr_->getUnsafe<ComponentT>(locCached(e));
    }

    template <typename ComponentT>
    MADRONA_HD inline ComponentT &get(Loc l)
    {
        return state_mgr_->getUnsafe<ComponentT>(l);
    }

    template <typename ComponentT>
    MADRONA_HD inline ResultRef<ComponentT> getSafe(Entity e)
    {
        return state_mgr_->get<ComponentT>(e);
    }

    template <typename ComponentT>
    MADRONA_HD inline ResultRef<ComponentT> getCheck(Entity e)
    {
        return state_mgr_->get<ComponentT>(e);
    }

    template <typename ComponentT>
    MADRONA_HD inline ResultRef<ComponentT> getCheck(Loc l)
    {
        return state_mgr_->get<ComponentT>(l);
    }

    template <typename ComponentT>
    MADRONA_HD inline ComponentT &getDirect(int32_t column_idx, Loc l)
    {
        return state_mgr_->getDirect<ComponentT>(column_idx, l);
    }

    template <typename SingletonT>
    MADRONA_HD inline SingletonT &singleton()
    {
        return state_mgr_->getSingleton<SingletonT>(world_id_);
    }

    MADRONA_HD inline void *tmpAlloc(uint64_t num_bytes)
    {
        return state_mgr_->tmpAlloc(num_bytes);
    }

    template <typename... ComponentTs, typename Fn>
    MADRONA_HD inline void iterateQuery(const Query<ComponentTs...> &query,
                                            Fn &&fn)
    {
        state_mgr_->iterateQuery<sizeof...(ComponentTs)>(
            (uint32_t)world_id_.idx, query.getSharedRef(),
            [&](int32_t row, auto... raw_ptrs) {
                fn(((ComponentTs *)raw_ptrs)[row]...);
            });
    }

    MADRONA_HD inline WorldID worldID() const { return world_id_; }

    MADRONA_HD inline WorldBase &data() const { return *data_; }

    MADRONA_HD inline StateManager *getStateManager() { return state_mgr_; }

protected:
    WorldBase *data_;

private:
    // One-entry (entity -> Loc) cache, held in registers.  Systems typically
    // touch several components of the same entity back to back; each hit
    // removes one dependent memory round trip (~0.3-0.5 us when the thread is
    // alone in its wavefront, as in per-world reset systems).  Rows only move
    // in the sort node, i.e. between kernels, so an entry cannot go stale
    // while a Context is alive except through destroyEntity (handled above).
    MADRONA_HD inline Loc locCached(Entity e)
    {
        if (e.id == cached_id_ && e.gen == cached_gen_) {
            return cached_loc_;
        }

        const mwhip::EntitySlot &slot =
            mwhip::entitiesOf(state_mgr_)[e.id];
        Loc loc { slot.loc.archetype, slot.loc.row };
        cached_id_ = e.id;
        cached_gen_ = e.gen;
        cached_loc_ = loc;
        return loc;
    }

    StateManager *state_mgr_;
    WorldID world_id_;
    bool exclusive_world_;
    int32_t cached_id_;
    uint32_t cached_gen_;
    Loc cached_loc_;
};

}

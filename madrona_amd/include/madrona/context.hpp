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
};

class Context {
public:
    MADRONA_HD inline Context(WorldBase *world_data, const WorkerInit &init)
        : data_(world_data), state_mgr_(init.stateMgr), world_id_(init.worldID)
    {}

    template <typename ArchetypeT>
    MADRONA_HD inline Entity makeEntity()
    {
        return makeEntity(TypeTracker::typeID<ArchetypeT>());
    }

    MADRONA_HD inline Entity makeEntity(uint32_t archetype_id)
    {
        return state_mgr_->makeEntityNow(world_id_, archetype_id);
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
        state_mgr_->destroyEntityNow(world_id_, e);
    }

    MADRONA_HD inline Loc loc(Entity e) const
    {
        return state_mgr_->getLoc(e);
    }

    template <typename ComponentT>
    MADRONA_HD inline ComponentT &get(Entity e)
    {
        return state_mgr_->getUnsafe<ComponentT>(e);
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
    StateManager *state_mgr_;
    WorldID world_id_;
};

}

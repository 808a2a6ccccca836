// Core ECS handle types.  API contract: reference include/madrona/ecs.hpp:17-75
// (Entity{gen,id}, Loc{archetype,row}, Archetype<>, Bundle<>, WorldID,
// WorldBase) -- field order and sizes are part of the exported-tensor ABI.
#pragma once

#include <madrona/fwd.hpp>
#include <madrona/macros.hpp>
#include <madrona/types.hpp>

namespace madrona {

// generation + slot of the entity store; column 0 of every archetype table
struct Entity {
    uint32_t gen;
    int32_t id;

    MADRONA_HD static constexpr inline Entity none() { return Entity { 0xFFFFFFFFu, -1 }; }
};

MADRONA_HD inline bool operator==(Entity a, Entity b) { return a.gen == b.gen && a.id == b.id; }
MADRONA_HD inline bool operator!=(Entity a, Entity b) { return !(a == b); }

// where an entity's row currently is
struct Loc {
    uint32_t archetype;
    int32_t row;

    MADRONA_HD static inline Loc none() { return Loc { 0xFFFFFFFFu, 0 }; }
    MADRONA_HD inline bool valid() const { return archetype != 0xFFFFFFFFu; }
};

MADRONA_HD inline bool operator==(Loc a, Loc b) { return a.row == b.row && a.archetype == b.archetype; }
MADRONA_HD inline bool operator!=(Loc a, Loc b) { return !(a == b); }

// column 1 of every archetype table; -1 tags a destroyed row
struct WorldID { int32_t idx; };

struct ComponentID { uint32_t id; };
struct ArchetypeID { uint32_t id; };
struct IndexHelper { uint32_t prev, next; };

// type lists: an archetype's columns, a reusable group of components
template <typename... ComponentTs>
struct Archetype { using Base = Archetype<ComponentTs...>; };

template <typename... ComponentTs>
struct Bundle { using Base = Bundle<ComponentTs...>; };

// Base class of the simulator's per-world state object.
class WorldBase {
public:
    MADRONA_HD inline WorldBase(Context &) {}
    WorldBase(const WorldBase &) = delete;
};

}

// Base transform components.  API contract: reference
// include/madrona/components.hpp:9-41 + src/core/base.cpp (registerTypes).
#pragma once

#include <madrona/math.hpp>
#include <madrona/fwd.hpp>
#include <madrona/taskgraph.hpp>

namespace madrona::base {

// thin wrappers: distinct component types over the math types
struct Position : math::Vector3 { MADRONA_HD Position(math::Vector3 v) : Vector3(v) {} };
struct Rotation : math::Quat { MADRONA_HD Rotation(math::Quat q) : Quat(q) {} };
struct Scale : math::Diag3x3 { MADRONA_HD Scale(math::Diag3x3 d) : Diag3x3(d) {} };

// index into the asset tables (ObjectManager, render meshes)
struct ObjectID { int32_t idx; };

struct ObjectInstance : Bundle<Position, Rotation, Scale, ObjectID> {};

namespace detail {
template <typename... Ts>
MADRONA_HOST_API inline void registerAll(ECSRegistry &registry)
{
    (registry.registerComponent<Ts>(), ...);
}
}

// component ids are handed out in this order, as in src/core/base.cpp
MADRONA_HOST_API inline void registerTypes(ECSRegistry &registry)
{
    detail::registerAll<Position, Rotation, Scale, ObjectID>(registry);
    registry.registerBundle<ObjectInstance>();
}

}

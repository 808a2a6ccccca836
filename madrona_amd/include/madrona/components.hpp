// Base transform components.  API contract: reference
// include/madrona/components.hpp:9-41 + src/core/base.cpp (registerTypes).
#pragma once

#include <madrona/math.hpp>
#include <madrona/fwd.hpp>
#include <madrona/taskgraph.hpp>

namespace madrona {
namespace base {

struct Position : math::Vector3 {
    MADRONA_HD Position(math::Vector3 v) : Vector3(v) {}
};

struct Rotation : math::Quat {
    MADRONA_HD Rotation(math::Quat q) : Quat(q) {}
};

struct Scale : math::Diag3x3 {
    MADRONA_HD Scale(math::Diag3x3 d) : Diag3x3(d) {}
};

struct ObjectID {
    int32_t idx;
};

struct ObjectInstance : Bundle<
    Position,
    Rotation,
    Scale,
    ObjectID
> {};

MADRONA_HOST_API inline void registerTypes(ECSRegistry &registry)
{
    registry.registerComponent<Position>();
    registry.registerComponent<Rotation>();
    registry.registerComponent<Scale>();
    registry.registerComponent<ObjectID>();

    registry.registerBundle<ObjectInstance>();
}

}
}

// PhysicsSystem with Solver::TGS (reference src/physics/tgs.cpp):
// bodies of three response types in free flight under gravity, random forces
// and torques -- what the reference's TGS path does to a world is its two
// integrators (its constraint side is a skeleton), so that is what this
// simulator pins: velocities with the body-space gyroscopic term for a box and
// an anisotropic slab, positions / rotations, the BVH kept up to date behind
// them.  Compiled unchanged against the reference CPU backend (oracle) and the
// HIP backend.
#pragma once

#include <madrona/taskgraph_builder.hpp>
#include <madrona/custom_context.hpp>
#include <madrona/components.hpp>
#include <madrona/math.hpp>
#include <madrona/rand.hpp>
#include <madrona/physics.hpp>

namespace tgsdrop {

using madrona::Entity;
using madrona::base::Position;
using madrona::base::Rotation;
using madrona::math::Vector3;

namespace consts {
inline constexpr int32_t numBodies = 12;
inline constexpr int32_t numAnchors = 2;
inline constexpr float deltaT = 0.04f;
inline constexpr int32_t numSubsteps = 4;
}

enum class ExportID : uint32_t { StepCount, NumExports };

// per body: the stream its kicks are drawn from
struct Kick {
    madrona::RandKey key;
    uint32_t step;
};
struct StepCount { int32_t n; };

struct Body : public madrona::Archetype<madrona::phys::RigidBody, Kick> {};
struct Anchor : public madrona::Archetype<madrona::phys::RigidBody> {};

class Engine;

struct Sim : public madrona::WorldBase {
    struct Config {
        uint32_t seed;
        uint32_t worldBase;
        madrona::phys::ObjectManager *rigidBodyObjMgr;
    };
    struct WorldInit {};

    static void registerTypes(madrona::ECSRegistry &registry, const Config &cfg);
    static void setupTasks(madrona::TaskGraphManager &taskgraph_mgr,
                           const Config &cfg);

    Sim(Engine &ctx, const Config &cfg, const WorldInit &init);

    Entity bodies[consts::numBodies];
    Entity anchors[consts::numAnchors];
};

class Engine : public madrona::CustomContext<Engine, Sim> {
public:
    using CustomContext::CustomContext;
};

}

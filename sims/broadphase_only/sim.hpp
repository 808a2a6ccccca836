// Standalone broadphase: bodies move kinematically, the BVH is kept up to date
// and the overlapping pairs are left in the CandidateTemporary table for the
// test to read (PhysicsSystem::setupStandaloneBroadphaseOverlapTasks /
// ...CleanupTasks -- the API gpu_hideseek-style simulators use without the
// solver).  Two rigid-body archetypes so that candidate order across
// archetypes is exercised; periodic BVH resets exercise the rebuild.
#pragma once

#include <madrona/taskgraph_builder.hpp>
#include <madrona/custom_context.hpp>
#include <madrona/components.hpp>
#include <madrona/math.hpp>
#include <madrona/rand.hpp>
#include <madrona/physics.hpp>

namespace bponly {

using madrona::Entity;
using madrona::base::Position;
using madrona::math::Vector3;

namespace consts {
inline constexpr int32_t numBoxes = 14;
inline constexpr int32_t numPillars = 4;
inline constexpr float arena = 5.f;
inline constexpr float deltaT = 0.05f;
inline constexpr int32_t rebuildPeriod = 16;
}

enum class ExportID : uint32_t { StepCount, NumExports };

struct Drift { Vector3 v; };
struct StepCount { int32_t n; };

struct Box : public madrona::Archetype<madrona::phys::RigidBody, Drift> {};
struct Pillar : public madrona::Archetype<madrona::phys::RigidBody> {};

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

    Entity boxes[consts::numBoxes];
    Entity pillars[consts::numPillars];
};

class Engine : public madrona::CustomContext<Engine, Sim> {
public:
    using CustomContext::CustomContext;
};

}

// Standalone broadphase: bodies move kinematically, the BVH is kept up to date
// and the overlapping pairs are left in the CandidateTemporary table for the
// test to read (PhysicsSystem::setupStandaloneBroadphaseOverlapTasks /
// ...CleanupTasks -- the API gpu_hideseek-style simulators use without the
// solver).  Two rigid-body archetypes so that candidate order across
// archetypes is exercised; periodic BVH resets exercise the rebuild.
//
// flags bit 0 ("ray mode"): the number of boxes differs per world (10 .. 100: trees
// of more than 64 leaves next to small ones) and every world holds 1 .. 3
// sensors that each cast a fan of 32 rays through the world's BVH
// (BVH::traceRay; on the MI355X backend the rays of a fan share their origin:
// BVH::traceRayShared, 32 lanes per sensor, so that the two halves of a
// wavefront regularly work on different trees with different leaf counts).
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
inline constexpr int32_t numBoxes = 14;          // (flags bit 0 clear)
inline constexpr int32_t maxBoxes = 100;
inline constexpr int32_t maxSensors = 3;
inline constexpr int32_t raysPerSensor = 32;
inline constexpr int32_t numPillars = 4;
inline constexpr float arena = 5.f;
inline constexpr float deltaT = 0.05f;
inline constexpr int32_t rebuildPeriod = 16;
}

enum class ExportID : uint32_t { StepCount, NumExports };

struct Drift { Vector3 v; };
struct StepCount { int32_t n; };

// what a sensor's rays met: distance (0: nothing within reach), the entity id
// and the surface normal
struct RayFan {
    float hitT[consts::raysPerSensor];
    int32_t hitEntity[consts::raysPerSensor];
    Vector3 hitNormal[consts::raysPerSensor];
};

struct Box : public madrona::Archetype<madrona::phys::RigidBody, Drift> {};
struct Pillar : public madrona::Archetype<madrona::phys::RigidBody> {};
struct Sensor : public madrona::Archetype<Position, Drift, RayFan> {};

class Engine;

struct Sim : public madrona::WorldBase {
    struct Config {
        uint32_t seed;
        uint32_t worldBase;
        madrona::phys::ObjectManager *rigidBodyObjMgr;
        uint32_t flags;
    };
    struct WorldInit {};

    static void registerTypes(madrona::ECSRegistry &registry, const Config &cfg);
    static void setupTasks(madrona::TaskGraphManager &taskgraph_mgr,
                           const Config &cfg);

    Sim(Engine &ctx, const Config &cfg, const WorldInit &init);

    Entity boxes[consts::maxBoxes];
    Entity pillars[consts::numPillars];
    int32_t numBoxes;
};

class Engine : public madrona::CustomContext<Engine, Sim> {
public:
    using CustomContext::CustomContext;
};

}

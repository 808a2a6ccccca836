// A pit of spheres, boxes and wedges: the sphere primitive paths of the
// narrowphase (sphere-sphere, sphere-plane, sphere-hull through GJK) that no
// ray-casting simulator can exercise (the reference's BVH::traceRay asserts on
// spheres).  No agents: bodies are kicked by forces drawn from the world's RNG.
// Per world: plane + 4 walls + 6 spheres + 4 boxes + 1 L-shaped block made of
// two hull primitives + 3 "coins" (12- and 16-sided prisms: hulls and faces
// larger than the fused kernel's LDS staging, so its fallback paths run) = 19
// bodies in two archetypes.  Written only against the public Madrona API.
#pragma once

#include <madrona/taskgraph_builder.hpp>
#include <madrona/custom_context.hpp>
#include <madrona/components.hpp>
#include <madrona/math.hpp>
#include <madrona/rand.hpp>
#include <madrona/physics.hpp>

namespace ballpit {

using madrona::Entity;
using madrona::RandKey;
using madrona::RNG;
using madrona::base::Position;
using madrona::base::Rotation;
using madrona::base::Scale;
using madrona::base::ObjectID;
using madrona::math::Vector3;
using madrona::math::Quat;
using madrona::phys::Velocity;
using madrona::phys::ResponseType;
using madrona::phys::ExternalForce;
using madrona::phys::ExternalTorque;
using madrona::phys::RigidBody;

namespace consts {
inline constexpr int32_t numSpheres = 6;
inline constexpr int32_t numBoxes = 5;
inline constexpr int32_t numCoins = 3;
inline constexpr int32_t numMovable = numSpheres + numBoxes + numCoins;
inline constexpr int32_t numWalls = 4;
inline constexpr int32_t numJoints = 8;
inline constexpr int32_t episodeLen = 150;
inline constexpr float pitSize = 10.f;
inline constexpr float wallThickness = 0.5f;
inline constexpr float wallHeight = 3.f;
inline constexpr float deltaT = 0.04f;
inline constexpr int32_t numPhysicsSubsteps = 4;
// crowd mode (Config::numExtra): extra spheres / boxes spread over a wider
// field, up to this many -- worlds of more than 64 and more than 128 bodies,
// i.e. the multi-chunk loops of the step kernels and the in-place BVH rebuild
inline constexpr int32_t maxExtra = 150;
inline constexpr int32_t maxRigidBodies = 24 + maxExtra;
}

enum class ExportID : uint32_t {
    Reset,
    StepsRemaining,
    QueryProbe,
    NumExports,
};

enum class SimObject : int32_t {
    Sphere,
    Box,
    Coin12,     // 12-gon prism: more vertices than the wave path keeps in LDS
    Coin16,     // 16-gon prism: its cap faces outgrow the clipping scratch
    Wall,
    Plane,
    LBlock,     // two hull primitives in one object
    NumObjects,
};

struct WorldReset {
    int32_t reset;
};

struct StepsRemaining {
    int32_t t;
};

struct LevelState {
    Entity movable[consts::numMovable];
    // fixed joints tying consecutive objects into a chain (more joints per
    // world than the step kernel stages next to the CU, and joints that share
    // bodies: they must be solved in order)
    Entity joints[consts::numJoints];
    Entity extra[consts::maxExtra];
};

// Running digest of PhysicsSystem box queries made every step (probeSystem):
// the CPU build asks findEntitiesWithinAABB, the GPU build the wave-cooperative
// findFirstEntitiesWithinAABBsWave -- equal digests = equal answers on worlds
// with spheres, multi-primitive objects and (crowd mode) more leaves than lanes.
struct QueryProbe {
    uint32_t digest;
    int32_t found;
};

// which movable object this is (kick schedule)
struct KickIndex {
    int32_t idx;
};

struct MovableObject : public madrona::Archetype<
    RigidBody,
    KickIndex
> {};

struct StaticObject : public madrona::Archetype<
    RigidBody
> {};

class Engine;

struct Sim : public madrona::WorldBase {
    struct Config {
        uint32_t seed;
        uint32_t worldBase;
        uint32_t autoResetDenom;
        uint32_t numExtra;
        madrona::phys::ObjectManager *rigidBodyObjMgr;
        // 1: every other joint of the chain is a hinge
        uint32_t hingeMode;
        // 1: PhysicsSystem::checkEntityAABBOverlap steers the kicks
        uint32_t overlapMode;
        // >= 0: only this world (global index) gets the numExtra bodies
        int32_t crowdedWorld;
    };

    struct WorldInit {};

    static void registerTypes(madrona::ECSRegistry &registry,
                              const Config &cfg);

    static void setupTasks(madrona::TaskGraphManager &taskgraph_mgr,
                           const Config &cfg);

    Sim(Engine &ctx, const Config &cfg, const WorldInit &init);

    RandKey initRandKey;
    RNG rng;
    RNG resetRng;
    uint32_t curWorldEpisode;
    uint32_t autoResetDenom;
    int32_t numExtra;
    uint32_t hingeMode;
    uint32_t overlapMode;
    Entity floorPlane;
    Entity walls[consts::numWalls];
};

class Engine : public madrona::CustomContext<Engine, Sim> {
public:
    using CustomContext::CustomContext;
};

}

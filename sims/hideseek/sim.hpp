// Config 4 of BASELINE.json: Hide-and-Seek-shaped worlds: 5
// agents (3 hiders, 2 seekers), 9 movable boxes, 2 ramps (wedge hulls), 12
// walls and a floor plane = 29 rigid bodies per world, with XPBD + BVH
// broadphase, lock/unlock of objects, line-of-sight rewards through
// BVH::traceRay, a 30-ray lidar, and ~2.8 KB of observations per world (the
// columns a multi-GPU run all-gathers every step).
// Written only against the public Madrona API; the same source builds against
// the reference headers + CPU backend (the oracle) and madrona_amd's overlay.
#pragma once

#include <madrona/taskgraph_builder.hpp>
#include <madrona/custom_context.hpp>
#include <madrona/components.hpp>
#include <madrona/math.hpp>
#include <madrona/rand.hpp>
#include <madrona/physics.hpp>

namespace hideseek {

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
inline constexpr int32_t numHiders = 3;
inline constexpr int32_t numSeekers = 2;
inline constexpr int32_t numAgents = numHiders + numSeekers;
inline constexpr int32_t numBoxes = 9;
inline constexpr int32_t numRamps = 2;
inline constexpr int32_t numMovable = numBoxes + numRamps;
inline constexpr int32_t numBorderWalls = 4;
inline constexpr int32_t numInteriorWalls = 8;
inline constexpr int32_t numLidarSamples = 30;
inline constexpr int32_t episodeLen = 240;
inline constexpr int32_t prepLen = 96;          // seekers see nothing yet
// the arena is a gridDim x gridDim board of cellSize squares centred at 0
inline constexpr int32_t gridDim = 6;
inline constexpr float cellSize = 4.f;
inline constexpr float arenaSize = cellSize * gridDim;
inline constexpr float wallThickness = 0.5f;
inline constexpr float wallHeight = 2.5f;
inline constexpr float deltaT = 0.04f;
inline constexpr int32_t numPhysicsSubsteps = 4;
inline constexpr int32_t numMoveAmountBuckets = 4;
inline constexpr int32_t numMoveAngleBuckets = 8;
inline constexpr int32_t numTurnBuckets = 5;
inline constexpr int32_t maxRigidBodies = 40;
}

enum class ExportID : uint32_t {
    Reset,
    Action,
    Reward,
    Done,
    SelfObservation,
    AgentObservations,
    BoxObservations,
    RampObservations,
    Lidar,
    StepsRemaining,
    NumExports,
};

// index into the ObjectManager built by the manager (mgr.cpp)
enum class SimObject : int32_t {
    Box,
    LongBox,
    Ramp,
    Wall,
    Agent,
    Plane,
    LBlock,     // a box made of two hull primitives (rays meet both)
    NumObjects,
};

enum class EntityType : uint32_t {
    None,
    Box,
    Ramp,
    Wall,
    Hider,
    Seeker,
    NumTypes,
};

// ---- agent interface ---------------------------------------------------------
struct WorldReset {
    int32_t reset;
};

struct Action {
    int32_t moveAmount;  // [0, 3]
    int32_t moveAngle;   // [0, 7]
    int32_t rotate;      // [-2, 2]
    int32_t lock;        // 1 = toggle the lock of the object in front
};

struct Reward {
    float v;
};

struct Done {
    int32_t v;
};

struct SelfObservation {
    float x;
    float y;
    float z;
    float facing;
    float vx;
    float vy;
    float isHider;
    float prepRemaining;
};

struct AgentObservation {
    float dx;
    float dy;
    float isHider;
    float visible;
};

struct AgentObservations {
    AgentObservation obs[consts::numAgents - 1];
};

struct ObjectObservation {
    float dx;
    float dy;
    float dz;
    float speed2;
    float locked;
};

struct BoxObservations {
    ObjectObservation obs[consts::numBoxes];
};

struct RampObservations {
    ObjectObservation obs[consts::numRamps];
};

struct LidarSample {
    float depth;
    float encodedType;
};

struct Lidar {
    LidarSample samples[consts::numLidarSamples];
};

struct StepsRemaining {
    uint32_t t;
};

struct OtherAgents {
    Entity e[consts::numAgents - 1];
};

// seen[i] != 0: OtherAgents::e[i] is in line of sight
struct Visibility {
    uint32_t seen[consts::numAgents - 1];
};

// ---- level objects -------------------------------------------------------------
struct LockState {
    int32_t locked;      // ResponseType is Static while locked
    int32_t byHiders;    // which team holds the lock
};

struct LevelState {
    Entity boxes[consts::numBoxes];
    Entity ramps[consts::numRamps];
    Entity walls[consts::numInteriorWalls];
};

// ---- archetypes ----------------------------------------------------------------
struct Agent : public madrona::Archetype<
    RigidBody,
    Action,
    Reward,
    Done,
    SelfObservation,
    AgentObservations,
    BoxObservations,
    RampObservations,
    Lidar,
    StepsRemaining,
    OtherAgents,
    Visibility,
    EntityType
> {};

struct MovableObject : public madrona::Archetype<
    RigidBody,
    LockState,
    EntityType
> {};

struct StaticObject : public madrona::Archetype<
    RigidBody,
    EntityType
> {};

class Engine;

struct Sim : public madrona::WorldBase {
    struct Config {
        uint32_t seed;
        uint32_t worldBase;
        // a world resets itself with probability 1/autoResetDenom per step
        // (0 disables) in addition to episode timeouts / external resets
        uint32_t autoResetDenom;
        madrona::phys::ObjectManager *rigidBodyObjMgr;
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
    Entity floorPlane;
    Entity borders[consts::numBorderWalls];
    Entity agents[consts::numAgents];
};

class Engine : public madrona::CustomContext<Engine, Sim> {
public:
    using CustomContext::CustomContext;
};

}

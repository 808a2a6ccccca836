// Config 2 of BASELINE.json: Escape-Room-shaped ECS, physics off
// (SURVEY.md §8d).  A synthetic simulator written only against the public
// Madrona API; compiles unchanged against the reference headers (CPU
// TaskGraphExecutor = the oracle) and against madrona_amd's overlay (HIP).
//
// World shape: 2 agents, 3 rooms each with 1 door, 2 buttons and 6 movable
// cubes => 2 Agent + 18 PhysicsEntity + 3 DoorEntity + 6 ButtonEntity rows.
// The rigid-body bundle carries the same columns (and bytes) as the
// reference's physics RigidBody bundle + XPBD solver state so table rows have
// the size SURVEY.md §8 quotes; a kinematic integrator stands in for the
// physics nodes, which config 3 adds.
#pragma once

#include <madrona/taskgraph_builder.hpp>
#include <madrona/custom_context.hpp>
#include <madrona/components.hpp>
#include <madrona/math.hpp>
#include <madrona/rand.hpp>

namespace escape {

using madrona::Entity;
using madrona::RandKey;
using madrona::RNG;
using madrona::base::Position;
using madrona::base::Rotation;
using madrona::base::Scale;
using madrona::base::ObjectID;
using madrona::math::Vector3;
using madrona::math::Quat;

namespace consts {
inline constexpr int32_t numAgents = 2;
inline constexpr int32_t numRooms = 3;
inline constexpr int32_t numButtonsPerRoom = 2;
inline constexpr int32_t numCubesPerRoom = 6;
inline constexpr int32_t numLidarSamples = 30;
inline constexpr int32_t episodeLen = 200;
inline constexpr float worldWidth = 20.f;
inline constexpr float roomLength = 13.f;
inline constexpr float worldLength = roomLength * numRooms;
inline constexpr float agentRadius = 1.f;
inline constexpr float buttonWidth = 1.3f;
inline constexpr float deltaT = 0.04f;
inline constexpr float doorSpeed = 20.f;
inline constexpr float rewardPerDist = 0.05f;
inline constexpr float slackReward = -0.005f;
inline constexpr int32_t numMoveAmountBuckets = 4;
inline constexpr int32_t numMoveAngleBuckets = 8;
inline constexpr int32_t numTurnBuckets = 5;
}

enum class ExportID : uint32_t {
    Reset,
    Action,
    Reward,
    Done,
    SelfObservation,
    PartnerObservation,
    RoomEntityObservations,
    DoorObservation,
    Lidar,
    StepsRemaining,
    NumExports,
};

enum class SimObject : int32_t {
    Cube,
    Wall,
    Door,
    Agent,
    Button,
    Plane,
    NumObjects,
};

enum class EntityType : uint32_t {
    None,
    Button,
    Cube,
    Wall,
    Agent,
    Door,
    NumTypes,
};

enum class ResponseType : uint32_t {
    Dynamic,
    Kinematic,
    Static,
};

// ---- rigid-body shaped components (sizes match the reference's physics.hpp /
// xpbd.cpp components) ---------------------------------------
struct Velocity {
    Vector3 linear;
    Vector3 angular;
};

struct ExternalForce : Vector3 {
    ExternalForce(Vector3 v) : Vector3(v) {}
};

struct ExternalTorque : Vector3 {
    ExternalTorque(Vector3 v) : Vector3(v) {}
};

struct SubstepPrevState {
    Vector3 prevPosition;
    Quat prevRotation;
};

struct PreSolvePositional {
    Vector3 x;
    Quat q;
};

struct PreSolveVelocity {
    Vector3 v;
    Vector3 omega;
};

struct SolverState : madrona::Bundle<
    SubstepPrevState,
    PreSolvePositional,
    PreSolveVelocity
> {};

struct RigidBody : madrona::Bundle<
    madrona::base::ObjectInstance,
    ResponseType,
    Velocity,
    ExternalForce,
    ExternalTorque,
    SolverState
> {};

// ---- agent interface ---------------------------------------------------------
struct WorldReset {
    int32_t reset;
};

struct Action {
    int32_t moveAmount;  // [0, 3]
    int32_t moveAngle;   // [0, 7]
    int32_t rotate;      // [-2, 2]
    int32_t grab;        // unused (no physics joints in this config)
};

struct Reward {
    float v;
};

struct Done {
    int32_t v;
};

struct SelfObservation {
    float roomX;
    float roomY;
    float globalX;
    float globalY;
    float globalZ;
    float maxY;
    float facing;        // z component of the heading quaternion
    float isGrabbing;
};

struct PartnerObservation {
    float dx;
    float dy;
    float isGrabbing;
};

struct EntityObservation {
    float dx;
    float dy;
    float encodedType;
};

struct RoomEntityObservations {
    EntityObservation obs[consts::numCubesPerRoom + consts::numButtonsPerRoom + 1];
};

struct DoorObservation {
    float dx;
    float dy;
    float isOpen;
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

struct Progress {
    float maxY;
};

struct OtherAgents {
    Entity e[consts::numAgents - 1];
};

struct GrabState {
    Entity constraintEntity;
};

// ---- level objects -------------------------------------------------------------
struct OpenState {
    int32_t isOpen;
};

struct DoorProperties {
    Entity buttons[4];
    int32_t numButtons;
    int32_t isPersistent;
};

struct ButtonState {
    int32_t isPressed;
};

struct Room {
    Entity door;
    Entity buttons[consts::numButtonsPerRoom];
    Entity cubes[consts::numCubesPerRoom];
};

struct LevelState {
    Room rooms[consts::numRooms];
};

// ---- archetypes ----------------------------------------------------------------
struct Agent : public madrona::Archetype<
    RigidBody,
    Action,
    Reward,
    Done,
    SelfObservation,
    PartnerObservation,
    RoomEntityObservations,
    DoorObservation,
    Lidar,
    StepsRemaining,
    Progress,
    OtherAgents,
    GrabState,
    EntityType
> {};

struct PhysicsEntity : public madrona::Archetype<
    RigidBody,
    EntityType
> {};

struct DoorEntity : public madrona::Archetype<
    RigidBody,
    OpenState,
    DoorProperties,
    EntityType
> {};

struct ButtonEntity : public madrona::Archetype<
    madrona::base::ObjectInstance,
    ButtonState,
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
    Entity agents[consts::numAgents];
};

class Engine : public madrona::CustomContext<Engine, Sim> {
public:
    using CustomContext::CustomContext;
};

}

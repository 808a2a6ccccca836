// Config 3 of BASELINE.json: Escape-Room-shaped worlds with rigid-body physics
// (LBVH broadphase + SAT narrowphase + XPBD).
// Written only against the public Madrona API (madrona/physics.hpp included);
// the same source builds against the reference headers + CPU backend (the
// oracle) and against madrona_amd's overlay (HIP).
//
// World shape: floor plane, 4 border walls, 2 agents, 3 rooms each with 2 wall
// segments around a door, 2 buttons and 4 movable cubes:
//   2 Agent + 23 PhysicsEntity + 3 DoorEntity rigid bodies, 6 ButtonEntity.
#pragma once

#include <madrona/taskgraph_builder.hpp>
#include <madrona/custom_context.hpp>
#include <madrona/components.hpp>
#include <madrona/math.hpp>
#include <madrona/rand.hpp>
#include <madrona/physics.hpp>

// BASELINE config 5 = this simulator + the batch ray caster: the variant in
// sims/escape_room_render compiles these sources with ESCPHYS_RENDER defined
// (every body drawable, a camera on each agent, 64 x 64 RGB-D per agent).
#ifdef ESCPHYS_RENDER
#include <madrona/render/ecs.hpp>
#define ESCPHYS_DRAWABLE , madrona::render::Renderable
#define ESCPHYS_VIEWER , madrona::render::RenderCamera
#else
#define ESCPHYS_DRAWABLE
#define ESCPHYS_VIEWER
#endif

namespace escphys {

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
inline constexpr int32_t numAgents = 2;
inline constexpr int32_t numRooms = 3;
inline constexpr int32_t numButtonsPerRoom = 2;
inline constexpr int32_t numCubesPerRoom = 4;
inline constexpr int32_t numBorderWalls = 4;
inline constexpr int32_t numLidarSamples = 30;
inline constexpr int32_t episodeLen = 200;
inline constexpr float worldWidth = 20.f;
inline constexpr float roomLength = 13.f;
inline constexpr float worldLength = roomLength * numRooms;
inline constexpr float wallWidth = 1.f;
inline constexpr float wallHeight = 2.f;
inline constexpr float doorWidth = 4.f;
inline constexpr float agentRadius = 1.f;
inline constexpr float buttonWidth = 1.3f;
inline constexpr float cubeSize = 1.5f;
inline constexpr float deltaT = 0.04f;
inline constexpr int32_t numPhysicsSubsteps = 4;
inline constexpr float doorSpeed = 20.f;
inline constexpr float rewardPerDist = 0.05f;
inline constexpr float slackReward = -0.005f;
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
    PartnerObservation,
    RoomEntityObservations,
    DoorObservation,
    Lidar,
    StepsRemaining,
#ifdef ESCPHYS_RENDER
    RGB,
    Depth,
#endif
    NumExports,
};

// index into the ObjectManager built by the manager (mgr.cpp)
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

// ---- agent interface ---------------------------------------------------------
struct WorldReset {
    int32_t reset;
};

struct Action {
    int32_t moveAmount;  // [0, 3]
    int32_t moveAngle;   // [0, 7]
    int32_t rotate;      // [-2, 2]
    int32_t grab;        // 0 = keep, 1 = toggle
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
    float facing;
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
    Entity walls[2];
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
    ESCPHYS_DRAWABLE
    ESCPHYS_VIEWER
> {};

struct PhysicsEntity : public madrona::Archetype<
    RigidBody,
    EntityType
    ESCPHYS_DRAWABLE
> {};

struct DoorEntity : public madrona::Archetype<
    RigidBody,
    OpenState,
    DoorProperties,
    EntityType
    ESCPHYS_DRAWABLE
> {};

struct ButtonEntity : public madrona::Archetype<
    madrona::base::ObjectInstance,
    ButtonState,
    EntityType
    ESCPHYS_DRAWABLE
> {};

#ifdef ESCPHYS_RENDER
// the world's one light
struct SunEntity : public madrona::Archetype<
    Position,
    madrona::render::LightDescDirection,
    madrona::render::LightDescType,
    madrona::render::LightDescShadow,
    madrona::render::LightDescCutoffAngle,
    madrona::render::LightDescIntensity,
    madrona::render::LightDescActive,
    madrona::render::LightCarrier
> {};
#endif

class Engine;

struct Sim : public madrona::WorldBase {
    struct Config {
        uint32_t seed;
        uint32_t worldBase;
        // a world resets itself with probability 1/autoResetDenom per step
        // (0 disables) in addition to episode timeouts / external resets
        uint32_t autoResetDenom;
        madrona::phys::ObjectManager *rigidBodyObjMgr;
#ifdef ESCPHYS_RENDER
        uint32_t sunCastsShadows;
        // reference CPU backend: where its render-prep systems append their
        // records (the manager's stand-in for the Vulkan renderer); null on
        // the GPU backend, whose render entities' rows are the records
        const madrona::render::RenderECSBridge *bridge;
#endif
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

#ifdef MADRONA_GPU_MODE
    // what grabQuerySystem found in front of each agent this step, for
    // grabSystem (sim.cpp)
    Entity grabTargets[consts::numAgents];
#endif
};

class Engine : public madrona::CustomContext<Engine, Sim> {
public:
    using CustomContext::CustomContext;
};

}

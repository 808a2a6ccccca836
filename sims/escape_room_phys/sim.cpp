#include "sim.hpp"

#include <cstring>

#ifdef MADRONA_GPU_MODE
#include <madrona/mw_gpu_entry.hpp>
#endif

// SIM_WAVE_API: this backend's wave-cooperative extensions (ordered create /
// destroy by a wavefront per world, box queries with a lane per BVH leaf,
// per-system occupancy) -- API the reference does not have.  -DSIM_PORTABLE
// builds the simulator WITHOUT them even under MADRONA_GPU_MODE: the sources as
// an unchanged reference simulator has them (plain makeEntity / destroyEntity /
// findEntitiesWithinAABB, one lane per world; only the reference's own GPU
// conventions remain: CustomParallelForNode for the ray systems,
// RecycleEntitiesNode).  lib<sim>_portable_hip.so, the `portable_sim` bench line.
#if defined(MADRONA_GPU_MODE) && !defined(SIM_PORTABLE)
#define SIM_WAVE_API 1
#endif
// (measurement: -DSIM_NO_ROW_CHAIN keeps the two door systems two nodes)
#if defined(SIM_WAVE_API) && !defined(SIM_NO_ROW_CHAIN)
#define SIM_ROW_CHAIN 1
#endif

using namespace madrona;
using namespace madrona::math;
using namespace madrona::phys;

#ifdef ESCPHYS_RENDER
namespace RenderingSystem = madrona::render::RenderingSystem;
#define ESCPHYS_IF_RENDER(...) __VA_ARGS__
#else
#define ESCPHYS_IF_RENDER(...)
#endif

namespace escphys {

// cos / sin of k * 2pi / 8 and of k * 2pi / 30 as literals: libm results differ
// in the last ulp between glibc and the device math library, literals do not.
static constexpr float kMoveSin[8] = {
    0.f, 0.70710678f, 1.f, 0.70710678f, 0.f, -0.70710678f, -1.f, -0.70710678f,
};
static constexpr float kMoveCos[8] = {
    1.f, 0.70710678f, 0.f, -0.70710678f, -1.f, -0.70710678f, 0.f, 0.70710678f,
};

static constexpr float kLidarCos[consts::numLidarSamples] = {
    1.f, 0.9781476f, 0.91354546f, 0.80901699f, 0.66913061f, 0.5f,
    0.30901699f, 0.10452846f, -0.10452846f, -0.30901699f, -0.5f,
    -0.66913061f, -0.80901699f, -0.91354546f, -0.9781476f, -1.f,
    -0.9781476f, -0.91354546f, -0.80901699f, -0.66913061f, -0.5f,
    -0.30901699f, -0.10452846f, 0.10452846f, 0.30901699f, 0.5f,
    0.66913061f, 0.80901699f, 0.91354546f, 0.9781476f,
};
static constexpr float kLidarSin[consts::numLidarSamples] = {
    0.f, 0.20791169f, 0.40673664f, 0.58778525f, 0.74314483f, 0.8660254f,
    0.95105652f, 0.9945219f, 0.9945219f, 0.95105652f, 0.8660254f,
    0.74314483f, 0.58778525f, 0.40673664f, 0.20791169f, 0.f,
    -0.20791169f, -0.40673664f, -0.58778525f, -0.74314483f, -0.8660254f,
    -0.95105652f, -0.9945219f, -0.9945219f, -0.95105652f, -0.8660254f,
    -0.74314483f, -0.58778525f, -0.40673664f, -0.20791169f,
};

void Sim::registerTypes(ECSRegistry &registry, const Config &cfg)
{
    (void)cfg;
    base::registerTypes(registry);
    PhysicsSystem::registerTypes(registry);
    ESCPHYS_IF_RENDER(RenderingSystem::registerTypes(registry, cfg.bridge);)

    registry.registerComponent<Action>();
    registry.registerComponent<Reward>();
    registry.registerComponent<Done>();
    registry.registerComponent<SelfObservation>();
    registry.registerComponent<PartnerObservation>();
    registry.registerComponent<RoomEntityObservations>();
    registry.registerComponent<DoorObservation>();
    registry.registerComponent<Lidar>();
    registry.registerComponent<StepsRemaining>();
    registry.registerComponent<Progress>();
    registry.registerComponent<OtherAgents>();
    registry.registerComponent<GrabState>();
    registry.registerComponent<EntityType>();
    registry.registerComponent<OpenState>();
    registry.registerComponent<DoorProperties>();
    registry.registerComponent<ButtonState>();

    registry.registerSingleton<WorldReset>();
    registry.registerSingleton<LevelState>();

    registry.registerArchetype<Agent>();
    registry.registerArchetype<PhysicsEntity>();
    registry.registerArchetype<DoorEntity>();
    registry.registerArchetype<ButtonEntity>();
#ifdef ESCPHYS_RENDER
    registry.registerArchetype<SunEntity>();
    registry.exportColumn<render::RaycastOutputArchetype, render::RGBOutputBuffer>(
        (uint32_t)ExportID::RGB);
    registry.exportColumn<render::RaycastOutputArchetype, render::DepthOutputBuffer>(
        (uint32_t)ExportID::Depth);
#endif

    registry.exportSingleton<WorldReset>((uint32_t)ExportID::Reset);
    registry.exportColumn<Agent, Action>((uint32_t)ExportID::Action);
    registry.exportColumn<Agent, Reward>((uint32_t)ExportID::Reward);
    registry.exportColumn<Agent, Done>((uint32_t)ExportID::Done);
    registry.exportColumn<Agent, SelfObservation>(
        (uint32_t)ExportID::SelfObservation);
    registry.exportColumn<Agent, PartnerObservation>(
        (uint32_t)ExportID::PartnerObservation);
    registry.exportColumn<Agent, RoomEntityObservations>(
        (uint32_t)ExportID::RoomEntityObservations);
    registry.exportColumn<Agent, DoorObservation>(
        (uint32_t)ExportID::DoorObservation);
    registry.exportColumn<Agent, Lidar>((uint32_t)ExportID::Lidar);
    registry.exportColumn<Agent, StepsRemaining>(
        (uint32_t)ExportID::StepsRemaining);
}

// ---------------------------------------------------------------------------
// level generation
// ---------------------------------------------------------------------------
static inline float randInRange(RNG &rng, float lo, float hi)
{
    return lo + rng.sampleUniform() * (hi - lo);
}

// Fills the RigidBody bundle and (re)registers the body with the broadphase.
static inline void setupRigidBody(Engine &ctx, Entity e, Vector3 pos, Quat rot,
                                  SimObject obj, EntityType type,
                                  ResponseType response, Diag3x3 scale)
{
    ObjectID obj_id { (int32_t)obj };

    ctx.get<Position>(e) = pos;
    ctx.get<Rotation>(e) = rot;
    ctx.get<Scale>(e) = scale;
    ctx.get<ObjectID>(e) = obj_id;
    ctx.get<ResponseType>(e) = response;
    ctx.get<Velocity>(e) = Velocity { Vector3::zero(), Vector3::zero() };
    ctx.get<ExternalForce>(e) = Vector3::zero();
    ctx.get<ExternalTorque>(e) = Vector3::zero();
    ctx.get<broadphase::LeafID>(e) =
        PhysicsSystem::registerEntity(ctx, e, obj_id);
    ctx.get<EntityType>(e) = type;
    ESCPHYS_IF_RENDER(RenderingSystem::makeEntityRenderable(ctx, e);)
}

static inline void registerRigidBodyEntity(Engine &ctx, Entity e)
{
    ctx.get<broadphase::LeafID>(e) =
        PhysicsSystem::registerEntity(ctx, e, ctx.get<ObjectID>(e));
}

static Entity makeWall(Engine &ctx, float x_min, float x_max, float y)
{
    Entity wall = ctx.makeEntity<PhysicsEntity>();
    setupRigidBody(ctx, wall,
        Vector3 { (x_min + x_max) * 0.5f, y, consts::wallHeight * 0.5f },
        Quat { 1, 0, 0, 0 }, SimObject::Wall, EntityType::Wall,
        ResponseType::Static,
        Diag3x3 { x_max - x_min, consts::wallWidth, consts::wallHeight });
    return wall;
}

static void generateLevel(Engine &ctx, RNG &rng)
{
    LevelState &level = ctx.singleton<LevelState>();

    const float half_width = consts::worldWidth / 2.f;

    for (int32_t r = 0; r < consts::numRooms; r++) {
        Room &room = level.rooms[r];
        const float y_min = (float)r * consts::roomLength;
        const float y_max = y_min + consts::roomLength;

        for (int32_t b = 0; b < consts::numButtonsPerRoom; b++) {
            Entity button = ctx.makeEntity<ButtonEntity>();
            Vector3 pos {
                randInRange(rng, -half_width + 2.f, half_width - 2.f),
                randInRange(rng, y_min + 2.f, y_max - 3.f),
                0.f,
            };
            ctx.get<Position>(button) = pos;
            ctx.get<Rotation>(button) = Quat { 1, 0, 0, 0 };
            ctx.get<Scale>(button) = Diag3x3 {
                consts::buttonWidth, consts::buttonWidth, 0.2f,
            };
            ctx.get<ObjectID>(button) = ObjectID { (int32_t)SimObject::Button };
            ctx.get<ButtonState>(button).isPressed = 0;
            ctx.get<EntityType>(button) = EntityType::Button;
            ESCPHYS_IF_RENDER(RenderingSystem::makeEntityRenderable(ctx, button);)
            room.buttons[b] = button;
        }

        // wall with a door gap
        float door_x = randInRange(rng, -half_width + 4.f, half_width - 4.f);
        room.walls[0] = makeWall(ctx, -half_width,
                                 door_x - consts::doorWidth * 0.5f, y_max);
        room.walls[1] = makeWall(ctx, door_x + consts::doorWidth * 0.5f,
                                 half_width, y_max);

        Entity door = ctx.makeEntity<DoorEntity>();
        setupRigidBody(ctx, door, Vector3 { door_x, y_max, 0.f },
            Quat { 1, 0, 0, 0 }, SimObject::Door, EntityType::Door,
            ResponseType::Static,
            Diag3x3 { consts::doorWidth * 0.8f, consts::wallWidth,
                      2.f * consts::wallHeight });
        ctx.get<OpenState>(door).isOpen = 0;
        DoorProperties &props = ctx.get<DoorProperties>(door);
        for (int32_t b = 0; b < 4; b++) {
            props.buttons[b] = b < consts::numButtonsPerRoom ?
                room.buttons[b] : Entity::none();
        }
        props.numButtons = 1 + rng.sampleI32(0, consts::numButtonsPerRoom);
        props.isPersistent = rng.sampleBool() ? 1 : 0;
        room.door = door;

        for (int32_t c = 0; c < consts::numCubesPerRoom; c++) {
            Entity cube = ctx.makeEntity<PhysicsEntity>();
            Vector3 pos {
                randInRange(rng, -half_width + 1.5f, half_width - 1.5f),
                randInRange(rng, y_min + 4.f, y_max - 2.5f),
                consts::cubeSize * 0.5f,
            };
            setupRigidBody(ctx, cube, pos, Quat { 1, 0, 0, 0 },
                SimObject::Cube, EntityType::Cube, ResponseType::Dynamic,
                Diag3x3 { consts::cubeSize, consts::cubeSize,
                          consts::cubeSize });
            room.cubes[c] = cube;
        }
    }
}

static void resetPersistentEntities(Engine &ctx, RNG &rng)
{
    Sim &sim = ctx.data();
    const float half_width = consts::worldWidth / 2.f;

    registerRigidBodyEntity(ctx, sim.floorPlane);
    for (int32_t i = 0; i < consts::numBorderWalls; i++) {
        registerRigidBodyEntity(ctx, sim.borders[i]);
    }

    for (int32_t i = 0; i < consts::numAgents; i++) {
        Entity agent = sim.agents[i];
        registerRigidBodyEntity(ctx, agent);

        // agents start in their own half so they never spawn interpenetrating
        float x_lo = i == 0 ? -half_width + 2.f : 1.5f;
        float x_hi = i == 0 ? -1.5f : half_width - 2.f;
        Vector3 pos {
            randInRange(rng, x_lo, x_hi),
            randInRange(rng, 1.5f, 3.f),
            1.f,
        };
        // heading: rotation about z by one of the 8 move angles; half-angle
        // terms from the tables via the half-angle identities (libm free)
        int32_t heading = rng.sampleI32(0, 8);
        float c = kMoveCos[heading];
        float s = kMoveSin[heading];
        float ch = sqrtf((1.f + c) * 0.5f);
        float sh = sqrtf((1.f - c) * 0.5f);
        if (s < 0.f) sh = -sh;
        Quat rot = Quat { ch, 0.f, 0.f, sh }.normalize();

        ctx.get<Position>(agent) = pos;
        ctx.get<Rotation>(agent) = rot;
        ctx.get<Velocity>(agent) = Velocity { Vector3::zero(), Vector3::zero() };
        ctx.get<ExternalForce>(agent) = Vector3::zero();
        ctx.get<ExternalTorque>(agent) = Vector3::zero();
        ctx.get<Action>(agent) = Action { 0, 0, 0, 0 };
        ctx.get<Progress>(agent).maxY = pos.y;
        ctx.get<StepsRemaining>(agent).t = consts::episodeLen;
        ctx.get<GrabState>(agent).constraintEntity = Entity::none();
        ctx.get<Reward>(agent).v = 0.f;
        ctx.get<Done>(agent).v = 0;
    }
}

static void createPersistentEntities(Engine &ctx)
{
    Sim &sim = ctx.data();
    const float half_width = consts::worldWidth / 2.f;
    const float w = consts::wallWidth;

    sim.floorPlane = ctx.makeEntity<PhysicsEntity>();
    setupRigidBody(ctx, sim.floorPlane, Vector3 { 0, 0, 0 },
        Quat { 1, 0, 0, 0 }, SimObject::Plane, EntityType::None,
        ResponseType::Static, Diag3x3 { 1, 1, 1 });

    // back, front, left, right
    const Vector3 border_pos[consts::numBorderWalls] = {
        { 0.f, -w * 0.5f, consts::wallHeight * 0.5f },
        { 0.f, consts::worldLength + 1.5f * w, consts::wallHeight * 0.5f },
        { -half_width - w * 0.5f, consts::worldLength * 0.5f,
          consts::wallHeight * 0.5f },
        { half_width + w * 0.5f, consts::worldLength * 0.5f,
          consts::wallHeight * 0.5f },
    };
    const Diag3x3 border_scale[consts::numBorderWalls] = {
        { consts::worldWidth + 2.f * w, w, consts::wallHeight },
        { consts::worldWidth + 2.f * w, w, consts::wallHeight },
        { w, consts::worldLength + 4.f * w, consts::wallHeight },
        { w, consts::worldLength + 4.f * w, consts::wallHeight },
    };
    for (int32_t i = 0; i < consts::numBorderWalls; i++) {
        sim.borders[i] = ctx.makeEntity<PhysicsEntity>();
        setupRigidBody(ctx, sim.borders[i], border_pos[i],
            Quat { 1, 0, 0, 0 }, SimObject::Wall, EntityType::Wall,
            ResponseType::Static, border_scale[i]);
    }

    for (int32_t i = 0; i < consts::numAgents; i++) {
        Entity agent = ctx.makeEntity<Agent>();
        sim.agents[i] = agent;

        setupRigidBody(ctx, agent, Vector3 { 0, 0, 1.f }, Quat { 1, 0, 0, 0 },
            SimObject::Agent, EntityType::Agent, ResponseType::Dynamic,
            Diag3x3 { 1.5f, 1.5f, 2.f });
        // (the reference Escape Room's camera: 100 degrees, at head height)
        ESCPHYS_IF_RENDER(RenderingSystem::attachEntityToView(ctx, agent, 100.f,
            0.001f, 1.5f * math::up);)
    }

    for (int32_t i = 0; i < consts::numAgents; i++) {
        OtherAgents &others = ctx.get<OtherAgents>(sim.agents[i]);
        int32_t out = 0;
        for (int32_t j = 0; j < consts::numAgents; j++) {
            if (j != i) {
                others.e[out++] = sim.agents[j];
            }
        }
    }
}

static void initWorld(Engine &ctx)
{
    Sim &sim = ctx.data();

    // every body re-registers with an emptied BVH (rebuilt on the next update)
    PhysicsSystem::reset(ctx);

    // a fresh RNG stream per (world, episode)
    // the stream lives in registers while the level is generated (the world
    // object is in memory that every component store might alias)
    RNG rng(rand::split_i(sim.initRandKey, sim.curWorldEpisode++));

    resetPersistentEntities(ctx, rng);
    generateLevel(ctx, rng);

    sim.rng = rng;
}

static void cleanupWorld(Engine &ctx)
{
    Sim &sim = ctx.data();
    for (int32_t i = 0; i < consts::numAgents; i++) {
        GrabState &grab = ctx.get<GrabState>(sim.agents[i]);
        if (grab.constraintEntity != Entity::none()) {
            ctx.destroyEntity(grab.constraintEntity);
            grab.constraintEntity = Entity::none();
        }
    }

    LevelState &level = ctx.singleton<LevelState>();
    for (int32_t r = 0; r < consts::numRooms; r++) {
        Room &room = level.rooms[r];
#ifdef ESCPHYS_RENDER
        // (render entities go first, as the API asks)
        for (int32_t c = 0; c < consts::numCubesPerRoom; c++) {
            RenderingSystem::cleanupRenderableEntity(ctx, room.cubes[c]);
        }
        RenderingSystem::cleanupRenderableEntity(ctx, room.walls[0]);
        RenderingSystem::cleanupRenderableEntity(ctx, room.walls[1]);
        RenderingSystem::cleanupRenderableEntity(ctx, room.door);
        for (int32_t b = 0; b < consts::numButtonsPerRoom; b++) {
            RenderingSystem::cleanupRenderableEntity(ctx, room.buttons[b]);
        }
#endif
        for (int32_t c = 0; c < consts::numCubesPerRoom; c++) {
            ctx.destroyEntity(room.cubes[c]);
        }
        ctx.destroyEntity(room.walls[0]);
        ctx.destroyEntity(room.walls[1]);
        ctx.destroyEntity(room.door);
        for (int32_t b = 0; b < consts::numButtonsPerRoom; b++) {
            ctx.destroyEntity(room.buttons[b]);
        }
    }
}

// ---------------------------------------------------------------------------
// systems
// ---------------------------------------------------------------------------
inline void movementSystem(Engine &,
                           Action &action,
                           Rotation &rot,
                           ExternalForce &external_force,
                           ExternalTorque &external_torque)
{
    constexpr float move_max = 1000.f;
    constexpr float turn_max = 320.f;

    Quat cur_rot = rot;

    float move_amount = (float)action.moveAmount *
        (move_max / (float)(consts::numMoveAmountBuckets - 1));

    int32_t angle = action.moveAngle & (consts::numMoveAngleBuckets - 1);
    float f_x = move_amount * kMoveSin[angle];
    float f_y = move_amount * kMoveCos[angle];

    constexpr float turn_delta_per_bucket =
        turn_max / (float)(consts::numTurnBuckets / 2);
    float t_z = turn_delta_per_bucket * (float)action.rotate;

    external_force = cur_rot.rotateVec(Vector3 { f_x, f_y, 0.f });
    external_torque = Vector3 { 0.f, 0.f, t_z };
}

// Grab / release the cube in front of each agent with a fixed joint.  One
// invocation per world, agents in order: entity creation order inside a world
// is then the same on every backend (parallel per-agent creation would make
// joint ids depend on scheduling).
// (kept out of line on the GPU: inlined into grabSystem, the joint record and the
// entity bookkeeping push its kernel to 178 registers, i.e. two wavefronts per
// SIMD for a system that is nothing but dependent loads)
#ifdef MADRONA_GPU_MODE
#define ESCPHYS_COLD __attribute__((noinline))
#else
#define ESCPHYS_COLD
#endif

ESCPHYS_COLD static void attachGrab(Engine &ctx, Entity e, Entity grab_entity,
                                    Vector3 pos, Quat rot, GrabState &grab)
{
    Vector3 other_pos = ctx.get<Position>(grab_entity);
    Quat other_rot = ctx.get<Rotation>(grab_entity);

    Vector3 r1 = Vector3 { 0.f, 1.25f, 0.f };
    Vector3 r2 = Vector3::zero();
    Quat attach1 { 1, 0, 0, 0 };
    Quat attach2 = (other_rot.inv() * rot).normalize();
    float separation = (other_pos - pos).length() - 1.25f;

    grab.constraintEntity = PhysicsSystem::makeFixedJoint(
        ctx, e, grab_entity, attach1, attach2, r1, r2, separation);
}

ESCPHYS_COLD static void releaseGrab(Engine &ctx, Entity held, GrabState &grab)
{
    ctx.destroyEntity(held);
    grab.constraintEntity = Entity::none();
}

#ifdef SIM_WAVE_API
// The GPU graph splits the grab in two.  This node runs a wavefront per world
// (CustomParallelForNode<..., 64, 1, ...>): the box queries of all agents that
// reach for something, a lane per BVH leaf (grabbing moves no body, so the
// queries see what the CPU's sequential loop sees).  grabSystem then runs one
// lane per world and only creates / destroys joints: lanes of a wavefront that
// append to one table share one atomic, waves of their own would queue on it.
inline void grabQuerySystem(Engine &ctx, LevelState &)
{
    Sim &sim = ctx.data();

    AABB boxes[consts::numAgents];
    int32_t owner[consts::numAgents];
    int32_t num_boxes = 0;
    for (int32_t i = 0; i < consts::numAgents; i++) {
        Entity e = sim.agents[i];
        if (ctx.get<Action>(e).grab == 0 ||
                ctx.get<GrabState>(e).constraintEntity != Entity::none()) {
            continue;
        }
        Vector3 reach = ctx.get<Position>(e) +
            ctx.get<Rotation>(e).rotateVec(Vector3 { 0.f, 1.75f, 0.f });
        boxes[num_boxes] = AABB {
            reach - Vector3 { 1.f, 1.f, 1.f },
            reach + Vector3 { 1.f, 1.f, 1.f },
        };
        owner[num_boxes++] = i;
    }
    Entity first[consts::numAgents];
    PhysicsSystem::findFirstEntitiesWithinAABBsWave<consts::numAgents>(
        ctx, boxes, num_boxes, first, [&](Entity other) {
            return ctx.get<EntityType>(other) == EntityType::Cube;
        });
    if (threadIdx.x % 64 == 0) {
        for (int32_t i = 0; i < consts::numAgents; i++) {
            sim.grabTargets[i] = Entity::none();
        }
        for (int32_t b = 0; b < num_boxes; b++) {
            sim.grabTargets[owner[b]] = first[b];
        }
    }
}
#endif

#ifdef SIM_WAVE_API
// (the joint of attachGrab with the cube's pose already read)
ESCPHYS_COLD static void attachGrabAt(Engine &ctx, Entity e, Entity grab_entity,
                                      Vector3 pos, Quat rot, Vector3 other_pos,
                                      Quat other_rot, GrabState &grab)
{
    Vector3 r1 = Vector3 { 0.f, 1.25f, 0.f };
    Vector3 r2 = Vector3::zero();
    Quat attach1 { 1, 0, 0, 0 };
    Quat attach2 = (other_rot.inv() * rot).normalize();
    float separation = (other_pos - pos).length() - 1.25f;

    grab.constraintEntity = PhysicsSystem::makeFixedJoint(
        ctx, e, grab_entity, attach1, attach2, r1, r2, separation);
}

// One lane per world, 128 wavefronts on the whole chip: the system costs what
// its longest chain of dependent loads costs (a cold miss is ~1.2 us), and read
// where it is used -- agent after agent, behind each `if` -- everything an agent
// may need was ~11 of them per agent.  Nothing the first agent's grab or release
// writes (its own GrabState, a joint row) is read for the second: both agents'
// handles, rows and their targets' poses are fetched up front, in three rounds.
inline void grabSystem(Engine &ctx, LevelState &)
{
    Sim &sim = ctx.data();

    Entity agent[consts::numAgents];
    Entity target[consts::numAgents];
    for (int32_t i = 0; i < consts::numAgents; i++) {
        agent[i] = sim.agents[i];
        target[i] = sim.grabTargets[i];
    }
    int32_t grab_action[consts::numAgents];
    GrabState *grab[consts::numAgents];
    Entity held[consts::numAgents];
    Vector3 pos[consts::numAgents];
    Quat rot[consts::numAgents];
    Vector3 target_pos[consts::numAgents];
    Quat target_rot[consts::numAgents];
    for (int32_t i = 0; i < consts::numAgents; i++) {
        // (no target: the agent's own row, read and not used)
        const Entity other = target[i] != Entity::none() ? target[i] : agent[i];
        grab_action[i] = ctx.get<Action>(agent[i]).grab;
        grab[i] = &ctx.get<GrabState>(agent[i]);
        held[i] = grab[i]->constraintEntity;
        pos[i] = ctx.get<Position>(agent[i]);
        rot[i] = ctx.get<Rotation>(agent[i]);
        target_pos[i] = ctx.get<Position>(other);
        target_rot[i] = ctx.get<Rotation>(other);
    }

    for (int32_t i = 0; i < consts::numAgents; i++) {
        if (grab_action[i] == 0) {
            continue;
        }
        if (held[i] != Entity::none()) {
            releaseGrab(ctx, held[i], *grab[i]);
            continue;
        }
        if (target[i] != Entity::none()) {
            attachGrabAt(ctx, agent[i], target[i], pos[i], rot[i], target_pos[i],
                         target_rot[i], *grab[i]);
        }
    }
}
#else
inline void grabSystem(Engine &ctx, LevelState &)
{
    Sim &sim = ctx.data();

    for (int32_t i = 0; i < consts::numAgents; i++) {
        Entity e = sim.agents[i];
        if (ctx.get<Action>(e).grab == 0) {
            continue;
        }

        GrabState &grab = ctx.get<GrabState>(e);
        if (grab.constraintEntity != Entity::none()) {
            releaseGrab(ctx, grab.constraintEntity, grab);
            continue;
        }

        Vector3 pos = ctx.get<Position>(e);
        Quat rot = ctx.get<Rotation>(e);

#ifdef SIM_WAVE_API
        Entity grab_entity = sim.grabTargets[i];
#else
        Vector3 reach = pos + rot.rotateVec(Vector3 { 0.f, 1.75f, 0.f });
        AABB reach_box {
            reach - Vector3 { 1.f, 1.f, 1.f },
            reach + Vector3 { 1.f, 1.f, 1.f },
        };

        Entity grab_entity = Entity::none();
        PhysicsSystem::findEntitiesWithinAABB(ctx, reach_box,
            [&](Entity other) {
                if (grab_entity != Entity::none()) {
                    return;
                }
                if (ctx.get<EntityType>(other) == EntityType::Cube) {
                    grab_entity = other;
                }
            });
#endif

        if (grab_entity != Entity::none()) {
            attachGrab(ctx, e, grab_entity, pos, rot, grab);
        }
    }
}
#endif

// agents stop dead every step: removes the need for drag
inline void agentZeroVelSystem(Engine &,
                               Velocity &vel,
                               Action &)
{
    vel.linear.x = 0.f;
    vel.linear.y = 0.f;
    vel.linear.z = fminf(vel.linear.z, 0.f);

    vel.angular = Vector3::zero();
}

inline void buttonSystem(Engine &ctx,
                         Position &pos,
                         ButtonState &state)
{
    const Sim &sim = ctx.data();

    bool pressed = false;
    for (int32_t i = 0; i < consts::numAgents; i++) {
        Vector3 agent_pos = ctx.get<Position>(sim.agents[i]);
        float dx = fabsf(agent_pos.x - pos.x);
        float dy = fabsf(agent_pos.y - pos.y);
        if (dx < consts::buttonWidth && dy < consts::buttonWidth) {
            pressed = true;
        }
    }

    state.isPressed = pressed ? 1 : 0;
}

inline void doorOpenSystem(Engine &ctx,
                           OpenState &open_state,
                           const DoorProperties &props)
{
    bool all_pressed = true;
    for (int32_t i = 0; i < props.numButtons; i++) {
        Entity button = props.buttons[i];
        all_pressed = all_pressed && ctx.get<ButtonState>(button).isPressed != 0;
    }

    if (all_pressed) {
        open_state.isOpen = 1;
    } else if (props.isPersistent == 0) {
        open_state.isOpen = 0;
    }
}

inline void setDoorPositionSystem(Engine &,
                                  Position &pos,
                                  OpenState &open_state)
{
    if (open_state.isOpen != 0) {
        if (pos.z > -4.5f) {
            pos.z += -consts::doorSpeed * consts::deltaT;
        }
    } else if (pos.z < 0.f) {
        pos.z += consts::doorSpeed * consts::deltaT;
    }

    if (pos.z >= 0.f) {
        pos.z = 0.f;
    }
}

#ifdef SIM_ROW_CHAIN
// Both door systems in one node: each only touches its own door's row (the
// button states doorOpenSystem reaches for are written by neither), so row i
// may run the second right behind the first -- one launch instead of two
// (madrona::mwhip::rowChain).  The portable sources
// and the CPU build keep the two nodes.
inline constexpr auto doorSystem = madrona::mwhip::rowChain<
    doorOpenSystem, setDoorPositionSystem, Engine,
    OpenState, DoorProperties, Position>;
#endif

inline void rewardSystem(Engine &,
                         Position &pos,
                         Progress &progress,
                         Reward &out_reward)
{
    float reward_pos = fminf(pos.y, consts::worldLength * 2.f);

    float old_max_y = progress.maxY;
    float new_progress = reward_pos - old_max_y;

    float reward;
    if (new_progress > 0.f) {
        reward = new_progress * consts::rewardPerDist;
        progress.maxY = reward_pos;
    } else {
        reward = consts::slackReward;
    }

    out_reward.v = reward;
}

inline void stepTrackerSystem(Engine &,
                              StepsRemaining &steps_remaining,
                              Done &done)
{
    int32_t num_remaining = (int32_t)--steps_remaining.t;
    if (num_remaining == consts::episodeLen - 1) {
        done.v = 0;
    } else if (num_remaining == 0) {
        done.v = 1;
    }
}

#ifdef SIM_WAVE_API
// The reset of a world on the GPU backends: 64 lanes per world
// (CustomParallelForNode<..., 64, 1>), lane i destroys / creates / fills in
// entity i instead of one lane doing all of them in sequence.  What comes out is
// what cleanupWorld() + initWorld() above leave, bit for bit:
//   * entity ids: Context::destroyEntityOrdered / makeEntityOrdered hand ids
//     back and out in lane order, and the lanes are laid out in the order of
//     the loops above (render variant: every body is followed by its render
//     entity, as makeEntityRenderable creates it inside setupRigidBody);
//   * rows: a table's rows in lane order;
//   * BVH leaves: initWorld() registers floor, borders, agents, then per room
//     walls, door, cubes -- leaf = position in that sequence (reserveLeafAt);
//   * random numbers: sample i of an episode is split_i(episode key, i) and the
//     loops draw a fixed number per entity:
//       agent a:  3 a + { x, y, heading }
//       room r:   base = 3 numAgents + r * (2 numButtons + 3 + 2 numCubes)
//         button b: base + 2 b + { x, y };  door gap: base + 2 numButtons;
//         door: ... + 1 numButtons, + 2 isPersistent;  cube c: ... + 3 + 2 c + { x, y }
static inline void resetWorldWave(Engine &ctx)
{
    Sim &sim = ctx.data();
    LevelState &level = ctx.singleton<LevelState>();
    broadphase::BVH &bvh = ctx.singleton<broadphase::BVH>();

#ifdef ESCPHYS_RENDER
    constexpr int32_t stride = 2;       // body, its render entity, body, ...
#else
    constexpr int32_t stride = 1;
#endif
    constexpr int32_t per_room =
        consts::numButtonsPerRoom + 2 + 1 + consts::numCubesPerRoom;
    constexpr int32_t num_bodies = consts::numRooms * per_room;
    constexpr int32_t num_persistent = 1 + consts::numBorderWalls + consts::numAgents;
    constexpr int32_t leaves_per_room = 2 + 1 + consts::numCubesPerRoom;
    constexpr uint32_t draws_per_room =
        2u * consts::numButtonsPerRoom + 3u + 2u * consts::numCubesPerRoom;
    constexpr uint32_t num_draws = 3u * consts::numAgents +
        (uint32_t)consts::numRooms * draws_per_room;
    static_assert(num_bodies * stride <= 64 && num_persistent <= 64 &&
                  2 + num_bodies * stride <= 64);

    const int32_t lane = (int32_t)(threadIdx.x % 64);

    // ---- cleanupWorld() ----
    // the agents' grab joints first
    {
        Entity joint = Entity::none();
        if (lane < consts::numAgents) {
            GrabState &grab = ctx.get<GrabState>(sim.agents[lane]);
            joint = grab.constraintEntity;
            grab.constraintEntity = Entity::none();
        }
        ctx.destroyEntityOrdered(joint, joint != Entity::none());
    }
    // then per room: (render variant: the render entities of) cubes, walls,
    // door, buttons; then those bodies themselves
    {
        constexpr int32_t group = per_room * stride;
        const int32_t r = lane / group;
        const int32_t in_room = lane % group;
        const int32_t g = in_room % per_room;
        const bool want = lane < consts::numRooms * group;
        Entity e = Entity::none();
        if (want) {
            const Room &room = level.rooms[r];
            e = g < consts::numCubesPerRoom ? room.cubes[g] :
                (g < consts::numCubesPerRoom + 2 ? room.walls[g - consts::numCubesPerRoom] :
                 (g == consts::numCubesPerRoom + 2 ? room.door :
                  room.buttons[g - consts::numCubesPerRoom - 3]));
#ifdef ESCPHYS_RENDER
            if (in_room < per_room) {
                e = ctx.get<render::Renderable>(e).renderEntity;
            }
#endif
        }
        ctx.destroyEntityOrdered(e, want);
    }

    // ---- initWorld() ----
    if (lane == 0) {
        // every body re-registers with an emptied BVH
        PhysicsSystem::reset(ctx);
        bvh.setNumLeaves(num_persistent + consts::numRooms * leaves_per_room);
    }

    const RandKey episode = rand::split_i(sim.initRandKey, sim.curWorldEpisode);
    auto key = [&](uint32_t i) { return rand::split_i(episode, i); };
    auto in_range = [&](uint32_t i, float lo, float hi) {
        return lo + rand::sampleUniform(key(i)) * (hi - lo);
    };
    const float half_width = consts::worldWidth / 2.f;

    // resetPersistentEntities(): floor, borders, agents re-register (leaves 0 ..),
    // the agents get a new pose
    if (lane < num_persistent) {
        const Entity e = lane == 0 ? sim.floorPlane :
            (lane <= consts::numBorderWalls ? sim.borders[lane - 1] :
             sim.agents[lane - 1 - consts::numBorderWalls]);
        ctx.get<broadphase::LeafID>(e) =
            bvh.reserveLeafAt(lane, e, ctx.get<ObjectID>(e));

        if (lane > consts::numBorderWalls) {
            const int32_t i = lane - 1 - consts::numBorderWalls;
            const uint32_t base = 3u * (uint32_t)i;
            Entity agent = e;

            float x_lo = i == 0 ? -half_width + 2.f : 1.5f;
            float x_hi = i == 0 ? -1.5f : half_width - 2.f;
            Vector3 pos {
                in_range(base, x_lo, x_hi),
                in_range(base + 1u, 1.5f, 3.f),
                1.f,
            };
            int32_t heading = rand::sampleI32(key(base + 2u), 0, 8);
            float c = kMoveCos[heading];
            float sn = kMoveSin[heading];
            float ch = sqrtf((1.f + c) * 0.5f);
            float sh = sqrtf((1.f - c) * 0.5f);
            if (sn < 0.f) sh = -sh;
            Quat rot = Quat { ch, 0.f, 0.f, sh }.normalize();

            ctx.get<Position>(agent) = pos;
            ctx.get<Rotation>(agent) = rot;
            ctx.get<Velocity>(agent) = Velocity { Vector3::zero(), Vector3::zero() };
            ctx.get<ExternalForce>(agent) = Vector3::zero();
            ctx.get<ExternalTorque>(agent) = Vector3::zero();
            ctx.get<Action>(agent) = Action { 0, 0, 0, 0 };
            ctx.get<Progress>(agent).maxY = pos.y;
            ctx.get<StepsRemaining>(agent).t = consts::episodeLen;
            ctx.get<GrabState>(agent).constraintEntity = Entity::none();
            ctx.get<Reward>(agent).v = 0.f;
            ctx.get<Done>(agent).v = 0;
        }
    }

    // generateLevel(): per room buttons, walls, door, cubes
    const int32_t body = lane / stride;
    const bool on_level = body < num_bodies;
    const bool is_render = stride == 2 && (lane & 1) != 0;
    const int32_t r = body / per_room;
    const int32_t k = body % per_room;
    const bool is_button = k < consts::numButtonsPerRoom;
    const bool is_wall = !is_button && k < consts::numButtonsPerRoom + 2;
    const bool is_door = k == consts::numButtonsPerRoom + 2;

    uint32_t archetype = is_button ? TypeTracker::typeID<ButtonEntity>() :
        (is_door ? TypeTracker::typeID<DoorEntity>() :
                   TypeTracker::typeID<PhysicsEntity>());
#ifdef ESCPHYS_RENDER
    if (is_render) {
        archetype = TypeTracker::typeID<render::RenderableArchetype>();
    }
#endif
    const Entity e = ctx.makeEntityOrdered(archetype, on_level);

    // (the door wants its room's buttons, a body its render entity)
    Entity room_buttons[consts::numButtonsPerRoom];
    for (int32_t b = 0; b < consts::numButtonsPerRoom; b++) {
        const int32_t src = ((on_level ? r : 0) * per_room + b) * stride;
        room_buttons[b].gen = (uint32_t)__shfl((int)e.gen, src, 64);
        room_buttons[b].id = __shfl(e.id, src, 64);
    }
    Entity render_entity;
    render_entity.gen = (uint32_t)__shfl((int)e.gen, (lane + 1) & 63, 64);
    render_entity.id = __shfl(e.id, (lane + 1) & 63, 64);
    (void)render_entity;

    if (on_level && is_render) {
#ifdef ESCPHYS_RENDER
        // (RenderingSystem::makeEntityRenderable's part on the render entity)
        render::InstanceData &inst = ctx.get<render::InstanceData>(e);
        inst.matID = render::MaterialOverride::UseDefaultMaterial;
        inst.color = 0;
#endif
    } else if (on_level) {
        Room &room = level.rooms[r];
        const float y_min = (float)r * consts::roomLength;
        const float y_max = y_min + consts::roomLength;
        const uint32_t base =
            3u * consts::numAgents + (uint32_t)r * draws_per_room;
        const float door_x = in_range(base + 2u * consts::numButtonsPerRoom,
                                      -half_width + 4.f, half_width - 4.f);
        const int32_t first_leaf = num_persistent + r * leaves_per_room;

        // setupRigidBody() with the leaf this body has in the sequence
        auto fill_body = [&](Vector3 pos, SimObject obj, EntityType type,
                             ResponseType response, Diag3x3 scale, int32_t leaf) {
            ObjectID obj_id { (int32_t)obj };
            ctx.get<Position>(e) = pos;
            ctx.get<Rotation>(e) = Quat { 1, 0, 0, 0 };
            ctx.get<Scale>(e) = scale;
            ctx.get<ObjectID>(e) = obj_id;
            ctx.get<ResponseType>(e) = response;
            ctx.get<Velocity>(e) = Velocity { Vector3::zero(), Vector3::zero() };
            ctx.get<ExternalForce>(e) = Vector3::zero();
            ctx.get<ExternalTorque>(e) = Vector3::zero();
            ctx.get<broadphase::LeafID>(e) = bvh.reserveLeafAt(leaf, e, obj_id);
            ctx.get<EntityType>(e) = type;
            ESCPHYS_IF_RENDER(
                ctx.get<render::Renderable>(e).renderEntity = render_entity;)
        };

        if (is_button) {
            const uint32_t at = base + 2u * (uint32_t)k;
            Vector3 pos {
                in_range(at, -half_width + 2.f, half_width - 2.f),
                in_range(at + 1u, y_min + 2.f, y_max - 3.f),
                0.f,
            };
            ctx.get<Position>(e) = pos;
            ctx.get<Rotation>(e) = Quat { 1, 0, 0, 0 };
            ctx.get<Scale>(e) = Diag3x3 {
                consts::buttonWidth, consts::buttonWidth, 0.2f,
            };
            ctx.get<ObjectID>(e) = ObjectID { (int32_t)SimObject::Button };
            ctx.get<ButtonState>(e).isPressed = 0;
            ctx.get<EntityType>(e) = EntityType::Button;
            ESCPHYS_IF_RENDER(
                ctx.get<render::Renderable>(e).renderEntity = render_entity;)
            room.buttons[k] = e;
        } else if (is_wall) {
            const int32_t side = k - consts::numButtonsPerRoom;
            // makeWall()
            const float x_min = side == 0 ? -half_width :
                                            door_x + consts::doorWidth * 0.5f;
            const float x_max = side == 0 ? door_x - consts::doorWidth * 0.5f :
                                            half_width;
            fill_body(
                Vector3 { (x_min + x_max) * 0.5f, y_max, consts::wallHeight * 0.5f },
                SimObject::Wall, EntityType::Wall, ResponseType::Static,
                Diag3x3 { x_max - x_min, consts::wallWidth, consts::wallHeight },
                first_leaf + side);
            room.walls[side] = e;
        } else if (is_door) {
            const uint32_t at = base + 2u * consts::numButtonsPerRoom;
            fill_body(Vector3 { door_x, y_max, 0.f }, SimObject::Door,
                EntityType::Door, ResponseType::Static,
                Diag3x3 { consts::doorWidth * 0.8f, consts::wallWidth,
                          2.f * consts::wallHeight },
                first_leaf + 2);
            ctx.get<OpenState>(e).isOpen = 0;
            DoorProperties &props = ctx.get<DoorProperties>(e);
            for (int32_t b = 0; b < 4; b++) {
                props.buttons[b] = b < consts::numButtonsPerRoom ?
                    room_buttons[b] : Entity::none();
            }
            props.numButtons = 1 + rand::sampleI32(key(at + 1u), 0,
                                                   consts::numButtonsPerRoom);
            props.isPersistent = rand::sampleBool(key(at + 2u)) ? 1 : 0;
            room.door = e;
        } else {
            const int32_t c = k - consts::numButtonsPerRoom - 3;
            const uint32_t at =
                base + 2u * consts::numButtonsPerRoom + 3u + 2u * (uint32_t)c;
            Vector3 pos {
                in_range(at, -half_width + 1.5f, half_width - 1.5f),
                in_range(at + 1u, y_min + 4.f, y_max - 2.5f),
                consts::cubeSize * 0.5f,
            };
            fill_body(pos, SimObject::Cube, EntityType::Cube,
                ResponseType::Dynamic,
                Diag3x3 { consts::cubeSize, consts::cubeSize, consts::cubeSize },
                first_leaf + 3 + c);
            room.cubes[c] = e;
        }
    }

    if (lane == 0) {
        // what `sim.rng = rng` leaves after the sequential draws
        struct RNGState { RandKey k; uint32_t count; };
        static_assert(sizeof(RNGState) == sizeof(RNG));
        RNGState state { episode, num_draws };
        memcpy(&sim.rng, &state, sizeof(RNG));
        sim.curWorldEpisode += 1;
    }
}
#endif

inline void resetSystem(Engine &ctx, WorldReset &reset)
{
    Sim &sim = ctx.data();

    int32_t should_reset = reset.reset;

    for (int32_t i = 0; i < consts::numAgents; i++) {
        if (ctx.get<Done>(sim.agents[i]).v != 0) {
            should_reset = 1;
        }
    }

#ifdef SIM_WAVE_API
    // 64 lanes per world: lane 0 advances the reset stream, everybody learns
    // the outcome
    int32_t auto_reset = 0;
    if (sim.autoResetDenom != 0 && threadIdx.x % 64 == 0) {
        auto_reset =
            sim.resetRng.sampleI32(0, (int32_t)sim.autoResetDenom) == 0 ? 1 : 0;
    }
    if (__shfl(auto_reset, 0, 64) != 0) {
        should_reset = 1;
    }

    if (should_reset != 0) {
        if (threadIdx.x % 64 == 0) {
            reset.reset = 0;
        }
        resetWorldWave(ctx);
    }
#else
    if (sim.autoResetDenom != 0) {
        if (sim.resetRng.sampleI32(0, (int32_t)sim.autoResetDenom) == 0) {
            should_reset = 1;
        }
    }

    if (should_reset != 0) {
        reset.reset = 0;
        cleanupWorld(ctx);
        initWorld(ctx);
    }
#endif
}

inline void collectObservationsSystem(Engine &ctx,
                                      Position &pos,
                                      Rotation &rot,
                                      const Progress &progress,
                                      const GrabState &grab,
                                      const OtherAgents &other_agents,
                                      SelfObservation &self_obs,
                                      PartnerObservation &partner_obs,
                                      RoomEntityObservations &room_ent_obs,
                                      DoorObservation &door_obs)
{
    const LevelState &level = ctx.singleton<LevelState>();

    int32_t room_idx = (int32_t)(pos.y / consts::roomLength);
    if (room_idx < 0) room_idx = 0;
    if (room_idx > consts::numRooms - 1) room_idx = consts::numRooms - 1;
    const Room &room = level.rooms[room_idx];

    const float room_y_min = (float)room_idx * consts::roomLength;

    self_obs.roomX = pos.x / (consts::worldWidth / 2.f);
    self_obs.roomY = (pos.y - room_y_min) / consts::roomLength;
    self_obs.globalX = pos.x / consts::worldWidth;
    self_obs.globalY = pos.y / consts::worldLength;
    self_obs.globalZ = pos.z / 10.f;
    self_obs.maxY = progress.maxY / consts::worldLength;
    self_obs.facing = rot.z;
    self_obs.isGrabbing =
        grab.constraintEntity != Entity::none() ? 1.f : 0.f;

    Quat to_view = rot.inv();

    {
        Entity other = other_agents.e[0];
        Vector3 other_pos = ctx.get<Position>(other);
        Vector3 rel = to_view.rotateVec(other_pos - pos);
        partner_obs.dx = rel.x / consts::worldLength;
        partner_obs.dy = rel.y / consts::worldLength;
        partner_obs.isGrabbing =
            ctx.get<GrabState>(other).constraintEntity != Entity::none() ?
                1.f : 0.f;
    }

    int32_t out = 0;
    for (int32_t c = 0; c < consts::numCubesPerRoom; c++) {
        Entity e = room.cubes[c];
        Vector3 rel = to_view.rotateVec(ctx.get<Position>(e) - pos);
        room_ent_obs.obs[out++] = EntityObservation {
            rel.x / consts::worldLength, rel.y / consts::worldLength,
            (float)ctx.get<EntityType>(e) / (float)EntityType::NumTypes,
        };
    }
    for (int32_t b = 0; b < consts::numButtonsPerRoom; b++) {
        Entity e = room.buttons[b];
        Vector3 rel = to_view.rotateVec(ctx.get<Position>(e) - pos);
        room_ent_obs.obs[out++] = EntityObservation {
            rel.x / consts::worldLength, rel.y / consts::worldLength,
            (float)ctx.get<EntityType>(e) / (float)EntityType::NumTypes,
        };
    }
    room_ent_obs.obs[out] = EntityObservation { 0.f, 0.f, 0.f };

    {
        Entity door = room.door;
        Vector3 rel = to_view.rotateVec(ctx.get<Position>(door) - pos);
        door_obs.dx = rel.x / consts::worldLength;
        door_obs.dy = rel.y / consts::worldLength;
        door_obs.isOpen = ctx.get<OpenState>(door).isOpen != 0 ? 1.f : 0.f;
    }
}

// 30 rays per agent through the world's BVH (hulls + floor plane): depth and
// the type of what was hit, like the Escape Room's lidar.  On the GPU backends
// the node runs 32 threads per agent, one ray each (the reference simulators'
// convention: CustomParallelForNode<..., 32, 1, ...> + threadIdx.x % 32).
inline void lidarSystem(Engine &ctx,
                        Entity e,
                        Lidar &lidar)
{
    Vector3 pos = ctx.get<Position>(e);
    Quat rot = ctx.get<Rotation>(e);
    broadphase::BVH &bvh = ctx.singleton<broadphase::BVH>();

    // from inside the agent's own hull (only back faces are met: a miss)
    Vector3 ray_o = pos + 0.5f * math::up;

#ifdef SIM_WAVE_API
    broadphase::BVH::RayGroupScratch *ray_scratch = broadphase::rayGroupScratch();
#endif
    auto traceRay = [&](int32_t i) {
        Vector3 ray_d = rot.rotateVec(
            Vector3 { kLidarCos[i], kLidarSin[i], 0.f }).normalize();

        float hit_t;
        Vector3 hit_normal;
#ifdef SIM_WAVE_API
        // (the agent's 30 rays start at the same point: this backend's
        // shared-origin flavour, same hits)
        Entity hit_entity = bvh.traceRayShared(ray_scratch, ray_o, ray_d,
                                               &hit_t, &hit_normal, 200.f);
#else
        Entity hit_entity =
            bvh.traceRay(ray_o, ray_d, &hit_t, &hit_normal, 200.f);
#endif

        if (hit_entity == Entity::none()) {
            lidar.samples[i] = LidarSample { 0.f, 0.f };
        } else {
            EntityType hit_type = ctx.get<EntityType>(hit_entity);
            lidar.samples[i] = LidarSample {
                hit_t / 200.f,
                (float)hit_type / (float)EntityType::NumTypes,
            };
        }
    };

#ifdef MADRONA_GPU_MODE
    int32_t idx = (int32_t)(threadIdx.x % 32);
    if (idx < consts::numLidarSamples) {
        traceRay(idx);
    }
#else
    for (int32_t i = 0; i < consts::numLidarSamples; i++) {
        traceRay(i);
    }
#endif
}

#ifdef MADRONA_GPU_MODE
// ---------------------------------------------------------------------------
// What each system reads and writes per row (see
// sims/escape_room/sim.cpp and madrona::mwhip::systemIO, taskgraph.inl).
// bodiesPerWorld: the leaf boxes of the world's BVH a ray / box query walks
// over (per-world data, read once per row; the object manager's meshes are the
// shared read-only tables SURVEY §8d excludes).
// ---------------------------------------------------------------------------
}
#define ESCPHYS_SYSTEM_IO(fn, ...) \
    template <> inline constexpr madrona::mwhip::SystemIOBytes \
        madrona::mwhip::systemIO<escphys::fn> = \
            madrona::mwhip::declareIO<__VA_ARGS__>()
namespace escphys_io {
using madrona::Entity;
using madrona::math::AABB;
using madrona::mwhip::Reads;
using madrona::mwhip::Times;
using madrona::mwhip::Writes;
// floor + 4 borders + 2 agents + 3 x (2 walls + door + 4 cubes)
inline constexpr uint32_t bodiesPerWorld = 28;
}
ESCPHYS_SYSTEM_IO(movementSystem,
    escphys_io::Reads<escphys::Action, escphys::Rotation>,
    escphys_io::Writes<escphys::ExternalForce, escphys::ExternalTorque>);
#ifdef SIM_WAVE_API
// (per world) both agents' action + grab state + pose, the leaf boxes the box
// queries test, the type of the entity found
ESCPHYS_SYSTEM_IO(grabQuerySystem,
    escphys_io::Reads<
        escphys_io::Times<escphys::Action, escphys::consts::numAgents>,
        escphys_io::Times<escphys::GrabState, escphys::consts::numAgents>,
        escphys_io::Times<escphys::Position, escphys::consts::numAgents>,
        escphys_io::Times<escphys::Rotation, escphys::consts::numAgents>,
        escphys_io::Times<escphys_io::AABB, escphys_io::bodiesPerWorld>,
        escphys_io::Times<escphys::EntityType, escphys::consts::numAgents>>,
    escphys_io::Writes<escphys_io::Times<escphys_io::Entity,
                                         escphys::consts::numAgents>>);
#endif
// (per world) the same agents' rows; a joint row created or destroyed
ESCPHYS_SYSTEM_IO(grabSystem,
    escphys_io::Reads<
        escphys_io::Times<escphys::Action, escphys::consts::numAgents>,
        escphys_io::Times<escphys::GrabState, escphys::consts::numAgents>,
        escphys_io::Times<escphys::Position, escphys::consts::numAgents>,
        escphys_io::Times<escphys::Rotation, escphys::consts::numAgents>,
        escphys_io::Times<escphys_io::Entity, escphys::consts::numAgents>>,
    escphys_io::Writes<escphys::GrabState, madrona::phys::JointConstraint>);
ESCPHYS_SYSTEM_IO(agentZeroVelSystem,
    escphys_io::Reads<>, escphys_io::Writes<escphys::Velocity>);
ESCPHYS_SYSTEM_IO(buttonSystem,
    escphys_io::Reads<escphys::Position,
                      escphys_io::Times<escphys::Position, escphys::consts::numAgents>>,
    escphys_io::Writes<escphys::ButtonState>);
ESCPHYS_SYSTEM_IO(doorOpenSystem,
    escphys_io::Reads<escphys::DoorProperties,
                      escphys_io::Times<escphys::ButtonState,
                                        escphys::consts::numButtonsPerRoom>>,
    escphys_io::Writes<escphys::OpenState>);
ESCPHYS_SYSTEM_IO(setDoorPositionSystem,
    escphys_io::Reads<escphys::Position, escphys::OpenState>,
    escphys_io::Writes<escphys::Position>);
#ifdef SIM_ROW_CHAIN
// (the chain of the two: OpenState is read -- a persistent door keeps it -- and
// written once, not written, stored, and read back)
ESCPHYS_SYSTEM_IO(doorSystem,
    escphys_io::Reads<escphys::DoorProperties,
                escphys_io::Times<escphys::ButtonState, escphys::consts::numButtonsPerRoom>,
                escphys::OpenState, escphys::Position>,
    escphys_io::Writes<escphys::OpenState, escphys::Position>);
#endif
ESCPHYS_SYSTEM_IO(rewardSystem,
    escphys_io::Reads<escphys::Position, escphys::Progress>,
    escphys_io::Writes<escphys::Progress, escphys::Reward>);
ESCPHYS_SYSTEM_IO(stepTrackerSystem,
    escphys_io::Reads<escphys::StepsRemaining>,
    escphys_io::Writes<escphys::StepsRemaining, escphys::Done>);
// a world that does not reset (the work of a reset is level generation)
ESCPHYS_SYSTEM_IO(resetSystem,
    escphys_io::Reads<escphys::WorldReset,
                      escphys_io::Times<escphys::Done, escphys::consts::numAgents>>,
    escphys_io::Writes<>);
ESCPHYS_SYSTEM_IO(collectObservationsSystem,
    escphys_io::Reads<escphys::Position, escphys::Rotation, escphys::Progress,
                      escphys::GrabState, escphys::OtherAgents, escphys::Room,
                      escphys::Position, escphys::GrabState,
                      escphys_io::Times<escphys::Position,
                          escphys::consts::numCubesPerRoom +
                          escphys::consts::numButtonsPerRoom>,
                      escphys_io::Times<escphys::EntityType,
                          escphys::consts::numCubesPerRoom +
                          escphys::consts::numButtonsPerRoom>,
                      escphys::Position, escphys::OpenState>,
    escphys_io::Writes<escphys::SelfObservation, escphys::PartnerObservation,
                       escphys::RoomEntityObservations, escphys::DoorObservation>);
// per agent (30 rays share the origin): its handle and pose, the world's leaf
// boxes, the type of what each ray hit
ESCPHYS_SYSTEM_IO(lidarSystem,
    escphys_io::Reads<escphys_io::Entity, escphys::Position, escphys::Rotation,
                      escphys_io::Times<escphys_io::AABB, escphys_io::bodiesPerWorld>,
                      escphys_io::Times<escphys::EntityType,
                                        escphys::consts::numLidarSamples>>,
    escphys_io::Writes<escphys::Lidar>);
#undef ESCPHYS_SYSTEM_IO
// (131 registers with the leaf tests' loads batched -- one too many for the
// four wavefronts per SIMD the ray systems want:
// profiles/r03_lidar_occupancy_variants.jsonl)
template <> inline constexpr unsigned
    madrona::mwhip::systemWavesPerSIMD<escphys::lidarSystem> = 4;
#ifdef SIM_WAVE_API
// (131 registers as well; a wavefront per world of dependent loads: 50 -> 44 us;
// five / six / eight wavefronts per SIMD: 44.0 / 45.5 / 50.4 us with 27 / 109 /
// 439 spilled dwords, profiles/r06_grabq_variants.jsonl)
template <> inline constexpr unsigned
    madrona::mwhip::systemWavesPerSIMD<escphys::grabQuerySystem> = 4;
#endif
namespace escphys {
#endif

void Sim::setupTasks(TaskGraphManager &taskgraph_mgr, const Config &)
{
    TaskGraphBuilder &builder = taskgraph_mgr.init(0);

    auto move_sys = builder.addToGraph<ParallelForNode<Engine,
        movementSystem,
            Action,
            Rotation,
            ExternalForce,
            ExternalTorque
        >>({});

    auto broadphase_setup_sys =
        PhysicsSystem::setupBroadphaseTasks(builder, {move_sys});

#ifdef SIM_WAVE_API
    // 64 lanes per world: the grab queries test a BVH leaf per lane
    auto grab_query_sys = builder.addToGraph<CustomParallelForNode<Engine,
        grabQuerySystem, 64, 1,
            LevelState
        >>({broadphase_setup_sys});
#else
    auto grab_query_sys = broadphase_setup_sys;
#endif
    auto grab_sys = builder.addToGraph<ParallelForNode<Engine,
        grabSystem,
            LevelState
        >>({grab_query_sys});

    auto substep_sys = PhysicsSystem::setupPhysicsStepTasks(builder,
        {grab_sys}, consts::numPhysicsSubsteps);

    // (the real dependencies, not a chain -- sims/escape_room/sim.cpp: zeroing
    // the agents' velocities, the button -> door chain, the reward and the step
    // counter all follow the physics step and touch disjoint components)
    auto phys_done =
        PhysicsSystem::setupCleanupTasks(builder, {substep_sys});

    auto agent_zero_vel = builder.addToGraph<ParallelForNode<Engine,
        agentZeroVelSystem,
            Velocity,
            Action
        >>({phys_done});

    auto button_sys = builder.addToGraph<ParallelForNode<Engine,
        buttonSystem,
            Position,
            ButtonState
        >>({phys_done});

    auto reward_sys = builder.addToGraph<ParallelForNode<Engine,
        rewardSystem,
            Position,
            Progress,
            Reward
        >>({phys_done});

    auto done_sys = builder.addToGraph<ParallelForNode<Engine,
        stepTrackerSystem,
            StepsRemaining,
            Done
        >>({phys_done});

    // (the door chain is registered behind the nodes that share the button
    // system's dependency: the executor runs those in one launch)
#ifdef SIM_ROW_CHAIN
    auto set_door_pos_sys = builder.addToGraph<ParallelForNode<Engine,
        doorSystem,
            OpenState,
            DoorProperties,
            Position
        >>({button_sys});
#else
    auto door_open_sys = builder.addToGraph<ParallelForNode<Engine,
        doorOpenSystem,
            OpenState,
            DoorProperties
        >>({button_sys});

    auto set_door_pos_sys = builder.addToGraph<ParallelForNode<Engine,
        setDoorPositionSystem,
            Position,
            OpenState
        >>({door_open_sys});
#endif

#ifdef SIM_WAVE_API
    // 64 lanes per world: lane i resets entity i (resetWorldWave)
    auto reset_sys = builder.addToGraph<CustomParallelForNode<Engine,
        resetSystem, 64, 1,
#else
    auto reset_sys = builder.addToGraph<ParallelForNode<Engine,
        resetSystem,
#endif
            WorldReset
        >>({agent_zero_vel, set_door_pos_sys, reward_sys, done_sys});

#ifdef MADRONA_GPU_MODE
    auto recycle_sys = builder.addToGraph<RecycleEntitiesNode>({reset_sys});
    auto post_reset = recycle_sys;
#else
    auto post_reset = reset_sys;
#endif

    auto compact_cubes = builder.addToGraph<
        CompactArchetypeNode<PhysicsEntity>>({post_reset});
    auto compact_doors = builder.addToGraph<
        CompactArchetypeNode<DoorEntity>>({compact_cubes});
    auto compact_buttons = builder.addToGraph<
        CompactArchetypeNode<ButtonEntity>>({compact_doors});

    // a reset world needs its BVH before observations / the next step
    auto post_reset_broadphase =
        PhysicsSystem::setupBroadphaseTasks(builder, {compact_buttons});

    auto collect_obs = builder.addToGraph<ParallelForNode<Engine,
        collectObservationsSystem,
            Position,
            Rotation,
            Progress,
            GrabState,
            OtherAgents,
            SelfObservation,
            PartnerObservation,
            RoomEntityObservations,
            DoorObservation
        >>({post_reset_broadphase});

#ifdef MADRONA_GPU_MODE
    auto lidar = builder.addToGraph<CustomParallelForNode<Engine,
        lidarSystem, 32, 1,
#else
    auto lidar = builder.addToGraph<ParallelForNode<Engine,
        lidarSystem,
#endif
            Entity,
            Lidar
        >>({post_reset_broadphase});      // (beside the observations: both only read)

    (void)collect_obs;
    (void)lidar;

#ifdef ESCPHYS_RENDER
    // instance / view records, Morton order, world grouping for the ray caster
    RenderingSystem::setupTasks(builder, {collect_obs, lidar}, false);
#endif
}

Sim::Sim(Engine &ctx, const Config &cfg, const WorldInit &)
    : WorldBase(ctx)
{
    uint32_t global_world = cfg.worldBase + (uint32_t)ctx.worldID().idx;

    initRandKey = rand::split_i(rand::initKey(cfg.seed), global_world);
    resetRng = RNG(rand::split_i(initRandKey, 0x7E5E7u));
    curWorldEpisode = 0;
    autoResetDenom = cfg.autoResetDenom;

    ctx.singleton<WorldReset>().reset = 0;

    PhysicsSystem::init(ctx, cfg.rigidBodyObjMgr, consts::deltaT,
        consts::numPhysicsSubsteps, -9.8f * math::up,
        consts::maxRigidBodies);

#ifdef ESCPHYS_RENDER
    RenderingSystem::init(ctx, cfg.bridge);
#endif

    createPersistentEntities(ctx);
    initWorld(ctx);

#ifdef ESCPHYS_RENDER
    Entity sun = ctx.makeEntity<SunEntity>();
    ctx.get<Position>(sun) = Vector3 { 0.f, 0.f, 30.f };
    ctx.get<render::LightDescDirection>(sun) = render::LightDescDirection(
        Vector3 { 0.3713907f, 0.5570860f, -0.7427814f });
    ctx.get<render::LightDescType>(sun).type = render::LightDesc::Directional;
    ctx.get<render::LightDescShadow>(sun).castShadow = cfg.sunCastsShadows != 0u;
    ctx.get<render::LightDescCutoffAngle>(sun).cutoff = -1.f;
    ctx.get<render::LightDescIntensity>(sun).intensity = 1.f;
    ctx.get<render::LightDescActive>(sun).active = true;
    RenderingSystem::makeEntityLightCarrier(ctx, sun);
#endif
}

#ifdef MADRONA_GPU_MODE
MADRONA_BUILD_MWGPU_ENTRY(Engine, Sim, Sim::Config, Sim::WorldInit);
#endif

}

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

using namespace madrona;
using namespace madrona::math;
using namespace madrona::phys;

namespace hideseek {

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

// rotation about z by k * 90 degrees
static constexpr float kQuarterW[4] = { 1.f, 0.70710678f, 0.f, 0.70710678f };
static constexpr float kQuarterZ[4] = { 0.f, 0.70710678f, 1.f, -0.70710678f };

void Sim::registerTypes(ECSRegistry &registry, const Config &)
{
    base::registerTypes(registry);
    PhysicsSystem::registerTypes(registry);

    registry.registerComponent<Action>();
    registry.registerComponent<Reward>();
    registry.registerComponent<Done>();
    registry.registerComponent<SelfObservation>();
    registry.registerComponent<AgentObservations>();
    registry.registerComponent<BoxObservations>();
    registry.registerComponent<RampObservations>();
    registry.registerComponent<Lidar>();
    registry.registerComponent<StepsRemaining>();
    registry.registerComponent<OtherAgents>();
    registry.registerComponent<Visibility>();
    registry.registerComponent<EntityType>();
    registry.registerComponent<LockState>();

    registry.registerSingleton<WorldReset>();
    registry.registerSingleton<LevelState>();

    registry.registerArchetype<Agent>();
    registry.registerArchetype<MovableObject>();
    registry.registerArchetype<StaticObject>();

    registry.exportSingleton<WorldReset>((uint32_t)ExportID::Reset);
    registry.exportColumn<Agent, Action>((uint32_t)ExportID::Action);
    registry.exportColumn<Agent, Reward>((uint32_t)ExportID::Reward);
    registry.exportColumn<Agent, Done>((uint32_t)ExportID::Done);
    registry.exportColumn<Agent, SelfObservation>(
        (uint32_t)ExportID::SelfObservation);
    registry.exportColumn<Agent, AgentObservations>(
        (uint32_t)ExportID::AgentObservations);
    registry.exportColumn<Agent, BoxObservations>(
        (uint32_t)ExportID::BoxObservations);
    registry.exportColumn<Agent, RampObservations>(
        (uint32_t)ExportID::RampObservations);
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
}

static inline void registerRigidBodyEntity(Engine &ctx, Entity e)
{
    ctx.get<broadphase::LeafID>(e) =
        PhysicsSystem::registerEntity(ctx, e, ctx.get<ObjectID>(e));
}

static inline float cellCenter(int32_t idx)
{
    return ((float)idx + 0.5f) * consts::cellSize - consts::arenaSize * 0.5f;
}

// Objects go to distinct cells of the board (a partial shuffle of the cell
// list), walls onto the lines between cells.
static void generateLevel(Engine &ctx, RNG &rng)
{
    Sim &sim = ctx.data();
    LevelState &level = ctx.singleton<LevelState>();

    constexpr int32_t num_cells = consts::gridDim * consts::gridDim;
    constexpr int32_t num_placed = consts::numAgents + consts::numMovable;
    int32_t cells[num_cells];
    for (int32_t i = 0; i < num_cells; i++) {
        cells[i] = i;
    }
    for (int32_t i = 0; i < num_placed; i++) {
        int32_t j = i + rng.sampleI32(0, num_cells - i);
        int32_t tmp = cells[i];
        cells[i] = cells[j];
        cells[j] = tmp;
    }

    auto cellPos = [&](int32_t slot, float z) {
        int32_t cell = cells[slot];
        return Vector3 {
            cellCenter(cell % consts::gridDim) + randInRange(rng, -0.25f, 0.25f),
            cellCenter(cell / consts::gridDim) + randInRange(rng, -0.25f, 0.25f),
            z,
        };
    };

    int32_t slot = 0;
    for (int32_t i = 0; i < consts::numAgents; i++) {
        Entity agent = sim.agents[i];
        Vector3 pos = cellPos(slot++, 1.f);

        // heading: rotation about z by one of the 8 move angles; half-angle
        // terms from the tables via the half-angle identities (libm free)
        int32_t heading = rng.sampleI32(0, 8);
        float c = kMoveCos[heading];
        float s = kMoveSin[heading];
        float ch = sqrtf((1.f + c) * 0.5f);
        float sh = sqrtf((1.f - c) * 0.5f);
        if (s < 0.f) sh = -sh;

        ctx.get<Position>(agent) = pos;
        ctx.get<Rotation>(agent) = Quat { ch, 0.f, 0.f, sh }.normalize();
    }

    for (int32_t i = 0; i < consts::numBoxes; i++) {
        const bool is_long = i >= consts::numBoxes - 3;
        Entity box = ctx.makeEntity<MovableObject>();

        Vector3 pos = cellPos(slot++, 0.75f);
        // one box in ten starts 1 cm inside the floor
        if (rng.sampleUniform() < 0.1f) {
            pos.z -= 0.01f;
        }
        int32_t quarter = is_long ? rng.sampleI32(0, 2) : 0;

        if (i == consts::numBoxes - 1) {
            // the last "box" is an L-shaped compound of two hulls
            setupRigidBody(ctx, box, pos,
                Quat { kQuarterW[quarter], 0.f, 0.f, kQuarterZ[quarter] },
                SimObject::LBlock, EntityType::Box, ResponseType::Dynamic,
                Diag3x3 { 1.4f, 1.4f, 1.4f });
        } else {
            setupRigidBody(ctx, box, pos,
                Quat { kQuarterW[quarter], 0.f, 0.f, kQuarterZ[quarter] },
                is_long ? SimObject::LongBox : SimObject::Box, EntityType::Box,
                ResponseType::Dynamic,
                is_long ? Diag3x3 { 3.f, 1.2f, 1.5f } :
                          Diag3x3 { 1.5f, 1.5f, 1.5f });
        }
        ctx.get<LockState>(box) = LockState { 0, 0 };
        level.boxes[i] = box;
    }

    for (int32_t i = 0; i < consts::numRamps; i++) {
        Entity ramp = ctx.makeEntity<MovableObject>();

        Vector3 pos = cellPos(slot++, 0.8f);
        int32_t quarter = rng.sampleI32(0, 4);

        setupRigidBody(ctx, ramp, pos,
            Quat { kQuarterW[quarter], 0.f, 0.f, kQuarterZ[quarter] },
            SimObject::Ramp, EntityType::Ramp, ResponseType::Dynamic,
            Diag3x3 { 2.5f, 3.f, 1.5f });
        ctx.get<LockState>(ramp) = LockState { 0, 0 };
        level.ramps[i] = ramp;
    }

    for (int32_t i = 0; i < consts::numInteriorWalls; i++) {
        Entity wall = ctx.makeEntity<StaticObject>();

        int32_t along_y = rng.sampleI32(0, 2);
        int32_t line = 1 + rng.sampleI32(0, consts::gridDim - 1);
        int32_t seg = rng.sampleI32(0, consts::gridDim);

        float line_coord =
            (float)line * consts::cellSize - consts::arenaSize * 0.5f;
        float seg_coord = cellCenter(seg);

        Vector3 pos = along_y != 0 ?
            Vector3 { line_coord, seg_coord, consts::wallHeight * 0.5f } :
            Vector3 { seg_coord, line_coord, consts::wallHeight * 0.5f };
        Diag3x3 scale = along_y != 0 ?
            Diag3x3 { consts::wallThickness, consts::cellSize,
                      consts::wallHeight } :
            Diag3x3 { consts::cellSize, consts::wallThickness,
                      consts::wallHeight };

        setupRigidBody(ctx, wall, pos, Quat { 1, 0, 0, 0 }, SimObject::Wall,
            EntityType::Wall, ResponseType::Static, scale);
        level.walls[i] = wall;
    }
}

static void resetPersistentEntities(Engine &ctx)
{
    Sim &sim = ctx.data();

    registerRigidBodyEntity(ctx, sim.floorPlane);
    for (int32_t i = 0; i < consts::numBorderWalls; i++) {
        registerRigidBodyEntity(ctx, sim.borders[i]);
    }

    for (int32_t i = 0; i < consts::numAgents; i++) {
        Entity agent = sim.agents[i];
        registerRigidBodyEntity(ctx, agent);

        ctx.get<Velocity>(agent) = Velocity { Vector3::zero(), Vector3::zero() };
        ctx.get<ExternalForce>(agent) = Vector3::zero();
        ctx.get<ExternalTorque>(agent) = Vector3::zero();
        ctx.get<Action>(agent) = Action { 0, 0, 0, 0 };
        ctx.get<StepsRemaining>(agent).t = consts::episodeLen;
        ctx.get<Visibility>(agent) = Visibility { { 0, 0, 0, 0 } };
        ctx.get<Reward>(agent).v = 0.f;
        ctx.get<Done>(agent).v = 0;
    }
}

static void createPersistentEntities(Engine &ctx)
{
    Sim &sim = ctx.data();
    const float half = consts::arenaSize * 0.5f;
    const float t = consts::wallThickness;

    sim.floorPlane = ctx.makeEntity<StaticObject>();
    setupRigidBody(ctx, sim.floorPlane, Vector3 { 0, 0, 0 },
        Quat { 1, 0, 0, 0 }, SimObject::Plane, EntityType::None,
        ResponseType::Static, Diag3x3 { 1, 1, 1 });

    const Vector3 border_pos[consts::numBorderWalls] = {
        { 0.f, -half - t * 0.5f, consts::wallHeight * 0.5f },
        { 0.f, half + t * 0.5f, consts::wallHeight * 0.5f },
        { -half - t * 0.5f, 0.f, consts::wallHeight * 0.5f },
        { half + t * 0.5f, 0.f, consts::wallHeight * 0.5f },
    };
    const Diag3x3 border_scale[consts::numBorderWalls] = {
        { consts::arenaSize + 2.f * t, t, consts::wallHeight },
        { consts::arenaSize + 2.f * t, t, consts::wallHeight },
        { t, consts::arenaSize + 2.f * t, consts::wallHeight },
        { t, consts::arenaSize + 2.f * t, consts::wallHeight },
    };
    for (int32_t i = 0; i < consts::numBorderWalls; i++) {
        sim.borders[i] = ctx.makeEntity<StaticObject>();
        setupRigidBody(ctx, sim.borders[i], border_pos[i],
            Quat { 1, 0, 0, 0 }, SimObject::Wall, EntityType::Wall,
            ResponseType::Static, border_scale[i]);
    }

    for (int32_t i = 0; i < consts::numAgents; i++) {
        Entity agent = ctx.makeEntity<Agent>();
        sim.agents[i] = agent;

        setupRigidBody(ctx, agent, Vector3 { 0, 0, 1.f }, Quat { 1, 0, 0, 0 },
            SimObject::Agent,
            i < consts::numHiders ? EntityType::Hider : EntityType::Seeker,
            ResponseType::Dynamic, Diag3x3 { 1.2f, 1.2f, 2.f });
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

    // a fresh RNG stream per (world, episode), kept in registers while the
    // level is generated (the world object is in memory that every component
    // store might alias)
    RNG rng(rand::split_i(sim.initRandKey, sim.curWorldEpisode++));

    resetPersistentEntities(ctx);
    generateLevel(ctx, rng);

    sim.rng = rng;
}

static void cleanupWorld(Engine &ctx)
{
    LevelState &level = ctx.singleton<LevelState>();
    for (int32_t i = 0; i < consts::numBoxes; i++) {
        ctx.destroyEntity(level.boxes[i]);
    }
    for (int32_t i = 0; i < consts::numRamps; i++) {
        ctx.destroyEntity(level.ramps[i]);
    }
    for (int32_t i = 0; i < consts::numInteriorWalls; i++) {
        ctx.destroyEntity(level.walls[i]);
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
    constexpr float move_max = 800.f;
    constexpr float turn_max = 240.f;

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

// Lock / unlock the movable object in front of each agent: a locked object is
// a Static body until a member of the locking team releases it.  One
// invocation per world, agents in order, so the outcome does not depend on how
// a backend schedules the agents of a world.
inline void lockSystem(Engine &ctx, LevelState &)
{
    Sim &sim = ctx.data();

#ifdef SIM_WAVE_API
    // 64 lanes per world (CustomParallelForNode<..., 64, 1, ...>).  First the
    // box queries of all agents that press the lock button, a lane per BVH leaf
    // (locking moves nothing, so they see what the sequential loop sees); then
    // lane 0 toggles the locks agent by agent, in the CPU's order.
    Entity found[consts::numAgents];
    {
        AABB boxes[consts::numAgents];
        int32_t owner[consts::numAgents];
        int32_t num_boxes = 0;
        for (int32_t i = 0; i < consts::numAgents; i++) {
            Entity e = sim.agents[i];
            found[i] = Entity::none();
            if (ctx.get<Action>(e).lock == 0) {
                continue;
            }
            Vector3 reach = ctx.get<Position>(e) +
                ctx.get<Rotation>(e).rotateVec(Vector3 { 0.f, 1.6f, 0.f });
            boxes[num_boxes] = AABB {
                reach - Vector3 { 1.f, 1.f, 1.f },
                reach + Vector3 { 1.f, 1.f, 1.f },
            };
            owner[num_boxes++] = i;
        }
        Entity first[consts::numAgents];
        PhysicsSystem::findFirstEntitiesWithinAABBsWave<consts::numAgents>(
            ctx, boxes, num_boxes, first, [&](Entity other) {
                EntityType type = ctx.get<EntityType>(other);
                return type == EntityType::Box || type == EntityType::Ramp;
            });
        for (int32_t b = 0; b < num_boxes; b++) {
            found[owner[b]] = first[b];
        }
    }
    if (threadIdx.x % 64 != 0) {
        return;
    }
#endif

    for (int32_t i = 0; i < consts::numAgents; i++) {
        Entity e = sim.agents[i];
        if (ctx.get<Action>(e).lock == 0) {
            continue;
        }
        const int32_t is_hider = i < consts::numHiders ? 1 : 0;

#ifdef SIM_WAVE_API
        Entity target = found[i];
#else
        Vector3 pos = ctx.get<Position>(e);
        Quat rot = ctx.get<Rotation>(e);

        Vector3 reach = pos + rot.rotateVec(Vector3 { 0.f, 1.6f, 0.f });
        AABB reach_box {
            reach - Vector3 { 1.f, 1.f, 1.f },
            reach + Vector3 { 1.f, 1.f, 1.f },
        };

        Entity target = Entity::none();
        PhysicsSystem::findEntitiesWithinAABB(ctx, reach_box,
            [&](Entity other) {
                if (target != Entity::none()) {
                    return;
                }
                EntityType type = ctx.get<EntityType>(other);
                if (type == EntityType::Box || type == EntityType::Ramp) {
                    target = other;
                }
            });
#endif

        if (target == Entity::none()) {
            continue;
        }

        LockState &lock = ctx.get<LockState>(target);
        if (lock.locked == 0) {
            lock.locked = 1;
            lock.byHiders = is_hider;
            ctx.get<ResponseType>(target) = ResponseType::Static;
            ctx.get<Velocity>(target) =
                Velocity { Vector3::zero(), Vector3::zero() };
        } else if (lock.byHiders == is_hider) {
            lock.locked = 0;
            ctx.get<ResponseType>(target) = ResponseType::Dynamic;
        }
    }
}

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

// Which of the other agents are in line of sight: one ray per other agent
// through the world's BVH, from inside the agent's own hull.  On the GPU
// backends the node runs 4 threads per agent, one ray each.
inline void visibilitySystem(Engine &ctx,
                             Position &pos,
                             const OtherAgents &others,
                             const StepsRemaining &steps,
                             EntityType &type,
                             Visibility &visibility)
{
    broadphase::BVH &bvh = ctx.singleton<broadphase::BVH>();

    const bool prep =
        steps.t > (uint32_t)(consts::episodeLen - consts::prepLen);
    const bool blind = prep && type == EntityType::Seeker;

    Vector3 ray_o = pos + 0.5f * math::up;

    auto look = [&](int32_t i) {
        uint32_t seen = 0;
        if (!blind) {
            Entity other = others.e[i];
            Vector3 to_other =
                (ctx.get<Position>(other) + 0.5f * math::up) - ray_o;
            float dist = to_other.length();
            if (dist < 1e-3f) {
                seen = 1;
            } else {
                Vector3 ray_d = to_other / dist;

                float hit_t;
                Vector3 hit_normal;
                Entity hit =
                    bvh.traceRay(ray_o, ray_d, &hit_t, &hit_normal, dist);
                seen = hit == other ? 1 : 0;
            }
        }
        visibility.seen[i] = seen;
    };

#ifdef MADRONA_GPU_MODE
    look((int32_t)(threadIdx.x % 4));
#else
    for (int32_t i = 0; i < consts::numAgents - 1; i++) {
        look(i);
    }
#endif
}

// hiders lose a point while a seeker sees them, seekers win one while they see
// a hider; nothing during the preparation phase
inline void rewardSystem(Engine &ctx,
                         const OtherAgents &others,
                         const Visibility &visibility,
                         const StepsRemaining &steps,
                         EntityType &type,
                         Reward &out_reward)
{
    if (steps.t > (uint32_t)(consts::episodeLen - consts::prepLen)) {
        out_reward.v = 0.f;
        return;
    }

    bool sees_opponent = false;
    for (int32_t i = 0; i < consts::numAgents - 1; i++) {
        if (visibility.seen[i] != 0 &&
                ctx.get<EntityType>(others.e[i]) != type) {
            sees_opponent = true;
        }
    }

    if (type == EntityType::Hider) {
        out_reward.v = sees_opponent ? -1.f : 1.f;
    } else {
        out_reward.v = sees_opponent ? 1.f : -1.f;
    }
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
// entity i.  Same result as cleanupWorld() + initWorld() above, bit for bit:
// ids through Context::destroyEntityOrdered / makeEntityOrdered (lane order =
// the order of the loops above), BVH leaves by position in the registration
// sequence (floor, borders, agents, boxes, ramps, walls), random numbers by
// sample index -- sample i of an episode is split_i(episode key, i), and the
// loops draw a fixed number per object:
//   cell shuffle: 0 .. numAgents + numMovable - 1
//   agent a:   S + 3 a + { x, y, heading }                    (S = shuffle draws)
//   box i:     A + 3 i (+ i - 6 for the three long ones) + { x, y, sunk, [quarter] }
//   ramp i:    B + 3 i + { x, y, quarter };   wall i:  R + 3 i + { along_y, line, seg }
static inline void resetWorldWave(Engine &ctx)
{
    Sim &sim = ctx.data();
    LevelState &level = ctx.singleton<LevelState>();
    broadphase::BVH &bvh = ctx.singleton<broadphase::BVH>();

    constexpr int32_t num_level =
        consts::numBoxes + consts::numRamps + consts::numInteriorWalls;
    constexpr int32_t num_persistent = 1 + consts::numBorderWalls + consts::numAgents;
    constexpr int32_t num_cells = consts::gridDim * consts::gridDim;
    constexpr int32_t num_placed = consts::numAgents + consts::numMovable;
    constexpr int32_t num_long = 3;
    constexpr uint32_t draws_shuffle = (uint32_t)num_placed;
    constexpr uint32_t draws_agents = draws_shuffle + 3u * consts::numAgents;
    constexpr uint32_t draws_boxes = draws_agents + 3u * consts::numBoxes + num_long;
    constexpr uint32_t draws_ramps = draws_boxes + 3u * consts::numRamps;
    constexpr uint32_t num_draws = draws_ramps + 3u * consts::numInteriorWalls;
    static_assert(num_level <= 64 && num_persistent <= 64);

    const int32_t lane = (int32_t)(threadIdx.x % 64);
    const bool on_level = lane < num_level;
    const bool is_box = lane < consts::numBoxes;
    const bool is_ramp = !is_box && lane < consts::numBoxes + consts::numRamps;

    // ---- cleanupWorld(): boxes, ramps, walls ----
    {
        Entity e = Entity::none();
        if (on_level) {
            e = is_box ? level.boxes[lane] :
                (is_ramp ? level.ramps[lane - consts::numBoxes] :
                 level.walls[lane - consts::numBoxes - consts::numRamps]);
        }
        ctx.destroyEntityOrdered(e, on_level);
    }

    // ---- initWorld() ----
    if (lane == 0) {
        PhysicsSystem::reset(ctx);
        bvh.setNumLeaves(num_persistent + num_level);
    }

    const RandKey episode = rand::split_i(sim.initRandKey, sim.curWorldEpisode);
    auto key = [&](uint32_t i) { return rand::split_i(episode, i); };
    auto in_range = [&](uint32_t i, float lo, float hi) {
        return lo + rand::sampleUniform(key(i)) * (hi - lo);
    };

    // the partial shuffle of the cell list (every lane, the same)
    int32_t cells[num_cells];
    for (int32_t i = 0; i < num_cells; i++) {
        cells[i] = i;
    }
    for (int32_t i = 0; i < num_placed; i++) {
        int32_t j = i + rand::sampleI32(key((uint32_t)i), 0, num_cells - i);
        int32_t tmp = cells[i];
        cells[i] = cells[j];
        cells[j] = tmp;
    }
    // slot: agents, boxes, ramps in that order; the draws of its x and y
    auto cell_pos = [&](int32_t slot, uint32_t at, float z) {
        int32_t cell = cells[slot];
        return Vector3 {
            cellCenter(cell % consts::gridDim) + in_range(at, -0.25f, 0.25f),
            cellCenter(cell / consts::gridDim) + in_range(at + 1u, -0.25f, 0.25f),
            z,
        };
    };

    // resetPersistentEntities() + the agents' part of generateLevel()
    if (lane < num_persistent) {
        const Entity e = lane == 0 ? sim.floorPlane :
            (lane <= consts::numBorderWalls ? sim.borders[lane - 1] :
             sim.agents[lane - 1 - consts::numBorderWalls]);
        ctx.get<broadphase::LeafID>(e) =
            bvh.reserveLeafAt(lane, e, ctx.get<ObjectID>(e));

        if (lane > consts::numBorderWalls) {
            const int32_t i = lane - 1 - consts::numBorderWalls;
            Entity agent = e;
            ctx.get<Velocity>(agent) = Velocity { Vector3::zero(), Vector3::zero() };
            ctx.get<ExternalForce>(agent) = Vector3::zero();
            ctx.get<ExternalTorque>(agent) = Vector3::zero();
            ctx.get<Action>(agent) = Action { 0, 0, 0, 0 };
            ctx.get<StepsRemaining>(agent).t = consts::episodeLen;
            ctx.get<Visibility>(agent) = Visibility { { 0, 0, 0, 0 } };
            ctx.get<Reward>(agent).v = 0.f;
            ctx.get<Done>(agent).v = 0;

            const uint32_t at = draws_shuffle + 3u * (uint32_t)i;
            Vector3 pos = cell_pos(i, at, 1.f);
            int32_t heading = rand::sampleI32(key(at + 2u), 0, 8);
            float c = kMoveCos[heading];
            float sn = kMoveSin[heading];
            float ch = sqrtf((1.f + c) * 0.5f);
            float sh = sqrtf((1.f - c) * 0.5f);
            if (sn < 0.f) sh = -sh;
            ctx.get<Position>(agent) = pos;
            ctx.get<Rotation>(agent) = Quat { ch, 0.f, 0.f, sh }.normalize();
        }
    }

    // generateLevel(): boxes, ramps (MovableObject), walls (StaticObject)
    const uint32_t archetype = is_box || is_ramp ?
        TypeTracker::typeID<MovableObject>() : TypeTracker::typeID<StaticObject>();
    const Entity e = ctx.makeEntityOrdered(archetype, on_level);

    if (on_level) {
        // setupRigidBody() with the leaf this body has in the sequence
        auto fill_body = [&](Vector3 pos, Quat rot, SimObject obj, EntityType type,
                             ResponseType response, Diag3x3 scale) {
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
                bvh.reserveLeafAt(num_persistent + lane, e, obj_id);
            ctx.get<EntityType>(e) = type;
        };

        if (is_box) {
            const int32_t i = lane;
            const bool is_long = i >= consts::numBoxes - num_long;
            const uint32_t at = draws_agents + 3u * (uint32_t)i +
                (is_long ? (uint32_t)(i - (consts::numBoxes - num_long)) : 0u);
            Vector3 pos = cell_pos(consts::numAgents + i, at, 0.75f);
            if (rand::sampleUniform(key(at + 2u)) < 0.1f) {
                pos.z -= 0.01f;
            }
            int32_t quarter = is_long ? rand::sampleI32(key(at + 3u), 0, 2) : 0;
            if (i == consts::numBoxes - 1) {
                fill_body(pos,
                    Quat { kQuarterW[quarter], 0.f, 0.f, kQuarterZ[quarter] },
                    SimObject::LBlock, EntityType::Box, ResponseType::Dynamic,
                    Diag3x3 { 1.4f, 1.4f, 1.4f });
            } else {
                fill_body(pos,
                    Quat { kQuarterW[quarter], 0.f, 0.f, kQuarterZ[quarter] },
                    is_long ? SimObject::LongBox : SimObject::Box,
                    EntityType::Box, ResponseType::Dynamic,
                    is_long ? Diag3x3 { 3.f, 1.2f, 1.5f } :
                              Diag3x3 { 1.5f, 1.5f, 1.5f });
            }
            ctx.get<LockState>(e) = LockState { 0, 0 };
            level.boxes[i] = e;
        } else if (is_ramp) {
            const int32_t i = lane - consts::numBoxes;
            const uint32_t at = draws_boxes + 3u * (uint32_t)i;
            Vector3 pos = cell_pos(consts::numAgents + consts::numBoxes + i, at, 0.8f);
            int32_t quarter = rand::sampleI32(key(at + 2u), 0, 4);
            fill_body(pos,
                Quat { kQuarterW[quarter], 0.f, 0.f, kQuarterZ[quarter] },
                SimObject::Ramp, EntityType::Ramp, ResponseType::Dynamic,
                Diag3x3 { 2.5f, 3.f, 1.5f });
            ctx.get<LockState>(e) = LockState { 0, 0 };
            level.ramps[i] = e;
        } else {
            const int32_t i = lane - consts::numBoxes - consts::numRamps;
            const uint32_t at = draws_ramps + 3u * (uint32_t)i;
            int32_t along_y = rand::sampleI32(key(at), 0, 2);
            int32_t line = 1 + rand::sampleI32(key(at + 1u), 0, consts::gridDim - 1);
            int32_t seg = rand::sampleI32(key(at + 2u), 0, consts::gridDim);

            float line_coord =
                (float)line * consts::cellSize - consts::arenaSize * 0.5f;
            float seg_coord = cellCenter(seg);

            Vector3 pos = along_y != 0 ?
                Vector3 { line_coord, seg_coord, consts::wallHeight * 0.5f } :
                Vector3 { seg_coord, line_coord, consts::wallHeight * 0.5f };
            Diag3x3 scale = along_y != 0 ?
                Diag3x3 { consts::wallThickness, consts::cellSize,
                          consts::wallHeight } :
                Diag3x3 { consts::cellSize, consts::wallThickness,
                          consts::wallHeight };
            fill_body(pos, Quat { 1, 0, 0, 0 }, SimObject::Wall, EntityType::Wall,
                      ResponseType::Static, scale);
            level.walls[i] = e;
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

static inline ObjectObservation observeObject(Engine &ctx, Entity e,
                                              Vector3 pos, Quat to_view)
{
    Vector3 rel = to_view.rotateVec(ctx.get<Position>(e) - pos);
    Vector3 v = ctx.get<Velocity>(e).linear;
    return ObjectObservation {
        rel.x / consts::arenaSize,
        rel.y / consts::arenaSize,
        rel.z / consts::arenaSize,
        v.length2() / 100.f,
        ctx.get<LockState>(e).locked != 0 ? 1.f : 0.f,
    };
}

inline void collectObservationsSystem(Engine &ctx,
                                      Position &pos,
                                      Rotation &rot,
                                      const Velocity &vel,
                                      const OtherAgents &other_agents,
                                      const Visibility &visibility,
                                      const StepsRemaining &steps,
                                      EntityType &type,
                                      SelfObservation &self_obs,
                                      AgentObservations &agent_obs,
                                      BoxObservations &box_obs,
                                      RampObservations &ramp_obs)
{
    const LevelState &level = ctx.singleton<LevelState>();

    int32_t prep_remaining = (int32_t)steps.t -
        (consts::episodeLen - consts::prepLen);
    if (prep_remaining < 0) prep_remaining = 0;

    self_obs.x = pos.x / consts::arenaSize;
    self_obs.y = pos.y / consts::arenaSize;
    self_obs.z = pos.z / 10.f;
    self_obs.facing = rot.z;
    self_obs.vx = vel.linear.x / 10.f;
    self_obs.vy = vel.linear.y / 10.f;
    self_obs.isHider = type == EntityType::Hider ? 1.f : 0.f;
    self_obs.prepRemaining = (float)prep_remaining / (float)consts::prepLen;

    Quat to_view = rot.inv();

    for (int32_t i = 0; i < consts::numAgents - 1; i++) {
        Entity other = other_agents.e[i];
        Vector3 rel = to_view.rotateVec(ctx.get<Position>(other) - pos);
        agent_obs.obs[i] = AgentObservation {
            rel.x / consts::arenaSize,
            rel.y / consts::arenaSize,
            ctx.get<EntityType>(other) == EntityType::Hider ? 1.f : 0.f,
            visibility.seen[i] != 0 ? 1.f : 0.f,
        };
    }

    for (int32_t i = 0; i < consts::numBoxes; i++) {
        box_obs.obs[i] = observeObject(ctx, level.boxes[i], pos, to_view);
    }
    for (int32_t i = 0; i < consts::numRamps; i++) {
        ramp_obs.obs[i] = observeObject(ctx, level.ramps[i], pos, to_view);
    }
}

// 30 rays per agent through the world's BVH: depth and the type of what was
// hit.  On the GPU backends the node runs 32 threads per agent, one ray each
// (the reference simulators' convention: CustomParallelForNode<..., 32, 1, ...>
// + threadIdx.x % 32).
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

#ifdef SIM_WAVE_API
// (this backend) the lock system is a chain of dependent loads with a wavefront
// per world: four wavefronts per SIMD instead of the three its 154 registers
// allow (109 -> 99 us at 8192 worlds; the ray systems do not profit)
}
template <> inline constexpr unsigned
    madrona::mwhip::systemWavesPerSIMD<hideseek::lockSystem> = 4;
// (the ray systems: 131 / 133 registers with the leaf tests' loads batched
// -- one too many for four wavefronts per SIMD, which is what they want:
// profiles/r03_lidar_occupancy_variants.jsonl)
template <> inline constexpr unsigned
    madrona::mwhip::systemWavesPerSIMD<hideseek::lidarSystem> = 4;
template <> inline constexpr unsigned
    madrona::mwhip::systemWavesPerSIMD<hideseek::visibilitySystem> = 4;
namespace hideseek {
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
    // 64 lanes per world: the lock queries test a BVH leaf per lane
    auto lock_sys = builder.addToGraph<CustomParallelForNode<Engine,
        lockSystem, 64, 1,
#else
    auto lock_sys = builder.addToGraph<ParallelForNode<Engine,
        lockSystem,
#endif
            LevelState
        >>({broadphase_setup_sys});

    auto substep_sys = PhysicsSystem::setupPhysicsStepTasks(builder,
        {lock_sys}, consts::numPhysicsSubsteps);

    // (the real dependencies, not a chain -- sims/escape_room/sim.cpp: the
    // velocity reset and the step counter touch different components)
    auto phys_done =
        PhysicsSystem::setupCleanupTasks(builder, {substep_sys});

    auto agent_zero_vel = builder.addToGraph<ParallelForNode<Engine,
        agentZeroVelSystem,
            Velocity,
            Action
        >>({phys_done});

    auto done_sys = builder.addToGraph<ParallelForNode<Engine,
        stepTrackerSystem,
            StepsRemaining,
            Done
        >>({phys_done});

#ifdef SIM_WAVE_API
    // 64 lanes per world: lane i resets entity i (resetWorldWave)
    auto reset_sys = builder.addToGraph<CustomParallelForNode<Engine,
        resetSystem, 64, 1,
#else
    auto reset_sys = builder.addToGraph<ParallelForNode<Engine,
        resetSystem,
#endif
            WorldReset
        >>({agent_zero_vel, done_sys});

#ifdef MADRONA_GPU_MODE
    auto recycle_sys = builder.addToGraph<RecycleEntitiesNode>({reset_sys});
    auto post_reset = recycle_sys;
#else
    auto post_reset = reset_sys;
#endif

    auto compact_movable = builder.addToGraph<
        CompactArchetypeNode<MovableObject>>({post_reset});
    auto compact_static = builder.addToGraph<
        CompactArchetypeNode<StaticObject>>({compact_movable});

    // line of sight, observations and lidar see the post-step (or freshly
    // reset) world: refresh the BVH first
    auto post_reset_broadphase =
        PhysicsSystem::setupBroadphaseTasks(builder, {compact_static});

    static_assert(consts::numAgents - 1 == 4);
#ifdef MADRONA_GPU_MODE
    auto visibility_sys = builder.addToGraph<CustomParallelForNode<Engine,
        visibilitySystem, 4, 1,
#else
    auto visibility_sys = builder.addToGraph<ParallelForNode<Engine,
        visibilitySystem,
#endif
            Position,
            OtherAgents,
            StepsRemaining,
            EntityType,
            Visibility
        >>({post_reset_broadphase});

    auto reward_sys = builder.addToGraph<ParallelForNode<Engine,
        rewardSystem,
            OtherAgents,
            Visibility,
            StepsRemaining,
            EntityType,
            Reward
        >>({visibility_sys});

    auto collect_obs = builder.addToGraph<ParallelForNode<Engine,
        collectObservationsSystem,
            Position,
            Rotation,
            Velocity,
            OtherAgents,
            Visibility,
            StepsRemaining,
            EntityType,
            SelfObservation,
            AgentObservations,
            BoxObservations,
            RampObservations
        >>({visibility_sys});     // (beside the reward: both read Visibility)

#ifdef MADRONA_GPU_MODE
    auto lidar = builder.addToGraph<CustomParallelForNode<Engine,
        lidarSystem, 32, 1,
#else
    auto lidar = builder.addToGraph<ParallelForNode<Engine,
        lidarSystem,
#endif
            Entity,
            Lidar
        >>({post_reset_broadphase});      // (rays need the tree, nothing else)

    (void)reward_sys;
    (void)collect_obs;
    (void)lidar;
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

    createPersistentEntities(ctx);
    initWorld(ctx);
}

#ifdef MADRONA_GPU_MODE
MADRONA_BUILD_MWGPU_ENTRY(Engine, Sim, Sim::Config, Sim::WorldInit);
#endif

}

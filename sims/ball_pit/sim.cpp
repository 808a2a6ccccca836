#include "sim.hpp"

#ifdef MADRONA_GPU_MODE
#include <madrona/mw_gpu_entry.hpp>
#endif

using namespace madrona;
using namespace madrona::math;
using namespace madrona::phys;

namespace ballpit {

// rotation about z by k * 90 degrees
static constexpr float kQuarterW[4] = { 1.f, 0.70710678f, 0.f, 0.70710678f };
static constexpr float kQuarterZ[4] = { 0.f, 0.70710678f, 1.f, -0.70710678f };

void Sim::registerTypes(ECSRegistry &registry, const Config &)
{
    base::registerTypes(registry);
    PhysicsSystem::registerTypes(registry);

    registry.registerComponent<KickIndex>();

    registry.registerSingleton<WorldReset>();
    registry.registerSingleton<StepsRemaining>();
    registry.registerSingleton<LevelState>();
    registry.registerSingleton<QueryProbe>();

    registry.registerArchetype<MovableObject>();
    registry.registerArchetype<StaticObject>();

    registry.exportSingleton<WorldReset>((uint32_t)ExportID::Reset);
    registry.exportSingleton<StepsRemaining>(
        (uint32_t)ExportID::StepsRemaining);
    registry.exportSingleton<QueryProbe>((uint32_t)ExportID::QueryProbe);
}

static inline float randInRange(RNG &rng, float lo, float hi)
{
    return lo + rng.sampleUniform() * (hi - lo);
}

static inline void setupRigidBody(Engine &ctx, Entity e, Vector3 pos, Quat rot,
                                  SimObject obj, ResponseType response,
                                  Diag3x3 scale)
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
}

// The movable objects start on a 4 x 4 board of cells at staggered heights, so
// they fall onto the floor and onto each other.
static void generateLevel(Engine &ctx, RNG &rng)
{
    LevelState &level = ctx.singleton<LevelState>();

    constexpr int32_t board = 4;
    constexpr float cell = consts::pitSize / (float)board;

    for (int32_t i = 0; i < consts::numMovable; i++) {
        Entity e = ctx.makeEntity<MovableObject>();

        int32_t slot = (i * 5 + 3) % (board * board);
        Vector3 pos {
            ((float)(slot % board) + 0.5f) * cell - consts::pitSize * 0.5f +
                randInRange(rng, -0.4f, 0.4f),
            ((float)(slot / board) + 0.5f) * cell - consts::pitSize * 0.5f +
                randInRange(rng, -0.4f, 0.4f),
            randInRange(rng, 1.f, 4.f),
        };

        if (i < consts::numSpheres) {
            float r = randInRange(rng, 0.5f, 1.1f);
            setupRigidBody(ctx, e, pos, Quat { 1, 0, 0, 0 }, SimObject::Sphere,
                           ResponseType::Dynamic, Diag3x3 { r, r, r });
        } else if (i == consts::numSpheres + consts::numBoxes - 1) {
            // the compound object: every pair it is in yields one candidate
            // per primitive
            int32_t quarter = rng.sampleI32(0, 4);
            float sc = randInRange(rng, 0.8f, 1.2f);
            setupRigidBody(ctx, e, pos,
                Quat { kQuarterW[quarter], 0.f, 0.f, kQuarterZ[quarter] },
                SimObject::LBlock, ResponseType::Dynamic,
                Diag3x3 { sc, sc, sc });
        } else if (i < consts::numSpheres + consts::numBoxes) {
            int32_t quarter = rng.sampleI32(0, 4);
            setupRigidBody(ctx, e, pos,
                Quat { kQuarterW[quarter], 0.f, 0.f, kQuarterZ[quarter] },
                SimObject::Box, ResponseType::Dynamic,
                Diag3x3 { randInRange(rng, 0.8f, 1.6f),
                          randInRange(rng, 0.8f, 1.6f),
                          randInRange(rng, 0.8f, 1.6f) });
        } else {
            int32_t quarter = rng.sampleI32(0, 4);
            // coins start flat, some of them above one another
            const int32_t coin = i - consts::numSpheres - consts::numBoxes;
            if (coin == 2) {
                Vector3 below = ctx.get<Position>(level.movable[i - 1]);
                pos.x = below.x + randInRange(rng, -0.2f, 0.2f);
                pos.y = below.y + randInRange(rng, -0.2f, 0.2f);
                pos.z = below.z + 0.8f;
            }
            float sc = randInRange(rng, 0.8f, 1.3f);
            setupRigidBody(ctx, e, pos,
                Quat { kQuarterW[quarter], 0.f, 0.f, kQuarterZ[quarter] },
                coin == 0 ? SimObject::Coin12 : SimObject::Coin16,
                ResponseType::Dynamic, Diag3x3 { sc, sc, 1.f });
        }

        ctx.get<KickIndex>(e).idx = i;
        level.movable[i] = e;
    }

    // crowd mode: a field of extra spheres and boxes around the pit
    for (int32_t i = 0; i < ctx.data().numExtra; i++) {
        Entity e = ctx.makeEntity<MovableObject>();

        constexpr int32_t field = 13;
        Vector3 pos {
            ((float)(i % field) - 6.f) * 3.f + randInRange(rng, -0.5f, 0.5f),
            ((float)(i / field) - 6.f) * 3.f + randInRange(rng, -0.5f, 0.5f),
            randInRange(rng, 1.f, 3.f),
        };
        if (i % 2 == 0) {
            float r = randInRange(rng, 0.5f, 1.f);
            setupRigidBody(ctx, e, pos, Quat { 1, 0, 0, 0 }, SimObject::Sphere,
                           ResponseType::Dynamic, Diag3x3 { r, r, r });
        } else {
            setupRigidBody(ctx, e, pos, Quat { 1, 0, 0, 0 }, SimObject::Box,
                ResponseType::Dynamic,
                Diag3x3 { randInRange(rng, 0.8f, 1.4f),
                          randInRange(rng, 0.8f, 1.4f),
                          randInRange(rng, 0.8f, 1.4f) });
        }
        ctx.get<KickIndex>(e).idx = consts::numMovable + i;
        level.extra[i] = e;
    }

    // movable[k] -- movable[k + 1], rigid at their initial distance
    for (int32_t k = 0; k < consts::numJoints; k++) {
        Entity a = level.movable[k];
        Entity b = level.movable[k + 1];
        Vector3 pa = ctx.get<Position>(a);
        Vector3 pb = ctx.get<Position>(b);
        Quat qa = ctx.get<Rotation>(a);
        Quat qb = ctx.get<Rotation>(b);
        if (ctx.data().hingeMode != 0 && k % 2 == 1) {
            // hinge about the x axis of `a` as both bodies see it now, pivot
            // half way between the two centres
            Quat a_to_b = (qb.inv() * qa).normalize();
            Vector3 mid = (pa + pb) * 0.5f;
            level.joints[k] = PhysicsSystem::makeHingeJoint(ctx, a, b,
                Vector3 { 1, 0, 0 }, a_to_b.rotateVec(Vector3 { 1, 0, 0 }),
                Vector3 { 0, 1, 0 }, a_to_b.rotateVec(Vector3 { 0, 1, 0 }),
                qa.inv().rotateVec(mid - pa), qb.inv().rotateVec(mid - pb));
            continue;
        }
        level.joints[k] = PhysicsSystem::makeFixedJoint(ctx, a, b,
            Quat { 1, 0, 0, 0 }, (qb.inv() * qa).normalize(),
            Vector3::zero(), Vector3::zero(), (pb - pa).length());
    }
}

static void createPersistentEntities(Engine &ctx)
{
    Sim &sim = ctx.data();
    const float half = consts::pitSize * 0.5f;
    const float t = consts::wallThickness;

    sim.floorPlane = ctx.makeEntity<StaticObject>();
    setupRigidBody(ctx, sim.floorPlane, Vector3 { 0, 0, 0 },
        Quat { 1, 0, 0, 0 }, SimObject::Plane, ResponseType::Static,
        Diag3x3 { 1, 1, 1 });

    const Vector3 wall_pos[consts::numWalls] = {
        { 0.f, -half - t * 0.5f, consts::wallHeight * 0.5f },
        { 0.f, half + t * 0.5f, consts::wallHeight * 0.5f },
        { -half - t * 0.5f, 0.f, consts::wallHeight * 0.5f },
        { half + t * 0.5f, 0.f, consts::wallHeight * 0.5f },
    };
    const Diag3x3 wall_scale[consts::numWalls] = {
        { consts::pitSize + 2.f * t, t, consts::wallHeight },
        { consts::pitSize + 2.f * t, t, consts::wallHeight },
        { t, consts::pitSize + 2.f * t, consts::wallHeight },
        { t, consts::pitSize + 2.f * t, consts::wallHeight },
    };
    for (int32_t i = 0; i < consts::numWalls; i++) {
        sim.walls[i] = ctx.makeEntity<StaticObject>();
        setupRigidBody(ctx, sim.walls[i], wall_pos[i], Quat { 1, 0, 0, 0 },
            SimObject::Wall, ResponseType::Static, wall_scale[i]);
    }
}

static void initWorld(Engine &ctx)
{
    Sim &sim = ctx.data();

    PhysicsSystem::reset(ctx);

    RNG rng(rand::split_i(sim.initRandKey, sim.curWorldEpisode++));

    ctx.get<broadphase::LeafID>(sim.floorPlane) = PhysicsSystem::registerEntity(
        ctx, sim.floorPlane, ctx.get<ObjectID>(sim.floorPlane));
    for (int32_t i = 0; i < consts::numWalls; i++) {
        ctx.get<broadphase::LeafID>(sim.walls[i]) =
            PhysicsSystem::registerEntity(ctx, sim.walls[i],
                                          ctx.get<ObjectID>(sim.walls[i]));
    }

    generateLevel(ctx, rng);

    ctx.singleton<StepsRemaining>().t = consts::episodeLen;
    sim.rng = rng;
}

static void cleanupWorld(Engine &ctx)
{
    LevelState &level = ctx.singleton<LevelState>();
    for (int32_t k = 0; k < consts::numJoints; k++) {
        ctx.destroyEntity(level.joints[k]);
    }
    for (int32_t i = 0; i < consts::numMovable; i++) {
        ctx.destroyEntity(level.movable[i]);
    }
    for (int32_t i = 0; i < ctx.data().numExtra; i++) {
        ctx.destroyEntity(level.extra[i]);
    }
}

// ---------------------------------------------------------------------------
// systems
// ---------------------------------------------------------------------------
// Every 8 steps one object per world (round robin) is kicked sideways and up.
// One invocation per world: the world's RNG stream advances in a fixed order.
inline void kickSystem(Engine &ctx, LevelState &level)
{
    Sim &sim = ctx.data();
    int32_t t = ctx.singleton<StepsRemaining>().t;

    for (int32_t i = 0; i < consts::numMovable; i++) {
        ctx.get<ExternalForce>(level.movable[i]) = Vector3::zero();
        ctx.get<ExternalTorque>(level.movable[i]) = Vector3::zero();
    }

    if (t % 8 != 0) {
        return;
    }

    RNG rng = sim.rng;
    int32_t target = (t / 8) % consts::numMovable;
    Vector3 force {
        randInRange(rng, -300.f, 300.f),
        randInRange(rng, -300.f, 300.f),
        randInRange(rng, 0.f, 250.f),
    };
    Vector3 torque {
        randInRange(rng, -20.f, 20.f),
        randInRange(rng, -20.f, 20.f),
        randInRange(rng, -20.f, 20.f),
    };
    sim.rng = rng;

    if (sim.overlapMode != 0) {
        // PhysicsSystem::checkEntityAABBOverlap decides the kick's direction:
        // objects whose hull reaches into the centre column are pushed down
        // instead of up (spheres never overlap: the test only looks at hulls)
        math::AABB column {
            .pMin = Vector3 { -3.f, -3.f, 0.f },
            .pMax = Vector3 { 3.f, 3.f, 8.f },
        };
        if (PhysicsSystem::checkEntityAABBOverlap(ctx, column,
                                                  level.movable[target])) {
            force.z = -force.z;
        }
    }

    ctx.get<ExternalForce>(level.movable[target]) = force;
    ctx.get<ExternalTorque>(level.movable[target]) = torque;
}

// Four box queries per world and step around objects picked by the step count:
// the first dynamic body findEntitiesWithinAABB reports in each (QueryProbe).
inline void probeSystem(Engine &ctx, QueryProbe &probe)
{
    constexpr int32_t num_boxes = 4;
    const LevelState &level = ctx.singleton<LevelState>();
    const int32_t t = ctx.singleton<StepsRemaining>().t;

    math::AABB boxes[num_boxes];
    for (int32_t k = 0; k < num_boxes; k++) {
        const Entity anchor =
            level.movable[(uint32_t)(t * 3 + k * 5) % (uint32_t)consts::numMovable];
        const Vector3 centre = ctx.get<Position>(anchor) +
            Vector3 { ((float)k - 1.5f) * 0.7f, 0.3f, 0.f };
        const float half = 0.6f + 0.45f * (float)k;
        boxes[k] = math::AABB {
            .pMin = centre - Vector3 { half, half, half },
            .pMax = centre + Vector3 { half, half, half },
        };
    }
    auto accept = [&](Entity e) {
        return ctx.get<ResponseType>(e) == ResponseType::Dynamic;
    };

    Entity first[num_boxes];
#ifdef MADRONA_GPU_MODE
    // (CustomParallelForNode<..., 64, 1, ...>: a wavefront per world)
    PhysicsSystem::findFirstEntitiesWithinAABBsWave<num_boxes>(
        ctx, boxes, num_boxes, first, accept);
    if (threadIdx.x % 64 != 0) {
        return;
    }
#else
    for (int32_t k = 0; k < num_boxes; k++) {
        first[k] = Entity::none();
        PhysicsSystem::findEntitiesWithinAABB(ctx, boxes[k], [&](Entity e) {
            if (first[k] == Entity::none() && accept(e)) {
                first[k] = e;
            }
        });
    }
#endif

    for (int32_t k = 0; k < num_boxes; k++) {
        probe.digest = probe.digest * 0x9E3779B1u +
            (uint32_t)(first[k].id + 1) * 31u + first[k].gen;
        probe.found += first[k] != Entity::none() ? 1 : 0;
    }
}

inline void resetSystem(Engine &ctx, WorldReset &reset)
{
    Sim &sim = ctx.data();

    int32_t should_reset = reset.reset;

    StepsRemaining &steps = ctx.singleton<StepsRemaining>();
    steps.t -= 1;
    if (steps.t <= 0) {
        should_reset = 1;
    }

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
}

void Sim::setupTasks(TaskGraphManager &taskgraph_mgr, const Config &)
{
    TaskGraphBuilder &builder = taskgraph_mgr.init(0);

    auto kick_sys = builder.addToGraph<ParallelForNode<Engine,
        kickSystem,
            LevelState
        >>({});

    auto broadphase_setup_sys =
        PhysicsSystem::setupBroadphaseTasks(builder, {kick_sys});

#ifdef MADRONA_GPU_MODE
    auto probe_sys = builder.addToGraph<CustomParallelForNode<Engine,
        probeSystem, 64, 1,
#else
    auto probe_sys = builder.addToGraph<ParallelForNode<Engine,
        probeSystem,
#endif
            QueryProbe
        >>({broadphase_setup_sys});

    auto substep_sys = PhysicsSystem::setupPhysicsStepTasks(builder,
        {probe_sys}, consts::numPhysicsSubsteps);

    auto phys_done =
        PhysicsSystem::setupCleanupTasks(builder, {substep_sys});

    auto reset_sys = builder.addToGraph<ParallelForNode<Engine,
        resetSystem,
            WorldReset
        >>({phys_done});

#ifdef MADRONA_GPU_MODE
    auto recycle_sys = builder.addToGraph<RecycleEntitiesNode>({reset_sys});
    auto post_reset = recycle_sys;
#else
    auto post_reset = reset_sys;
#endif

    auto compact_movable = builder.addToGraph<
        CompactArchetypeNode<MovableObject>>({post_reset});

    auto post_reset_broadphase =
        PhysicsSystem::setupBroadphaseTasks(builder, {compact_movable});

    (void)post_reset_broadphase;
}

Sim::Sim(Engine &ctx, const Config &cfg, const WorldInit &)
    : WorldBase(ctx)
{
    uint32_t global_world = cfg.worldBase + (uint32_t)ctx.worldID().idx;

    initRandKey = rand::split_i(rand::initKey(cfg.seed), global_world);
    resetRng = RNG(rand::split_i(initRandKey, 0x7E5E7u));
    curWorldEpisode = 0;
    autoResetDenom = cfg.autoResetDenom;
    numExtra = (int32_t)cfg.numExtra < consts::maxExtra ?
        (int32_t)cfg.numExtra : consts::maxExtra;
    if (cfg.crowdedWorld >= 0 && (int32_t)global_world != cfg.crowdedWorld) {
        numExtra = 0;
    }
    hingeMode = cfg.hingeMode;
    overlapMode = cfg.overlapMode;

    ctx.singleton<WorldReset>().reset = 0;
    ctx.singleton<QueryProbe>() = QueryProbe { 0u, 0 };

    PhysicsSystem::init(ctx, cfg.rigidBodyObjMgr, consts::deltaT,
        consts::numPhysicsSubsteps, -9.8f * math::up,
        consts::maxRigidBodies - consts::maxExtra + numExtra);

    createPersistentEntities(ctx);
    initWorld(ctx);
}

#ifdef MADRONA_GPU_MODE
MADRONA_BUILD_MWGPU_ENTRY(Engine, Sim, Sim::Config, Sim::WorldInit);
#endif

}

#include "sim.hpp"

#ifdef MADRONA_GPU_MODE
#include <madrona/mw_gpu_entry.hpp>
#endif

using namespace madrona;
using namespace madrona::math;
using namespace madrona::phys;
using madrona::base::Rotation;
using madrona::base::Scale;
using madrona::base::ObjectID;

namespace bponly {

void Sim::registerTypes(ECSRegistry &registry, const Config &)
{
    base::registerTypes(registry);
    PhysicsSystem::registerTypes(registry);

    registry.registerComponent<Drift>();
    registry.registerComponent<RayFan>();
    registry.registerSingleton<StepCount>();

    registry.registerArchetype<Box>();
    registry.registerArchetype<Pillar>();
    registry.registerArchetype<Sensor>();

    registry.exportSingleton<StepCount>((uint32_t)ExportID::StepCount);
}

static inline void setupBody(Engine &ctx, Entity e, Vector3 pos, Diag3x3 scale,
                             ResponseType response)
{
    ObjectID obj_id { 0 };
    ctx.get<Position>(e) = pos;
    ctx.get<Rotation>(e) = Quat { 1, 0, 0, 0 };
    ctx.get<Scale>(e) = scale;
    ctx.get<ObjectID>(e) = obj_id;
    ctx.get<ResponseType>(e) = response;
    ctx.get<Velocity>(e) = Velocity { Vector3::zero(), Vector3::zero() };
    ctx.get<ExternalForce>(e) = Vector3::zero();
    ctx.get<ExternalTorque>(e) = Vector3::zero();
    ctx.get<broadphase::LeafID>(e) =
        PhysicsSystem::registerEntity(ctx, e, obj_id);
}

inline void driftSystem(Engine &,
                        Position &pos,
                        Velocity &vel,
                        Drift &drift)
{
    Vector3 p = pos;
    Vector3 v = drift.v;

    p += consts::deltaT * v;
    if (p.x < -consts::arena) { p.x = -consts::arena; v.x = -v.x; }
    if (p.x > consts::arena) { p.x = consts::arena; v.x = -v.x; }
    if (p.y < -consts::arena) { p.y = -consts::arena; v.y = -v.y; }
    if (p.y > consts::arena) { p.y = consts::arena; v.y = -v.y; }

    pos = p;
    drift.v = v;
    vel.linear = v;     // the BVH sweeps leaf boxes along the velocity
}

// every rebuildPeriod steps the world empties its BVH and registers its bodies
// again (in a different order): the tree is rebuilt on the next update
inline void reregisterSystem(Engine &ctx, StepCount &steps)
{
    Sim &sim = ctx.data();

    steps.n += 1;
    if (steps.n % consts::rebuildPeriod != 0) {
        return;
    }

    PhysicsSystem::reset(ctx);
    for (int32_t i = sim.numBoxes - 1; i >= 0; i--) {
        Entity e = sim.boxes[i];
        ctx.get<broadphase::LeafID>(e) =
            PhysicsSystem::registerEntity(ctx, e, ctx.get<ObjectID>(e));
    }
    for (int32_t i = 0; i < consts::numPillars; i++) {
        Entity e = sim.pillars[i];
        ctx.get<broadphase::LeafID>(e) =
            PhysicsSystem::registerEntity(ctx, e, ctx.get<ObjectID>(e));
    }
}

// sensors drift like the boxes (no body: they are not in the BVH)
inline void sensorDriftSystem(Engine &, Position &pos, Drift &drift, RayFan &)
{
    Vector3 p = pos;
    Vector3 v = drift.v;
    p += consts::deltaT * v;
    if (p.x < -consts::arena) { p.x = -consts::arena; v.x = -v.x; }
    if (p.x > consts::arena) { p.x = consts::arena; v.x = -v.x; }
    if (p.y < -consts::arena) { p.y = -consts::arena; v.y = -v.y; }
    if (p.y > consts::arena) { p.y = consts::arena; v.y = -v.y; }
    pos = p;
    drift.v = v;
}

// 32 rays from the sensor's position: a fan in the plane plus a tilt that
// differs per ray, so that rays leave through tops and sides of the boxes
inline void raySystem(Engine &ctx, const Position &pos, RayFan &fan)
{
    broadphase::BVH &bvh = ctx.singleton<broadphase::BVH>();
    Vector3 ray_o = pos;

#if defined(MADRONA_GPU_MODE) && !defined(SIM_PORTABLE)
    broadphase::BVH::RayGroupScratch *ray_scratch = broadphase::rayGroupScratch();
#endif
    auto trace = [&](int32_t i) {
        // (directions from integers: no transcendental functions, whose last
        // bit differs between libm and the device library)
        Vector3 ray_d = Vector3 {
            (float)((i * 7) % 11 - 5) + 0.5f,
            (float)((i * 3) % 13 - 6) + 0.25f,
            0.75f * (float)((i % 5) - 2) }.normalize();
        float hit_t;
        Vector3 hit_normal;
#if defined(MADRONA_GPU_MODE) && !defined(SIM_PORTABLE)
        Entity hit = bvh.traceRayShared(ray_scratch, ray_o, ray_d, &hit_t,
                                        &hit_normal, 40.f);
#else
        Entity hit = bvh.traceRay(ray_o, ray_d, &hit_t, &hit_normal, 40.f);
#endif
        if (hit == Entity::none()) {
            fan.hitT[i] = 0.f;
            fan.hitEntity[i] = -1;
            fan.hitNormal[i] = Vector3::zero();
        } else {
            fan.hitT[i] = hit_t;
            fan.hitEntity[i] = hit.id;
            fan.hitNormal[i] = hit_normal;
        }
    };

#ifdef MADRONA_GPU_MODE
    trace((int32_t)(threadIdx.x % 32));
#else
    for (int32_t i = 0; i < consts::raysPerSensor; i++) {
        trace(i);
    }
#endif
}

void Sim::setupTasks(TaskGraphManager &taskgraph_mgr, const Config &cfg)
{
    TaskGraphBuilder &builder = taskgraph_mgr.init(0);

    // last step's pairs stay readable until the next step starts
    auto cleanup =
        PhysicsSystem::setupStandaloneBroadphaseCleanupTasks(builder, {});

    auto drift = builder.addToGraph<ParallelForNode<Engine,
        driftSystem, Position, Velocity, Drift>>({cleanup});

    auto reregister = builder.addToGraph<ParallelForNode<Engine,
        reregisterSystem, StepCount>>({drift});

    auto bvh = PhysicsSystem::setupBroadphaseTasks(builder, {reregister});

    if ((cfg.flags & 1u) != 0) {
        auto sensor_drift = builder.addToGraph<ParallelForNode<Engine,
            sensorDriftSystem, Position, Drift, RayFan>>({bvh});
#ifdef MADRONA_GPU_MODE
        bvh = builder.addToGraph<CustomParallelForNode<Engine,
            raySystem, 32, 1, Position, RayFan>>({sensor_drift});
#else
        bvh = builder.addToGraph<ParallelForNode<Engine,
            raySystem, Position, RayFan>>({sensor_drift});
#endif
    }

    auto overlaps =
        PhysicsSystem::setupStandaloneBroadphaseOverlapTasks(builder, {bvh});

#ifdef MADRONA_GPU_MODE
    // group the pairs by world (stable) so they can be read per world
    overlaps = builder.addToGraph<
        SortArchetypeNode<CandidateTemporary, WorldID>>({overlaps});
#endif
    (void)overlaps;
}

Sim::Sim(Engine &ctx, const Config &cfg, const WorldInit &)
    : WorldBase(ctx)
{
    uint32_t global_world = cfg.worldBase + (uint32_t)ctx.worldID().idx;
    RNG rng(rand::split_i(rand::initKey(cfg.seed), global_world));

    ctx.singleton<StepCount>().n = 0;

    const bool ray_mode = (cfg.flags & 1u) != 0;
    // (ray mode: 10 .. 100 boxes, by the global world index)
    numBoxes = ray_mode ?
        10 + (int32_t)((global_world * 37u) % 91u) : consts::numBoxes;

    PhysicsSystem::init(ctx, cfg.rigidBodyObjMgr, consts::deltaT, 1,
                        -9.8f * math::up,
                        ray_mode ? consts::maxBoxes + consts::numPillars : 32);

    for (int32_t i = 0; i < consts::numPillars; i++) {
        pillars[i] = ctx.makeEntity<Pillar>();
        setupBody(ctx, pillars[i],
            Vector3 { (i & 1) ? 2.5f : -2.5f, (i & 2) ? 2.5f : -2.5f, 1.f },
            Diag3x3 { 1.f, 1.f, 2.f }, ResponseType::Static);
    }

    for (int32_t i = 0; i < numBoxes; i++) {
        boxes[i] = ctx.makeEntity<Box>();
        float size = 0.6f + rng.sampleUniform();
        setupBody(ctx, boxes[i],
            Vector3 {
                (rng.sampleUniform() * 2.f - 1.f) * consts::arena,
                (rng.sampleUniform() * 2.f - 1.f) * consts::arena,
                size * 0.5f,
            },
            Diag3x3 { size, size, size }, ResponseType::Dynamic);
        ctx.get<Drift>(boxes[i]).v = Vector3 {
            rng.sampleUniform() * 4.f - 2.f, rng.sampleUniform() * 4.f - 2.f, 0.f,
        };
    }

    if (ray_mode) {
        const int32_t num_sensors = 1 + (int32_t)(global_world % 3u);
        for (int32_t i = 0; i < num_sensors; i++) {
            Entity e = ctx.makeEntity<Sensor>();
            ctx.get<Position>(e) = Vector3 {
                (rng.sampleUniform() * 2.f - 1.f) * consts::arena,
                (rng.sampleUniform() * 2.f - 1.f) * consts::arena,
                0.3f + rng.sampleUniform(),
            };
            ctx.get<Drift>(e).v = Vector3 {
                rng.sampleUniform() * 3.f - 1.5f, rng.sampleUniform() * 3.f - 1.5f,
                0.f,
            };
            RayFan &fan = ctx.get<RayFan>(e);
            for (int32_t r = 0; r < consts::raysPerSensor; r++) {
                fan.hitT[r] = 0.f;
                fan.hitEntity[r] = -1;
                fan.hitNormal[r] = Vector3::zero();
            }
        }
    }
}

#ifdef MADRONA_GPU_MODE
MADRONA_BUILD_MWGPU_ENTRY(Engine, Sim, Sim::Config, Sim::WorldInit);
#endif

}

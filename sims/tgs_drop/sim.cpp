#include "sim.hpp"

#ifdef MADRONA_GPU_MODE
#include <madrona/mw_gpu_entry.hpp>
#endif

using namespace madrona;
using namespace madrona::math;
using namespace madrona::phys;
using madrona::base::Scale;
using madrona::base::ObjectID;

namespace tgsdrop {

static constexpr PhysicsSystem::Solver kSolver = PhysicsSystem::Solver::TGS;

void Sim::registerTypes(ECSRegistry &registry, const Config &)
{
    base::registerTypes(registry);
    PhysicsSystem::registerTypes(registry, kSolver);

    registry.registerComponent<Kick>();
    registry.registerSingleton<StepCount>();

    registry.registerArchetype<Body>();
    registry.registerArchetype<Anchor>();

    registry.exportSingleton<StepCount>((uint32_t)ExportID::StepCount);
}

static inline void setupBody(Engine &ctx, Entity e, Vector3 pos, Quat rot,
                             Diag3x3 scale, int32_t obj, ResponseType response)
{
    ObjectID obj_id { obj };
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

// a fresh force and torque per body and step
inline void kickSystem(Engine &,
                       Kick &kick,
                       ExternalForce &force,
                       ExternalTorque &torque)
{
    RNG rng(rand::split_i(kick.key, kick.step));
    kick.step += 1;

    force = Vector3 {
        rng.sampleUniform() * 40.f - 20.f,
        rng.sampleUniform() * 40.f - 20.f,
        rng.sampleUniform() * 30.f,
    };
    torque = Vector3 {
        rng.sampleUniform() * 6.f - 3.f,
        rng.sampleUniform() * 6.f - 3.f,
        rng.sampleUniform() * 6.f - 3.f,
    };
}

inline void countSystem(Engine &, StepCount &steps)
{
    steps.n += 1;
}

void Sim::setupTasks(TaskGraphManager &taskgraph_mgr, const Config &)
{
    TaskGraphBuilder &builder = taskgraph_mgr.init(0);

    auto kick = builder.addToGraph<ParallelForNode<Engine,
        kickSystem, Kick, ExternalForce, ExternalTorque>>({});
    auto count = builder.addToGraph<ParallelForNode<Engine,
        countSystem, StepCount>>({kick});

    auto bvh = PhysicsSystem::setupBroadphaseTasks(builder, {count});
    auto step = PhysicsSystem::setupPhysicsStepTasks(
        builder, {bvh}, consts::numSubsteps, kSolver);
    auto cleanup = PhysicsSystem::setupCleanupTasks(builder, {step});
    (void)cleanup;
}

Sim::Sim(Engine &ctx, const Config &cfg, const WorldInit &)
    : WorldBase(ctx)
{
    uint32_t global_world = cfg.worldBase + (uint32_t)ctx.worldID().idx;
    RandKey world_key = rand::split_i(rand::initKey(cfg.seed), global_world);
    RNG rng(world_key);

    ctx.singleton<StepCount>().n = 0;

    PhysicsSystem::init(ctx, cfg.rigidBodyObjMgr, consts::deltaT,
                        consts::numSubsteps, -9.8f * math::up, 32, kSolver);

    for (int32_t i = 0; i < consts::numAnchors; i++) {
        anchors[i] = ctx.makeEntity<Anchor>();
        setupBody(ctx, anchors[i], Vector3 { i == 0 ? -3.f : 3.f, 0.f, 1.f },
                  Quat { 1, 0, 0, 0 }, Diag3x3 { 1.f, 1.f, 2.f }, 0,
                  ResponseType::Static);
    }

    for (int32_t i = 0; i < consts::numBodies; i++) {
        bodies[i] = ctx.makeEntity<Body>();
        // a tilted start: rotation about a random axis
        Vector3 axis {
            rng.sampleUniform() * 2.f - 1.f, rng.sampleUniform() * 2.f - 1.f,
            rng.sampleUniform() * 2.f - 1.f,
        };
        float angle = rng.sampleUniform();
        Quat rot = Quat { 1.f, angle * axis.x, angle * axis.y, angle * axis.z }
            .normalize();
        float size = 0.5f + rng.sampleUniform();
        setupBody(ctx, bodies[i],
            Vector3 {
                rng.sampleUniform() * 8.f - 4.f, rng.sampleUniform() * 8.f - 4.f,
                2.f + rng.sampleUniform() * 3.f,
            },
            rot, Diag3x3 { size, size, size },
            // every third body is the anisotropic slab
            i % 3 == 2 ? 1 : 0,
            // every fourth is kinematic: no gravity, forces still act
            i % 4 == 3 ? ResponseType::Kinematic : ResponseType::Dynamic);
        ctx.get<Velocity>(bodies[i]).angular = Vector3 {
            rng.sampleUniform() * 4.f - 2.f, rng.sampleUniform() * 4.f - 2.f,
            rng.sampleUniform() * 4.f - 2.f,
        };
        ctx.get<Kick>(bodies[i]) = Kick {
            rand::split_i(world_key, 1000u + (uint32_t)i), 0u,
        };
    }
}

#ifdef MADRONA_GPU_MODE
MADRONA_BUILD_MWGPU_ENTRY(Engine, Sim, Sim::Config, Sim::WorldInit);
#endif

}

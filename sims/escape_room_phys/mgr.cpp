#include "escape_room_phys/sim.hpp"

#include <madrona/physics_loader.hpp>
#include <madrona/physics_assets.hpp>
#include <madrona/importer.hpp>
#include <madrona/stack_alloc.hpp>

struct SimTraits;
#include "common/sim_c_api.h"
#ifdef ESCPHYS_RENDER
#include "common/mesh_set.hpp"
#include "common/cpu_render_bridge.hpp"
#ifndef SIM_BACKEND_REF_CPU
#include "common/render_config.hpp"
#endif
#endif

#include <vector>
#include <string>
#include <memory>
#include <array>

namespace {

using namespace madrona;
using namespace madrona::phys;
using escphys::SimObject;

// A unit cube hull (quads, outward CCW) stands in for every mesh asset of the
// real Escape Room (no file importers here): instances scale it per body.
struct CubeMesh {
    math::Vector3 positions[8] = {
        { -0.5f, -0.5f, -0.5f }, { 0.5f, -0.5f, -0.5f },
        { 0.5f, 0.5f, -0.5f }, { -0.5f, 0.5f, -0.5f },
        { -0.5f, -0.5f, 0.5f }, { 0.5f, -0.5f, 0.5f },
        { 0.5f, 0.5f, 0.5f }, { -0.5f, 0.5f, 0.5f },
    };
    uint32_t indices[24] = {
        0, 3, 2, 1,     // -z
        4, 5, 6, 7,     // +z
        0, 1, 5, 4,     // -y
        2, 3, 7, 6,     // +y
        0, 4, 7, 3,     // -x
        1, 2, 6, 5,     // +x
    };
    uint32_t faceCounts[6] = { 4, 4, 4, 4, 4, 4 };
};

// Loaders stay alive for the life of the process: worlds keep pointing at the
// ObjectManager they own.
std::vector<std::unique_ptr<PhysicsLoader>> &loaders()
{
    static std::vector<std::unique_ptr<PhysicsLoader>> list;
    return list;
}

ObjectManager *loadPhysicsObjects(const SimCreateArgs &args)
{
#ifdef SIM_BACKEND_REF_CPU
    (void)args;
    auto loader = std::make_unique<PhysicsLoader>(ExecMode::CPU, 10);
#else
    auto loader = std::make_unique<PhysicsLoader>(ExecMode::CUDA, 10,
                                                  args.gpu_id);
#endif

    CubeMesh cube;
    imp::SourceMesh hull_mesh {};
    hull_mesh.positions = cube.positions;
    hull_mesh.indices = cube.indices;
    hull_mesh.faceCounts = cube.faceCounts;
    hull_mesh.numVertices = 8;
    hull_mesh.numFaces = 6;

    std::array<SourceCollisionPrimitive, (size_t)SimObject::NumObjects> prims {};
    std::array<SourceCollisionObject, (size_t)SimObject::NumObjects> objs {};

    auto setup_hull = [&](SimObject obj, float inv_mass,
                          RigidBodyFrictionData friction) {
        SourceCollisionPrimitive &prim = prims[(size_t)obj];
        prim.type = CollisionPrimitive::Type::Hull;
        prim.hullInput.hullIDX = 0;

        objs[(size_t)obj] = SourceCollisionObject {
            Span<const SourceCollisionPrimitive>(&prim, 1),
            inv_mass,
            friction,
        };
    };

    setup_hull(SimObject::Cube, 0.075f, { 0.5f, 0.75f });
    setup_hull(SimObject::Wall, 0.f, { 0.5f, 0.5f });
    setup_hull(SimObject::Door, 0.f, { 0.5f, 0.5f });
    setup_hull(SimObject::Agent, 1.f, { 0.5f, 0.5f });
    setup_hull(SimObject::Button, 1.f, { 0.5f, 0.5f });

    {
        SourceCollisionPrimitive &prim = prims[(size_t)SimObject::Plane];
        prim.type = CollisionPrimitive::Type::Plane;
        objs[(size_t)SimObject::Plane] = SourceCollisionObject {
            Span<const SourceCollisionPrimitive>(&prim, 1),
            0.f,
            { 0.5f, 0.5f },
        };
    }

    StackAlloc tmp_alloc;
    RigidBodyAssets rigid_body_assets;
    CountT num_rigid_body_data_bytes;
    void *rigid_body_data = RigidBodyAssets::processRigidBodyAssets(
        Span<const imp::SourceMesh>(&hull_mesh, 1),
        Span<const SourceCollisionObject>(objs.data(), (CountT)objs.size()),
        false, tmp_alloc, &rigid_body_assets, &num_rigid_body_data_bytes);

    if (rigid_body_data == nullptr) {
        FATAL("Invalid collision hull input");
    }

    // agents turn about z only: infinite inertia about x and y
    rigid_body_assets.metadatas[(size_t)SimObject::Agent]
        .mass.invInertiaTensor.x = 0.f;
    rigid_body_assets.metadatas[(size_t)SimObject::Agent]
        .mass.invInertiaTensor.y = 0.f;

    loader->loadRigidBodies(rigid_body_assets);
    free(rigid_body_data);

    ObjectManager *mgr = &loader->getObjectManager();
    loaders().push_back(std::move(loader));
    return mgr;
}

#ifdef ESCPHYS_RENDER
// What the ray caster draws for each SimObject: boxes for the cube-hull bodies
// (instances scale them), a 352-triangle ellipsoid for the agents, a quad for
// the floor.
const simmesh::MeshSet &meshes()
{
    static const simmesh::MeshSet set = [] {
        simmesh::MeshSet m;
        const int32_t cube = m.material(0.8f, 0.45f, 0.2f);
        const int32_t wall = m.material(0.55f, 0.55f, 0.6f);
        const int32_t door = m.material(0.2f, 0.6f, 0.25f);
        const int32_t agent = m.material(0.9f, 0.85f, 0.2f);
        const int32_t button = m.material(0.85f, 0.15f, 0.15f);
        const int32_t floor = m.material(0.35f, 0.3f, 0.28f);
        const int32_t mats[] = { cube, wall, door, agent, button, floor };
        for (int32_t obj = 0; obj < (int32_t)SimObject::NumObjects; obj++) {
            if (obj == (int32_t)SimObject::Agent) {
                m.ellipsoid(0.f, 0.f, 0.f, 0.5f, 0.5f, 0.5f, 16, 12);
            } else if (obj == (int32_t)SimObject::Plane) {
                const uint32_t a = m.vert(-40.f, -20.f, 0.f);
                m.vert(40.f, -20.f, 0.f);
                m.vert(40.f, 60.f, 0.f);
                m.vert(-40.f, 60.f, 0.f);
                m.quad(a, a + 1, a + 2, a + 3);
            } else {
                m.box(-0.5f, -0.5f, -0.5f, 0.5f, 0.5f, 0.5f);
            }
            m.endObject(mats[obj]);
        }
        return m;
    }();
    return set;
}

uint32_t g_resolution = 0;
#ifdef SIM_BACKEND_REF_CPU
// floor + 4 borders + 2 agents + 3 rooms x 9 = 34 instances, 2 views per world
constexpr uint32_t kMaxRecordsPerWorld = 64;
simmgr::CpuRenderBridge *g_bridge = nullptr;
#endif
#endif

}

struct SimTraits {
    using Sim = escphys::Sim;
    using Engine = escphys::Engine;

    static constexpr uint32_t numExports =
        (uint32_t)escphys::ExportID::NumExports;
    static constexpr uint32_t numTaskGraphs = 1;

    // flags: low 16 bits = autoResetDenom (0 disables random resets); render
    // variant: bits 16-23 = output resolution (0: 64), bit 24 = depth only,
    // bit 25 = the sun casts shadows, bit 26 = no ray caster (render-prep
    // systems only: without render-target entities the entity ids are the CPU
    // backend's, whose CPU mode has no ray caster either)
    static Sim::Config makeConfig(const SimCreateArgs &args)
    {
#if defined(ESCPHYS_RENDER) && defined(SIM_BACKEND_REF_CPU)
        delete g_bridge;
        g_bridge = new simmgr::CpuRenderBridge(
            args.num_worlds, kMaxRecordsPerWorld, 64,
            escphys::consts::numAgents, kMaxRecordsPerWorld);
#endif
        return Sim::Config {
            args.seed, args.world_base, args.flags & 0xFFFFu,
            loadPhysicsObjects(args),
#ifdef ESCPHYS_RENDER
            (args.flags >> 25) & 1u,
#ifdef SIM_BACKEND_REF_CPU
            &g_bridge->bridge,
#else
            nullptr,
#endif
#endif
        };
    }

#if defined(ESCPHYS_RENDER) && defined(SIM_BACKEND_REF_CPU)
    static void preStep() { g_bridge->beginStep(); }
#endif

#ifdef ESCPHYS_RENDER
    static const simmesh::MeshSet &renderMeshes() { return meshes(); }

#ifndef SIM_BACKEND_REF_CPU
    static madrona::Optional<madrona::CudaBatchRenderConfig> renderConfig(
        const SimCreateArgs &args)
    {
        madrona::CudaBatchRenderConfig cfg {};
        cfg.renderMode = (args.flags >> 24) & 1u ?
            madrona::CudaBatchRenderConfig::RenderMode::Depth :
            madrona::CudaBatchRenderConfig::RenderMode::RGBD;
        if (((args.flags >> 26) & 1u) != 0u) {
            g_resolution = 0;
            return madrona::Optional<madrona::CudaBatchRenderConfig>::none();
        }
        cfg.renderResolution = (args.flags >> 16) & 0xFFu;
        if (cfg.renderResolution == 0) cfg.renderResolution = 64;
        g_resolution = cfg.renderResolution;
        cfg.maxViewsPerWorld = escphys::consts::numAgents;
        simmgr::meshesToRenderConfig(meshes(), cfg);
        return madrona::Optional<madrona::CudaBatchRenderConfig>::make(cfg);
    }
#endif
#endif

    static void makeInits(const SimCreateArgs &, Sim::WorldInit *) {}

    template <typename T>
    static void describeTensors(T &out, uint32_t num_worlds);
    template <typename T>
    static void describeColumns(T &cols);
};

#include "common/mgr_impl.inl"

template <typename T>
void SimTraits::describeTensors(T &out, uint32_t num_worlds)
{
    using escphys::ExportID;
    namespace c = escphys::consts;
    int64_t W = num_worlds;
    int64_t A = c::numAgents;
    out.push_back({ "reset", SIM_I32, { W, 1 }, (uint32_t)ExportID::Reset });
    out.push_back({ "action", SIM_I32, { W, A, 4 }, (uint32_t)ExportID::Action });
    out.push_back({ "reward", SIM_F32, { W, A, 1 }, (uint32_t)ExportID::Reward });
    out.push_back({ "done", SIM_I32, { W, A, 1 }, (uint32_t)ExportID::Done });
    out.push_back({ "self_obs", SIM_F32, { W, A, 8 },
                    (uint32_t)ExportID::SelfObservation });
    out.push_back({ "partner_obs", SIM_F32, { W, A, 3 },
                    (uint32_t)ExportID::PartnerObservation });
    out.push_back({ "room_ent_obs", SIM_F32,
                    { W, A, c::numCubesPerRoom + c::numButtonsPerRoom + 1, 3 },
                    (uint32_t)ExportID::RoomEntityObservations });
    out.push_back({ "door_obs", SIM_F32, { W, A, 3 },
                    (uint32_t)ExportID::DoorObservation });
    out.push_back({ "lidar", SIM_F32, { W, A, c::numLidarSamples, 2 },
                    (uint32_t)ExportID::Lidar });
    out.push_back({ "steps_remaining", SIM_I32, { W, A, 1 },
                    (uint32_t)ExportID::StepsRemaining });
#if defined(ESCPHYS_RENDER) && !defined(SIM_BACKEND_REF_CPU)
    const int64_t R = g_resolution;
    if (R != 0) {
        out.push_back({ "rgb", SIM_U8, { W * A, R, R, 4 }, (uint32_t)ExportID::RGB });
        out.push_back({ "depth", SIM_F32, { W * A, R, R }, (uint32_t)ExportID::Depth });
    }
#endif
}

template <typename T>
void SimTraits::describeColumns(T &cols)
{
    using namespace escphys;
    using madrona::Entity;
    using madrona::phys::broadphase::LeafID;

    cols.template add<Agent, Entity>("Agent.Entity", false);
    cols.template add<Agent, Position>("Agent.Position", true);
    cols.template add<Agent, Rotation>("Agent.Rotation", true);
    cols.template add<Agent, Velocity>("Agent.Velocity", true);
    cols.template add<Agent, LeafID>("Agent.LeafID", false);
    cols.template add<Agent, ExternalForce>("Agent.ExternalForce", true);
    cols.template add<Agent, Action>("Agent.Action", false);
    cols.template add<Agent, Reward>("Agent.Reward", true);
    cols.template add<Agent, Done>("Agent.Done", false);
    cols.template add<Agent, SelfObservation>("Agent.SelfObservation", true);
    cols.template add<Agent, PartnerObservation>("Agent.PartnerObservation", true);
    cols.template add<Agent, RoomEntityObservations>(
        "Agent.RoomEntityObservations", true);
    cols.template add<Agent, DoorObservation>("Agent.DoorObservation", true);
    cols.template add<Agent, Lidar>("Agent.Lidar", true);
    cols.template add<Agent, StepsRemaining>("Agent.StepsRemaining", false);
    cols.template add<Agent, Progress>("Agent.Progress", true);
    cols.template add<Agent, GrabState>("Agent.GrabState", false);

    cols.template add<PhysicsEntity, Entity>("PhysicsEntity.Entity", false);
    cols.template add<PhysicsEntity, Position>("PhysicsEntity.Position", true);
    cols.template add<PhysicsEntity, Rotation>("PhysicsEntity.Rotation", true);
    cols.template add<PhysicsEntity, Velocity>("PhysicsEntity.Velocity", true);
    cols.template add<PhysicsEntity, LeafID>("PhysicsEntity.LeafID", false);
    cols.template add<PhysicsEntity, EntityType>("PhysicsEntity.EntityType", false);

    cols.template add<DoorEntity, Entity>("DoorEntity.Entity", false);
    cols.template add<DoorEntity, Position>("DoorEntity.Position", true);
    cols.template add<DoorEntity, OpenState>("DoorEntity.OpenState", false);
    cols.template add<DoorEntity, DoorProperties>("DoorEntity.DoorProperties", false);

    cols.template add<ButtonEntity, Entity>("ButtonEntity.Entity", false);
    cols.template add<ButtonEntity, Position>("ButtonEntity.Position", true);
    cols.template add<ButtonEntity, ButtonState>("ButtonEntity.ButtonState", false);
#ifdef ESCPHYS_RENDER
    using namespace madrona::render;
    // the light table is filled the same way on both backends
    cols.template add<LightArchetype, LightDesc>("Light.LightDesc", false);
    cols.template add<PhysicsEntity, Renderable>("PhysicsEntity.Renderable", false);
    cols.template add<ButtonEntity, Renderable>("ButtonEntity.Renderable", false);
    cols.template add<Agent, Renderable>("Agent.Renderable", false);
    cols.template add<Agent, RenderCamera>("Agent.RenderCamera", false);
#ifndef SIM_BACKEND_REF_CPU
    // GPU mode only: the render entities' rows ARE the renderer's records
    cols.template add<RenderableArchetype, InstanceData>("Renderable.InstanceData", false);
    cols.template add<RenderableArchetype, MortonCode>("Renderable.MortonCode", false);
    cols.template add<RenderCameraArchetype, PerspectiveCameraData>("Camera.PerspectiveCameraData", false);
    cols.template add<RenderCameraArchetype, RenderOutputIndex>("Camera.RenderOutputIndex", false);
#endif
#endif
}

#ifdef ESCPHYS_RENDER
// CPU mode: the instance (kind 0, 64 B) / view (kind 1, 48 B) records the
// reference appended to its bridge during the last step, arrival order, + the
// (world << 32 | entity id) key of each.  Returns the count (-1 on HIP: there
// the render entities' rows are the records).
extern "C" SIM_API int64_t render_prep_bridge_records(int32_t kind, void *dst,
                                                      uint64_t *keys_dst,
                                                      uint64_t max_records)
{
#ifdef SIM_BACKEND_REF_CPU
    if (g_bridge == nullptr) return -1;
    return g_bridge->records(kind, dst, keys_dst, max_records);
#else
    (void)kind; (void)dst; (void)keys_dst; (void)max_records;
    return -1;
#endif
}
#endif

#include "hideseek/sim.hpp"

#include <madrona/physics_loader.hpp>
#include <madrona/physics_assets.hpp>
#include <madrona/importer.hpp>
#include <madrona/stack_alloc.hpp>

struct SimTraits;
#include "common/sim_c_api.h"

#include <vector>
#include <string>
#include <memory>
#include <array>

namespace {

using namespace madrona;
using namespace madrona::phys;
using hideseek::SimObject;

// Two source hulls (no file importers here), scaled per instance: a unit cube
// and a unit wedge (right-triangle cross-section: the slope rises from y = -0.5
// to y = +0.5; 6 vertices, 3 quads + 2 triangles).
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

struct WedgeMesh {
    math::Vector3 positions[6] = {
        { -0.5f, -0.5f, -0.5f }, { 0.5f, -0.5f, -0.5f },
        { 0.5f, 0.5f, -0.5f }, { -0.5f, 0.5f, -0.5f },
        { -0.5f, 0.5f, 0.5f }, { 0.5f, 0.5f, 0.5f },
    };
    uint32_t indices[18] = {
        0, 3, 2, 1,     // -z
        2, 3, 4, 5,     // +y
        0, 1, 5, 4,     // slope (-y, +z)
        0, 4, 3,        // -x
        1, 2, 5,        // +x
    };
    uint32_t faceCounts[5] = { 4, 4, 4, 3, 3 };
};

// An axis-aligned box hull [lo, hi] (same face / winding layout as CubeMesh).
struct BoxMesh {
    math::Vector3 positions[8];
    uint32_t indices[24] = {
        0, 3, 2, 1,  4, 5, 6, 7,  0, 1, 5, 4,  2, 3, 7, 6,  0, 4, 7, 3,
        1, 2, 6, 5,
    };
    uint32_t faceCounts[6] = { 4, 4, 4, 4, 4, 4 };

    BoxMesh(math::Vector3 lo, math::Vector3 hi)
        : positions {
              { lo.x, lo.y, lo.z }, { hi.x, lo.y, lo.z },
              { hi.x, hi.y, lo.z }, { lo.x, hi.y, lo.z },
              { lo.x, lo.y, hi.z }, { hi.x, lo.y, hi.z },
              { hi.x, hi.y, hi.z }, { lo.x, hi.y, hi.z },
          }
    {}
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
    WedgeMesh wedge;
    // the L-shaped block: a bar along x and a post standing on one end
    BoxMesh bar({ -1.f, -0.3f, -0.5f }, { 1.f, 0.3f, 0.1f });
    BoxMesh post({ 0.4f, -0.3f, 0.1f }, { 1.f, 0.3f, 0.9f });
    std::array<imp::SourceMesh, 4> hull_meshes {};
    hull_meshes[2].positions = bar.positions;
    hull_meshes[2].indices = bar.indices;
    hull_meshes[2].faceCounts = bar.faceCounts;
    hull_meshes[2].numVertices = 8;
    hull_meshes[2].numFaces = 6;
    hull_meshes[3].positions = post.positions;
    hull_meshes[3].indices = post.indices;
    hull_meshes[3].faceCounts = post.faceCounts;
    hull_meshes[3].numVertices = 8;
    hull_meshes[3].numFaces = 6;
    hull_meshes[0].positions = cube.positions;
    hull_meshes[0].indices = cube.indices;
    hull_meshes[0].faceCounts = cube.faceCounts;
    hull_meshes[0].numVertices = 8;
    hull_meshes[0].numFaces = 6;
    hull_meshes[1].positions = wedge.positions;
    hull_meshes[1].indices = wedge.indices;
    hull_meshes[1].faceCounts = wedge.faceCounts;
    hull_meshes[1].numVertices = 6;
    hull_meshes[1].numFaces = 5;

    std::array<SourceCollisionPrimitive, (size_t)SimObject::NumObjects> prims {};
    std::array<SourceCollisionObject, (size_t)SimObject::NumObjects> objs {};

    auto setup_hull = [&](SimObject obj, uint32_t hull_idx, float inv_mass,
                          RigidBodyFrictionData friction) {
        SourceCollisionPrimitive &prim = prims[(size_t)obj];
        prim.type = CollisionPrimitive::Type::Hull;
        prim.hullInput.hullIDX = hull_idx;

        objs[(size_t)obj] = SourceCollisionObject {
            Span<const SourceCollisionPrimitive>(&prim, 1),
            inv_mass,
            friction,
        };
    };

    setup_hull(SimObject::Box, 0, 0.1f, { 0.5f, 0.75f });
    setup_hull(SimObject::LongBox, 0, 0.06f, { 0.5f, 0.75f });
    setup_hull(SimObject::Ramp, 1, 0.05f, { 0.6f, 0.8f });
    setup_hull(SimObject::Wall, 0, 0.f, { 0.5f, 0.5f });
    setup_hull(SimObject::Agent, 0, 1.f, { 0.5f, 0.5f });

    std::array<SourceCollisionPrimitive, 2> l_prims {};
    for (uint32_t i = 0; i < 2; i++) {
        l_prims[i].type = CollisionPrimitive::Type::Hull;
        l_prims[i].hullInput.hullIDX = 2 + i;
    }
    objs[(size_t)SimObject::LBlock] = SourceCollisionObject {
        Span<const SourceCollisionPrimitive>(l_prims.data(), 2),
        0.08f,
        { 0.5f, 0.75f },
    };

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
        Span<const imp::SourceMesh>(hull_meshes.data(),
                                    (CountT)hull_meshes.size()),
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

}

struct SimTraits {
    using Sim = hideseek::Sim;
    using Engine = hideseek::Engine;

    static constexpr uint32_t numExports =
        (uint32_t)hideseek::ExportID::NumExports;
    static constexpr uint32_t numTaskGraphs = 1;

    // flags: low 16 bits = autoResetDenom (0 disables random resets)
    static Sim::Config makeConfig(const SimCreateArgs &args)
    {
        return Sim::Config {
            args.seed, args.world_base, args.flags & 0xFFFFu,
            loadPhysicsObjects(args),
        };
    }

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
    using hideseek::ExportID;
    namespace c = hideseek::consts;
    int64_t W = num_worlds;
    int64_t A = c::numAgents;
    out.push_back({ "reset", SIM_I32, { W, 1 }, (uint32_t)ExportID::Reset });
    out.push_back({ "action", SIM_I32, { W, A, 4 }, (uint32_t)ExportID::Action });
    out.push_back({ "reward", SIM_F32, { W, A, 1 }, (uint32_t)ExportID::Reward });
    out.push_back({ "done", SIM_I32, { W, A, 1 }, (uint32_t)ExportID::Done });
    out.push_back({ "self_obs", SIM_F32, { W, A, 8 },
                    (uint32_t)ExportID::SelfObservation });
    out.push_back({ "agent_obs", SIM_F32, { W, A, A - 1, 4 },
                    (uint32_t)ExportID::AgentObservations });
    out.push_back({ "box_obs", SIM_F32, { W, A, c::numBoxes, 5 },
                    (uint32_t)ExportID::BoxObservations });
    out.push_back({ "ramp_obs", SIM_F32, { W, A, c::numRamps, 5 },
                    (uint32_t)ExportID::RampObservations });
    out.push_back({ "lidar", SIM_F32, { W, A, c::numLidarSamples, 2 },
                    (uint32_t)ExportID::Lidar });
    out.push_back({ "steps_remaining", SIM_I32, { W, A, 1 },
                    (uint32_t)ExportID::StepsRemaining });
}

template <typename T>
void SimTraits::describeColumns(T &cols)
{
    using namespace hideseek;
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
    cols.template add<Agent, AgentObservations>("Agent.AgentObservations", true);
    cols.template add<Agent, BoxObservations>("Agent.BoxObservations", true);
    cols.template add<Agent, RampObservations>("Agent.RampObservations", true);
    cols.template add<Agent, Lidar>("Agent.Lidar", true);
    cols.template add<Agent, StepsRemaining>("Agent.StepsRemaining", false);
    cols.template add<Agent, Visibility>("Agent.Visibility", false);

    cols.template add<MovableObject, Entity>("MovableObject.Entity", false);
    cols.template add<MovableObject, Position>("MovableObject.Position", true);
    cols.template add<MovableObject, Rotation>("MovableObject.Rotation", true);
    cols.template add<MovableObject, Velocity>("MovableObject.Velocity", true);
    cols.template add<MovableObject, ResponseType>(
        "MovableObject.ResponseType", false);
    cols.template add<MovableObject, LeafID>("MovableObject.LeafID", false);
    cols.template add<MovableObject, LockState>("MovableObject.LockState", false);

    cols.template add<StaticObject, Entity>("StaticObject.Entity", false);
    cols.template add<StaticObject, Position>("StaticObject.Position", true);
    cols.template add<StaticObject, Scale>("StaticObject.Scale", true);
    cols.template add<StaticObject, LeafID>("StaticObject.LeafID", false);
}

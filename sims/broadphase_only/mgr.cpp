#include "broadphase_only/sim.hpp"

#include <madrona/physics_loader.hpp>
#include <madrona/physics_assets.hpp>
#include <madrona/importer.hpp>
#include <madrona/stack_alloc.hpp>

#ifdef SIM_BACKEND_REF_CPU
// the candidate archetype is private to the reference's physics sources
#include <physics_impl.hpp>
#endif

struct SimTraits;
#include "common/sim_c_api.h"

#include <vector>
#include <string>
#include <memory>

namespace {

using namespace madrona;
using namespace madrona::phys;

std::vector<std::unique_ptr<PhysicsLoader>> &loaders()
{
    static std::vector<std::unique_ptr<PhysicsLoader>> list;
    return list;
}

ObjectManager *loadCube(const SimCreateArgs &args)
{
#ifdef SIM_BACKEND_REF_CPU
    (void)args;
    auto loader = std::make_unique<PhysicsLoader>(ExecMode::CPU, 4);
#else
    auto loader = std::make_unique<PhysicsLoader>(ExecMode::CUDA, 4, args.gpu_id);
#endif

    math::Vector3 positions[8] = {
        { -0.5f, -0.5f, -0.5f }, { 0.5f, -0.5f, -0.5f },
        { 0.5f, 0.5f, -0.5f }, { -0.5f, 0.5f, -0.5f },
        { -0.5f, -0.5f, 0.5f }, { 0.5f, -0.5f, 0.5f },
        { 0.5f, 0.5f, 0.5f }, { -0.5f, 0.5f, 0.5f },
    };
    uint32_t indices[24] = {
        0, 3, 2, 1, 4, 5, 6, 7, 0, 1, 5, 4, 2, 3, 7, 6, 0, 4, 7, 3, 1, 2, 6, 5,
    };
    uint32_t face_counts[6] = { 4, 4, 4, 4, 4, 4 };

    imp::SourceMesh hull_mesh {};
    hull_mesh.positions = positions;
    hull_mesh.indices = indices;
    hull_mesh.faceCounts = face_counts;
    hull_mesh.numVertices = 8;
    hull_mesh.numFaces = 6;

    SourceCollisionPrimitive prim {};
    prim.type = CollisionPrimitive::Type::Hull;
    prim.hullInput.hullIDX = 0;
    SourceCollisionObject obj {
        Span<const SourceCollisionPrimitive>(&prim, 1), 1.f, { 0.5f, 0.5f },
    };

    StackAlloc tmp_alloc;
    RigidBodyAssets assets;
    CountT num_bytes;
    void *data = RigidBodyAssets::processRigidBodyAssets(
        Span<const imp::SourceMesh>(&hull_mesh, 1),
        Span<const SourceCollisionObject>(&obj, 1),
        false, tmp_alloc, &assets, &num_bytes);
    if (data == nullptr) {
        FATAL("Invalid collision hull input");
    }

    loader->loadRigidBodies(assets);
    free(data);

    ObjectManager *mgr = &loader->getObjectManager();
    loaders().push_back(std::move(loader));
    return mgr;
}

}

struct SimTraits {
    using Sim = bponly::Sim;
    using Engine = bponly::Engine;

    static constexpr uint32_t numExports = (uint32_t)bponly::ExportID::NumExports;
    static constexpr uint32_t numTaskGraphs = 1;

    static Sim::Config makeConfig(const SimCreateArgs &args)
    {
        return Sim::Config { args.seed, args.world_base, loadCube(args), args.flags };
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
    out.push_back({ "step_count", SIM_I32, { (int64_t)num_worlds, 1 },
                    (uint32_t)bponly::ExportID::StepCount });
}

template <typename T>
void SimTraits::describeColumns(T &cols)
{
    using namespace bponly;
    using madrona::Entity;
    using madrona::phys::CandidateTemporary;
    using madrona::phys::CandidateCollision;

    cols.template add<Box, Entity>("Box.Entity", false);
    cols.template add<Box, Position>("Box.Position", true);
    cols.template add<Box, madrona::phys::broadphase::LeafID>("Box.LeafID", false);
    cols.template add<Pillar, Entity>("Pillar.Entity", false);
    cols.template add<Sensor, Position>("Sensor.Position", true);
    cols.template add<Sensor, RayFan>("Sensor.RayFan", false);
    cols.template add<CandidateTemporary, CandidateCollision>(
        "Candidates.CandidateCollision", false);
}

#include "tgs_drop/sim.hpp"

#include <madrona/physics_loader.hpp>
#include <madrona/physics_assets.hpp>
#include <madrona/importer.hpp>
#include <madrona/stack_alloc.hpp>

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

// object 0: a unit cube; object 1: a slab (2 x 1 x 0.5) with three different
// principal moments, so that the gyroscopic term of the integrator is not zero
ObjectManager *loadObjects(const SimCreateArgs &args)
{
#ifdef SIM_BACKEND_REF_CPU
    (void)args;
    auto loader = std::make_unique<PhysicsLoader>(ExecMode::CPU, 4);
#else
    auto loader = std::make_unique<PhysicsLoader>(ExecMode::CUDA, 4, args.gpu_id);
#endif

    auto box = [](float x, float y, float z, math::Vector3 *out) {
        const float sx[8] = { -1, 1, 1, -1, -1, 1, 1, -1 };
        const float sy[8] = { -1, -1, 1, 1, -1, -1, 1, 1 };
        const float sz[8] = { -1, -1, -1, -1, 1, 1, 1, 1 };
        for (int i = 0; i < 8; i++) {
            out[i] = math::Vector3 { sx[i] * x, sy[i] * y, sz[i] * z };
        }
    };
    math::Vector3 cube_pos[8], slab_pos[8];
    box(0.5f, 0.5f, 0.5f, cube_pos);
    box(1.f, 0.5f, 0.25f, slab_pos);
    uint32_t indices[24] = {
        0, 3, 2, 1, 4, 5, 6, 7, 0, 1, 5, 4, 2, 3, 7, 6, 0, 4, 7, 3, 1, 2, 6, 5,
    };
    uint32_t face_counts[6] = { 4, 4, 4, 4, 4, 4 };

    imp::SourceMesh meshes[2] {};
    for (int m = 0; m < 2; m++) {
        meshes[m].positions = m == 0 ? cube_pos : slab_pos;
        meshes[m].indices = indices;
        meshes[m].faceCounts = face_counts;
        meshes[m].numVertices = 8;
        meshes[m].numFaces = 6;
    }

    SourceCollisionPrimitive prims[2] {};
    SourceCollisionObject objs[2] {};
    for (int m = 0; m < 2; m++) {
        prims[m].type = CollisionPrimitive::Type::Hull;
        prims[m].hullInput.hullIDX = (uint32_t)m;
        objs[m] = SourceCollisionObject {
            Span<const SourceCollisionPrimitive>(&prims[m], 1),
            m == 0 ? 1.f : 0.4f, { 0.5f, 0.5f },
        };
    }

    StackAlloc tmp_alloc;
    RigidBodyAssets assets;
    CountT num_bytes;
    void *data = RigidBodyAssets::processRigidBodyAssets(
        Span<const imp::SourceMesh>(meshes, 2),
        Span<const SourceCollisionObject>(objs, 2),
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
    using Sim = tgsdrop::Sim;
    using Engine = tgsdrop::Engine;

    static constexpr uint32_t numExports = (uint32_t)tgsdrop::ExportID::NumExports;
    static constexpr uint32_t numTaskGraphs = 1;

    static Sim::Config makeConfig(const SimCreateArgs &args)
    {
        return Sim::Config { args.seed, args.world_base, loadObjects(args) };
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
                    (uint32_t)tgsdrop::ExportID::StepCount });
}

template <typename T>
void SimTraits::describeColumns(T &cols)
{
    using namespace tgsdrop;
    using madrona::Entity;
    using namespace madrona::phys;

    cols.template add<Body, Entity>("Body.Entity", false);
    cols.template add<Body, Position>("Body.Position", true);
    cols.template add<Body, Rotation>("Body.Rotation", true);
    cols.template add<Body, Velocity>("Body.Velocity", true);
    cols.template add<Body, ExternalForce>("Body.ExternalForce", true);
    cols.template add<Body, ExternalTorque>("Body.ExternalTorque", true);
    cols.template add<Body, broadphase::LeafID>("Body.LeafID", false);
    cols.template add<Anchor, Entity>("Anchor.Entity", false);
    cols.template add<Anchor, Position>("Anchor.Position", true);
    cols.template add<Anchor, Velocity>("Anchor.Velocity", true);
}

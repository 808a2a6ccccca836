#include "ball_pit/sim.hpp"

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
using ballpit::SimObject;

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

// A regular n-gon prism ("coin") of radius r and height h, axis along z.
struct PrismMesh {
    std::vector<math::Vector3> positions;
    std::vector<uint32_t> indices;
    std::vector<uint32_t> faceCounts;

    PrismMesh(uint32_t n, float r, float h)
    {
        for (uint32_t ring = 0; ring < 2; ring++) {
            for (uint32_t i = 0; i < n; i++) {
                float angle = 6.2831853f * (float)i / (float)n;
                positions.push_back(math::Vector3 {
                    r * cosf(angle), r * sinf(angle),
                    ring == 0 ? -0.5f * h : 0.5f * h,
                });
            }
        }
        // bottom cap (seen from below), top cap, then the sides
        indices.push_back(0);
        for (uint32_t i = n - 1; i >= 1; i--) {
            indices.push_back(i);
        }
        faceCounts.push_back(n);
        for (uint32_t i = 0; i < n; i++) {
            indices.push_back(n + i);
        }
        faceCounts.push_back(n);
        for (uint32_t i = 0; i < n; i++) {
            uint32_t j = (i + 1) % n;
            indices.push_back(i);
            indices.push_back(j);
            indices.push_back(n + j);
            indices.push_back(n + i);
            faceCounts.push_back(4);
        }
    }
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
    BoxMesh bar({ -1.f, -0.3f, -0.3f }, { 1.f, 0.3f, 0.3f });
    BoxMesh post({ 0.4f, -0.3f, 0.3f }, { 1.f, 0.3f, 1.2f });
    PrismMesh coin12(12, 0.9f, 0.5f);
    PrismMesh coin16(16, 1.f, 0.5f);
    std::array<imp::SourceMesh, 6> hull_meshes {};
    auto set_prism = [&](uint32_t idx, PrismMesh &m) {
        hull_meshes[idx].positions = m.positions.data();
        hull_meshes[idx].indices = m.indices.data();
        hull_meshes[idx].faceCounts = m.faceCounts.data();
        hull_meshes[idx].numVertices = (uint32_t)m.positions.size();
        hull_meshes[idx].numFaces = (uint32_t)m.faceCounts.size();
    };
    set_prism(4, coin12);
    set_prism(5, coin16);
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

    setup_hull(SimObject::Box, 0, 0.2f, { 0.5f, 0.75f });
    setup_hull(SimObject::Coin12, 4, 0.3f, { 0.6f, 0.8f });
    setup_hull(SimObject::Coin16, 5, 0.25f, { 0.6f, 0.8f });
    setup_hull(SimObject::Wall, 0, 0.f, { 0.5f, 0.5f });

    std::array<SourceCollisionPrimitive, 2> l_prims {};
    for (uint32_t i = 0; i < 2; i++) {
        l_prims[i].type = CollisionPrimitive::Type::Hull;
        l_prims[i].hullInput.hullIDX = 2 + i;
    }
    objs[(size_t)SimObject::LBlock] = SourceCollisionObject {
        Span<const SourceCollisionPrimitive>(l_prims.data(), 2),
        0.2f,
        { 0.5f, 0.7f },
    };

    {
        SourceCollisionPrimitive &prim = prims[(size_t)SimObject::Sphere];
        prim.type = CollisionPrimitive::Type::Sphere;
        prim.sphere.radius = 1.f;
        objs[(size_t)SimObject::Sphere] = SourceCollisionObject {
            Span<const SourceCollisionPrimitive>(&prim, 1),
            0.25f,
            { 0.5f, 0.6f },
        };
    }

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

    loader->loadRigidBodies(rigid_body_assets);
    free(rigid_body_data);

    ObjectManager *mgr = &loader->getObjectManager();
    loaders().push_back(std::move(loader));
    return mgr;
}

}

struct SimTraits {
    using Sim = ballpit::Sim;
    using Engine = ballpit::Engine;

    static constexpr uint32_t numExports =
        (uint32_t)ballpit::ExportID::NumExports;
    static constexpr uint32_t numTaskGraphs = 1;

    // flags: low 16 bits = autoResetDenom (0 disables random resets), bits
    // 16-23 = extra bodies per world (crowd mode), bit 24 = every other joint
    // is a hinge, bit 25 = checkEntityAABBOverlap steers the kicks, bit 26 = only
    // ONE world is crowded, the one whose global index is in bits 27-31
    static Sim::Config makeConfig(const SimCreateArgs &args)
    {
        return Sim::Config {
            args.seed, args.world_base, args.flags & 0xFFFFu,
            (args.flags >> 16) & 0xFFu,
            loadPhysicsObjects(args),
            (args.flags >> 24) & 1u,
            (args.flags >> 25) & 1u,
            ((args.flags >> 26) & 1u) != 0 ?
                (int32_t)((args.flags >> 27) & 31u) : -1,
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
    using ballpit::ExportID;
    int64_t W = num_worlds;
    out.push_back({ "reset", SIM_I32, { W, 1 }, (uint32_t)ExportID::Reset });
    out.push_back({ "steps_remaining", SIM_I32, { W, 1 },
                    (uint32_t)ExportID::StepsRemaining });
    out.push_back({ "query_probe", SIM_I32, { W, 2 },
                    (uint32_t)ExportID::QueryProbe });
}

template <typename T>
void SimTraits::describeColumns(T &cols)
{
    using namespace ballpit;
    using madrona::Entity;
    using madrona::phys::broadphase::LeafID;

    cols.template add<MovableObject, Entity>("MovableObject.Entity", false);
    cols.template add<MovableObject, Position>("MovableObject.Position", true);
    cols.template add<MovableObject, Rotation>("MovableObject.Rotation", true);
    cols.template add<MovableObject, Scale>("MovableObject.Scale", true);
    cols.template add<MovableObject, Velocity>("MovableObject.Velocity", true);
    cols.template add<MovableObject, ExternalForce>(
        "MovableObject.ExternalForce", true);
    cols.template add<MovableObject, LeafID>("MovableObject.LeafID", false);
    cols.template add<MovableObject, KickIndex>("MovableObject.KickIndex", false);

    cols.template add<StaticObject, Entity>("StaticObject.Entity", false);
    cols.template add<StaticObject, Position>("StaticObject.Position", true);
    cols.template add<StaticObject, LeafID>("StaticObject.LeafID", false);
}

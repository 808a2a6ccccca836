#include "render_prep/sim.hpp"

struct SimTraits;
#include "common/sim_c_api.h"
#include "common/mesh_set.hpp"

#include <cmath>
#include <cstring>
#include <vector>

#ifndef SIM_BACKEND_REF_CPU
#include "common/render_config.hpp"
#endif

#include "common/cpu_render_bridge.hpp"

namespace {

// flags: bit 4 = movers stand on a coarse grid (Morton-code ties);
// bit 0 = RenderingSystem::setupTasks(update_visual_properties = true),
// bit 1 = depth only, bit 2 = crowded worlds, bit 3 = the geometry goes to the
// executor in the reference's asset form (MeshBVHData / MaterialData) instead
// of as plain triangles,
// bits 8-15 = ray caster output resolution (HIP backend; 0 = ray caster off)
constexpr uint32_t kMaxRecordsPerWorld = 128;

#ifdef SIM_BACKEND_REF_CPU
simmgr::CpuRenderBridge *g_bridge = nullptr;
#endif

// object-space root boxes of the four "models" (HIP backend, ray caster on)
const float kRootAABBs[renderprep::consts::numObjects * 6] = {
    -0.5f, -0.5f, -0.5f, 0.5f, 0.5f, 0.5f,
    -1.f, -0.25f, 0.f, 1.f, 0.25f, 2.f,
    -0.75f, -0.75f, -0.1f, 0.75f, 0.75f, 0.1f,
    0.f, 0.f, 0.f, 1.5f, 1.f, 0.5f,
};

// Triangle meshes of the four "models", inside those boxes: a cube, an
// ellipsoid (168 triangles: a bottom-level BVH several levels deep), a flat
// cylinder and a wedge.  Materials: one per object for the first three; the
// wedge's mesh has none of its own (-1): its triangles carry theirs -- a
// textured one (a checker, sampled at the hit's interpolated uv) on the floor
// and the slope, a plain one on the sides, none on the back; a fifth material
// is only ever reached through MaterialOverride, a sixth (textured) too.
const simmesh::MeshSet &meshes()
{
    static const simmesh::MeshSet set = [] {
        simmesh::MeshSet m;
        const int32_t red = m.material(0.8f, 0.2f, 0.2f);
        const int32_t green = m.material(0.2f, 0.7f, 0.3f);
        const int32_t blue = m.material(0.25f, 0.35f, 0.9f);
        m.material(0.9f, 0.8f, 0.1f);
        m.material(0.5f, 0.5f, 0.5f);
        const int32_t checker = m.checkerTexture(16, 8, 2);
        const int32_t stripes = m.checkerTexture(5, 7, 1);
        const int32_t tiled = m.material(1.f, 0.9f, 0.8f, checker);
        const int32_t plain = m.material(0.3f, 0.8f, 0.8f);
        m.material(0.7f, 1.f, 0.6f, stripes);

        m.box(-0.5f, -0.5f, -0.5f, 0.5f, 0.5f, 0.5f);
        m.endObject(red);
        m.ellipsoid(0.f, 0.f, 1.f, 1.f, 0.25f, 1.f, 12, 8);
        m.endObject(green);
        m.cylinder(0.75f, -0.1f, 0.1f, 16);
        m.endObject(blue);
        // wedge in [0, 1.5] x [0, 1] x [0, 0.5]
        // (uvs run past [0, 1]: wrap addressing)
        m.vert(0.f, 0.f, 0.f); m.uv(0.f, 0.f);
        m.vert(1.5f, 0.f, 0.f); m.uv(1.5f, 0.f);
        m.vert(0.f, 1.f, 0.f); m.uv(0.f, 1.f);
        m.vert(1.5f, 1.f, 0.f); m.uv(1.5f, 1.f);
        m.vert(0.f, 0.f, 0.5f); m.uv(-0.25f, 0.1f);
        m.vert(0.f, 1.f, 0.5f); m.uv(-0.25f, 0.9f);
        m.currentTriangleMaterial = tiled;
        m.quad(0, 2, 3, 1);         // floor
        m.currentTriangleMaterial = -1;
        m.quad(0, 4, 5, 2);         // back
        m.currentTriangleMaterial = tiled;
        m.quad(1, 3, 5, 4);         // slope
        m.currentTriangleMaterial = plain;
        m.tri(0, 1, 4); m.tri(2, 5, 3);
        m.currentTriangleMaterial = -1;
        m.endObject(-1);
        return m;
    }();
    return set;
}

uint32_t g_resolution = 0;

}

struct SimTraits {
    using Sim = renderprep::Sim;
    using Engine = renderprep::Engine;

    static constexpr uint32_t numExports =
        (uint32_t)renderprep::ExportID::NumExports;
    static constexpr uint32_t numTaskGraphs = 1;

    static Sim::Config makeConfig(const SimCreateArgs &args)
    {
        const madrona::render::RenderECSBridge *bridge = nullptr;
#ifdef SIM_BACKEND_REF_CPU
        delete g_bridge;
        g_bridge = new simmgr::CpuRenderBridge(
            args.num_worlds, kMaxRecordsPerWorld, 64,
            renderprep::consts::numViewers, renderprep::consts::maxMovers);
        bridge = &g_bridge->bridge;
#endif
        return Sim::Config { args.seed, args.world_base, args.flags & 1u,
                             (args.flags >> 2) & 1u, (args.flags >> 4) & 1u,
                             bridge };
    }

    static void makeInits(const SimCreateArgs &, Sim::WorldInit *) {}

    static const simmesh::MeshSet &renderMeshes() { return meshes(); }

#ifdef SIM_BACKEND_REF_CPU
    // the renderer zeroes the append counters before every step
    static void preStep() { g_bridge->beginStep(); }
#else
    static madrona::Optional<madrona::CudaBatchRenderConfig> renderConfig(
        const SimCreateArgs &args)
    {
        madrona::CudaBatchRenderConfig cfg {};
        cfg.renderMode = madrona::CudaBatchRenderConfig::RenderMode::RGBD;
        cfg.renderResolution = (args.flags >> 8) & 0xFFu;
        g_resolution = cfg.renderResolution;
        cfg.objectRootAABBs = madrona::Span<const float>(
            kRootAABBs, (madrona::CountT)(renderprep::consts::numObjects * 6));
        if ((args.flags & 2u) != 0u) {
            cfg.renderMode = madrona::CudaBatchRenderConfig::RenderMode::Depth;
        }
        if ((args.flags & 8u) != 0u) {
            // (kept alive with the process: the executor copies what it needs
            // while it is constructed)
            simmgr::meshesToReferenceAssets(meshes(), cfg);
        } else {
            simmgr::meshesToRenderConfig(meshes(), cfg);
        }
        return madrona::Optional<madrona::CudaBatchRenderConfig>::make(cfg);
    }
#endif

    template <typename T>
    static void describeTensors(T &out, uint32_t num_worlds);
    template <typename T>
    static void describeColumns(T &cols);
};

#include "common/mgr_impl.inl"

template <typename T>
void SimTraits::describeTensors(T &out, uint32_t num_worlds)
{
    out.push_back({ "roster", SIM_I32, { (int64_t)num_worlds, 2 },
                    (uint32_t)renderprep::ExportID::Roster });
#ifndef SIM_BACKEND_REF_CPU
    if (g_resolution != 0u) {
        // rows of the render-target table = output slots = rows of the camera
        // table (two viewers per world)
        const int64_t views = (int64_t)num_worlds * renderprep::consts::numViewers;
        out.push_back({ "rgb", SIM_U8,
                        { views, (int64_t)g_resolution, (int64_t)g_resolution, 4 },
                        (uint32_t)renderprep::ExportID::RGB });
        out.push_back({ "depth", SIM_F32,
                        { views, (int64_t)g_resolution, (int64_t)g_resolution },
                        (uint32_t)renderprep::ExportID::Depth });
    }
#endif
}

template <typename T>
void SimTraits::describeColumns(T &cols)
{
    using namespace renderprep;
    using namespace madrona::render;

    // simulator side: identical rows in identical order on both backends
    cols.template add<Mover, Position>("Mover.Position", true);
    cols.template add<Mover, Rotation>("Mover.Rotation", true);
    cols.template add<Mover, Renderable>("Mover.Renderable", false);
    cols.template add<Mover, MaterialOverride>("Mover.MaterialOverride", false);
    cols.template add<Mover, ColorOverride>("Mover.ColorOverride", false);
    cols.template add<Viewer, Position>("Viewer.Position", true);
    cols.template add<Lamp, Position>("Lamp.Position", true);
    // render side.  The light table is filled the same way on both backends;
    // Morton codes too, but the reference's CPU sort by Morton code is not a
    // sort: compared per world as multisets.
    cols.template add<LightArchetype, LightDesc>("Light.LightDesc", false);
    cols.template add<RenderableArchetype, MortonCode>("Renderable.MortonCode", false);
#ifndef SIM_BACKEND_REF_CPU
    // GPU mode only: the render entities' rows ARE the renderer's records
    cols.template add<RenderableArchetype, madrona::Entity>("Renderable.Entity", false);
    cols.template add<RenderableArchetype, InstanceData>("Renderable.InstanceData", false);
    cols.template add<RenderableArchetype, TLBVHNode>("Renderable.TLBVHNode", true);
    cols.template add<RenderCameraArchetype, PerspectiveCameraData>("Camera.PerspectiveCameraData", false);
    cols.template add<RenderCameraArchetype, RenderOutputIndex>("Camera.RenderOutputIndex", false);
    cols.template add<RenderCameraArchetype, RenderOutputRef>("Camera.RenderOutputRef", false);
#endif
}

// CPU mode: the records of the last step as the reference appended them to the
// bridge (arrival order) + the (world << 32 | entity id) key of each.
// kind 0 = instances (64 B each), 1 = views (48 B each).  Returns the count.
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

#include "render_prep/sim.hpp"

struct SimTraits;
#include "common/sim_c_api.h"

#include <cmath>
#include <cstring>
#include <vector>

#ifndef SIM_BACKEND_REF_CPU
#include <madrona/mw_gpu.hpp>
#endif

#ifdef SIM_BACKEND_REF_CPU
// the reference's private bridge struct, where it lies (oracle build only)
#include "ecs_interop.hpp"
#include <madrona/sync.hpp>
#endif

namespace {

// flags: bit 0 = RenderingSystem::setupTasks(update_visual_properties = true),
// bits 8-15 = ray caster output resolution (HIP backend; 0 = ray caster off)
constexpr uint32_t kMaxRecordsPerWorld = 64;

#ifdef SIM_BACKEND_REF_CPU
// What the reference's Vulkan renderer would own: the buffers its CPU-mode
// systems append instance / view records to (src/render/ecs_interop.hpp).
struct CpuBridge {
    madrona::render::RenderECSBridge bridge {};
    std::vector<madrona::render::PerspectiveCameraData> views;
    std::vector<madrona::render::InstanceData> instances;
    std::vector<uint64_t> instanceKeys, viewKeys;
    uint32_t totalViews = 0, totalInstances = 0;
    madrona::AtomicU32 viewCounter { 0 };
    madrona::AtomicU32 instanceCounter { 0 };
};
CpuBridge *g_bridge = nullptr;
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
// cylinder and a wedge.  Materials: one per object, the wedge has none.
struct Meshes {
    std::vector<float> vertices;
    std::vector<uint32_t> indices;
    std::vector<uint32_t> vertexOffsets { 0 };
    std::vector<uint32_t> triangleOffsets { 0 };
    std::vector<float> materialColors;
    std::vector<int32_t> objectMaterials;

    uint32_t vert(float x, float y, float z)
    {
        vertices.insert(vertices.end(), { x, y, z });
        return (uint32_t)(vertices.size() / 3) - vertexOffsets.back() - 1u;
    }
    void tri(uint32_t a, uint32_t b, uint32_t c)
    {
        indices.insert(indices.end(), { a, b, c });
    }
    void quad(uint32_t a, uint32_t b, uint32_t c, uint32_t d)
    {
        tri(a, b, c);
        tri(a, c, d);
    }
    void endObject()
    {
        vertexOffsets.push_back((uint32_t)(vertices.size() / 3));
        triangleOffsets.push_back((uint32_t)(indices.size() / 3));
    }

    Meshes()
    {
        // 0: cube
        for (int i = 0; i < 8; i++) {
            vert(i & 1 ? 0.5f : -0.5f, i & 2 ? 0.5f : -0.5f, i & 4 ? 0.5f : -0.5f);
        }
        quad(0, 2, 3, 1); quad(4, 5, 7, 6); quad(0, 1, 5, 4);
        quad(2, 6, 7, 3); quad(0, 4, 6, 2); quad(1, 3, 7, 5);
        endObject();

        // 1: ellipsoid, centre (0, 0, 1), radii (1, 0.25, 1)
        {
            const int slices = 12, stacks = 8;
            const uint32_t south = vert(0.f, 0.f, 0.f);
            for (int st = 1; st < stacks; st++) {
                const float phi = 3.14159265f * (float)st / (float)stacks;
                for (int sl = 0; sl < slices; sl++) {
                    const float th = 6.2831853f * (float)sl / (float)slices;
                    vert(sinf(phi) * cosf(th), 0.25f * sinf(phi) * sinf(th),
                         1.f - cosf(phi));
                }
            }
            const uint32_t north = vert(0.f, 0.f, 2.f);
            auto ring = [&](int st, int sl) {
                return 1u + (uint32_t)((st - 1) * slices + sl % slices);
            };
            for (int sl = 0; sl < slices; sl++) {
                tri(south, ring(1, sl + 1), ring(1, sl));
                tri(north, ring(stacks - 1, sl), ring(stacks - 1, sl + 1));
                for (int st = 1; st < stacks - 1; st++) {
                    quad(ring(st, sl), ring(st, sl + 1), ring(st + 1, sl + 1),
                         ring(st + 1, sl));
                }
            }
        }
        endObject();

        // 2: cylinder, radius 0.75, z in [-0.1, 0.1]
        {
            const int segs = 16;
            const uint32_t lo = vert(0.f, 0.f, -0.1f), hi = vert(0.f, 0.f, 0.1f);
            for (int i = 0; i < segs; i++) {
                const float th = 6.2831853f * (float)i / (float)segs;
                vert(0.75f * cosf(th), 0.75f * sinf(th), -0.1f);
                vert(0.75f * cosf(th), 0.75f * sinf(th), 0.1f);
            }
            for (int i = 0; i < segs; i++) {
                const uint32_t a = 2u + 2u * (uint32_t)i;
                const uint32_t b = 2u + 2u * (uint32_t)((i + 1) % segs);
                tri(lo, b, a);
                tri(hi, a + 1, b + 1);
                quad(a, b, b + 1, a + 1);
            }
        }
        endObject();

        // 3: wedge in [0, 1.5] x [0, 1] x [0, 0.5]
        vert(0.f, 0.f, 0.f); vert(1.5f, 0.f, 0.f); vert(0.f, 1.f, 0.f);
        vert(1.5f, 1.f, 0.f); vert(0.f, 0.f, 0.5f); vert(0.f, 1.f, 0.5f);
        quad(0, 2, 3, 1); quad(0, 4, 5, 2); quad(1, 3, 5, 4);
        tri(0, 1, 4); tri(2, 5, 3);
        endObject();

        materialColors = { 0.8f, 0.2f, 0.2f,  0.2f, 0.7f, 0.3f,  0.25f, 0.35f, 0.9f,
                           0.9f, 0.8f, 0.1f,  0.5f, 0.5f, 0.5f };
        objectMaterials = { 0, 1, 2, -1 };
    }
};

const Meshes &meshes()
{
    static const Meshes m;
    return m;
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
        g_bridge = new CpuBridge();
        const size_t cap = (size_t)args.num_worlds * kMaxRecordsPerWorld;
        g_bridge->views.resize(cap);
        g_bridge->instances.resize(cap);
        g_bridge->instanceKeys.resize(cap);
        g_bridge->viewKeys.resize(cap);
        auto &b = g_bridge->bridge;
        b.views = g_bridge->views.data();
        b.instances = g_bridge->instances.data();
        b.totalNumViews = &g_bridge->totalViews;
        b.totalNumInstances = &g_bridge->totalInstances;
        b.totalNumViewsCPUInc = &g_bridge->viewCounter;
        b.totalNumInstancesCPUInc = &g_bridge->instanceCounter;
        b.instancesWorldIDs = g_bridge->instanceKeys.data();
        b.viewsWorldIDs = g_bridge->viewKeys.data();
        b.renderWidth = 64;
        b.renderHeight = 64;
        b.maxViewsPerworld = renderprep::consts::numViewers;
        b.maxInstancesPerWorld = renderprep::consts::maxMovers;
        b.isGPUBackend = false;
        bridge = &b;
#endif
        return Sim::Config { args.seed, args.world_base, args.flags & 1u,
                             bridge };
    }

    static void makeInits(const SimCreateArgs &, Sim::WorldInit *) {}

#ifdef SIM_BACKEND_REF_CPU
    // the renderer zeroes the append counters before every step
    static void preStep()
    {
        g_bridge->viewCounter.store_relaxed(0);
        g_bridge->instanceCounter.store_relaxed(0);
    }
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
        const Meshes &m = meshes();
        cfg.geoBVHData.vertices = madrona::Span<const madrona::math::Vector3>(
            (const madrona::math::Vector3 *)m.vertices.data(),
            (madrona::CountT)(m.vertices.size() / 3));
        cfg.geoBVHData.indices = madrona::Span<const uint32_t>(
            m.indices.data(), (madrona::CountT)m.indices.size());
        cfg.geoBVHData.objectVertexOffsets = madrona::Span<const uint32_t>(
            m.vertexOffsets.data(), (madrona::CountT)m.vertexOffsets.size());
        cfg.geoBVHData.objectTriangleOffsets = madrona::Span<const uint32_t>(
            m.triangleOffsets.data(), (madrona::CountT)m.triangleOffsets.size());
        cfg.materialData.materialColors =
            madrona::Span<const madrona::math::Vector3>(
                (const madrona::math::Vector3 *)m.materialColors.data(),
                (madrona::CountT)(m.materialColors.size() / 3));
        cfg.materialData.objectMaterials = madrona::Span<const int32_t>(
            m.objectMaterials.data(), (madrona::CountT)m.objectMaterials.size());
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

// The meshes and materials the ray caster was given (for the test's brute-force
// oracle).  Any pointer may be NULL; returns the number of objects and, through
// counts, { vertices, triangles, materials }.
extern "C" SIM_API int32_t render_prep_geometry(float *vertices, uint32_t *indices,
                                                uint32_t *vertex_offsets,
                                                uint32_t *triangle_offsets,
                                                float *material_colors,
                                                int32_t *object_materials,
                                                uint32_t *counts)
{
    const Meshes &m = meshes();
    if (vertices) memcpy(vertices, m.vertices.data(), m.vertices.size() * 4);
    if (indices) memcpy(indices, m.indices.data(), m.indices.size() * 4);
    if (vertex_offsets) {
        memcpy(vertex_offsets, m.vertexOffsets.data(), m.vertexOffsets.size() * 4);
    }
    if (triangle_offsets) {
        memcpy(triangle_offsets, m.triangleOffsets.data(),
               m.triangleOffsets.size() * 4);
    }
    if (material_colors) {
        memcpy(material_colors, m.materialColors.data(),
               m.materialColors.size() * 4);
    }
    if (object_materials) {
        memcpy(object_materials, m.objectMaterials.data(),
               m.objectMaterials.size() * 4);
    }
    if (counts) {
        counts[0] = (uint32_t)(m.vertices.size() / 3);
        counts[1] = (uint32_t)(m.indices.size() / 3);
        counts[2] = (uint32_t)(m.materialColors.size() / 3);
    }
    return renderprep::consts::numObjects;
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
    const uint64_t n = kind == 0 ? g_bridge->instanceCounter.load_relaxed() :
                                   g_bridge->viewCounter.load_relaxed();
    if (n > max_records) return -2;
    if (kind == 0) {
        memcpy(dst, g_bridge->instances.data(), n * 64);
        memcpy(keys_dst, g_bridge->instanceKeys.data(), n * 8);
    } else {
        memcpy(dst, g_bridge->views.data(), n * 48);
        memcpy(keys_dst, g_bridge->viewKeys.data(), n * 8);
    }
    return (int64_t)n;
#else
    (void)kind; (void)dst; (void)keys_dst; (void)max_records;
    return -1;
#endif
}

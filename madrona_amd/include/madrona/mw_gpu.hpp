// Host-side executor of the MI355X backend.
//
// API contract: reference include/madrona/mw_gpu.hpp:25-164 -- StateConfig,
// CompileConfig, MWCudaLaunchGraph and MWCudaExecutor keep their names,
// fields and method signatures so a simulator's Manager compiles unchanged
// ("Cuda" in the names is historical: everything below drives HIP).  The
// class is a header-only shim over the C ABI in include/mwhip.h; errors abort
// like the reference's REQ_CUDA / FATAL.
//
// Differences that are visible to callers:
//  * CompileConfig::userSources / userCompileFlags are accepted and ignored:
//    simulator device code is compiled offline by hipcc into the same shared
//    object; there is no runtime compiler.
//  * initCUDA returns an opaque context value carrying the gpu id.
#pragma once

#include <madrona/macros.hpp>
#include <madrona/span.hpp>
#include <madrona/optional.hpp>
#include <madrona/types.hpp>
#include <madrona/math.hpp>
#include <madrona/render/cuda_batch_render_assets.hpp>

#include <mwhip.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

// Defined by MADRONA_BUILD_MWGPU_ENTRY in the simulator's device TU.
extern "C" const mwhip_user_entry *madronaMWHipUserEntry();

namespace madrona {

struct StateConfig {
    void *worldInitPtr;
    uint32_t numWorldInitBytes;
    void *userConfigPtr;
    uint32_t numUserConfigBytes;
    uint32_t numWorldDataBytes;
    uint32_t worldDataAlignment;
    uint32_t numWorlds;
    uint32_t numTaskGraphs;
    uint32_t numExportedBuffers;
};

struct CompileConfig {
    enum class OptMode : uint32_t {
        Optimize,
        LTO,
        Debug,
    };

    Span<const char *const> userSources;
    Span<const char *const> userCompileFlags;
    OptMode optMode = OptMode::LTO;
};

// Batch ray caster configuration (reference mw_gpu.hpp:77-96).  Geometry and
// materials arrive either in the reference's own form -- geoBVHData /
// materialData, the types of <madrona/render/cuda_batch_render_assets.hpp>:
// 4-wide quantised mesh BVHs with de-indexed vertices, per-triangle materials
// and texture objects, as a renderer's asset processor produces them -- or, for
// applications without such a processor (it needs Embree), as plain indexed
// triangles: geoTriangles / triangleMaterials, an addition of this backend.
// Either way the executor builds its own bottom-level trees from the triangles
// (any valid BVH gives the same image); of a MeshBVHData it reads the leaves.
namespace render {

// indexed triangles of the renderable objects, host memory
struct TriangleMeshData {
    Span<const math::Vector3> vertices = {};        // all objects
    Span<const uint32_t> indices = {};              // 3 per triangle, object-local
    Span<const uint32_t> objectVertexOffsets = {};  // [numObjects + 1]
    Span<const uint32_t> objectTriangleOffsets = {};// [numObjects + 1]
    Span<const math::Vector2> vertexUVs = {};       // per vertex, or empty
    // per triangle (all objects): the material of triangles of objects whose
    // objectMaterials entry is -1 (reference MeshBVH::leafMats); or empty
    Span<const int32_t> triangleMaterials = {};
};

struct TriangleMaterialData {
    Span<const math::Vector3> materialColors = {};
    Span<const int32_t> objectMaterials = {};       // [numObjects], -1: per triangle / none
    Span<const int32_t> materialTextures = {};      // per material: texture or -1; or empty
    Span<const TextureRGBA8> textures = {};
};

}

struct CudaBatchRenderConfig {
    enum class RenderMode : uint32_t {
        RGBD,
        Depth,
    };

    RenderMode renderMode = RenderMode::RGBD;
    render::MeshBVHData geoBVHData = {};
    render::MaterialData materialData = {};
    // 6 floats per object id (min xyz, max xyz), host memory; empty: taken from
    // the geometry
    Span<const float> objectRootAABBs = {};
    // the ray caster's outputs are renderResolution x renderResolution
    uint32_t renderResolution = 0;
    // (reference mw_gpu.hpp:89-90.  Its ray caster hands nearPlane down as
    // BVHParams::nearSphere -> TraceInfo::tMin, bvh_raycast.cpp:983, and never
    // reads tMin again: accepted here, with the same effect)
    float nearPlane = 0.f;
    float farPlane = 0.f;
    // ---- this backend ----
    // most views a world will hold, 0 = unknown: sizes the render-target table,
    // 8 * renderResolution^2 bytes a view
    uint32_t maxViewsPerWorld = 0;
    // the geometry as plain triangles (used when geoBVHData holds no meshes)
    render::TriangleMeshData geoTriangles = {};
    render::TriangleMaterialData triangleMaterials = {};
    // entries of materialData.materials (the reference's struct carries no
    // count); 0: one past the largest material id the meshes name
    uint32_t numMaterials = 0;
};

// Opaque device context handle (the reference returns a CUcontext)
struct MWHipContext {
    int32_t gpuID;
};
using CUcontext = MWHipContext;

class MWCudaExecutor;

class MWCudaLaunchGraph {
public:
    MWCudaLaunchGraph() : exec_(nullptr), graph_(0) {}
    MWCudaLaunchGraph(const MWCudaLaunchGraph &) = delete;
    MWCudaLaunchGraph(MWCudaLaunchGraph &&o) : exec_(o.exec_), graph_(o.graph_)
    {
        o.exec_ = nullptr;
        o.graph_ = 0;
    }

    ~MWCudaLaunchGraph()
    {
        if (exec_ != nullptr) {
            mwhip_free_launch_graph(exec_, graph_);
        }
    }

    MWCudaLaunchGraph &operator=(MWCudaLaunchGraph &&o)
    {
        if (this != &o) {
            if (exec_ != nullptr) {
                mwhip_free_launch_graph(exec_, graph_);
            }
            exec_ = o.exec_;
            graph_ = o.graph_;
            o.exec_ = nullptr;
            o.graph_ = 0;
        }
        return *this;
    }

    uint64_t handle() const { return graph_; }

private:
    MWCudaLaunchGraph(mwhip_exec *exec, uint64_t graph)
        : exec_(exec), graph_(graph)
    {}

    mwhip_exec *exec_;
    uint64_t graph_;

friend class MWCudaExecutor;
};

namespace detail {

// A renderer's MeshBVHData / MaterialData (host or device memory) as the
// triangle description the executor's C ABI takes: every leaf child of every
// node names `triSize` consecutive triangles, three de-indexed BVHVertex each
// (reference mesh_bvh.hpp:33-47, bvh_raycast.cpp:303-313).
struct ReferenceAssets {
    std::vector<float> vertices;            // 3 per triangle corner
    std::vector<float> uvs;
    std::vector<uint32_t> indices;          // 0, 1, 2, ... per object
    std::vector<uint32_t> vertexOffsets { 0u };
    std::vector<uint32_t> triangleOffsets { 0u };
    std::vector<int32_t> triangleMaterials;
    std::vector<int32_t> objectMaterials;
    std::vector<float> materialColors;
    std::vector<int32_t> materialTextures;
    std::vector<render::TextureRGBA8> textureDescs;

    template <typename T>
    static std::vector<T> fetch(const T *src, uint64_t count)
    {
        std::vector<T> host(count);
        if (count != 0 && mwhip_memcpy_any(host.data(), src,
                                           count * sizeof(T)) != 0) {
            fprintf(stderr, "madrona_amd: cannot read the render assets: %s\n",
                    mwhip_last_error());
            abort();
        }
        return host;
    }

    template <typename ConfigT>
    void read(const ConfigT &cfg)
    {
        const render::MeshBVHData &geo = cfg.geoBVHData;
        const std::vector<MeshBVH> meshes = fetch(geo.meshBVHs, geo.numBVHs);
        int32_t max_material = -1;
        for (const MeshBVH &mesh : meshes) {
            const std::vector<QBVHNode> nodes = fetch(mesh.nodes, mesh.numNodes);
            const std::vector<MeshBVH::BVHVertex> verts =
                fetch(mesh.vertices, mesh.numVerts);
            const std::vector<MeshBVH::LeafMaterial> leaf_mats =
                mesh.materialIDX == -1 && mesh.leafMats != nullptr ?
                    fetch(mesh.leafMats, mesh.numVerts / 3u) :
                    std::vector<MeshBVH::LeafMaterial>();
            uint32_t corner = 0;
            for (const QBVHNode &node : nodes) {
                for (uint32_t c = 0; c < (uint32_t)MADRONA_BVH_WIDTH; c++) {
                    if (!node.hasChild(c) || !node.isLeaf(c)) continue;
                    const uint32_t first = node.leafIDX(c);
                    for (uint32_t t = 0; t < node.triSize[c]; t++) {
                        for (uint32_t k = 0; k < 3u; k++) {
                            const MeshBVH::BVHVertex &v =
                                verts[(size_t)(first + t) * 3u + k];
                            vertices.insert(vertices.end(),
                                            { v.pos.x, v.pos.y, v.pos.z });
                            uvs.insert(uvs.end(), { v.uv.x, v.uv.y });
                            indices.push_back(corner++);
                        }
                        const int32_t m = leaf_mats.empty() ? -1 :
                            leaf_mats[first + t].material[0].matIDX;
                        triangleMaterials.push_back(m);
                        max_material = m > max_material ? m : max_material;
                    }
                }
            }
            vertexOffsets.push_back((uint32_t)(vertices.size() / 3));
            triangleOffsets.push_back((uint32_t)(indices.size() / 3));
            objectMaterials.push_back(mesh.materialIDX);
            max_material = mesh.materialIDX > max_material ?
                mesh.materialIDX : max_material;
        }

        const uint32_t num_materials = cfg.numMaterials != 0 ?
            cfg.numMaterials : (uint32_t)(max_material + 1);
        const std::vector<Material> materials =
            fetch(cfg.materialData.materials,
                  cfg.materialData.materials != nullptr ? num_materials : 0u);
        for (const Material &m : materials) {
            materialColors.insert(materialColors.end(),
                                  { m.color.x, m.color.y, m.color.z });
            materialTextures.push_back(m.textureIdx);
        }
        const std::vector<cudaTextureObject_t> handles = fetch(
            cfg.materialData.textures, cfg.materialData.numTextureBuffers);
        for (cudaTextureObject_t h : handles) {
            textureDescs.push_back(*(const render::TextureRGBA8 *)(uintptr_t)h);
        }
    }

    void describe(mwhip_render_geometry &geometry,
                  std::vector<mwhip_texture> &textures) const
    {
        geometry.num_objects = (uint32_t)triangleOffsets.size() - 1u;
        geometry.vertices = vertices.data();
        geometry.indices = indices.data();
        geometry.object_vertex_offset = vertexOffsets.data();
        geometry.object_triangle_offset = triangleOffsets.data();
        geometry.vertex_uv = uvs.data();
        geometry.triangle_material = triangleMaterials.data();
        geometry.object_material = objectMaterials.data();
        geometry.num_materials = (uint32_t)(materialColors.size() / 3);
        geometry.material_color = materialColors.data();
        if (!textureDescs.empty()) {
            geometry.material_texture = materialTextures.data();
            for (const render::TextureRGBA8 &t : textureDescs) {
                textures.push_back({ t.width, t.height, t.pixels });
            }
            geometry.num_textures = (uint32_t)textures.size();
            geometry.textures = textures.data();
        }
    }
};

}

class MWCudaExecutor {
    using ReferenceAssets = detail::ReferenceAssets;
public:
    static CUcontext initCUDA(int gpu_id) { return CUcontext { gpu_id }; }

    MWCudaExecutor() : exec_(nullptr), num_taskgraphs_(0) {}

    MWCudaExecutor(const StateConfig &state_cfg,
                   const CompileConfig &,
                   CUcontext ctx,
                   Optional<CudaBatchRenderConfig> render_cfg =
                       Optional<CudaBatchRenderConfig>::none())
        : exec_(nullptr), num_taskgraphs_(state_cfg.numTaskGraphs)
    {
        mwhip_state_config cfg {};
        mwhip_render_geometry geometry {};
        std::vector<mwhip_texture> textures;
        ReferenceAssets from_reference;
        if (render_cfg.has_value()) {
            const render::TriangleMeshData &geo = render_cfg->geoTriangles;
            if (render_cfg->geoBVHData.numBVHs != 0) {
                // the reference's asset form: read the leaves of its mesh BVHs
                from_reference.read(*render_cfg);
                from_reference.describe(geometry, textures);
                cfg.render_geometry = &geometry;
            } else if (geo.objectTriangleOffsets.size() > 1) {
                geometry.num_objects =
                    (uint32_t)geo.objectTriangleOffsets.size() - 1u;
                geometry.vertices = (const float *)geo.vertices.data();
                geometry.indices = geo.indices.data();
                geometry.object_vertex_offset = geo.objectVertexOffsets.data();
                geometry.object_triangle_offset =
                    geo.objectTriangleOffsets.data();
                geometry.vertex_uv = geo.vertexUVs.size() != 0 ?
                    (const float *)geo.vertexUVs.data() : nullptr;
                geometry.triangle_material = geo.triangleMaterials.size() != 0 ?
                    geo.triangleMaterials.data() : nullptr;
                const render::TriangleMaterialData &mats =
                    render_cfg->triangleMaterials;
                geometry.num_materials = (uint32_t)mats.materialColors.size();
                geometry.material_color =
                    (const float *)mats.materialColors.data();
                geometry.object_material = mats.objectMaterials.size() != 0 ?
                    mats.objectMaterials.data() : nullptr;
                if (mats.materialTextures.size() != 0) {
                    geometry.material_texture = mats.materialTextures.data();
                    for (const render::TextureRGBA8 &t : mats.textures) {
                        textures.push_back({ t.width, t.height, t.pixels });
                    }
                    geometry.num_textures = (uint32_t)textures.size();
                    geometry.textures = textures.data();
                }
                cfg.render_geometry = &geometry;
            }
            cfg.raycast_output_resolution = render_cfg->renderResolution;
            cfg.raycast_max_views_per_world = render_cfg->maxViewsPerWorld;
            cfg.raycast_rgbd = render_cfg->renderMode ==
                CudaBatchRenderConfig::RenderMode::RGBD ? 1u : 0u;
            cfg.object_root_aabbs = render_cfg->objectRootAABBs.data();
            cfg.num_object_root_aabbs =
                (uint32_t)render_cfg->objectRootAABBs.size() / 6u;
        }
        cfg.world_init_ptr = state_cfg.worldInitPtr;
        cfg.num_world_init_bytes = state_cfg.numWorldInitBytes;
        cfg.user_config_ptr = state_cfg.userConfigPtr;
        cfg.num_user_config_bytes = state_cfg.numUserConfigBytes;
        cfg.num_world_data_bytes = state_cfg.numWorldDataBytes;
        cfg.world_data_alignment = state_cfg.worldDataAlignment;
        cfg.num_worlds = state_cfg.numWorlds;
        cfg.num_task_graphs = state_cfg.numTaskGraphs;
        cfg.num_exported_buffers = state_cfg.numExportedBuffers;
        cfg.gpu_id = ctx.gpuID;

        req(mwhip_create(&cfg, madronaMWHipUserEntry(), &exec_),
            "MWCudaExecutor");
    }

    MWCudaExecutor(const MWCudaExecutor &) = delete;
    MWCudaExecutor(MWCudaExecutor &&o)
        : exec_(o.exec_), num_taskgraphs_(o.num_taskgraphs_)
    {
        o.exec_ = nullptr;
    }

    ~MWCudaExecutor()
    {
        if (exec_ != nullptr) {
            mwhip_destroy(exec_);
        }
    }

    MWCudaExecutor &operator=(MWCudaExecutor &&o)
    {
        if (this != &o) {
            if (exec_ != nullptr) {
                mwhip_destroy(exec_);
            }
            exec_ = o.exec_;
            num_taskgraphs_ = o.num_taskgraphs_;
            o.exec_ = nullptr;
        }
        return *this;
    }

    template <EnumType EnumT>
    MWCudaLaunchGraph buildLaunchGraph(EnumT taskgraph_id,
                                       const char *stat_name = nullptr)
    {
        return buildLaunchGraph((uint32_t)taskgraph_id, stat_name);
    }

    MWCudaLaunchGraph buildLaunchGraph(uint32_t taskgraph_id,
                                       const char *stat_name = nullptr)
    {
        return buildLaunchGraph(Span<const uint32_t>(&taskgraph_id, 1),
                                stat_name);
    }

    MWCudaLaunchGraph buildLaunchGraph(Span<const uint32_t> taskgraph_ids,
                                       const char *stat_name = nullptr)
    {
        uint64_t graph = 0;
        req(mwhip_build_launch_graph(exec_, taskgraph_ids.data(),
            (uint32_t)taskgraph_ids.size(), stat_name, &graph),
            "buildLaunchGraph");
        return MWCudaLaunchGraph(exec_, graph);
    }

    MWCudaLaunchGraph buildLaunchGraphAllTaskGraphs()
    {
        std::vector<uint32_t> ids(num_taskgraphs_);
        for (uint32_t i = 0; i < num_taskgraphs_; i++) ids[i] = i;
        return buildLaunchGraph(
            Span<const uint32_t>(ids.data(), (CountT)ids.size()));
    }

    // TLAS build + ray cast of every view into the RaycastOutputArchetype
    // columns (reference mw_gpu.hpp:150, cuda_exec.cpp:2527-2700); run it after
    // the step graph
    MWCudaLaunchGraph buildRenderGraph()
    {
        uint64_t graph = 0;
        req(mwhip_build_render_graph(exec_, &graph), "buildRenderGraph");
        return MWCudaLaunchGraph(exec_, graph);
    }

    // synchronous (reference cuda_exec.cpp:2756-2794)
    void run(MWCudaLaunchGraph &launch_graph)
    {
        req(mwhip_run(exec_, launch_graph.graph_), "run");
    }

    // strm is a hipStream_t
    void runAsync(MWCudaLaunchGraph &launch_graph, void *strm)
    {
        req(mwhip_run_async(exec_, launch_graph.graph_, strm), "runAsync");
    }

    // device pointer, owned by the executor
    void *getExported(CountT slot) const
    {
        return mwhip_get_exported(exec_, (uint32_t)slot);
    }

    mwhip_exec *handle() const { return exec_; }

private:
    static void req(int rc, const char *what)
    {
        if (rc != 0) {
            fprintf(stderr, "madrona_amd: %s failed (%d): %s\n", what, rc,
                    mwhip_last_error());
            abort();
        }
    }

    mwhip_exec *exec_;
    uint32_t num_taskgraphs_;
};

using MWHipExecutor = MWCudaExecutor;
using MWHipLaunchGraph = MWCudaLaunchGraph;

}

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

#include <mwhip.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
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

// Batch ray caster configuration (reference mw_gpu.hpp:77-96).  The geometry
// side (geoBVHData / materialData) is not built on this backend yet (SURVEY
// 8f-1); what the render-prep systems need of it is one object-space root AABB
// per object id.
namespace render {

// Triangle geometry of the renderable objects, host memory (the reference's
// MeshBVHData holds Embree-built QBVHs already on the device,
// render/cuda_batch_render_assets.hpp:8-20; this executor builds its own
// bottom-level BVHs from the triangles).
struct MeshBVHData {
    Span<const math::Vector3> vertices = {};        // all objects
    Span<const uint32_t> indices = {};              // 3 per triangle, object-local
    Span<const uint32_t> objectVertexOffsets = {};  // [numObjects + 1]
    Span<const uint32_t> objectTriangleOffsets = {};// [numObjects + 1]
};

// One untextured material per object (the reference: per mesh, optionally
// textured, cuda_batch_render_assets.hpp:22-28).
struct MaterialData {
    Span<const math::Vector3> materialColors = {};
    Span<const int32_t> objectMaterials = {};       // [numObjects], -1: none
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
    // geoBVHData
    Span<const float> objectRootAABBs = {};
    // the ray caster's outputs are renderResolution x renderResolution
    uint32_t renderResolution = 0;
    float nearPlane = 0.f;
    float farPlane = 0.f;
    // (this backend) most views a world will hold, 0 = unknown: sizes the
    // render-target table, 8 * renderResolution^2 bytes a view
    uint32_t maxViewsPerWorld = 0;
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

class MWCudaExecutor {
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
        if (render_cfg.has_value()) {
            const render::MeshBVHData &geo = render_cfg->geoBVHData;
            if (geo.objectTriangleOffsets.size() > 1) {
                geometry.num_objects =
                    (uint32_t)geo.objectTriangleOffsets.size() - 1u;
                geometry.vertices = (const float *)geo.vertices.data();
                geometry.indices = geo.indices.data();
                geometry.object_vertex_offset = geo.objectVertexOffsets.data();
                geometry.object_triangle_offset =
                    geo.objectTriangleOffsets.data();
                const render::MaterialData &mats = render_cfg->materialData;
                geometry.num_materials = (uint32_t)mats.materialColors.size();
                geometry.material_color =
                    (const float *)mats.materialColors.data();
                geometry.object_material = mats.objectMaterials.size() != 0 ?
                    mats.objectMaterials.data() : nullptr;
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

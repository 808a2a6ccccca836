// Shared implementation of sims/common/sim_c_api.h.
//
// A simulator's mgr.cpp defines `struct SimTraits` (types, tensor list, dump
// list) and then includes this file.  Compiled twice:
//   -DSIM_BACKEND_REF_CPU  -> reference madrona::TaskGraphExecutor (the oracle;
//                             reference include/madrona/mw_cpu.hpp:73-110)
//   otherwise              -> madrona::MWCudaExecutor as provided by
//                             madrona_amd's <madrona/mw_gpu.hpp> (HIP backend;
//                             mirrors reference include/madrona/mw_gpu.hpp:98-164)
// The Manager code below is deliberately written the way the reference's
// simulators write theirs (StateConfig{...}, buildLaunchGraphAllTaskGraphs,
// run, getExported) so it doubles as the "links unchanged" check.
#pragma once

#include "sim_c_api.h"
#include "mesh_set.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#ifdef SIM_BACKEND_REF_CPU
#include <madrona/mw_cpu.hpp>
#else
#include <madrona/mw_gpu.hpp>
#include <mwhip.h>
#endif

namespace simmgr {

struct TensorDesc {
    std::string name;
    int32_t dtype;
    std::vector<int64_t> dims;
    uint32_t slot;
};

static inline uint64_t dtypeBytes(int32_t dtype)
{
    switch (dtype) {
    case SIM_U8: case SIM_I8: return 1;
    case SIM_I16: case SIM_F16: return 2;
    case SIM_I32: case SIM_F32: return 4;
    case SIM_I64: return 8;
    default: return 0;
    }
}

#ifdef SIM_BACKEND_REF_CPU
using DumpFn = std::pair<void *, uint32_t> (*)(madrona::StateManager &,
                                                uint32_t world);
#endif

struct ColumnDesc {
    std::string name;
    uint32_t elemBytes;
    int32_t isFloat;
#ifdef SIM_BACKEND_REF_CPU
    DumpFn fn;
#else
    uint32_t archetypeID;
    uint32_t componentID;
#endif
};

struct ColumnList {
    std::vector<ColumnDesc> cols;

    template <typename ArchetypeT, typename ComponentT>
    void add(const char *name, bool is_float)
    {
        ColumnDesc d;
        d.name = name;
        d.elemBytes = (uint32_t)sizeof(ComponentT);
        d.isFloat = is_float ? 1 : 0;
#ifdef SIM_BACKEND_REF_CPU
        d.fn = [](madrona::StateManager &mgr, uint32_t world)
                -> std::pair<void *, uint32_t> {
            if constexpr (std::is_same_v<ComponentT, madrona::Entity>) {
                auto *ptr = mgr.getWorldEntities<ArchetypeT>(world);
                auto n = mgr.numRows<ArchetypeT>(world);
                return { (void *)ptr, (uint32_t)n };
            } else {
                auto [ptr, n] = mgr.getWorldComponentsAndCount<
                    ArchetypeT, ComponentT>(world);
                return { (void *)ptr, n };
            }
        };
#else
        d.archetypeID = madrona::TypeTracker::typeID<ArchetypeT>();
        d.componentID = madrona::TypeTracker::typeID<ComponentT>();
#endif
        cols.push_back(std::move(d));
    }
};

}

namespace simmgr {
// optional: the meshes the simulator gives the ray caster
template <typename T>
static const simmesh::MeshSet *renderMeshes()
{
    if constexpr (requires { T::renderMeshes(); }) {
        return &T::renderMeshes();
    } else {
        return nullptr;
    }
}

// optional hook of a simulator's traits, run before every step
template <typename T>
static void preStep()
{
    if constexpr (requires { T::preStep(); }) {
        T::preStep();
    }
}
}

#ifndef SIM_BACKEND_REF_CPU
namespace simmgr {
template <typename T>
static madrona::Optional<madrona::CudaBatchRenderConfig> renderConfig(
    const SimCreateArgs &args)
{
    if constexpr (requires { T::renderConfig(args); }) {
        return T::renderConfig(args);
    } else {
        return madrona::Optional<madrona::CudaBatchRenderConfig>::none();
    }
}

// a step = every task graph back to back, unless the simulator says its other
// task graphs are test probes (Traits::stepIsTaskGraph0)
template <typename T>
static madrona::MWCudaLaunchGraph *buildStepGraph(madrona::MWCudaExecutor &exec)
{
    if constexpr (requires { T::stepIsTaskGraph0; }) {
        return new madrona::MWCudaLaunchGraph(exec.buildLaunchGraph(0u));
    } else {
        return new madrona::MWCudaLaunchGraph(
            exec.buildLaunchGraphAllTaskGraphs());
    }
}
}
#endif

struct SimHandle {
    using Traits = SimTraits;
    using Sim = typename Traits::Sim;
    using Engine = typename Traits::Engine;
    using Config = typename Sim::Config;
    using WorldInit = typename Sim::WorldInit;

    uint32_t numWorlds;
    std::vector<simmgr::TensorDesc> tensors;
    simmgr::ColumnList columns;

#ifdef SIM_BACKEND_REF_CPU
    using Exec = madrona::TaskGraphExecutor<Engine, Sim, Config, WorldInit>;
    Exec *exec;
#else
    madrona::MWCudaExecutor *exec;
    madrona::MWCudaLaunchGraph *stepGraph;
    std::vector<madrona::MWCudaLaunchGraph *> probeGraphs;
    madrona::MWCudaLaunchGraph *renderGraph;
#endif
};

extern "C" {

SimHandle *sim_create(const SimCreateArgs *args)
{
    using Traits = SimTraits;
    using Sim = typename Traits::Sim;
    using Config = typename Sim::Config;
    using WorldInit = typename Sim::WorldInit;

    SimHandle *h = new SimHandle {};
    h->numWorlds = args->num_worlds;

    Config cfg = Traits::makeConfig(*args);
    std::vector<WorldInit> inits(args->num_worlds);
    Traits::makeInits(*args, inits.data());

#ifdef SIM_BACKEND_REF_CPU
    h->exec = new SimHandle::Exec(
        madrona::ThreadPoolExecutor::Config {
            .numWorlds = args->num_worlds,
            .numExportedBuffers = Traits::numExports,
            .numWorkers = args->num_workers,
        },
        cfg, inits.data(), Traits::numTaskGraphs);
#else
    auto ctx = madrona::MWCudaExecutor::initCUDA(args->gpu_id);

    // (simulators with a batch renderer pass Optional<CudaBatchRenderConfig>,
    // reference mw_gpu.hpp:106-110)
    madrona::Optional<madrona::CudaBatchRenderConfig> render_cfg =
        simmgr::renderConfig<Traits>(*args);

    h->exec = new madrona::MWCudaExecutor({
        .worldInitPtr = inits.data(),
        .numWorldInitBytes = (uint32_t)sizeof(WorldInit),
        .userConfigPtr = (void *)&cfg,
        .numUserConfigBytes = (uint32_t)sizeof(Config),
        .numWorldDataBytes = (uint32_t)sizeof(Sim),
        .worldDataAlignment = (uint32_t)alignof(Sim),
        .numWorlds = args->num_worlds,
        .numTaskGraphs = Traits::numTaskGraphs,
        .numExportedBuffers = Traits::numExports,
    }, {
        {}, {}, madrona::CompileConfig::OptMode::LTO,
    }, ctx, render_cfg);

    h->stepGraph = simmgr::buildStepGraph<Traits>(*h->exec);
#endif

    Traits::describeTensors(h->tensors, args->num_worlds);
    Traits::describeColumns(h->columns);

    return h;
}

void sim_destroy(SimHandle *h)
{
    if (!h) return;
#ifndef SIM_BACKEND_REF_CPU
    delete h->stepGraph;
    for (auto *g : h->probeGraphs) delete g;
    delete h->renderGraph;
#endif
    delete h->exec;
    delete h;
}

const char *sim_backend(SimHandle *)
{
#ifdef SIM_BACKEND_REF_CPU
    return "ref_cpu";
#else
    return "hip";
#endif
}

void sim_step(SimHandle *h, uint32_t num_steps)
{
    for (uint32_t i = 0; i < num_steps; i++) {
        simmgr::preStep<SimTraits>();
#ifdef SIM_BACKEND_REF_CPU
        h->exec->run();
#else
        h->exec->run(*h->stepGraph);
#endif
    }
}

uint32_t sim_num_tensors(SimHandle *h)
{
    return (uint32_t)h->tensors.size();
}

int sim_tensor_info(SimHandle *h, uint32_t idx, SimTensorInfo *out)
{
    if (idx >= h->tensors.size()) return -1;
    const auto &t = h->tensors[idx];
    out->name = t.name.c_str();
    out->dtype = t.dtype;
    out->ndim = (int32_t)t.dims.size();
    for (int i = 0; i < 4; i++) {
        out->dims[i] = i < (int)t.dims.size() ? t.dims[i] : 1;
    }
#ifdef SIM_BACKEND_REF_CPU
    out->on_device = 0;
#else
    out->on_device = 1;
#endif
    return 0;
}

void *sim_tensor_ptr(SimHandle *h, uint32_t idx)
{
    if (idx >= h->tensors.size()) return nullptr;
    return h->exec->getExported(h->tensors[idx].slot);
}

static uint64_t simTensorBytes(const simmgr::TensorDesc &t)
{
    uint64_t n = simmgr::dtypeBytes(t.dtype);
    for (int64_t d : t.dims) n *= (uint64_t)d;
    return n;
}

int sim_tensor_read(SimHandle *h, uint32_t idx, void *dst, uint64_t num_bytes)
{
    if (idx >= h->tensors.size()) return -1;
    if (num_bytes > simTensorBytes(h->tensors[idx])) return -2;
    void *src = sim_tensor_ptr(h, idx);
#ifdef SIM_BACKEND_REF_CPU
    memcpy(dst, src, num_bytes);
    return 0;
#else
    return mwhip_memcpy_d2h(dst, src, num_bytes);
#endif
}

int sim_tensor_write(SimHandle *h, uint32_t idx, const void *src,
                     uint64_t num_bytes)
{
    if (idx >= h->tensors.size()) return -1;
    if (num_bytes > simTensorBytes(h->tensors[idx])) return -2;
    void *dst = sim_tensor_ptr(h, idx);
#ifdef SIM_BACKEND_REF_CPU
    memcpy(dst, src, num_bytes);
    return 0;
#else
    return mwhip_memcpy_h2d(dst, src, num_bytes);
#endif
}

uint32_t sim_num_columns(SimHandle *h)
{
    return (uint32_t)h->columns.cols.size();
}

int sim_column_info(SimHandle *h, uint32_t idx, SimColumnInfo *out)
{
    if (idx >= h->columns.cols.size()) return -1;
    const auto &c = h->columns.cols[idx];
    out->name = c.name.c_str();
    out->elem_bytes = c.elemBytes;
    out->is_float = c.isFloat;
    return 0;
}

int64_t sim_column_dump(SimHandle *h, uint32_t idx, void *dst,
                        uint64_t dst_bytes, int32_t *world_counts)
{
    if (idx >= h->columns.cols.size()) return -1;
    const auto &c = h->columns.cols[idx];
#ifdef SIM_BACKEND_REF_CPU
    madrona::StateManager &mgr = *h->exec->getWorldContext(0).getStateManager();
    uint64_t off = 0;
    int64_t total = 0;
    for (uint32_t w = 0; w < h->numWorlds; w++) {
        auto [ptr, n] = c.fn(mgr, w);
        uint64_t nb = (uint64_t)n * c.elemBytes;
        if (off + nb > dst_bytes) return -2;
        memcpy((char *)dst + off, ptr, nb);
        off += nb;
        total += n;
        world_counts[w] = (int32_t)n;
    }
    return total;
#else
    return mwhip_dump_column(h->exec->handle(), c.archetypeID, c.componentID,
                             dst, dst_bytes, world_counts);
#endif
}

int sim_hip_run_taskgraph(SimHandle *h, uint32_t taskgraph_id)
{
#ifdef SIM_BACKEND_REF_CPU
    (void)h; (void)taskgraph_id;
    return -1;
#else
    if (taskgraph_id >= SimTraits::numTaskGraphs) return -1;
    if (h->probeGraphs.size() <= taskgraph_id) {
        h->probeGraphs.resize(taskgraph_id + 1, nullptr);
    }
    if (h->probeGraphs[taskgraph_id] == nullptr) {
        h->probeGraphs[taskgraph_id] = new madrona::MWCudaLaunchGraph(
            h->exec->buildLaunchGraph(taskgraph_id));
    }
    h->exec->run(*h->probeGraphs[taskgraph_id]);
    return 0;
#endif
}

int32_t sim_render_geometry(float *vertices, uint32_t *indices,
                            uint32_t *vertex_offsets, uint32_t *triangle_offsets,
                            float *material_colors, int32_t *object_materials,
                            uint32_t *counts)
{
    const simmesh::MeshSet *m = simmgr::renderMeshes<SimTraits>();
    if (m == nullptr) {
        return 0;
    }
    auto copy = [](auto *dst, const auto &src) {
        if (dst != nullptr && !src.empty()) {
            memcpy(dst, src.data(), src.size() * sizeof(src[0]));
        }
    };
    copy(vertices, m->vertices);
    copy(indices, m->indices);
    copy(vertex_offsets, m->vertexOffsets);
    copy(triangle_offsets, m->triangleOffsets);
    copy(material_colors, m->materialColors);
    copy(object_materials, m->objectMaterials);
    if (counts != nullptr) {
        counts[0] = (uint32_t)(m->vertices.size() / 3);
        counts[1] = (uint32_t)(m->indices.size() / 3);
        counts[2] = (uint32_t)(m->materialColors.size() / 3);
    }
    return (int32_t)m->numObjects();
}

int32_t sim_render_geometry_ex(float *vertex_uvs, int32_t *triangle_materials,
                               int32_t *material_textures, uint32_t *texture_dims,
                               uint8_t *texels, uint32_t *counts)
{
    const simmesh::MeshSet *m = simmgr::renderMeshes<SimTraits>();
    if (m == nullptr) {
        return 0;
    }
    auto copy = [](auto *dst, const auto &src) {
        if (dst != nullptr && !src.empty()) {
            memcpy(dst, src.data(), src.size() * sizeof(src[0]));
        }
    };
    copy(vertex_uvs, m->vertexUVs);
    copy(triangle_materials, m->triangleMaterials);
    copy(material_textures, m->materialTextures);
    size_t bytes = 0;
    for (size_t t = 0; t < m->textures.size(); t++) {
        if (texture_dims != nullptr) {
            texture_dims[2 * t] = m->textures[t].width;
            texture_dims[2 * t + 1] = m->textures[t].height;
        }
        if (texels != nullptr) {
            memcpy(texels + bytes, m->textures[t].rgba8.data(),
                   m->textures[t].rgba8.size());
        }
        bytes += m->textures[t].rgba8.size();
    }
    if (counts != nullptr) {
        counts[0] = (uint32_t)m->textures.size();
        counts[1] = (uint32_t)bytes;
    }
    return (int32_t)m->textures.size();
}

uint64_t sim_hip_render_graph(SimHandle *h)
{
#ifdef SIM_BACKEND_REF_CPU
    (void)h;
    return 0;
#else
    if (h->renderGraph == nullptr) {
        h->renderGraph =
            new madrona::MWCudaLaunchGraph(h->exec->buildRenderGraph());
    }
    return h->renderGraph->handle();
#endif
}

int sim_hip_render(SimHandle *h)
{
#ifdef SIM_BACKEND_REF_CPU
    (void)h;
    return -1;
#else
    (void)sim_hip_render_graph(h);
    h->exec->run(*h->renderGraph);
    return 0;
#endif
}

int64_t sim_column_dump_raw(SimHandle *h, uint32_t idx, void *dst,
                            uint64_t dst_bytes)
{
#ifdef SIM_BACKEND_REF_CPU
    (void)h; (void)idx; (void)dst; (void)dst_bytes;
    return -1;
#else
    if (idx >= h->columns.cols.size()) return -1;
    const auto &c = h->columns.cols[idx];
    return mwhip_dump_column_raw(h->exec->handle(), c.archetypeID,
                                 c.componentID, dst, dst_bytes);
#endif
}

uint64_t sim_hip_step_graph(SimHandle *h)
{
#ifdef SIM_BACKEND_REF_CPU
    (void)h;
    return 0;
#else
    return h->stepGraph->handle();
#endif
}

void *sim_hip_exec(SimHandle *h)
{
#ifdef SIM_BACKEND_REF_CPU
    (void)h;
    return nullptr;
#else
    return (void *)h->exec->handle();
#endif
}

}

/* C API every synthetic simulator in sims/ exposes, identically, from
 *   (a) oracle/_ref/lib<sim>_ref.so   -- linked against the REFERENCE CPU backend
 *       (madrona::TaskGraphExecutor, reference include/madrona/mw_cpu.hpp:73-110)
 *   (b) madrona_amd/_build/lib<sim>_hip.so -- linked against libmadrona_hip.so
 *       (madrona::MWHipExecutor, the MI355X backend).
 * Tests / bench.py load both with ctypes and diff what they return.
 */
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIM_API __attribute__((visibility("default")))

typedef struct SimHandle SimHandle;

typedef struct SimCreateArgs {
    uint32_t num_worlds;
    uint32_t seed;
    int32_t gpu_id;          /* HIP backend only */
    uint32_t num_workers;    /* reference CPU backend only; 1 => bit-exact ids */
    uint32_t world_base;     /* global index of local world 0 (multi-GPU sharding) */
    uint32_t flags;          /* simulator specific */
} SimCreateArgs;

/* dtype codes follow madrona::py::TensorElementType
 * (reference include/madrona/py/utils.hpp:58-66) */
enum { SIM_U8 = 0, SIM_I8 = 1, SIM_I16 = 2, SIM_I32 = 3, SIM_I64 = 4,
       SIM_F16 = 5, SIM_F32 = 6 };

typedef struct SimTensorInfo {
    const char *name;
    int32_t dtype;
    int32_t ndim;
    int64_t dims[4];
    int32_t on_device;   /* 1: ptr is a device pointer */
} SimTensorInfo;

typedef struct SimColumnInfo {
    const char *name;        /* "Archetype.Component" */
    uint32_t elem_bytes;
    int32_t is_float;        /* 1: compare with fp tolerance, 0: bit exact */
} SimColumnInfo;

SIM_API SimHandle *sim_create(const SimCreateArgs *args);
SIM_API void sim_destroy(SimHandle *h);
SIM_API const char *sim_backend(SimHandle *h);     /* "ref_cpu" | "hip" */
SIM_API void sim_step(SimHandle *h, uint32_t num_steps);

SIM_API uint32_t sim_num_tensors(SimHandle *h);
SIM_API int sim_tensor_info(SimHandle *h, uint32_t slot, SimTensorInfo *out);
SIM_API void *sim_tensor_ptr(SimHandle *h, uint32_t slot);
/* copy tensor `slot` to / from host memory regardless of backend */
SIM_API int sim_tensor_read(SimHandle *h, uint32_t slot, void *dst, uint64_t num_bytes);
SIM_API int sim_tensor_write(SimHandle *h, uint32_t slot, const void *src, uint64_t num_bytes);

/* Parity dumps: column idx of the registered dump list
 * copied to host, rows grouped by world in world order; world_counts[w] rows
 * for world w.  Returns total rows, or -1 on error / -2 if dst is too small. */
SIM_API uint32_t sim_num_columns(SimHandle *h);
SIM_API int sim_column_info(SimHandle *h, uint32_t idx, SimColumnInfo *out);
SIM_API int64_t sim_column_dump(SimHandle *h, uint32_t idx, void *dst, uint64_t dst_bytes,
                        int32_t *world_counts);

/* HIP backend only (-1 on the reference): replays ONE task graph of the
 * simulator (a test probe: e.g. a lone sort node), and copies a column in TABLE
 * order -- every row below the table's row count, destroyed ones included, no
 * grouping by world -- to look at the table between nodes.  Returns rows. */
SIM_API int sim_hip_run_taskgraph(SimHandle *h, uint32_t taskgraph_id);
SIM_API int64_t sim_column_dump_raw(SimHandle *h, uint32_t idx, void *dst,
                                    uint64_t dst_bytes);

/* The meshes and materials the simulator hands to the batch ray caster (for the
 * tests' oracle).  Any pointer may be NULL; returns the number of objects (0:
 * the simulator draws nothing) and, through counts, { vertices, triangles,
 * materials }. */
SIM_API int32_t sim_render_geometry(float *vertices, uint32_t *indices,
                                    uint32_t *vertex_offsets,
                                    uint32_t *triangle_offsets,
                                    float *material_colors,
                                    int32_t *object_materials, uint32_t *counts);

/* ... and their per-vertex uvs, per-triangle materials (read for objects whose
 * object material is -1), per-material texture ids and RGBA8 textures.
 * texture_dims: width, height per texture; texels: all textures back to back.
 * counts: { textures, texel bytes }.  Returns the number of textures. */
SIM_API int32_t sim_render_geometry_ex(float *vertex_uvs, int32_t *triangle_materials,
                                       int32_t *material_textures,
                                       uint32_t *texture_dims, uint8_t *texels,
                                       uint32_t *counts);

/* HIP backend only: the batch ray caster's pass (MWCudaExecutor::
 * buildRenderGraph) over the tables as the last step left them; outputs land in
 * the simulator's "rgb" / "depth" tensors.  sim_hip_render runs it and waits;
 * sim_hip_render_graph returns its launch-graph handle (0 on the reference). */
SIM_API int sim_hip_render(SimHandle *h);
SIM_API uint64_t sim_hip_render_graph(SimHandle *h);

/* HIP backend only (NULL/0 on the reference): opaque mwhip_exec* for profiling */
SIM_API void *sim_hip_exec(SimHandle *h);
/* launch-graph handle of the per-step graph inside that executor (0 on the reference) */
SIM_API uint64_t sim_hip_step_graph(SimHandle *h);

#ifdef __cplusplus
}
#endif

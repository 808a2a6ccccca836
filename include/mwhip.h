/* mwhip.h -- C ABI of libmadrona_hip.so, the MI355X (gfx950) many-world ECS
 * task-graph backend.
 *
 * The reference has no FFI for this path: its boundary is the C++ class
 * madrona::MWCudaExecutor (reference include/madrona/mw_gpu.hpp:98-164) plus
 * device code that it JIT-compiles with NVRTC.  This header is the C-ABI that
 * a maintainer would bind instead; madrona_amd/include/madrona/mw_gpu.hpp is a
 * header-only C++ shim over it with the reference's class/method names, so
 * existing simulators compile unchanged.  Every entry point cites the
 * reference interface it replaces.
 *
 * Conventions: plain pointers and sizes only; functions returning int return
 * 0 on success and a negative code on error (the reference aborts through
 * FATAL()/REQ_CUDA(); the C++ shim restores that behaviour by aborting on a
 * non-zero code).  One executor drives one GPU from one host thread.
 */
#ifndef MWHIP_H
#define MWHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MWHIP_ABI_VERSION 7u   /* 7: mwhip_node_desc::pfor_group_kernel (a node's body is only called from the group kernel of the code object that defines it); 6: mwhip_node_desc::write_mask (same-dependency nodes whose signatures clash keep their own launches); 5: mwhip_node_desc::pfor_body, mwhip_pfor_body(), mwhip_set_pfor_group_kernel() (side-by-side ParallelFor nodes in one launch); 4: io_declared */

typedef struct mwhip_exec mwhip_exec; /* opaque; == MWCudaExecutor::Impl */

/* == madrona::StateConfig (reference include/madrona/mw_gpu.hpp:25-51), POD,
 * plus the gpu id that MWCudaExecutor::initCUDA(gpu_id) (:104) selected. */
typedef struct mwhip_state_config {
    const void *world_init_ptr;     /* host: num_worlds * num_world_init_bytes */
    uint32_t num_world_init_bytes;
    const void *user_config_ptr;    /* host */
    uint32_t num_user_config_bytes;
    uint32_t num_world_data_bytes;
    uint32_t world_data_alignment;
    uint32_t num_worlds;
    uint32_t num_task_graphs;
    uint32_t num_exported_buffers;
    int32_t gpu_id;
    /* == the parts of madrona::CudaBatchRenderConfig (mw_gpu.hpp:77-96) the
     * render-prep ECS systems need (Optional<CudaBatchRenderConfig> of the
     * MWCudaExecutor constructor, cuda_exec.cpp:2333): side of the square
     * ray-caster outputs (0 = ray caster off), colour + depth or depth only,
     * and one object-space root AABB (6 floats: min xyz, max xyz) per object id
     * for the top-level BVH leaf boxes (host pointer, copied; may be NULL). */
    uint32_t raycast_output_resolution;
    uint32_t raycast_rgbd;
    const float *object_root_aabbs;
    uint32_t num_object_root_aabbs;
    uint32_t pad_;
    /* triangle geometry + materials of the renderable objects (== the
     * geoBVHData / materialData of CudaBatchRenderConfig, which the reference
     * fills from Embree-built MeshBVHs): host memory, copied; the executor
     * builds its own bottom-level BVHs.  NULL: no ray caster. */
    const struct mwhip_render_geometry *render_geometry;
    /* most views a world will ever hold (0: unknown).  Sizes the render-target
     * table: at 64 x 64 RGB-D a row is 32 KiB, and a table of unknown size gets
     * the executor's default rows per world. */
    uint32_t raycast_max_views_per_world;
    uint32_t pad2_;
} mwhip_state_config;

typedef struct mwhip_render_geometry {
    uint32_t num_objects;
    uint32_t num_materials;
    const float *vertices;              /* xyz per vertex, all objects */
    const uint32_t *indices;            /* 3 per triangle, object-local vertex ids */
    const uint32_t *object_vertex_offset;   /* [num_objects + 1] into vertices */
    const uint32_t *object_triangle_offset; /* [num_objects + 1] into indices / 3 */
    const int32_t *object_material;     /* [num_objects] material id of the whole
                                         * mesh (reference MeshBVH::materialIDX,
                                         * mesh_bvh.hpp:304); -1: per triangle
                                         * (triangle_material) or, without those,
                                         * none (white); NULL: -1 everywhere */
    const float *material_color;        /* rgb per material */
    /* ---- per-triangle materials and textures (reference
     * MeshBVH::LeafMaterial / BVHVertex::uv, mesh_bvh.hpp:167-178; Material,
     * :148-156; bvh_raycast.cpp:772-800).  All optional (NULL / 0). ---- */
    const float *vertex_uv;             /* uv per vertex, all objects */
    const int32_t *triangle_material;   /* per triangle, all objects: material id
                                         * or -1 (none: white); read for objects
                                         * whose object_material is -1 */
    const int32_t *material_texture;    /* [num_materials] texture id or -1 */
    uint32_t num_textures;
    uint32_t pad_;
    const struct mwhip_texture *textures;   /* [num_textures] */
} mwhip_render_geometry;

/* An RGBA8 texture, row 0 first (what the reference uploads into a cudaArray of
 * uchar4 and samples through a texture object with wrap addressing, linear
 * filtering and normalised coordinates, render/asset_processor.cpp:312-345). */
typedef struct mwhip_texture {
    uint32_t width;
    uint32_t height;
    const uint8_t *rgba8;               /* width * height * 4 bytes, host memory */
} mwhip_texture;

/* Which tables the batch ray caster reads and writes: set once by
 * RenderingSystem::registerTypes (the type ids are assigned there).
 * Replaces the pointers the reference's render BVH kernels find in their
 * BVHParams (src/mw/device/bvh_raycast.cpp, src/mw/cuda_exec.cpp). */
typedef struct mwhip_render_layout {
    uint32_t renderable_archetype;      /* InstanceData, MortonCode, TLBVHNode */
    uint32_t camera_archetype;          /* PerspectiveCameraData */
    uint32_t light_archetype;           /* LightDesc */
    uint32_t output_archetype;          /* RGBOutputBuffer, DepthOutputBuffer */
    uint32_t instance_component;
    uint32_t morton_component;
    uint32_t tlbvh_component;
    uint32_t camera_component;
    uint32_t light_component;
    uint32_t rgb_component;
    uint32_t depth_component;
    uint32_t pad_;
} mwhip_render_layout;

/* What the simulator's offline-compiled HIP translation unit hands to the
 * executor.  Replaces CompileConfig::userSources + the three entry kernels
 * instantiated by MADRONA_BUILD_MWGPU_ENTRY (reference
 * src/mw/device/include/madrona/mw_gpu_entry.hpp:12-91).  registerTypes and
 * setupTasks run on the HOST here (they only describe types and graph
 * topology); world constructors run on the device. */
typedef struct mwhip_user_entry {
    uint32_t abi_version;
    /* calls WorldT::registerTypes(ECSRegistry&, cfg) -> mwhip_register_* */
    void (*register_types)(mwhip_exec *exec, const void *user_cfg_host);
    /* calls WorldT::setupTasks(TaskGraphManager&, cfg) -> mwhip_tg_add_node */
    void (*setup_tasks)(mwhip_exec *exec, const void *user_cfg_host);
    /* host stub of __global__ void(ecs_state*, const void *cfg_dev,
     *   const void *inits_dev, int32_t num_worlds): placement-new of
     *   WorldT(ctx, cfg, init[w]) for one world per thread
     *   (== entryKernels::initWorlds, mw_gpu_entry.hpp:37-56) */
    const void *init_worlds_kernel;
    /* stores the device ecs_state pointer into the user module's
     * __device__ global (== GPUImplConsts::get().stateManagerAddr) */
    void (*bind_device_state)(void *ecs_state_dev);
} mwhip_user_entry;

/* MWCudaExecutor::MWCudaExecutor(state_cfg, compile_cfg, cu_ctx)
 * (reference src/mw/cuda_exec.cpp:2333-2420): allocates the ECS, runs
 * registerTypes, constructs all worlds on the device, runs setupTasks. */
int mwhip_create(const mwhip_state_config *cfg, const mwhip_user_entry *entry,
                 mwhip_exec **out);
/* MWCudaExecutor::~MWCudaExecutor (cuda_exec.cpp:2484-2530) */
void mwhip_destroy(mwhip_exec *exec);
/* last error text for this thread ("" if none) */
const char *mwhip_last_error(void);

/* ---- ECS registry: StateManager::register* ------------------------------
 * reference src/mw/device/state.cpp:154-378, device state.inl:7-158 */
int mwhip_register_component(mwhip_exec *exec, uint32_t component_id,
                             uint32_t alignment, uint32_t num_bytes);
/* bundle ids carry bit 31 (reference state.hpp:197); nested bundles are
 * flattened in place */
int mwhip_register_bundle(mwhip_exec *exec, uint32_t bundle_id,
                          const uint32_t *component_ids, uint32_t num_components);
/* archetype_flags: the reference's ArchetypeFlags (ecs_flags.hpp) plus, in bit
 * 31, "registered by registerSingleton": exactly one row per world, no head
 * room, never grows. */
#define MWHIP_ARCHETYPE_SINGLETON 0x80000000u
int mwhip_register_archetype(mwhip_exec *exec, uint32_t archetype_id,
                             const uint32_t *component_ids,
                             const uint32_t *component_flags, /* may be NULL */
                             uint32_t num_components, uint32_t archetype_flags,
                             uint32_t max_num_entities_per_world);
/* registerSingleton<T>: archetype with one row per world, entity ids assigned
 * in world order (reference device state.inl:136-151, CPU state.inl:163-179) */
int mwhip_register_singleton(mwhip_exec *exec, uint32_t archetype_id,
                             uint32_t component_id);
/* ECSRegistry::exportColumn (registry.inl:47-51): returns the device address
 * of the column, stable for the executor's lifetime, also stored in `slot` */
void *mwhip_export_column(mwhip_exec *exec, uint32_t archetype_id,
                          uint32_t component_id, int32_t slot);
/* StateManager::makeQuery (device/state.cpp:380-440): appends
 * [archetype, col idx per component]* to the query table; returns its offset */
#define MWHIP_QUERY_ALL_SINGLETON 1u  /* every matched archetype has 1 row/world */
int mwhip_make_query(mwhip_exec *exec, const uint32_t *component_ids,
                     uint32_t num_components, uint32_t *offset_out,
                     uint32_t *num_matching_out, uint32_t *flags_out);
/* Executor-owned device scratch (freed by mwhip_destroy). */
void *mwhip_alloc_device(mwhip_exec *exec, uint64_t num_bytes, int zero);
/* Device memory that outlives any executor (asset tables uploaded before the
 * executor exists: PhysicsLoader, reference src/physics/physics_loader.cpp).
 * gpu_id selects the device; return nullptr / nonzero on failure. */
void *mwhip_raw_alloc(int gpu_id, uint64_t num_bytes);
void mwhip_raw_free(int gpu_id, void *device_ptr);
int mwhip_raw_copy_h2d(int gpu_id, void *dst_device, const void *src_host,
                       uint64_t num_bytes);
int mwhip_raw_copy_d2h(int gpu_id, void *dst_host, const void *src_device,
                       uint64_t num_bytes);
/* Small table of module-private device pointers inside ecs_state
 * (ecs_state::moduleData[slot], slot < 4); e.g. the physics module's scratch. */
int mwhip_set_module_data(mwhip_exec *exec, uint32_t slot, void *device_ptr);
void *mwhip_get_module_data(mwhip_exec *exec, uint32_t slot);
/* rows every column of the archetype's table can ever hold (the address space
 * reserved for it; memory is mapped behind it as the table grows) */
uint32_t mwhip_archetype_capacity(mwhip_exec *exec, uint32_t archetype_id);
/* how many times a table has been grown so far (tests / monitoring) */
uint32_t mwhip_num_table_growths(mwhip_exec *exec);
/* device address of the archetype's table header (mwhip::TableHdr) */
void *mwhip_table_header(mwhip_exec *exec, uint32_t archetype_id);
/* copies `count` words of the query table starting at `offset` */
int mwhip_get_query_data(mwhip_exec *exec, uint32_t offset, uint32_t count,
                         uint32_t *out);
/* address of the device-resident ecs_state (valid after mwhip_create's
 * register phase) and of per-world user data */
void *mwhip_device_state(mwhip_exec *exec);
void *mwhip_world_data(mwhip_exec *exec, uint32_t world_idx);
uint32_t mwhip_num_worlds(const mwhip_exec *exec);
/* the ray caster's output resolution / RGBD flag the executor was created with
 * (registerTypes sizes the render-target components from them,
 * reference src/render/ecs_system.cpp:385-404) */
void mwhip_render_config(const mwhip_exec *exec, uint32_t *resolution_out,
                         uint32_t *rgbd_out);
uint32_t mwhip_render_max_views(const mwhip_exec *exec);
/* Host-side preview of what mwhip_create builds from a geometry description (no
 * GPU needed): per object the number of bottom-level BVH nodes, whether the mesh
 * is its own axis-aligned bounds seen from outside (such objects are intersected
 * as slabs + the two triangles of the entry face), and the object-space bounds
 * (6 floats each).  Any output may be NULL.  Returns 0, or -1 with
 * mwhip_last_error() set when the description is malformed. */
int mwhip_render_geometry_info(const struct mwhip_render_geometry *geometry,
                               uint32_t *num_nodes_out, uint32_t *is_box_out,
                               float *bounds_out);
int mwhip_set_render_layout(mwhip_exec *exec, const mwhip_render_layout *layout);
/* MWCudaExecutor::buildRenderGraph (reference mw_gpu.hpp:140, cuda_exec.cpp:
 * 2294-2331): a launch graph that builds every world's top-level BVH over its
 * (Morton-sorted) instances and ray-casts every view into the render-target
 * columns.  Run it after the step graph (mwhip_run / mwhip_run_async). */
int mwhip_build_render_graph(mwhip_exec *exec, uint64_t *graph_out);
uint32_t mwhip_num_task_graphs(const mwhip_exec *exec);

/* ---- task graph: TaskGraph::Builder -------------------------------------
 * reference src/mw/device/taskgraph_utils.cpp:30-146 + taskgraph.inl:59-111 */
enum mwhip_node_kind {
    MWHIP_NODE_KERNEL = 0,        /* user/system kernel: ParallelFor, custom node */
    MWHIP_NODE_SORT_ARCHETYPE = 1,/* SortArchetypeNode<A,C> (sort_archetype.cpp) */
    MWHIP_NODE_CLEAR_TMP = 2,     /* ClearTmpNode<A> (taskgraph_utils.cpp:171-190) */
    MWHIP_NODE_RESET_TMP_ALLOC = 3,/* ResetTmpAllocNode (:216-230) */
    MWHIP_NODE_RECYCLE = 4,       /* RecycleEntitiesNode (:192-214); no-op here */
    /* In-place exclusive prefix sum over up to 8 device arrays treated as one
     * sequence (node_data = mwhip_scan_params).  Building block of the
     * deterministic "count -> scan -> fill" emission of temporaries that
     * replaces arrival-order atomics. */
    MWHIP_NODE_EXCLUSIVE_SCAN = 5
};

#define MWHIP_SCAN_MAX_SEGMENTS 8
typedef struct mwhip_scan_params {
    uint32_t num_segments;
    uint32_t capacity;              /* total_out is clamped to this; overflow raises
                                     * the table-overflow device error */
    uint32_t *data[MWHIP_SCAN_MAX_SEGMENTS];         /* device, scanned in place */
    const int32_t *lengths[MWHIP_SCAN_MAX_SEGMENTS]; /* device-resident lengths */
    int32_t *total_out;             /* device; e.g. a table's row count */
    uint32_t *needs_sort_out;       /* optional device flag set to 1 if total > 0 */
} mwhip_scan_params;

enum mwhip_count_mode {
    MWHIP_COUNT_QUERY_ROWS = 0,   /* one invocation per row of the query's tables */
    MWHIP_COUNT_FIXED = 1,        /* fixed_count invocations */
    MWHIP_COUNT_PER_WORLD = 2     /* one invocation per world */
};

/* Query resolution passed BY VALUE in the kernel-argument segment of
 * ParallelFor kernels: the matched tables' header addresses and the column
 * index of every component, so a kernel starts with one round trip (row count +
 * column pointers) instead of walking ecs_state -> query table -> table.
 * Queries matching more archetypes / components than fit set num_inline = 0
 * and the kernel walks the query table (mwhip_make_query) instead. */
#define MWHIP_PFOR_MAX_INLINE 4
#define MWHIP_PFOR_MAX_COMPONENTS 24
typedef struct mwhip_pfor_args {
    uint32_t num_matching;
    uint32_t num_inline;
    void *tables[MWHIP_PFOR_MAX_INLINE];            /* device table headers */
    uint16_t columns[MWHIP_PFOR_MAX_INLINE][MWHIP_PFOR_MAX_COMPONENTS];
    /* Non-NULL for nodes whose system can append rows (makeEntity /
     * makeTemporary reachable from the kernel): per-node device state through
     * which the first workgroup to arrive fixes the row count of every matched
     * table for the whole launch -- the reference evaluates numInvocations
     * once per node (device taskgraph.inl:164-188), so rows created during a
     * node are never visited by it.  The kernel is then launched with
     * 4 * num_matching bytes of dynamic LDS. */
    void *row_sync;
} mwhip_pfor_args;

typedef struct mwhip_node_desc {
    uint32_t kind;
    const char *name;             /* for profiles; copied */
    /* MWHIP_NODE_KERNEL: host stub of
     *   __global__ void(ecs_state*, void *node_data_dev, uint32_t a0, uint32_t a1)
     * or, when wants_pfor_args != 0 (count_mode MWHIP_COUNT_QUERY_ROWS),
     *   __global__ void(ecs_state*, void *node_data_dev, uint32_t a0, uint32_t a1,
     *                   mwhip_pfor_args query) */
    const void *kernel;
    uint32_t wants_pfor_args;
    int32_t node_data_id;         /* from mwhip_tg_add_node_data, or -1 */
    uint32_t arg0, arg1;
    uint32_t count_mode;
    uint32_t fixed_count;
    uint32_t threads_per_invocation;
    uint32_t query_offset;        /* MWHIP_COUNT_QUERY_ROWS */
    uint32_t num_matching;
    uint32_t bytes_per_row;       /* algorithmic bytes for rooflines */
    /* SORT / CLEAR_TMP */
    uint32_t archetype_id;
    uint32_t component_id;
    /* bytes_per_row comes from a read / write set declared next to the system
     * (1; madrona::mwhip::systemIO, taskgraph.inl) or from the signature rule
     * (0: const T & = read, T & = read + write -- an upper bound).  SURVEY §8d;
     * the reference's per-row contract is device taskgraph.inl:164-300. */
    uint32_t io_declared;
    /* ParallelFor nodes (items_per_invocation == 1): device address of the
     * node's body as a __device__ function (mwhip_pfor_body), or NULL.  Nodes
     * that name the SAME dependencies, cannot append rows and carry a body are
     * run side by side in ONE launch of the simulator's group kernel
     * (mwhip_set_pfor_group_kernel): blockIdx.y selects the node.  The reference
     * runs independent nodes of its task graph one after the other
     * (device taskgraph.cpp:142-317); results are the same by the independence
     * the simulator declared with its dependency lists. */
    const void *pfor_body;
    /* ParallelFor nodes: bit i set = the system may write component i of its
     * query (in query order): a non-const reference in its signature.  The
     * runtime does not put two nodes into one launch when one of them writes a
     * component the other one names on a table both match (the reference runs
     * them one after the other, device taskgraph.cpp:142-317: same-dependency
     * siblings may still rely on registration order).  0 with pfor_body set
     * means "writes nothing"; nodes built without a signature pass ~0u. */
    uint32_t write_mask;
    /* ParallelFor nodes with a pfor_body: the group kernel of the code object
     * that defines the body (its registers, scratch and LDS were sized with this
     * body in it).  Only nodes that name the SAME group kernel share a launch,
     * and that kernel is the one launched.  NULL: the executor's default,
     * mwhip_set_pfor_group_kernel(). */
    const void *pfor_group_kernel;
} mwhip_node_desc;

/* Members of a grouped launch, in device memory (the group kernel's argument). */
#define MWHIP_PFOR_GROUP_MAX 6
typedef struct mwhip_pfor_group {
    uint32_t count;
    uint32_t pad_;
    const void *body[MWHIP_PFOR_GROUP_MAX];        /* void (*)(ecs_state *, uint32_t,
                                                     * uint32_t, const mwhip_pfor_args *) */
    uint32_t query_offset[MWHIP_PFOR_GROUP_MAX];
    uint32_t num_matching_and_flags[MWHIP_PFOR_GROUP_MAX];
    mwhip_pfor_args query[MWHIP_PFOR_GROUP_MAX];
} mwhip_pfor_group;

/* (the reference's TaskGraph::maxNodeDataBytes is 256; larger here because a
 * node's data is what its kernel reaches with one load from its arguments) */
#define MWHIP_MAX_NODE_DATA_BYTES 2048u

/* TaskGraph::Builder::constructNodeData (taskgraph.inl:43-57): copies a node
 * data block (<= MWHIP_MAX_NODE_DATA_BYTES) to the device; several nodes may
 * share one block.
 * Returns the data id >= 0 or a negative error. */
int32_t mwhip_tg_add_node_data(mwhip_exec *exec, uint32_t taskgraph_id,
                               const void *data, uint32_t num_bytes);
/* device address of a node data block (TaskGraph::getNodeData) */
void *mwhip_tg_node_data(mwhip_exec *exec, uint32_t taskgraph_id, int32_t data_id);
/* TaskGraph::Builder::registerNode (taskgraph_utils.cpp:30-72).  Nodes are
 * ordered with the reference's own "first unqueued node whose dependencies
 * are queued" rule (taskgraph_utils.cpp:74-146).  Returns node id >= 0. */
int32_t mwhip_tg_add_node(mwhip_exec *exec, uint32_t taskgraph_id,
                          const mwhip_node_desc *desc, const int32_t *deps,
                          uint32_t num_deps);

/* Device address of a ParallelFor kernel's body: launches `kernel` (host stub of
 * a parallelForKernel instantiation) once in its report mode
 * (num_matching_and_flags = 0xFFFFFFFF: thread 0 stores the address of the
 * instantiation's __device__ body at node_data_dev and returns).  NULL on
 * error. */
const void *mwhip_pfor_body(mwhip_exec *exec, const void *kernel);
/* The simulator's group kernel -- __global__ void(ecs_state *, const
 * mwhip_pfor_group *), in the SAME code object as the bodies (device function
 * addresses are only called from the module that defines them). */
int mwhip_set_pfor_group_kernel(mwhip_exec *exec, const void *kernel);

/* compute units of the executor's device (kernel nodes that size a persistent
 * grid themselves) */
uint32_t mwhip_device_cus(const mwhip_exec *exec);

/* ---- execution ----------------------------------------------------------*/
/* MWCudaExecutor::buildLaunchGraph(Span<const uint32_t>, stat_name)
 * (cuda_exec.cpp:2174-2292): captures one hipGraph running the given task
 * graphs back to back.  graph_out is a handle owned by the executor. */
int mwhip_build_launch_graph(mwhip_exec *exec, const uint32_t *taskgraph_ids,
                             uint32_t num_taskgraphs, const char *stat_name,
                             uint64_t *graph_out);
void mwhip_free_launch_graph(mwhip_exec *exec, uint64_t graph);
/* MWCudaExecutor::run (cuda_exec.cpp:2756-2794): synchronous */
int mwhip_run(mwhip_exec *exec, uint64_t graph);
/* MWCudaExecutor::runAsync (:2796-2800).  Replays of ONE executor's launch
 * graphs run one at a time, in the order they were queued -- queue them on one
 * stream (the executor's own, mwhip_stream, or one of the caller's): nodes that
 * can append rows agree on their row counts through a per-replay tag, and two
 * graphs of one executor in flight at once on different streams would not
 * (they time out with a device error, they do not hang). */
int mwhip_run_async(mwhip_exec *exec, uint64_t graph, void *hip_stream);
/* the executor's private stream (cu::makeStream, cuda_exec.cpp:2342) */
void *mwhip_stream(mwhip_exec *exec);
/* Waits for everything queued on the executor's private stream (replays started
 * with mwhip_run_async on mwhip_stream()) and returns the health of the last
 * replay like mwhip_run does (new: the reference leaves this to the caller's
 * cudaStreamSynchronize). */
int mwhip_synchronize(mwhip_exec *exec);
/* MWCudaExecutor::getExported (cuda_exec.cpp:2802-2805) */
void *mwhip_get_exported(const mwhip_exec *exec, uint32_t slot);

/* ---- introspection for parity dumps / measurement (new) ----------------- */
int32_t mwhip_num_rows(mwhip_exec *exec, uint32_t archetype_id);
/* Copies column `component_id` of `archetype_id`, rows grouped by world in
 * world order (== concatenating the reference CPU backend's per-world
 * tables, state.inl:368-377); world_counts[num_worlds] receives rows/world.
 * Returns total rows, -1 on error, -2 if dst is too small. */
int64_t mwhip_dump_column(mwhip_exec *exec, uint32_t archetype_id,
                          uint32_t component_id, void *dst, uint64_t dst_bytes,
                          int32_t *world_counts);
/* The same column in TABLE order: every row below the table's row count
 * (destroyed rows included), no grouping by world.  Returns rows, -1 on error,
 * -2 if dst is too small. */
int64_t mwhip_dump_column_raw(mwhip_exec *exec, uint32_t archetype_id,
                              uint32_t component_id, void *dst,
                              uint64_t dst_bytes);
int mwhip_memcpy_d2h(void *dst_host, const void *src_dev, uint64_t num_bytes);
int mwhip_memcpy_h2d(void *dst_dev, const void *src_host, uint64_t num_bytes);
/* to host memory from wherever `src` lives (host or device: a renderer's asset
 * buffers, reference render/cuda_batch_render_assets.hpp, may be either) */
int mwhip_memcpy_any(void *dst_host, const void *src, uint64_t num_bytes);

/* Packs `num_columns` exported columns into one row-major record per row:
 * dst[row] = column 0's words | column 1's words | ...  (4-byte words;
 * words_per_row[c] of them from src_columns[c] + row * words_per_row[c]).  Queued
 * on the executor's stream, i.e. ordered after the replays queued before it --
 * the send buffer of the one observation all-gather per step of a multi-GPU run
 * (the reference hands the column itself to ncclAllGather).
 * Pointers are device pointers; the descriptor arrays are read before the call
 * returns. */
#define MWHIP_PACK_MAX_COLUMNS 16
/* The same packing as the LAST node of a copy of launch graph `base_graph`
 * (a foreign kernel between two replays costs ~35 us of lost launch
 * pipelining on this runtime; inside the graph it costs its own ~4 us).
 * Callers that overlap the collective with the next replay build two such
 * graphs, one per send buffer, and alternate. */
int mwhip_build_launch_graph_with_pack(mwhip_exec *exec, uint64_t base_graph,
                                       uint32_t num_columns,
                                       const void *const *src_columns,
                                       const uint32_t *words_per_row,
                                       uint32_t num_rows, void *dst,
                                       uint64_t *graph_out);
/* Makes `hip_stream` wait for every replay queued so far (mwhip_run_async on
 * the executor's stream) without putting anything on the executor's stream: the
 * last kernel of each replay bumps a counter in signal memory that the waiting
 * stream polls (hipStreamWaitValue32). */
int mwhip_stream_wait_replays(mwhip_exec *exec, void *hip_stream);
int mwhip_pack_rows(mwhip_exec *exec, uint32_t num_columns,
                    const void *const *src_columns,
                    const uint32_t *words_per_row, uint32_t num_rows,
                    void *dst);

/* Device-resident input ring: the k-th replay of a step graph of this executor
 * (any graph from mwhip_build_launch_graph; replays of the render graph neither
 * read nor advance the rings; k = 0, 1, ...) after this call starts by copying slot k % num_slots
 * of `ring` (num_slots x slot_bytes, device memory, whole dwords) into `dst` -- normally an exported
 * action column (mwhip_exported) --, so that a policy's outputs for the next
 * steps can be queued with the replays that consume them; nothing foreign sits
 * on the executor's stream between two graph launches (which costs 20-35 us of
 * launch pipelining each).  ring == NULL removes the ring of
 * `dst`.  The ring belongs to the caller and must stay allocated until it is
 * removed or the executor destroyed.  Rebuilds the launch graphs (handles stay
 * valid); waits for the executor's stream.  (No reference counterpart: its managers write the action
 * tensor between steps, include/madrona/mw_gpu.hpp:146 runAsync + PyTorch.) */
int mwhip_set_input_ring(mwhip_exec *exec, void *dst, const void *ring,
                         uint64_t slot_bytes, uint32_t num_slots);

/* Queues a one-wave marker kernel (benchWindowMarker) on the executor's stream:
 * a pair of them brackets a measurement window in a rocprofv3 kernel trace
 * (profiles/summarize_rocprof.py trims to it).  Measurement only. */
int mwhip_mark_window(mwhip_exec *exec, uint32_t id);

/* Per-kernel timing with HIP events on the executor's stream (replaces the
 * reference's device tracing, mw_gpu/tracing.hpp).  Runs the launch graph's
 * kernels eagerly `reps` times with an event pair around every kernel. */
typedef struct mwhip_kernel_stat {
    const char *name;       /* node name + kernel role, owned by the executor */
    uint32_t node_kind;
    uint32_t archetype_id;
    double avg_us;          /* mean duration of this kernel per step */
    double algo_bytes;      /* mean algorithmic bytes per launch */
    double rows;            /* mean rows / invocations processed per launch */
    uint32_t io_declared;   /* algo_bytes from a declared read / write set (1),
                             * from the signature rule (0) */
    uint32_t workgroups;    /* grid of the launch */
    uint32_t node_index;    /* position of the node in its task graph's execution
                             * order (the key of MADRONA_MWHIP_EXEC_CONFIG_FILE),
                             * 0xFFFFFFFF for kernels that are not a node's own */
    uint32_t pad_;
} mwhip_kernel_stat;
int32_t mwhip_profile(mwhip_exec *exec, uint64_t graph, uint32_t reps,
                      mwhip_kernel_stat *out, uint32_t max_out);

#ifdef __cplusplus
}
#endif
#endif /* MWHIP_H */

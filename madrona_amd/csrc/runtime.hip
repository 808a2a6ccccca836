// libmadrona_hip.so -- host runtime of the MI355X many-world ECS backend.
// Implements the C ABI declared in include/mwhip.h.
//
// Replaces, for the hot path only, the reference's GPU executor host side
// (src/mw/cuda_exec.cpp: state allocation :1721-1948, graph build :2174-2292,
// run :2756-2794) and the device-side registry / task-graph builder
// (src/mw/device/state.cpp:154-440, taskgraph_utils.cpp:30-146).  There is no
// runtime compiler, no megakernel and no host<->device mailbox: registration
// and graph construction are host code, every node is its own kernel, and a
// step is one hipGraph replay on the executor's private stream.
#include "exec_internal.hpp"

// ---- this translation unit: the C ABI (registry, queries, task-graph builder, create /
// destroy, run, introspection, profiling) ----

static thread_local std::string g_lastError;

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
MWHIP_RT int fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_lastError = buf;
    if (getenv("MADRONA_MWHIP_VERBOSE") != nullptr) {
        fprintf(stderr, "[mwhip] error %d: %s\n", code, buf);
    }
    return code;
}

MWHIP_RT uint32_t envU32(const char *name, uint32_t fallback)
{
    const char *v = getenv(name);
    if (v == nullptr || *v == '\0') return fallback;
    return (uint32_t)strtoul(v, nullptr, 10);
}

// ---------------------------------------------------------------------------
// registry
// ---------------------------------------------------------------------------
extern "C" const char *mwhip_last_error(void)
{
    return g_lastError.c_str();
}

extern "C" int mwhip_register_component(mwhip_exec *exec, uint32_t id,
                                        uint32_t alignment, uint32_t num_bytes)
{
    if (!exec->registrationOpen) {
        return fail(-1, "registerComponent outside registerTypes");
    }
    if (id >= kMaxComponents) {
        return fail(-2, "component id %u exceeds the limit of %u", id,
                    kMaxComponents);
    }
    if (exec->components.size() <= id) {
        exec->components.resize(id + 1);
    }
    exec->components[id] = ComponentRec { true, alignment, num_bytes };
    return 0;
}

static int flattenComponents(mwhip_exec *exec, const uint32_t *ids, uint32_t n,
                             std::vector<uint32_t> &out)
{
    for (uint32_t i = 0; i < n; i++) {
        uint32_t id = ids[i];
        if ((id & kBundleMask) != 0u) {
            auto it = exec->bundles.find(id);
            if (it == exec->bundles.end()) {
                return fail(-3, "bundle 0x%x used before registerBundle", id);
            }
            out.insert(out.end(), it->second.begin(), it->second.end());
        } else {
            if (id >= exec->components.size() ||
                    !exec->components[id].registered) {
                return fail(-3, "component %u used before registerComponent",
                            id);
            }
            out.push_back(id);
        }
    }
    return 0;
}

extern "C" int mwhip_register_bundle(mwhip_exec *exec, uint32_t bundle_id,
                                     const uint32_t *component_ids,
                                     uint32_t num_components)
{
    if (!exec->registrationOpen) {
        return fail(-1, "registerBundle outside registerTypes");
    }
    std::vector<uint32_t> flat;
    int rc = flattenComponents(exec, component_ids, num_components, flat);
    if (rc != 0) return rc;
    exec->bundles[bundle_id | kBundleMask] = std::move(flat);
    return 0;
}

static uint32_t defaultRowsPerWorld()
{
    return envU32("MADRONA_MWHIP_ROWS_PER_WORLD", 64);
}

extern "C" int mwhip_register_archetype(mwhip_exec *exec, uint32_t id,
                                        const uint32_t *component_ids,
                                        const uint32_t *component_flags,
                                        uint32_t num_components,
                                        uint32_t archetype_flags,
                                        uint32_t max_per_world)
{
    (void)component_flags;
    if (!exec->registrationOpen) {
        return fail(-1, "registerArchetype outside registerTypes");
    }
    if (id >= kMaxArchetypes) {
        return fail(-2, "archetype id %u exceeds the limit of %u", id,
                    kMaxArchetypes);
    }
    if (exec->archetypes.size() <= id) {
        exec->archetypes.resize(id + 1);
    }

    ArchetypeRec &arch = exec->archetypes[id];
    if (arch.registered) {
        return 0;   // idempotent, like the reference's CPU registry
    }

    arch = ArchetypeRec {};
    arch.id = id;
    arch.flags = archetype_flags;
    arch.maxPerWorld = max_per_world;
    int rc = flattenComponents(exec, component_ids, num_components, arch.comps);
    if (rc != 0) return rc;

    arch.numColumns = 2u + (uint32_t)arch.comps.size();
    if (arch.numColumns > kMaxColumns) {
        return fail(-2, "archetype %u has %u columns (limit %u)", id,
                    arch.numColumns, kMaxColumns);
    }

    const uint64_t W = exec->cfg.num_worlds;
    // Rows of one world that were destroyed and re-created coexist until the
    // next compaction, hence the 2x head room over the declared maximum.
    // (Only singletons are pinned to exactly one row per world: an ordinary
    // archetype declared with a maximum of 1 still needs two rows per world
    // while its entity is destroyed and re-created inside one step.)
    const bool pinned_rows = (archetype_flags & MWHIP_ARCHETYPE_SINGLETON) != 0u;
    uint64_t rows_per_world = pinned_rows ? 1 :
        (max_per_world > 0 ? 2ull * max_per_world : defaultRowsPerWorld());
    uint64_t capacity = std::max<uint64_t>(W * rows_per_world, 64);
    if (capacity > 0x7FFFFFF0ull) {
        return fail(-2, "archetype %u capacity overflow", id);
    }
    // Tables whose row count is not pinned to one per world can grow: address
    // space for tableGrowth x the initial rows (MADRONA_MWHIP_TABLE_GROWTH,
    // default 4; 1 = plain allocations), see growTables().
    uint64_t reserved = capacity;
    if (!pinned_rows && exec->tableGrowth > 1) {
        reserved = std::min<uint64_t>(capacity * exec->tableGrowth, 0x7FFFFFF0ull);
        // (test hook) start smaller than the rows the simulator declared
        if (const char *div = getenv("MADRONA_MWHIP_INITIAL_CAPACITY_DIV")) {
            uint64_t d = strtoull(div, nullptr, 10);
            if (d > 1) capacity = std::max<uint64_t>(capacity / d, 64);
        }
    }
    arch.capacity = (uint32_t)capacity;
    arch.reservedCapacity = (uint32_t)reserved;
    const bool growable = reserved > capacity;

    // column 0 = Entity, column 1 = WorldID, then user components
    // (reference src/mw/device/state.cpp:269-341)
    arch.colBytes.push_back((uint32_t)sizeof(Entity));
    arch.colComponent.push_back(0);
    arch.colBytes.push_back((uint32_t)sizeof(WorldID));
    arch.colComponent.push_back(1);
    for (uint32_t c : arch.comps) {
        arch.colBytes.push_back(exec->components[c].bytes);
        arch.colComponent.push_back(c);
    }

    arch.rowBytes = 0;
    arch.primary.resize(arch.numColumns);
    arch.alt.resize(arch.numColumns);
    arch.colFlags.assign(arch.numColumns, 0u);
    arch.primaryVm.assign(arch.numColumns, nullptr);
    arch.altVm.assign(arch.numColumns, nullptr);
    for (uint32_t c = 0; c < arch.numColumns; c++) {
        arch.rowBytes += arch.colBytes[c];
        size_t bytes = (size_t)arch.capacity * arch.colBytes[c] + 16;
        if (growable) {
            size_t max_bytes =
                (size_t)arch.reservedCapacity * arch.colBytes[c] + 16;
            rc = vmAlloc(exec, &arch.primary[c], &arch.primaryVm[c], max_bytes,
                         bytes, true);
            if (rc != 0) return rc;
            rc = vmAlloc(exec, &arch.alt[c], &arch.altVm[c], max_bytes, bytes,
                         true);
            if (rc != 0) return rc;
            continue;
        }
        rc = devAlloc(exec, &arch.primary[c], bytes);
        if (rc != 0) return rc;
        rc = devAlloc(exec, &arch.alt[c], bytes);
        if (rc != 0) return rc;
    }

    rc = devAllocT(exec, &arch.worldOffsets, W);
    if (rc != 0) return rc;
    rc = devAllocT(exec, &arch.worldCounts, W);
    if (rc != 0) return rc;

    arch.registered = true;
    return 0;
}

extern "C" int mwhip_register_singleton(mwhip_exec *exec, uint32_t archetype_id,
                                        uint32_t component_id)
{
    (void)component_id;
    if (archetype_id >= exec->archetypes.size() ||
            !exec->archetypes[archetype_id].registered) {
        return fail(-3, "singleton archetype %u not registered", archetype_id);
    }
    ArchetypeRec &arch = exec->archetypes[archetype_id];
    if (!arch.singleton) {
        arch.singleton = true;
        arch.singletonOrdinal = (int32_t)exec->numSingletons++;
    }
    return 0;
}

MWHIP_RT int findColumn(const ArchetypeRec &arch, uint32_t component_id)
{
    if (component_id == 0) return 0;
    if (component_id == 1) return 1;
    for (size_t i = 0; i < arch.comps.size(); i++) {
        if (arch.comps[i] == component_id) {
            return (int)i + 2;
        }
    }
    return -1;
}

extern "C" void *mwhip_export_column(mwhip_exec *exec, uint32_t archetype_id,
                                     uint32_t component_id, int32_t slot)
{
    if (archetype_id >= exec->archetypes.size() ||
            !exec->archetypes[archetype_id].registered) {
        fail(-3, "exportColumn: archetype %u not registered", archetype_id);
        return nullptr;
    }
    ArchetypeRec &arch = exec->archetypes[archetype_id];
    int col = findColumn(arch, component_id);
    if (col < 0) {
        fail(-3, "exportColumn: archetype %u has no component %u",
             archetype_id, component_id);
        return nullptr;
    }
    if (slot < 0 || (uint32_t)slot >= exec->exported.size()) {
        fail(-3, "exportColumn: slot %d out of range (numExportedBuffers=%zu)",
             slot, exec->exported.size());
        return nullptr;
    }

    // exported columns keep their address across sorts
    arch.colFlags[col] |= kColumnPinned;
    exec->exported[slot] = arch.primary[col];
    return arch.primary[col];
}

extern "C" int mwhip_make_query(mwhip_exec *exec, const uint32_t *component_ids,
                                uint32_t num_components, uint32_t *offset_out,
                                uint32_t *num_matching_out, uint32_t *flags_out)
{
    std::vector<uint32_t> comps(component_ids, component_ids + num_components);
    for (const QueryRec &q : exec->queries) {
        if (q.comps == comps) {
            *offset_out = q.offset;
            *num_matching_out = q.numMatching;
            if (flags_out) *flags_out = q.flags;
            return 0;
        }
    }

    // same record layout and archetype order as the reference's makeQuery
    // (src/mw/device/state.cpp:380-440)
    QueryRec rec;
    rec.comps = comps;
    rec.offset = (uint32_t)exec->queryDataHost.size();
    rec.numMatching = 0;
    bool all_singleton = true;

    for (uint32_t a = 0; a < exec->archetypes.size(); a++) {
        const ArchetypeRec &arch = exec->archetypes[a];
        if (!arch.registered) continue;

        bool has_all = true;
        for (uint32_t c : comps) {
            if (c == 0) continue;   // Entity is in every table
            if (findColumn(arch, c) < 0) {
                has_all = false;
                break;
            }
        }
        if (!has_all) continue;

        rec.numMatching += 1;
        all_singleton = all_singleton && arch.singleton;
        exec->queryDataHost.push_back(a);
        for (uint32_t c : comps) {
            exec->queryDataHost.push_back((uint32_t)findColumn(arch, c));
        }
    }

    if (exec->queryDataHost.size() > exec->queryCapacity) {
        return fail(-2, "query table overflow");
    }

    if (exec->stateBuilt && exec->queryDataHost.size() > rec.offset) {
        HIPCHK(hipMemcpy(exec->hostState.queryData + rec.offset,
            exec->queryDataHost.data() + rec.offset,
            (exec->queryDataHost.size() - rec.offset) * sizeof(uint32_t),
            hipMemcpyHostToDevice));
    }

    rec.flags = (rec.numMatching > 0 && all_singleton) ?
        MWHIP_QUERY_ALL_SINGLETON : 0u;
    *offset_out = rec.offset;
    *num_matching_out = rec.numMatching;
    if (flags_out) *flags_out = rec.flags;
    exec->queries.push_back(std::move(rec));
    return 0;
}

extern "C" void *mwhip_raw_alloc(int gpu_id, uint64_t num_bytes)
{
    void *ptr = nullptr;
    if (hipSetDevice(gpu_id) != hipSuccess ||
            hipMalloc(&ptr, num_bytes == 0 ? 16 : num_bytes) != hipSuccess) {
        fail(-2, "raw_alloc of %llu bytes on gpu %d failed",
             (unsigned long long)num_bytes, gpu_id);
        return nullptr;
    }
    return ptr;
}

extern "C" void mwhip_raw_free(int gpu_id, void *device_ptr)
{
    if (device_ptr != nullptr && hipSetDevice(gpu_id) == hipSuccess) {
        (void)hipFree(device_ptr);
    }
}

extern "C" int mwhip_raw_copy_h2d(int gpu_id, void *dst_device,
                                  const void *src_host, uint64_t num_bytes)
{
    HIPCHK(hipSetDevice(gpu_id));
    if (num_bytes != 0) {
        HIPCHK(hipMemcpy(dst_device, src_host, num_bytes,
                         hipMemcpyHostToDevice));
    }
    return 0;
}

extern "C" int mwhip_raw_copy_d2h(int gpu_id, void *dst_host,
                                  const void *src_device, uint64_t num_bytes)
{
    HIPCHK(hipSetDevice(gpu_id));
    if (num_bytes != 0) {
        HIPCHK(hipMemcpy(dst_host, src_device, num_bytes,
                         hipMemcpyDeviceToHost));
    }
    return 0;
}

extern "C" void *mwhip_alloc_device(mwhip_exec *exec, uint64_t num_bytes, int zero)
{
    void *ptr = nullptr;
    if (devAlloc(exec, &ptr, num_bytes, zero != 0) != 0) {
        return nullptr;
    }
    return ptr;
}

extern "C" int mwhip_set_module_data(mwhip_exec *exec, uint32_t slot,
                                     void *device_ptr)
{
    if (slot >= 4) {
        return fail(-3, "module data slot %u out of range", slot);
    }
    exec->hostState.moduleData[slot] = device_ptr;
    if (exec->stateBuilt) {
        HIPCHK(hipMemcpy((char *)exec->stateDev +
            offsetof(EcsState, moduleData) + slot * sizeof(void *), &device_ptr,
            sizeof(void *), hipMemcpyHostToDevice));
    }
    return 0;
}

extern "C" void *mwhip_get_module_data(mwhip_exec *exec, uint32_t slot)
{
    return slot < 4 ? exec->hostState.moduleData[slot] : nullptr;
}

extern "C" uint32_t mwhip_archetype_capacity(mwhip_exec *exec,
                                             uint32_t archetype_id)
{
    if (archetype_id >= exec->archetypes.size() ||
            !exec->archetypes[archetype_id].registered) {
        return 0;
    }
    return exec->archetypes[archetype_id].reservedCapacity;
}

extern "C" void *mwhip_table_header(mwhip_exec *exec, uint32_t archetype_id)
{
    if (!exec->stateBuilt || archetype_id >= exec->archetypes.size()) {
        return nullptr;
    }
    return exec->hostState.tables + archetype_id;
}

extern "C" int mwhip_get_query_data(mwhip_exec *exec, uint32_t offset,
                                    uint32_t count, uint32_t *out)
{
    if ((size_t)offset + count > exec->queryDataHost.size()) {
        return fail(-3, "query data range out of bounds");
    }
    memcpy(out, exec->queryDataHost.data() + offset, count * sizeof(uint32_t));
    return 0;
}

extern "C" void *mwhip_device_state(mwhip_exec *exec) { return exec->stateDev; }

extern "C" void *mwhip_world_data(mwhip_exec *exec, uint32_t world_idx)
{
    return exec->hostState.worldData +
        (uint64_t)world_idx * exec->hostState.worldDataStride;
}

extern "C" uint32_t mwhip_num_worlds(const mwhip_exec *exec)
{
    return exec->cfg.num_worlds;
}

extern "C" void mwhip_render_config(const mwhip_exec *exec,
                                    uint32_t *resolution_out,
                                    uint32_t *rgbd_out)
{
    *resolution_out = exec->cfg.raycast_output_resolution;
    *rgbd_out = exec->cfg.raycast_rgbd;
}

extern "C" int mwhip_render_geometry_info(const mwhip_render_geometry *geometry,
                                          uint32_t *num_nodes_out,
                                          uint32_t *is_box_out, float *bounds_out)
{
    if (geometry == nullptr) {
        return fail(-1, "render_geometry_info: null geometry");
    }
    RenderGeometryHost built;
    std::string error;
    if (buildRenderGeometry(*geometry, built, error) != 0) {
        return fail(-1, "%s", error.c_str());
    }
    for (uint32_t obj = 0; obj < built.numObjects; obj++) {
        if (num_nodes_out != nullptr) {
            num_nodes_out[obj] =
                built.objectNodeOffset[obj + 1] - built.objectNodeOffset[obj];
        }
        if (is_box_out != nullptr) {
            is_box_out[obj] = built.objectIsBox[obj];
        }
    }
    if (bounds_out != nullptr) {
        memcpy(bounds_out, built.objectRootBox.data(),
               built.objectRootBox.size() * sizeof(float));
    }
    return 0;
}

extern "C" uint32_t mwhip_render_max_views(const mwhip_exec *exec)
{
    return exec->cfg.raycast_max_views_per_world;
}

extern "C" uint32_t mwhip_num_task_graphs(const mwhip_exec *exec)
{
    return exec->cfg.num_task_graphs;
}

extern "C" void *mwhip_stream(mwhip_exec *exec) { return exec->stream; }

extern "C" void *mwhip_get_exported(const mwhip_exec *exec, uint32_t slot)
{
    return slot < exec->exported.size() ? exec->exported[slot] : nullptr;
}

extern "C" int mwhip_create(const mwhip_state_config *cfg,
                            const mwhip_user_entry *entry, mwhip_exec **out)
{
    *out = nullptr;
    if (entry == nullptr || entry->abi_version != MWHIP_ABI_VERSION) {
        return fail(-1, "user entry ABI mismatch");
    }
    if (cfg->num_worlds == 0) {
        return fail(-1, "numWorlds must be > 0");
    }

    int device_count = 0;
    hipError_t res = hipGetDeviceCount(&device_count);
    if (res != hipSuccess || device_count == 0) {
        return fail(-11, "no HIP device available (the MI355X backend has no "
                    "CPU fallback)");
    }
    HIPCHK(hipSetDevice(cfg->gpu_id));

    g_lastError.clear();
    std::unique_ptr<mwhip_exec> exec(new mwhip_exec {});
    exec->cfg = *cfg;
    exec->entry = *entry;
    exec->exported.assign(cfg->num_exported_buffers, nullptr);
    exec->taskGraphs.resize(cfg->num_task_graphs);
    exec->checkAfterRun = envU32("MADRONA_MWHIP_CHECK", 1) != 0;
    exec->sortCarriesMisc = envU32("MADRONA_MWHIP_SORT_CARRIES_MISC", 1) != 0;
    exec->sortCompaction = envU32("MADRONA_MWHIP_SORT_COMPACT", 1);
    exec->eagerReplay = envU32("MADRONA_MWHIP_EAGER", 0) != 0;
    {
        hipDeviceProp_t prop {};
        HIPCHK(hipGetDeviceProperties(&prop, cfg->gpu_id));
        exec->numCUs = std::max(prop.multiProcessorCount, 1);
        int rc_cfg = loadExecConfigFile(exec.get());
        if (rc_cfg != 0) return rc_cfg;
    }
    exec->tableGrowth = std::max(envU32("MADRONA_MWHIP_TABLE_GROWTH", 4), 1u);
    HIPCHK(hipStreamCreateWithFlags(&exec->stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&exec->serviceStream, hipStreamNonBlocking));

    if (cfg->render_geometry != nullptr) {
        std::string error;
        if (buildRenderGeometry(*cfg->render_geometry, exec->renderGeometry,
                                error) != 0) {
            return fail(-2, "%s", error.c_str());
        }
        exec->haveRenderGeometry = true;
    }

    // ---- registerTypes (host) -------------------------------------------------
    exec->registrationOpen = true;
    entry->register_types(exec.get(), cfg->user_config_ptr);
    exec->registrationOpen = false;
    int rc = buildDeviceState(exec.get());
    if (rc != 0) return rc;
    entry->bind_device_state(exec->stateDev);

    // ---- world constructors (device) -----------------------------------------
    rc = devAlloc(exec.get(), &exec->userCfgDev,
                  std::max<uint32_t>(cfg->num_user_config_bytes, 16u));
    if (rc != 0) return rc;
    if (cfg->num_user_config_bytes > 0) {
        HIPCHK(hipMemcpy(exec->userCfgDev, cfg->user_config_ptr,
            cfg->num_user_config_bytes, hipMemcpyHostToDevice));
    }
    size_t init_bytes = (size_t)cfg->num_world_init_bytes * cfg->num_worlds;
    rc = devAlloc(exec.get(), &exec->worldInitsDev,
                  std::max<size_t>(init_bytes, 16));
    if (rc != 0) return rc;
    if (init_bytes > 0 && cfg->world_init_ptr != nullptr) {
        HIPCHK(hipMemcpy(exec->worldInitsDev, cfg->world_init_ptr, init_bytes,
                         hipMemcpyHostToDevice));
    }

    rc = constructWorlds(exec.get());
    if (rc != 0) return rc;
    rc = sortAllArchetypes(exec.get());
    if (rc != 0) return rc;

    // ---- setupTasks (host) ----------------------------------------------------
    g_lastError.clear();
    entry->setup_tasks(exec.get(), cfg->user_config_ptr);

    // messages of the world constructors; from here on a thread keeps the ring
    // drained while replays are queued without being waited for
    drainHostPrints(exec.get(), false);
    mwhip_exec *raw = exec.release();
    __atomic_store_n(&raw->growMailbox->serviceEnabled, 1u, __ATOMIC_RELEASE);
    raw->printThread = std::thread([raw]() {
        // table-memory requests are answered promptly (device threads wait for
        // them), messages are printed at leisure
        uint32_t tick = 0;
        while (!raw->printStop.load()) {
            serviceGrowRequests(raw);
            if (tick++ % 16u == 0u) {
                drainHostPrints(raw, true);
            }
            std::this_thread::sleep_for(std::chrono::microseconds(150));
        }
    });

    *out = raw;
    return 0;
}

extern "C" void mwhip_destroy(mwhip_exec *exec)
{
    if (exec == nullptr) return;
    (void)hipSetDevice(exec->cfg.gpu_id);
    (void)hipStreamSynchronize(exec->stream);
    // The service thread (host prints + on-demand growth) goes first: it maps
    // memory into the reserved ranges and walks exec->vmRanges / archetypes,
    // all of which are released below.
    if (exec->growMailbox != nullptr) {
        __atomic_store_n(&exec->growMailbox->serviceEnabled, 0u, __ATOMIC_RELEASE);
    }
    exec->printStop.store(true);
    if (exec->printThread.joinable()) exec->printThread.join();
    (void)hipStreamSynchronize(exec->serviceStream);
    drainHostPrints(exec, false);
#ifdef MADRONA_TRACING
    writeDeviceTrace(exec);
#endif
    for (auto &kv : exec->launchGraphs) {
        releaseLaunchGraph(*kv.second);
    }
    for (void *p : exec->allocations) {
        (void)hipFree(p);
    }
    vmFreeAll(exec);
    if (exec->printRing) (void)hipHostFree(exec->printRing);
    if (exec->growMailbox) (void)hipHostFree(exec->growMailbox);
    if (exec->statsHost) (void)hipHostFree(exec->statsHost);
    (void)hipStreamDestroy(exec->stream);
    (void)hipStreamDestroy(exec->serviceStream);
    delete exec;
}

// ---------------------------------------------------------------------------
// task graph
// ---------------------------------------------------------------------------
extern "C" int32_t mwhip_tg_add_node_data(mwhip_exec *exec, uint32_t tg_id,
                                          const void *data, uint32_t num_bytes)
{
    if (tg_id >= exec->taskGraphs.size()) {
        return fail(-3, "task graph %u out of range (numTaskGraphs=%zu)", tg_id,
                    exec->taskGraphs.size());
    }
    if (num_bytes > MWHIP_MAX_NODE_DATA_BYTES) {
        return fail(-2, "node data larger than %u bytes",
                    (unsigned)MWHIP_MAX_NODE_DATA_BYTES);
    }
    void *dev = nullptr;
    int rc = devAlloc(exec, &dev, std::max<uint32_t>(num_bytes, 256u));
    if (rc != 0) return rc;
    if (num_bytes > 0) {
        HIPCHK(hipMemcpy(dev, data, num_bytes, hipMemcpyHostToDevice));
    }
    TaskGraphRec &tg = exec->taskGraphs[tg_id];
    tg.dataDev.push_back(dev);
    return (int32_t)tg.dataDev.size() - 1;
}

extern "C" void *mwhip_tg_node_data(mwhip_exec *exec, uint32_t tg_id,
                                    int32_t data_id)
{
    if (tg_id >= exec->taskGraphs.size()) return nullptr;
    TaskGraphRec &tg = exec->taskGraphs[tg_id];
    if (data_id < 0 || (size_t)data_id >= tg.dataDev.size()) return nullptr;
    return tg.dataDev[data_id];
}

extern "C" const void *mwhip_pfor_body(mwhip_exec *exec, const void *kernel)
{
    if (kernel == nullptr) return nullptr;
    if (exec->pforBodyScratch == nullptr) {
        if (devAlloc(exec, &exec->pforBodyScratch, 16) != 0) return nullptr;
    }
    if (hipMemsetAsync(exec->pforBodyScratch, 0, 16, exec->stream) != hipSuccess) {
        return nullptr;
    }
    EcsState *state = exec->stateDev;
    void *report_to = exec->pforBodyScratch;
    uint32_t query_offset = 0, report_mode = 0xFFFFFFFFu;
    mwhip_pfor_args none {};
    void *args[] = { &state, &report_to, &query_offset, &report_mode, &none };
    if (hipLaunchKernel(kernel, dim3(1, 1, 1), dim3(64, 1, 1), args, 0,
                        exec->stream) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    const void *body = nullptr;
    if (hipStreamSynchronize(exec->stream) != hipSuccess ||
            hipMemcpy(&body, exec->pforBodyScratch, sizeof(body),
                      hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return body;
}

extern "C" int mwhip_set_pfor_group_kernel(mwhip_exec *exec, const void *kernel)
{
    exec->pforGroupKernel = kernel;
    return 0;
}

extern "C" int32_t mwhip_tg_add_node(mwhip_exec *exec, uint32_t tg_id,
                                     const mwhip_node_desc *desc,
                                     const int32_t *deps, uint32_t num_deps)
{
    if (tg_id >= exec->taskGraphs.size()) {
        return fail(-3, "task graph %u out of range (numTaskGraphs=%zu)", tg_id,
                    exec->taskGraphs.size());
    }
    TaskGraphRec &tg = exec->taskGraphs[tg_id];

    NodeRec node;
    node.desc = *desc;
    node.name = desc->name != nullptr ? desc->name : "node";
    node.desc.name = nullptr;
    for (uint32_t i = 0; i < num_deps; i++) {
        if (deps[i] < 0 || (size_t)deps[i] >= tg.nodes.size()) {
            return fail(-3, "node '%s' depends on unknown node %d",
                        node.name.c_str(), deps[i]);
        }
        node.deps.push_back(deps[i]);
    }
    if (desc->kind == MWHIP_NODE_KERNEL && desc->kernel == nullptr) {
        return fail(-3, "node '%s' has no kernel", node.name.c_str());
    }

    tg.nodes.push_back(std::move(node));
    tg.built = false;
    return (int32_t)tg.nodes.size() - 1;
}

// Same ordering rule as the reference builder (taskgraph_utils.cpp:74-146,
// identical on CPU: src/core/taskgraph.cpp:53-117): repeatedly take the first
// not-yet-queued node; queue it if all of its dependencies are queued.
static int topoSort(TaskGraphRec &tg)
{
    const size_t n = tg.nodes.size();
    tg.sorted.clear();
    if (n == 0) {
        tg.built = true;
        return 0;
    }

    std::vector<bool> queued(n, false);
    size_t remaining = n;
    size_t guard = 0;
    while (remaining > 0) {
        bool progressed = false;
        for (size_t i = 0; i < n; i++) {
            if (queued[i]) continue;
            bool ready = true;
            for (int32_t dep : tg.nodes[i].deps) {
                if (!queued[dep]) {
                    ready = false;
                    break;
                }
            }
            if (ready) {
                queued[i] = true;
                tg.sorted.push_back((int32_t)i);
                remaining--;
                progressed = true;
                break;
            }
        }
        if (!progressed || ++guard > n * n + 1) {
            return fail(-3, "task graph has a dependency cycle");
        }
    }
    tg.built = true;
    return 0;
}

extern "C" int mwhip_build_launch_graph(mwhip_exec *exec,
                                        const uint32_t *taskgraph_ids,
                                        uint32_t num_taskgraphs,
                                        const char *stat_name,
                                        uint64_t *graph_out)
{
    HIPCHK(hipSetDevice(exec->cfg.gpu_id));

    std::vector<uint32_t> ids(taskgraph_ids, taskgraph_ids + num_taskgraphs);
    for (uint32_t id : ids) {
        if (id >= exec->taskGraphs.size()) {
            return fail(-3, "task graph %u does not exist", id);
        }
        if (!exec->taskGraphs[id].built) {
            int rc = topoSort(exec->taskGraphs[id]);
            if (rc != 0) return rc;
        }
    }

    // the worlds have been constructed: size the tables for what they hold
    {
        int rc = growTablesFromDevice(exec);
        if (rc != 0) return rc;
    }

    std::unique_ptr<LaunchGraph> lg;
    int rc = instantiateLaunchGraph(exec, ids,
                                    stat_name != nullptr ? stat_name : "", lg);
    if (rc != 0) return rc;

    uint64_t handle = exec->nextGraphHandle++;
    exec->launchGraphs[handle] = std::move(lg);
    *graph_out = handle;
    return 0;
}

extern "C" int mwhip_set_render_layout(mwhip_exec *exec,
                                       const mwhip_render_layout *layout)
{
    if (layout == nullptr) {
        return fail(-2, "set_render_layout: null layout");
    }
    exec->renderLayout = *layout;
    exec->haveRenderLayout = true;
    return 0;
}

extern "C" int mwhip_build_render_graph(mwhip_exec *exec, uint64_t *graph_out)
{
    HIPCHK(hipSetDevice(exec->cfg.gpu_id));
    if (exec->cfg.raycast_output_resolution == 0) {
        return fail(-3, "buildRenderGraph: the executor was created without a "
                    "render configuration");
    }
    if (!exec->haveRenderGeometry) {
        return fail(-3, "buildRenderGraph: no geometry "
                    "(mwhip_state_config::render_geometry)");
    }
    if (!exec->haveRenderLayout) {
        return fail(-3, "buildRenderGraph: RenderingSystem::registerTypes did "
                    "not run (mwhip_set_render_layout)");
    }

    int rc = growTablesFromDevice(exec);
    if (rc != 0) return rc;

    LaunchGraph like {};
    like.isRender = true;
    std::unique_ptr<LaunchGraph> lg;
    rc = instantiateLaunchGraph(exec, {}, "render", lg, &like);
    if (rc != 0) return rc;

    uint64_t handle = exec->nextGraphHandle++;
    exec->launchGraphs[handle] = std::move(lg);
    *graph_out = handle;
    return 0;
}

extern "C" void mwhip_free_launch_graph(mwhip_exec *exec, uint64_t graph)
{
    auto it = exec->launchGraphs.find(graph);
    if (it == exec->launchGraphs.end()) return;
    (void)hipStreamSynchronize(exec->stream);
    releaseLaunchGraph(*it->second);
    exec->launchGraphs.erase(it);
}

static int checkHealth(mwhip_exec *exec)
{
    if (!exec->checkAfterRun) return 0;
    uint32_t flags = (uint32_t)exec->statsHost[0];
    if (flags != 0) {
        if ((flags & kErrTableOverflow) != 0u && exec->growMailbox != nullptr &&
                exec->growMailbox->failedRow != 0) {
            const GrowMailbox &mb = *exec->growMailbox;
            const ArchetypeRec *arch =
                (size_t)mb.failedArchetype < exec->archetypes.size() ?
                    &exec->archetypes[mb.failedArchetype] : nullptr;
            return fail(-5, "device error 0x%x: %s [archetype %d: row %d with %d "
                        "rows mapped, %u reserved]", flags, describeError(flags),
                        mb.failedArchetype, mb.failedRow, mb.failedCapacity,
                        arch != nullptr ? arch->reservedCapacity : 0u);
        }
        return fail(-5, "device error 0x%x: %s", flags, describeError(flags));
    }
    return 0;
}

static int replayGraph(mwhip_exec *exec, LaunchGraph &lg, hipStream_t stream);

extern "C" int mwhip_run(mwhip_exec *exec, uint64_t graph)
{
    auto it = exec->launchGraphs.find(graph);
    if (it == exec->launchGraphs.end()) {
        return fail(-3, "unknown launch graph");
    }
    {
        int launch_rc = replayGraph(exec, *it->second, exec->stream);
        if (launch_rc != 0) return launch_rc;
    }
    exec->replaysLaunched++;
    HIPCHK(hipStreamSynchronize(exec->stream));
    drainHostPrints(exec, false);
    int rc = checkHealth(exec);
    if (rc != 0) return rc;
#ifdef MADRONA_TRACING
    rc = collectDeviceTrace(exec);
    if (rc != 0) return rc;
#endif
    return growTablesAfterReplay(exec);
}

// One replay of a step graph on `stream`: the instantiated hipGraph, or -- 
// MADRONA_MWHIP_EAGER=1, measurement -- the same launches one by one (a graph
// kernel node costs ~4 us on this runtime whatever the kernel does; a
// dependent launch on a stream 1.5 us of device time and 3-4 us of host time,
// which a step of a millisecond hides).
static int replayGraph(mwhip_exec *exec, LaunchGraph &lg, hipStream_t stream)
{
    if (!exec->eagerReplay) {
        HIPCHK(hipGraphLaunch(lg.graphExec, stream));
        return 0;
    }
    for (KernelLaunch &k : lg.launches) {
        int rc = launchOne(exec, k, stream);
        if (rc != 0) return rc;
    }
    return 0;
}

extern "C" int mwhip_run_async(mwhip_exec *exec, uint64_t graph, void *hip_stream)
{
#ifdef MADRONA_TRACING
    // every step's log is read back before the next one overwrites it
    if ((hipStream_t)hip_stream == exec->stream) {
        return mwhip_run(exec, graph);
    }
#endif
    auto it = exec->launchGraphs.find(graph);
    if (it == exec->launchGraphs.end()) {
        return fail(-3, "unknown launch graph");
    }
    // health (and table sizes) as of the last completed replay
    int rc = checkHealth(exec);
    if (rc != 0) return rc;

    // Growth reacts to what COMPLETED replays reported: replays queued behind
    // them run with the tables as they are.  While some table is filling up
    // (its per-step peak above a quarter of what is mapped) at most two
    // replays stay in flight, so that growth keeps ahead of it; otherwise the
    // queue may run as deep as the caller likes.
    if ((hipStream_t)hip_stream == exec->stream) {
        const uint32_t done = __atomic_load_n(
            (uint32_t *)&exec->statsHost[kStatsReplays], __ATOMIC_ACQUIRE);
        const uint32_t in_flight = exec->replaysLaunched - done;
        // (a table is "filling up" while its peak keeps setting records: a
        // population in steady state -- resets that destroy and re-create the
        // same number of rows -- does not throttle the queue)
        bool filling = false;
        {
            // the service thread changes arch.capacity under this lock (never
            // held across a stream wait: a replay in flight may be waiting for
            // that thread)
            std::lock_guard<std::mutex> capacities(exec->growMutex);
            for (uint32_t a = 0; a < exec->archetypes.size(); a++) {
                ArchetypeRec &arch = exec->archetypes[a];
                if (!arch.registered || arch.reservedCapacity <= arch.capacity) {
                    continue;
                }
                const int64_t peak = exec->statsHost[kStatsPeaks + a];
                if (peak > arch.peakSeen) {
                    arch.peakSeen = peak;
                    if (4 * peak > (int64_t)arch.capacity) {
                        arch.fillingUntil = exec->replaysLaunched + 4u;
                    }
                }
                if (exec->replaysLaunched < arch.fillingUntil) {
                    filling = true;
                }
            }
        }
        static const bool limit_queue =
            envU32("MADRONA_MWHIP_LIMIT_QUEUE", 1) != 0;
        if (in_flight >= 2u && limit_queue) {
            if (filling) {
                HIPCHK(hipStreamSynchronize(exec->stream));
                drainHostPrints(exec, false);
                rc = checkHealth(exec);
                if (rc != 0) return rc;
            }
        }
    }
    // (only on the executor's own stream: growing waits for that stream to
    // drain; replays queued on a caller's stream grow at mwhip_synchronize)
    if ((hipStream_t)hip_stream == exec->stream) {
        rc = growTablesAfterReplay(exec);
        if (rc != 0) return rc;
    }
    // (growing rebuilds the graphs: look the handle up again)
    it = exec->launchGraphs.find(graph);
    rc = replayGraph(exec, *it->second, (hipStream_t)hip_stream);
    if (rc != 0) return rc;
    exec->replaysLaunched++;
    return 0;
}

static int makePackArgs(uint32_t num_columns, const void *const *src_columns,
                        const uint32_t *words_per_row, uint32_t num_rows,
                        PackArgs *out)
{
    if (num_columns == 0 || num_columns > MWHIP_PACK_MAX_COLUMNS) {
        return fail(-2, "pack_rows: %u columns (1..%u)", num_columns,
                    (uint32_t)MWHIP_PACK_MAX_COLUMNS);
    }
    PackArgs args {};
    args.numColumns = num_columns;
    args.numRows = num_rows;
    for (uint32_t c = 0; c < num_columns; c++) {
        args.src[c] = (const uint32_t *)src_columns[c];
        args.words[c] = words_per_row[c];
        args.firstWord[c] = args.recordWords;
        args.recordWords += words_per_row[c];
    }
    *out = args;
    return 0;
}

extern "C" int mwhip_build_launch_graph_with_pack(
    mwhip_exec *exec, uint64_t base_graph, uint32_t num_columns,
    const void *const *src_columns, const uint32_t *words_per_row,
    uint32_t num_rows, void *dst, uint64_t *graph_out)
{
    auto it = exec->launchGraphs.find(base_graph);
    if (it == exec->launchGraphs.end()) {
        return fail(-3, "unknown launch graph");
    }
    HIPCHK(hipSetDevice(exec->cfg.gpu_id));

    LaunchGraph with_pack {};
    with_pack.hasPack = true;
    with_pack.packDst = dst;
    int rc = makePackArgs(num_columns, src_columns, words_per_row, num_rows,
                          &with_pack.pack);
    if (rc != 0) return rc;

    HIPCHK(hipStreamSynchronize(exec->stream));
    std::unique_ptr<LaunchGraph> lg;
    rc = instantiateLaunchGraph(exec, it->second->taskGraphIds,
                                it->second->statName, lg, &with_pack);
    if (rc != 0) return rc;

    uint64_t handle = exec->nextGraphHandle++;
    exec->launchGraphs[handle] = std::move(lg);
    *graph_out = handle;
    return 0;
}

static int rebuildAllLaunchGraphs(mwhip_exec *exec)
{
    HIPCHK(hipStreamSynchronize(exec->stream));
    for (auto &kv : exec->launchGraphs) {
        std::unique_ptr<LaunchGraph> fresh;
        int rc = instantiateLaunchGraph(exec, kv.second->taskGraphIds,
                                        kv.second->statName, fresh,
                                        kv.second.get());
        if (rc != 0) return rc;
        releaseLaunchGraph(*kv.second);
        kv.second = std::move(fresh);
    }
    return 0;
}

extern "C" uint32_t mwhip_device_cus(const mwhip_exec *exec)
{
    return exec->numCUs;
}

extern "C" int mwhip_set_input_ring(mwhip_exec *exec, void *dst, const void *ring,
                                    uint64_t slot_bytes, uint32_t num_slots)
{
    if (dst == nullptr) {
        return fail(-2, "set_input_ring: no destination");
    }
    HIPCHK(hipSetDevice(exec->cfg.gpu_id));
    auto &rings = exec->inputRings;
    rings.erase(std::remove_if(rings.begin(), rings.end(),
        [dst](const mwhip_exec::InputRing &r) { return r.dst == dst; }), rings.end());
    if (ring != nullptr) {
        if (num_slots == 0 || slot_bytes == 0 || slot_bytes % 4 != 0 ||
                slot_bytes / 4 > 0xFFFFFFFFull) {
            return fail(-2, "set_input_ring: %llu bytes x %u slots (whole dwords, "
                        "at least one slot)", (unsigned long long)slot_bytes, num_slots);
        }
        if (rings.size() >= 4) {
            return fail(-2, "set_input_ring: at most 4 rings");
        }
        // (every replay of every STEP graph of the executor counts; render
        // graphs do not)
        HIPCHK(hipStreamSynchronize(exec->stream));
        uint32_t done = 0;
        HIPCHK(hipMemcpy(&done, exec->replaySignal + kStepReplayWord, sizeof(done),
                         hipMemcpyDeviceToHost));
        rings.push_back({ (uint32_t *)dst, (const uint32_t *)ring,
                          (uint32_t)(slot_bytes / 4), num_slots, done });
    }
    return rebuildAllLaunchGraphs(exec);
}

// Another stream waits for every replay queued so far WITHOUT touching the
// executor's stream: the last kernel of each replay bumps a counter in signal
// memory and the waiting stream polls it (hipStreamWaitValue32).  An event
// recorded between two graph launches would do, but costs ~20 us of launch
// pipelining per step on this runtime; an event-record node inside the graph
// does not order a later hipStreamWaitEvent (measured: the wait returns early).
extern "C" int mwhip_stream_wait_replays(mwhip_exec *exec, void *hip_stream)
{
    if (exec->replaySignal == nullptr) {
        return fail(-3, "no replay signal");
    }
    HIPCHK(hipStreamWaitValue32((hipStream_t)hip_stream, exec->replaySignal,
                                exec->replaysLaunched, hipStreamWaitValueGte,
                                0xFFFFFFFFu));
    return 0;
}

extern "C" int mwhip_pack_rows(mwhip_exec *exec, uint32_t num_columns,
                               const void *const *src_columns,
                               const uint32_t *words_per_row, uint32_t num_rows,
                               void *dst)
{
    PackArgs args {};
    int rc = makePackArgs(num_columns, src_columns, words_per_row, num_rows,
                          &args);
    if (rc != 0) return rc;
    if (args.recordWords == 0 || num_rows == 0) {
        return 0;
    }

    const uint64_t total = (uint64_t)num_rows * args.recordWords;
    const uint32_t blocks =
        (uint32_t)std::min<uint64_t>((total + 255) / 256, 4096);
    HIPCHK(hipSetDevice(exec->cfg.gpu_id));
    uint32_t *dst_words = (uint32_t *)dst;
    void *pack_args[] = { &args, &dst_words };
    HIPCHK(hipLaunchKernel(packRowsKernelFn(), dim3(blocks), dim3(256), pack_args,
                           0, exec->stream));
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int mwhip_mark_window(mwhip_exec *exec, uint32_t id)
{
    HIPCHK(hipSetDevice(exec->cfg.gpu_id));
    uint32_t *no_signal = nullptr;
    void *mark_args[] = { &no_signal, &id };
    HIPCHK(hipLaunchKernel(benchWindowMarkerFn(), dim3(1), dim3(64), mark_args, 0,
                           exec->stream));
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" uint32_t mwhip_num_table_growths(mwhip_exec *exec)
{
    return exec->numGrowths;
}

extern "C" int mwhip_synchronize(mwhip_exec *exec)
{
    HIPCHK(hipStreamSynchronize(exec->stream));
    drainHostPrints(exec, false);
    int rc = checkHealth(exec);
    if (rc != 0) return rc;
#ifdef MADRONA_TRACING
    rc = collectDeviceTrace(exec);
    if (rc != 0) return rc;
#endif
    return growTablesAfterReplay(exec);
}

// ---------------------------------------------------------------------------
// introspection
// ---------------------------------------------------------------------------
extern "C" int32_t mwhip_num_rows(mwhip_exec *exec, uint32_t archetype_id)
{
    if (archetype_id >= exec->archetypes.size() ||
            !exec->archetypes[archetype_id].registered) {
        return -1;
    }
    TableHdr hdr;
    if (hipMemcpy(&hdr, exec->hostState.tables + archetype_id, sizeof(TableHdr),
                  hipMemcpyDeviceToHost) != hipSuccess) {
        return -1;
    }
    return hdr.numRows;
}

extern "C" int64_t mwhip_dump_column(mwhip_exec *exec, uint32_t archetype_id,
                                     uint32_t component_id, void *dst,
                                     uint64_t dst_bytes, int32_t *world_counts)
{
    if (archetype_id >= exec->archetypes.size() ||
            !exec->archetypes[archetype_id].registered) {
        fail(-3, "dump: archetype %u not registered", archetype_id);
        return -1;
    }
    (void)hipSetDevice(exec->cfg.gpu_id);
    (void)hipStreamSynchronize(exec->stream);

    const ArchetypeRec &arch = exec->archetypes[archetype_id];
    int col = findColumn(arch, component_id);
    if (col < 0) {
        fail(-3, "dump: archetype %u has no component %u", archetype_id,
             component_id);
        return -1;
    }

    // the live header knows which of the ping-pong buffers is current
    TableHdr hdr;
    if (hipMemcpy(&hdr, exec->hostState.tables + archetype_id, sizeof(TableHdr),
                  hipMemcpyDeviceToHost) != hipSuccess) {
        return -1;
    }

    const uint32_t W = exec->cfg.num_worlds;
    std::vector<int32_t> offsets(W), counts(W);
    if (hipMemcpy(offsets.data(), hdr.worldOffsets, W * sizeof(int32_t),
                  hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(counts.data(), hdr.worldCounts, W * sizeof(int32_t),
                  hipMemcpyDeviceToHost) != hipSuccess) {
        return -1;
    }

    const uint32_t elem = arch.colBytes[col];
    int64_t total = 0;
    bool contiguous = true;
    for (uint32_t w = 0; w < W; w++) {
        if (counts[w] < 0 || offsets[w] < 0) {
            fail(-3, "dump: table %u is not sorted by world", archetype_id);
            return -1;
        }
        if (counts[w] > 0 && offsets[w] != total) contiguous = false;
        total += counts[w];
        world_counts[w] = counts[w];
    }
    if ((uint64_t)total * elem > dst_bytes) {
        return -2;
    }

    if (contiguous) {
        if (total > 0 && hipMemcpy(dst, hdr.columns[col], (size_t)total * elem,
                hipMemcpyDeviceToHost) != hipSuccess) {
            return -1;
        }
    } else {
        char *out = (char *)dst;
        for (uint32_t w = 0; w < W; w++) {
            size_t nb = (size_t)counts[w] * elem;
            if (nb == 0) continue;
            if (hipMemcpy(out, (char *)hdr.columns[col] +
                    (size_t)offsets[w] * elem, nb,
                    hipMemcpyDeviceToHost) != hipSuccess) {
                return -1;
            }
            out += nb;
        }
    }
    return total;
}

extern "C" int64_t mwhip_dump_column_raw(mwhip_exec *exec,
                                         uint32_t archetype_id,
                                         uint32_t component_id, void *dst,
                                         uint64_t dst_bytes)
{
    if (archetype_id >= exec->archetypes.size() ||
            !exec->archetypes[archetype_id].registered) {
        fail(-3, "dump: archetype %u not registered", archetype_id);
        return -1;
    }
    (void)hipSetDevice(exec->cfg.gpu_id);
    (void)hipStreamSynchronize(exec->stream);

    const ArchetypeRec &arch = exec->archetypes[archetype_id];
    int col = findColumn(arch, component_id);
    if (col < 0) {
        fail(-3, "dump: archetype %u has no component %u", archetype_id,
             component_id);
        return -1;
    }
    TableHdr hdr;
    if (hipMemcpy(&hdr, exec->hostState.tables + archetype_id, sizeof(TableHdr),
                  hipMemcpyDeviceToHost) != hipSuccess) {
        return -1;
    }
    const uint64_t bytes = (uint64_t)hdr.numRows * arch.colBytes[col];
    if (bytes > dst_bytes) {
        return -2;
    }
    if (bytes > 0 && hipMemcpy(dst, hdr.columns[col], bytes,
                               hipMemcpyDeviceToHost) != hipSuccess) {
        return -1;
    }
    return hdr.numRows;
}

extern "C" int mwhip_memcpy_d2h(void *dst_host, const void *src_dev,
                                uint64_t num_bytes)
{
    HIPCHK(hipMemcpy(dst_host, src_dev, num_bytes, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int mwhip_memcpy_any(void *dst_host, const void *src,
                                uint64_t num_bytes)
{
    if (num_bytes == 0) return 0;
    hipPointerAttribute_t attr {};
    const hipError_t res = hipPointerGetAttributes(&attr, src);
    if (res != hipSuccess || attr.type == hipMemoryTypeUnregistered ||
            attr.type == hipMemoryTypeHost) {
        (void)hipGetLastError();    // (plain host memory is not an error)
        memcpy(dst_host, src, num_bytes);
        return 0;
    }
    HIPCHK(hipMemcpy(dst_host, src, num_bytes, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int mwhip_memcpy_h2d(void *dst_dev, const void *src_host,
                                uint64_t num_bytes)
{
    HIPCHK(hipMemcpy(dst_dev, src_host, num_bytes, hipMemcpyHostToDevice));
    return 0;
}

// ---------------------------------------------------------------------------
// per-kernel timing + algorithmic bytes
// ---------------------------------------------------------------------------
static int readRowCounts(mwhip_exec *exec, std::vector<int32_t> &rows)
{
    std::vector<TableHdr> hdrs(exec->tablesHost.size());
    HIPCHK(hipMemcpy(hdrs.data(), exec->hostState.tables,
        hdrs.size() * sizeof(TableHdr), hipMemcpyDeviceToHost));
    rows.resize(hdrs.size());
    for (size_t i = 0; i < hdrs.size(); i++) rows[i] = hdrs[i].numRows;
    return 0;
}

extern "C" int32_t mwhip_profile(mwhip_exec *exec, uint64_t graph, uint32_t reps,
                                 mwhip_kernel_stat *out, uint32_t max_out)
{
    auto it = exec->launchGraphs.find(graph);
    if (it == exec->launchGraphs.end()) {
        return fail(-3, "unknown launch graph");
    }
    LaunchGraph &lg = *it->second;
    const size_t n = lg.launches.size();
    if (reps == 0) reps = 1;

    HIPCHK(hipSetDevice(exec->cfg.gpu_id));
    HIPCHK(hipStreamSynchronize(exec->stream));

    // start/stop events attached to each dispatch: their difference is the
    // kernel's own begin/end timestamp pair (what rocprofv3 reports), without
    // the cost of separate event packets between kernels
    std::vector<hipEvent_t> ev_start(n), ev_stop(n);
    for (size_t i = 0; i < n; i++) {
        HIPCHK(hipEventCreate(&ev_start[i]));
        HIPCHK(hipEventCreate(&ev_stop[i]));
    }

    std::vector<double> total_us(n, 0.0), total_rows(n, 0.0),
        total_bytes(n, 0.0);

    // snapshot sort statistics to turn them into per-launch averages
    struct SiteSnap { SortState before, after; };
    std::vector<std::vector<SiteSnap>> snaps(lg.sortBatches.size());
    for (size_t b = 0; b < lg.sortBatches.size(); b++) {
        snaps[b].resize(lg.sortBatches[b]->sites.size());
        for (size_t s = 0; s < snaps[b].size(); s++) {
            HIPCHK(hipMemcpy(&snaps[b][s].before,
                lg.sortBatches[b]->sites[s].stateDev, sizeof(SortState),
                hipMemcpyDeviceToHost));
        }
    }

    for (uint32_t r = 0; r < reps; r++) {
        std::vector<int32_t> rows;
        int rc = readRowCounts(exec, rows);
        if (rc != 0) return rc;

        // queue everything behind the gate, then open it
        volatile int32_t *gate_host = exec->statsHost + kStatsGate;
        *gate_host = 0;
        __sync_synchronize();
        {
            int32_t *gate_dev = nullptr;
            HIPCHK(hipHostGetDevicePointer((void **)&gate_dev,
                (void *)(exec->statsHost + kStatsGate), 0));
            void *gargs[] = { &gate_dev };
            HIPCHK(hipLaunchKernel(gateKernelFn(), dim3(1), dim3(64),
                                   gargs, 0, exec->stream));
        }
        for (size_t i = 0; i < n; i++) {
            KernelLaunch &k = lg.launches[i];
            void *kargs[8];
            k.argPointers(kargs);
            hipError_t lres = hipExtLaunchKernel(k.fn, k.grid, k.block, kargs,
                k.dynamicLds, exec->stream, ev_start[i], ev_stop[i], 0);
            if (lres != hipSuccess) {
                *gate_host = 1;
                return fail(-10, "hipExtLaunchKernel -> %s",
                            hipGetErrorString(lres));
            }
        }
        *gate_host = 1;
        __sync_synchronize();
        HIPCHK(hipStreamSynchronize(exec->stream));
        // (the health kernel ran once more: keep the host's replay count in
        // step with the device's)
        HIPCHK(hipMemcpy(&exec->replaysLaunched, exec->replaySignal,
                         sizeof(uint32_t), hipMemcpyDeviceToHost));
        rc = checkHealth(exec);
        if (rc != 0) return rc;

        for (size_t i = 0; i < n; i++) {
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, ev_start[i], ev_stop[i]));
            total_us[i] += (double)ms * 1000.0;

            const KernelLaunch &k = lg.launches[i];
            if (k.kind == MWHIP_NODE_KERNEL &&
                    k.countMode == MWHIP_COUNT_QUERY_ROWS) {
                // rows at the start of the step (steady-state approximation)
                auto rowsOf = [&](uint32_t query_offset) {
                    double nrows = 0;
                    for (const QueryRec &q : exec->queries) {
                        if (q.offset != query_offset) continue;
                        const uint32_t *p =
                            exec->queryDataHost.data() + q.offset;
                        for (uint32_t m = 0; m < q.numMatching; m++) {
                            nrows += rows[p[0]];
                            p += 1 + q.comps.size();
                        }
                        break;
                    }
                    return nrows;
                };
                if (k.members.empty()) {
                    const double nrows = rowsOf(k.queryOffset);
                    total_rows[i] += nrows;
                    total_bytes[i] += nrows * k.bytesPerRow;
                } else {
                    // a grouped launch: the sum over its nodes
                    for (const KernelLaunch::Member &mem : k.members) {
                        const double nrows = rowsOf(mem.queryOffset);
                        total_rows[i] += nrows;
                        total_bytes[i] += nrows * mem.bytesPerRow;
                    }
                }
            }
        }
    }

    // sort kernels: bytes from the measured rows in / out of each site.  An
    // archetype sorted by several nodes of the graph (e.g. the contact table,
    // once per substep) has one SortState: split its counters evenly.
    auto shareCount = [&](const void *state_dev) {
        double count = 0;
        for (const auto &other : lg.sortBatches) {
            for (const auto &site : other->sites) {
                if ((const void *)site.stateDev == state_dev) count += 1;
            }
        }
        return count < 1 ? 1.0 : count;
    };
    for (size_t b = 0; b < lg.sortBatches.size(); b++) {
        const SortBatch &batch = *lg.sortBatches[b];
        double hist = 0, pass0 = 0, passn = 0, gather = 0, fin = 0, rows_in = 0;
        double passes_small = 0;
        for (size_t s = 0; s < batch.sites.size(); s++) {
            HIPCHK(hipMemcpy(&snaps[b][s].after, batch.sites[s].stateDev,
                sizeof(SortState), hipMemcpyDeviceToHost));
            const double share =
                (double)reps * shareCount(batch.sites[s].stateDev);
            double n_in = (double)(snaps[b][s].after.statRowsIn -
                                   snaps[b][s].before.statRowsIn) / share;
            double n_out = (double)(snaps[b][s].after.statRowsOut -
                                    snaps[b][s].before.statRowsOut) / share;
            double runs = (double)(snaps[b][s].after.statRuns -
                                   snaps[b][s].before.statRuns) / share;
            rows_in += n_in;
            hist += 4.0 * n_in;
            pass0 += 4.0 * n_in + 8.0 * n_in;
            passn += 16.0 * n_in;
            // index read + every column read and written once + Loc remap +
            // the world's offset / count pair (world sorts)
            gather += 4.0 * n_out + 2.0 * batch.sites[s].rowBytes * n_out +
                4.0 * n_out +
                (batch.sites[s].worldSort ?
                     8.0 * exec->cfg.num_worlds * runs : 0.0);
            passes_small += (batch.sites[s].numPasses - 1) * 16.0 * n_in;
        }
        uint32_t pass_idx = 0;
        for (size_t i = 0; i < n; i++) {
            const KernelLaunch &k = lg.launches[i];
            if (k.sortBatch != &batch) continue;
            double bytes = 0;
            switch (k.sortRole) {
            case SortRole::Histogram: bytes = hist; pass_idx = 0; break;
            case SortRole::Onesweep:
                bytes = pass_idx == 0 ? pass0 : passn;
                pass_idx++;
                break;
            case SortRole::Gather: bytes = gather; break;
            case SortRole::Finalize: bytes = fin; break;
            // The compaction chain stands in for histogram + key passes: the
            // node keeps SURVEY 8d's algorithmic bytes (what a sort node is
            // priced at), split over its two kernels the way the radix chain
            // splits them; what they move is 8 N + 8 N'.
            case SortRole::CompactPrepare: bytes = hist + pass0; break;
            case SortRole::CompactScatter: bytes = passes_small; break;
            case SortRole::Small:
                bytes = hist + pass0 + passes_small + gather;
                break;
            default: break;
            }
            total_bytes[i] = bytes * reps;
            total_rows[i] = rows_in * reps;
        }
    }

    for (size_t i = 0; i < n; i++) {
        (void)hipEventDestroy(ev_start[i]);
        (void)hipEventDestroy(ev_stop[i]);
    }

    lg.statNames.resize(n);
    uint32_t count = (uint32_t)std::min<size_t>(n, max_out);
    for (uint32_t i = 0; i < count; i++) {
        const KernelLaunch &k = lg.launches[i];
        lg.statNames[i] = k.name;
        if (k.role[0] != '\0') {
            lg.statNames[i] += ":";
            lg.statNames[i] += k.role;
        }
        out[i].name = lg.statNames[i].c_str();
        out[i].node_kind = k.kind;
        out[i].archetype_id = k.archetype;
        out[i].avg_us = total_us[i] / reps;
        out[i].algo_bytes = total_bytes[i] / reps;
        out[i].rows = total_rows[i] / reps;
        out[i].io_declared = lg.launches[i].ioDeclared;
        out[i].workgroups = lg.launches[i].grid.x * lg.launches[i].grid.y;
        out[i].node_index = lg.launches[i].nodeIndex;
        out[i].pad_ = 0;
    }
    return (int32_t)n;
}


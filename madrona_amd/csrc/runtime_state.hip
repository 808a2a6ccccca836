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

// ---- this translation unit: device memory and state, world construction, table / entity
// store / scratch growth and its service thread, host prints, device traces ----

__thread std::vector<void *> *t_allocScope = nullptr;

MWHIP_RT int devAlloc(mwhip_exec *exec, void **out, size_t bytes, bool zero)
{
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes == 0) bytes = 256;
    HIPCHK(hipMalloc(out, bytes));
    (t_allocScope != nullptr ? *t_allocScope : exec->allocations).push_back(*out);
    if (zero) {
        HIPCHK(hipMemset(*out, 0, bytes));
    }
    return 0;
}

static int vmEnsure(mwhip_exec *exec, VmRange &r, size_t bytes, bool zero)
{
    bytes = (bytes + kVmChunk - 1) / kVmChunk * kVmChunk;
    if (bytes > r.reserved) {
        return fail(-4, "growable range: %zu bytes requested, %zu reserved",
                    bytes, r.reserved);
    }

    hipMemAllocationProp prop {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = exec->cfg.gpu_id;
    hipMemAccessDesc access {};
    access.location = prop.location;
    access.flags = hipMemAccessFlagsProtReadWrite;

    while (r.mapped < bytes) {
        hipMemGenericAllocationHandle_t chunk;
        HIPCHK(hipMemCreate(&chunk, kVmChunk, &prop, 0));
        HIPCHK(hipMemMap(r.base + r.mapped, kVmChunk, 0, chunk, 0));
        HIPCHK(hipMemSetAccess(r.base + r.mapped, kVmChunk, &access, 1));
        if (zero) {
            // on the executor's side stream: a synchronous hipMemset waits for
            // the device to drain, and the service thread maps memory for
            // kernels that are waiting for exactly that
            HIPCHK(hipMemsetAsync(r.base + r.mapped, 0, kVmChunk,
                                  exec->serviceStream));
        }
        r.chunks.push_back(chunk);
        r.mapped += kVmChunk;
    }
    if (zero) {
        HIPCHK(hipStreamSynchronize(exec->serviceStream));
    }
    return 0;
}

MWHIP_RT int vmAlloc(mwhip_exec *exec, void **out, VmRange **range_out,
                   size_t reserve_bytes, size_t map_bytes, bool zero)
{
    std::unique_ptr<VmRange> r(new VmRange {});
    r->reserved = (reserve_bytes + kVmChunk - 1) / kVmChunk * kVmChunk;
    void *base = nullptr;
    HIPCHK(hipMemAddressReserve(&base, r->reserved, kVmChunk, nullptr, 0));
    r->base = (char *)base;
    exec->vmRanges.push_back(std::move(r));
    VmRange *range = exec->vmRanges.back().get();

    int rc = vmEnsure(exec, *range, map_bytes, zero);
    if (rc != 0) return rc;
    *out = range->base;
    *range_out = range;
    return 0;
}

MWHIP_RT void vmFreeAll(mwhip_exec *exec)
{
    for (auto &r : exec->vmRanges) {
        for (size_t i = 0; i < r->chunks.size(); i++) {
            (void)hipMemUnmap(r->base + i * kVmChunk, kVmChunk);
            (void)hipMemRelease(r->chunks[i]);
        }
        (void)hipMemAddressFree(r->base, r->reserved);
    }
    exec->vmRanges.clear();
}

// ---------------------------------------------------------------------------
// state construction
// ---------------------------------------------------------------------------
// Bottom-level BVHs, triangles and materials of the ray caster -> device.
static int uploadRenderGeometry(mwhip_exec *exec)
{
    const RenderGeometryHost &g = exec->renderGeometry;
    RenderGeometryDev &d = exec->renderGeometryDev;
    d.numObjects = g.numObjects;
    d.numMaterials = g.numMaterials;

    auto upload = [exec](const void *src, size_t bytes, const void **out) -> int {
        void *dev = nullptr;
        int rc = devAlloc(exec, &dev, bytes, false);
        if (rc != 0) return rc;
        if (bytes != 0) {
            HIPCHK(hipMemcpy(dev, src, bytes, hipMemcpyHostToDevice));
        }
        *out = dev;
        return 0;
    };
    int rc = upload(g.nodes.data(), g.nodes.size() * sizeof(BvhNode),
                    (const void **)&d.nodes);
    if (rc != 0) return rc;
    rc = upload(g.triangleVertices.data(), g.triangleVertices.size() * 4,
                (const void **)&d.triangleVertices);
    if (rc != 0) return rc;
    rc = upload(g.objectNodeOffset.data(), g.objectNodeOffset.size() * 4,
                (const void **)&d.objectNodeOffset);
    if (rc != 0) return rc;
    rc = upload(g.objectTriangleOffset.data(), g.objectTriangleOffset.size() * 4,
                (const void **)&d.objectTriangleOffset);
    if (rc != 0) return rc;
    rc = upload(g.objectMaterial.data(), g.objectMaterial.size() * 4,
                (const void **)&d.objectMaterial);
    if (rc != 0) return rc;
    rc = upload(g.objectRootBox.data(), g.objectRootBox.size() * 4,
                (const void **)&d.objectBounds);
    if (rc != 0) return rc;
    rc = upload(g.objectIsBox.data(), g.objectIsBox.size() * 4,
                (const void **)&d.objectIsBox);
    if (rc != 0) return rc;
    rc = upload(g.objectBoxFaces.data(), g.objectBoxFaces.size() * 4,
                (const void **)&d.objectBoxFaces);
    if (rc != 0) return rc;
    rc = upload(g.materialColor.data(), g.materialColor.size() * 4,
                (const void **)&d.materialColor);
    if (rc != 0) return rc;
    d.triangleUV = nullptr;
    d.triangleMaterial = nullptr;
    d.materialTexture = nullptr;
    d.textureInfo = nullptr;
    d.texels = nullptr;
    d.numTextures = (uint32_t)(g.textureInfo.size() / 4);
    if (!g.triangleUV.empty()) {
        rc = upload(g.triangleUV.data(), g.triangleUV.size() * 4,
                    (const void **)&d.triangleUV);
        if (rc != 0) return rc;
    }
    // (only what the shading can reach is uploaded -- the ray cast kernel is
    // compiled without the per-hit material / texture lookup when neither
    // pointer is set: per-triangle materials are consulted for objects without
    // a material of their own, textures through a material that has one)
    bool some_object_without_material = false;
    for (int32_t m : g.objectMaterial) {
        some_object_without_material = some_object_without_material || m < 0;
    }
    bool some_textured_material = false;
    for (int32_t t : g.materialTexture) {
        some_textured_material = some_textured_material || t >= 0;
    }
    if (!g.triangleMaterial.empty() && some_object_without_material) {
        rc = upload(g.triangleMaterial.data(), g.triangleMaterial.size() * 4,
                    (const void **)&d.triangleMaterial);
        if (rc != 0) return rc;
    }
    if (!g.materialTexture.empty() && some_textured_material) {
        rc = upload(g.materialTexture.data(), g.materialTexture.size() * 4,
                    (const void **)&d.materialTexture);
        if (rc != 0) return rc;
        rc = upload(g.textureInfo.data(), g.textureInfo.size() * 4,
                    (const void **)&d.textureInfo);
        if (rc != 0) return rc;
        rc = upload(g.texels.data(), g.texels.size() * 4, (const void **)&d.texels);
        if (rc != 0) return rc;
    }
    return 0;
}

MWHIP_RT int buildDeviceState(mwhip_exec *exec)
{
    const uint32_t W = exec->cfg.num_worlds;
    EcsState &hs = exec->hostState;

    hs.numArchetypeSlots = (uint32_t)exec->archetypes.size();
    hs.numComponentSlots = (uint32_t)exec->components.size();
    hs.numWorlds = (int32_t)W;

    // ---- table headers + dense (archetype, component) -> column lookup -----
    exec->tablesHost.assign(std::max<uint32_t>(hs.numArchetypeSlots, 1u),
                            TableHdr {});
    std::vector<uint16_t> lookup(
        (size_t)std::max<uint32_t>(hs.numArchetypeSlots, 1u) *
            std::max<uint32_t>(hs.numComponentSlots, 1u), kNoColumn);
    std::vector<void *> col_ptrs(lookup.size(), nullptr);

    for (uint32_t a = 0; a < hs.numArchetypeSlots; a++) {
        const ArchetypeRec &arch = exec->archetypes[a];
        TableHdr &hdr = exec->tablesHost[a];
        if (!arch.registered) continue;

        for (uint32_t c = 0; c < arch.numColumns; c++) {
            hdr.columns[c] = arch.primary[c];
            hdr.columnsAlt[c] = arch.alt[c];
            hdr.columnBytes[c] = arch.colBytes[c];
            hdr.columnFlags[c] = arch.colFlags[c];
            hdr.columnComponent[c] = (uint16_t)arch.colComponent[c];
            lookup[(size_t)a * hs.numComponentSlots + arch.colComponent[c]] =
                (uint16_t)c;
            col_ptrs[(size_t)a * hs.numComponentSlots + arch.colComponent[c]] =
                arch.primary[c];
        }
        hdr.numColumns = (int32_t)arch.numColumns;
        hdr.numRows = arch.singleton ? (int32_t)W : 0;
        hdr.capacity = (int32_t)arch.capacity;
        hdr.needsSort = 0;
        hdr.worldOffsets = arch.worldOffsets;
        hdr.worldCounts = arch.worldCounts;
        hdr.maxPerWorld = arch.maxPerWorld;
        hdr.registered = 1;
        hdr.rowBytes = arch.rowBytes;
    }

    int rc = devAllocT(exec, &hs.tables, exec->tablesHost.size());
    if (rc != 0) return rc;
    rc = devAllocT(exec, &hs.colLookup, lookup.size());
    if (rc != 0) return rc;
    HIPCHK(hipMemcpy(hs.colLookup, lookup.data(),
        lookup.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    rc = devAllocT(exec, &hs.colPtr, col_ptrs.size());
    if (rc != 0) return rc;
    HIPCHK(hipMemcpy(hs.colPtr, col_ptrs.data(),
        col_ptrs.size() * sizeof(void *), hipMemcpyHostToDevice));

    rc = devAllocT(exec, &hs.queryData, exec->queryCapacity);
    if (rc != 0) return rc;

    // ---- entity store ---------------------------------------------------------
    const uint64_t singleton_ids = (uint64_t)exec->numSingletons * W;
    exec->singletonIdEnd =
        (uint32_t)((singleton_ids + kIdsPerBlock - 1) / kIdsPerBlock *
                   kIdsPerBlock);
    const uint32_t blocks_per_world =
        envU32("MADRONA_MWHIP_ID_BLOCKS_PER_WORLD", 4);
    uint64_t entity_capacity = (uint64_t)exec->singletonIdEnd +
        (uint64_t)W * blocks_per_world * kIdsPerBlock + kIdsPerBlock;
    if (entity_capacity > 0x7FFFFFF0ull) {
        return fail(-2, "entity store overflow");
    }
    const uint64_t default_entity_capacity = entity_capacity;
    // (test hook) start with a fraction of the ids the defaults provide
    if (const char *div = getenv("MADRONA_MWHIP_INITIAL_ID_CAPACITY_DIV")) {
        uint64_t d = strtoull(div, nullptr, 10);
        if (d > 1) {
            entity_capacity = std::max<uint64_t>(
                exec->singletonIdEnd + ((entity_capacity - exec->singletonIdEnd) / d +
                    kIdsPerBlock - 1) / kIdsPerBlock * kIdsPerBlock,
                exec->singletonIdEnd + kIdsPerBlock);
        }
    }
    hs.entityCapacity = (int32_t)entity_capacity;
    if (exec->tableGrowth > 1) {
        // growable like the tables: address space for 16 x the ids
        const uint64_t reserve_ids =
            std::min<uint64_t>(default_entity_capacity * 16, 0x7FFFFFF0ull);
        rc = vmAlloc(exec, (void **)&hs.entities, &exec->entityVm,
                     reserve_ids * sizeof(EntitySlot),
                     entity_capacity * sizeof(EntitySlot), true);
        if (rc != 0) return rc;
    } else {
        rc = devAllocT(exec, &hs.entities, entity_capacity);
        if (rc != 0) return rc;
    }
    rc = devAllocT(exec, &hs.worldCaches, W);
    if (rc != 0) return rc;
    rc = devAllocT(exec, &hs.initBlockBase, W);
    if (rc != 0) return rc;

    // ---- per-world user data + scratch allocator --------------------------------
    uint32_t align = std::max<uint32_t>(exec->cfg.world_data_alignment, 16u);
    hs.worldDataStride =
        (exec->cfg.num_world_data_bytes + align - 1) / align * align;
    rc = devAlloc(exec, (void **)&hs.worldData,
                  (size_t)hs.worldDataStride * W);
    if (rc != 0) return rc;

    hs.tmpCapacity =
        (unsigned long long)envU32("MADRONA_MWHIP_TMP_MB", 64) << 20;
    if (exec->tableGrowth > 1) {
        rc = vmAlloc(exec, (void **)&hs.tmpBase, &exec->tmpVm,
                     hs.tmpCapacity * 16, hs.tmpCapacity, false);
        if (rc != 0) return rc;
    } else {
        rc = devAlloc(exec, (void **)&hs.tmpBase, hs.tmpCapacity, false);
        if (rc != 0) return rc;
    }
    hs.tmpOffset = 0;

    hs.persistCapacity = (unsigned long long)W *
        envU32("MADRONA_MWHIP_PERSIST_KB_PER_WORLD", 16) * 1024ull + (1ull << 20);
    rc = devAlloc(exec, (void **)&hs.persistBase, hs.persistCapacity, false);
    if (rc != 0) return rc;
    hs.persistOffset = 0;

    hs.idFreeHead = 0xFFFFFFFFull;      // {gen 0, head sentinel}
    hs.numIds = (int32_t)exec->singletonIdEnd;
    hs.initMode = 0;
    hs.errorFlags = 0;
    hs.hostExec = nullptr;

    // ---- singletons: one row per world, ids in (singleton, world) order -------
    // (reference CPU state.inl:163-179: k-th created singleton entity gets id k)
    std::vector<Entity> ents(W);
    std::vector<int32_t> iota(W), ones(W, 1);
    std::vector<EntitySlot> slots(W);
    for (uint32_t a = 0; a < hs.numArchetypeSlots; a++) {
        const ArchetypeRec &arch = exec->archetypes[a];
        if (!arch.registered || !arch.singleton) continue;

        const uint32_t base = (uint32_t)arch.singletonOrdinal * W;
        for (uint32_t w = 0; w < W; w++) {
            ents[w] = Entity { 0, (int32_t)(base + w) };
            iota[w] = (int32_t)w;
            slots[w].loc.archetype = a;
            slots[w].loc.row = (int32_t)w;
            slots[w].gen = 0;
        }
        HIPCHK(hipMemcpy(arch.primary[0], ents.data(), W * sizeof(Entity),
                         hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(arch.primary[1], iota.data(), W * sizeof(int32_t),
                         hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(arch.worldOffsets, iota.data(), W * sizeof(int32_t),
                         hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(arch.worldCounts, ones.data(), W * sizeof(int32_t),
                         hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(hs.entities + base, slots.data(),
                         W * sizeof(EntitySlot), hipMemcpyHostToDevice));
    }

    HIPCHK(hipMemcpy(hs.tables, exec->tablesHost.data(),
        exec->tablesHost.size() * sizeof(TableHdr), hipMemcpyHostToDevice));
    if (!exec->queryDataHost.empty()) {
        HIPCHK(hipMemcpy(hs.queryData, exec->queryDataHost.data(),
            exec->queryDataHost.size() * sizeof(uint32_t),
            hipMemcpyHostToDevice));
    }

    // replay counter in signal memory (hipStreamWaitValue32 polls it; ParallelFor
    // nodes derive their per-launch tag from it)
    if (hipExtMallocWithFlags((void **)&exec->replaySignal, 256,
                              hipMallocSignalMemory) != hipSuccess) {
        (void)hipGetLastError();
        HIPCHK(hipMalloc((void **)&exec->replaySignal, 256));
    }
    exec->allocations.push_back(exec->replaySignal);
    HIPCHK(hipMemset(exec->replaySignal, 0, 256));
    hs.replayCounter = exec->replaySignal;
#ifdef MADRONA_TRACING
    {
        // (only the index needs clearing; -1 until a traced graph starts a step)
        HIPCHK(hipMalloc(&exec->deviceTracing, sizeof(mwGPU::DeviceTracing)));
        exec->allocations.push_back(exec->deviceTracing);
        const int32_t off = -1;
        HIPCHK(hipMemcpy(exec->deviceTracing, &off, sizeof(off), hipMemcpyHostToDevice));
        HIPCHK(hipMalloc(&exec->traceCursor, 256));
        exec->allocations.push_back(exec->traceCursor);
        HIPCHK(hipMemset(exec->traceCursor, 0, 256));
        hs.deviceTracing = exec->deviceTracing;
        hs.traceCursor = exec->traceCursor;
    }
#endif

    // batch ray caster configuration (render-prep systems read it on the device)
    hs.raycastOutputResolution = exec->cfg.raycast_output_resolution;
    hs.raycastRGBD = exec->cfg.raycast_rgbd;
    {
        // object-space root boxes (TLBVH leaves are made from them): as given,
        // else those of the geometry handed over for the ray caster
        const float *boxes = exec->cfg.object_root_aabbs;
        uint32_t num_boxes = exec->cfg.num_object_root_aabbs;
        if ((boxes == nullptr || num_boxes == 0) && exec->haveRenderGeometry) {
            boxes = exec->renderGeometry.objectRootBox.data();
            num_boxes = exec->renderGeometry.numObjects;
        }
        if (boxes != nullptr && num_boxes != 0) {
            void *aabbs_dev = nullptr;
            const size_t bytes = (size_t)num_boxes * 24;
            rc = devAlloc(exec, &aabbs_dev, bytes, false);
            if (rc != 0) return rc;
            HIPCHK(hipMemcpy(aabbs_dev, boxes, bytes, hipMemcpyHostToDevice));
            hs.moduleData[2] = aabbs_dev;
        }
    }
    if (exec->haveRenderGeometry) {
        rc = uploadRenderGeometry(exec);
        if (rc != 0) return rc;
    }
    // (caller's memory: not valid after mwhip_create returns)
    exec->cfg.object_root_aabbs = nullptr;
    exec->cfg.render_geometry = nullptr;

    // device -> host requests for table memory
    HIPCHK(hipHostMalloc((void **)&exec->growMailbox, sizeof(GrowMailbox),
                         hipHostMallocMapped));
    memset((void *)exec->growMailbox, 0, sizeof(GrowMailbox));
    for (uint32_t a = 0; a < exec->archetypes.size() && a < kMaxArchetypes; a++) {
        if (exec->archetypes[a].registered) {
            exec->growMailbox->capacity[a] = (int32_t)exec->archetypes[a].capacity;
        }
    }
    exec->growMailbox->capacity[kGrowSlotEntities] = hs.entityCapacity;
    exec->growMailbox->capacity[kGrowSlotTmp] = (int32_t)(hs.tmpCapacity >> 10);
    HIPCHK(hipHostGetDevicePointer((void **)&hs.growMailbox,
                                   exec->growMailbox, 0));

    // device -> host message ring of mwGPU::HostPrint
    HIPCHK(hipHostMalloc((void **)&exec->printRing, sizeof(HostPrintRing),
                         hipHostMallocMapped));
    memset((void *)exec->printRing, 0, sizeof(HostPrintRing));
    HIPCHK(hipHostGetDevicePointer((void **)&hs.hostPrintRing,
                                   exec->printRing, 0));

    rc = devAllocT(exec, &exec->stateDev, 1);
    if (rc != 0) return rc;
    HIPCHK(hipMemcpy(exec->stateDev, &hs, sizeof(EcsState),
                     hipMemcpyHostToDevice));

    // [0] error flags, [1] id high-water mark, rows / per-step peak rows per
    // archetype, profiling gate flag, replays completed (kStats* above)
    HIPCHK(hipHostMalloc((void **)&exec->statsHost,
        kStatsWords * sizeof(int32_t), hipHostMallocMapped));
    memset(exec->statsHost, 0, kStatsWords * sizeof(int32_t));

    exec->stateBuilt = true;
    return 0;
}

// ---------------------------------------------------------------------------
// mwGPU::HostPrint, host side: "{}" placeholders, one line per record
// (reference HostPrintCPU, src/mw/cuda_exec.cpp: same placeholder syntax)
// ---------------------------------------------------------------------------
static void printRecord(const HostPrintRecord &rec)
{
    std::string out;
    uint32_t next_arg = 0;
    char num[64];
    for (const char *p = rec.fmt; *p != '\0' &&
             p < rec.fmt + HostPrintRecord::maxChars; p++) {
        if (p[0] == '{' && p[1] == '}' && next_arg < rec.numArgs &&
                next_arg < (uint32_t)HostPrintRecord::maxArgs) {
            const uint64_t v = rec.args[next_arg];
            switch (rec.types[next_arg]) {
            case HostPrintRecord::I32:
                snprintf(num, sizeof(num), "%d", (int32_t)(int64_t)v); break;
            case HostPrintRecord::U32:
                snprintf(num, sizeof(num), "%u", (uint32_t)v); break;
            case HostPrintRecord::I64:
                snprintf(num, sizeof(num), "%" PRId64, (int64_t)v); break;
            case HostPrintRecord::U64:
                snprintf(num, sizeof(num), "%" PRIu64, v); break;
            case HostPrintRecord::Float: {
                uint32_t bits = (uint32_t)v;
                float f;
                memcpy(&f, &bits, sizeof(f));
                snprintf(num, sizeof(num), "%f", f);
            } break;
            default:
                snprintf(num, sizeof(num), "%p", (void *)(uintptr_t)v); break;
            }
            out += num;
            next_arg++;
            p++;
        } else {
            out += *p;
        }
    }
    printf("%s\n", out.c_str());
}

// Prints completed records in ticket order and stops at the first incomplete
// one (its writer is still running).  in_flight: called between replays by the
// service thread; otherwise the stream has been waited for, every record below
// head is complete, and the drop count is reported.
MWHIP_RT void drainHostPrints(mwhip_exec *exec, bool in_flight)
{
    HostPrintRing *ring = exec->printRing;
    if (ring == nullptr) return;
    std::lock_guard<std::mutex> guard(exec->printMutex);

    const uint64_t head = __atomic_load_n(&ring->head, __ATOMIC_ACQUIRE);
    uint64_t tail = ring->tail;
    bool printed = false;
    while (tail < head) {
        HostPrintRecord &rec = ring->records[tail % HostPrintRing::numRecords];
        const uint64_t seq = __atomic_load_n(&rec.seq, __ATOMIC_ACQUIRE);
        if (seq == tail + 1) {
            printRecord(rec);
            printed = true;
        } else {
            // Not there yet: its writer is still filling it in (tickets are
            // only taken when the ring has room, host_print.hpp: there are no
            // holes to skip).  After a replay the host waited for, every
            // writer has finished and this does not happen.
            (void)in_flight;
            break;
        }
        tail += 1;
        __atomic_store_n(&ring->tail, tail, __ATOMIC_RELEASE);
    }
    if (!in_flight) {
        const uint64_t dropped =
            __atomic_exchange_n(&ring->dropped, 0ull, __ATOMIC_RELAXED);
        if (dropped != 0) {
            printf("madrona_amd: HostPrint ring overflow, %" PRIu64
                   " message(s) dropped\n", dropped);
            printed = true;
        }
    }
    if (printed) fflush(stdout);
}

template <typename T>
static int pokeState(mwhip_exec *exec, T EcsState::*field, const T &value)
{
    exec->hostState.*field = value;
    char *dst = (char *)exec->stateDev +
        ((char *)&(exec->hostState.*field) - (char *)&exec->hostState);
    HIPCHK(hipMemcpy(dst, &value, sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

static int fetchError(mwhip_exec *exec, uint32_t *flags)
{
    HIPCHK(hipMemcpy(flags, (char *)exec->stateDev +
        offsetof(EcsState, errorFlags), sizeof(uint32_t),
        hipMemcpyDeviceToHost));
    return 0;
}

MWHIP_RT const char *describeError(uint32_t flags)
{
    if (flags & kErrTableOverflow) {
        return "an archetype table ran out of rows (raise "
               "max_num_entities_per_world or MADRONA_MWHIP_ROWS_PER_WORLD)";
    }
    if (flags & kErrEntityOverflow) {
        return "the entity id store is full (raise "
               "MADRONA_MWHIP_ID_BLOCKS_PER_WORLD)";
    }
    if (flags & kErrTmpOverflow) {
        return "Context::tmpAlloc scratch exhausted (raise MADRONA_MWHIP_TMP_MB)";
    }
    if (flags & kErrSortLookback) {
        return "sort look-back timed out";
    }
    if (flags & kErrPersistOverflow) {
        return "persistent world-constructor allocations exhausted (raise "
               "MADRONA_MWHIP_PERSIST_KB_PER_WORLD)";
    }
    if (flags & kErrPhysics) {
        return "physics capacity exceeded (BVH leaves / traversal stack / "
               "hull scratch)";
    }
    if (flags & kErrInitBlocks) {
        return "world constructors are not deterministic";
    }
    if (flags & kErrRender) {
        return "ray caster: a world holds more than 1024 instances, the "
               "instance table was not world-sorted, or a traversal stack "
               "overflowed";
    }
    return "unknown device error";
}

// ---------------------------------------------------------------------------
// world construction (two passes, deterministic id blocks)
// ---------------------------------------------------------------------------
static int launchInitWorlds(mwhip_exec *exec)
{
    const int32_t W = (int32_t)exec->cfg.num_worlds;
    EcsState *state = exec->stateDev;
    const void *cfg = exec->userCfgDev;
    const void *inits = exec->worldInitsDev;
    void *args[] = { &state, &cfg, &inits, (void *)&W };
    HIPCHK(hipLaunchKernel(exec->entry.init_worlds_kernel,
        dim3((uint32_t)((W + 63) / 64), 1, 1), dim3(64, 1, 1), args, 0,
        exec->stream));
    HIPCHK(hipStreamSynchronize(exec->stream));
    return 0;
}

static int resetForInitPass(mwhip_exec *exec)
{
    const uint32_t W = exec->cfg.num_worlds;
    EcsState &hs = exec->hostState;

    HIPCHK(hipMemcpy(hs.tables, exec->tablesHost.data(),
        exec->tablesHost.size() * sizeof(TableHdr), hipMemcpyHostToDevice));
    HIPCHK(hipMemset(hs.worldCaches, 0, W * sizeof(IdCache)));
    std::vector<IdCache> caches(W);
    for (IdCache &c : caches) {
        c.freeHead = kIdSentinel;
        c.numFree = 0;
        c.overflowHead = kIdSentinel;
        c.numOverflow = 0;
        c.lock = 0;
        c.initBlocksUsed = 0;
        c.runtimeBlocksUsed = 0;
    }
    HIPCHK(hipMemcpy(hs.worldCaches, caches.data(), W * sizeof(IdCache),
                     hipMemcpyHostToDevice));

    HIPCHK(hipMemset(hs.entities + exec->singletonIdEnd, 0,
        (size_t)(hs.entityCapacity - (int32_t)exec->singletonIdEnd) *
            sizeof(EntitySlot)));
    HIPCHK(hipMemset(hs.worldData, 0, (size_t)hs.worldDataStride * W));

    // singleton user data back to zero
    for (const ArchetypeRec &arch : exec->archetypes) {
        if (arch.registered && arch.singleton) {
            HIPCHK(hipMemset(arch.primary[2], 0,
                             (size_t)arch.capacity * arch.colBytes[2]));
        }
    }

    int rc = pokeState(exec, &EcsState::numIds, (int32_t)exec->singletonIdEnd);
    if (rc != 0) return rc;
    rc = pokeState(exec, &EcsState::tmpOffset, 0ull);
    if (rc != 0) return rc;
    rc = pokeState(exec, &EcsState::persistOffset, 0ull);
    if (rc != 0) return rc;
    rc = pokeState(exec, &EcsState::idFreeHead, 0xFFFFFFFFull);
    if (rc != 0) return rc;
    return pokeState(exec, &EcsState::errorFlags, 0u);
}

// Entity store / scratch region: more memory mapped behind them (addresses do
// not change), the mailbox told; the device header follows at the next replay
// boundary (growTables).  Caller holds growMutex (or runs before the service
// thread exists).
static int growEntityStore(mwhip_exec *exec, uint64_t new_ids)
{
    EcsState &hs = exec->hostState;
    if (exec->entityVm == nullptr) {
        return fail(-4, "the entity store is a fixed allocation "
                    "(MADRONA_MWHIP_TABLE_GROWTH=1)");
    }
    new_ids = (new_ids + kIdsPerBlock - 1) / kIdsPerBlock * kIdsPerBlock;
    new_ids = std::min<uint64_t>(new_ids,
                                 exec->entityVm->reserved / sizeof(EntitySlot));
    if (new_ids <= (uint64_t)hs.entityCapacity) {
        return fail(-4, "the entity store's reserved address space is used up "
                    "(%d ids)", hs.entityCapacity);
    }
    int rc = vmEnsure(exec, *exec->entityVm, new_ids * sizeof(EntitySlot), true);
    if (rc != 0) return rc;
    hs.entityCapacity = (int32_t)new_ids;
    if (exec->growMailbox != nullptr) {
        __atomic_store_n(&exec->growMailbox->capacity[kGrowSlotEntities],
                         (int32_t)new_ids, __ATOMIC_RELEASE);
    }
    exec->numGrowths++;
    exec->headersStale = true;
    if (getenv("MADRONA_MWHIP_DEBUG_GROWTH") != nullptr) {
        fprintf(stderr, "madrona_amd: entity store now holds %llu ids\n",
                (unsigned long long)new_ids);
    }
    return 0;
}

static int growTmpRegion(mwhip_exec *exec, uint64_t new_bytes)
{
    EcsState &hs = exec->hostState;
    if (exec->tmpVm == nullptr) {
        return fail(-4, "the scratch region is a fixed allocation "
                    "(MADRONA_MWHIP_TABLE_GROWTH=1)");
    }
    new_bytes = std::min<uint64_t>((new_bytes + kVmChunk - 1) / kVmChunk * kVmChunk,
                                   exec->tmpVm->reserved);
    if (new_bytes <= hs.tmpCapacity) {
        return fail(-4, "the scratch region's reserved address space is used up "
                    "(%llu bytes)", (unsigned long long)hs.tmpCapacity);
    }
    int rc = vmEnsure(exec, *exec->tmpVm, new_bytes, false);
    if (rc != 0) return rc;
    hs.tmpCapacity = new_bytes;
    if (exec->growMailbox != nullptr) {
        __atomic_store_n(&exec->growMailbox->capacity[kGrowSlotTmp],
                         (int32_t)(new_bytes >> 10), __ATOMIC_RELEASE);
    }
    exec->numGrowths++;
    exec->headersStale = true;
    if (getenv("MADRONA_MWHIP_DEBUG_GROWTH") != nullptr) {
        fprintf(stderr, "madrona_amd: scratch region now %llu MiB\n",
                (unsigned long long)(new_bytes >> 20));
    }
    return 0;
}

MWHIP_RT int constructWorlds(mwhip_exec *exec)
{
    const uint32_t W = exec->cfg.num_worlds;
    EcsState &hs = exec->hostState;

    // pass 1: run the constructors to learn how many id blocks each world
    // takes from the global store
    int rc = resetForInitPass(exec);
    if (rc != 0) return rc;
    rc = pokeState(exec, &EcsState::initMode, 1u);
    if (rc != 0) return rc;
    rc = launchInitWorlds(exec);
    if (rc != 0) return rc;

    uint32_t err = 0;
    rc = fetchError(exec, &err);
    if (rc != 0) return rc;
    // (both recoverable conditions may be raised by one pass: test bits, handle
    // the persistent region first -- after its overflow every world aliases
    // persistBase, which can produce secondary flags that the rerun clears)
    if ((err & kErrPersistOverflow) != 0u) {
        // The constructors asked for more persistent memory (BVH arrays, ...)
        // than MADRONA_MWHIP_PERSIST_KB_PER_WORLD provides.  persistAlloc kept
        // counting, so the offset is what they need: size the region for it
        // and run pass 1 again.
        unsigned long long needed = 0;
        HIPCHK(hipMemcpy(&needed, (char *)exec->stateDev +
            offsetof(EcsState, persistOffset), sizeof(needed),
            hipMemcpyDeviceToHost));
        unsigned long long capacity = needed + needed / 8 + (1ull << 20);
        char *region = nullptr;
        rc = devAlloc(exec, (void **)&region, capacity, false);
        if (rc != 0) return rc;
        rc = pokeState(exec, &EcsState::persistBase, region);
        if (rc != 0) return rc;
        rc = pokeState(exec, &EcsState::persistCapacity, capacity);
        if (rc != 0) return rc;

        rc = resetForInitPass(exec);
        if (rc != 0) return rc;
        rc = pokeState(exec, &EcsState::initMode, 1u);
        if (rc != 0) return rc;
        rc = launchInitWorlds(exec);
        if (rc != 0) return rc;
        rc = fetchError(exec, &err);
        if (rc != 0) return rc;
    }
    // Constructors that create more rows than a table was given at
    // registration: full tables grow (they live in reserved address space,
    // growTables) and pass 1 runs again, until everything fits or the
    // reservations are exhausted.
    // Likewise the entity store when the constructors take more id blocks than
    // it was sized for.
    constexpr uint32_t kGrowable = kErrTableOverflow | kErrEntityOverflow;
    for (int attempt = 0; (err & kGrowable) != 0u &&
             (err & ~kGrowable) == 0u && attempt < 8; attempt++) {
        const uint32_t before = exec->numGrowths;
        if ((err & kErrEntityOverflow) != 0u && exec->entityVm != nullptr) {
            int32_t wanted = 0;     // (pass 1 keeps counting past the end)
            HIPCHK(hipMemcpy(&wanted, (char *)exec->stateDev +
                offsetof(EcsState, numIds), sizeof(wanted),
                hipMemcpyDeviceToHost));
            rc = growEntityStore(exec, std::max<uint64_t>(
                (uint64_t)wanted + (uint64_t)wanted / 4,
                2ull * (uint64_t)hs.entityCapacity));
            if (rc != 0) return rc;
            rc = pokeState(exec, &EcsState::entityCapacity, hs.entityCapacity);
            if (rc != 0) return rc;
        }
        if ((err & kErrTableOverflow) != 0u) {
            rc = growTablesFromDevice(exec);
            if (rc != 0) return rc;
        }
        if (exec->numGrowths == before) {
            break;      // nothing left to grow: report the overflow
        }

        rc = resetForInitPass(exec);
        if (rc != 0) return rc;
        rc = pokeState(exec, &EcsState::initMode, 1u);
        if (rc != 0) return rc;
        rc = launchInitWorlds(exec);
        if (rc != 0) return rc;
        rc = fetchError(exec, &err);
        if (rc != 0) return rc;
    }
    if (err != 0) {
        return fail(-4, "world construction failed: %s", describeError(err));
    }

    std::vector<IdCache> caches(W);
    HIPCHK(hipMemcpy(caches.data(), hs.worldCaches, W * sizeof(IdCache),
                     hipMemcpyDeviceToHost));

    // pass 2: replay with world-major block bases, i.e. the order in which the
    // reference CPU backend's sequential constructor loop
    // (include/madrona/mw_cpu.inl:42-46) would have grabbed them
    std::vector<int32_t> bases(W);
    int64_t next = exec->singletonIdEnd;
    for (uint32_t w = 0; w < W; w++) {
        bases[w] = (int32_t)next;
        next += (int64_t)caches[w].initBlocksUsed * kIdsPerBlock;
    }
    if (next + (int64_t)kIdsPerBlock > hs.entityCapacity) {
        if (exec->entityVm == nullptr) {
            return fail(-4, "world construction failed: %s",
                        describeError(kErrEntityOverflow));
        }
        rc = growEntityStore(exec, (uint64_t)next + (uint64_t)next / 4 +
                                   kIdsPerBlock);
        if (rc != 0) return rc;
        rc = pokeState(exec, &EcsState::entityCapacity, hs.entityCapacity);
        if (rc != 0) return rc;
    }

    rc = resetForInitPass(exec);
    if (rc != 0) return rc;
    HIPCHK(hipMemcpy(hs.initBlockBase, bases.data(), W * sizeof(int32_t),
                     hipMemcpyHostToDevice));
    rc = pokeState(exec, &EcsState::numIds, (int32_t)next);
    if (rc != 0) return rc;
    rc = pokeState(exec, &EcsState::initMode, 2u);
    if (rc != 0) return rc;
    rc = launchInitWorlds(exec);
    if (rc != 0) return rc;

    std::vector<IdCache> caches2(W);
    HIPCHK(hipMemcpy(caches2.data(), hs.worldCaches, W * sizeof(IdCache),
                     hipMemcpyDeviceToHost));
    for (uint32_t w = 0; w < W; w++) {
        if (caches2[w].initBlocksUsed != caches[w].initBlocksUsed) {
            return fail(-4, "world %u constructor is not deterministic "
                "(%d id blocks, then %d)", w, caches[w].initBlocksUsed,
                caches2[w].initBlocksUsed);
        }
    }

    rc = fetchError(exec, &err);
    if (rc != 0) return rc;
    if (err != 0) {
        return fail(-4, "world construction failed: %s", describeError(err));
    }

    rc = pokeState(exec, &EcsState::runtimeIdBase, (int32_t)next);
    if (rc != 0) return rc;
    return pokeState(exec, &EcsState::initMode, 0u);
}

// MADRONA_MWHIP_EXEC_CONFIG_FILE: { "<node index>": <workgroups per CU>, ... } --
// the format of the reference's exec-config file (cuda_exec.cpp:2115-2172:
// node index -> blocks per SM).  Written by madrona_amd/scripts/profile.py.
MWHIP_RT int loadExecConfigFile(mwhip_exec *exec)
{
    const char *path = getenv("MADRONA_MWHIP_EXEC_CONFIG_FILE");
    if (path == nullptr || path[0] == '\0') {
        return 0;
    }
    FILE *f = fopen(path, "rb");
    if (f == nullptr) {
        return fail(-2, "MADRONA_MWHIP_EXEC_CONFIG_FILE: cannot open %s", path);
    }
    std::string text;
    char buf[4096];
    size_t got;
    while ((got = fread(buf, 1, sizeof(buf), f)) > 0) {
        text.append(buf, got);
    }
    fclose(f);

    // a flat object of "digits": digits pairs
    size_t at = 0;
    auto skip = [&]() {
        while (at < text.size() && (isspace((unsigned char)text[at]) ||
                                    text[at] == ',')) at++;
    };
    skip();
    if (at >= text.size() || text[at] != '{') {
        return fail(-2, "MADRONA_MWHIP_EXEC_CONFIG_FILE points to invalid file");
    }
    at++;
    for (;;) {
        skip();
        if (at < text.size() && text[at] == '}') break;
        if (at >= text.size() || text[at] != '"') {
            return fail(-2, "MADRONA_MWHIP_EXEC_CONFIG_FILE points to invalid file");
        }
        at++;
        unsigned long long node = 0, value = 0;
        size_t digits = 0;
        while (at < text.size() && isdigit((unsigned char)text[at])) {
            node = node * 10 + (unsigned)(text[at++] - '0');
            digits++;
        }
        if (digits == 0 || at >= text.size() || text[at] != '"' || node > 16384) {
            return fail(-2, "MADRONA_MWHIP_EXEC_CONFIG_FILE points to invalid file");
        }
        at++;
        skip();
        if (at >= text.size() || text[at] != ':') {
            return fail(-2, "MADRONA_MWHIP_EXEC_CONFIG_FILE points to invalid file");
        }
        at++;
        skip();
        digits = 0;
        while (at < text.size() && isdigit((unsigned char)text[at])) {
            value = value * 10 + (unsigned)(text[at++] - '0');
            digits++;
        }
        if (digits == 0 || value > 64) {
            return fail(-2, "MADRONA_MWHIP_EXEC_CONFIG_FILE points to invalid file");
        }
        if (node >= exec->nodeWorkgroupsPerCU.size()) {
            exec->nodeWorkgroupsPerCU.resize(node + 1, 0u);
        }
        exec->nodeWorkgroupsPerCU[node] = (uint32_t)value;
    }
    return 0;
}

#ifdef MADRONA_TRACING
#endif
// Table growth, between replays (the stream is idle).  A table whose live rows
// fill more than half of its backed capacity gets more memory mapped behind
// every column (primary and ping-pong twin) and behind its sort buffers --
// addresses do not change --, the capacity in its device header is raised, and
// the launch graphs are rebuilt so that grids follow the new size.  Half,
// because rows destroyed and re-created in one step coexist until the step's
// compaction; a single step that outruns the head room still raises
// kErrTableOverflow, as a fixed-capacity table does.
// rows_of(a) = live rows of archetype a, or -1.
// Maps memory for `new_capacity` rows behind every column, twin and sort buffer
// of archetype a (addresses do not change) and tells the device through the
// mailbox.  The table's device header and the launch graphs are brought up to
// date by refreshAfterGrowth(), with the stream idle.  Caller holds growMutex.
static int mapTableRows(mwhip_exec *exec, uint32_t a, uint64_t new_capacity)
{
    ArchetypeRec &arch = exec->archetypes[a];
    for (uint32_t c = 0; c < arch.numColumns; c++) {
        size_t bytes = (size_t)new_capacity * arch.colBytes[c] + 16;
        int rc = vmEnsure(exec, *arch.primaryVm[c], bytes, true);
        if (rc != 0) return rc;
        rc = vmEnsure(exec, *arch.altVm[c], bytes, true);
        if (rc != 0) return rc;
    }
    for (VmRange *r : arch.sortVm) {
        if (r != nullptr) {
            int rc = vmEnsure(exec, *r, (size_t)new_capacity * 4, false);
            if (rc != 0) return rc;
        }
    }
    arch.capacity = (uint32_t)new_capacity;
    exec->tablesHost[a].capacity = (int32_t)new_capacity;
    if (exec->growMailbox != nullptr && a < kMaxArchetypes) {
        __atomic_store_n(&exec->growMailbox->capacity[a], (int32_t)new_capacity,
                         __ATOMIC_RELEASE);
    }
    exec->numGrowths++;
    exec->headersStale = true;
    if (getenv("MADRONA_MWHIP_DEBUG_GROWTH") != nullptr) {
        fprintf(stderr, "madrona_amd: archetype %u now has %llu rows mapped\n",
                a, (unsigned long long)new_capacity);
    }
    return 0;
}

// Service thread: answers the device's requests while a replay is running.
MWHIP_RT void serviceGrowRequests(mwhip_exec *exec)
{
    GrowMailbox *mb = exec->growMailbox;
    if (mb == nullptr) return;
    bool pending = false;
    for (uint32_t a = 0; a < exec->archetypes.size() && a < kMaxArchetypes; a++) {
        if (__atomic_load_n(&mb->requested[a], __ATOMIC_RELAXED) >
                mb->capacity[a]) {
            pending = true;
            break;
        }
    }
    for (uint32_t slot : { kGrowSlotEntities, kGrowSlotTmp }) {
        if (__atomic_load_n(&mb->requested[slot], __ATOMIC_RELAXED) >
                mb->capacity[slot]) {
            pending = true;
        }
    }
    if (!pending) return;

    std::lock_guard<std::mutex> guard(exec->growMutex);
    (void)hipSetDevice(exec->cfg.gpu_id);
    {
        const int64_t ids = __atomic_load_n(&mb->requested[kGrowSlotEntities],
                                            __ATOMIC_RELAXED);
        if (ids > mb->capacity[kGrowSlotEntities] && exec->entityVm != nullptr) {
            // a layer of run-time blocks (one per world) at least
            (void)growEntityStore(exec, std::max<uint64_t>(
                2ull * (uint64_t)ids, 2ull * (uint64_t)exec->hostState.entityCapacity));
        }
        const int64_t kib = __atomic_load_n(&mb->requested[kGrowSlotTmp],
                                            __ATOMIC_RELAXED);
        if (kib > mb->capacity[kGrowSlotTmp] && exec->tmpVm != nullptr) {
            (void)growTmpRegion(exec, std::max<uint64_t>(
                2ull * ((uint64_t)kib << 10), 2ull * exec->hostState.tmpCapacity));
        }
    }
    for (uint32_t a = 0; a < exec->archetypes.size() && a < kMaxArchetypes; a++) {
        ArchetypeRec &arch = exec->archetypes[a];
        const int64_t wanted = __atomic_load_n(&mb->requested[a], __ATOMIC_RELAXED);
        if (!arch.registered || wanted <= (int64_t)arch.capacity ||
                arch.reservedCapacity <= arch.capacity) {
            continue;
        }
        uint64_t new_capacity = std::max<uint64_t>(2ull * arch.capacity,
                                                   2ull * (uint64_t)wanted);
        new_capacity = std::min<uint64_t>(new_capacity, arch.reservedCapacity);
        if (getenv("MADRONA_MWHIP_DEBUG_GROWTH") != nullptr) {
            fprintf(stderr, "madrona_amd: on-demand growth of archetype %u: "
                    "%u -> %llu rows (wanted %lld)\n", a, arch.capacity,
                    (unsigned long long)new_capacity, (long long)wanted);
        }
        if (mapTableRows(exec, a, new_capacity) != 0) {
            if (getenv("MADRONA_MWHIP_DEBUG_GROWTH") != nullptr) {
                fprintf(stderr, "madrona_amd: mapping failed: %s\n",
                        mwhip_last_error());
            }
            // the waiting threads time out and raise the overflow flag
            return;
        }
    }
}

template <typename RowsFn>
static int growTables(mwhip_exec *exec, RowsFn &&rows_of)
{
    // Anything to do?  If so the stream is drained BEFORE the lock is taken for
    // the work: a replay in flight may be waiting for the service thread, which
    // needs the same lock.
    auto wants_growth = [&](uint32_t a) {
        const ArchetypeRec &arch = exec->archetypes[a];
        if (!arch.registered || arch.reservedCapacity <= arch.capacity) {
            return false;
        }
        int64_t rows = rows_of(a);
        return rows >= 0 && 2 * rows > (int64_t)arch.capacity;
    };
    {
        std::lock_guard<std::mutex> peek(exec->growMutex);
        bool needed = exec->headersStale;
        for (uint32_t a = 0; a < exec->archetypes.size() && !needed; a++) {
            needed = wants_growth(a);
        }
        if (!needed) {
            return 0;
        }
    }
    HIPCHK(hipStreamSynchronize(exec->stream));

    std::lock_guard<std::mutex> guard(exec->growMutex);
    bool grew = exec->headersStale;
    for (uint32_t a = 0; a < exec->archetypes.size(); a++) {
        ArchetypeRec &arch = exec->archetypes[a];
        if (!arch.registered || arch.reservedCapacity <= arch.capacity) {
            continue;
        }
        int64_t rows = rows_of(a);
        if (rows < 0 || 2 * rows <= (int64_t)arch.capacity) {
            continue;
        }

        uint64_t new_capacity = std::max<uint64_t>(2ull * arch.capacity,
                                                   3ull * (uint64_t)rows);
        new_capacity = std::min<uint64_t>(new_capacity, arch.reservedCapacity);
        int rc = mapTableRows(exec, a, new_capacity);
        if (rc != 0) return rc;
        grew = true;
    }

    if (!grew) {
        return 0;
    }

    // device headers follow what is mapped (also after on-demand growth by the
    // service thread), then the graphs are rebuilt for the new sizes
    for (uint32_t a = 0; a < exec->archetypes.size(); a++) {
        const ArchetypeRec &arch = exec->archetypes[a];
        if (!arch.registered) continue;
        int32_t cap = (int32_t)arch.capacity;
        HIPCHK(hipMemcpy((char *)(exec->hostState.tables + a) +
                             offsetof(TableHdr, capacity),
                         &cap, sizeof(cap), hipMemcpyHostToDevice));
    }
    {
        int rc = pokeState(exec, &EcsState::entityCapacity,
                           exec->hostState.entityCapacity);
        if (rc != 0) return rc;
        rc = pokeState(exec, &EcsState::tmpCapacity, exec->hostState.tmpCapacity);
        if (rc != 0) return rc;
    }
    exec->headersStale = false;

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

// row counts read back from the device's table headers
MWHIP_RT int growTablesFromDevice(mwhip_exec *exec)
{
    HIPCHK(hipStreamSynchronize(exec->stream));
    std::vector<TableHdr> hdrs(exec->tablesHost.size());
    HIPCHK(hipMemcpy(hdrs.data(), exec->hostState.tables,
                     hdrs.size() * sizeof(TableHdr), hipMemcpyDeviceToHost));
    return growTables(exec, [&hdrs](uint32_t a) -> int64_t {
        return a < hdrs.size() ? hdrs[a].numRows : -1;
    });
}

// A table sorted by the single-launch path has grown past half of what that
// path is meant for: its graphs are rebuilt with the chain (the single launch
// stays correct at any size, it is just one workgroup).
MWHIP_RT int sortsOutgrown(mwhip_exec *exec)
{
    bool rebuild = false;
    // a table whose appended tails keep exceeding what the compaction chain's
    // one workgroup sorts quickly goes back to the radix chain
    for (uint32_t a = 0; a < exec->archetypes.size() && a < kMaxArchetypes; a++) {
        ArchetypeRec &arch = exec->archetypes[a];
        if (!arch.registered || arch.noCompact || exec->sortCompaction != 1) continue;
        if ((uint32_t)std::max(exec->statsHost[kStatsTails + a], 0) >
                sortCompactTailLimit()) {
            exec->statsHost[kStatsTails + a] = 0;   // (counted once per report)
            if (++arch.longTails >= 3u) {
                arch.noCompact = true;
                rebuild = true;
            }
        }
    }
    for (auto &kv : exec->launchGraphs) {
        for (const auto &batch : kv.second->sortBatches) {
            if (!batch->small) continue;
            for (const SortSiteHost &site : batch->sites) {
                int64_t rows = site.archetype < kMaxArchetypes ?
                    exec->statsHost[kStatsRows + site.archetype] : 0;
                ArchetypeRec &arch = exec->archetypes[site.archetype];
                if (rows * 2 > (int64_t)sortSmallRowLimit() && !arch.bigSort) {
                    arch.bigSort = true;
                    rebuild = true;
                }
                // One workgroup moving a few thousand rows is slower than the
                // compaction chain's three launches (8192 Escape-Room worlds,
                // ~2 K joints re-sorted every step: 50 us against 27); the one
                // launch wins while the table is tiny or mostly idle (4 us
                // against 3 x 4 when nothing changed).  Three reports in a
                // row above the mark move a world-sorted table to the chain.
                // (One launch on 16-32 workgroups -- every workgroup ordering
                // all keys in LDS, then moving its share -- was built in round 4
                // and measured at 29 us: the device-scope hand-off between the
                // workgroups costs more than the launches it saves.  Removed in
                // round 5; profiles/r04_sort_variants.jsonl.)
                if (!arch.bigSort && site.worldSort &&
                        compactionEligible(exec, site.archetype, 1u) &&
                        rows >= (int64_t)sortSmallBusyRows()) {
                    if (++arch.smallBusy >= 3u) {
                        arch.bigSort = true;
                        rebuild = true;
                    }
                } else {
                    arch.smallBusy = 0;
                }
            }
        }
    }
    if (!rebuild) {
        return 0;
    }
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

// row counts the last completed replay reported (statsKernel)
MWHIP_RT int growTablesAfterReplay(mwhip_exec *exec)
{
    int rc = growTables(exec, [exec](uint32_t a) -> int64_t {
        return a < kMaxArchetypes ? exec->statsHost[kStatsPeaks + a] : -1;
    });
    if (rc != 0) return rc;
    return sortsOutgrown(exec);
}

#ifdef MADRONA_TRACING
// The step that just completed -> exec->traceLogs (the first 100 steps, like the
// reference's DeviceTracingManager, cuda_exec.cpp:204-257), with a nodeFinish
// record per kernel: its latest blockWait.
MWHIP_RT int collectDeviceTrace(mwhip_exec *exec)
{
    using mwGPU::DeviceEvent;
    using Log = mwGPU::DeviceTracing::DeviceLog;
    if (exec->deviceTracing == nullptr || exec->traceSteps >= 100u) return 0;
    int32_t count = 0;
    HIPCHK(hipMemcpy(&count, exec->deviceTracing, sizeof(count), hipMemcpyDeviceToHost));
    if (count <= 0) return 0;       // nothing logged, or the step overflowed
    count = std::min<int32_t>(count, (int32_t)mwGPU::DeviceTracing::maxLogSize);
    std::vector<Log> step((size_t)count);
    HIPCHK(hipMemcpy(step.data(),
        (const char *)exec->deviceTracing + offsetof(mwGPU::DeviceTracing, device_logs_),
        step.size() * sizeof(Log), hipMemcpyDeviceToHost));
    // (slots reserved for a kernel whose workgroups log nothing)
    step.erase(std::remove_if(step.begin(), step.end(), [](const Log &l) {
        return (uint32_t)l.event == mwGPU::DeviceTracing::unusedSlot;
    }), step.end());
    // a step begins with the calibration record: slot 0 of the log
    for (size_t i = 0; i < step.size(); i++) {
        step[i].padding = (uint32_t)i;
    }
    std::vector<Log> finish;
    for (const Log &l : step) {
        if (l.event == DeviceEvent::nodeStart) {
            if (finish.size() <= l.nodeID) finish.resize(l.nodeID + 1, Log {});
            Log f = l;      // (a kernel whose workgroups log nothing: zero length)
            f.event = DeviceEvent::nodeFinish;
            finish[l.nodeID] = f;
        }
    }
    for (const Log &l : step) {
        if (l.event == DeviceEvent::blockWait && l.nodeID < finish.size() &&
                l.cycleCount >= finish[l.nodeID].cycleCount) {
            Log &f = finish[l.nodeID];
            f.cycleCount = l.cycleCount;
            f.smID = l.smID;
            f.warpID = l.warpID;
            f.blockID = l.blockID;
        }
    }
    uint32_t next = (uint32_t)step.size();
    for (Log &f : finish) {
        if (f.event != DeviceEvent::nodeFinish) continue;
        f.padding = next++;
        step.push_back(f);
    }
    exec->traceLogs.insert(exec->traceLogs.end(), step.begin(), step.end());
    exec->traceSteps++;
    // (until the next traced graph starts its step)
    const int32_t off = -1;
    HIPCHK(hipMemcpy(exec->deviceTracing, &off, sizeof(off), hipMemcpyHostToDevice));
    return 0;
}

MWHIP_RT void writeDeviceTrace(mwhip_exec *exec)
{
    if (exec->traceLogs.empty()) return;
    const char *dir = getenv("MADRONA_MWHIP_TRACE_DIR");
    const std::string path = dir != nullptr ? std::string(dir) + "/" : "/tmp/";
    ::madrona::WriteToFile<mwGPU::DeviceTracing::DeviceLog>(
        exec->traceLogs.data(), exec->traceLogs.size(), path,
        "_madrona_device_tracing");
    std::string names;
    for (const std::string &n : exec->traceNames) {
        names += n + "\n";
    }
    ::madrona::WriteToFile((void *)names.data(), names.size(), path,
                           "_madrona_device_tracing_nodes");
}

#endif

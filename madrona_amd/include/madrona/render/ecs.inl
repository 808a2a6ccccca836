// RenderingSystem on the MI355X backend.  Behaviour: reference
// src/render/ecs_system.cpp in MADRONA_GPU_MODE (:51-98 Morton codes, :100-159
// instance records, :317-347 view records, :349-380 exported counts, :385-597
// registration and the task graph), re-stated against this backend's tables.
#pragma once

#include <madrona/mw_gpu/const.hpp>
#include <madrona/crash.hpp>

namespace madrona::render {

namespace RenderingSystem {

// per-world singleton: where the renderer wants the totals, and the aspect ratio
// view records are built with
struct RenderingSystemState {
    uint32_t *totalNumViews;
    uint32_t *totalNumInstances;
    uint32_t *voxels;
    float aspectRatio;
};

namespace detail {

// 30-bit Morton code of a position: x, y, z contribute 10 bits each, z highest
// in every triple.  As in the reference (ecs_system.cpp:51-83) the 10 bits are
// the LOW bits of the coordinate's IEEE-754 pattern (1024 is clamped to 1023):
// instances are ordered by it bit for bit the same, whatever one thinks of the
// choice of bits.
MADRONA_HD inline uint32_t spreadBits10(uint32_t v)
{
    v = v == 1024u ? 1023u : v;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

MADRONA_HD inline uint32_t mortonOf(math::Vector3 p)
{
    return (spreadBits10(__builtin_bit_cast(uint32_t, p.z)) << 2) |
           (spreadBits10(__builtin_bit_cast(uint32_t, p.y)) << 1) |
            spreadBits10(__builtin_bit_cast(uint32_t, p.x));
}

// resolution of the ray caster's square outputs, 0 when it is off
MADRONA_HD inline uint32_t raycastResolution()
{
    return mwGPU::GPUImplConsts::get().raycastOutputResolution;
}

// the render entity's record follows its owner; with the ray caster on, so does
// the world-space box of the object's geometry (TLBVH leaf)
MADRONA_HD inline InstanceData &writeInstance(Context &ctx, Entity render_entity,
                                              math::Vector3 pos, math::Quat rot,
                                              math::Diag3x3 scale,
                                              int32_t object_id)
{
    InstanceData &inst = ctx.get<InstanceData>(render_entity);
    inst.position = pos;
    inst.rotation = rot;
    inst.scale = scale;
    inst.worldIDX = ctx.worldID().idx;
    inst.objectID = object_id;

    const auto consts = mwGPU::GPUImplConsts::get();
    if (consts.raycastOutputResolution != 0u && consts.meshBVHsAddr != nullptr) {
        // one object-space root box per object id (CudaBatchRenderConfig)
        const math::AABB root = ((const math::AABB *)consts.meshBVHsAddr)[object_id];
        ctx.get<TLBVHNode>(render_entity).aabb = root.applyTRS(pos, rot, scale);
    }
    return inst;
}

}

// ---- systems ------------------------------------------------------------------
inline void instanceTransformUpdate(Context &ctx, Entity,
                                    const base::Position &pos,
                                    const base::Rotation &rot,
                                    const base::Scale &scale,
                                    const base::ObjectID &obj_id,
                                    const Renderable &renderable)
{
    if (renderable.renderEntity == Entity::none()) {
        return;
    }
    detail::writeInstance(ctx, renderable.renderEntity, pos, rot, scale,
                          obj_id.idx);
}

inline void instanceTransformUpdateWithMat(Context &ctx, Entity,
                                           const base::Position &pos,
                                           const base::Rotation &rot,
                                           const base::Scale &scale,
                                           const base::ObjectID &obj_id,
                                           const MaterialOverride &mat,
                                           const ColorOverride &color,
                                           const Renderable &renderable)
{
    if (renderable.renderEntity == Entity::none()) {
        return;
    }
    InstanceData &inst = detail::writeInstance(
        ctx, renderable.renderEntity, pos, rot, scale, obj_id.idx);
    inst.matID = mat.matID;
    inst.color = color.color;
}

inline void lightUpdate(Context &ctx, Entity, const base::Position &pos,
                        const LightDescDirection &dir,
                        const LightDescType &type,
                        const LightDescShadow &shadow,
                        const LightDescCutoffAngle &angle,
                        const LightDescIntensity &intensity,
                        const LightDescActive &active, LightCarrier &carrier)
{
    if (carrier.light == Entity::none()) {
        return;
    }
    LightDesc &light = ctx.get<LightDesc>(carrier.light);
    light.type = type.type;
    light.castShadow = shadow.castShadow;
    light.position = pos;
    light.direction = dir;
    light.cutoff = angle.cutoff;
    light.intensity = intensity.intensity;
    light.active = active.active;
}

// camera pose of a view + the output slot it renders into: the row of the
// viewing entity in its own (world-sorted) table, so outputs line up with the
// simulator's exported per-agent tensors
inline void viewTransformUpdate(Context &ctx, Entity e,
                                const base::Position &pos,
                                const base::Rotation &rot,
                                const RenderCamera &cam)
{
    PerspectiveCameraData &view =
        ctx.get<PerspectiveCameraData>(cam.cameraEntity);
    ctx.get<RenderOutputIndex>(cam.cameraEntity).index =
        (uint32_t)ctx.loc(e).row;

    view.position = pos + cam.cameraOffset;
    view.rotation = rot.inv();
    view.worldIDX = ctx.worldID().idx;
}

inline void mortonCodeUpdate(Context &ctx, Entity, const base::Position &pos,
                             const Renderable &renderable)
{
    if (renderable.renderEntity == Entity::none()) {
        return;
    }
    ctx.get<MortonCode>(renderable.renderEntity) = detail::mortonOf(pos);
}

// world 0 publishes the table sizes the renderer dispatches over
inline void exportCounts(Context &ctx, RenderingSystemState &state)
{
    if (ctx.worldID().idx != 0 || state.totalNumViews == nullptr) {
        return;
    }
    StateManager *mgr = ctx.getStateManager();
    *state.totalNumViews = mgr->getArchetypeNumRows<RenderCameraArchetype>();
    *state.totalNumInstances = mgr->getArchetypeNumRows<RenderableArchetype>();
}

// ---- registration / graph --------------------------------------------------------
MADRONA_HOST_API inline void registerTypes(ECSRegistry &registry,
                                           const RenderECSBridge *bridge)
{
    // (the body is compiled for the device too, although it only ever runs on
    // the host: the device pass has to see every registerComponent<T> so that
    // T's id variable exists in the simulator's code object)
#if MADRONA_ON_HOST
    if (bridge != nullptr) {
        FATAL("RenderingSystem: no Vulkan bridge on this backend, pass nullptr");
    }
#else
    (void)bridge;
#endif

    // render targets exist even with the ray caster off (4 bytes each); depth is
    // always produced, colour only in RGBD mode
    const uint32_t res = registry.raycastOutputResolution();
    uint32_t depth_bytes = res * res * 4u;
    uint32_t rgb_bytes = depth_bytes;
    if (depth_bytes == 0u) {
        rgb_bytes = depth_bytes = 4u;
    } else if (!registry.raycastRGBD()) {
        rgb_bytes = 4u;
    }

    registry.registerComponent<RenderCamera>();
    registry.registerComponent<Renderable>();
    registry.registerComponent<PerspectiveCameraData>();
    registry.registerComponent<InstanceData>();
    registry.registerComponent<MortonCode>();
    registry.registerComponent<MaterialOverride>();
    registry.registerComponent<ColorOverride>();
    registry.registerComponent<LightDesc>();
    registry.registerComponent<LightDescDirection>();
    registry.registerComponent<LightDescType>();
    registry.registerComponent<LightDescShadow>();
    registry.registerComponent<LightDescCutoffAngle>();
    registry.registerComponent<LightDescIntensity>();
    registry.registerComponent<LightDescActive>();
    registry.registerComponent<LightCarrier>();
    registry.registerComponent<RGBOutputBuffer>(rgb_bytes);
    registry.registerComponent<DepthOutputBuffer>(depth_bytes);
    registry.registerComponent<RenderOutputIndex>();
    registry.registerComponent<RenderOutputRef>();
    registry.registerComponent<TLBVHNode>();

    // (render targets are 8 * res^2 bytes a row: sized by the configuration's
    // promise when there is one)
    const CountT max_views = (CountT)registry.raycastMaxViewsPerWorld();
    registry.registerArchetype<RaycastOutputArchetype>(
        ComponentMetadataSelector<> {}, ArchetypeFlags::None, max_views);
    registry.registerArchetype<LightArchetype>();
    registry.registerArchetype<RenderCameraArchetype>(
        ComponentMetadataSelector<> {}, ArchetypeFlags::None, max_views);
    registry.registerArchetype<RenderableArchetype>();

    registry.registerSingleton<RenderingSystemState>();

    registry.setRenderLayout<RenderableArchetype, RenderCameraArchetype,
        LightArchetype, RaycastOutputArchetype, InstanceData, MortonCode,
        TLBVHNode, PerspectiveCameraData, LightDesc, RGBOutputBuffer,
        DepthOutputBuffer>();
}

MADRONA_HOST_API inline TaskGraphNodeID setupTasks(
    TaskGraphBuilder &builder, Span<const TaskGraphNodeID> deps,
    bool update_visual_properties)
{
    using namespace base;

    auto node = builder.addToGraph<ParallelForNode<Context,
        instanceTransformUpdate,
            Entity, Position, Rotation, Scale, ObjectID, Renderable>>(deps);

    if (update_visual_properties) {
        node = builder.addToGraph<ParallelForNode<Context,
            instanceTransformUpdateWithMat,
                Entity, Position, Rotation, Scale, ObjectID, MaterialOverride,
                ColorOverride, Renderable>>({node});
        node = builder.addToGraph<ParallelForNode<Context,
            lightUpdate,
                Entity, Position, LightDescDirection, LightDescType,
                LightDescShadow, LightDescCutoffAngle, LightDescIntensity,
                LightDescActive, LightCarrier>>({node});
    }

    node = builder.addToGraph<ParallelForNode<Context,
        viewTransformUpdate,
            Entity, Position, Rotation, RenderCamera>>({node});
    node = builder.addToGraph<ParallelForNode<Context,
        mortonCodeUpdate,
            Entity, Position, Renderable>>({node});

    // instances: order by Morton code, then group by world (stable: Morton
    // order survives inside each world) dropping the destroyed rows.  The
    // reference compacts the table before the Morton sort as well
    // (ecs_system.cpp:551-566); the table that leaves the three chains is the
    // same without it -- both sorts are stable, and a compaction keeps the
    // relative order of a world's rows, which is all that decides ties between
    // equal codes -- so that chain (two key passes and a gather of every column,
    // every step) is left out.
    // (MADRONA_MWHIP_RENDER_PRECOMPACT=1 builds the reference's graph, that
    // chain included: tests/test_render_prep_gpu.py compares the two tables
    // row for row under churn and Morton-code ties)
#if MADRONA_ON_HOST
    if (const char *pre = getenv("MADRONA_MWHIP_RENDER_PRECOMPACT");
            pre != nullptr && atoi(pre) != 0) {
        node = builder.addToGraph<
            CompactArchetypeNode<RenderableArchetype>>({node});
    }
#endif
    node = builder.addToGraph<
        SortArchetypeNode<RenderableArchetype, MortonCode>>({node});
    node = builder.addToGraph<ResetTmpAllocNode>({node});
    node = builder.addToGraph<CompactArchetypeNode<RenderableArchetype>>({node});

    // views: group by world, then global output-slot order (slots of one world
    // are consecutive rows of the viewers' table, so worlds stay grouped)
    node = builder.addToGraph<
        CompactArchetypeNode<RenderCameraArchetype>>({node});
    node = builder.addToGraph<
        SortArchetypeNode<RenderCameraArchetype, RenderOutputIndex>>({node});
    node = builder.addToGraph<ResetTmpAllocNode>({node});

    node = builder.addToGraph<CompactArchetypeNode<LightArchetype>>({node});

    return builder.addToGraph<ParallelForNode<Context,
        exportCounts, RenderingSystemState>>({node});
}

// ---- per-world / per-entity API ------------------------------------------------------
MADRONA_HD inline void init(Context &ctx, const RenderECSBridge *)
{
    RenderingSystemState &state = ctx.singleton<RenderingSystemState>();
    state.totalNumViews = nullptr;
    state.totalNumInstances = nullptr;
    state.voxels = nullptr;
    // square outputs (the reference takes width / height from its bridge)
    state.aspectRatio = 1.f;
}

MADRONA_HD inline uint32_t *getVoxelPtr(Context &ctx)
{
    return ctx.singleton<RenderingSystemState>().voxels;
}

MADRONA_HD inline void makeEntityRenderable(Context &ctx, Entity e)
{
    Entity render_entity = ctx.makeEntity<RenderableArchetype>();
    ctx.get<Renderable>(e).renderEntity = render_entity;

    InstanceData &inst = ctx.get<InstanceData>(render_entity);
    inst.matID = MaterialOverride::UseDefaultMaterial;
    inst.color = 0;
}

MADRONA_HD inline void disableEntityRenderable(Context &ctx, Entity e)
{
    ctx.get<Renderable>(e).renderEntity = Entity::none();
}

MADRONA_HD inline void attachEntityToView(Context &ctx, Entity e,
                                          float vfov_degrees, float z_near,
                                          const math::Vector3 &camera_offset)
{
    const float fov_scale =
        1.f / tanf(math::toRadians(vfov_degrees * 0.5f));

    Entity camera_entity = ctx.makeEntity<RenderCameraArchetype>();
    ctx.get<RenderCamera>(e) =
        RenderCamera { camera_entity, fov_scale, z_near, camera_offset };

    const float aspect = ctx.singleton<RenderingSystemState>().aspectRatio;
    PerspectiveCameraData &view =
        ctx.get<PerspectiveCameraData>(camera_entity);
    view.position = math::Vector3::zero();
    view.rotation = math::Quat { 1.f, 0.f, 0.f, 0.f };
    view.xScale = fov_scale / aspect;
    view.yScale = -fov_scale;
    view.zNear = z_near;
    view.worldIDX = ctx.worldID().idx;
    view.pad = 0;

    if (detail::raycastResolution() != 0u) {
        ctx.get<RenderOutputRef>(camera_entity).outputEntity =
            ctx.makeEntity<RaycastOutputArchetype>();
    }
}

MADRONA_HD inline void cleanupViewingEntity(Context &ctx, Entity e)
{
    ctx.destroyEntity(ctx.get<RenderCamera>(e).cameraEntity);
}

MADRONA_HD inline void cleanupRenderableEntity(Context &ctx, Entity e)
{
    ctx.destroyEntity(ctx.get<Renderable>(e).renderEntity);
}

MADRONA_HD inline void makeEntityLightCarrier(Context &ctx, Entity e)
{
    Entity light_entity = ctx.makeEntity<LightArchetype>();
    ctx.get<LightCarrier>(e).light = light_entity;

    LightDesc &light = ctx.get<LightDesc>(light_entity);
    light.type = ctx.get<LightDescType>(e).type;
    light.castShadow = ctx.get<LightDescShadow>(e).castShadow;
    light.position = ctx.get<base::Position>(e);
    light.direction = ctx.get<LightDescDirection>(e);
    light.cutoff = ctx.get<LightDescCutoffAngle>(e).cutoff;
    light.intensity = ctx.get<LightDescIntensity>(e).intensity;
    light.active = ctx.get<LightDescActive>(e).active;
}

}

}

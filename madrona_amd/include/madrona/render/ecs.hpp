// Render-prep side of the ECS (BASELINE config 5): the
// components a simulator attaches to what should be drawn / to what looks, the
// render-side archetypes they feed, and RenderingSystem -- the systems that
// copy transforms into InstanceData / PerspectiveCameraData every step and the
// sort chains that leave those tables grouped by world, instances in Morton
// order, views in output-slot order, ready for a batch renderer.
//
// API contract: reference include/madrona/render/ecs.hpp:7-211 (component and
// archetype names, field layouts -- a renderer reads these tables as raw
// memory --, RenderingSystem entry points) and src/render/ecs_system.cpp
// (behaviour in MADRONA_GPU_MODE: the render entity's own row IS the renderer's
// record; the reference's CPU mode appends to bridge buffers instead).
// There is no Vulkan bridge on this backend: `bridge` must be nullptr (the
// tables then live in executor memory like any other) -- an importing renderer
// is SURVEY 8f-1.  The ray caster's inputs are configured on the executor
// (CudaBatchRenderConfig, madrona/mw_gpu.hpp): output resolution, and one
// object-space root AABB per object id for the TLBVH leaf boxes.
#pragma once

#include <madrona/math.hpp>
#include <madrona/components.hpp>
#include <madrona/taskgraph_builder.hpp>

namespace madrona::render {

// ---- attached by the simulator ------------------------------------------------
// an entity that looks
struct RenderCamera {
    Entity cameraEntity;
    float fovScale;             // 1 / tan(fovy / 2)
    float zNear;
    math::Vector3 cameraOffset;
};

// an entity that is drawn
struct Renderable {
    Entity renderEntity;
};

struct LightCarrier {
    Entity light;
};

struct MaterialOverride {
    enum { UseDefaultMaterial = -1, UseOverrideColor = -2 };
    int32_t matID;
};

struct ColorOverride {
    uint32_t color;
};

// ---- what the renderer reads (layouts are part of the contract) -----------------
using MortonCode = uint32_t;

struct alignas(16) PerspectiveCameraData {
    math::Vector3 position;
    math::Quat rotation;
    float xScale;
    float yScale;
    float zNear;
    int32_t worldIDX;
    uint32_t pad;
};

struct alignas(16) InstanceData {
    math::Vector3 position;
    math::Quat rotation;
    math::Diag3x3 scale;
    int32_t matID;      // -1: the model's material, -2: `color` below
    int32_t objectID;
    int32_t worldIDX;
    uint32_t color;
};

static_assert(sizeof(PerspectiveCameraData) == 48);
static_assert(sizeof(InstanceData) == 64);

struct LightDesc {
    enum Type : bool { Directional = true, Spotlight = false };

    Type type;
    bool castShadow;
    math::Vector3 position;     // spotlight only
    math::Vector3 direction;
    float cutoff;               // spotlight only
    float intensity;
    bool active;
};

// per-field light inputs on the carrier entity
struct LightDescDirection : math::Vector3 {
    MADRONA_HD LightDescDirection(math::Vector3 v) : Vector3(v) {}
};
struct LightDescType { LightDesc::Type type; };
struct LightDescShadow { bool castShadow; };
struct LightDescCutoffAngle { float cutoff; };
struct LightDescIntensity { float intensity; };
struct LightDescActive { bool active; };

// render targets: registered with a run-time size (resolution^2 * 4 bytes)
struct RenderOutputBuffer { char buffer[1]; };
struct RGBOutputBuffer : RenderOutputBuffer {};
struct DepthOutputBuffer : RenderOutputBuffer {};

struct RenderOutputRef { Entity outputEntity; };
struct RenderOutputIndex { uint32_t index; };

// leaf box of the top-level BVH over a world's instances
struct alignas(16) TLBVHNode { math::AABB aabb; };

struct LightArchetype : public Archetype<LightDesc> {};

struct RenderableArchetype : public Archetype<
    InstanceData,
    MortonCode,     // instances are ordered by it inside a world (LBVH build)
    TLBVHNode
> {};

struct RenderCameraArchetype : public Archetype<
    PerspectiveCameraData,
    RenderOutputRef,
    RenderOutputIndex
> {};

struct RaycastOutputArchetype : public Archetype<
    RGBOutputBuffer,
    DepthOutputBuffer
> {};

struct RenderECSBridge;     // Vulkan interop: not available on this backend

namespace RenderingSystem {

MADRONA_HOST_API inline void registerTypes(ECSRegistry &registry,
                                           const RenderECSBridge *bridge);

MADRONA_HOST_API inline TaskGraphNodeID setupTasks(
    TaskGraphBuilder &builder, Span<const TaskGraphNodeID> deps,
    bool update_visual_properties = false);

MADRONA_HD inline void init(Context &ctx, const RenderECSBridge *bridge);

MADRONA_HD inline uint32_t *getVoxelPtr(Context &ctx);

MADRONA_HD inline void makeEntityRenderable(Context &ctx, Entity e);
MADRONA_HD inline void disableEntityRenderable(Context &ctx, Entity e);
MADRONA_HD inline void attachEntityToView(Context &ctx, Entity e,
                                          float vfov_degrees, float z_near,
                                          const math::Vector3 &camera_offset);
MADRONA_HD inline void cleanupViewingEntity(Context &ctx, Entity e);
MADRONA_HD inline void cleanupRenderableEntity(Context &ctx, Entity e);
MADRONA_HD inline void makeEntityLightCarrier(Context &ctx, Entity e);

}

}

#include "ecs.inl"

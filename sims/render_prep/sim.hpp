// Render-prep test simulator (BASELINE config 5's ECS side):
// worlds of drifting, appearing, disappearing and hidden renderable entities,
// two viewers and a lamp, stepped through madrona::render::RenderingSystem --
// instance / view / light records, Morton codes, and the sort chains of its
// task graph (reference src/render/ecs_system.cpp:486-597), including the only
// SortArchetypeNode over non-WorldID keys any reference system uses.  Compiled
// unchanged against the reference (CPU mode: records go to RenderECSBridge
// buffers) and against the HIP backend (GPU mode: records are the render
// entities' own rows).
#pragma once

#include <madrona/taskgraph_builder.hpp>
#include <madrona/custom_context.hpp>
#include <madrona/components.hpp>
#include <madrona/rand.hpp>
#include <madrona/render/ecs.hpp>

namespace renderprep {

using madrona::Entity;
using madrona::RNG;
using madrona::base::ObjectID;
using madrona::base::Position;
using madrona::base::Rotation;
using madrona::base::Scale;

namespace consts {
inline constexpr int32_t maxMovers = 96;
// population cap of an ordinary (not crowded) world: its entities fit the one
// block of 64 ids a world takes during construction
inline constexpr int32_t sparseMovers = 24;
inline constexpr int32_t numViewers = 2;
inline constexpr int32_t numObjects = 4;
inline constexpr float arena = 7.f;
}

enum class ExportID : uint32_t {
    Roster,
    // ray caster outputs, one row per view (HIP backend; the reference's CPU
    // mode has no ray caster)
    RGB,
    Depth,
    NumExports,
};

struct Drift {
    madrona::math::Vector3 v;
};

// the world's movers; hidden[i]: the mover exists but is not drawn
struct Roster {
    int32_t numMovers;
    int32_t numHidden;
    Entity movers[consts::maxMovers];
    int32_t hidden[consts::maxMovers];
};

struct RosterCounts {
    int32_t numMovers;
    int32_t numDrawn;
};

struct Mover : public madrona::Archetype<
    Position, Rotation, Scale, ObjectID, Drift,
    madrona::render::Renderable,
    madrona::render::MaterialOverride,
    madrona::render::ColorOverride
> {};

struct Viewer : public madrona::Archetype<
    Position, Rotation,
    madrona::render::RenderCamera
> {};

struct Lamp : public madrona::Archetype<
    Position,
    madrona::render::LightDescDirection,
    madrona::render::LightDescType,
    madrona::render::LightDescShadow,
    madrona::render::LightDescCutoffAngle,
    madrona::render::LightDescIntensity,
    madrona::render::LightDescActive,
    madrona::render::LightCarrier
> {};

class Engine;

struct Sim : public madrona::WorldBase {
    struct Config {
        uint32_t seed;
        uint32_t worldBase;
        // RenderingSystem::setupTasks(update_visual_properties)
        uint32_t updateVisuals;
        // crowded worlds: 70 .. 95 movers from the start (more instances than
        // the ray caster keeps in LDS per world)
        uint32_t dense;
        // movers stand on a coarse grid and do not drift: many of a world's
        // instances share a position, hence a Morton code (ties in the sort)
        uint32_t ties;
        // reference CPU backend: the buffers its systems append to
        const madrona::render::RenderECSBridge *bridge;
    };

    struct WorldInit {};

    static void registerTypes(madrona::ECSRegistry &registry,
                              const Config &cfg);
    static void setupTasks(madrona::TaskGraphManager &taskgraph_mgr,
                           const Config &cfg);

    Sim(Engine &ctx, const Config &cfg, const WorldInit &init);

    RNG rng;
    Entity viewers[consts::numViewers];
    Entity lamp;
    Entity sun;     // every third world: a second, directional light
    int32_t moverCap;
    uint32_t ties;
    uint32_t step;
};

class Engine : public madrona::CustomContext<Engine, Sim> {
public:
    using CustomContext::CustomContext;
};

}

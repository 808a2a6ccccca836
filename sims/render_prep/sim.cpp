#include "sim.hpp"

#ifdef MADRONA_GPU_MODE
#include <madrona/mw_gpu_entry.hpp>
#endif

using namespace madrona;
using namespace madrona::math;
using namespace madrona::render;

namespace renderprep {

void Sim::registerTypes(ECSRegistry &registry, const Config &cfg)
{
    base::registerTypes(registry);
    RenderingSystem::registerTypes(registry, cfg.bridge);

    registry.registerComponent<Drift>();
    registry.registerSingleton<Roster>();
    registry.registerSingleton<RosterCounts>();

    registry.registerArchetype<Mover>();
    registry.registerArchetype<Viewer>();
    registry.registerArchetype<Lamp>();

    registry.exportSingleton<RosterCounts>((uint32_t)ExportID::Roster);
#ifdef MADRONA_GPU_MODE
    registry.exportColumn<RaycastOutputArchetype, RGBOutputBuffer>(
        (uint32_t)ExportID::RGB);
    registry.exportColumn<RaycastOutputArchetype, DepthOutputBuffer>(
        (uint32_t)ExportID::Depth);
#endif
}

static inline float randInRange(RNG &rng, float lo, float hi)
{
    return lo + rng.sampleUniform() * (hi - lo);
}

static Entity makeMover(Engine &ctx, RNG &rng)
{
    Entity e = ctx.makeEntity<Mover>();
    ctx.get<Position>(e) = Vector3 {
        randInRange(rng, -consts::arena, consts::arena),
        randInRange(rng, -consts::arena, consts::arena),
        randInRange(rng, 0.f, 4.f),
    };
    ctx.get<Rotation>(e) = Quat { 1.f, 0.f, 0.f, 0.f };
    ctx.get<Scale>(e) = Diag3x3 {
        randInRange(rng, 0.5f, 2.f), randInRange(rng, 0.5f, 2.f),
        randInRange(rng, 0.5f, 2.f),
    };
    ctx.get<ObjectID>(e) = ObjectID { rng.sampleI32(0, consts::numObjects) };
    ctx.get<Drift>(e).v = Vector3 {
        randInRange(rng, -1.f, 1.f), randInRange(rng, -1.f, 1.f), 0.f,
    };
    if (ctx.data().ties != 0u) {
        // a 3 x 3 grid of standing places: equal positions, equal Morton codes
        ctx.get<Position>(e) = Vector3 {
            8.f * (float)rng.sampleI32(-1, 2), 8.f * (float)rng.sampleI32(-1, 2),
            0.f,
        };
        ctx.get<Drift>(e).v = Vector3::zero();
    }
    // every third mover overrides its colour, every fifth its material (with
    // an untextured one, or -- style 10 -- the textured material 7, which then
    // samples the mover's own uvs: all zero except on the wedge)
    int32_t style = rng.sampleI32(0, 15);
    ctx.get<MaterialOverride>(e).matID = style % 5 == 0 ? (style == 10 ? 7 : 3) :
        (style % 3 == 0 ? (int32_t)MaterialOverride::UseOverrideColor :
                          (int32_t)MaterialOverride::UseDefaultMaterial);
    ctx.get<ColorOverride>(e).color = 0xFF000000u | (uint32_t)(style * 1118481);
    RenderingSystem::makeEntityRenderable(ctx, e);
    return e;
}

// x, y drift and bounce off the arena; a fixed small rotation per step (no
// trigonometry at run time: both backends must agree to the bit)
inline void driftSystem(Engine &, Position &pos, Rotation &rot, Drift &drift)
{
    Vector3 p = pos;
    p.x += drift.v.x * 0.125f;
    p.y += drift.v.y * 0.125f;
    if (p.x > consts::arena || p.x < -consts::arena) drift.v.x = -drift.v.x;
    if (p.y > consts::arena || p.y < -consts::arena) drift.v.y = -drift.v.y;
    pos = p;

    const Quat spin = Quat { 1.f, 0.01f, 0.02f, 0.03f }.normalize();
    rot = (rot * spin).normalize();
}

// one lane per world: movers appear, disappear, hide and show
inline void churnSystem(Engine &ctx, Roster &roster)
{
    Sim &sim = ctx.data();
    RNG rng = sim.rng;

    int32_t action = rng.sampleI32(0, 8);
    if (action <= 1 && roster.numMovers > 2) {
        // a mover leaves: its render entity first, as the API asks
        int32_t victim = rng.sampleI32(0, roster.numMovers);
        Entity e = roster.movers[victim];
        if (roster.hidden[victim] == 0) {
            RenderingSystem::cleanupRenderableEntity(ctx, e);
        } else {
            roster.numHidden -= 1;
        }
        ctx.destroyEntity(e);
        roster.movers[victim] = roster.movers[roster.numMovers - 1];
        roster.hidden[victim] = roster.hidden[roster.numMovers - 1];
        roster.numMovers -= 1;
    } else if (action <= 4 && roster.numMovers < sim.moverCap) {
        roster.movers[roster.numMovers] = makeMover(ctx, rng);
        roster.hidden[roster.numMovers] = 0;
        roster.numMovers += 1;
    } else if (action == 5 && roster.numMovers > 0) {
        // toggle visibility of one mover
        int32_t pick = rng.sampleI32(0, roster.numMovers);
        Entity e = roster.movers[pick];
        if (roster.hidden[pick] == 0) {
            RenderingSystem::cleanupRenderableEntity(ctx, e);
            RenderingSystem::disableEntityRenderable(ctx, e);
            roster.hidden[pick] = 1;
            roster.numHidden += 1;
        } else {
            RenderingSystem::makeEntityRenderable(ctx, e);
            roster.hidden[pick] = 0;
            roster.numHidden -= 1;
        }
    }

    sim.rng = rng;
    sim.step += 1;

    RosterCounts &counts = ctx.singleton<RosterCounts>();
    counts.numMovers = roster.numMovers;
    counts.numDrawn = roster.numMovers - roster.numHidden;
}

// the viewers circle slowly, the lamp flickers
inline void viewerSystem(Engine &, Position &pos, Rotation &rot,
                         RenderCamera &)
{
    pos.x = pos.x * 0.99f + 0.05f;
    const Quat yaw = Quat { 1.f, 0.f, 0.f, 0.02f }.normalize();
    rot = (rot * yaw).normalize();
}

inline void lampSystem(Engine &ctx, Position &pos, LightDescIntensity &power,
                       LightDescActive &active)
{
    uint32_t t = ctx.data().step;
    pos.z = 6.f + (float)(t % 7u) * 0.25f;
    power.intensity = 1.f + (float)(t % 5u) * 0.125f;
    active.active = t % 11u != 0u;
}

void Sim::setupTasks(TaskGraphManager &taskgraph_mgr, const Config &cfg)
{
    TaskGraphBuilder &builder = taskgraph_mgr.init(0);

    auto drift = builder.addToGraph<ParallelForNode<Engine,
        driftSystem, Position, Rotation, Drift>>({});
    auto churn = builder.addToGraph<ParallelForNode<Engine,
        churnSystem, Roster>>({drift});
    auto viewers = builder.addToGraph<ParallelForNode<Engine,
        viewerSystem, Position, Rotation, RenderCamera>>({churn});
    auto lamp = builder.addToGraph<ParallelForNode<Engine,
        lampSystem, Position, LightDescIntensity, LightDescActive>>({viewers});
    auto compact = builder.addToGraph<CompactArchetypeNode<Mover>>({lamp});

    RenderingSystem::setupTasks(builder, {compact}, cfg.updateVisuals != 0);
}

Sim::Sim(Engine &ctx, const Config &cfg, const WorldInit &)
    : WorldBase(ctx)
{
    uint32_t global_world = cfg.worldBase + (uint32_t)ctx.worldID().idx;
    RNG init_rng(rand::split_i(rand::initKey(cfg.seed), global_world));
    step = 0;

    moverCap = cfg.dense != 0u ? consts::maxMovers : consts::sparseMovers;
    ties = cfg.ties;

    RenderingSystem::init(ctx, cfg.bridge);

    Roster &roster = ctx.singleton<Roster>();
    roster.numMovers = 0;
    roster.numHidden = 0;
    int32_t initial = cfg.dense != 0u ? 70 + (int32_t)(global_world * 7u % 26u) :
                                        3 + (int32_t)(global_world * 5u % 14u);
    for (int32_t i = 0; i < initial; i++) {
        roster.movers[i] = makeMover(ctx, init_rng);
        roster.hidden[i] = 0;
        roster.numMovers += 1;
    }
    RosterCounts &counts = ctx.singleton<RosterCounts>();
    counts.numMovers = roster.numMovers;
    counts.numDrawn = roster.numMovers;

    for (int32_t i = 0; i < consts::numViewers; i++) {
        Entity v = ctx.makeEntity<Viewer>();
        ctx.get<Position>(v) = Vector3 {
            randInRange(init_rng, -4.f, 4.f), randInRange(init_rng, -4.f, 4.f),
            1.5f,
        };
        ctx.get<Rotation>(v) = Quat { 1.f, 0.f, 0.f, 0.f };
        RenderingSystem::attachEntityToView(ctx, v, 90.f, 0.125f,
                                            Vector3 { 0.f, 0.f, 0.25f * (float)i });
        viewers[i] = v;
    }

    lamp = ctx.makeEntity<Lamp>();
    ctx.get<Position>(lamp) = Vector3 { 0.f, 0.f, 6.f };
    ctx.get<LightDescDirection>(lamp) =
        LightDescDirection(Vector3 { 0.25f, 0.5f, -1.f });
    ctx.get<LightDescType>(lamp).type = LightDesc::Spotlight;
    ctx.get<LightDescShadow>(lamp).castShadow = global_world % 2u == 0u;
    ctx.get<LightDescCutoffAngle>(lamp).cutoff = 0.75f;
    ctx.get<LightDescIntensity>(lamp).intensity = 1.f;
    ctx.get<LightDescActive>(lamp).active = true;
    RenderingSystem::makeEntityLightCarrier(ctx, lamp);

    sun = Entity::none();
    if (global_world % 3u == 0u) {
        sun = ctx.makeEntity<Lamp>();
        ctx.get<Position>(sun) = Vector3 { 0.f, 0.f, 50.f };
        ctx.get<LightDescDirection>(sun) =
            LightDescDirection(Vector3 { -0.5f, 0.25f, -0.75f });
        ctx.get<LightDescType>(sun).type = LightDesc::Directional;
        ctx.get<LightDescShadow>(sun).castShadow = global_world % 2u == 1u;
        ctx.get<LightDescCutoffAngle>(sun).cutoff = -1.f;
        ctx.get<LightDescIntensity>(sun).intensity = 1.f;
        ctx.get<LightDescActive>(sun).active = true;
        RenderingSystem::makeEntityLightCarrier(ctx, sun);
    }

    // crowded worlds are also lit from all sides: more lights than the ray caster
    // keeps next to the CU
    if (cfg.dense != 0u) {
        for (int32_t i = 0; i < 10; i++) {
            Entity extra = ctx.makeEntity<Lamp>();
            ctx.get<Position>(extra) = Vector3 { (float)i, 0.f, 20.f };
            ctx.get<LightDescDirection>(extra) = LightDescDirection(Vector3 {
                0.125f * (float)(i - 5), 0.0625f * (float)((i * 3) % 7 - 3),
                -0.25f - 0.0625f * (float)i });
            ctx.get<LightDescType>(extra).type = LightDesc::Directional;
            ctx.get<LightDescShadow>(extra).castShadow = i == 9;
            ctx.get<LightDescCutoffAngle>(extra).cutoff = -1.f;
            ctx.get<LightDescIntensity>(extra).intensity = 1.f;
            ctx.get<LightDescActive>(extra).active = true;
            RenderingSystem::makeEntityLightCarrier(ctx, extra);
        }
    }

    rng = init_rng;
}

#ifdef MADRONA_GPU_MODE
MADRONA_BUILD_MWGPU_ENTRY(Engine, Sim, Sim::Config, Sim::WorldInit);
#endif

}

#include "sim.hpp"

#ifdef MADRONA_GPU_MODE
#include <madrona/mw_gpu_entry.hpp>
#include <madrona/mw_gpu/host_print.hpp>
#endif

using namespace madrona;

namespace sortstress {

void Sim::registerTypes(ECSRegistry &registry, const Config &)
{
    registry.registerComponent<Tag8>();
    registry.registerComponent<Half>();
    registry.registerComponent<Key>();
    registry.registerComponent<Pair>();
    registry.registerComponent<Vec3>();
    registry.registerComponent<Quad>();
    registry.registerComponent<Blob20>();
    registry.registerComponent<Wide>();
    registry.registerSingleton<Churn>();

    registry.registerArchetype<Item>();
    registry.registerArchetype<Scratch>();

    registry.exportSingleton<Churn>((uint32_t)ExportID::Churn);
}

static inline void fillItem(Engine &ctx, Entity e, RNG &rng)
{
    uint32_t bits = (uint32_t)rng.sampleI32(0, 0x7FFFFFFF);
    uint32_t id_bits = ctx.data().mixIds != 0 ? (uint32_t)e.id : 0u;

    ctx.get<Tag8>(e).v = (uint8_t)(bits & 0xFF);
    ctx.get<Half>(e).v = (uint16_t)(bits >> 8);
    ctx.get<Key>(e).v = bits;
    ctx.get<Pair>(e).v = ((uint64_t)bits << 32) | (uint64_t)id_bits;

    Vec3 &v3 = ctx.get<Vec3>(e);
    for (int i = 0; i < 3; i++) v3.v[i] = rng.sampleUniform();

    Quad &q = ctx.get<Quad>(e);
    for (int i = 0; i < 4; i++) q.v[i] = (float)i + v3.v[0];

    Blob20 &b = ctx.get<Blob20>(e);
    for (int i = 0; i < 5; i++) b.v[i] = bits * (uint32_t)(i + 1);

    Wide &w = ctx.get<Wide>(e);
    for (int i = 0; i < 60; i++) w.v[i] = (float)(bits & 0xFFFF) + (float)i;
}

inline void churnSystem(Engine &ctx, Churn &churn)
{
    Sim &sim = ctx.data();
    RNG &rng = sim.rng;

    if (sim.burst == 1u || sim.burst == 3u) {
        // burst mode: no churn; the bursting worlds fill up at steps 3 and 30
        // and empty again at steps 13 and 40
        if (sim.burst == 1u && (churn.step == 2u || churn.step == 29u)) {
            while (sim.numItems < 31) {
                Entity e = ctx.makeEntity<Item>();
                fillItem(ctx, e, rng);
                sim.items[sim.numItems++] = e;
            }
        }
        if (sim.burst == 1u && (churn.step == 12u || churn.step == 39u)) {
            while (sim.numItems > 1) {
                ctx.destroyEntity(sim.items[--sim.numItems]);
            }
        }
        churn.step += 1;
        churn.numItems = (uint32_t)sim.numItems;
        return;
    }

    int32_t num_destroy = rng.sampleI32(0, consts::maxChurn + 1);
    if (sim.rampUp != 0 && sim.numItems < consts::maxItems) {
        num_destroy = 0;
    }
    for (int32_t i = 0; i < num_destroy && sim.numItems > 0; i++) {
        int32_t victim = rng.sampleI32(0, sim.numItems);
        ctx.destroyEntity(sim.items[victim]);
        sim.items[victim] = sim.items[sim.numItems - 1];
        sim.numItems -= 1;
    }

    int32_t num_create = rng.sampleI32(0, consts::maxChurn + 1);
    for (int32_t i = 0; i < num_create && sim.numItems < consts::maxItems; i++) {
        Entity e = ctx.makeEntity<Item>();
        fillItem(ctx, e, rng);
        sim.items[sim.numItems++] = e;
    }

    int32_t num_scratch = rng.sampleI32(0, consts::maxScratch + 1);
    for (int32_t i = 0; i < num_scratch; i++) {
        Loc loc = ctx.makeTemporary<Scratch>();
        ctx.get<Key>(loc).v = churn.step * 1000u + (uint32_t)i;
        Vec3 &v = ctx.get<Vec3>(loc);
        v.v[0] = rng.sampleUniform();
        v.v[1] = (float)i;
        v.v[2] = (float)churn.step;
    }

    churn.step += 1;
    churn.numItems = (uint32_t)sim.numItems;

#ifdef MADRONA_GPU_MODE
    // device-side printf through the executor's message ring (every argument
    // type the channel carries); world 3 speaks once, at its third step
    if (ctx.worldID().idx == 3 && churn.step == 3u) {
        mwGPU::HostPrint::log(
            "sort_stress: world {} step {} items {} mean {} key {} ptr {}",
            ctx.worldID().idx, churn.step, (int64_t)sim.numItems,
            (float)sim.numItems * 0.5f, (uint64_t)0x123456789ABCull,
            (void *)&sim);
    }
    if (sim.chatty != 0 && churn.step == 2u) {
        mwGPU::HostPrint::log("sort_stress: chatty world {}", ctx.worldID().idx);
    }
#endif
}

inline void touchSystem(Engine &ctx, Entity e, Key &key, Vec3 &v3, Wide &wide)
{
    uint32_t id_bits = ctx.data().mixIds != 0 ? (uint32_t)e.id : 0u;
    key.v = key.v * 1664525u + 1013904223u + id_bits;
    v3.v[1] = v3.v[1] * 0.5f + 0.25f;
    wide.v[(key.v >> 8) % 60] += 1.f;
}

inline void scratchSystem(Engine &ctx, Key &key, Vec3 &v3)
{
    v3.v[2] = v3.v[2] + (float)ctx.worldID().idx + (float)(key.v & 7u);
}

// The same per-row update twice: as an items_per_invocation = 4
// CustomParallelForNode (Fn(WorldID *, Cs *..., count), reference device
// taskgraph.inl:229-266 -- that node type only exists in the reference's
// device headers) and, for the CPU oracle, as an ordinary ParallelForNode.
static inline void scaleOne(Half &h, Pair &p)
{
    h.v = (uint16_t)(h.v * 3u + 1u);
    p.v += (uint64_t)h.v << 8;
}

#ifdef MADRONA_GPU_MODE
inline void scaleBatch(WorldID *worlds, Half *halves, Pair *pairs,
                       int32_t count)
{
    for (int32_t i = 0; i < count; i++) {
        if (worlds[i].idx != -1) {
            scaleOne(halves[i], pairs[i]);
        }
    }
}
#else
inline void scaleSystem(Engine &, Half &h, Pair &p)
{
    scaleOne(h, p);
}
#endif

// Works in per-step scratch memory (Context::tmpAlloc; freed by
// ResetTmpAllocNode): fills a buffer, reads it back in another order.
inline void scratchSumSystem(Engine &ctx, Churn &churn)
{
    constexpr int32_t n = 24;
    uint32_t *buf = (uint32_t *)ctx.tmpAlloc(sizeof(uint32_t) * n);
    for (int32_t i = 0; i < n; i++) {
        buf[i] = churn.step * 2654435761u + (uint32_t)i * 40503u +
            churn.numItems;
    }
    uint32_t sum = 0;
    for (int32_t i = n - 1; i >= 0; i--) {
        sum = sum * 31u + buf[i];
    }
    churn.scratchSum = sum;
}

// Three nodes that name the same dependency (Config::siblings).  The second
// reads what the first writes: right only behind it.  The reference runs nodes
// in registration order whatever they declared; this backend runs nodes with
// equal dependencies in one launch UNLESS their signatures clash (a non-const
// reference to a component the other names, on a shared table) -- the first two
// must stay apart, the last two may share a launch.
inline void siblingWriteSystem(Engine &, Quad &q)
{
    q.v[0] = q.v[0] * 0.5f + 1.f;
}

inline void siblingReadSystem(Engine &, const Quad &q, Vec3 &v)
{
    v.v[2] = q.v[0] + q.v[1];
}

inline void siblingOtherSystem(Engine &, Blob20 &b)
{
    b.v[4] += 3u;
}

void Sim::setupTasks(TaskGraphManager &taskgraph_mgr, const Config &cfg)
{
    TaskGraphBuilder &builder = taskgraph_mgr.init(TaskGraphID::Step);

    auto clear_tmp = builder.addToGraph<ClearTmpNode<Scratch>>({});

    auto churn_sys = builder.addToGraph<ParallelForNode<Engine,
        churnSystem, Churn>>({clear_tmp});

    auto compact = builder.addToGraph<CompactArchetypeNode<Item>>({churn_sys});

#ifdef MADRONA_GPU_MODE
    // as the reference's XPBD graph does for Contact temporaries
    // (src/physics/xpbd.cpp:1107-1113): group temporaries by world
    auto sort_tmp = builder.addToGraph<
        SortArchetypeNode<Scratch, WorldID>>({compact});
    auto post_sort = builder.addToGraph<ResetTmpAllocNode>({sort_tmp});
#else
    auto post_sort = compact;
#endif

    auto touch_sys = builder.addToGraph<ParallelForNode<Engine,
        touchSystem, Entity, Key, Vec3, Wide>>({post_sort});

    auto scratch_sys = builder.addToGraph<ParallelForNode<Engine,
        scratchSystem, Key, Vec3>>({touch_sys});

#ifdef MADRONA_GPU_MODE
    auto scale_sys = builder.addToGraph<CustomParallelForNode<Engine,
        scaleBatch, 1, 4, Half, Pair>>({scratch_sys});
#else
    auto scale_sys = builder.addToGraph<ParallelForNode<Engine,
        scaleSystem, Half, Pair>>({scratch_sys});
#endif

    auto sum_sys = builder.addToGraph<ParallelForNode<Engine,
        scratchSumSystem, Churn>>({scale_sys});
    auto reset_tmp = builder.addToGraph<ResetTmpAllocNode>({sum_sys});
    if (cfg.siblings != 0) {
        builder.addToGraph<ParallelForNode<Engine,
            siblingWriteSystem, Quad>>({reset_tmp});
        builder.addToGraph<ParallelForNode<Engine,
            siblingReadSystem, Quad, Vec3>>({reset_tmp});
        builder.addToGraph<ParallelForNode<Engine,
            siblingOtherSystem, Blob20>>({reset_tmp});
    }

#ifdef MADRONA_GPU_MODE
    {
        TaskGraphBuilder &b = taskgraph_mgr.init(TaskGraphID::ChurnOnly);
        auto clear = b.addToGraph<ClearTmpNode<Scratch>>({});
        b.addToGraph<ParallelForNode<Engine, churnSystem, Churn>>({clear});
    }
    {
        TaskGraphBuilder &b = taskgraph_mgr.init(TaskGraphID::SortByKey);
        b.addToGraph<SortArchetypeNode<Item, Key>>({});
    }
    {
        TaskGraphBuilder &b = taskgraph_mgr.init(TaskGraphID::CompactOnly);
        auto c = b.addToGraph<CompactArchetypeNode<Item>>({});
        b.addToGraph<SortArchetypeNode<Scratch, WorldID>>({c});
    }
#endif
}

Sim::Sim(Engine &ctx, const Config &cfg, const WorldInit &)
    : WorldBase(ctx)
{
    uint32_t global_world = cfg.worldBase + (uint32_t)ctx.worldID().idx;
    rng = RNG(rand::split_i(rand::initKey(cfg.seed), global_world));
    mixIds = cfg.coldStart == 0 ? 1u : 0u;
    rampUp = cfg.rampUp;
    chatty = cfg.chatty;
    numItems = 0;

    Churn &churn = ctx.singleton<Churn>();
    churn.step = 0;
    churn.numItems = 0;
    churn.scratchSum = 0;

    // ragged start; with coldStart some worlds begin empty
    int32_t initial = cfg.coldStart != 0 ?
        (int32_t)((global_world * 7u) % (uint32_t)consts::maxItems) :
        1 + (int32_t)((global_world * 7u) % (uint32_t)(consts::maxItems - 1));
    if (cfg.rampUp != 0 || cfg.burst != 0) {
        initial = 1;
    }
    burst = 0;
    if (cfg.burst != 0) {
        burst = global_world < 100u ? 1u :
            (global_world >= 200u && global_world < 264u) ? 2u : 3u;
    }
    for (int32_t i = 0; i < initial; i++) {
        Entity e = ctx.makeEntity<Item>();
        fillItem(ctx, e, rng);
        items[numItems++] = e;
    }
}

#ifdef MADRONA_GPU_MODE
MADRONA_BUILD_MWGPU_ENTRY(Engine, Sim, Sim::Config, Sim::WorldInit);
#endif

}

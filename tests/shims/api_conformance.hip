// TEST INFRASTRUCTURE.  API-conformance translation unit: includes every header
// of the overlay that the DEVICE side of a real Madrona simulator includes
// (madrona_escape_room / gpu_hideseek style sim.hpp + sim.cpp: taskgraph
// builder, custom context, components, math, rand, physics, the mwGPU
// helpers of SURVEY.md 8b) and touches the symbols such code names, so that a
// missing or mis-declared one is a compile error here rather than in a user's
// build.  Compiled for gfx950 + host by madrona_amd/Makefile; the simulator in
// it runs in tests/test_api_conformance.py (-m gpu part).
#include <madrona/mwhip/user_prelude.hpp>
#include <cstdio>
#include <string>
#pragma clang force_cuda_host_device begin
#include <madrona/taskgraph_builder.hpp>
#include <madrona/custom_context.hpp>
#include <madrona/components.hpp>
#include <madrona/math.hpp>
#include <madrona/rand.hpp>
#include <madrona/physics.hpp>
#include <madrona/memory.hpp>
#include <madrona/sync.hpp>
#include <madrona/inline_array.hpp>
#include <madrona/mw_gpu/const.hpp>
#include <madrona/mw_gpu/host_print.hpp>
#include <madrona/mw_gpu_entry.hpp>

using namespace madrona;

namespace conformance {

struct Counter {
    uint32_t hits;
    float sum;
    uint32_t locked;
    uint32_t scratch;
};

struct Thing : public Archetype<Counter> {};

class Engine;

struct Sim : public WorldBase {
    struct Config { uint32_t unused; };
    struct WorldInit { uint8_t unused; };

    static void registerTypes(ECSRegistry &registry, const Config &)
    {
        registry.registerComponent<Counter>();
        registry.registerArchetype<Thing>();
        registry.exportColumn<Thing, Counter>(0);
    }

    static void setupTasks(TaskGraphManager &mgr, const Config &);

    Sim(Engine &ctx, const Config &, const WorldInit &);

    Entity thing;
};

class Engine : public CustomContext<Engine, Sim> {
public:
    using CustomContext::CustomContext;
};

// every mwGPU:: / sync / container symbol SURVEY 8b lists, used the way
// src/render/ecs_system.cpp and the reference's simulators use them
inline void touchSystem(Engine &ctx, Counter &c)
{
    // sync.hpp
    AtomicU32Ref hits(c.hits);
    hits.fetch_add_relaxed(1u);
    AtomicFloatRef sum(c.sum);
    sum.fetch_add<sync::relaxed>(0.5f);
    AtomicU32Ref lock_word(c.locked);
    uint32_t expected = 0;
    if (lock_word.compare_exchange_weak<sync::acquire, sync::relaxed>(
            expected, 1u)) {
        lock_word.store<sync::release>(0u);
    }
    AtomicI32 local_counter { 0 };
    local_counter.fetch_add_relaxed(2);
    SpinLock lock;
    if (lock.tryLock()) {
        lock.unlock();
    }

    // memory.hpp: TmpAllocator / HostAllocator
    uint32_t *tmp = (uint32_t *)mwGPU::TmpAllocator::get().alloc(64);
    tmp[0] = (uint32_t)local_counter.load_relaxed();
    c.scratch = tmp[0] +
        (uint32_t)mwGPU::getHostAllocator()->roundUpAlloc(1) / 256u;

    // mw_gpu/const.hpp
    auto consts = mwGPU::GPUImplConsts::get();
    if (consts.stateManagerAddr != (void *)mwGPU::getStateManager() ||
            consts.numWorlds == 0u) {
        c.scratch = 0xDEADu;
    }

    // inline_array.hpp
    InlineArray<int32_t, 4> small;
    small.push_back(3);
    small.emplace_back(4);
    c.scratch += (uint32_t)small.size();

    // mw_gpu/host_print.hpp
    if (ctx.worldID().idx == 0 && c.hits == 1u) {
        mwGPU::HostPrint::log("conformance: world {} hits {} sum {}",
                              ctx.worldID().idx, c.hits, c.sum);
    }
}

void Sim::setupTasks(TaskGraphManager &mgr, const Config &)
{
    TaskGraphBuilder &builder = mgr.init(0);
    auto touch = builder.addToGraph<ParallelForNode<Engine, touchSystem,
                                                    Counter>>({});
    auto sort = builder.addToGraph<SortArchetypeNode<Thing, WorldID>>({touch});
    auto compact = builder.addToGraph<CompactArchetypeNode<Thing>>({sort});
    auto recycle = builder.addToGraph<RecycleEntitiesNode>({compact});
    builder.addToGraph<ResetTmpAllocNode>({recycle});
}

Sim::Sim(Engine &ctx, const Config &, const WorldInit &)
    : WorldBase(ctx)
{
    thing = ctx.makeEntity<Thing>();
    ctx.get<Counter>(thing) = Counter { 0, 0.f, 0, 0 };
}

MADRONA_BUILD_MWGPU_ENTRY(Engine, Sim, Sim::Config, Sim::WorldInit);

}

// madrona::mwhip::rowChain on plain objects (runs on the host, CPU test): each
// system gets the components of the node's list its signature names, by
// reference, const reference or value, first system first
namespace conformance_chain {
struct Ctx { int order; };
struct Open { int v; };
struct Props { int v; };
struct Pos { int v; };
inline void openSystem(Ctx &ctx, Open &open, const Props &props)
{
    open.v = props.v + 1;
    ctx.order = ctx.order * 10 + 1;
}
inline void moveSystem(Ctx &ctx, Pos &pos, Open open)
{
    pos.v = open.v * 2;
    ctx.order = ctx.order * 10 + 2;
}
inline constexpr auto chained = madrona::mwhip::rowChain<
    openSystem, moveSystem, Ctx, Open, Props, Pos>;
}
#pragma clang force_cuda_host_device end

extern "C" __attribute__((visibility("default")))
int conf_row_chain(char *name_out, int32_t name_cap)
{
    using namespace conformance_chain;
    Ctx ctx { 0 };
    Open open { -1 };
    Props props { 20 };
    Pos pos { -1 };
    chained(ctx, open, props, pos);
    if (open.v != 21 || pos.v != 42 || props.v != 20 || ctx.order != 12) {
        return 1;
    }
    std::string name = madrona::mwhip::systemName<chained>();
    snprintf(name_out, (size_t)name_cap, "%s", name.c_str());
    return 0;
}

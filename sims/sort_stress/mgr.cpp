#include "sort_stress/sim.hpp"

struct SimTraits;
#include "common/sim_c_api.h"

#include <vector>
#include <string>

struct SimTraits {
    using Sim = sortstress::Sim;
    using Engine = sortstress::Engine;

    static constexpr uint32_t numExports =
        (uint32_t)sortstress::ExportID::NumExports;
    static constexpr uint32_t numTaskGraphs =
        (uint32_t)sortstress::TaskGraphID::NumTaskGraphs;
    // sim_step replays task graph 0 only (the others are test probes)
    static constexpr bool stepIsTaskGraph0 = true;

    static Sim::Config makeConfig(const SimCreateArgs &args)
    {
        return Sim::Config { args.seed, args.world_base, args.flags & 1u,
                             (args.flags >> 1) & 1u,
                             (args.flags >> 2) & 1u,
                             (args.flags >> 3) & 1u,
                             (args.flags >> 4) & 1u };
    }

    static void makeInits(const SimCreateArgs &, Sim::WorldInit *) {}

    template <typename T>
    static void describeTensors(T &out, uint32_t num_worlds);
    template <typename T>
    static void describeColumns(T &cols);
};

#include "common/mgr_impl.inl"

template <typename T>
void SimTraits::describeTensors(T &out, uint32_t num_worlds)
{
    out.push_back({ "churn", SIM_I32, { (int64_t)num_worlds, 3 },
                    (uint32_t)sortstress::ExportID::Churn });
}

template <typename T>
void SimTraits::describeColumns(T &cols)
{
    using namespace sortstress;
    using madrona::Entity;
    using madrona::WorldID;

    cols.template add<Item, Entity>("Item.Entity", false);
    cols.template add<Item, Tag8>("Item.Tag8", false);
    cols.template add<Item, Half>("Item.Half", false);
    cols.template add<Item, Key>("Item.Key", false);
    cols.template add<Item, Pair>("Item.Pair", false);
    cols.template add<Item, Vec3>("Item.Vec3", true);
    cols.template add<Item, Quad>("Item.Quad", true);
    cols.template add<Item, Blob20>("Item.Blob20", false);
    cols.template add<Item, Wide>("Item.Wide", true);
#ifndef SIM_BACKEND_REF_CPU
    // (the global tables of the HIP backend carry the world id as column 1;
    // the reference CPU backend keeps one table per world)
    cols.template add<Item, WorldID>("Item.WorldID", false);
#endif
    cols.template add<Scratch, Key>("Scratch.Key", false);
    cols.template add<Scratch, Vec3>("Scratch.Vec3", true);
}

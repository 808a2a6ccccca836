#include "cartpole/sim.hpp"

struct SimTraits;
#include "common/sim_c_api.h"

#include <vector>
#include <string>

namespace simmgr { struct TensorDesc; struct ColumnList; }

struct SimTraits {
    using Sim = cartpole::Sim;
    using Engine = cartpole::Engine;

    static constexpr uint32_t numExports =
        (uint32_t)cartpole::ExportID::NumExports;
    static constexpr uint32_t numTaskGraphs = 1;

    static Sim::Config makeConfig(const SimCreateArgs &args)
    {
        return Sim::Config { args.seed, args.world_base };
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
    using cartpole::ExportID;
    int64_t W = num_worlds;
    out.push_back({ "state", SIM_F32, { W, 4 }, (uint32_t)ExportID::State });
    out.push_back({ "action", SIM_I32, { W, 1 }, (uint32_t)ExportID::Action });
    out.push_back({ "reward", SIM_F32, { W, 1 }, (uint32_t)ExportID::Reward });
    out.push_back({ "done", SIM_I32, { W, 1 }, (uint32_t)ExportID::Done });
}

template <typename T>
void SimTraits::describeColumns(T &cols)
{
    using namespace cartpole;
    cols.template add<Cart, madrona::Entity>("Cart.Entity", false);
    cols.template add<Cart, CartState>("Cart.CartState", true);
    cols.template add<Cart, Action>("Cart.Action", false);
    cols.template add<Cart, Reward>("Cart.Reward", true);
    cols.template add<Cart, Done>("Cart.Done", false);
    cols.template add<Cart, EpisodeInfo>("Cart.EpisodeInfo", false);
}

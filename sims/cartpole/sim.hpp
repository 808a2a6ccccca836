// Config 1 of BASELINE.json: cartpole-style 2-component ECS.
// Backend-agnostic: this file and sim.cpp compile unchanged against the
// reference headers (CPU TaskGraphExecutor) and against madrona_amd's header
// overlay (MI355X HIP backend, MADRONA_GPU_MODE defined).
#pragma once

#include <madrona/taskgraph_builder.hpp>
#include <madrona/custom_context.hpp>
#include <madrona/rand.hpp>

namespace cartpole {

using madrona::Entity;
using madrona::RandKey;

enum class ExportID : uint32_t {
    State,
    Action,
    Reward,
    Done,
    NumExports,
};

// x, xDot, theta, thetaDot
struct CartState {
    float x;
    float xDot;
    float theta;
    float thetaDot;
};

// 0 = push left, 1 = push right, <0 = sample from the world's RNG
struct Action {
    int32_t push;
};

struct Reward {
    float v;
};

struct Done {
    int32_t v;
};

struct EpisodeInfo {
    uint32_t episodeIdx;
    uint32_t stepIdx;
};

struct Cart : public madrona::Archetype<
    CartState, Action, Reward, Done, EpisodeInfo
> {};

class Engine;

struct Sim : public madrona::WorldBase {
    struct Config {
        uint32_t seed;
        uint32_t worldBase;
    };

    struct WorldInit {};

    static void registerTypes(madrona::ECSRegistry &registry,
                              const Config &cfg);

    static void setupTasks(madrona::TaskGraphManager &taskgraph_mgr,
                           const Config &cfg);

    Sim(Engine &ctx, const Config &cfg, const WorldInit &init);

    RandKey worldKey;
    Entity cart;
};

class Engine : public madrona::CustomContext<Engine, Sim> {
public:
    using CustomContext::CustomContext;
};

}

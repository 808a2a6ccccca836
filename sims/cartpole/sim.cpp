#include "sim.hpp"

#ifdef MADRONA_GPU_MODE
#include <madrona/mw_gpu_entry.hpp>
#endif

using namespace madrona;

namespace cartpole {

// libm's sinf/cosf differ in the last ulp between glibc (reference CPU
// backend) and the device math library; the pole dynamics amplify that.
// Use range-reduced polynomials built only from + - * so both backends
// produce the same bits when compiled with -ffp-contract=off.
static inline float polySin(float x)
{
    // valid for |x| <= pi/2 (the pole angle never leaves +-0.21 rad + margin)
    float x2 = x * x;
    float p = -1.9515295891e-4f;
    p = p * x2 + 8.3321608736e-3f;
    p = p * x2 - 1.6666654611e-1f;
    return x + x * x2 * p;
}

static inline float polyCos(float x)
{
    float x2 = x * x;
    float p = 2.443315711809948e-5f;
    p = p * x2 - 1.388731625493765e-3f;
    p = p * x2 + 4.166664568298827e-2f;
    return 1.f - 0.5f * x2 + x2 * x2 * p;
}

static inline void resetCart(const Sim &sim, CartState &state,
                             EpisodeInfo &info)
{
    RandKey ep_key = rand::split_i(sim.worldKey, info.episodeIdx);

    // uniform in [-0.05, 0.05)
    auto sample = [&](uint32_t i) {
        return rand::sampleUniform(rand::split_i(ep_key, i)) * 0.1f - 0.05f;
    };

    state.x = sample(0);
    state.xDot = sample(1);
    state.theta = sample(2);
    state.thetaDot = sample(3);

    info.stepIdx = 0;
}

void Sim::registerTypes(ECSRegistry &registry, const Config &)
{
    registry.registerComponent<CartState>();
    registry.registerComponent<Action>();
    registry.registerComponent<Reward>();
    registry.registerComponent<Done>();
    registry.registerComponent<EpisodeInfo>();

    registry.registerArchetype<Cart>();

    registry.exportColumn<Cart, CartState>((uint32_t)ExportID::State);
    registry.exportColumn<Cart, Action>((uint32_t)ExportID::Action);
    registry.exportColumn<Cart, Reward>((uint32_t)ExportID::Reward);
    registry.exportColumn<Cart, Done>((uint32_t)ExportID::Done);
}

inline void cartpoleStep(Engine &ctx,
                         CartState &state,
                         Action &action,
                         Reward &reward,
                         Done &done,
                         EpisodeInfo &info)
{
    constexpr float gravity = 9.8f;
    constexpr float mass_cart = 1.f;
    constexpr float mass_pole = 0.1f;
    constexpr float total_mass = mass_cart + mass_pole;
    constexpr float half_len = 0.5f;
    constexpr float pole_mass_len = mass_pole * half_len;
    constexpr float force_mag = 10.f;
    constexpr float dt = 0.02f;
    constexpr float theta_limit = 12.f * 2.f * 3.14159265358979f / 360.f;
    constexpr float x_limit = 2.4f;

    const Sim &sim = ctx.data();

    int32_t push = action.push;
    if (push < 0) {
        RandKey ep_key = rand::split_i(sim.worldKey, info.episodeIdx);
        RandKey act_key = rand::split_i(ep_key, 1000u + info.stepIdx);
        push = rand::sampleI32(act_key, 0, 2);
    }

    float force = push == 1 ? force_mag : -force_mag;

    float cos_t = polyCos(state.theta);
    float sin_t = polySin(state.theta);

    float temp = (force + pole_mass_len * state.thetaDot * state.thetaDot *
                  sin_t) / total_mass;
    float theta_acc = (gravity * sin_t - cos_t * temp) /
        (half_len * (4.f / 3.f - mass_pole * cos_t * cos_t / total_mass));
    float x_acc = temp - pole_mass_len * theta_acc * cos_t / total_mass;

    state.x = state.x + dt * state.xDot;
    state.xDot = state.xDot + dt * x_acc;
    state.theta = state.theta + dt * state.thetaDot;
    state.thetaDot = state.thetaDot + dt * theta_acc;

    info.stepIdx += 1;

    bool failed = state.x < -x_limit || state.x > x_limit ||
        state.theta < -theta_limit || state.theta > theta_limit;
    bool timeout = info.stepIdx >= 200;

    reward.v = failed ? 0.f : 1.f;
    done.v = (failed || timeout) ? 1 : 0;

    if (failed || timeout) {
        info.episodeIdx += 1;
        resetCart(sim, state, info);
    }
}

void Sim::setupTasks(TaskGraphManager &taskgraph_mgr, const Config &)
{
    TaskGraphBuilder &builder = taskgraph_mgr.init(0);

    builder.addToGraph<ParallelForNode<Engine, cartpoleStep,
        CartState, Action, Reward, Done, EpisodeInfo>>({});
}

Sim::Sim(Engine &ctx, const Config &cfg, const WorldInit &)
    : WorldBase(ctx)
{
    uint32_t global_world = cfg.worldBase + (uint32_t)ctx.worldID().idx;
    worldKey = rand::split_i(rand::initKey(cfg.seed), global_world);

    cart = ctx.makeEntity<Cart>();

    CartState &state = ctx.get<CartState>(cart);
    EpisodeInfo &info = ctx.get<EpisodeInfo>(cart);
    info.episodeIdx = 0;
    resetCart(*this, state, info);

    ctx.get<Action>(cart).push = -1;
    ctx.get<Reward>(cart).v = 0.f;
    ctx.get<Done>(cart).v = 0;
}

#ifdef MADRONA_GPU_MODE
MADRONA_BUILD_MWGPU_ENTRY(Engine, Sim, Sim::Config, Sim::WorldInit);
#endif

}

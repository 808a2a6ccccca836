#include "sim.hpp"

#include <cstring>

#ifdef MADRONA_GPU_MODE
#include <madrona/mw_gpu_entry.hpp>
#endif

// SIM_WAVE_API: this backend's wave-cooperative extensions (ordered create /
// destroy by a wavefront per world, box queries with a lane per BVH leaf,
// per-system occupancy) -- API the reference does not have.  -DSIM_PORTABLE
// builds the simulator WITHOUT them even under MADRONA_GPU_MODE: the sources as
// an unchanged reference simulator has them (plain makeEntity / destroyEntity /
// findEntitiesWithinAABB, one lane per world; only the reference's own GPU
// conventions remain: CustomParallelForNode for the ray systems,
// RecycleEntitiesNode).  lib<sim>_portable_hip.so, the `portable_sim` bench line.
#if defined(MADRONA_GPU_MODE) && !defined(SIM_PORTABLE)
#define SIM_WAVE_API 1
#endif
// (measurement: -DSIM_NO_ROW_CHAIN keeps the two door systems two nodes)
#if defined(SIM_WAVE_API) && !defined(SIM_NO_ROW_CHAIN)
#define SIM_ROW_CHAIN 1
#endif

using namespace madrona;
using namespace madrona::math;

namespace escape {

// cos / sin of k * 2pi / 8 and of k * 2pi / 30 as literals: libm results differ
// in the last ulp between glibc and the device math library, literals do not.
static constexpr float kMoveSin[8] = {
    0.f, 0.70710678f, 1.f, 0.70710678f, 0.f, -0.70710678f, -1.f, -0.70710678f,
};
static constexpr float kMoveCos[8] = {
    1.f, 0.70710678f, 0.f, -0.70710678f, -1.f, -0.70710678f, 0.f, 0.70710678f,
};

static constexpr float kLidarCos[consts::numLidarSamples] = {
    1.f, 0.9781476f, 0.91354546f, 0.80901699f, 0.66913061f, 0.5f,
    0.30901699f, 0.10452846f, -0.10452846f, -0.30901699f, -0.5f,
    -0.66913061f, -0.80901699f, -0.91354546f, -0.9781476f, -1.f,
    -0.9781476f, -0.91354546f, -0.80901699f, -0.66913061f, -0.5f,
    -0.30901699f, -0.10452846f, 0.10452846f, 0.30901699f, 0.5f,
    0.66913061f, 0.80901699f, 0.91354546f, 0.9781476f,
};
static constexpr float kLidarSin[consts::numLidarSamples] = {
    0.f, 0.20791169f, 0.40673664f, 0.58778525f, 0.74314483f, 0.8660254f,
    0.95105652f, 0.9945219f, 0.9945219f, 0.95105652f, 0.8660254f,
    0.74314483f, 0.58778525f, 0.40673664f, 0.20791169f, 0.f,
    -0.20791169f, -0.40673664f, -0.58778525f, -0.74314483f, -0.8660254f,
    -0.95105652f, -0.9945219f, -0.9945219f, -0.95105652f, -0.8660254f,
    -0.74314483f, -0.58778525f, -0.40673664f, -0.20791169f,
};

void Sim::registerTypes(ECSRegistry &registry, const Config &)
{
    base::registerTypes(registry);

    registry.registerComponent<ResponseType>();
    registry.registerComponent<Velocity>();
    registry.registerComponent<ExternalForce>();
    registry.registerComponent<ExternalTorque>();
    registry.registerComponent<SubstepPrevState>();
    registry.registerComponent<PreSolvePositional>();
    registry.registerComponent<PreSolveVelocity>();
    registry.registerBundle<SolverState>();
    registry.registerBundle<RigidBody>();

    registry.registerComponent<Action>();
    registry.registerComponent<Reward>();
    registry.registerComponent<Done>();
    registry.registerComponent<SelfObservation>();
    registry.registerComponent<PartnerObservation>();
    registry.registerComponent<RoomEntityObservations>();
    registry.registerComponent<DoorObservation>();
    registry.registerComponent<Lidar>();
    registry.registerComponent<StepsRemaining>();
    registry.registerComponent<Progress>();
    registry.registerComponent<OtherAgents>();
    registry.registerComponent<GrabState>();
    registry.registerComponent<EntityType>();
    registry.registerComponent<OpenState>();
    registry.registerComponent<DoorProperties>();
    registry.registerComponent<ButtonState>();

    registry.registerSingleton<WorldReset>();
    registry.registerSingleton<LevelState>();

    registry.registerArchetype<Agent>();
    registry.registerArchetype<PhysicsEntity>();
    registry.registerArchetype<DoorEntity>();
    registry.registerArchetype<ButtonEntity>();

    registry.exportSingleton<WorldReset>((uint32_t)ExportID::Reset);
    registry.exportColumn<Agent, Action>((uint32_t)ExportID::Action);
    registry.exportColumn<Agent, Reward>((uint32_t)ExportID::Reward);
    registry.exportColumn<Agent, Done>((uint32_t)ExportID::Done);
    registry.exportColumn<Agent, SelfObservation>(
        (uint32_t)ExportID::SelfObservation);
    registry.exportColumn<Agent, PartnerObservation>(
        (uint32_t)ExportID::PartnerObservation);
    registry.exportColumn<Agent, RoomEntityObservations>(
        (uint32_t)ExportID::RoomEntityObservations);
    registry.exportColumn<Agent, DoorObservation>(
        (uint32_t)ExportID::DoorObservation);
    registry.exportColumn<Agent, Lidar>((uint32_t)ExportID::Lidar);
    registry.exportColumn<Agent, StepsRemaining>(
        (uint32_t)ExportID::StepsRemaining);
}

// ---------------------------------------------------------------------------
// level generation
// ---------------------------------------------------------------------------
static inline float randInRange(RNG &rng, float lo, float hi)
{
    return lo + rng.sampleUniform() * (hi - lo);
}

template <typename ArchetypeT>
static inline void setupRigidBody(Engine &ctx, Entity e, Vector3 pos, Quat rot,
                                  SimObject obj, EntityType type,
                                  ResponseType response, Diag3x3 scale)
{
    ctx.get<Position>(e) = pos;
    ctx.get<Rotation>(e) = rot;
    ctx.get<Scale>(e) = scale;
    ctx.get<ObjectID>(e) = ObjectID { (int32_t)obj };
    ctx.get<ResponseType>(e) = response;
    ctx.get<Velocity>(e) = Velocity { Vector3::zero(), Vector3::zero() };
    ctx.get<ExternalForce>(e) = Vector3::zero();
    ctx.get<ExternalTorque>(e) = Vector3::zero();
    ctx.get<SubstepPrevState>(e) = SubstepPrevState { pos, rot };
    ctx.get<PreSolvePositional>(e) = PreSolvePositional { pos, rot };
    ctx.get<PreSolveVelocity>(e) =
        PreSolveVelocity { Vector3::zero(), Vector3::zero() };
    ctx.get<EntityType>(e) = type;
}

static void generateLevel(Engine &ctx, RNG &rng)
{
    LevelState &level = ctx.singleton<LevelState>();

    const float half_width = consts::worldWidth / 2.f;

    for (int32_t r = 0; r < consts::numRooms; r++) {
        Room &room = level.rooms[r];
        const float y_min = (float)r * consts::roomLength;
        const float y_max = y_min + consts::roomLength;

        for (int32_t b = 0; b < consts::numButtonsPerRoom; b++) {
            Entity button = ctx.makeEntity<ButtonEntity>();
            Vector3 pos {
                randInRange(rng, -half_width + 2.f, half_width - 2.f),
                randInRange(rng, y_min + 2.f, y_max - 3.f),
                0.f,
            };
            ctx.get<Position>(button) = pos;
            ctx.get<Rotation>(button) = Quat::id();
            ctx.get<Scale>(button) = Diag3x3 {
                consts::buttonWidth, consts::buttonWidth, 0.2f,
            };
            ctx.get<ObjectID>(button) = ObjectID { (int32_t)SimObject::Button };
            ctx.get<ButtonState>(button).isPressed = 0;
            ctx.get<EntityType>(button) = EntityType::Button;
            room.buttons[b] = button;
        }

        Entity door = ctx.makeEntity<DoorEntity>();
        setupRigidBody<DoorEntity>(ctx, door,
            Vector3 { randInRange(rng, -half_width + 3.f, half_width - 3.f),
                      y_max, 0.f },
            Quat::id(), SimObject::Door, EntityType::Door,
            ResponseType::Static, Diag3x3 { 3.f * 0.8f, 1.f, 1.75f });
        ctx.get<OpenState>(door).isOpen = 0;
        DoorProperties &props = ctx.get<DoorProperties>(door);
        for (int32_t b = 0; b < 4; b++) {
            props.buttons[b] = b < consts::numButtonsPerRoom ?
                room.buttons[b] : Entity::none();
        }
        // rooms alternate between "any agent on both buttons" persistent and
        // non-persistent doors
        props.numButtons = 1 + rng.sampleI32(0, consts::numButtonsPerRoom);
        props.isPersistent = rng.sampleBool() ? 1 : 0;
        room.door = door;

        for (int32_t c = 0; c < consts::numCubesPerRoom; c++) {
            Entity cube = ctx.makeEntity<PhysicsEntity>();
            Vector3 pos {
                randInRange(rng, -half_width + 1.5f, half_width - 1.5f),
                randInRange(rng, y_min + 1.5f, y_max - 2.5f),
                0.75f,
            };
            setupRigidBody<PhysicsEntity>(ctx, cube, pos, Quat::id(),
                SimObject::Cube, EntityType::Cube, ResponseType::Dynamic,
                Diag3x3 { 1.5f, 1.5f, 1.5f });
            // give the cubes something to do under the kinematic integrator
            ctx.get<Velocity>(cube).linear = Vector3 {
                randInRange(rng, -1.f, 1.f), randInRange(rng, -1.f, 1.f), 0.f,
            };
            ctx.get<Velocity>(cube).angular = Vector3 {
                0.f, 0.f, randInRange(rng, -0.5f, 0.5f),
            };
            room.cubes[c] = cube;
        }
    }
}

static void resetAgents(Engine &ctx, RNG &rng)
{
    Sim &sim = ctx.data();
    const float half_width = consts::worldWidth / 2.f;

    for (int32_t i = 0; i < consts::numAgents; i++) {
        Entity agent = sim.agents[i];

        Vector3 pos {
            randInRange(rng, -half_width + 2.f, half_width - 2.f),
            randInRange(rng, 1.5f, 3.f),
            0.f,
        };
        // heading: rotation about z by one of the 8 move angles
        int32_t heading = rng.sampleI32(0, 8);
        // half angle = heading * pi/8: use the 16-step table entries we have
        // via the double-angle identities to stay libm free
        float c = kMoveCos[heading];
        float s = kMoveSin[heading];
        // q = (cos(a/2), 0, 0, sin(a/2)) with cos(a/2) = sqrt((1+c)/2)
        float ch = sqrtf((1.f + c) * 0.5f);
        float sh = sqrtf((1.f - c) * 0.5f);
        if (s < 0.f) sh = -sh;
        Quat rot = Quat { ch, 0.f, 0.f, sh }.normalize();

        ctx.get<Position>(agent) = pos;
        ctx.get<Rotation>(agent) = rot;
        ctx.get<Velocity>(agent) = Velocity { Vector3::zero(), Vector3::zero() };
        ctx.get<ExternalForce>(agent) = Vector3::zero();
        ctx.get<ExternalTorque>(agent) = Vector3::zero();
        ctx.get<SubstepPrevState>(agent) = SubstepPrevState { pos, rot };
        ctx.get<PreSolvePositional>(agent) = PreSolvePositional { pos, rot };
        ctx.get<PreSolveVelocity>(agent) =
            PreSolveVelocity { Vector3::zero(), Vector3::zero() };
        ctx.get<Action>(agent) = Action { 0, 0, 0, 0 };
        ctx.get<Progress>(agent).maxY = pos.y;
        ctx.get<StepsRemaining>(agent).t = consts::episodeLen;
        ctx.get<GrabState>(agent).constraintEntity = Entity::none();
        ctx.get<Reward>(agent).v = 0.f;
        ctx.get<Done>(agent).v = 0;
    }
}

static void createPersistentEntities(Engine &ctx)
{
    Sim &sim = ctx.data();

    for (int32_t i = 0; i < consts::numAgents; i++) {
        Entity agent = ctx.makeEntity<Agent>();
        sim.agents[i] = agent;

        ctx.get<Scale>(agent) = Diag3x3 { 1.f, 1.f, 1.f };
        ctx.get<ObjectID>(agent) = ObjectID { (int32_t)SimObject::Agent };
        ctx.get<ResponseType>(agent) = ResponseType::Dynamic;
        ctx.get<EntityType>(agent) = EntityType::Agent;
    }

    for (int32_t i = 0; i < consts::numAgents; i++) {
        OtherAgents &others = ctx.get<OtherAgents>(sim.agents[i]);
        int32_t out = 0;
        for (int32_t j = 0; j < consts::numAgents; j++) {
            if (j != i) {
                others.e[out++] = sim.agents[j];
            }
        }
    }
}

static void initWorld(Engine &ctx)
{
    Sim &sim = ctx.data();

    // a fresh RNG stream per (world, episode)
    // the stream lives in registers while the level is generated (the world
    // object is in memory that every component store might alias)
    RNG rng(rand::split_i(sim.initRandKey, sim.curWorldEpisode++));

    resetAgents(ctx, rng);
    generateLevel(ctx, rng);

    sim.rng = rng;
}

static void cleanupWorld(Engine &ctx)
{
    LevelState &level = ctx.singleton<LevelState>();
    for (int32_t r = 0; r < consts::numRooms; r++) {
        Room &room = level.rooms[r];
        for (int32_t c = 0; c < consts::numCubesPerRoom; c++) {
            ctx.destroyEntity(room.cubes[c]);
        }
        ctx.destroyEntity(room.door);
        for (int32_t b = 0; b < consts::numButtonsPerRoom; b++) {
            ctx.destroyEntity(room.buttons[b]);
        }
    }
}

// ---------------------------------------------------------------------------
// systems
// ---------------------------------------------------------------------------
inline void movementSystem(Engine &,
                           Action &action,
                           Rotation &rot,
                           ExternalForce &external_force,
                           ExternalTorque &external_torque)
{
    constexpr float move_max = 1000.f;
    constexpr float turn_max = 320.f;

    Quat cur_rot = rot;

    float move_amount = (float)action.moveAmount *
        (move_max / (float)(consts::numMoveAmountBuckets - 1));

    int32_t angle = action.moveAngle & (consts::numMoveAngleBuckets - 1);
    float f_x = move_amount * kMoveSin[angle];
    float f_y = move_amount * kMoveCos[angle];

    constexpr float turn_delta_per_bucket =
        turn_max / (float)(consts::numTurnBuckets / 2);
    float t_z = turn_delta_per_bucket * (float)action.rotate;

    external_force = cur_rot.rotateVec(Vector3 { f_x, f_y, 0.f });
    external_torque = Vector3 { 0.f, 0.f, t_z };
}

// Stand-in for the physics substep: semi-implicit Euler with damping, walls
// as position clamps.  Touches the same columns substepRigidBodies does
// (reference src/physics/xpbd.cpp:100-185).
inline void kinematicStepSystem(Engine &,
                                Position &position,
                                Rotation &rotation,
                                Velocity &velocity,
                                const ResponseType &response,
                                ExternalForce &external_force,
                                ExternalTorque &external_torque,
                                SubstepPrevState &prev_state,
                                PreSolvePositional &presolve_pos,
                                PreSolveVelocity &presolve_vel)
{
    if (response != ResponseType::Dynamic) {
        return;
    }

    constexpr float h = consts::deltaT;
    constexpr float inv_mass = 1.f / 50.f;
    constexpr float inv_inertia = 1.f / 60.f;
    constexpr float damping = 0.9f;

    Vector3 x = position;
    Quat q = rotation;

    prev_state.prevPosition = x;
    prev_state.prevRotation = q;

    Vector3 v = velocity.linear;
    Vector3 omega = velocity.angular;

    v += h * inv_mass * Vector3(external_force);
    omega += h * inv_inertia * Vector3(external_torque);
    v *= damping;
    omega *= damping;

    x += h * v;

    Quat apply_omega = Quat::fromAngularVec(0.5f * h * omega);
    q += apply_omega * q;
    q = q.normalize();

    // arena walls
    const float x_lim = consts::worldWidth / 2.f - consts::agentRadius;
    if (x.x < -x_lim) { x.x = -x_lim; v.x = -v.x; }
    if (x.x > x_lim) { x.x = x_lim; v.x = -v.x; }
    if (x.y < consts::agentRadius) { x.y = consts::agentRadius; v.y = -v.y; }
    if (x.y > consts::worldLength - consts::agentRadius) {
        x.y = consts::worldLength - consts::agentRadius;
        v.y = -v.y;
    }

    presolve_pos.x = x;
    presolve_pos.q = q;
    presolve_vel.v = v;
    presolve_vel.omega = omega;

    position = x;
    rotation = q;
    velocity.linear = v;
    velocity.angular = omega;

    external_force = Vector3::zero();
    external_torque = Vector3::zero();
}

inline void buttonSystem(Engine &ctx,
                         Position &pos,
                         ButtonState &state)
{
    const Sim &sim = ctx.data();

    bool pressed = false;
    for (int32_t i = 0; i < consts::numAgents; i++) {
        Vector3 agent_pos = ctx.get<Position>(sim.agents[i]);
        float dx = fabsf(agent_pos.x - pos.x);
        float dy = fabsf(agent_pos.y - pos.y);
        if (dx < consts::buttonWidth && dy < consts::buttonWidth) {
            pressed = true;
        }
    }

    state.isPressed = pressed ? 1 : 0;
}

inline void doorOpenSystem(Engine &ctx,
                           OpenState &open_state,
                           const DoorProperties &props)
{
    bool all_pressed = true;
    for (int32_t i = 0; i < props.numButtons; i++) {
        Entity button = props.buttons[i];
        all_pressed = all_pressed && ctx.get<ButtonState>(button).isPressed != 0;
    }

    if (all_pressed) {
        open_state.isOpen = 1;
    } else if (props.isPersistent == 0) {
        open_state.isOpen = 0;
    }
}

inline void setDoorPositionSystem(Engine &,
                                  Position &pos,
                                  OpenState &open_state)
{
    if (open_state.isOpen != 0) {
        if (pos.z > -4.5f) {
            pos.z += -consts::doorSpeed * consts::deltaT;
        }
    } else if (pos.z < 0.f) {
        pos.z += consts::doorSpeed * consts::deltaT;
    }

    if (pos.z >= 0.f) {
        pos.z = 0.f;
    }
}

#ifdef SIM_ROW_CHAIN
// Both door systems in one node: each only touches its own door's row (the
// button states doorOpenSystem reaches for are written by neither), so row i
// may run the second right behind the first -- one launch instead of two
// (madrona::mwhip::rowChain).  The portable sources
// and the CPU build keep the two nodes.
inline constexpr auto doorSystem = madrona::mwhip::rowChain<
    doorOpenSystem, setDoorPositionSystem, Engine,
    OpenState, DoorProperties, Position>;
#endif

inline void rewardSystem(Engine &,
                         Position &pos,
                         Progress &progress,
                         Reward &out_reward)
{
    float reward_pos = fminf(pos.y, consts::worldLength * 2.f);

    float old_max_y = progress.maxY;
    float new_progress = reward_pos - old_max_y;

    float reward;
    if (new_progress > 0.f) {
        reward = new_progress * consts::rewardPerDist;
        progress.maxY = reward_pos;
    } else {
        reward = consts::slackReward;
    }

    out_reward.v = reward;
}

inline void stepTrackerSystem(Engine &,
                              StepsRemaining &steps_remaining,
                              Done &done)
{
    int32_t num_remaining = (int32_t)--steps_remaining.t;
    if (num_remaining == consts::episodeLen - 1) {
        done.v = 0;
    } else if (num_remaining == 0) {
        done.v = 1;
    }
}

#ifdef SIM_WAVE_API
// The reset of a world on the GPU backends: the reset system runs 64 lanes per
// world (CustomParallelForNode<..., 64, 1>, the reference simulators' way of
// spreading heavy per-world work over a warp) and lane i destroys / creates /
// fills in entity i, instead of one lane doing 27 of each in sequence.  What
// comes out is what cleanupWorld() + initWorld() above leave: the same entities
// with the same ids (Context::destroyEntityOrdered / makeEntityOrdered hand ids
// back and out in lane order = the order of the loops above), the same rows in
// the same order, and the same random numbers -- the RNG is counter based,
// sample i of an episode is split_i(episode key, i), and the loops above draw a
// fixed number per entity, so every lane computes the indices of its own draws:
//   agent a:   3 a + { 0: x, 1: y, 2: heading }
//   room r:    base = 3 numAgents + r * (2 numButtons + 3 + 5 numCubes)
//     button b:  base + 2 b + { 0: x, 1: y }
//     door:      base + 2 numButtons + { 0: x, 1: numButtons, 2: isPersistent }
//     cube c:    base + 2 numButtons + 3 + 5 c + { 0: x, 1: y, 2: vx, 3: vy, 4: wz }
static inline void resetWorldWave(Engine &ctx)
{
    Sim &sim = ctx.data();
    LevelState &level = ctx.singleton<LevelState>();

    constexpr int32_t per_room = consts::numButtonsPerRoom + 1 + consts::numCubesPerRoom;
    constexpr int32_t num_level = consts::numRooms * per_room;
    constexpr uint32_t draws_per_room =
        2u * consts::numButtonsPerRoom + 3u + 5u * consts::numCubesPerRoom;
    constexpr uint32_t num_draws = 3u * consts::numAgents +
        (uint32_t)consts::numRooms * draws_per_room;
    static_assert(num_level + consts::numAgents <= 64);

    const int32_t lane = (int32_t)(threadIdx.x % 64);
    const int32_t r = lane / per_room;
    const int32_t k = lane % per_room;
    const bool in_level = lane < num_level;

    // ---- cleanupWorld(): per room the cubes, the door, the buttons ----
    {
        Entity e = Entity::none();
        if (in_level) {
            const Room &room = level.rooms[r];
            e = k < consts::numCubesPerRoom ? room.cubes[k] :
                (k == consts::numCubesPerRoom ? room.door :
                 room.buttons[k - consts::numCubesPerRoom - 1]);
        }
        ctx.destroyEntityOrdered(e, in_level);
    }

    // ---- initWorld() ----
    const RandKey episode = rand::split_i(sim.initRandKey, sim.curWorldEpisode);
    auto key = [&](uint32_t i) { return rand::split_i(episode, i); };
    auto in_range = [&](uint32_t i, float lo, float hi) {
        return lo + rand::sampleUniform(key(i)) * (hi - lo);
    };
    const float half_width = consts::worldWidth / 2.f;

    // resetAgents(): one lane per agent, behind the level's lanes
    if (lane >= num_level && lane < num_level + consts::numAgents) {
        const int32_t i = lane - num_level;
        const uint32_t base = 3u * (uint32_t)i;
        Entity agent = sim.agents[i];

        Vector3 pos {
            in_range(base, -half_width + 2.f, half_width - 2.f),
            in_range(base + 1u, 1.5f, 3.f),
            0.f,
        };
        int32_t heading = rand::sampleI32(key(base + 2u), 0, 8);
        float c = kMoveCos[heading];
        float sn = kMoveSin[heading];
        float ch = sqrtf((1.f + c) * 0.5f);
        float sh = sqrtf((1.f - c) * 0.5f);
        if (sn < 0.f) sh = -sh;
        Quat rot = Quat { ch, 0.f, 0.f, sh }.normalize();

        ctx.get<Position>(agent) = pos;
        ctx.get<Rotation>(agent) = rot;
        ctx.get<Velocity>(agent) = Velocity { Vector3::zero(), Vector3::zero() };
        ctx.get<ExternalForce>(agent) = Vector3::zero();
        ctx.get<ExternalTorque>(agent) = Vector3::zero();
        ctx.get<SubstepPrevState>(agent) = SubstepPrevState { pos, rot };
        ctx.get<PreSolvePositional>(agent) = PreSolvePositional { pos, rot };
        ctx.get<PreSolveVelocity>(agent) =
            PreSolveVelocity { Vector3::zero(), Vector3::zero() };
        ctx.get<Action>(agent) = Action { 0, 0, 0, 0 };
        ctx.get<Progress>(agent).maxY = pos.y;
        ctx.get<StepsRemaining>(agent).t = consts::episodeLen;
        ctx.get<GrabState>(agent).constraintEntity = Entity::none();
        ctx.get<Reward>(agent).v = 0.f;
        ctx.get<Done>(agent).v = 0;
    }

    // generateLevel(): per room the buttons, the door, the cubes -- in that
    // order, which is the order ids are handed out in
    const bool is_button = in_level && k < consts::numButtonsPerRoom;
    const bool is_door = in_level && k == consts::numButtonsPerRoom;
    [[maybe_unused]] const bool is_cube = in_level && k > consts::numButtonsPerRoom;
    const uint32_t archetype = is_button ? TypeTracker::typeID<ButtonEntity>() :
        (is_door ? TypeTracker::typeID<DoorEntity>() :
                   TypeTracker::typeID<PhysicsEntity>());
    const Entity e = ctx.makeEntityOrdered(archetype, in_level);

    // (the door wants its room's buttons)
    Entity room_buttons[consts::numButtonsPerRoom];
    for (int32_t b = 0; b < consts::numButtonsPerRoom; b++) {
        const int32_t src = (r < consts::numRooms ? r : 0) * per_room + b;
        room_buttons[b].gen = (uint32_t)__shfl((int)e.gen, src, 64);
        room_buttons[b].id = __shfl(e.id, src, 64);
    }

    if (in_level) {
        Room &room = level.rooms[r];
        const float y_min = (float)r * consts::roomLength;
        const float y_max = y_min + consts::roomLength;
        const uint32_t base =
            3u * consts::numAgents + (uint32_t)r * draws_per_room;

        if (is_button) {
            const uint32_t at = base + 2u * (uint32_t)k;
            Vector3 pos {
                in_range(at, -half_width + 2.f, half_width - 2.f),
                in_range(at + 1u, y_min + 2.f, y_max - 3.f),
                0.f,
            };
            ctx.get<Position>(e) = pos;
            ctx.get<Rotation>(e) = Quat::id();
            ctx.get<Scale>(e) = Diag3x3 {
                consts::buttonWidth, consts::buttonWidth, 0.2f,
            };
            ctx.get<ObjectID>(e) = ObjectID { (int32_t)SimObject::Button };
            ctx.get<ButtonState>(e).isPressed = 0;
            ctx.get<EntityType>(e) = EntityType::Button;
            room.buttons[k] = e;
        } else if (is_door) {
            const uint32_t at = base + 2u * consts::numButtonsPerRoom;
            setupRigidBody<DoorEntity>(ctx, e,
                Vector3 { in_range(at, -half_width + 3.f, half_width - 3.f),
                          y_max, 0.f },
                Quat::id(), SimObject::Door, EntityType::Door,
                ResponseType::Static, Diag3x3 { 3.f * 0.8f, 1.f, 1.75f });
            ctx.get<OpenState>(e).isOpen = 0;
            DoorProperties &props = ctx.get<DoorProperties>(e);
            for (int32_t b = 0; b < 4; b++) {
                props.buttons[b] = b < consts::numButtonsPerRoom ?
                    room_buttons[b] : Entity::none();
            }
            props.numButtons = 1 + rand::sampleI32(key(at + 1u), 0,
                                                   consts::numButtonsPerRoom);
            props.isPersistent = rand::sampleBool(key(at + 2u)) ? 1 : 0;
            room.door = e;
        } else {
            const int32_t c = k - consts::numButtonsPerRoom - 1;
            const uint32_t at =
                base + 2u * consts::numButtonsPerRoom + 3u + 5u * (uint32_t)c;
            Vector3 pos {
                in_range(at, -half_width + 1.5f, half_width - 1.5f),
                in_range(at + 1u, y_min + 1.5f, y_max - 2.5f),
                0.75f,
            };
            setupRigidBody<PhysicsEntity>(ctx, e, pos, Quat::id(),
                SimObject::Cube, EntityType::Cube, ResponseType::Dynamic,
                Diag3x3 { 1.5f, 1.5f, 1.5f });
            ctx.get<Velocity>(e).linear = Vector3 {
                in_range(at + 2u, -1.f, 1.f), in_range(at + 3u, -1.f, 1.f), 0.f,
            };
            ctx.get<Velocity>(e).angular = Vector3 {
                0.f, 0.f, in_range(at + 4u, -0.5f, 0.5f),
            };
            room.cubes[c] = e;
        }
    }

    if (lane == 0) {
        // what `sim.rng = rng` leaves after the sequential draws: the episode's
        // stream, advanced past them (RNG = { key, samples drawn })
        struct RNGState { RandKey k; uint32_t count; };
        static_assert(sizeof(RNGState) == sizeof(RNG));
        RNGState state { episode, num_draws };
        memcpy(&sim.rng, &state, sizeof(RNG));
        sim.curWorldEpisode += 1;
    }
}
#endif

inline void resetSystem(Engine &ctx, WorldReset &reset)
{
    Sim &sim = ctx.data();

    int32_t should_reset = reset.reset;

    for (int32_t i = 0; i < consts::numAgents; i++) {
        if (ctx.get<Done>(sim.agents[i]).v != 0) {
            should_reset = 1;
        }
    }

#ifdef SIM_WAVE_API
    // 64 lanes per world: lane 0 advances the reset stream, everybody learns
    // the outcome
    int32_t auto_reset = 0;
    if (sim.autoResetDenom != 0 && threadIdx.x % 64 == 0) {
        auto_reset =
            sim.resetRng.sampleI32(0, (int32_t)sim.autoResetDenom) == 0 ? 1 : 0;
    }
    if (__shfl(auto_reset, 0, 64) != 0) {
        should_reset = 1;
    }

    if (should_reset != 0) {
        if (threadIdx.x % 64 == 0) {
            reset.reset = 0;
        }
        resetWorldWave(ctx);
    }
#else
    if (sim.autoResetDenom != 0) {
        if (sim.resetRng.sampleI32(0, (int32_t)sim.autoResetDenom) == 0) {
            should_reset = 1;
        }
    }

    if (should_reset != 0) {
        reset.reset = 0;
        cleanupWorld(ctx);
        initWorld(ctx);
    }
#endif
}

inline void collectObservationsSystem(Engine &ctx,
                                      Position &pos,
                                      Rotation &rot,
                                      const Progress &progress,
                                      const GrabState &grab,
                                      const OtherAgents &other_agents,
                                      SelfObservation &self_obs,
                                      PartnerObservation &partner_obs,
                                      RoomEntityObservations &room_ent_obs,
                                      DoorObservation &door_obs)
{
    const LevelState &level = ctx.singleton<LevelState>();

    int32_t room_idx = (int32_t)(pos.y / consts::roomLength);
    if (room_idx < 0) room_idx = 0;
    if (room_idx > consts::numRooms - 1) room_idx = consts::numRooms - 1;
    const Room &room = level.rooms[room_idx];

    const float room_y_min = (float)room_idx * consts::roomLength;

    self_obs.roomX = pos.x / (consts::worldWidth / 2.f);
    self_obs.roomY = (pos.y - room_y_min) / consts::roomLength;
    self_obs.globalX = pos.x / consts::worldWidth;
    self_obs.globalY = pos.y / consts::worldLength;
    self_obs.globalZ = pos.z / 10.f;
    self_obs.maxY = progress.maxY / consts::worldLength;
    self_obs.facing = rot.z;
    self_obs.isGrabbing =
        grab.constraintEntity != Entity::none() ? 1.f : 0.f;

    Quat to_view = rot.inv();

    {
        Entity other = other_agents.e[0];
        Vector3 other_pos = ctx.get<Position>(other);
        Vector3 rel = to_view.rotateVec(other_pos - pos);
        partner_obs.dx = rel.x / consts::worldLength;
        partner_obs.dy = rel.y / consts::worldLength;
        partner_obs.isGrabbing =
            ctx.get<GrabState>(other).constraintEntity != Entity::none() ?
                1.f : 0.f;
    }

    int32_t out = 0;
    for (int32_t c = 0; c < consts::numCubesPerRoom; c++) {
        Entity e = room.cubes[c];
        Vector3 rel = to_view.rotateVec(ctx.get<Position>(e) - pos);
        room_ent_obs.obs[out++] = EntityObservation {
            rel.x / consts::worldLength, rel.y / consts::worldLength,
            (float)ctx.get<EntityType>(e) / (float)EntityType::NumTypes,
        };
    }
    for (int32_t b = 0; b < consts::numButtonsPerRoom; b++) {
        Entity e = room.buttons[b];
        Vector3 rel = to_view.rotateVec(ctx.get<Position>(e) - pos);
        room_ent_obs.obs[out++] = EntityObservation {
            rel.x / consts::worldLength, rel.y / consts::worldLength,
            (float)ctx.get<EntityType>(e) / (float)EntityType::NumTypes,
        };
    }
    room_ent_obs.obs[out] = EntityObservation { 0.f, 0.f, 0.f };

    {
        Entity door = room.door;
        Vector3 rel = to_view.rotateVec(ctx.get<Position>(door) - pos);
        door_obs.dx = rel.x / consts::worldLength;
        door_obs.dy = rel.y / consts::worldLength;
        door_obs.isOpen = ctx.get<OpenState>(door).isOpen != 0 ? 1.f : 0.f;
    }
}

// Analytic stand-in for the BVH ray cast: distance along each of 30 view rays
// to the arena walls and to the other agent (circle).
inline void lidarSystem(Engine &ctx,
                        Entity e,
                        Lidar &lidar)
{
    Vector3 pos = ctx.get<Position>(e);
    Quat rot = ctx.get<Rotation>(e);
    Vector3 other_pos = ctx.get<Position>(ctx.get<OtherAgents>(e).e[0]);

    const float x_lim = consts::worldWidth / 2.f;

    for (int32_t i = 0; i < consts::numLidarSamples; i++) {
        Vector3 dir = rot.rotateVec(
            Vector3 { kLidarCos[i], kLidarSin[i], 0.f });

        float t_hit = 200.f;
        float hit_type = (float)EntityType::None;

        // walls
        if (dir.x > 1e-6f) {
            float t = (x_lim - pos.x) / dir.x;
            if (t < t_hit) { t_hit = t; hit_type = (float)EntityType::Wall; }
        } else if (dir.x < -1e-6f) {
            float t = (-x_lim - pos.x) / dir.x;
            if (t < t_hit) { t_hit = t; hit_type = (float)EntityType::Wall; }
        }
        if (dir.y > 1e-6f) {
            float t = (consts::worldLength - pos.y) / dir.y;
            if (t < t_hit) { t_hit = t; hit_type = (float)EntityType::Wall; }
        } else if (dir.y < -1e-6f) {
            float t = (0.f - pos.y) / dir.y;
            if (t < t_hit) { t_hit = t; hit_type = (float)EntityType::Wall; }
        }

        // other agent: |pos + t dir - c|^2 = r^2
        Vector3 oc = pos - other_pos;
        float b = 2.f * (oc.x * dir.x + oc.y * dir.y);
        float c = oc.x * oc.x + oc.y * oc.y -
            consts::agentRadius * consts::agentRadius;
        float t1, t2;
        if (solveQuadraticUnsafe(1.f, b, c, &t1, &t2)) {
            if (t1 > 0.f && t1 < t_hit) {
                t_hit = t1;
                hit_type = (float)EntityType::Agent;
            }
        }

        lidar.samples[i] = LidarSample {
            t_hit / 200.f,
            hit_type / (float)EntityType::NumTypes,
        };
    }
}

#ifdef MADRONA_GPU_MODE
// ---------------------------------------------------------------------------
// What each system reads and writes per row ("each node declares
// its read/write set next to the kernel"; madrona::mwhip::systemIO,
// taskgraph.inl) -- the algorithmic bytes the bench prices the ParallelFor
// nodes at.  A row that returns early moves less: the set is what a row that
// does the system's work moves.
// ---------------------------------------------------------------------------
}
#define ESCAPE_SYSTEM_IO(fn, ...) \
    template <> inline constexpr madrona::mwhip::SystemIOBytes \
        madrona::mwhip::systemIO<escape::fn> = \
            madrona::mwhip::declareIO<__VA_ARGS__>()
namespace escape_io {
using namespace escape;
using madrona::Entity;
using madrona::mwhip::Reads;
using madrona::mwhip::Times;
using madrona::mwhip::Writes;
}
ESCAPE_SYSTEM_IO(movementSystem,
    escape_io::Reads<escape::Action, escape::Rotation>,
    escape_io::Writes<escape::ExternalForce, escape::ExternalTorque>);
// (a Dynamic row; the others stop after their ResponseType)
ESCAPE_SYSTEM_IO(kinematicStepSystem,
    escape_io::Reads<escape::Position, escape::Rotation, escape::Velocity,
                     escape::ResponseType, escape::ExternalForce,
                     escape::ExternalTorque>,
    escape_io::Writes<escape::Position, escape::Rotation, escape::Velocity,
                      escape::ExternalForce, escape::ExternalTorque,
                      escape::SubstepPrevState, escape::PreSolvePositional,
                      escape::PreSolveVelocity>);
// own position + the position of both agents (ctx.get)
ESCAPE_SYSTEM_IO(buttonSystem,
    escape_io::Reads<escape::Position,
                     escape_io::Times<escape::Position, escape::consts::numAgents>>,
    escape_io::Writes<escape::ButtonState>);
// the door's properties + the state of its (up to four, here two) buttons
ESCAPE_SYSTEM_IO(doorOpenSystem,
    escape_io::Reads<escape::DoorProperties,
                     escape_io::Times<escape::ButtonState,
                                      escape::consts::numButtonsPerRoom>>,
    escape_io::Writes<escape::OpenState>);
ESCAPE_SYSTEM_IO(setDoorPositionSystem,
    escape_io::Reads<escape::Position, escape::OpenState>,
    escape_io::Writes<escape::Position>);
#ifdef SIM_ROW_CHAIN
// (the chain of the two: OpenState is read -- a persistent door keeps it -- and
// written once, not written, stored, and read back)
ESCAPE_SYSTEM_IO(doorSystem,
    escape_io::Reads<escape::DoorProperties,
                escape_io::Times<escape::ButtonState, escape::consts::numButtonsPerRoom>,
                escape::OpenState, escape::Position>,
    escape_io::Writes<escape::OpenState, escape::Position>);
#endif
ESCAPE_SYSTEM_IO(rewardSystem,
    escape_io::Reads<escape::Position, escape::Progress>,
    escape_io::Writes<escape::Progress, escape::Reward>);
ESCAPE_SYSTEM_IO(stepTrackerSystem,
    escape_io::Reads<escape::StepsRemaining>,
    escape_io::Writes<escape::StepsRemaining, escape::Done>);
// a world that does not reset: its flag and both agents' Done (the work of a
// reset is the simulator's level generation, not priced here)
ESCAPE_SYSTEM_IO(resetSystem,
    escape_io::Reads<escape::WorldReset,
                     escape_io::Times<escape::Done, escape::consts::numAgents>>,
    escape_io::Writes<>);
// the agent's row, its room (LevelState singleton), the partner's position and
// grab state, position + type of the room's cubes and buttons, the door
ESCAPE_SYSTEM_IO(collectObservationsSystem,
    escape_io::Reads<escape::Position, escape::Rotation, escape::Progress,
                     escape::GrabState, escape::OtherAgents, escape::Room,
                     escape::Position, escape::GrabState,
                     escape_io::Times<escape::Position,
                         escape::consts::numCubesPerRoom +
                         escape::consts::numButtonsPerRoom>,
                     escape_io::Times<escape::EntityType,
                         escape::consts::numCubesPerRoom +
                         escape::consts::numButtonsPerRoom>,
                     escape::Position, escape::OpenState>,
    escape_io::Writes<escape::SelfObservation, escape::PartnerObservation,
                      escape::RoomEntityObservations, escape::DoorObservation>);
// (the analytic stand-in for the ray cast: own pose, the other agent's position)
ESCAPE_SYSTEM_IO(lidarSystem,
    escape_io::Reads<escape_io::Entity, escape::Position, escape::Rotation,
                     escape::OtherAgents, escape::Position>,
    escape_io::Writes<escape::Lidar>);
#undef ESCAPE_SYSTEM_IO
namespace escape {
#endif

// The dependencies are the real ones, not a chain: the button -> door chain, the
// reward and the step counter all follow the kinematic step and touch disjoint
// components (ButtonState / OpenState / door Position; Progress / Reward;
// StepsRemaining / Done -- the agents' Position is only read); observations and
// lidar both only read the state the step left.  A backend that runs nodes one
// after the other in the builder's order (the reference's) is unaffected; the
// MI355X executor runs nodes that wait for the same launches side by side
// (DESIGN.md section 15.7).
void Sim::setupTasks(TaskGraphManager &taskgraph_mgr, const Config &)
{
    TaskGraphBuilder &builder = taskgraph_mgr.init(0);

    auto move_sys = builder.addToGraph<ParallelForNode<Engine,
        movementSystem,
            Action,
            Rotation,
            ExternalForce,
            ExternalTorque
        >>({});

    auto kinematic_sys = builder.addToGraph<ParallelForNode<Engine,
        kinematicStepSystem,
            Position,
            Rotation,
            Velocity,
            ResponseType,
            ExternalForce,
            ExternalTorque,
            SubstepPrevState,
            PreSolvePositional,
            PreSolveVelocity
        >>({move_sys});

    auto button_sys = builder.addToGraph<ParallelForNode<Engine,
        buttonSystem,
            Position,
            ButtonState
        >>({kinematic_sys});

    auto reward_sys = builder.addToGraph<ParallelForNode<Engine,
        rewardSystem,
            Position,
            Progress,
            Reward
        >>({kinematic_sys});

    auto done_sys = builder.addToGraph<ParallelForNode<Engine,
        stepTrackerSystem,
            StepsRemaining,
            Done
        >>({kinematic_sys});

    // (the door chain is registered behind the nodes that share the button
    // system's dependency: the executor runs those in one launch)
#ifdef SIM_ROW_CHAIN
    auto set_door_pos_sys = builder.addToGraph<ParallelForNode<Engine,
        doorSystem,
            OpenState,
            DoorProperties,
            Position
        >>({button_sys});
#else
    auto door_open_sys = builder.addToGraph<ParallelForNode<Engine,
        doorOpenSystem,
            OpenState,
            DoorProperties
        >>({button_sys});

    auto set_door_pos_sys = builder.addToGraph<ParallelForNode<Engine,
        setDoorPositionSystem,
            Position,
            OpenState
        >>({door_open_sys});
#endif

#ifdef SIM_WAVE_API
    // 64 lanes per world: lane i resets entity i (resetWorldWave)
    auto reset_sys = builder.addToGraph<CustomParallelForNode<Engine,
        resetSystem, 64, 1,
#else
    auto reset_sys = builder.addToGraph<ParallelForNode<Engine,
        resetSystem,
#endif
            WorldReset
        >>({set_door_pos_sys, reward_sys, done_sys});

#ifdef MADRONA_GPU_MODE
    auto recycle_sys = builder.addToGraph<RecycleEntitiesNode>({reset_sys});
    auto post_reset = recycle_sys;
#else
    auto post_reset = reset_sys;
#endif

    auto compact_cubes = builder.addToGraph<
        CompactArchetypeNode<PhysicsEntity>>({post_reset});
    auto compact_doors = builder.addToGraph<
        CompactArchetypeNode<DoorEntity>>({compact_cubes});
    auto compact_buttons = builder.addToGraph<
        CompactArchetypeNode<ButtonEntity>>({compact_doors});

    auto collect_obs = builder.addToGraph<ParallelForNode<Engine,
        collectObservationsSystem,
            Position,
            Rotation,
            Progress,
            GrabState,
            OtherAgents,
            SelfObservation,
            PartnerObservation,
            RoomEntityObservations,
            DoorObservation
        >>({compact_buttons});

    auto lidar = builder.addToGraph<ParallelForNode<Engine,
        lidarSystem,
            Entity,
            Lidar
        >>({compact_buttons});

    (void)collect_obs;
    (void)lidar;
}

Sim::Sim(Engine &ctx, const Config &cfg, const WorldInit &)
    : WorldBase(ctx)
{
    uint32_t global_world = cfg.worldBase + (uint32_t)ctx.worldID().idx;

    initRandKey = rand::split_i(rand::initKey(cfg.seed), global_world);
    resetRng = RNG(rand::split_i(initRandKey, 0x7E5E7u));
    curWorldEpisode = 0;
    autoResetDenom = cfg.autoResetDenom;

    ctx.singleton<WorldReset>().reset = 0;

    createPersistentEntities(ctx);
    initWorld(ctx);
}

#ifdef MADRONA_GPU_MODE
MADRONA_BUILD_MWGPU_ENTRY(Engine, Sim, Sim::Config, Sim::WorldInit);
#endif

}

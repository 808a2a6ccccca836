#include "escape_room/sim.hpp"

struct SimTraits;
#include "common/sim_c_api.h"

#include <vector>
#include <string>

struct SimTraits {
    using Sim = escape::Sim;
    using Engine = escape::Engine;

    static constexpr uint32_t numExports =
        (uint32_t)escape::ExportID::NumExports;
    static constexpr uint32_t numTaskGraphs = 1;

    // flags: low 16 bits = autoResetDenom (0 disables random resets)
    static Sim::Config makeConfig(const SimCreateArgs &args)
    {
        return Sim::Config { args.seed, args.world_base, args.flags & 0xFFFFu };
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
    using escape::ExportID;
    namespace c = escape::consts;
    int64_t W = num_worlds;
    int64_t A = c::numAgents;
    out.push_back({ "reset", SIM_I32, { W, 1 }, (uint32_t)ExportID::Reset });
    out.push_back({ "action", SIM_I32, { W, A, 4 }, (uint32_t)ExportID::Action });
    out.push_back({ "reward", SIM_F32, { W, A, 1 }, (uint32_t)ExportID::Reward });
    out.push_back({ "done", SIM_I32, { W, A, 1 }, (uint32_t)ExportID::Done });
    out.push_back({ "self_obs", SIM_F32, { W, A, 8 },
                    (uint32_t)ExportID::SelfObservation });
    out.push_back({ "partner_obs", SIM_F32, { W, A, 3 },
                    (uint32_t)ExportID::PartnerObservation });
    out.push_back({ "room_ent_obs", SIM_F32,
                    { W, A, c::numCubesPerRoom + c::numButtonsPerRoom + 1, 3 },
                    (uint32_t)ExportID::RoomEntityObservations });
    out.push_back({ "door_obs", SIM_F32, { W, A, 3 },
                    (uint32_t)ExportID::DoorObservation });
    out.push_back({ "lidar", SIM_F32, { W, A, c::numLidarSamples, 2 },
                    (uint32_t)ExportID::Lidar });
    out.push_back({ "steps_remaining", SIM_I32, { W, A, 1 },
                    (uint32_t)ExportID::StepsRemaining });
}

template <typename T>
void SimTraits::describeColumns(T &cols)
{
    using namespace escape;
    using madrona::Entity;

    cols.template add<Agent, Entity>("Agent.Entity", false);
    cols.template add<Agent, Position>("Agent.Position", true);
    cols.template add<Agent, Rotation>("Agent.Rotation", true);
    cols.template add<Agent, Velocity>("Agent.Velocity", true);
    cols.template add<Agent, ExternalForce>("Agent.ExternalForce", true);
    cols.template add<Agent, PreSolvePositional>("Agent.PreSolvePositional", true);
    cols.template add<Agent, Action>("Agent.Action", false);
    cols.template add<Agent, Reward>("Agent.Reward", true);
    cols.template add<Agent, Done>("Agent.Done", false);
    cols.template add<Agent, SelfObservation>("Agent.SelfObservation", true);
    cols.template add<Agent, PartnerObservation>("Agent.PartnerObservation", true);
    cols.template add<Agent, RoomEntityObservations>(
        "Agent.RoomEntityObservations", true);
    cols.template add<Agent, DoorObservation>("Agent.DoorObservation", true);
    cols.template add<Agent, Lidar>("Agent.Lidar", true);
    cols.template add<Agent, StepsRemaining>("Agent.StepsRemaining", false);
    cols.template add<Agent, Progress>("Agent.Progress", true);
    cols.template add<Agent, OtherAgents>("Agent.OtherAgents", false);

    cols.template add<PhysicsEntity, Entity>("PhysicsEntity.Entity", false);
    cols.template add<PhysicsEntity, Position>("PhysicsEntity.Position", true);
    cols.template add<PhysicsEntity, Rotation>("PhysicsEntity.Rotation", true);
    cols.template add<PhysicsEntity, Velocity>("PhysicsEntity.Velocity", true);
    cols.template add<PhysicsEntity, SubstepPrevState>(
        "PhysicsEntity.SubstepPrevState", true);
    cols.template add<PhysicsEntity, EntityType>("PhysicsEntity.EntityType", false);

    cols.template add<DoorEntity, Entity>("DoorEntity.Entity", false);
    cols.template add<DoorEntity, Position>("DoorEntity.Position", true);
    cols.template add<DoorEntity, OpenState>("DoorEntity.OpenState", false);
    cols.template add<DoorEntity, DoorProperties>("DoorEntity.DoorProperties", false);

    cols.template add<ButtonEntity, Entity>("ButtonEntity.Entity", false);
    cols.template add<ButtonEntity, Position>("ButtonEntity.Position", true);
    cols.template add<ButtonEntity, ButtonState>("ButtonEntity.ButtonState", false);
}

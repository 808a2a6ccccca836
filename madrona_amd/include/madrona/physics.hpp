// Rigid-body physics on the task graph: BVH broadphase, SAT narrowphase, XPBD.
//
// API contract: reference include/madrona/physics.hpp:13-228 (component types,
// RigidBody bundle, ObjectManager, namespace PhysicsSystem) -- same names,
// layouts and column order (RGDCols, src/physics/physics_impl.hpp:43-58), so
// simulators that embed RigidBody as the first bundle of their archetypes and
// call PhysicsSystem::{registerTypes,init,registerEntity,reset,setup*Tasks}
// compile unchanged.
//
// The implementation (physics.inl, phys_impl/*.hpp) is device code inlined into
// the simulator's translation unit.  It keeps the evaluation order of the
// reference CPU build (src/physics/{broadphase,narrowphase,xpbd}.cpp) so fp32
// state matches the CPU oracle bit for bit under -ffp-contract=off, and it emits
// candidates and contacts in the CPU backend's deterministic order (count ->
// scan -> fill) instead of atomic arrival order.
#pragma once

#include <madrona/math.hpp>
#include <madrona/components.hpp>
#include <madrona/span.hpp>
#include <madrona/taskgraph_builder.hpp>
#include <madrona/context.hpp>
#include <madrona/crash.hpp>

#include <madrona/broadphase.hpp>
#include <madrona/geo.hpp>

namespace madrona::phys {

struct ExternalForce : math::Vector3 {
    MADRONA_HD ExternalForce(math::Vector3 v) : Vector3(v) {}
};

struct ExternalTorque : math::Vector3 {
    MADRONA_HD ExternalTorque(math::Vector3 v) : Vector3(v) {}
};

enum class ResponseType : uint32_t {
    Dynamic,
    Kinematic,
    Static,
};

struct Velocity {
    math::Vector3 linear;
    math::Vector3 angular;
};

struct SolverBundleAlias {};

struct RigidBody : Bundle<
    base::ObjectInstance,
    ResponseType,
    broadphase::LeafID,
    Velocity,
    ExternalForce,
    ExternalTorque,
    SolverBundleAlias
> {};

struct CandidateCollision {
    Loc a;
    Loc b;
    uint32_t aPrim;
    uint32_t bPrim;
};

struct ContactConstraint {
    Loc ref;
    Loc alt;
    math::Vector4 points[4];
    int32_t numPoints;
    math::Vector3 normal;
};

struct JointConstraint {
    enum class Type {
        Fixed,
        Hinge
    };

    struct Fixed {
        math::Quat attachRot1;
        math::Quat attachRot2;
        float separation;
    };

    struct Hinge {
        math::Vector3 a1Local;
        math::Vector3 a2Local;
        math::Vector3 b1Local;
        math::Vector3 b2Local;
    };

    Entity e1;
    Entity e2;
    Type type;

    union {
        Fixed fixed;
        Hinge hinge;
    };

    math::Vector3 r1;
    math::Vector3 r2;
};

struct CollisionEvent {
    Entity a;
    Entity b;
};

struct CollisionEventTemporary : Archetype<CollisionEvent> {};

// Per object state
struct RigidBodyMassData {
    float invMass;
    math::Vector3 invInertiaTensor;
    math::Vector3 toCenterOfMass;
    math::Quat toInteriaFrame;
};

struct RigidBodyFrictionData {
    float muS;
    float muD;
};

struct RigidBodyMetadata {
    RigidBodyMassData mass;
    RigidBodyFrictionData friction;
};

struct CollisionPrimitive {
    enum class Type : uint32_t {
        Sphere = 1 << 0,
        Hull = 1 << 1,
        Plane = 1 << 2,
    };

    struct Sphere {
        float radius;
    };

    struct Hull {
        geo::HalfEdgeMesh halfEdgeMesh;
    };

    struct Plane {};

    Type type;
    union {
        Sphere sphere;
        Plane plane;
        Hull hull;
    };
};

struct ObjectManager {
    CollisionPrimitive *collisionPrimitives;
    math::AABB *primitiveAABBs;

    math::AABB *rigidBodyAABBs;
    uint32_t *rigidBodyPrimitiveOffsets;
    uint32_t *rigidBodyPrimitiveCounts;
    RigidBodyMetadata *metadata;
};

struct ObjectData {
    ObjectManager *mgr;
};

namespace PhysicsSystem {

enum class Solver : uint32_t {
    XPBD,
    TGS,    // not available in this backend yet
};

MADRONA_HD inline void init(Context &ctx,
                            ObjectManager *obj_mgr,
                            float delta_t,
                            CountT num_substeps,
                            math::Vector3 gravity,
                            CountT max_dynamic_objects,
                            Solver solver = Solver::XPBD);

MADRONA_HD inline void reset(Context &ctx);

MADRONA_HD inline broadphase::LeafID registerEntity(Context &ctx,
                                                    Entity e,
                                                    base::ObjectID obj_id);

template <typename Fn>
MADRONA_HD inline void findEntitiesWithinAABB(Context &ctx,
                                              math::AABB aabb,
                                              Fn &&fn);

MADRONA_HD inline bool checkEntityAABBOverlap(Context &ctx,
                                              math::AABB aabb,
                                              Entity e);

MADRONA_HD inline Entity makeFixedJoint(Context &ctx,
                                        Entity e1, Entity e2,
                                        math::Quat attach_rot1,
                                        math::Quat attach_rot2,
                                        math::Vector3 r1, math::Vector3 r2,
                                        float separation);

MADRONA_HD inline Entity makeHingeJoint(Context &ctx,
                                        Entity e1, Entity e2,
                                        math::Vector3 a1_local,
                                        math::Vector3 a2_local,
                                        math::Vector3 b1_local,
                                        math::Vector3 b2_local,
                                        math::Vector3 r1, math::Vector3 r2);

MADRONA_HOST_API inline void registerTypes(ECSRegistry &registry,
                                           Solver solver = Solver::XPBD);

MADRONA_HOST_API inline TaskGraphNodeID setupBroadphaseTasks(
    TaskGraphBuilder &builder,
    Span<const TaskGraphNodeID> deps);

MADRONA_HOST_API inline TaskGraphNodeID setupPhysicsStepTasks(
    TaskGraphBuilder &builder,
    Span<const TaskGraphNodeID> deps,
    CountT num_substeps,
    Solver solver = Solver::XPBD);

MADRONA_HOST_API inline TaskGraphNodeID setupCleanupTasks(
    TaskGraphBuilder &builder,
    Span<const TaskGraphNodeID> deps);

MADRONA_HOST_API inline TaskGraphNodeID setupStandaloneBroadphaseOverlapTasks(
    TaskGraphBuilder &builder,
    Span<const TaskGraphNodeID> deps);

MADRONA_HOST_API inline TaskGraphNodeID setupStandaloneBroadphaseCleanupTasks(
    TaskGraphBuilder &builder,
    Span<const TaskGraphNodeID> deps);

}

// Column indices of the RigidBody bundle inside any archetype that starts with
// it (reference src/physics/physics_impl.hpp:43-58).
namespace RGDCols {
    constexpr inline CountT Position = 2;
    constexpr inline CountT Rotation = 3;
    constexpr inline CountT Scale = 4;
    constexpr inline CountT ObjectID = 5;
    constexpr inline CountT ResponseType = 6;
    constexpr inline CountT LeafID = 7;
    constexpr inline CountT Velocity = 8;
    constexpr inline CountT ExternalForce = 9;
    constexpr inline CountT ExternalTorque = 10;
    constexpr inline CountT SolverBase = 11;

    constexpr inline CountT CandidateCollision = 2;
    constexpr inline CountT ContactConstraint = 2;
    constexpr inline CountT JointConstraint = 2;
}

}

#include "physics.inl"

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

// ---- per-body components (members of the RigidBody bundle) ------------------
enum class ResponseType : uint32_t { Dynamic, Kinematic, Static };

struct Velocity { math::Vector3 linear, angular; };

// forces / torques accumulated by the simulator for the next step
struct ExternalForce : math::Vector3 {
    MADRONA_HD ExternalForce(math::Vector3 v) : Vector3(v) {}
};
struct ExternalTorque : math::Vector3 {
    MADRONA_HD ExternalTorque(math::Vector3 v) : Vector3(v) {}
};

// replaced by the solver's own per-body state when the bundle is registered
struct SolverBundleAlias {};

struct RigidBody : Bundle<base::ObjectInstance, ResponseType, broadphase::LeafID,
                          Velocity, ExternalForce, ExternalTorque,
                          SolverBundleAlias> {};

// ---- what the broadphase / narrowphase / solver pass to each other -----------
struct CandidateCollision {
    Loc a, b;               // the two bodies
    uint32_t aPrim, bPrim;  // primitive of each body to test
};

struct ContactConstraint {
    Loc ref, alt;               // body owning the reference feature, the other
    math::Vector4 points[4];    // xyz = world position, w = penetration depth
    int32_t numPoints;
    math::Vector3 normal;
};

struct JointConstraint {
    enum class Type { Fixed, Hinge };

    struct Fixed {
        math::Quat attachRot1, attachRot2;
        float separation;
    };

    struct Hinge {
        math::Vector3 a1Local, a2Local, b1Local, b2Local;
    };

    Entity e1, e2;
    Type type;
    union {
        Fixed fixed;
        Hinge hinge;
    };
    math::Vector3 r1, r2;
};

struct CollisionEvent { Entity a, b; };
struct CollisionEventTemporary : Archetype<CollisionEvent> {};

// ---- collision assets: one entry per object id -------------------------------
struct RigidBodyMassData {
    float invMass;
    math::Vector3 invInertiaTensor, toCenterOfMass;
    math::Quat toInteriaFrame;
};

struct RigidBodyFrictionData { float muS, muD; };

struct RigidBodyMetadata {
    RigidBodyMassData mass;
    RigidBodyFrictionData friction;
};

struct CollisionPrimitive {
    // values are OR-ed into the narrowphase test id
    enum class Type : uint32_t { Sphere = 1 << 0, Hull = 1 << 1, Plane = 1 << 2 };

    struct Sphere { float radius; };
    struct Hull { geo::HalfEdgeMesh halfEdgeMesh; };
    struct Plane {};

    Type type;
    union {
        Sphere sphere;
        Plane plane;
        Hull hull;
    };
};

// arrays indexed by primitive (first two) and by object id (the rest); filled by
// PhysicsLoader, device resident
// What the LDS-resident step kernels keep next to the CU of the object
// manager's geometry -- the first primitives and as many of their object-space
// hull meshes as fit a small arena -- laid out once by PhysicsLoader, so that a
// world stages it with one coalesced copy instead of walking the primitive
// list (a dependent round trip per hull, per world, per step).
struct PrimImage {
    static constexpr uint32_t maxPrims = 8;
    static constexpr uint32_t arenaDwords = 192;

    uint32_t numPrims;                  // covers primitives [0, numPrims); 0: none
    uint32_t arenaUsed;
    uint32_t pad_[2];
    CollisionPrimitive prims[maxPrims]; // hull pointers as in collisionPrimitives
    math::AABB primAABBs[maxPrims];
    // per hull primitive: dword offsets into `arena` of its face planes, half
    // edges, vertices and face base half edges, or -1 (mesh stays in HBM)
    int32_t meshOffset[maxPrims][4];
    alignas(16) uint32_t arena[arenaDwords];
};

struct ObjectManager {
    CollisionPrimitive *collisionPrimitives;
    math::AABB *primitiveAABBs;
    math::AABB *rigidBodyAABBs;
    uint32_t *rigidBodyPrimitiveOffsets, *rigidBodyPrimitiveCounts;
    RigidBodyMetadata *metadata;
    // (not in the reference's struct; nullptr when the manager was not filled
    // by this backend's PhysicsLoader)
    const PrimImage *primImage;
};

struct ObjectData { ObjectManager *mgr; };

namespace PhysicsSystem {

enum class Solver : uint32_t {
    XPBD,
    TGS,    // phys_impl/tgs.hpp
};

// ---- world constructor / reset ---------------------------------------------------
MADRONA_HD inline void init(Context &ctx, ObjectManager *obj_mgr, float delta_t,
                            CountT num_substeps, math::Vector3 gravity,
                            CountT max_dynamic_objects,
                            Solver solver = Solver::XPBD);
MADRONA_HD inline void reset(Context &ctx);
MADRONA_HD inline broadphase::LeafID registerEntity(
    Context &ctx, Entity e, base::ObjectID obj_id);

// ---- queries against the world's BVH ---------------------------------------------
template <typename Fn>
MADRONA_HD inline void findEntitiesWithinAABB(Context &ctx, math::AABB aabb,
                                              Fn &&fn);
#if defined(__HIPCC__)
// This backend, device code only: for each of num_boxes query boxes the FIRST
// entity findEntitiesWithinAABB would report for which accept(entity) holds
// (Entity::none() if there is none), found by the 64 lanes of the calling
// wavefront together: a lane per BVH leaf answering for all boxes, instead of a
// tree walk per box with a chain of dependent loads per hit.  Every lane calls
// with the same boxes; `accept` is evaluated by different lanes for different
// entities and must not have side effects; the queries see the state as it is
// when the call is made.  For systems that run a wavefront per world
// (CustomParallelForNode<..., 64, 1, ...>), after setupBroadphaseTasks has
// brought the leaves up to date with the poses (leaves are culled by their own
// box, which then contains the body with 100 dt^2 to spare).
template <int MAX_BOXES, typename Fn>
MADRONA_DEVICE inline void findFirstEntitiesWithinAABBsWave(Context &ctx,
                                                            const math::AABB *boxes,
                                                            int32_t num_boxes,
                                                            Entity *out,
                                                            Fn &&accept);
template <typename Fn>
MADRONA_DEVICE inline Entity findFirstEntityWithinAABBWave(Context &ctx,
                                                           math::AABB aabb,
                                                           Fn &&accept);
#endif

MADRONA_HD inline bool checkEntityAABBOverlap(Context &ctx, math::AABB aabb,
                                              Entity e);

// ---- joints (entities of the solver's joint archetype) ---------------------------
MADRONA_HD inline Entity makeFixedJoint(
    Context &ctx, Entity e1, Entity e2,
    math::Quat attach_rot1, math::Quat attach_rot2,
    math::Vector3 r1, math::Vector3 r2, float separation);
MADRONA_HD inline Entity makeHingeJoint(
    Context &ctx, Entity e1, Entity e2,
    math::Vector3 a1_local, math::Vector3 a2_local,
    math::Vector3 b1_local, math::Vector3 b2_local,
    math::Vector3 r1, math::Vector3 r2);

// ---- registration and task graph (host) ------------------------------------------
MADRONA_HOST_API inline void registerTypes(ECSRegistry &registry,
                                           Solver solver = Solver::XPBD);
MADRONA_HOST_API inline TaskGraphNodeID setupBroadphaseTasks(
    TaskGraphBuilder &builder, Span<const TaskGraphNodeID> deps);
MADRONA_HOST_API inline TaskGraphNodeID setupPhysicsStepTasks(
    TaskGraphBuilder &builder, Span<const TaskGraphNodeID> deps,
    CountT num_substeps, Solver solver = Solver::XPBD);
MADRONA_HOST_API inline TaskGraphNodeID setupCleanupTasks(
    TaskGraphBuilder &builder, Span<const TaskGraphNodeID> deps);
MADRONA_HOST_API inline TaskGraphNodeID setupStandaloneBroadphaseOverlapTasks(
    TaskGraphBuilder &builder, Span<const TaskGraphNodeID> deps);
MADRONA_HOST_API inline TaskGraphNodeID setupStandaloneBroadphaseCleanupTasks(
    TaskGraphBuilder &builder, Span<const TaskGraphNodeID> deps);

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

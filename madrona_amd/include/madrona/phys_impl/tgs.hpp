// The second solver behind PhysicsSystem::Solver (reference
// src/physics/tgs.cpp).  In the reference its constraint side is a skeleton --
// prepareContacts / prepareJoints / warmStart* / solveContacts / solveJoints
// have empty bodies (tgs.cpp:59-92, 146-170, 198-208) -- and what a step does
// to the world is its two integrators, per substep:
//   integrateVelocities   v += h g (dynamic bodies) + h m^-1 f; the angular
//                         velocity advanced in body space with the gyroscopic
//                         term (tgs.cpp:93-145)
//   integratePositions    x += h v; q += (h / 2) omega q, renormalised
//                         (tgs.cpp:172-196)
// Bodies therefore move freely: nothing resolves contacts.  The reference still
// runs its narrowphase first and clears the contacts at the end of the step
// (tgs.cpp:225-236, 299-302); those rows are temporaries without entity ids that
// nothing reads, so this backend does not produce them.
// (included by physics.inl inside namespace madrona::phys, behind
// PhysicsSystemState)

namespace tgs {

struct Contact : Archetype<ContactConstraint> {};
struct Joint : Archetype<JointConstraint> {};

// per-body solver state of this solver: none (reference tgs.cpp:17-18), so a
// RigidBody has no columns behind RGDCols::SolverBase
struct TGSRigidBodyState : Bundle<> {};

// (the reference keeps two Query objects here; kept so that the singleton, and
// with it every later entity id, exists)
struct SolverState {
    uint32_t unused[8];
};

inline void integrateVelocities(Context &ctx,
                                base::Rotation q,
                                ResponseType response_type,
                                ExternalForce ext_force,
                                ExternalTorque ext_torque,
                                base::ObjectID obj_id,
                                Velocity &vel)
{
    using namespace math;

    if (response_type == ResponseType::Static) {
        return;
    }

    const PhysicsSystemState &sys = ctx.singleton<PhysicsSystemState>();
    const RigidBodyMetadata &metadata =
        ctx.singleton<ObjectData>().mgr->metadata[obj_id.idx];
    const float h = sys.h;
    const float inv_m = metadata.mass.invMass;
    const Diag3x3 inv_I = Diag3x3::fromVec(metadata.mass.invInertiaTensor);

    Vector3 v = vel.linear;
    if (response_type == ResponseType::Dynamic) {
        v += h * sys.g;
    }
    v += h * inv_m * ext_force;

    // body-space angular update with the gyroscopic term; an infinite inertia
    // about an axis (inverse 0) takes part as 0
    const Diag3x3 I {
        inv_I.d0 == 0 ? 0.f : 1.f / inv_I.d0,
        inv_I.d1 == 0 ? 0.f : 1.f / inv_I.d1,
        inv_I.d2 == 0 ? 0.f : 1.f / inv_I.d2,
    };
    const Vector3 omega = vel.angular;
    const Quat to_local = q.inv();
    const Vector3 tau_local = to_local.rotateVec(ext_torque);
    Vector3 omega_local = to_local.rotateVec(omega);
    // (the reference multiplies the inertia with the WORLD-space omega here,
    // tgs.cpp:135-136: reproduced)
    omega_local += h * inv_I * (tau_local - cross(omega_local, I * omega));

    vel.linear = v;
    vel.angular = q.rotateVec(omega_local);
}

inline void integratePositions(Context &ctx,
                               base::Position &pos,
                               base::Rotation &rot,
                               Velocity vel)
{
    using namespace math;

    const float h = ctx.singleton<PhysicsSystemState>().h;

    Vector3 x = pos;
    x += h * vel.linear;

    Quat q = rot;
    q += Quat::fromAngularVec(0.5f * h * vel.angular) * q;

    pos = x;
    rot = q.normalize();
}

}

// XPBD rigid-body solver math (Mueller et al., "Detailed Rigid Body Simulation
// with Extended Position Based Dynamics"): substep integration, positional
// contact + static friction, joints, velocity update, restitution and dynamic
// friction.  Pure functions over plain values (host + device) so they can be
// checked on the CPU; the ECS-facing kernels are in physics.inl.
//
// Numerics follow the reference solver step for step (src/physics/xpbd.cpp:
// integration :100-185, positional updates :206-275, contacts :304-550, joints
// :552-718, velocities :738-779, restitution / friction :781-1039) so that
// fp32 state equals the CPU oracle's under -ffp-contract=off.
#pragma once

#include <madrona/math.hpp>
#include <madrona/physics.hpp>

namespace madrona::phys::xpbd {

using math::Vector3;
using math::Vector4;
using math::Quat;

struct XPBDContactState {
    float lambdaN[4];
};

struct SubstepPrevState {
    math::Vector3 prevPosition;
    math::Quat prevRotation;
};

struct PreSolvePositional {
    math::Vector3 x;
    math::Quat q;
};

struct PreSolveVelocity {
    math::Vector3 v;
    math::Vector3 omega;
};

struct Contact : Archetype<ContactConstraint, XPBDContactState> {};
struct Joint : Archetype<JointConstraint> {};

struct XPBDRigidBodyState : Bundle<
    SubstepPrevState,
    PreSolvePositional,
    PreSolveVelocity
> {};

namespace XPBDCols {
    constexpr inline CountT SubstepPrevState = RGDCols::SolverBase;
    constexpr inline CountT PreSolvePositional = RGDCols::SolverBase + 1;
    constexpr inline CountT PreSolveVelocity = RGDCols::SolverBase + 2;
}

// Everything the solver reads / writes for one body, gathered once.
struct BodyState {
    Vector3 x;
    Quat q;
    float invMass;
    Vector3 invInertia;
};

MADRONA_HD inline Vector3 multDiag(Vector3 diag, Vector3 v)
{
    return Vector3 { diag.x * v.x, diag.y * v.y, diag.z * v.z };
}

// ---------------------------------------------------------------------------
// integration
// ---------------------------------------------------------------------------
struct SubstepResult {
    Vector3 x;
    Quat q;
    Vector3 v;
    Vector3 omega;
};

// semi-implicit Euler with the gyroscopic term in the body frame
MADRONA_HD inline SubstepResult integrateBody(Vector3 x, Quat q, Vector3 v,
                                              Vector3 omega, float inv_m,
                                              Vector3 inv_I, Vector3 ext_force,
                                              Vector3 ext_torque, Vector3 gravity,
                                              float h, bool apply_gravity)
{
    if (apply_gravity) {
        v += h * gravity;
    }

    v += h * inv_m * ext_force;
    x += h * v;

    Vector3 I {
        (inv_I.x == 0) ? 0.0f : 1.0f / inv_I.x,
        (inv_I.y == 0) ? 0.0f : 1.0f / inv_I.y,
        (inv_I.z == 0) ? 0.0f : 1.0f / inv_I.z,
    };

    Quat to_local = q.inv();

    Vector3 tau_ext_local = to_local.rotateVec(ext_torque);
    Vector3 omega_local = to_local.rotateVec(omega);

    Vector3 I_omega_local = multDiag(I, omega_local);

    omega_local += h * multDiag(inv_I,
        tau_ext_local - (cross(omega_local, I_omega_local)));

    omega = q.rotateVec(omega_local);

    Quat apply_omega = Quat::fromAngularVec(0.5f * h * omega);
    q += apply_omega * q;
    q = q.normalize();

    return SubstepResult { x, q, v, omega };
}

// ---------------------------------------------------------------------------
// positional constraints
// ---------------------------------------------------------------------------
MADRONA_HD inline float generalizedInverseMass(Vector3 torque_axis,
                                               Vector3 rot_axis, float inv_m)
{
    return inv_m + dot(torque_axis, rot_axis);
}

MADRONA_HD inline float computePositionalLambda(
    Vector3 torque_axis1, Vector3 torque_axis2,
    Vector3 rot_axis1, Vector3 rot_axis2,
    float inv_m1, float inv_m2, float c, float alpha_tilde)
{
    float w1 = generalizedInverseMass(torque_axis1, rot_axis1, inv_m1);
    float w2 = generalizedInverseMass(torque_axis2, rot_axis2, inv_m2);
    return -c / (w1 + w2 + alpha_tilde);
}

MADRONA_HD inline void applyPositionalImpulse(
    Vector3 &x1, Vector3 &x2, Quat &q1, Quat &q2,
    Vector3 rot_axis_local1, Vector3 rot_axis_local2,
    float inv_m1, float inv_m2, Vector3 n, float delta_lambda)
{
    x1 += delta_lambda * inv_m1 * n;
    x2 -= delta_lambda * inv_m2 * n;

    float half_lambda = 0.5f * delta_lambda;

    Vector3 q1_update_angular = q1.rotateVec(half_lambda * rot_axis_local1);
    Vector3 q2_update_angular = q2.rotateVec(half_lambda * rot_axis_local2);

    q1 += Quat::fromAngularVec(q1_update_angular) * q1;
    q2 -= Quat::fromAngularVec(q2_update_angular) * q2;

    q1 = q1.normalize();
    q2 = q2.normalize();
}

// distance constraint of magnitude c along n_world at body-local anchors r1, r2
MADRONA_HD inline float applyPositionalUpdate(
    Vector3 &x1, Vector3 &x2, Quat &q1, Quat &q2,
    Vector3 r1, Vector3 r2, float inv_m1, float inv_m2,
    Vector3 inv_I1, Vector3 inv_I2, Vector3 n_world,
    float c, float alpha_tilde)
{
    Vector3 n_local1 = q1.inv().rotateVec(n_world);
    Vector3 n_local2 = q2.inv().rotateVec(n_world);

    Vector3 torque_axis_local1 = cross(r1, n_local1);
    Vector3 torque_axis_local2 = cross(r2, n_local2);

    Vector3 rot_axis_local1 = multDiag(inv_I1, torque_axis_local1);
    Vector3 rot_axis_local2 = multDiag(inv_I2, torque_axis_local2);

    float lambda = computePositionalLambda(
        torque_axis_local1, torque_axis_local2,
        rot_axis_local1, rot_axis_local2,
        inv_m1, inv_m2, c, alpha_tilde);

    applyPositionalImpulse(x1, x2, q1, q2, rot_axis_local1, rot_axis_local2,
                           inv_m1, inv_m2, n_world, lambda);

    return lambda;
}

struct AngularUpdate {
    Quat q1;
    Quat q2;
};

MADRONA_HD inline AngularUpdate computeAngularUpdate(
    Quat q1, Quat q2, Vector3 inv_I1, Vector3 inv_I2,
    Vector3 n1, Vector3 n2, float theta, float alpha_tilde)
{
    Vector3 local_rot_axis1 = multDiag(inv_I1, n1);
    Vector3 local_rot_axis2 = multDiag(inv_I2, n2);

    float w1 = dot(n1, local_rot_axis1);
    float w2 = dot(n2, local_rot_axis2);

    float delta_lambda = -theta / (w1 + w2 + alpha_tilde);
    float half_lambda = 0.5f * delta_lambda;

    return AngularUpdate {
        Quat::fromAngularVec(q1.rotateVec(half_lambda * local_rot_axis1)),
        Quat::fromAngularVec(q2.rotateVec(half_lambda * local_rot_axis2)),
    };
}

MADRONA_HD inline void applyAngularUpdate(Quat &q1, Quat &q2, AngularUpdate u)
{
    q1 = (q1 + u.q1 * q1).normalize();
    q2 = (q2 - u.q2 * q2).normalize();
}

// ---------------------------------------------------------------------------
// contacts
// ---------------------------------------------------------------------------
// non-penetration along n_world, then static friction on the tangential drift
MADRONA_HD inline void solveContactPoint(
    Vector3 &x1, Vector3 &x2, Quat &q1, Quat &q2,
    SubstepPrevState prev1, SubstepPrevState prev2,
    float inv_m1, float inv_m2, Vector3 inv_I1, Vector3 inv_I2,
    Vector3 r1, Vector3 r2, Vector3 n_world, float avg_mu_s,
    float *lambda_n_out, float *lambda_t_out)
{
    Vector3 p1 = q1.rotateVec(r1) + x1;
    Vector3 p2 = q2.rotateVec(r2) + x2;

    float d = dot(p1 - p2, n_world);
    if (d <= 0) {
        return;
    }

    float lambda_n = applyPositionalUpdate(
        x1, x2, q1, q2, r1, r2, inv_m1, inv_m2, inv_I1, inv_I2, n_world, d, 0);
    *lambda_n_out = lambda_n;

    Vector3 p1_hat = prev1.prevRotation.rotateVec(r1) + prev1.prevPosition;
    Vector3 p2_hat = prev2.prevRotation.rotateVec(r2) + prev2.prevPosition;

    // re-evaluate after the normal correction so friction absorbs its drift
    p1 = q1.rotateVec(r1) + x1;
    p2 = q2.rotateVec(r2) + x2;

    Vector3 delta_p = (p1 - p1_hat) - (p2 - p2_hat);
    Vector3 delta_p_t = delta_p - dot(delta_p, n_world) * n_world;

    float tangential_magnitude = delta_p_t.length();
    if (tangential_magnitude > 0.f) {
        Vector3 t_world = delta_p_t / tangential_magnitude;
        Vector3 t_local1 = q1.inv().rotateVec(t_world);
        Vector3 t_local2 = q2.inv().rotateVec(t_world);

        Vector3 friction_torque_axis_local1 = cross(r1, t_local1);
        Vector3 friction_torque_axis_local2 = cross(r2, t_local2);

        Vector3 friction_rot_axis_local1 =
            multDiag(inv_I1, friction_torque_axis_local1);
        Vector3 friction_rot_axis_local2 =
            multDiag(inv_I2, friction_torque_axis_local2);

        float lambda_t = computePositionalLambda(
            friction_torque_axis_local1, friction_torque_axis_local2,
            friction_rot_axis_local1, friction_rot_axis_local2,
            inv_m1, inv_m2, tangential_magnitude, 0);
        float lambda_threshold = lambda_n * avg_mu_s;

        if (lambda_t > lambda_threshold) {
            *lambda_t_out = lambda_t;

            applyPositionalImpulse(
                x1, x2, q1, q2,
                friction_rot_axis_local1, friction_rot_axis_local2,
                inv_m1, inv_m2, t_world, lambda_t);
        }
    }
}

struct LocalContacts {
    Vector3 r1;
    Vector3 r2;
};

MADRONA_HD inline LocalContacts getLocalSpaceContacts(
    const PreSolvePositional &presolve_pos1,
    const PreSolvePositional &presolve_pos2,
    Vector3 contact1, float penetration_depth, Vector3 contact_normal)
{
    Vector3 contact2 = contact1 - contact_normal * penetration_depth;

    return LocalContacts {
        presolve_pos1.q.inv().rotateVec(contact1 - presolve_pos1.x),
        presolve_pos2.q.inv().rotateVec(contact2 - presolve_pos2.x),
    };
}

// depth-weighted average of the manifold; true when all depths are zero
// (ContactT: a ContactConstraint, or anything with its ref / alt / points[] /
// numPoints / normal -- the LDS step hands in a view of its packed contacts)
template <typename ContactT>
MADRONA_HD inline bool getAvgContact(const ContactT &contact,
                                     Vector3 *avg_out, float *penetration_out)
{
    Vector3 avg_contact = Vector3::zero();

    float max_penetration = -FLT_MAX;
    float penetration_sum = 0.f;
    for (CountT i = 0; i < contact.numPoints; i++) {
        Vector4 pt = contact.points[i];
        if (pt.w > max_penetration) {
            max_penetration = pt.w;
        }
        penetration_sum += pt.w;
    }

    if (penetration_sum == 0.f) {
        return true;
    }

    for (CountT i = 0; i < contact.numPoints; i++) {
        Vector4 pt = contact.points[i];
        avg_contact += pt.w / penetration_sum * pt.xyz();
    }

    *avg_out = avg_contact;
    *penetration_out = max_penetration;
    return false;
}

// ---------------------------------------------------------------------------
// joints
// ---------------------------------------------------------------------------
MADRONA_HD inline void applyJointOrientationConstraint(
    Quat &q1, Quat &q2, Quat attach_q1, Quat attach_q2,
    Vector3 inv_I1, Vector3 inv_I2)
{
    Quat orientation1 = (q1 * attach_q1).normalize();
    Quat orientation2 = (q2 * attach_q2).normalize();

    Quat diff = orientation1 * orientation2.inv();

    Vector3 delta_q = 2.f * Vector3 { diff.x, diff.y, diff.z };
    float delta_q_magnitude = delta_q.length();

    if (delta_q_magnitude > 0) {
        delta_q /= delta_q_magnitude;
        Vector3 delta_q_local1 = q1.inv().rotateVec(delta_q);
        Vector3 delta_q_local2 = q2.inv().rotateVec(delta_q);

        applyAngularUpdate(q1, q2, computeAngularUpdate(
            q1, q2, inv_I1, inv_I2, delta_q_local1, delta_q_local2,
            delta_q_magnitude, 0));
    }
}

MADRONA_HD inline void applyJointAxisConstraint(
    Quat &q1, Quat &q2, Vector3 axis1_local, Vector3 axis2_local,
    Vector3 inv_I1, Vector3 inv_I2)
{
    Vector3 axis1 = q1.rotateVec(axis1_local);
    Vector3 axis2 = q2.rotateVec(axis2_local);

    Vector3 delta_q = cross(axis1, axis2);
    float delta_q_magnitude = delta_q.length();

    if (delta_q_magnitude > 0) {
        delta_q /= delta_q_magnitude;
        Vector3 delta_q_local1 = q1.inv().rotateVec(delta_q);
        Vector3 delta_q_local2 = q2.inv().rotateVec(delta_q);

        applyAngularUpdate(q1, q2, computeAngularUpdate(
            q1, q2, inv_I1, inv_I2, delta_q_local1, delta_q_local2,
            delta_q_magnitude, 0));
    }
}

MADRONA_HD inline void solveJoint(const JointConstraint &joint,
                                  Vector3 &x1, Vector3 &x2, Quat &q1, Quat &q2,
                                  float inv_m1, float inv_m2,
                                  Vector3 inv_I1, Vector3 inv_I2)
{
    Vector3 pos_correction;
    if (joint.type == JointConstraint::Type::Fixed) {
        JointConstraint::Fixed fixed_data = joint.fixed;

        applyJointOrientationConstraint(
            q1, q2, fixed_data.attachRot1, fixed_data.attachRot2,
            inv_I1, inv_I2);

        Vector3 r1_world = q1.rotateVec(joint.r1) + x1;
        Vector3 r2_world = q2.rotateVec(joint.r2) + x2;
        Vector3 delta_r = r2_world - r1_world;

        Quat axes_rot = (q1 * fixed_data.attachRot1).normalize();

        Vector3 a1 = axes_rot.rotateVec(math::fwd);
        Vector3 b1 = axes_rot.rotateVec(math::right);
        Vector3 c1 = cross(a1, b1);

        // fixed separation along a1, none along the other two axes
        pos_correction = Vector3::zero();
        float a_separation = dot(delta_r, a1);
        pos_correction -= (a_separation - fixed_data.separation) * a1;
        float b_separation = dot(delta_r, b1);
        pos_correction -= b_separation * b1;
        float c_separation = dot(delta_r, c1);
        pos_correction -= c_separation * c1;
    } else {
        JointConstraint::Hinge hinge_data = joint.hinge;

        applyJointAxisConstraint(q1, q2, hinge_data.a1Local,
                                 hinge_data.a2Local, inv_I1, inv_I2);

        Vector3 r1_world = q1.rotateVec(joint.r1) + x1;
        Vector3 r2_world = q2.rotateVec(joint.r2) + x2;

        pos_correction = r2_world - r1_world;
    }

    float pos_correction_magnitude = pos_correction.length();
    if (pos_correction_magnitude > 0.f) {
        pos_correction /= pos_correction_magnitude;

        applyPositionalUpdate(
            x1, x2, q1, q2, joint.r1, joint.r2, inv_m1, inv_m2,
            inv_I1, inv_I2, pos_correction, pos_correction_magnitude, 0);
    }
}

// ---------------------------------------------------------------------------
// velocities
// ---------------------------------------------------------------------------
MADRONA_HD inline Velocity deriveVelocity(Vector3 x, Quat q,
                                          const SubstepPrevState &prev_state,
                                          float h)
{
    Vector3 x_prev = prev_state.prevPosition;
    Quat q_prev = prev_state.prevRotation;

    // identical rotations must give exactly zero angular velocity
    Quat delta_q;
    if (q.w != q_prev.w || q.x != q_prev.x ||
            q.y != q_prev.y || q.z != q_prev.z) {
        delta_q = q * q_prev.inv();
    } else {
        delta_q = Quat { 1, 0, 0, 0 };
    }

    Vector3 new_omega = 2.f / h * Vector3 { delta_q.x, delta_q.y, delta_q.z };

    return Velocity {
        (x - x_prev) / h,
        delta_q.w > 0.f ? new_omega : -new_omega,
    };
}

MADRONA_HD inline Vector3 computeRelativeVelocity(
    Vector3 v1, Vector3 v2, Vector3 omega1, Vector3 omega2,
    Vector3 dir1, Vector3 dir2)
{
    return (v1 + cross(omega1, dir1)) - (v2 + cross(omega2, dir2));
}

MADRONA_HD inline void applyFrictionVelocityUpdate(
    Vector3 &v1, Vector3 &v2, Vector3 &omega1, Vector3 &omega2,
    Quat q1, Quat q2, float inv_m1, float inv_m2,
    Vector3 inv_I1, Vector3 inv_I2, Vector3 n, float mu_d, float h,
    Vector3 r1_local, Vector3 r2_local, Vector3 r1_world, Vector3 r2_world,
    float lambda)
{
    Vector3 v = computeRelativeVelocity(
        v1, v2, omega1, omega2, r1_world, r2_world);

    float vn = dot(n, v);
    Vector3 vt = v - n * vn;

    float vt_len = vt.length();
    if (vt_len == 0.f) {
        return;
    }

    Vector3 delta_world = vt / vt_len;

    Vector3 delta_local1 = q1.inv().rotateVec(delta_world);
    Vector3 delta_local2 = q2.inv().rotateVec(delta_world);

    Vector3 friction_torque_axis_local1 = cross(r1_local, delta_local1);
    Vector3 friction_torque_axis_local2 = cross(r2_local, delta_local2);

    Vector3 friction_rot_axis_local1 =
        multDiag(inv_I1, friction_torque_axis_local1);
    Vector3 friction_rot_axis_local2 =
        multDiag(inv_I2, friction_torque_axis_local2);

    float w1 = generalizedInverseMass(
        friction_torque_axis_local1, friction_rot_axis_local1, inv_m1);
    float w2 = generalizedInverseMass(
        friction_torque_axis_local2, friction_rot_axis_local2, inv_m2);

    float inv_mass_scale = 1.f / (w1 + w2);

    float dynamic_friction_magnitude =
        mu_d * fabsf(lambda) * inv_mass_scale / h;

    float corrected_magnitude = -fminf(dynamic_friction_magnitude, vt_len);

    float impulse_magnitude = corrected_magnitude * inv_mass_scale;
    if (impulse_magnitude == 0.f) {
        return;
    }

    v1 += delta_world * impulse_magnitude * inv_m1;
    v2 -= delta_world * impulse_magnitude * inv_m2;

    omega1 += q1.rotateVec(impulse_magnitude * friction_rot_axis_local1);
    omega2 -= q2.rotateVec(impulse_magnitude * friction_rot_axis_local2);
}

MADRONA_HD inline void applyRestitutionVelocityUpdate(
    Vector3 &v1, Vector3 &v2, Vector3 &omega1, Vector3 &omega2,
    Quat q1, Quat q2, float inv_m1, float inv_m2,
    Vector3 inv_I1, Vector3 inv_I2, Vector3 n, float restitution_threshold,
    Vector3 r1_world, Vector3 r2_world,
    Vector3 restitution_torque_axis_local1,
    Vector3 restitution_torque_axis_local2, float vn_bar)
{
    Vector3 v = computeRelativeVelocity(
        v1, v2, omega1, omega2, r1_world, r2_world);

    float vn = dot(n, v);

    float e = 0.3f;     // fixed restitution coefficient, as in the reference
    if (fabsf(vn_bar) <= restitution_threshold) {
        e = 0.f;
    }

    float restitution_magnitude = fminf(-e * vn_bar, 0) - vn;

    Vector3 restitution_rot_axis_local1 =
        multDiag(inv_I1, restitution_torque_axis_local1);
    Vector3 restitution_rot_axis_local2 =
        multDiag(inv_I2, restitution_torque_axis_local2);

    float w1 = generalizedInverseMass(
        restitution_torque_axis_local1, restitution_rot_axis_local1, inv_m1);
    float w2 = generalizedInverseMass(
        restitution_torque_axis_local2, restitution_rot_axis_local2, inv_m2);

    float inv_mass_scale = 1.f / (w1 + w2);

    float impulse_magnitude = restitution_magnitude * inv_mass_scale;
    if (impulse_magnitude == 0.f) {
        return;
    }

    v1 += n * impulse_magnitude * inv_m1;
    v2 -= n * impulse_magnitude * inv_m2;

    omega1 += q1.rotateVec(impulse_magnitude * restitution_rot_axis_local1);
    omega2 -= q2.rotateVec(impulse_magnitude * restitution_rot_axis_local2);
}

// ---------------------------------------------------------------------------
// Constraint solves against a body store: gather the two bodies, solve, scatter
// (reference xpbd.cpp handleContact :454-548, handleJointConstraint :607-718,
// solveVelocitiesForContact :918-1036).  The store decides where body state
// lives: ECS columns in HBM (EcsBodyStore) or a world's LDS block
// (phys_impl/world_step.inl); constraints name bodies by Loc either way.
// ---------------------------------------------------------------------------
struct BodyConstants {
    float invMass;
    Vector3 invInertia;
    RigidBodyFrictionData friction;
};

MADRONA_HD inline BodyConstants bodyConstants(const RigidBodyMetadata &metadata,
                                              ResponseType resp_type)
{
    BodyConstants c {
        metadata.mass.invMass,
        metadata.mass.invInertiaTensor,
        metadata.friction,
    };

    if (resp_type == ResponseType::Static) {
        c.invMass = 0.f;
        c.invInertia = Vector3::zero();
    }

    return c;
}

struct EcsBodyStore {
    Context &ctx;
    const ObjectManager &objMgr;

    MADRONA_HD inline Vector3 &position(Loc l)
    {
        return ctx.getDirect<base::Position>(RGDCols::Position, l);
    }

    MADRONA_HD inline Quat &rotation(Loc l)
    {
        return ctx.getDirect<base::Rotation>(RGDCols::Rotation, l);
    }

    MADRONA_HD inline Velocity &velocity(Loc l)
    {
        return ctx.getDirect<Velocity>(RGDCols::Velocity, l);
    }

    MADRONA_HD inline SubstepPrevState prevState(Loc l)
    {
        return ctx.getDirect<SubstepPrevState>(XPBDCols::SubstepPrevState, l);
    }

    MADRONA_HD inline PreSolvePositional presolvePositional(Loc l)
    {
        return ctx.getDirect<PreSolvePositional>(
            XPBDCols::PreSolvePositional, l);
    }

    MADRONA_HD inline PreSolveVelocity presolveVelocity(Loc l)
    {
        return ctx.getDirect<PreSolveVelocity>(XPBDCols::PreSolveVelocity, l);
    }

    MADRONA_HD inline BodyConstants constants(Loc l)
    {
        base::ObjectID obj_id =
            ctx.getDirect<base::ObjectID>(RGDCols::ObjectID, l);
        ResponseType resp_type =
            ctx.getDirect<ResponseType>(RGDCols::ResponseType, l);
        return bodyConstants(objMgr.metadata[obj_id.idx], resp_type);
    }
};

template <typename StoreT, typename ContactT>
MADRONA_HD inline void handleContact(StoreT &store,
                                     const ContactT &contact,
                                     float *lambdas)
{
    Vector3 *x1_ptr = &store.position(contact.ref);
    Vector3 *x2_ptr = &store.position(contact.alt);
    Quat *q1_ptr = &store.rotation(contact.ref);
    Quat *q2_ptr = &store.rotation(contact.alt);

    SubstepPrevState prev1 = store.prevState(contact.ref);
    SubstepPrevState prev2 = store.prevState(contact.alt);

    PreSolvePositional presolve_pos1 = store.presolvePositional(contact.ref);
    PreSolvePositional presolve_pos2 = store.presolvePositional(contact.alt);

    BodyConstants c1 = store.constants(contact.ref);
    BodyConstants c2 = store.constants(contact.alt);

    Vector3 x1 = *x1_ptr;
    Vector3 x2 = *x2_ptr;
    Quat q1 = *q1_ptr;
    Quat q2 = *q2_ptr;

    float avg_mu_s = 0.5f * (c1.friction.muS + c2.friction.muS);

    Vector3 avg_contact_pos;
    float contact_pos_penetration;
    if (getAvgContact(contact, &avg_contact_pos, &contact_pos_penetration)) {
        return;
    }

    LocalContacts local = getLocalSpaceContacts(
        presolve_pos1, presolve_pos2, avg_contact_pos,
        contact_pos_penetration, contact.normal);

    float lambda_n = 0.f;
    float lambda_t = 0.f;

    solveContactPoint(x1, x2, q1, q2, prev1, prev2,
                      c1.invMass, c2.invMass, c1.invInertia, c2.invInertia,
                      local.r1, local.r2, contact.normal, avg_mu_s,
                      &lambda_n, &lambda_t);

    lambdas[0] = lambda_n;

    *x1_ptr = x1;
    *x2_ptr = x2;
    *q1_ptr = q1;
    *q2_ptr = q2;
}

template <typename StoreT>
MADRONA_HD inline void handleJointConstraint(StoreT &store, Loc l1, Loc l2,
                                             const JointConstraint &joint)
{
    Vector3 *x1_ptr = &store.position(l1);
    Vector3 *x2_ptr = &store.position(l2);
    Quat *q1_ptr = &store.rotation(l1);
    Quat *q2_ptr = &store.rotation(l2);

    Vector3 x1 = *x1_ptr;
    Vector3 x2 = *x2_ptr;
    Quat q1 = *q1_ptr;
    Quat q2 = *q2_ptr;

    BodyConstants c1 = store.constants(l1);
    BodyConstants c2 = store.constants(l2);

    solveJoint(joint, x1, x2, q1, q2, c1.invMass, c2.invMass,
               c1.invInertia, c2.invInertia);

    *x1_ptr = x1;
    *x2_ptr = x2;
    *q1_ptr = q1;
    *q2_ptr = q2;
}

template <typename StoreT, typename ContactT>
MADRONA_HD inline void solveVelocitiesForContact(
    StoreT &store, const ContactT &contact, const float *lambda_n,
    float h, float restitution_threshold)
{
    Velocity *v1_out = &store.velocity(contact.ref);
    Velocity *v2_out = &store.velocity(contact.alt);

    Quat q1 = store.rotation(contact.ref);
    Quat q2 = store.rotation(contact.alt);

    PreSolvePositional presolve_pos1 = store.presolvePositional(contact.ref);
    PreSolvePositional presolve_pos2 = store.presolvePositional(contact.alt);

    PreSolveVelocity presolve_vel1 = store.presolveVelocity(contact.ref);
    PreSolveVelocity presolve_vel2 = store.presolveVelocity(contact.alt);

    BodyConstants c1 = store.constants(contact.ref);
    BodyConstants c2 = store.constants(contact.alt);

    Vector3 v1 = v1_out->linear;
    Vector3 omega1 = v1_out->angular;
    Vector3 v2 = v2_out->linear;
    Vector3 omega2 = v2_out->angular;

    float mu_d = 0.5f * (c1.friction.muD + c2.friction.muD);

    {
        Vector3 avg_contact_pos;
        float contact_pos_penetration;
        if (getAvgContact(contact, &avg_contact_pos,
                          &contact_pos_penetration)) {
            return;
        }

        LocalContacts local = getLocalSpaceContacts(
            presolve_pos1, presolve_pos2, avg_contact_pos,
            contact_pos_penetration, contact.normal);

        Vector3 r1_presolve = presolve_pos1.q.rotateVec(local.r1);
        Vector3 r2_presolve = presolve_pos2.q.rotateVec(local.r2);

        Vector3 v_bar = computeRelativeVelocity(
            presolve_vel1.v, presolve_vel2.v,
            presolve_vel1.omega, presolve_vel2.omega,
            r1_presolve, r2_presolve);

        float vn_bar = dot(contact.normal, v_bar);

        Vector3 r1_world = q1.rotateVec(local.r1);
        Vector3 r2_world = q2.rotateVec(local.r2);

        Vector3 restitution_torque_axis_local1 =
            cross(local.r1, q1.inv().rotateVec(contact.normal));
        Vector3 restitution_torque_axis_local2 =
            cross(local.r2, q2.inv().rotateVec(contact.normal));

        applyRestitutionVelocityUpdate(
            v1, v2, omega1, omega2, q1, q2, c1.invMass, c2.invMass,
            c1.invInertia, c2.invInertia, contact.normal,
            restitution_threshold, r1_world, r2_world,
            restitution_torque_axis_local1, restitution_torque_axis_local2,
            vn_bar);
    }

    float penetration_sum = 0.f;
    for (CountT i = 0; i < contact.numPoints; i++) {
        penetration_sum += contact.points[i].w;
    }

    for (CountT i = 0; i < contact.numPoints; i++) {
        LocalContacts local = getLocalSpaceContacts(
            presolve_pos1, presolve_pos2, contact.points[i].xyz(),
            contact.points[i].w, contact.normal);

        Vector3 r1_world = q1.rotateVec(local.r1);
        Vector3 r2_world = q2.rotateVec(local.r2);

        applyFrictionVelocityUpdate(
            v1, v2, omega1, omega2, q1, q2, c1.invMass, c2.invMass,
            c1.invInertia, c2.invInertia, contact.normal, mu_d, h,
            local.r1, local.r2, r1_world, r2_world,
            lambda_n[0] * (contact.points[i].w / penetration_sum));
    }

    *v1_out = Velocity { v1, omega1 };
    *v2_out = Velocity { v2, omega2 };
}

#if defined(__HIPCC__)
// ---------------------------------------------------------------------------
// The same solves by TWO lanes per constraint (this backend, device only).
// A contact's position solve and velocity solve are symmetric in its two
// bodies: each body's anchor, torque axis, generalised inverse mass and its
// own update are computed from that body's state alone, and the bodies meet in
// a handful of scalars (the gap along the normal, w1 + w2, the relative
// velocity).  One lane per constraint does both bodies one after the other --
// about a thousand instructions a level, and the step kernel is bound by the
// instructions it issues.  Here lane `2t` of a team
// holds the constraint's first body (`second` = false) and lane `2t + 1` its
// second; what the other body contributed arrives by a lane exchange, and what
// involves both is evaluated by both lanes with the operands in the order of
// the one-lane routines above: same expressions on the same operands, the same
// bits.  The second body's updates are subtractions there (x2 -= ..., q2 -= ...);
// a - b and a + (-b) are the same IEEE operation.
// Both lanes of a team must be in the call together (the level loops of
// phys_impl/step_lds.inl see to it).
// ---------------------------------------------------------------------------
namespace paired {

// the other lane of the team: lanes 2t and 2t + 1 swap inside their quad (a DPP
// move: no LDS round trip)
__device__ inline float partner(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(
        __builtin_bit_cast(int, v), 0xB1 /* quad_perm:[1,0,3,2] */, 0xF, 0xF,
        true));
}

__device__ inline Vector3 partner(Vector3 v)
{
    return Vector3 { partner(v.x), partner(v.y), partner(v.z) };
}

// (what body 1 has) - (what body 2 has), from this lane's value
__device__ inline Vector3 firstMinusSecond(bool second, Vector3 mine)
{
    const Vector3 theirs = partner(mine);
    return (second ? theirs : mine) - (second ? mine : theirs);
}

// (what body 2 has) - (what body 1 has): not the negation of the above when
// the two are equal (+0 either way)
__device__ inline Vector3 secondMinusFirst(bool second, Vector3 mine)
{
    const Vector3 theirs = partner(mine);
    return (second ? mine : theirs) - (second ? theirs : mine);
}

// w1 + w2 (+ alpha_tilde: computePositionalLambda adds it even when it is 0)
__device__ inline float bothInverseMasses(bool second, float w_mine)
{
    const float w_theirs = partner(w_mine);
    return (second ? w_theirs : w_mine) + (second ? w_mine : w_theirs);
}

__device__ inline Vector3 signedFor(bool second, Vector3 v)
{
    return Vector3 { second ? -v.x : v.x, second ? -v.y : v.y, second ? -v.z : v.z };
}

// applyPositionalImpulse, this lane's body
__device__ inline void applyPositionalImpulse(bool second, Vector3 &x, Quat &q,
                                              Vector3 rot_axis_local, float inv_m,
                                              Vector3 n, float delta_lambda)
{
    x += signedFor(second, delta_lambda * inv_m * n);

    float half_lambda = 0.5f * delta_lambda;
    Vector3 q_update_angular = q.rotateVec(half_lambda * rot_axis_local);
    Quat dq = Quat::fromAngularVec(q_update_angular) * q;
    q += Quat { second ? -dq.w : dq.w, second ? -dq.x : dq.x,
                second ? -dq.y : dq.y, second ? -dq.z : dq.z };
    q = q.normalize();
}

// applyPositionalUpdate
__device__ inline float applyPositionalUpdate(bool second, Vector3 &x, Quat &q,
                                             Vector3 r, float inv_m, Vector3 inv_I,
                                             Vector3 n_world, float c,
                                             float alpha_tilde)
{
    Vector3 n_local = q.inv().rotateVec(n_world);
    Vector3 torque_axis_local = cross(r, n_local);
    Vector3 rot_axis_local = multDiag(inv_I, torque_axis_local);

    float w = generalizedInverseMass(torque_axis_local, rot_axis_local, inv_m);
    float lambda = -c / (bothInverseMasses(second, w) + alpha_tilde);

    applyPositionalImpulse(second, x, q, rot_axis_local, inv_m, n_world, lambda);
    return lambda;
}

// solveContactPoint
__device__ inline void solveContactPoint(bool second, Vector3 &x, Quat &q,
                                         SubstepPrevState prev, float inv_m,
                                         Vector3 inv_I, Vector3 r, Vector3 n_world,
                                         float avg_mu_s, float *lambda_n_out,
                                         float *lambda_t_out)
{
    Vector3 p = q.rotateVec(r) + x;

    float d = dot(firstMinusSecond(second, p), n_world);
    if (d <= 0) {
        return;
    }

    float lambda_n = applyPositionalUpdate(second, x, q, r, inv_m, inv_I,
                                           n_world, d, 0);
    *lambda_n_out = lambda_n;

    Vector3 p_hat = prev.prevRotation.rotateVec(r) + prev.prevPosition;

    p = q.rotateVec(r) + x;

    Vector3 delta_p = firstMinusSecond(second, p - p_hat);
    Vector3 delta_p_t = delta_p - dot(delta_p, n_world) * n_world;

    float tangential_magnitude = delta_p_t.length();
    if (tangential_magnitude > 0.f) {
        Vector3 t_world = delta_p_t / tangential_magnitude;
        Vector3 t_local = q.inv().rotateVec(t_world);

        Vector3 friction_torque_axis_local = cross(r, t_local);
        Vector3 friction_rot_axis_local =
            multDiag(inv_I, friction_torque_axis_local);

        float w = generalizedInverseMass(friction_torque_axis_local,
                                         friction_rot_axis_local, inv_m);
        float lambda_t =
            -tangential_magnitude / (bothInverseMasses(second, w) + 0);
        float lambda_threshold = lambda_n * avg_mu_s;

        if (lambda_t > lambda_threshold) {
            *lambda_t_out = lambda_t;

            applyPositionalImpulse(second, x, q, friction_rot_axis_local, inv_m,
                                   t_world, lambda_t);
        }
    }
}

// getLocalSpaceContacts, this lane's body
__device__ inline Vector3 localSpaceContact(bool second,
                                            const PreSolvePositional &presolve_pos,
                                            Vector3 contact1,
                                            float penetration_depth,
                                            Vector3 contact_normal)
{
    Vector3 contact2 = contact1 - contact_normal * penetration_depth;
    return presolve_pos.q.inv().rotateVec(
        (second ? contact2 : contact1) - presolve_pos.x);
}

// handleContact; lambdas[0] is the first lane's to keep
template <typename StoreT, typename ContactT>
__device__ inline void handleContact(bool second, StoreT &store,
                                     const ContactT &contact, float *lambdas)
{
    const Loc me = second ? contact.alt : contact.ref;
    Vector3 *x_ptr = &store.position(me);
    Quat *q_ptr = &store.rotation(me);

    SubstepPrevState prev = store.prevState(me);
    PreSolvePositional presolve_pos = store.presolvePositional(me);
    BodyConstants c = store.constants(me);

    Vector3 x = *x_ptr;
    Quat q = *q_ptr;

    const float mu_theirs = partner(c.friction.muS);
    float avg_mu_s = 0.5f * ((second ? mu_theirs : c.friction.muS) +
                             (second ? c.friction.muS : mu_theirs));

    Vector3 avg_contact_pos;
    float contact_pos_penetration;
    if (getAvgContact(contact, &avg_contact_pos, &contact_pos_penetration)) {
        return;
    }

    Vector3 r = localSpaceContact(second, presolve_pos, avg_contact_pos,
                                  contact_pos_penetration, contact.normal);

    float lambda_n = 0.f;
    float lambda_t = 0.f;

    solveContactPoint(second, x, q, prev, c.invMass, c.invInertia, r,
                      contact.normal, avg_mu_s, &lambda_n, &lambda_t);

    lambdas[0] = lambda_n;

    *x_ptr = x;
    *q_ptr = q;
}

__device__ inline Quat partner(Quat v)
{
    return Quat { partner(v.w), partner(v.x), partner(v.y), partner(v.z) };
}

// computeAngularUpdate + applyAngularUpdate, this lane's body (n: delta_q in its
// frame)
__device__ inline void applyAngularUpdate(bool second, Quat &q, Vector3 inv_I,
                                          Vector3 n, float theta, float alpha_tilde)
{
    Vector3 local_rot_axis = multDiag(inv_I, n);
    float w = dot(n, local_rot_axis);

    float delta_lambda = -theta / (bothInverseMasses(second, w) + alpha_tilde);
    float half_lambda = 0.5f * delta_lambda;

    Quat u = Quat::fromAngularVec(q.rotateVec(half_lambda * local_rot_axis));
    Quat dq = u * q;
    q = (q + Quat { second ? -dq.w : dq.w, second ? -dq.x : dq.x,
                    second ? -dq.y : dq.y, second ? -dq.z : dq.z }).normalize();
}

// solveJoint
__device__ inline void solveJoint(bool second, const JointConstraint &joint,
                                  Vector3 &x, Quat &q, float inv_m, Vector3 inv_I)
{
    const Vector3 r = second ? joint.r2 : joint.r1;

    Vector3 pos_correction;
    if (joint.type == JointConstraint::Type::Fixed) {
        JointConstraint::Fixed fixed_data = joint.fixed;
        const Quat attach_q = second ? fixed_data.attachRot2 : fixed_data.attachRot1;

        // applyJointOrientationConstraint
        {
            Quat orientation = (q * attach_q).normalize();
            Quat theirs = partner(orientation);
            Quat diff = (second ? theirs : orientation) *
                (second ? orientation : theirs).inv();

            Vector3 delta_q = 2.f * Vector3 { diff.x, diff.y, diff.z };
            float delta_q_magnitude = delta_q.length();

            if (delta_q_magnitude > 0) {
                delta_q /= delta_q_magnitude;
                Vector3 delta_q_local = q.inv().rotateVec(delta_q);

                applyAngularUpdate(second, q, inv_I, delta_q_local,
                                   delta_q_magnitude, 0);
            }
        }

        Vector3 r_world = q.rotateVec(r) + x;
        // (r2_world - r1_world)
        Vector3 delta_r = secondMinusFirst(second, r_world);

        // (the first body's, after its update)
        Quat axes_mine = (q * fixed_data.attachRot1).normalize();
        Quat axes_theirs = partner(axes_mine);
        Quat axes_rot = second ? axes_theirs : axes_mine;

        Vector3 a1 = axes_rot.rotateVec(math::fwd);
        Vector3 b1 = axes_rot.rotateVec(math::right);
        Vector3 c1 = cross(a1, b1);

        pos_correction = Vector3::zero();
        float a_separation = dot(delta_r, a1);
        pos_correction -= (a_separation - fixed_data.separation) * a1;
        float b_separation = dot(delta_r, b1);
        pos_correction -= b_separation * b1;
        float c_separation = dot(delta_r, c1);
        pos_correction -= c_separation * c1;
    } else {
        JointConstraint::Hinge hinge_data = joint.hinge;

        // applyJointAxisConstraint
        {
            Vector3 axis = q.rotateVec(second ? hinge_data.a2Local :
                                                hinge_data.a1Local);
            Vector3 theirs = partner(axis);
            Vector3 delta_q = cross(second ? theirs : axis, second ? axis : theirs);
            float delta_q_magnitude = delta_q.length();

            if (delta_q_magnitude > 0) {
                delta_q /= delta_q_magnitude;
                Vector3 delta_q_local = q.inv().rotateVec(delta_q);

                applyAngularUpdate(second, q, inv_I, delta_q_local,
                                   delta_q_magnitude, 0);
            }
        }

        Vector3 r_world = q.rotateVec(r) + x;
        pos_correction = secondMinusFirst(second, r_world);
    }

    float pos_correction_magnitude = pos_correction.length();
    if (pos_correction_magnitude > 0.f) {
        pos_correction /= pos_correction_magnitude;

        applyPositionalUpdate(second, x, q, r, inv_m, inv_I, pos_correction,
                              pos_correction_magnitude, 0);
    }
}

// handleJointConstraint: `me` is this lane's end of the joint
template <typename StoreT>
__device__ inline void handleJointConstraint(bool second, StoreT &store, Loc me,
                                             const JointConstraint &joint)
{
    Vector3 *x_ptr = &store.position(me);
    Quat *q_ptr = &store.rotation(me);

    Vector3 x = *x_ptr;
    Quat q = *q_ptr;

    BodyConstants c = store.constants(me);

    solveJoint(second, joint, x, q, c.invMass, c.invInertia);

    *x_ptr = x;
    *q_ptr = q;
}

// (v1 + omega1 x dir1) - (v2 + omega2 x dir2): computeRelativeVelocity
__device__ inline Vector3 relativeVelocity(bool second, Vector3 v, Vector3 omega,
                                           Vector3 dir)
{
    return firstMinusSecond(second, v + cross(omega, dir));
}

__device__ inline void applyRestitutionVelocityUpdate(
    bool second, Vector3 &v, Vector3 &omega, Quat q, float inv_m, Vector3 inv_I,
    Vector3 n, float restitution_threshold, Vector3 r_world,
    Vector3 restitution_torque_axis_local, float vn_bar)
{
    Vector3 v_rel = relativeVelocity(second, v, omega, r_world);

    float vn = dot(n, v_rel);

    float e = 0.3f;
    if (fabsf(vn_bar) <= restitution_threshold) {
        e = 0.f;
    }

    float restitution_magnitude = fminf(-e * vn_bar, 0) - vn;

    Vector3 restitution_rot_axis_local =
        multDiag(inv_I, restitution_torque_axis_local);

    float w = generalizedInverseMass(restitution_torque_axis_local,
                                     restitution_rot_axis_local, inv_m);

    float inv_mass_scale = 1.f / bothInverseMasses(second, w);

    float impulse_magnitude = restitution_magnitude * inv_mass_scale;
    if (impulse_magnitude == 0.f) {
        return;
    }

    v += signedFor(second, n * impulse_magnitude * inv_m);
    omega += signedFor(second,
        q.rotateVec(impulse_magnitude * restitution_rot_axis_local));
}

__device__ inline void applyFrictionVelocityUpdate(
    bool second, Vector3 &v, Vector3 &omega, Quat q, float inv_m, Vector3 inv_I,
    Vector3 n, float mu_d, float h, Vector3 r_local, Vector3 r_world,
    float lambda)
{
    Vector3 v_rel = relativeVelocity(second, v, omega, r_world);

    float vn = dot(n, v_rel);
    Vector3 vt = v_rel - n * vn;

    float vt_len = vt.length();
    if (vt_len == 0.f) {
        return;
    }

    Vector3 delta_world = vt / vt_len;

    Vector3 delta_local = q.inv().rotateVec(delta_world);

    Vector3 friction_torque_axis_local = cross(r_local, delta_local);
    Vector3 friction_rot_axis_local =
        multDiag(inv_I, friction_torque_axis_local);

    float w = generalizedInverseMass(friction_torque_axis_local,
                                     friction_rot_axis_local, inv_m);

    float inv_mass_scale = 1.f / bothInverseMasses(second, w);

    float dynamic_friction_magnitude =
        mu_d * fabsf(lambda) * inv_mass_scale / h;

    float corrected_magnitude = -fminf(dynamic_friction_magnitude, vt_len);

    float impulse_magnitude = corrected_magnitude * inv_mass_scale;
    if (impulse_magnitude == 0.f) {
        return;
    }

    v += signedFor(second, delta_world * impulse_magnitude * inv_m);
    omega += signedFor(second,
        q.rotateVec(impulse_magnitude * friction_rot_axis_local));
}

// solveVelocitiesForContact
template <typename StoreT, typename ContactT>
__device__ inline void solveVelocitiesForContact(
    bool second, StoreT &store, const ContactT &contact, const float *lambda_n,
    float h, float restitution_threshold)
{
    const Loc me = second ? contact.alt : contact.ref;
    Velocity *v_out = &store.velocity(me);

    Quat q = store.rotation(me);

    PreSolvePositional presolve_pos = store.presolvePositional(me);
    PreSolveVelocity presolve_vel = store.presolveVelocity(me);
    BodyConstants c = store.constants(me);

    Vector3 v = v_out->linear;
    Vector3 omega = v_out->angular;

    const float mu_theirs = partner(c.friction.muD);
    float mu_d = 0.5f * ((second ? mu_theirs : c.friction.muD) +
                         (second ? c.friction.muD : mu_theirs));

    {
        Vector3 avg_contact_pos;
        float contact_pos_penetration;
        if (getAvgContact(contact, &avg_contact_pos,
                          &contact_pos_penetration)) {
            return;
        }

        Vector3 r = localSpaceContact(second, presolve_pos, avg_contact_pos,
                                      contact_pos_penetration, contact.normal);

        Vector3 r_presolve = presolve_pos.q.rotateVec(r);

        Vector3 v_bar = relativeVelocity(second, presolve_vel.v,
                                         presolve_vel.omega, r_presolve);

        float vn_bar = dot(contact.normal, v_bar);

        Vector3 r_world = q.rotateVec(r);

        Vector3 restitution_torque_axis_local =
            cross(r, q.inv().rotateVec(contact.normal));

        applyRestitutionVelocityUpdate(
            second, v, omega, q, c.invMass, c.invInertia, contact.normal,
            restitution_threshold, r_world, restitution_torque_axis_local,
            vn_bar);
    }

    float penetration_sum = 0.f;
    for (CountT i = 0; i < contact.numPoints; i++) {
        penetration_sum += contact.points[i].w;
    }

    for (CountT i = 0; i < contact.numPoints; i++) {
        Vector3 r = localSpaceContact(second, presolve_pos,
                                      contact.points[i].xyz(),
                                      contact.points[i].w, contact.normal);

        Vector3 r_world = q.rotateVec(r);

        applyFrictionVelocityUpdate(
            second, v, omega, q, c.invMass, c.invInertia, contact.normal, mu_d,
            h, r, r_world,
            lambda_n[0] * (contact.points[i].w / penetration_sum));
    }

    *v_out = Velocity { v, omega };
}

}
#endif

// ECS-backed conveniences
MADRONA_HD inline void handleContact(Context &ctx,
                                     const ObjectManager &obj_mgr,
                                     const ContactConstraint &contact,
                                     float *lambdas)
{
    EcsBodyStore store { ctx, obj_mgr };
    handleContact(store, contact, lambdas);
}

MADRONA_HD inline void handleJointConstraint(Context &ctx,
                                             const ObjectManager &obj_mgr,
                                             const JointConstraint &joint)
{
    EcsBodyStore store { ctx, obj_mgr };
    handleJointConstraint(store, ctx.loc(joint.e1), ctx.loc(joint.e2), joint);
}

MADRONA_HD inline void solveVelocitiesForContact(
    Context &ctx, const ObjectManager &obj_mgr,
    const ContactConstraint &contact, const float *lambda_n,
    float h, float restitution_threshold)
{
    EcsBodyStore store { ctx, obj_mgr };
    solveVelocitiesForContact(store, contact, lambda_n, h,
                              restitution_threshold);
}

}

// The fused rigid-body step: ONE kernel per simulation step, one wavefront per
// world (included by physics.inl inside namespace madrona::phys::kernels).  The
// code is split by what it does: step_wave.inl (wavefront primitives),
// step_narrowphase.inl (cooperative SAT / clipping / manifolds), step_hbm.inl
// (the step out of HBM + shared pieces), step_lds.inl (the step with the world
// in LDS), step_order.inl (the order / frame kernel in front of it).
//
// Worlds are independent, and everything the physics step does is per world:
// find candidate pairs, then per substep integrate, collide, solve positions,
// derive velocities, solve velocities.  The reference GPU backend runs those as
// ~8 megakernel nodes per substep over globally compacted tables (atomically
// appended, then radix sorted by world).  Here a 64-lane wavefront owns a world
// for the whole step:
//   * lanes = bodies / candidates / contacts of that world; phases are
//     separated by workgroup-scope fences (a wave never leaves its CU), not by
//     kernel boundaries -- no launches, no sorts, no scans in between;
//   * candidates and contacts live in per-world segments of module-private
//     arrays (fixed stride), written in the CPU backend's order with wave
//     ballots + prefix sums, so XPBD's Gauss-Seidel sweep sees the CPU sequence;
//   * the sequential sweep itself is parallelised exactly: contact i must wait
//     only for earlier contacts that touch one of its non-static bodies, so
//     contacts are assigned dependency levels and each level runs in parallel.
//     Constraints of one level touch disjoint bodies, therefore commute bit for
//     bit with the sequential order of the CPU solver (xpbd.cpp:720-736).
//
// Requirement: every rigid-body archetype is grouped by world when the physics
// step starts (the simulator compacts after creating / destroying bodies, as
// the reference's GPU simulators do); otherwise kErrPhysics is raised.

using namespace narrowphase;


#include "step_wave.inl"
#include "step_narrowphase.inl"
#include "step_hbm.inl"
#include "step_lds.inl"
#include "step_order.inl"

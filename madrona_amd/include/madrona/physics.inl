#pragma once

#include <vector>
#include <madrona/phys_impl/narrowphase.hpp>
#include <madrona/phys_impl/gjk.hpp>
#include <madrona/phys_impl/xpbd.hpp>

namespace madrona::phys {

struct PhysicsSystemState {
    float deltaT;
    float h;
    math::Vector3 g;
    float gMagnitude;
    float restitutionThreshold;
    uint32_t contactArchetypeID;
    uint32_t jointArchetypeID;
};

struct CandidateTemporary : Archetype<CandidateCollision> {};

#include <madrona/phys_impl/tgs.hpp>

namespace xpbd {

// The reference keeps its Query objects here (xpbd.cpp:21-24); this backend
// resolves queries when the graph is built, the singleton is kept so that the
// same number of singleton entities (hence the same entity ids) exist.
struct SolverState {
    uint32_t unused[8];
};

}

// ---------------------------------------------------------------------------
// Device-global scratch of the physics module (ecs_state::moduleData[0]),
// allocated by setupPhysicsStepTasks.  Candidates and contacts are emitted in
// the CPU backend's order -- (body archetype, row, traversal order) -- by
// counting per creator row, scanning, and filling, instead of appending with
// atomics in arrival order: XPBD is Gauss-Seidel over contacts
// in table order, so the order is part of the result.
// ---------------------------------------------------------------------------
struct PhysicsScratch {
    static constexpr uint32_t maxBodyArchetypes = MWHIP_SCAN_MAX_SEGMENTS;

    uint32_t numBodyArchetypes;
    uint32_t bodyArchetypes[maxBodyArchetypes];
    uint32_t *bodyCounts[maxBodyArchetypes];    // per body row: #candidates -> offset

    uint32_t candidateArchetype;
    uint32_t contactArchetype;
    uint32_t jointArchetype;

    // fused per-world step (phys_impl/world_step.inl): fixed-stride per-world
    // segments, world w owns [w * stride, (w + 1) * stride)
    uint32_t candidatesPerWorld;
    uint32_t contactsPerWorld;
    CandidateCollision *worldCandidates;
    ContactConstraint *worldContacts;
    float *worldLambdas;
    // per world: world-space planes + vertices of one hull, for the pairs of the
    // LDS step kernels whose faces outgrow the LDS clipping scratch (taken one
    // lane at a time; phys_impl/narrowphase.hpp: 128 elements each)
    static constexpr uint32_t hullScratchBytes = 128u * (16u + 12u);
    char *worldHullScratch;
};

// What a step of ANY world needs to find its rows, resolved once per launch
// (physicsOrderKernel, the single-workgroup kernel in front of the step) instead
// of once per world by every wavefront of the step kernel: the addresses of the
// rigid-body tables' columns and world ranges, of the singleton columns, of the
// joint table and the entity store, and the object manager.  A wavefront of
// the step kernel runs alone on its SIMD with nothing to hide a cache miss
// behind, every kernel starts with cold caches, and a dependent miss is ~1.5 us
// here: through the tables (table header -> column address -> row, singleton ->
// object -> array -> element) loading a world was a chain of ~20 of them, with
// the frame it is 5 (world -> row ranges / tree -> rows -> what the rows name
// -> the leaf's slot in its parent node).
struct PhysicsFrame {
    static constexpr uint32_t maxArchetypes = PhysicsScratch::maxBodyArchetypes;
    // Entity, WorldID, the RigidBody bundle (RGDCols), the XPBD solver state
    static constexpr uint32_t numColumns = 14;

    uint32_t numArchetypes;
    uint32_t unsorted;                  // some rigid-body table needs its world sort
    uint32_t archetype[maxArchetypes];
    const int32_t *worldOffsets[maxArchetypes];
    const int32_t *worldCounts[maxArchetypes];
    void *columns[maxArchetypes][numColumns];
    // singleton columns (indexed by world)
    const broadphase::BVH *trees;
    const PhysicsSystemState *systemStates;
    const ObjectData *objectData;
    const ObjectManager *objMgr;        // world 0's manager and a copy of it
    ObjectManager objMgrCopy;
    // the joint table (sorted by world right before the step)
    const int32_t *jointOffsets;
    const int32_t *jointCounts;
    const JointConstraint *joints;
    const mwhip::EntitySlot *entities;
};

// node data of the fused per-world step kernel
struct PhysicsStepParams {
    int32_t numSubsteps;
    // LDS step kernels: contacts a world may hold per substep before it is
    // handed to the HBM kernel; 0 = what the block has room for
    // (MADRONA_MWHIP_PHYS_LDS_CONTACTS: lower it to exercise the fallback)
    uint32_t contactCap;
    // LDS step kernels: what every world cost the last time it was stepped
    // (100 MHz ticks), and the worlds sorted by it, heaviest first
    // (physicsOrderKernel)
    uint32_t *worldCost;
    int32_t *worldOrder;
    // LDS step kernels: [0] = how many worlds of this step the instantiation
    // could not hold (more bodies than MAXB, more contacts in a substep than
    // the block has room for), [1 ..] = their indices; zeroed by
    // physicsOrderKernel, filled by the LDS kernel, stepped by
    // physicsStepKernel in fallback mode right behind it
    int32_t *fallbackList;
};

// The step node's data: the parameters and, right behind them, the frame -- a
// kernel has the address of both from its arguments, so the frame's words are
// one load away, like the parameters (behind a pointer in the parameters they
// were two: 2 us per pair of worlds in front of everything else).
struct PhysicsStepNode {
    PhysicsStepParams params;
    PhysicsFrame frame;
};

namespace detail {

// per-world row budget of the physics temporaries, overridable from the
// environment (tables hold 2x the hint per world)
MADRONA_HOST_API inline CountT capacityHint(const char *env_name,
                                            CountT fallback)
{
#if MADRONA_ON_HOST
    const char *v = getenv(env_name);
    if (v != nullptr && atol(v) > 0) {
        return (CountT)atol(v);
    }
#else
    (void)env_name;
#endif
    return fallback;
}

MADRONA_HD inline PhysicsScratch *scratch(mwhip::EcsState *S)
{
    return (PhysicsScratch *)S->moduleData[0];
}

// Visits, in BVH traversal order, every (other body, #primitive pairs) that
// body `e` must be tested against (reference findIntersectingEntry,
// broadphase.cpp:930-993): each unordered pair once (lower entity id owns it),
// static-static pairs skipped.
template <typename Fn>
MADRONA_HD inline void forEachCandidate(Context &ctx, Entity e,
                                        broadphase::LeafID leaf_id, Loc a_loc,
                                        Fn &&fn)
{
    broadphase::BVH &bvh = ctx.singleton<broadphase::BVH>();
    const ObjectManager &obj_mgr = *ctx.singleton<ObjectData>().mgr;

    bool a_is_static =
        ctx.getDirect<ResponseType>(RGDCols::ResponseType, a_loc) ==
        ResponseType::Static;
    base::ObjectID a_obj =
        ctx.getDirect<base::ObjectID>(RGDCols::ObjectID, a_loc);
    CountT a_num_prims = (CountT)obj_mgr.rigidBodyPrimitiveCounts[a_obj.idx];

    bvh.findLeafIntersecting(leaf_id, [&](Entity other) {
        if (e.id < other.id) {
            Loc b_loc = ctx.loc(other);

            if (a_is_static &&
                ctx.getDirect<ResponseType>(RGDCols::ResponseType, b_loc) ==
                    ResponseType::Static) {
                return;
            }

            base::ObjectID b_obj =
                ctx.getDirect<base::ObjectID>(RGDCols::ObjectID, b_loc);
            CountT b_num_prims =
                (CountT)obj_mgr.rigidBodyPrimitiveCounts[b_obj.idx];

            fn(b_loc, a_num_prims, b_num_prims);
        }
    });
}

}

// ---------------------------------------------------------------------------
// BVH ray casts (need the ObjectManager types, hence defined here)
// ---------------------------------------------------------------------------
namespace broadphase {

namespace detail {

// ray in the plane's frame: normal (0, 0, 1), d = 0
MADRONA_HD inline bool traceRayIntoPlane(math::Vector3 ray_o,
                                         math::Vector3 ray_d,
                                         float t_min, float t_max,
                                         float *hit_t,
                                         math::Vector3 *hit_normal)
{
    float denom = ray_d.z;
    if (denom == 0) {
        return false;
    }

    float t = -ray_o.z / denom;
    if (t < t_min || t > t_max) {
        return false;
    }

    *hit_t = t;
    *hit_normal = math::Vector3 { 0, 0, 1 };
    return true;
}

// Clips the ray against the hull's half-spaces (Ericson, RTCD 5.3.8, with
// outward normals); a ray that only meets back faces is a miss.
MADRONA_HD inline bool traceRayIntoConvexPolyhedron(
    const geo::HalfEdgeMesh &convex_mesh,
    math::Vector3 ray_o, math::Vector3 ray_d,
    float t_min, float t_max,
    float *hit_t, math::Vector3 *hit_normal)
{
    float tfirst = t_min;
    float tlast = t_max;

    math::Vector3 closest_normal = math::Vector3::zero();

    // The planes a batch at a time, all loads of a batch on their way before
    // the first plane is clipped against (a cube is one batch): fetched where
    // they are used -- through a generic pointer, between the early exits --
    // every plane was a round trip of its own, and the clip of a six-sided box
    // six L2 latencies in a row with nothing else for the lane to do.
    const CountT num_faces = (CountT)convex_mesh.numFaces;
    const geo::Plane *face_planes = convex_mesh.facePlanes;
    constexpr CountT plane_batch = 6;
    for (CountT first_face = 0; first_face < num_faces; first_face += plane_batch) {
    geo::Plane batch_planes[plane_batch];
MADRONA_UNROLL
    for (CountT j = 0; j < plane_batch; j++) {
        const CountT f = first_face + j < num_faces ? first_face + j : num_faces - 1;
        batch_planes[j] = mwhip::loadGlobal(face_planes + f);
    }
MADRONA_UNROLL
    for (CountT j = 0; j < plane_batch; j++) {
        if (first_face + j >= num_faces) {
            break;
        }
        const geo::Plane plane = batch_planes[j];

        float denom = dot(plane.normal, ray_d);
        float neg_dist = plane.d - dot(plane.normal, ray_o);

        if (denom == 0.0f) {
            // parallel to the face: outside its half-space means no hit at all
            if (neg_dist < 0.0f) {
                return false;
            }
        } else {
            float t = neg_dist / denom;
            if (denom < 0.0f) {
                if (t >= tfirst) {      // entering
                    tfirst = t;
                    closest_normal = plane.normal;
                }
            } else {
                if (t <= tlast) {       // leaving
                    tlast = t;
                }
            }

            if (tfirst > tlast) {
                return false;
            }
        }
    }
    }

    if (closest_normal.x == 0 && closest_normal.y == 0 &&
            closest_normal.z == 0) {
        return false;
    }

    *hit_t = tfirst;
    *hit_normal = closest_normal;
    return true;
}

}

// What a leaf test needs of its leaf that does not depend on the ray's
// direction: the leaf's transform and primitives and the ray ORIGIN in the
// object's frame -- the same for every ray that starts at the same point (the
// 30 rays of an agent's lidar).  Same expressions as the reference's
// traceRayIntoLeaf (broadphase.cpp), evaluated once per (origin, leaf).
struct BVH::RayLeaf {
    math::Quat rot;             // leaf_transforms_[leaf].rot
    math::Diag3x3 scale;
    math::Vector3 objRayO;      // rot^-1 (o - pos) / scale
    int32_t primOffset;
    int32_t numPrims;
    math::AABB box;             // leaf_aabbs_[leaf]: what the traversal culls by
    int32_t leafIdx;
};

BVH::RayLeaf BVH::rayLeaf(int32_t leaf_idx, math::Vector3 world_ray_o) const
{
    using namespace math;

    // (global loads, the independent ones together: mwhip::loadGlobal)
    const ObjectManager *obj_mgr = mwhip::loadGlobal(&obj_mgr_);
    const base::ObjectID *leaf_obj_ids = mwhip::loadGlobal(&leaf_obj_ids_);
    const LeafTransform *leaf_transforms = mwhip::loadGlobal(&leaf_transforms_);
    const AABB *leaf_aabbs = mwhip::loadGlobal(&leaf_aabbs_);
    const uint32_t *prim_offsets =
        mwhip::loadGlobal(&obj_mgr->rigidBodyPrimitiveOffsets);
    const uint32_t *prim_counts =
        mwhip::loadGlobal(&obj_mgr->rigidBodyPrimitiveCounts);
    base::ObjectID obj_id = mwhip::loadGlobal(leaf_obj_ids + leaf_idx);
    LeafTransform leaf_txfm = mwhip::loadGlobal(leaf_transforms + leaf_idx);
    const AABB leaf_box = mwhip::loadGlobal(leaf_aabbs + leaf_idx);

    Quat rot_to_local = leaf_txfm.rot.inv();

    Vector3 obj_ray_o = rot_to_local.rotateVec(world_ray_o - leaf_txfm.pos);
    obj_ray_o.x /= leaf_txfm.scale.d0;
    obj_ray_o.y /= leaf_txfm.scale.d1;
    obj_ray_o.z /= leaf_txfm.scale.d2;

    return RayLeaf {
        leaf_txfm.rot, leaf_txfm.scale, obj_ray_o,
        (int32_t)mwhip::loadGlobal(prim_offsets + obj_id.idx),
        (int32_t)mwhip::loadGlobal(prim_counts + obj_id.idx),
        leaf_box, leaf_idx,
    };
}

bool BVH::traceRayIntoLeaf(int32_t leaf_idx,
                           math::Vector3 world_ray_o,
                           math::Vector3 world_ray_d,
                           float t_min,
                           float t_max,
                           float *hit_t,
                           math::Vector3 *hit_normal)
{
    return traceRayIntoLeaf(rayLeaf(leaf_idx, world_ray_o), world_ray_d, t_min,
                            t_max, hit_t, hit_normal);
}

bool BVH::traceRayIntoLeaf(const RayLeaf &leaf,
                           math::Vector3 world_ray_d,
                           float t_min,
                           float t_max,
                           float *hit_t,
                           math::Vector3 *hit_normal)
{
    using namespace math;

    struct { Quat rot; Diag3x3 scale; } leaf_txfm { leaf.rot, leaf.scale };
    const Vector3 obj_ray_o = leaf.objRayO;

    Vector3 obj_ray_d = leaf_txfm.rot.inv().rotateVec(world_ray_d);
    obj_ray_d.x /= leaf_txfm.scale.d0;
    obj_ray_d.y /= leaf_txfm.scale.d1;
    obj_ray_d.z /= leaf_txfm.scale.d2;

    Diag3x3 inv_obj_ray_d = Diag3x3::fromVec(1.f / obj_ray_d);

    Vector3 obj_hit_normal = Vector3::zero();

    CountT prim_offset = (CountT)leaf.primOffset;
    CountT num_prims = (CountT)leaf.numPrims;

    // (the primitive's box and its header in one round of loads, the planes of
    // a hull in the next: traceRayIntoConvexPolyhedron)
    const ObjectManager *obj_mgr = mwhip::loadGlobal(&obj_mgr_);
    const AABB *prim_aabbs = mwhip::loadGlobal(&obj_mgr->primitiveAABBs);
    const CollisionPrimitive *prims =
        mwhip::loadGlobal(&obj_mgr->collisionPrimitives);

    bool hit_leaf = false;
    for (CountT i = 0; i < num_prims; i++) {
        CountT prim_idx = prim_offset + i;

        AABB prim_aabb = mwhip::loadGlobal(prim_aabbs + prim_idx);
        const CollisionPrimitive *prim = prims + prim_idx;
        const CollisionPrimitive::Type prim_type = mwhip::loadGlobal(&prim->type);
        // (the bytes of a hull whatever the type: only read if it is one)
        const geo::HalfEdgeMesh prim_mesh =
            mwhip::loadGlobal(&prim->hull.halfEdgeMesh);
        if (!prim_aabb.rayIntersects(obj_ray_o, inv_obj_ray_d, 0.f, t_max)) {
            continue;
        }

        bool hit_prim = false;
        if (prim_type == CollisionPrimitive::Type::Hull) {
            hit_prim = detail::traceRayIntoConvexPolyhedron(
                prim_mesh, obj_ray_o, obj_ray_d, t_min, t_max,
                hit_t, &obj_hit_normal);
        } else if (prim_type == CollisionPrimitive::Type::Plane) {
            hit_prim = detail::traceRayIntoPlane(
                obj_ray_o, obj_ray_d, t_min, t_max, hit_t, &obj_hit_normal);
        }   // spheres cannot be ray cast (the reference asserts)

        if (hit_prim) {
            hit_leaf = true;
            t_max = *hit_t;
        }
    }

    if (!hit_leaf) {
        return false;
    }

    *hit_normal = leaf_txfm.rot.rotateVec(obj_hit_normal);
    return true;
}

#if defined(__HIPCC__)
// LDS of the rays that share an origin (traceRayShared): the ray-independent
// part of the leaf tests of one 64-leaf window, per group of 32 lanes.
struct BVH::RayGroupScratch {
    RayLeaf leaves[64];
};

// The scratch of the calling lane's group of 32 in a 1-D workgroup of at most
// 256 threads (what a CustomParallelForNode<..., 32, 1, ...> runs in); nullptr
// in any other block shape.  Only kernels that call this carry the LDS.
__device__ inline BVH::RayGroupScratch *rayGroupScratch()
{
    constexpr uint32_t groups_per_block = 8;
    __shared__ BVH::RayGroupScratch scratch[groups_per_block];
    if (blockDim.x > 32u * groups_per_block || blockDim.y != 1u ||
            blockDim.z != 1u) {
        return nullptr;
    }
    return &scratch[threadIdx.x / 32u];
}
#endif

Entity BVH::traceRay(math::Vector3 o,
                     math::Vector3 d,
                     float *out_hit_t,
                     math::Vector3 *out_hit_normal,
                     float t_max)
{
    return traceRayImpl<false>(nullptr, o, d, out_hit_t, out_hit_normal, t_max);
}

Entity BVH::traceRayShared(RayGroupScratch *scratch,
                           math::Vector3 o,
                           math::Vector3 d,
                           float *out_hit_t,
                           math::Vector3 *out_hit_normal,
                           float t_max)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (scratch != nullptr) {
        return traceRayImpl<true>(scratch, o, d, out_hit_t, out_hit_normal, t_max);
    }
#endif
    (void)scratch;
    return traceRayImpl<false>(nullptr, o, d, out_hit_t, out_hit_normal, t_max);
}

// SharedOrigin: the caller's group of 32 lanes offered LDS (traceRayShared);
// whether its rays really share origin and tree is still decided per call.
template <bool SharedOrigin>
Entity BVH::traceRayImpl([[maybe_unused]] RayGroupScratch *scratch,
                         math::Vector3 o,
                         math::Vector3 d,
                         float *out_hit_t,
                         math::Vector3 *out_hit_normal,
                         float t_max)
{
    using namespace math;

    Diag3x3 inv_d = Diag3x3::fromVec(d).inv();

    Entity closest_hit_entity = Entity::none();
    Vector3 closest_hit_normal = Vector3::zero();

    auto visitLeaf = [&](int32_t leaf_idx, const RayLeaf *shared_leaf) {
        float hit_t;
        Vector3 leaf_hit_normal;
        bool leaf_hit = shared_leaf != nullptr ?
            traceRayIntoLeaf(*shared_leaf, d, 0.f, t_max, &hit_t,
                             &leaf_hit_normal) :
            traceRayIntoLeaf(leaf_idx, o, d, 0.f, t_max, &hit_t,
                             &leaf_hit_normal);

        if (leaf_hit) {
            t_max = hit_t;
            closest_hit_entity =
                mwhip::loadGlobal(mwhip::loadGlobal(&leaf_entities_) + leaf_idx);
            closest_hit_normal = leaf_hit_normal;
        }
    };

    if (!force_rebuild_) {
        // Flat traversal.  The stack walk below meets the leaves in an order
        // fixed by the tree (culling removes visits, it never reorders them),
        // and a leaf is met iff the ray hits its slot box: every ancestor's
        // box contains it (refits only grow boxes) and the slab test is
        // monotone in the box, so the ancestors' tests cannot cull a leaf
        // whose own slot passes.  Visiting the leaves in the order recorded by
        // rebuild() therefore performs the same leaf tests in the same order
        // with the same t_max, without the dependent node -> child -> node
        // chain and without a stack.
        //
        // The box tested here is the leaf's OWN box (leaf_aabbs_), not its slot
        // in the parent node: slots only grow between rebuilds -- after a few
        // hundred steps of bodies moving about they cover much of the arena and
        // most rays "hit" most of them --, while a leaf test can only succeed
        // for a ray that passes the body's box, which contains the body with
        // 100 dt^2 (16 cm) to spare on every side (expandAABBWithMotion), and
        // is inside the slot and all its ancestors.  The leaves skipped this
        // way are ones the reference visits and finds nothing in; the hits,
        // their order and t_max are the same.  (One dependent load fewer, too.)
        //
        // Two passes per window of 64 leaves: every box against the initial
        // t_max (all rays of an agent read the same boxes: broadcast loads, no
        // divergence) leaves a bit mask of candidate leaves; then each ray
        // walks ITS OWN set bits, re-testing the box with the current t_max.
        // A wave then takes as many leaf-test steps as its busiest ray has
        // candidates (a handful), not one per leaf that any of its rays touches.
        const int32_t n = num_tree_leaves_;
        const int32_t *traversal_order = mwhip::loadGlobal(&dfs_leaves_);
        const AABB *leaf_boxes = mwhip::loadGlobal(&leaf_aabbs_);

        // Rays that share their origin (traceRayShared): the lanes of a
        // half-wavefront that are in this call together with the SAME origin
        // and tree (an agent's lidar: CustomParallelForNode<..., 32, 1, ...>,
        // lane = ray) compute the ray-independent part of every leaf test ONCE
        // per leaf -- a lane per leaf, into the caller's LDS scratch -- instead
        // of once per (ray, leaf): the leaf's transform and primitive range (a
        // chain of dependent loads) and the origin in the object's frame (a
        // quaternion rotation and three exact divisions of the ~15 a leaf test
        // costs).  Same expressions, same inputs, so the same bits (rayLeaf);
        // what a ray then does per candidate leaf only involves its direction.
        // The group's lanes re-read their active set at the top of every
        // window (a lane whose ray is done with a window early waits there:
        // wave_barrier), so the scratch of a window is only rewritten when
        // every lane of the group has left the previous one; the two halves of
        // a wavefront decide, fill and read independently (different trees and
        // leaf counts included: a half whose tree has fewer windows simply
        // leaves the loop earlier).
        [[maybe_unused]] bool share_origin = false;
        [[maybe_unused]] RayLeaf *group_leaves = nullptr;
#if defined(__HIP_DEVICE_COMPILE__)
        constexpr int32_t ray_group_lanes = 32;
        if constexpr (SharedOrigin) {
            const uint32_t lane = __builtin_amdgcn_mbcnt_hi(
                ~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            const uint64_t active = __builtin_amdgcn_ballot_w64(true);
            const uint32_t half_first = lane & ~(uint32_t)(ray_group_lanes - 1);
            const uint64_t half_mask =
                (active >> half_first) & ((1ull << ray_group_lanes) - 1ull);
            const int32_t leader =
                (int32_t)half_first + (int32_t)__builtin_ctzll(half_mask);
            const bool same =
                __shfl(o.x, leader, 64) == o.x && __shfl(o.y, leader, 64) == o.y &&
                __shfl(o.z, leader, 64) == o.z &&
                __shfl((uint32_t)(uintptr_t)this, leader, 64) ==
                    (uint32_t)(uintptr_t)this &&
                __shfl((uint32_t)((uintptr_t)this >> 32), leader, 64) ==
                    (uint32_t)((uintptr_t)this >> 32) &&
                __shfl((uint32_t)(uintptr_t)scratch, leader, 64) ==
                    (uint32_t)(uintptr_t)scratch;
            const uint64_t agree = __builtin_amdgcn_ballot_w64(same);
            const uint32_t num_active = (uint32_t)__builtin_popcountll(half_mask);
            // (worth it from a handful of rays on)
            share_origin = ((agree >> half_first) & half_mask) == half_mask &&
                num_active >= 8u;
            if (share_origin) {
                group_leaves = scratch->leaves;
            }
        }
#endif

        for (int32_t base = 0; base < n; base += 64) {
            const int32_t window = n - base < 64 ? n - base : 64;

#if defined(__HIP_DEVICE_COMPILE__)
            if (share_origin) {
                // the window's leaves over the group's active lanes
                const uint32_t lane = __builtin_amdgcn_mbcnt_hi(
                    ~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
                const uint64_t active = __builtin_amdgcn_ballot_w64(true);
                const uint32_t half_first = lane & ~(uint32_t)(ray_group_lanes - 1);
                const uint64_t half_mask =
                    (active >> half_first) & ((1ull << ray_group_lanes) - 1ull);
                const int32_t rank = (int32_t)__builtin_popcountll(
                    half_mask & ((1ull << (lane - half_first)) - 1ull));
                const int32_t num_active = (int32_t)__builtin_popcountll(half_mask);
                // (the previous window's entries are done with: every lane of
                // the group has left its candidate loop)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                for (int32_t j = rank; j < window; j += num_active) {
                    group_leaves[j] = rayLeaf(
                        mwhip::loadGlobal(traversal_order + base + j), o);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
#endif

            // (boxes fetched a group at a time: leaf index -> box is two
            // dependent round trips, paid once per group)
            uint64_t candidates = 0;
            constexpr int32_t group = 8;
#if defined(__HIP_DEVICE_COMPILE__)
            if (share_origin) {
                // (the boxes are in LDS with the rest of the leaves)
                for (int32_t g = 0; g < window; g += group) {
MADRONA_UNROLL
                    for (int32_t k = 0; k < group; k++) {
                        const int32_t j = g + k < window ? g + k : window - 1;
                        AABB box = group_leaves[j].box;
                        const bool hit = g + k < window &&
                            box.rayIntersects(o, inv_d, 0.f, t_max);
                        candidates |= (uint64_t)hit << ((g + k) & 63);
                    }
                }
            } else
#endif
            for (int32_t g = 0; g < window; g += group) {
                int32_t leaf[group];
MADRONA_UNROLL
                for (int32_t k = 0; k < group; k++) {
                    const int32_t j = g + k < window ? g + k : window - 1;
                    leaf[k] = mwhip::loadGlobal(traversal_order + base + j);
                }
                AABB box[group];
MADRONA_UNROLL
                for (int32_t k = 0; k < group; k++) {
                    box[k] = mwhip::loadGlobal(leaf_boxes + leaf[k]);
                }
MADRONA_UNROLL
                for (int32_t k = 0; k < group; k++) {
                    const bool hit = g + k < window &&
                        box[k].rayIntersects(o, inv_d, 0.f, t_max);
                    candidates |= (uint64_t)hit << ((g + k) & 63);
                }
            }

            // (until a leaf test shrinks t_max the mask IS the re-test: the box
            // is only fetched again after a hit)
            const float mask_t_max = t_max;
            while (candidates != 0) {
                const int32_t j = (int32_t)__builtin_ctzll(candidates);
                candidates &= candidates - 1;

#if defined(__HIP_DEVICE_COMPILE__)
                if (share_origin) {
                    const RayLeaf *shared_leaf = &group_leaves[j];
                    AABB leaf_box = shared_leaf->box;
                    if (t_max == mask_t_max ||
                            leaf_box.rayIntersects(o, inv_d, 0.f, t_max)) {
                        visitLeaf(shared_leaf->leafIdx, shared_leaf);
                    }
                    continue;
                }
#endif
                const int32_t leaf_idx =
                    mwhip::loadGlobal(traversal_order + base + j);
                bool visit = t_max == mask_t_max;
                if (!visit) {
                    AABB own_box = mwhip::loadGlobal(leaf_boxes + leaf_idx);
                    visit = own_box.rayIntersects(o, inv_d, 0.f, t_max);
                }
                if (visit) {
                    visitLeaf(leaf_idx, nullptr);
                }
            }
        }
    } else {
        // a rebuild is pending: walk whatever tree is there, like the reference
        NodeStack stack;
        stack.push(0);

        while (!stack.empty()) {
            const Node &node = nodes_[stack.pop()];
            for (CountT c = 0; c < 4; c++) {
                if (!node.hasChild(c)) {
                    continue;
                }

                if (!node.bounds(c).rayIntersects(o, inv_d, 0.f, t_max)) {
                    continue;
                }

                if (node.isLeaf(c)) {
                    visitLeaf(node.leafIDX(c), nullptr);
                } else {
                    stack.push(node.children[c]);
                }
            }
        }
    }

    if (closest_hit_entity == Entity::none()) {
        return Entity::none();
    }

    *out_hit_t = t_max;
    *out_hit_normal = closest_hit_normal;
    return closest_hit_entity;
}

}

// ---------------------------------------------------------------------------
// narrowphase: primitive-pair dispatch and contact generation (reference
// narrowphase.cpp narrowphaseDispatch :1214-1514, generateContacts :1516-1680,
// runNarrowphase :1682-1907, CPU flavour: one candidate per lane, hulls
// transformed to world space up front)
// ---------------------------------------------------------------------------
namespace narrowphase {

struct PrimitiveTransform {
    Vector3 pos;
    Quat rot;
    Diag3x3 scale;
};

// A candidate after primitive-type ordering (a's type <= b's type), ready for
// the type-specific test.
struct PairSetup {
    Loc aLoc;
    Loc bLoc;
    const CollisionPrimitive *aPrim;
    const CollisionPrimitive *bPrim;
    PrimitiveTransform a;
    PrimitiveTransform b;
    NarrowphaseTest test;
    bool aabbOverlap;       // false: the pair cannot touch
};

MADRONA_HD inline PairSetup setupPair(const ObjectManager &obj_mgr,
                                      Loc a_loc, Loc b_loc,
                                      uint32_t a_prim_idx, uint32_t b_prim_idx,
                                      PrimitiveTransform a_txfm,
                                      PrimitiveTransform b_txfm)
{
    const CollisionPrimitive *a_prim = &obj_mgr.collisionPrimitives[a_prim_idx];
    const CollisionPrimitive *b_prim = &obj_mgr.collisionPrimitives[b_prim_idx];

    uint32_t raw_type_a = static_cast<uint32_t>(a_prim->type);
    uint32_t raw_type_b = static_cast<uint32_t>(b_prim->type);

    if (raw_type_a > raw_type_b) {
        Loc tmp_loc = a_loc; a_loc = b_loc; b_loc = tmp_loc;
        const CollisionPrimitive *tmp_prim = a_prim;
        a_prim = b_prim; b_prim = tmp_prim;
        uint32_t tmp_idx = a_prim_idx; a_prim_idx = b_prim_idx;
        b_prim_idx = tmp_idx;
        uint32_t tmp_type = raw_type_a; raw_type_a = raw_type_b;
        raw_type_b = tmp_type;
        PrimitiveTransform tmp_txfm = a_txfm; a_txfm = b_txfm;
        b_txfm = tmp_txfm;
    }

    math::AABB a_obj_aabb = obj_mgr.primitiveAABBs[a_prim_idx];
    math::AABB b_obj_aabb = obj_mgr.primitiveAABBs[b_prim_idx];

    math::AABB a_world_aabb =
        a_obj_aabb.applyTRS(a_txfm.pos, a_txfm.rot, a_txfm.scale);
    math::AABB b_world_aabb =
        b_obj_aabb.applyTRS(b_txfm.pos, b_txfm.rot, b_txfm.scale);

    return PairSetup {
        a_loc, b_loc, a_prim, b_prim, a_txfm, b_txfm,
        NarrowphaseTest { raw_type_a | raw_type_b },
        a_world_aabb.intersects(b_world_aabb),
    };
}

MADRONA_HD inline PairSetup setupPair(Context &ctx,
                                      const ObjectManager &obj_mgr,
                                      const CandidateCollision &candidate)
{
    Loc a_loc = candidate.a;
    Loc b_loc = candidate.b;

    base::ObjectID a_obj =
        ctx.getDirect<base::ObjectID>(RGDCols::ObjectID, a_loc);
    base::ObjectID b_obj =
        ctx.getDirect<base::ObjectID>(RGDCols::ObjectID, b_loc);

    uint32_t a_prim_idx =
        obj_mgr.rigidBodyPrimitiveOffsets[a_obj.idx] + candidate.aPrim;
    uint32_t b_prim_idx =
        obj_mgr.rigidBodyPrimitiveOffsets[b_obj.idx] + candidate.bPrim;

    PrimitiveTransform a_txfm {
        ctx.getDirect<base::Position>(RGDCols::Position, a_loc),
        ctx.getDirect<base::Rotation>(RGDCols::Rotation, a_loc),
        Diag3x3(ctx.getDirect<base::Scale>(RGDCols::Scale, a_loc)),
    };
    PrimitiveTransform b_txfm {
        ctx.getDirect<base::Position>(RGDCols::Position, b_loc),
        ctx.getDirect<base::Rotation>(RGDCols::Rotation, b_loc),
        Diag3x3(ctx.getDirect<base::Scale>(RGDCols::Scale, b_loc)),
    };

    return setupPair(obj_mgr, a_loc, b_loc, a_prim_idx, b_prim_idx,
                     a_txfm, b_txfm);
}

MADRONA_HD inline void manifoldToContact(const Manifold &manifold,
                                         Loc ref_loc, Loc other_loc,
                                         ContactConstraint *out)
{
    out->ref = ref_loc;
    out->alt = other_loc;
    for (int i = 0; i < 4; i++) {
        out->points[i] = Vector4::fromVec3W(manifold.contactPoints[i],
                                            manifold.penetrationDepths[i]);
    }
    out->numPoints = manifold.numContactPoints;
    out->normal = manifold.normal;
}

MADRONA_HD inline void sphereToContact(const SphereContact &sphere,
                                       Loc a_loc, Loc b_loc,
                                       ContactConstraint *out)
{
    out->ref = b_loc;
    out->alt = a_loc;
    out->points[0] = Vector4::fromVec3W(sphere.pt, sphere.depth);
    out->points[1] = Vector4::zero();
    out->points[2] = Vector4::zero();
    out->points[3] = Vector4::zero();
    out->numPoints = 1;
    out->normal = sphere.normal;
}

// SAT feature -> manifold -> constraint (reference generateContacts,
// narrowphase.cpp:1516-1680).  tmp_a / tmp_b: clipping scratch, each large
// enough for the clipped incident polygon (Vector3s).
// (OutT, here and below: a ContactConstraint, or whatever manifoldToContact /
// sphereToContact have an overload for -- the LDS step's packed contacts)
template <typename HullA, typename HullB, typename OutT>
MADRONA_HD inline bool satToContact(const SATResult &sat,
                                    const HullA &a, const HullB &b,
                                    Loc a_loc, Loc b_loc,
                                    void *tmp_a, void *tmp_b,
                                    OutT *out)
{
    const Vector3 no_offset { 0, 0, 0 };
    const Quat no_rot { 1, 0, 0, 0 };

    if (sat.type == ContactType::SATFace) {
        uint32_t ref_face_idx_and_ref_mask = sat.contact.refFaceIdxOrEdgeIdxA;
        uint32_t incident_face_idx = sat.contact.incidentFaceIdxOrEdgeIdxB;

        uint32_t ref_face_idx = ref_face_idx_and_ref_mask & 0x7FFFFFFFu;
        bool a_is_ref = ref_face_idx == ref_face_idx_and_ref_mask;

        Plane ref_plane { sat.contact.normal, sat.contact.planeDOrSeparation };

        Manifold manifold = a_is_ref ?
            createFaceContact(ref_plane, (int32_t)ref_face_idx,
                (int32_t)incident_face_idx, a, b, tmp_a, tmp_b,
                no_offset, no_rot) :
            createFaceContact(ref_plane, (int32_t)ref_face_idx,
                (int32_t)incident_face_idx, b, a, tmp_a, tmp_b,
                no_offset, no_rot);

        // barely touching pairs can lose every clipped point to fp32
        if (manifold.numContactPoints == 0) {
            return false;
        }
        manifoldToContact(manifold, a_is_ref ? a_loc : b_loc,
                          a_is_ref ? b_loc : a_loc, out);
        return true;
    } else if (sat.type == ContactType::SATEdge) {
        Manifold manifold = createEdgeContact(
            sat.contact.normal, sat.contact.planeDOrSeparation,
            (int32_t)sat.contact.refFaceIdxOrEdgeIdxA,
            (int32_t)sat.contact.incidentFaceIdxOrEdgeIdxB,
            a, b, no_offset, no_rot);

        manifoldToContact(manifold, a_loc, b_loc, out);
        return true;
    }

    return false;
}

template <typename HullT, typename OutT>
// tmp_a / tmp_b hold one entry per vertex of the incident face; a caller with
// less room than the largest face passes its capacity and gets *too_big when
// the face SAT picked does not fit.
MADRONA_HD inline bool hullPlaneContact(const HullT &a_hull,
                                        const PrimitiveTransform &plane_txfm,
                                        Loc a_loc, Loc b_loc,
                                        void *tmp_a, void *tmp_b,
                                        OutT *out,
                                        CountT tmp_capacity = -1,
                                        bool *too_big = nullptr)
{
    constexpr Vector3 base_normal = { 0, 0, 1 };
    Vector3 plane_normal = plane_txfm.rot.rotateVec(base_normal);

    Plane plane { plane_normal, dot(plane_normal, plane_txfm.pos) };

    const SATResult sat = doSATPlane(plane, a_hull);
    if (sat.type != ContactType::SATPlane) {
        return false;
    }

    if (tmp_capacity >= 0) {
        CountT num_face_vertices = 0;
        uint32_t hedge_idx = a_hull.faceBaseHedge(
            (CountT)sat.contact.incidentFaceIdxOrEdgeIdxB);
        const uint32_t start_hedge_idx = hedge_idx;
        do {
            hedge_idx = a_hull.hedge(hedge_idx).next;
            num_face_vertices++;
        } while (hedge_idx != start_hedge_idx);

        if (num_face_vertices > tmp_capacity) {
            *too_big = true;
            return false;
        }
    }

    // the plane is always b and always the reference
    Manifold manifold = createFacePlaneContact(
        Plane { sat.contact.normal, sat.contact.planeDOrSeparation },
        (int32_t)sat.contact.incidentFaceIdxOrEdgeIdxB, a_hull,
        (Vector3 *)tmp_a, (float *)tmp_b,
        Vector3 { 0, 0, 0 }, Quat { 1, 0, 0, 0 });

    if (manifold.numContactPoints == 0) {
        return false;
    }
    manifoldToContact(manifold, b_loc, a_loc, out);
    return true;
}

template <typename OutT>
MADRONA_HD inline bool sphereSphereContact(const PairSetup &pair,
                                           OutT *out)
{
    float a_radius = pair.a.scale.d0 * pair.aPrim->sphere.radius;
    float b_radius = pair.b.scale.d0 * pair.bPrim->sphere.radius;

    Vector3 to_b = pair.b.pos - pair.a.pos;
    float dist = to_b.length();

    if (dist > a_radius + b_radius) {
        return false;
    }

    Vector3 normal = dist > 0.f ? to_b / dist : math::up;
    float penetration = a_radius + b_radius - dist;

    sphereToContact(SphereContact {
        normal, pair.a.pos + a_radius * normal, penetration,
    }, pair.aLoc, pair.bLoc, out);
    return true;
}

template <typename OutT>
MADRONA_HD inline bool spherePlaneContact(const PairSetup &pair,
                                          OutT *out)
{
    float sphere_radius = pair.a.scale.d0 * pair.aPrim->sphere.radius;

    constexpr Vector3 base_normal = { 0, 0, 1 };
    Vector3 plane_normal = pair.b.rot.rotateVec(base_normal);

    float d = plane_normal.dot(pair.b.pos);
    float t = plane_normal.dot(pair.a.pos) - d;

    float penetration = sphere_radius - t;
    if (penetration < 0) {
        return false;
    }

    sphereToContact(SphereContact {
        plane_normal, pair.a.pos - t * plane_normal, penetration,
    }, pair.aLoc, pair.bLoc, out);
    return true;
}

// Sphere (a) against hull (b).  `b_hull` is the hull in the frame centred on
// the sphere (translation b.pos - a.pos), so the sphere sits at the origin
// (reference narrowphaseDispatch, SphereHull case, narrowphase.cpp:1325-1409).
template <typename HullT, typename OutT>
MADRONA_HD inline bool sphereHullContact(const PairSetup &pair,
                                         const HullT &b_hull,
                                         OutT *out)
{
    float sphere_radius = pair.a.scale.d0 * pair.aPrim->sphere.radius;

    Vector3 to_hull_closest_pt;
    float hull_dist2 = gjk::hullClosestPointToOrigin(
        b_hull, 1e-10f, &to_hull_closest_pt);

    if (hull_dist2 > sphere_radius * sphere_radius) {
        return false;
    }

    SphereContact contact;
    if (hull_dist2 == 0.f) {
        // centre inside the hull: least-penetration face
        float max_sep = -FLT_MAX;
        Vector3 sep_normal = Vector3::zero();
        const CountT num_faces = b_hull.numFaces();
        for (CountT i = 0; i < num_faces; i++) {
            Plane plane = b_hull.plane(i);
            float face_dist = -plane.d;

            if (face_dist > max_sep) {
                max_sep = face_dist;
                sep_normal = plane.normal;
            }
        }

        // GJK said inside, SAT says (barely) outside
        if (max_sep > 0.f) {
            return false;
        }

        contact.normal = sep_normal;
        contact.pt = pair.a.pos + sep_normal * sphere_radius;
        contact.depth = -max_sep;
    } else {
        float to_hull_len = sqrtf(hull_dist2);

        float depth = sphere_radius - to_hull_len;
        Vector3 normal = to_hull_closest_pt / to_hull_len;

        contact.normal = -normal;
        contact.pt = pair.a.pos + normal * sphere_radius;
        contact.depth = depth;
    }

    sphereToContact(contact, pair.aLoc, pair.bLoc, out);
    return true;
}

// One primitive pair on one lane with the hulls transformed into caller
// scratch (the reference's CPU flavour: runNarrowphase, narrowphase.cpp:
// 1682-1907).  tmp_vertices / tmp_faces: max_tmp_elems entries each; the
// clipping scratch is the UNUSED TAIL of each buffer.  (It used to overlay the
// planes "once SAT is done" -- but the planes are read through Plane lvalues
// and the polygons written through Vector3 lvalues, so type-based alias
// analysis lets the compiler sink a plane load below the polygon stores; the
// device build did, for faces large enough to reach the plane it still needed.)
template <typename OutT>
MADRONA_HD inline bool collidePairStored(const PairSetup &pair,
                                         Vector3 *tmp_vertices,
                                         Plane *tmp_faces,
                                         CountT max_tmp_elems,
                                         OutT *out,
                                         bool *unsupported)
{
    if (!pair.aabbOverlap) {
        return false;
    }

    // clip_a: Vector3s behind the hulls' vertices, clip_b: behind their
    // planes; a clipped polygon has at most (vertices of both hulls) corners
    auto clipScratch = [&](CountT used_vertices, CountT used_faces,
                           void **clip_a, void **clip_b) {
        const CountT need = used_vertices;
        const CountT have_a = max_tmp_elems - used_vertices;
        const CountT have_b = (CountT)(((size_t)(max_tmp_elems - used_faces) *
            sizeof(Plane)) / sizeof(Vector3));
        *clip_a = tmp_vertices + used_vertices;
        *clip_b = tmp_faces + used_faces;
        return have_a >= need && have_b >= need;
    };
    void *clip_a = nullptr;
    void *clip_b = nullptr;

    switch (pair.test) {
    case NarrowphaseTest::SphereSphere:
        return sphereSphereContact(pair, out);
    case NarrowphaseTest::SpherePlane:
        return spherePlaneContact(pair, out);
    case NarrowphaseTest::HullHull: {
        const HalfEdgeMesh &a_he_mesh = pair.aPrim->hull.halfEdgeMesh;
        const HalfEdgeMesh &b_he_mesh = pair.bPrim->hull.halfEdgeMesh;

        if ((CountT)(a_he_mesh.numFaces + b_he_mesh.numFaces) > max_tmp_elems ||
            (CountT)(a_he_mesh.numVertices + b_he_mesh.numVertices) >
                max_tmp_elems) {
            *unsupported = true;
            return false;
        }
        if (!clipScratch(
                (CountT)(a_he_mesh.numVertices + b_he_mesh.numVertices),
                (CountT)(a_he_mesh.numFaces + b_he_mesh.numFaces),
                &clip_a, &clip_b)) {
            *unsupported = true;
            return false;
        }

        HullState a_hull = makeHullState(a_he_mesh, pair.a.pos, pair.a.rot,
            pair.a.scale, tmp_vertices, tmp_faces);
        HullState b_hull = makeHullState(b_he_mesh, pair.b.pos, pair.b.rot,
            pair.b.scale, tmp_vertices + a_he_mesh.numVertices,
            tmp_faces + a_he_mesh.numFaces);

        const SATResult sat = doSAT(a_hull, b_hull);
        return satToContact(sat, a_hull, b_hull, pair.aLoc, pair.bLoc,
                            clip_a, clip_b, out);
    }
    case NarrowphaseTest::HullPlane: {
        const HalfEdgeMesh &a_he_mesh = pair.aPrim->hull.halfEdgeMesh;

        if ((CountT)a_he_mesh.numFaces > max_tmp_elems ||
            (CountT)a_he_mesh.numVertices > max_tmp_elems) {
            *unsupported = true;
            return false;
        }
        if (!clipScratch((CountT)a_he_mesh.numVertices,
                         (CountT)a_he_mesh.numFaces, &clip_a, &clip_b)) {
            *unsupported = true;
            return false;
        }

        HullState a_hull = makeHullState(a_he_mesh, pair.a.pos, pair.a.rot,
            pair.a.scale, tmp_vertices, tmp_faces);

        return hullPlaneContact(a_hull, pair.b, pair.aLoc, pair.bLoc,
                                clip_a, clip_b, out);
    }
    case NarrowphaseTest::SphereHull: {
        const HalfEdgeMesh &b_he_mesh = pair.bPrim->hull.halfEdgeMesh;

        if ((CountT)b_he_mesh.numFaces > max_tmp_elems ||
            (CountT)b_he_mesh.numVertices > max_tmp_elems) {
            *unsupported = true;
            return false;
        }

        HullState b_hull = makeHullState(b_he_mesh, pair.b.pos - pair.a.pos,
            pair.b.rot, pair.b.scale, tmp_vertices, tmp_faces);

        return sphereHullContact(pair, b_hull, out);
    }
    case NarrowphaseTest::PlanePlane:   // planes are static, never paired
    default:
        *unsupported = true;
        return false;
    }
}

MADRONA_HD inline bool computeContact(Context &ctx,
                                      const ObjectManager &obj_mgr,
                                      const CandidateCollision &candidate,
                                      Vector3 *tmp_vertices, Plane *tmp_faces,
                                      CountT max_tmp_elems,
                                      ContactConstraint *out)
{
    PairSetup pair = setupPair(ctx, obj_mgr, candidate);

    bool unsupported = false;
    bool has_contact = collidePairStored(pair, tmp_vertices, tmp_faces,
                                         max_tmp_elems, out, &unsupported);

#if defined(__HIP_DEVICE_COMPILE__)
    if (unsupported) {
        mwhip::raiseError(ctx.getStateManager(), mwhip::kErrPhysics);
    }
#endif

    return has_contact;
}

}

// ---------------------------------------------------------------------------
// Systems that are plain ParallelFor nodes
// ---------------------------------------------------------------------------
namespace broadphase {

inline void updateLeafPositionsEntry(Context &ctx,
                                     const LeafID &leaf_id,
                                     const base::Position &pos,
                                     const base::Rotation &rot,
                                     const base::Scale &scale,
                                     const base::ObjectID &obj_id,
                                     const Velocity &vel)
{
    BVH &bvh = ctx.singleton<BVH>();
    ObjectManager &obj_mgr = *ctx.singleton<ObjectData>().mgr;
    math::AABB obj_aabb = obj_mgr.rigidBodyAABBs[obj_id.idx];

    bvh.updateLeafPosition(leaf_id, pos, rot, scale, vel.linear, obj_aabb);
}

// Leaf update + refit in one pass over the bodies (the reference runs them as
// two ParallelFor nodes, broadphase.cpp:892-1052): the refit of a leaf only
// needs that leaf's new box, and expands its ancestors with atomic min / max,
// so it can follow the update in the same thread.  A world whose tree is about
// to be rebuilt skips the refit: the rebuild derives every box from the leaf
// boxes, after which a refit changes nothing.
inline void updateLeafAndRefitEntry(Context &ctx,
                                    const LeafID &leaf_id,
                                    const base::Position &pos,
                                    const base::Rotation &rot,
                                    const base::Scale &scale,
                                    const base::ObjectID &obj_id,
                                    const Velocity &vel)
{
    BVH &bvh = ctx.singleton<BVH>();
    // (global loads: the chain manager -> table -> box runs next to the
    // tree's own rounds in updateLeafAndRefit instead of in front of them)
    const ObjectManager *obj_mgr =
        mwhip::loadGlobal(&ctx.singleton<ObjectData>().mgr);
    const math::AABB *body_aabbs = mwhip::loadGlobal(&obj_mgr->rigidBodyAABBs);
    math::AABB obj_aabb = mwhip::loadGlobal(body_aabbs + obj_id.idx);

    bvh.updateLeafAndRefit(leaf_id, pos, rot, scale, vel.linear, obj_aabb);
}

inline void updateBVHEntry(Context &, BVH &bvh)
{
    bvh.updateTree();
}

inline void refitEntry(Context &ctx, LeafID leaf_id)
{
    BVH &bvh = ctx.singleton<BVH>();
    bvh.refitLeaf(leaf_id, bvh.getLeafAABB(leaf_id));
}

}

}   // namespace madrona::phys

// What the leaf systems move per body (madrona::mwhip::systemIO,
// taskgraph.inl): the row's columns in, the leaf's box (24 B) and transform
// (40 B: position, rotation, scale) out; the refit adds the leaf's slot in its
// parent node (24 B, read + written when the box grew).  rigidBodyAABBs is a
// shared read-only table of the object manager: excluded.
template <> inline constexpr madrona::mwhip::SystemIOBytes
    madrona::mwhip::systemIO<madrona::phys::broadphase::updateLeafPositionsEntry> =
        madrona::mwhip::declareIO<
            madrona::mwhip::Reads<madrona::phys::broadphase::LeafID,
                madrona::base::Position, madrona::base::Rotation,
                madrona::base::Scale, madrona::base::ObjectID,
                madrona::phys::Velocity>,
            madrona::mwhip::Writes<madrona::math::AABB,
                madrona::mwhip::Times<float, 10>>>();
template <> inline constexpr madrona::mwhip::SystemIOBytes
    madrona::mwhip::systemIO<madrona::phys::broadphase::updateLeafAndRefitEntry> =
        madrona::mwhip::declareIO<
            madrona::mwhip::Reads<madrona::phys::broadphase::LeafID,
                madrona::base::Position, madrona::base::Rotation,
                madrona::base::Scale, madrona::base::ObjectID,
                madrona::phys::Velocity, madrona::math::AABB>,
            madrona::mwhip::Writes<madrona::math::AABB,
                madrona::mwhip::Times<float, 10>, madrona::math::AABB>>();

namespace madrona::phys {

namespace xpbd {

inline void substepRigidBodies(Context &ctx,
                               base::Position &pos,
                               base::Rotation &rot,
                               const Velocity &vel,
                               const base::ObjectID &obj_id,
                               ResponseType response_type,
                               ExternalForce &ext_force,
                               ExternalTorque &ext_torque,
                               SubstepPrevState &prev_state,
                               PreSolvePositional &presolve_pos,
                               PreSolveVelocity &presolve_vel)
{
    Vector3 x = pos;
    Quat q = rot;

    prev_state.prevPosition = x;
    prev_state.prevRotation = q;

    if (response_type == ResponseType::Static) {
        presolve_pos.x = x;
        presolve_pos.q = q;
        presolve_vel.v = Vector3::zero();
        presolve_vel.omega = Vector3::zero();
        return;
    }

    const PhysicsSystemState &physics_sys =
        ctx.singleton<PhysicsSystemState>();
    const ObjectManager &obj_mgr = *ctx.singleton<ObjectData>().mgr;
    const RigidBodyMetadata &metadata = obj_mgr.metadata[obj_id.idx];

    SubstepResult next = integrateBody(
        x, q, vel.linear, vel.angular, metadata.mass.invMass,
        metadata.mass.invInertiaTensor, ext_force, ext_torque, physics_sys.g,
        physics_sys.h, response_type == ResponseType::Dynamic);

    pos = next.x;
    rot = next.q;

    presolve_pos.x = next.x;
    presolve_pos.q = next.q;
    presolve_vel.v = next.v;
    presolve_vel.omega = next.omega;
}

inline void setVelocities(Context &ctx,
                          const base::Position &pos,
                          const base::Rotation &rot,
                          const SubstepPrevState &prev_state,
                          Velocity &vel)
{
    const PhysicsSystemState &physics_sys =
        ctx.singleton<PhysicsSystemState>();
    vel = deriveVelocity(pos, rot, prev_state, physics_sys.h);
}

}

// ---------------------------------------------------------------------------
// Custom kernels (device only)
// ---------------------------------------------------------------------------
#if defined(__HIPCC__)
namespace kernels {

using mwhip::EcsState;
using mwhip::TableHdr;

// broadphase, phase 1 and 3: per rigid-body row, count / write the candidate
// pairs it owns.  fill == false: bodyCounts[a][row] = #candidates;
// fill == true: write them at the scanned offset.
template <bool fill>
__global__ void __launch_bounds__(256)
candidateKernel(EcsState *S, void *, uint32_t, uint32_t)
{
    mwhip::TraceScope trace_scope(S);
    StateManager *state_mgr = static_cast<StateManager *>(S);
    PhysicsScratch *ps = detail::scratch(S);

    const int32_t tid = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    const int32_t stride = (int32_t)(gridDim.x * blockDim.x);

    TableHdr &cand_tbl = S->tables[ps->candidateArchetype];

    for (uint32_t a = 0; a < ps->numBodyArchetypes; a++) {
        const uint32_t archetype_id = ps->bodyArchetypes[a];
        TableHdr &tbl = S->tables[archetype_id];
        const int32_t num_rows = tbl.numRows;
        const Entity *entities = (const Entity *)tbl.columns[0];
        const WorldID *worlds = (const WorldID *)tbl.columns[1];
        const broadphase::LeafID *leaves =
            (const broadphase::LeafID *)tbl.columns[RGDCols::LeafID];
        uint32_t *counts = ps->bodyCounts[a];

        for (int32_t row = tid; row < num_rows; row += stride) {
            WorldID world_id = worlds[row];
            if (world_id.idx == -1) {
                if constexpr (!fill) counts[row] = 0;
                continue;
            }

            Context ctx = TaskGraph::makeContext<Context>(state_mgr, world_id);
            Loc a_loc { archetype_id, row };

            uint32_t n = 0;
            uint32_t out = fill ? counts[row] : 0u;
            detail::forEachCandidate(ctx, entities[row], leaves[row], a_loc,
                [&](Loc b_loc, CountT a_num_prims, CountT b_num_prims) {
                CountT total_checks = a_num_prims * b_num_prims;
                if constexpr (fill) {
                    for (CountT k = 0; k < total_checks; k++) {
                        uint32_t dst = out + n + (uint32_t)k;
                        if (dst >= (uint32_t)cand_tbl.capacity) {
                            break;      // scan already raised the overflow flag
                        }
                        ((Entity *)cand_tbl.columns[0])[dst] = Entity::none();
                        ((WorldID *)cand_tbl.columns[1])[dst] = world_id;
                        CandidateCollision &candidate = ((CandidateCollision *)
                            cand_tbl.columns[RGDCols::CandidateCollision])[dst];
                        candidate.a = a_loc;
                        candidate.b = b_loc;
                        candidate.aPrim = (uint32_t)(k / b_num_prims);
                        candidate.bPrim = (uint32_t)(k % b_num_prims);
                    }
                }
                n += (uint32_t)total_checks;
            });

            if constexpr (!fill) counts[row] = n;
        }
    }
}

#include <madrona/phys_impl/world_step.inl>

// LDS of one staged tree rebuild (<= 64 leaves)
struct BvhRebuildStaging {
    static constexpr int32_t maxLeaves = 64;
    static constexpr int32_t maxNodes = 21 + maxLeaves;    // numInternalNodes(64)
    alignas(16) char nodes[maxNodes * broadphase::BVH::nodeBytes];
    math::AABB leafAABBs[maxLeaves];
    math::Vector3 leafCenters[maxLeaves];
    uint32_t leafParents[maxLeaves];
    int32_t sortedLeaves[maxLeaves];
    int32_t traversalOrder[maxLeaves];
    union {
        // (the stack machine's stack / the breadth-first build's range records)
        broadphase::BVH::RebuildStackEntry
            buildStack[broadphase::BVH::rebuildStackSize];
        broadphase::BVH::SegmentedScratch segmented;
    };
    int32_t numNodes;
};

// The rebuild of ONE world's tree by its wavefront (the tree asked for it: a
// reset re-registered its bodies): the build runs out of LDS on a rebased copy
// of the tree (BVH::rebased) and the arrays go back to HBM once.  The top-down
// build is a chain of a few thousand dependent accesses to a few KB; from HBM
// that chain is pure latency (measured 230 us per step for ~40 rebuilding
// worlds with the reference's one-thread-per-world updateBVHEntry ParallelFor,
// broadphase.cpp:1003-1004).  Leaf boxes must be current (and visible to this
// wavefront).
// Check (tests, bvhUpdateKernel<true>): the breadth-first build's tree is
// compared, word for word, with the stack machine's from the same leaves;
// kErrPhysics if they differ.
template <bool Check = false>
__device__ inline void rebuildTreeStaged(broadphase::BVH &bvh, uint32_t lane,
                                         BvhRebuildStaging &staging,
                                         BvhRebuildStaging *check_staging = nullptr,
                                         EcsState *S = nullptr)
{
    const int32_t num_leaves = bvh.numLeaves();
    if (num_leaves > BvhRebuildStaging::maxLeaves ||
            bvh.nodeCapacity() > BvhRebuildStaging::maxNodes) {
        if (lane == 0) {
            bvh.updateTree();       // too large to stage: build in place
        }
        return;
    }

    for (int32_t i = (int32_t)lane; i < num_leaves; i += 64) {
        math::AABB aabb = bvh.rawLeafAABBs()[i];
        staging.leafAABBs[i] = aabb;
        staging.leafCenters[i] = (aabb.pMin + aabb.pMax) / 2.f;
        staging.sortedLeaves[i] = bvh.rawSortedLeaves()[i];
    }
    wave::phaseFence();

    broadphase::BVH local = bvh.rebased(staging.nodes,
        staging.leafAABBs, staging.leafParents, staging.sortedLeaves,
        staging.traversalOrder, staging.leafCenters);
#ifdef MADRONA_PHYS_SERIAL_BVH_REBUILD
    if (lane == 0) {
        staging.numNodes = local.rebuildStaged(staging.buildStack);
    }
    wave::phaseFence();
    const int32_t num_nodes = staging.numNodes;
#else
    // breadth first, every range of a level at once (broadphase.inl); the
    // stack machine only if the tree needs more range records than the scratch
    // holds (not expected: a range is a node, and the reference sizes its node
    // array with the same bound)
#ifdef MADRONA_PHYS_STACK_REBUILD
    int32_t num_nodes = -1;     // (measurement builds: rounds 2-4's build)
#else
    int32_t num_nodes = local.rebuildStagedSegmented(lane, &staging.segmented);
    wave::phaseFence();
#endif
    if (num_nodes < 0) {
        for (int32_t i = (int32_t)lane; i < num_leaves; i += 64) {
            staging.sortedLeaves[i] = bvh.rawSortedLeaves()[i];
        }
        wave::phaseFence();
        num_nodes = local.rebuildStagedWave(lane, staging.buildStack);
        wave::phaseFence();
    }
    if constexpr (Check) {
        BvhRebuildStaging &other = *check_staging;
        for (int32_t i = (int32_t)lane; i < num_leaves; i += 64) {
            other.leafAABBs[i] = staging.leafAABBs[i];
            other.leafCenters[i] = staging.leafCenters[i];
            other.sortedLeaves[i] = bvh.rawSortedLeaves()[i];
        }
        wave::phaseFence();
        broadphase::BVH reference_tree = bvh.rebased(other.nodes,
            other.leafAABBs, other.leafParents, other.sortedLeaves,
            other.traversalOrder, other.leafCenters);
        const int32_t ref_nodes =
            reference_tree.rebuildStagedWave(lane, other.buildStack);
        wave::phaseFence();
        bool same = ref_nodes == num_nodes;
        // (nodes the tree uses: the root's subtree, numbered 0 .. count - 1)
        int32_t used = 0;
        {
            // count by walking the stack machine's tree: every node has a
            // parent id below its own; the largest referenced child id + 1
            used = 1;
            for (int32_t n = 0; n < ref_nodes; n++) {
                const int32_t *words = (const int32_t *)(
                    other.nodes + (size_t)n * broadphase::BVH::nodeBytes);
                if (n >= used) break;
                for (int32_t c = 0; c < 4; c++) {
                    const int32_t child = words[24 + c];
                    if (child >= 0 && child + 1 > used) used = child + 1;
                }
            }
        }
        const uint32_t words_used =
            (uint32_t)used * (broadphase::BVH::nodeBytes / 4);
        for (uint32_t i = lane; i < words_used; i += 64) {
            same = same && ((const uint32_t *)staging.nodes)[i] ==
                           ((const uint32_t *)other.nodes)[i];
        }
        for (int32_t i = (int32_t)lane; i < num_leaves; i += 64) {
            same = same && staging.leafParents[i] == other.leafParents[i] &&
                staging.sortedLeaves[i] == other.sortedLeaves[i] &&
                staging.traversalOrder[i] == other.traversalOrder[i];
        }
        if (__builtin_amdgcn_ballot_w64(!same) != 0ull) {
            mwhip::raiseError(S, mwhip::kErrPhysics);
        }
    }
#endif

    waveCopyDwords(lane, 64u, (uint32_t *)bvh.rawNodes(),
        (const uint32_t *)staging.nodes,
        (uint32_t)num_nodes * (broadphase::BVH::nodeBytes / 4));
    for (int32_t i = (int32_t)lane; i < num_leaves; i += 64) {
        bvh.rawLeafParents()[i] = staging.leafParents[i];
        bvh.rawSortedLeaves()[i] = staging.sortedLeaves[i];
        bvh.rawTraversalOrder()[i] = staging.traversalOrder[i];
    }
    if (lane == 0) {
        bvh.finishRebuild(num_nodes);
    }
    wave::phaseFence();
}

// Leaf update + refit of EVERY body, a wavefront per world, a lane per leaf
// (what setupBroadphaseTasks / setupPostIntegrationTasks run: the reference's
// updateLeafPositionsEntry + refitEntry ParallelFor nodes, broadphase.cpp:
// 440-647 + 892-1052).  Rounds 1-4 ran it as a ParallelFor over the body rows
// (updateLeafAndRefitEntry): every ROW went tree -> its arrays -> the leaf's
// parent -> the slot, and manager -> table -> box, 16-24 us per launch, three
// launches per step.  Here the tree's words are fetched once per world and the
// lanes go leaf -> entity -> row -> components in batched rounds:
//   1  the tree (the singleton column is resolved once per wavefront)
//   2  its arrays, the object manager, the expansions, the rebuild flag
//   3  per lane: the leaf's entity, object id and parent; the manager's box table
//   4  the entity's slot in the store, the object's box, the leaf's slot bounds
//   5  the row's five column addresses
//   6  position, rotation, scale, object id, velocity
// then the same arithmetic and the same stores (BVH::applyLeafUpdate: the
// leaf's own slot plain, ancestors that have to grow with atomic min / max --
// order independent, so the boxes are those of any other schedule).  A leaf
// whose entity is gone (destroyed without a reset of the tree) is skipped: the
// ParallelFor had no row for it either.
// WithRebuild (setupBroadphaseTasks: the reference's updateBVHEntry node follows
// the leaf update there): a world whose tree asked for a rebuild builds it right
// behind its leaf update, in the same launch -- the ~40 us latency of a build (a
// step resets ~1 world in 200) then runs beside the other worlds' refresh
// instead of in a launch of its own with the chip idle.  The staging costs
// 14 KB of LDS (11 wavefronts per CU instead of 32); setupPostIntegrationTasks
// has no rebuild behind it and takes the plain variant (no LDS).
// (86 registers: five wavefronts per SIMD.  Capped at 64 for eight -- every world
// resident at once -- it spills 26 dwords and takes 32.6 us instead of 21.4:
// profiles/r06_refresh_variants.jsonl.  The node is bound by its scattered
// bytes, not by wavefronts in flight.)
template <bool WithRebuild>
__global__ void __launch_bounds__(64)
bvhRefreshKernel(EcsState *S, void *, uint32_t, uint32_t)
{
    mwhip::TraceScope trace_scope(S);
    using namespace base;
    using broadphase::BVH;
    using mwhip::loadGlobal;
    using mwhip::loadInvariant;

    StateManager *state_mgr = static_cast<StateManager *>(S);
    const uint32_t lane = wave::laneID();
    const int32_t num_worlds = S->numWorlds;

    BVH *trees = state_mgr->getSingletonColumn<BVH>();
    const mwhip::EntitySlot *entities = mwhip::entitiesOf(S);
    void *const *col_ptrs = loadInvariant(&S->colPtr);
    const uint32_t num_slots = loadInvariant(&S->numComponentSlots);
    const uint32_t id_pos = TypeTracker::typeID<Position>();
    const uint32_t id_rot = TypeTracker::typeID<Rotation>();
    const uint32_t id_scale = TypeTracker::typeID<Scale>();
    const uint32_t id_obj = TypeTracker::typeID<ObjectID>();
    const uint32_t id_vel = TypeTracker::typeID<Velocity>();

    for (int32_t world = (int32_t)blockIdx.x; world < num_worlds;
         world += (int32_t)gridDim.x) {
        const BVH::RefreshView view = BVH::loadRefreshView(trees + world);
        const math::AABB *body_aabbs = nullptr;
        for (int32_t leaf = (int32_t)lane; leaf < view.numLeaves; leaf += 64) {
            // ---- round 3 ----
            if (body_aabbs == nullptr) {
                body_aabbs = loadGlobal(&view.objMgr->rigidBodyAABBs);
            }
            const Entity e = loadGlobal(view.leafEntities + leaf);
            const ObjectID leaf_obj = loadGlobal(view.leafObjIDs + leaf);
            uint32_t leaf_parent = 0;
            if (view.refit.refit) {
                leaf_parent = loadGlobal(view.refit.leafParents + leaf);
            }
            __builtin_amdgcn_sched_barrier(0);

            // ---- round 4 ----
            const mwhip::EntitySlot slot_of_e =
                loadGlobal(entities + (e.id >= 0 ? e.id : 0));
            math::AABB obj_aabb = loadGlobal(body_aabbs + leaf_obj.idx);
            math::AABB slot = math::AABB::invalid();
            if (view.refit.refit) {
                slot = BVH::loadSlotBounds(view.refit.nodes, leaf_parent);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (e.id < 0 || slot_of_e.gen != e.gen) {
                continue;
            }
            const uint32_t arch = slot_of_e.loc.archetype;
            const int32_t row = slot_of_e.loc.row;

            // ---- round 5 ----
            const uint32_t base = arch * num_slots;
            const Position *col_pos =
                (const Position *)loadInvariant(&col_ptrs[base + id_pos]);
            const Rotation *col_rot =
                (const Rotation *)loadInvariant(&col_ptrs[base + id_rot]);
            const Scale *col_scale =
                (const Scale *)loadInvariant(&col_ptrs[base + id_scale]);
            const ObjectID *col_obj =
                (const ObjectID *)loadInvariant(&col_ptrs[base + id_obj]);
            const Velocity *col_vel =
                (const Velocity *)loadInvariant(&col_ptrs[base + id_vel]);
            __builtin_amdgcn_sched_barrier(0);
            if (col_pos == nullptr || col_rot == nullptr || col_scale == nullptr ||
                    col_obj == nullptr || col_vel == nullptr) {
                continue;       // (not a rigid body row: the query would not match)
            }

            // ---- round 6 ----
            const Position pos = loadGlobal(col_pos + row);
            const Rotation rot = loadGlobal(col_rot + row);
            const Scale scale = loadGlobal(col_scale + row);
            const ObjectID obj_id = loadGlobal(col_obj + row);
            const Velocity vel = loadGlobal(col_vel + row);
            __builtin_amdgcn_sched_barrier(0);
            if (obj_id.idx != leaf_obj.idx) {
                // (the component was changed after the body was registered:
                // the row is what the reference's system reads)
                obj_aabb = loadGlobal(body_aabbs + obj_id.idx);
            }

            BVH::applyLeafUpdate(view.refit, leaf, leaf_parent, slot, pos, rot,
                                 scale, vel.linear, obj_aabb);
        }
        if constexpr (WithRebuild) {
            if (!view.refit.refit) {
                // (a rebuild is pending: the leaf boxes just stored are what it
                // builds from)
                __shared__ BvhRebuildStaging staging;
                wave::phaseFence();
                rebuildTreeStaged(trees[world], lane, staging);
            }
        }
    }
}

// BVH rebuild for the worlds that asked for one, as a launch of its own (behind
// the ParallelFor flavour of the leaf update, MADRONA_MWHIP_BVH_REFRESH=0; the
// refresh kernel rebuilds in place, bvhRefreshKernel<true>).
template <bool Check>
__global__ void __launch_bounds__(64)
bvhUpdateKernel(EcsState *S, void *, uint32_t, uint32_t)
{
    mwhip::TraceScope trace_scope(S);
    __shared__ BvhRebuildStaging staging;
    [[maybe_unused]] BvhRebuildStaging *check_staging = nullptr;
    if constexpr (Check) {
        __shared__ BvhRebuildStaging second;
        check_staging = &second;
    }

    StateManager *state_mgr = static_cast<StateManager *>(S);
    const uint32_t lane = wave::laneID();
    const int32_t num_worlds = S->numWorlds;

    for (int32_t world = (int32_t)blockIdx.x; world < num_worlds;
         world += (int32_t)gridDim.x) {
        Context ctx = TaskGraph::makeContext<Context>(
            state_mgr, WorldID { world }, true);
        broadphase::BVH &bvh = ctx.singleton<broadphase::BVH>();

        if (!bvh.needsRebuild()) {
            continue;
        }
        rebuildTreeStaged<Check>(bvh, lane, staging, check_staging, S);
    }
}
}
#endif // __HIPCC__

}

// ---------------------------------------------------------------------------
// PhysicsSystem: the simulator-facing API (reference src/physics/physics.cpp)
// ---------------------------------------------------------------------------
namespace madrona::phys::PhysicsSystem {

MADRONA_HD inline void init(Context &ctx,
                            ObjectManager *obj_mgr,
                            float delta_t,
                            CountT num_substeps,
                            math::Vector3 gravity,
                            CountT max_dynamic_objects,
                            Solver solver)
{
    broadphase::BVH &bvh = ctx.singleton<broadphase::BVH>();

    // leaves are expanded by 2 * delta_t * velocity plus room for
    // acceleration within the step
    constexpr float max_inst_accel = 100.f;
    new (&bvh) broadphase::BVH(
        obj_mgr, max_dynamic_objects, 2.f * delta_t,
        max_inst_accel * delta_t * delta_t);

    float h = delta_t / (float)num_substeps;
    float g_mag = gravity.length();

    PhysicsSystemState &state = ctx.singleton<PhysicsSystemState>();
    state.deltaT = delta_t;
    state.h = h;
    state.g = gravity;
    state.gMagnitude = g_mag;
    state.restitutionThreshold = 2.f * g_mag * h;
    // (reference physics.cpp:113-137: the solver's archetypes and its state)
    if (solver == Solver::TGS) {
        state.contactArchetypeID = TypeTracker::typeID<tgs::Contact>();
        state.jointArchetypeID = TypeTracker::typeID<tgs::Joint>();
        tgs::SolverState &solver_state = ctx.singleton<tgs::SolverState>();
        for (int i = 0; i < 8; i++) {
            solver_state.unused[i] = 0;
        }
    } else {
        state.contactArchetypeID = TypeTracker::typeID<xpbd::Contact>();
        state.jointArchetypeID = TypeTracker::typeID<xpbd::Joint>();
        xpbd::SolverState &solver_state = ctx.singleton<xpbd::SolverState>();
        for (int i = 0; i < 8; i++) {
            solver_state.unused[i] = 0;
        }
    }

    ctx.singleton<ObjectData>() = ObjectData { obj_mgr };
}

MADRONA_HD inline void reset(Context &ctx)
{
    broadphase::BVH &bvh = ctx.singleton<broadphase::BVH>();
    bvh.rebuildOnUpdate();
    bvh.clearLeaves();
}

MADRONA_HD inline broadphase::LeafID registerEntity(Context &ctx,
                                                    Entity e,
                                                    base::ObjectID obj_id)
{
    return ctx.singleton<broadphase::BVH>().reserveLeaf(e, obj_id);
}

template <typename Fn>
MADRONA_HD inline void findEntitiesWithinAABB(Context &ctx,
                                              math::AABB aabb,
                                              Fn &&fn)
{
    // BVH boxes are conservative (expanded for motion): confirm against the
    // body's actual hull extents (reference physics.inl:8-24)
    // (every reported entity is confirmed against its hull, so the leaves can be
    // culled by their own boxes in traversal order: BVH::findIntersectingLeafBoxes)
    ctx.singleton<broadphase::BVH>().findIntersectingLeafBoxes(aabb, [&](Entity e) {
        if (checkEntityAABBOverlap(ctx, aabb, e)) {
            fn(e);
        }
    });
}

#if defined(__HIPCC__)
// checkEntityAABBOverlap(ctx, boxes[b], e) for every b < num_boxes at once (bit b
// of the result): the transformed primitive boxes and the hull's extents along
// the world axes do not depend on the query box and are computed once.  Same
// expressions in the same order as the single-box function, so the bits are
// what num_boxes separate calls return.
template <int MAX_BOXES>
MADRONA_DEVICE inline uint32_t checkEntityAABBsOverlap(Context &ctx,
                                                       const math::AABB *boxes,
                                                       int32_t num_boxes,
                                                       uint32_t candidates,
                                                       Entity e)
{
    using namespace math;
    static_assert(MAX_BOXES <= 32);

    const ObjectManager &obj_mgr = *ctx.singleton<ObjectData>().mgr;

    base::ObjectID e_obj_id = ctx.get<base::ObjectID>(e);
    Vector3 e_pos = ctx.get<base::Position>(e);
    Quat e_rot = ctx.get<base::Rotation>(e);
    Diag3x3 e_scale = ctx.get<base::Scale>(e);

    uint32_t num_prims = obj_mgr.rigidBodyPrimitiveCounts[e_obj_id.idx];
    uint32_t base_prim_offset = obj_mgr.rigidBodyPrimitiveOffsets[e_obj_id.idx];

    uint32_t result = 0;
    for (uint32_t prim_offset = 0; prim_offset < num_prims; prim_offset++) {
        uint32_t prim_idx = base_prim_offset + prim_offset;

        const CollisionPrimitive &prim = obj_mgr.collisionPrimitives[prim_idx];
        if (prim.type != CollisionPrimitive::Type::Hull) {
            continue;
        }

        AABB prim_aabb = obj_mgr.primitiveAABBs[prim_idx];
        AABB txfmed_aabb = prim_aabb.applyTRS(e_pos, e_rot, e_scale);

        // boxes this primitive can still decide (a box an earlier primitive
        // accepted stays accepted: the single-box function returns there)
        uint32_t open = 0;
        for (int32_t b = 0; b < num_boxes; b++) {
            if (((candidates & ~result) >> b & 1u) != 0u &&
                    txfmed_aabb.overlaps(boxes[b])) {
                open |= 1u << b;
            }
        }
        if (open == 0u) {
            continue;
        }

        // exact extent of the transformed hull along the world axes
        const Vector3 *vertices = prim.hull.halfEdgeMesh.vertices;
        CountT num_verts = (CountT)prim.hull.halfEdgeMesh.numVertices;

        const Vector3 axes[3] { right, fwd, up };
        float min_hull_projs[3] { FLT_MAX, FLT_MAX, FLT_MAX };
        float max_hull_projs[3] { -FLT_MAX, -FLT_MAX, -FLT_MAX };

#pragma clang loop unroll(disable)
        for (CountT vert_idx = 0; vert_idx < num_verts; vert_idx++) {
            Vector3 v = e_rot.rotateVec(e_scale * vertices[vert_idx]) + e_pos;

            for (CountT i = 0; i < 3; i++) {
                float proj = dot(v, axes[i]);
                if (proj < min_hull_projs[i]) {
                    min_hull_projs[i] = proj;
                }
                if (proj > max_hull_projs[i]) {
                    max_hull_projs[i] = proj;
                }
            }
        }

        for (int32_t b = 0; b < num_boxes; b++) {
            if ((open >> b & 1u) == 0u) {
                continue;
            }
            bool axes_overlap = true;
            for (CountT i = 0; i < 3; i++) {
                bool proj_overlap = max_hull_projs[i] > boxes[b].pMin[i] &&
                    boxes[b].pMax[i] > min_hull_projs[i];
                if (!proj_overlap) {
                    axes_overlap = false;
                }
            }
            if (axes_overlap) {
                result |= 1u << b;
            }
        }
    }

    return result;
}

template <int MAX_BOXES, typename Fn>
MADRONA_DEVICE inline void findFirstEntitiesWithinAABBsWave(Context &ctx,
                                                            const math::AABB *boxes,
                                                            int32_t num_boxes,
                                                            Entity *out,
                                                            Fn &&accept)
{
    const broadphase::BVH &bvh = ctx.singleton<broadphase::BVH>();
    const uint32_t lane = threadIdx.x % 64u;
    for (int32_t b = 0; b < num_boxes; b++) {
        out[b] = Entity::none();
    }
    if (num_boxes <= 0) {
        return;
    }

    if (bvh.needsRebuild()) {
        // (no traversal order yet: one lane walks the tree, box after box)
        for (int32_t b = 0; b < num_boxes; b++) {
            Entity found = Entity::none();
            if (lane == 0) {
                findEntitiesWithinAABB(ctx, boxes[b], [&](Entity e) {
                    if (found == Entity::none() && accept(e)) {
                        found = e;
                    }
                });
            }
            out[b] = Entity { (uint32_t)__shfl((int32_t)found.gen, 0, 64),
                              __shfl(found.id, 0, 64) };
        }
        return;
    }

    // A query reports its leaves as a subsequence of the tree's full traversal
    // order (BVH::traversalOrder), and every leaf it can accept is one it
    // reaches: the body's own box lies inside its slot box and all ancestors
    // (refits only grow them).  Lane i takes the i-th leaf of that order and
    // answers for every box; per box the first accepting lane wins.
    const int32_t n = bvh.numLeaves();
    const int32_t *order = bvh.traversalOrder();
    uint32_t open = num_boxes >= 32 ? 0xFFFFFFFFu : (1u << num_boxes) - 1u;
    for (int32_t base = 0; base < n && open != 0u; base += 64) {
        const int32_t i = base + (int32_t)lane;
        Entity e = Entity::none();
        uint32_t hits = 0;
        if (i < n) {
            const int32_t leaf = order[i];
            // (the leaf's own box, not its slot in the parent node: the slot
            // only grows between rebuilds, while the exact test below can only
            // succeed for a box that meets the body's own box -- which contains
            // the body with room to spare and lies inside the slot; the leaves
            // skipped are ones the reference visits and rejects)
            const math::AABB own = bvh.getLeafAABB(broadphase::LeafID { leaf });
            uint32_t reached = 0;
            for (int32_t b = 0; b < num_boxes; b++) {
                if ((open >> b & 1u) != 0u && boxes[b].overlaps(own)) {
                    reached |= 1u << b;
                }
            }
            if (reached != 0u) {
                e = bvh.leafEntity(leaf);
                hits = checkEntityAABBsOverlap<MAX_BOXES>(ctx, boxes, num_boxes,
                                                          reached, e);
                if (hits != 0u && !accept(e)) {
                    hits = 0u;
                }
            }
        }
        for (int32_t b = 0; b < num_boxes; b++) {
            if ((open >> b & 1u) == 0u) {
                continue;
            }
            const unsigned long long mask = __ballot((hits >> b & 1u) != 0u);
            if (mask != 0ull) {
                const int first = __builtin_ctzll(mask);
                out[b] = Entity { (uint32_t)__shfl((int32_t)e.gen, first, 64),
                                  __shfl(e.id, first, 64) };
                open &= ~(1u << b);
            }
        }
    }
}

template <typename Fn>
MADRONA_DEVICE inline Entity findFirstEntityWithinAABBWave(Context &ctx,
                                                           math::AABB aabb,
                                                           Fn &&accept)
{
    Entity found;
    findFirstEntitiesWithinAABBsWave<1>(ctx, &aabb, 1, &found,
                                        std::forward<Fn>(accept));
    return found;
}
#endif

MADRONA_HD inline bool checkEntityAABBOverlap(Context &ctx,
                                              math::AABB aabb,
                                              Entity e)
{
    using namespace math;

    const ObjectManager &obj_mgr = *ctx.singleton<ObjectData>().mgr;

    base::ObjectID e_obj_id = ctx.get<base::ObjectID>(e);
    Vector3 e_pos = ctx.get<base::Position>(e);
    Quat e_rot = ctx.get<base::Rotation>(e);
    Diag3x3 e_scale = ctx.get<base::Scale>(e);

    uint32_t num_prims = obj_mgr.rigidBodyPrimitiveCounts[e_obj_id.idx];
    uint32_t base_prim_offset = obj_mgr.rigidBodyPrimitiveOffsets[e_obj_id.idx];

    for (uint32_t prim_offset = 0; prim_offset < num_prims; prim_offset++) {
        uint32_t prim_idx = base_prim_offset + prim_offset;

        const CollisionPrimitive &prim = obj_mgr.collisionPrimitives[prim_idx];
        if (prim.type != CollisionPrimitive::Type::Hull) {
            continue;
        }

        AABB prim_aabb = obj_mgr.primitiveAABBs[prim_idx];
        AABB txfmed_aabb = prim_aabb.applyTRS(e_pos, e_rot, e_scale);

        if (!txfmed_aabb.overlaps(aabb)) {
            continue;
        }

        // exact extent of the transformed hull along the world axes
        const Vector3 *vertices = prim.hull.halfEdgeMesh.vertices;
        CountT num_verts = (CountT)prim.hull.halfEdgeMesh.numVertices;

        const Vector3 axes[3] { right, fwd, up };
        float min_hull_projs[3] { FLT_MAX, FLT_MAX, FLT_MAX };
        float max_hull_projs[3] { -FLT_MAX, -FLT_MAX, -FLT_MAX };

        // (rolled: unrolled eight times it takes 116 registers, and its callers are
        // latency-bound systems that want wavefronts in flight, not registers)
#pragma clang loop unroll(disable)
        for (CountT vert_idx = 0; vert_idx < num_verts; vert_idx++) {
            Vector3 v = e_rot.rotateVec(e_scale * vertices[vert_idx]) + e_pos;

            for (CountT i = 0; i < 3; i++) {
                float proj = dot(v, axes[i]);
                if (proj < min_hull_projs[i]) {
                    min_hull_projs[i] = proj;
                }
                if (proj > max_hull_projs[i]) {
                    max_hull_projs[i] = proj;
                }
            }
        }

        bool axes_overlap = true;
        for (CountT i = 0; i < 3; i++) {
            bool proj_overlap = max_hull_projs[i] > aabb.pMin[i] &&
                aabb.pMax[i] > min_hull_projs[i];
            if (!proj_overlap) {
                axes_overlap = false;
            }
        }

        if (axes_overlap) {
            return true;
        }
    }

    return false;
}

MADRONA_HD inline Entity makeFixedJoint(Context &ctx,
                                        Entity e1, Entity e2,
                                        math::Quat attach_rot1,
                                        math::Quat attach_rot2,
                                        math::Vector3 r1, math::Vector3 r2,
                                        float separation)
{
    const PhysicsSystemState &physics_sys =
        ctx.singleton<PhysicsSystemState>();
    Entity e = ctx.makeEntity(physics_sys.jointArchetypeID);

    JointConstraint &joint = ctx.get<JointConstraint>(e);
    joint.e1 = e1;
    joint.e2 = e2;
    joint.type = JointConstraint::Type::Fixed;
    joint.fixed.attachRot1 = attach_rot1;
    joint.fixed.attachRot2 = attach_rot2;
    joint.fixed.separation = separation;
    joint.r1 = r1;
    joint.r2 = r2;

    return e;
}

MADRONA_HD inline Entity makeHingeJoint(Context &ctx,
                                        Entity e1, Entity e2,
                                        math::Vector3 a1_local,
                                        math::Vector3 a2_local,
                                        math::Vector3 b1_local,
                                        math::Vector3 b2_local,
                                        math::Vector3 r1, math::Vector3 r2)
{
    const PhysicsSystemState &physics_sys =
        ctx.singleton<PhysicsSystemState>();
    Entity e = ctx.makeEntity(physics_sys.jointArchetypeID);

    JointConstraint &joint = ctx.get<JointConstraint>(e);
    joint.e1 = e1;
    joint.e2 = e2;
    joint.type = JointConstraint::Type::Hinge;
    joint.hinge.a1Local = a1_local;
    joint.hinge.a2Local = a2_local;
    joint.hinge.b1Local = b1_local;
    joint.hinge.b2Local = b2_local;
    joint.r1 = r1;
    joint.r2 = r2;

    return e;
}

MADRONA_HOST_API inline void registerTypes(ECSRegistry &registry, Solver solver)
{
    // Same order as the reference (physics.cpp:308-341, xpbd.cpp:1046-1060):
    // singleton registration order fixes singleton entity ids.  (Not inside a
    // host-only block: the device pass instantiates the per-type id symbols
    // from these calls.)
    registry.registerComponent<ResponseType>();
    registry.registerComponent<broadphase::LeafID>();
    registry.registerComponent<Velocity>();
    registry.registerComponent<ExternalForce>();
    registry.registerComponent<ExternalTorque>();

    registry.registerSingleton<broadphase::BVH>();

    registry.registerComponent<CollisionEvent>();
    registry.registerArchetype<CollisionEventTemporary>();

    registry.registerComponent<CandidateCollision>();
    registry.registerArchetype<CandidateTemporary>(
        ComponentMetadataSelector<> {}, ArchetypeFlags::None,
        detail::capacityHint("MADRONA_MWHIP_MAX_CANDIDATES_PER_WORLD", 128));

    registry.registerComponent<JointConstraint>();
    registry.registerComponent<ContactConstraint>();

    registry.registerSingleton<PhysicsSystemState>();
    registry.registerSingleton<ObjectData>();

    if (solver == Solver::TGS) {
        // reference tgs.cpp:29-44
        registry.registerArchetype<tgs::Joint>();
        registry.registerArchetype<tgs::Contact>();
        registry.registerBundle<tgs::TGSRigidBodyState>();
        registry.registerBundleAlias<SolverBundleAlias, tgs::TGSRigidBodyState>();
        registry.registerSingleton<tgs::SolverState>();
    } else {
        registry.registerComponent<xpbd::SubstepPrevState>();
        registry.registerComponent<xpbd::PreSolvePositional>();
        registry.registerComponent<xpbd::PreSolveVelocity>();
        registry.registerComponent<xpbd::XPBDContactState>();

        registry.registerArchetype<xpbd::Joint>();
        registry.registerArchetype<xpbd::Contact>(
            ComponentMetadataSelector<> {}, ArchetypeFlags::None,
            detail::capacityHint("MADRONA_MWHIP_MAX_CONTACTS_PER_WORLD", 64));

        registry.registerSingleton<xpbd::SolverState>();

        registry.registerBundle<xpbd::XPBDRigidBodyState>();
        registry.registerBundleAlias<SolverBundleAlias, xpbd::XPBDRigidBodyState>();
    }

    registry.registerBundle<RigidBody>();
}

MADRONA_HOST_API inline TaskGraphNodeID setupBroadphaseTasks(
    TaskGraphBuilder &builder,
    Span<const TaskGraphNodeID> deps);

namespace detail {

#if MADRONA_ON_HOST
inline PhysicsScratch *scratchHost(TaskGraphBuilder &builder,
                                   PhysicsScratch *host_copy_out)
{
    // one scratch block per executor (kept in the executor's module slot 0),
    // shared by every graph that has physics
    mwhip_exec *exec = builder.exec();
    if (void *existing = mwhip_get_module_data(exec, 0)) {
        mwhip::check(mwhip_memcpy_d2h(host_copy_out, existing,
                                      sizeof(PhysicsScratch)), "memcpy_d2h");
        return (PhysicsScratch *)existing;
    }

    StateManager &state_mgr = builder.stateManager();

    PhysicsScratch ps {};

    // body archetypes in query order == CPU iteration order
    auto body_query = state_mgr.query<Entity, broadphase::LeafID>();
    const QueryRef *ref = body_query.getSharedRef();
    if (ref->numMatchingArchetypes > PhysicsScratch::maxBodyArchetypes) {
        FATAL("madrona_amd physics: more than %u rigid body archetypes",
              PhysicsScratch::maxBodyArchetypes);
    }

    uint32_t query_words[PhysicsScratch::maxBodyArchetypes * 3];
    mwhip::check(mwhip_get_query_data(exec, ref->offset,
        ref->numMatchingArchetypes * 3, query_words), "get_query_data");

    ps.numBodyArchetypes = ref->numMatchingArchetypes;
    for (uint32_t i = 0; i < ps.numBodyArchetypes; i++) {
        uint32_t archetype_id = query_words[i * 3];
        ps.bodyArchetypes[i] = archetype_id;
        uint32_t capacity = mwhip_archetype_capacity(exec, archetype_id);
        ps.bodyCounts[i] = (uint32_t *)mwhip_alloc_device(
            exec, (uint64_t)capacity * sizeof(uint32_t), 1);
    }

    ps.candidateArchetype = TypeTracker::typeID<CandidateTemporary>();
    ps.contactArchetype = TypeTracker::typeID<xpbd::Contact>();
    ps.jointArchetype = TypeTracker::typeID<xpbd::Joint>();

    const uint64_t num_worlds = mwhip_num_worlds(exec);
    ps.candidatesPerWorld = (uint32_t)phys::detail::capacityHint(
        "MADRONA_MWHIP_MAX_CANDIDATES_PER_WORLD", 256);
    ps.contactsPerWorld = (uint32_t)phys::detail::capacityHint(
        "MADRONA_MWHIP_MAX_CONTACTS_PER_WORLD", 128);
    ps.worldCandidates = (CandidateCollision *)mwhip_alloc_device(exec,
        num_worlds * ps.candidatesPerWorld * sizeof(CandidateCollision), 0);
    ps.worldContacts = (ContactConstraint *)mwhip_alloc_device(exec,
        num_worlds * ps.contactsPerWorld * sizeof(ContactConstraint), 0);
    ps.worldLambdas = (float *)mwhip_alloc_device(exec,
        num_worlds * ps.contactsPerWorld * sizeof(float), 0);
    ps.worldHullScratch = (char *)mwhip_alloc_device(exec,
        num_worlds * PhysicsScratch::hullScratchBytes, 0);

    PhysicsScratch *dev = (PhysicsScratch *)mwhip_alloc_device(
        exec, sizeof(PhysicsScratch), 0);
    if (dev == nullptr || ps.worldCandidates == nullptr ||
            ps.worldContacts == nullptr || ps.worldLambdas == nullptr ||
            ps.worldHullScratch == nullptr) {
        FATAL("madrona_amd physics: scratch allocation failed: %s",
              mwhip_last_error());
    }
    mwhip::check(mwhip_memcpy_h2d(dev, &ps, sizeof(PhysicsScratch)),
                 "memcpy_h2d");
    mwhip::check(mwhip_set_module_data(exec, 0, dev), "set_module_data");

    *host_copy_out = ps;
    return dev;
}

inline int32_t *numRowsAddr(mwhip_exec *exec, uint32_t archetype_id)
{
    char *hdr = (char *)mwhip_table_header(exec, archetype_id);
    return (int32_t *)(hdr + offsetof(mwhip::TableHdr, numRows));
}

inline uint32_t *needsSortAddr(mwhip_exec *exec, uint32_t archetype_id)
{
    char *hdr = (char *)mwhip_table_header(exec, archetype_id);
    return (uint32_t *)(hdr + offsetof(mwhip::TableHdr, needsSort));
}

inline TaskGraphNodeID addKernelNode(TaskGraphBuilder &builder,
                                     const char *name, const void *kernel,
                                     uint32_t count_mode, uint32_t fixed_count,
                                     Span<const TaskGraphNodeID> deps)
{
    mwhip_node_desc desc {};
    desc.kind = MWHIP_NODE_KERNEL;
    desc.name = name;
    desc.kernel = kernel;
    desc.count_mode = count_mode;
    desc.fixed_count = fixed_count;
    desc.threads_per_invocation = 1;
    return builder.addRuntimeNode(desc, -1, deps);
}
#endif

// broadphase overlap: count -> scan -> fill (replaces the reference's
// findIntersectingEntry ParallelFor, broadphase.cpp:930-993)
MADRONA_HOST_API inline TaskGraphNodeID setupCandidateTasks(
    TaskGraphBuilder &builder, Span<const TaskGraphNodeID> deps)
{
#if defined(__HIPCC__)
    [[maybe_unused]] auto count_stub = [] __host__ () -> const void * {
        return (const void *)&kernels::candidateKernel<false>;
    };
    [[maybe_unused]] auto fill_stub = [] __host__ () -> const void * {
        return (const void *)&kernels::candidateKernel<true>;
    };
#else
    auto count_stub = []() -> const void * { return nullptr; };
    auto fill_stub = []() -> const void * { return nullptr; };
#endif

#if MADRONA_ON_HOST
    mwhip_exec *exec = builder.exec();
    PhysicsScratch ps;
    scratchHost(builder, &ps);

    uint64_t body_capacity = 0;
    for (uint32_t i = 0; i < ps.numBodyArchetypes; i++) {
        body_capacity += mwhip_archetype_capacity(exec, ps.bodyArchetypes[i]);
    }

    auto count = addKernelNode(builder, "physics:candidateCount", count_stub(),
        MWHIP_COUNT_FIXED, (uint32_t)body_capacity, deps);

    mwhip_scan_params scan {};
    scan.num_segments = ps.numBodyArchetypes;
    scan.capacity = mwhip_archetype_capacity(exec, ps.candidateArchetype);
    for (uint32_t i = 0; i < ps.numBodyArchetypes; i++) {
        scan.data[i] = ps.bodyCounts[i];
        scan.lengths[i] = numRowsAddr(exec, ps.bodyArchetypes[i]);
    }
    scan.total_out = numRowsAddr(exec, ps.candidateArchetype);
    scan.needs_sort_out = needsSortAddr(exec, ps.candidateArchetype);

    auto scan_data = builder.constructNodeData<mwhip_scan_params>(scan);
    mwhip_node_desc scan_desc {};
    scan_desc.kind = MWHIP_NODE_EXCLUSIVE_SCAN;
    scan_desc.name = "physics:candidateScan";
    scan_desc.fixed_count = (uint32_t)body_capacity;
    auto scanned = builder.addRuntimeNode(scan_desc, scan_data.id, {count});

    return addKernelNode(builder, "physics:candidateFill", fill_stub(),
        MWHIP_COUNT_FIXED, (uint32_t)body_capacity, {scanned});
#else
    (void)builder; (void)deps;
    MADRONA_DEVICE_STUB();
#endif
}

// the leaf update + refit of every body (kernels::bvhRefreshKernel; the
// ParallelFor over the body rows it replaces is still there for measurements:
// MADRONA_MWHIP_BVH_REFRESH=0)
// with_rebuild: trees that asked for a rebuild are rebuilt in the same launch
// (*rebuilt_out = true) -- MADRONA_MWHIP_BVH_REFRESH: 0 = the ParallelFor, 1 =
// the refresh kernel with a rebuild launch of its own behind it, 2 (default) =
// the refresh kernel rebuilds
MADRONA_HOST_API inline TaskGraphNodeID setupLeafRefreshTasks(
    TaskGraphBuilder &builder, Span<const TaskGraphNodeID> deps,
    bool with_rebuild = false, bool *rebuilt_out = nullptr)
{
    using namespace base;
    using broadphase::LeafID;

#if defined(__HIPCC__)
    [[maybe_unused]] auto refresh_stub = [] __host__ (bool rebuild) -> const void * {
        return rebuild ? (const void *)&kernels::bvhRefreshKernel<true> :
                         (const void *)&kernels::bvhRefreshKernel<false>;
    };
#else
    auto refresh_stub = [](bool) -> const void * { return nullptr; };
#endif

    if (rebuilt_out != nullptr) {
        *rebuilt_out = false;
    }
#if MADRONA_ON_HOST
    const char *refresh_env = getenv("MADRONA_MWHIP_BVH_REFRESH");
    const int mode = refresh_env == nullptr ? 2 : atoi(refresh_env);
    if (mode != 0) {
        const bool rebuild = with_rebuild && mode >= 2;
        mwhip_node_desc desc {};
        desc.kind = MWHIP_NODE_KERNEL;
        desc.name = rebuild ? "physics:bvhRefresh+rebuild" : "physics:bvhRefresh";
        desc.kernel = refresh_stub(rebuild);
        desc.count_mode = MWHIP_COUNT_PER_WORLD;
        desc.threads_per_invocation = 64;
        if (rebuilt_out != nullptr) {
            *rebuilt_out = rebuild;
        }
        return builder.addRuntimeNode(desc, -1, deps);
    }
#else
    (void)with_rebuild;
#endif
    return builder.addToGraph<ParallelForNode<Context,
        broadphase::updateLeafAndRefitEntry,
            LeafID, Position, Rotation, Scale, ObjectID, Velocity>>(deps);
}

MADRONA_HOST_API inline TaskGraphNodeID setupPostIntegrationTasks(
    TaskGraphBuilder &builder, Span<const TaskGraphNodeID> deps)
{
    return setupLeafRefreshTasks(builder, deps);
}

}

MADRONA_HOST_API inline TaskGraphNodeID setupBroadphaseTasks(
    TaskGraphBuilder &builder,
    Span<const TaskGraphNodeID> deps)
{
    using namespace base;
    using broadphase::LeafID;

#if defined(__HIPCC__)
    [[maybe_unused]] auto bvh_stub = [] __host__ (bool check) -> const void * {
        return check ? (const void *)&kernels::bvhUpdateKernel<true> :
                       (const void *)&kernels::bvhUpdateKernel<false>;
    };
#else
    auto bvh_stub = [](bool) -> const void * { return nullptr; };
#endif

    bool rebuilt = false;
    auto update_leaves = detail::setupLeafRefreshTasks(builder, deps, true, &rebuilt);

    TaskGraphNodeID bvh_update = update_leaves;
#if MADRONA_ON_HOST
    if (!rebuilt) {
        mwhip_node_desc desc {};
        desc.kind = MWHIP_NODE_KERNEL;
        desc.name = "physics:bvhUpdate";
        // MADRONA_MWHIP_BVH_CHECK=1 (with MADRONA_MWHIP_BVH_REFRESH=1): every
        // rebuild is done both ways and compared (tests)
        const char *check_env = getenv("MADRONA_MWHIP_BVH_CHECK");
        desc.kernel = bvh_stub(check_env != nullptr && atoi(check_env) != 0);
        desc.count_mode = MWHIP_COUNT_PER_WORLD;
        desc.threads_per_invocation = 64;
        bvh_update = builder.addRuntimeNode(desc, -1, {update_leaves});
    }
#endif

    // (no refit after the rebuild: rebuilt boxes already contain their leaves)
    return bvh_update;
}

MADRONA_HOST_API inline TaskGraphNodeID setupPhysicsStepTasks(
    TaskGraphBuilder &builder,
    Span<const TaskGraphNodeID> deps,
    CountT num_substeps,
    Solver solver)
{
    if (solver == Solver::TGS) {
        // reference tgs.cpp:225-304 without the stages that have no effect
        // (phys_impl/tgs.hpp): per substep the two integrators, then the leaf
        // update every solver is followed by
        TaskGraphNodeID cur {};
        bool first = true;
        for (CountT i = 0; i < num_substeps; i++) {
            cur = first ?
                builder.addToGraph<ParallelForNode<Context,
                    tgs::integrateVelocities, base::Rotation, ResponseType,
                    ExternalForce, ExternalTorque, base::ObjectID, Velocity>>(
                        deps) :
                builder.addToGraph<ParallelForNode<Context,
                    tgs::integrateVelocities, base::Rotation, ResponseType,
                    ExternalForce, ExternalTorque, base::ObjectID, Velocity>>(
                        {cur});
            first = false;
            cur = builder.addToGraph<ParallelForNode<Context,
                tgs::integratePositions, base::Position, base::Rotation,
                Velocity>>({cur});
        }
        return detail::setupPostIntegrationTasks(builder, {cur});
    }

#if defined(__HIPCC__)
    [[maybe_unused]] auto step_stub = [] __host__ (int max_bodies, int lanes)
            -> const void * {
        if (max_bodies == 32 && lanes == 32) {
            // two worlds per wavefront
            return (const void *)&kernels::physicsStepLdsKernel<32, 32>;
        }
        switch (max_bodies) {
        case 32: return (const void *)&kernels::physicsStepLdsKernel<32>;
        case 64: return (const void *)&kernels::physicsStepLdsKernel<64>;
        case 128: return (const void *)&kernels::physicsStepLdsKernel<128>;
        default: return (const void *)&kernels::physicsStepKernel;
        }
    };
    [[maybe_unused]] auto fallback_stub = [] __host__ () -> const void * {
        return (const void *)&kernels::physicsStepKernel;
    };
    [[maybe_unused]] auto order_stub = [] __host__ () -> const void * {
        return (const void *)&kernels::physicsOrderKernel;
    };
#else
    auto step_stub = [](int, int) -> const void * { return nullptr; };
    auto fallback_stub = []() -> const void * { return nullptr; };
    auto order_stub = []() -> const void * { return nullptr; };
#endif

    // joints are created / destroyed by the simulator between steps: group
    // them by world (stable) so each world finds its range
    auto cur_node = builder.addToGraph<
        SortArchetypeNode<xpbd::Joint, WorldID>>(deps);
    cur_node = builder.addToGraph<ResetTmpAllocNode>({cur_node});

#if MADRONA_ON_HOST
    // candidates -> per substep (integrate, narrowphase, position solve,
    // velocities, velocity solve): one wavefront per world, one launch
    PhysicsScratch ps;
    detail::scratchHost(builder, &ps);


    // Worlds small enough to live in LDS take the LDS-resident kernel; the
    // bound is the largest world at graph-build time + 1/16, rounded up
    // (MADRONA_MWHIP_PHYS_MAX_BODIES overrides; > 128 selects the generic
    // kernel that works out of HBM).
    mwhip_exec *exec = builder.exec();
    int max_bodies = (int)phys::detail::capacityHint(
        "MADRONA_MWHIP_PHYS_MAX_BODIES", 0);
    // bodies per world that are not Static (what the two-worlds-per-wavefront
    // layout keeps solver records for; world_step.inl, maxSolverBodies)
    int bulk_movable = 0;
    {
        // rows per world of every rigid-body archetype, as initialised
        const uint32_t num_worlds = mwhip_num_worlds(exec);
        std::vector<int32_t> per_world(num_worlds, 0);
        std::vector<int32_t> movable(num_worlds, 0);
        std::vector<int32_t> counts(num_worlds);
        std::vector<int32_t> resp;
        for (uint32_t i = 0; i < ps.numBodyArchetypes; i++) {
            int32_t rows = mwhip_num_rows(exec, ps.bodyArchetypes[i]);
            resp.resize((size_t)(rows > 0 ? rows : 1));
            static_assert(sizeof(ResponseType) == sizeof(int32_t));
            int64_t n = mwhip_dump_column(exec, ps.bodyArchetypes[i],
                TypeTracker::typeID<ResponseType>(), resp.data(),
                resp.size() * sizeof(int32_t), counts.data());
            if (n < 0) {
                FATAL("physics: cannot read the body tables");
            }
            size_t row = 0;
            for (uint32_t w = 0; w < num_worlds; w++) {
                per_world[w] += counts[w];
                for (int32_t r = 0; r < counts[w]; r++, row++) {
                    movable[w] += resp[row] != (int32_t)ResponseType::Static ? 1 : 0;
                }
            }
        }
        std::sort(movable.begin(), movable.end());
        if (!movable.empty()) {
            const size_t covered =
                (movable.size() - 1) - (movable.size() - 1) / 50;
            bulk_movable = movable[covered] + movable[covered] / 16;
        }
      if (max_bodies == 0) {
        // The instantiation is sized for the bulk of the worlds, not for the
        // largest one: what 98 % of them hold (+ 1/16); a world beyond it --
        // at build time or later -- is stepped by the HBM kernel in fallback
        // mode, in the same step (world_step.inl, FramedWorld::tooManyBodies).
        std::sort(per_world.begin(), per_world.end());
        const size_t covered = per_world.empty() ? 0 :
            (per_world.size() - 1) - (per_world.size() - 1) / 50;
        const int32_t bulk = per_world.empty() ? 1 :
            (per_world[covered] > 1 ? per_world[covered] : 1);
        max_bodies = bulk + bulk / 16;
      }
    }
    max_bodies = max_bodies <= 32 ? 32 : max_bodies <= 64 ? 64 :
                 max_bodies <= 128 ? 128 : 0;

    // Worlds of at most 32 bodies go two to a wavefront, one per half
    // (phys_impl/world_step.inl; measured on 8192 Escape-Room worlds: 830 ->
    // 712 us per step, profiles/r03_phys_variants.jsonl).
    // MADRONA_MWHIP_PHYS_LANES=64 keeps one world per wavefront.
    // (that layout keeps solver records for 20 of a world's 32 bodies: worlds
    // whose bulk has more bodies that move stay one to a wavefront)
    const int lanes_per_world = max_bodies == 32 && bulk_movable <= 20 &&
        phys::detail::capacityHint("MADRONA_MWHIP_PHYS_LANES", 32) == 32 ? 32 : 64;

    // The LDS kernels take their worlds heaviest first (physicsOrderKernel:
    // without it the launch ends on a few contact-rich worlds with most SIMDs
    // idle), through a frame of addresses that kernel resolves once per launch
    // (PhysicsFrame), and list the worlds they cannot hold for the HBM kernel
    // that follows them (fallbackList).  What rounds 3-4 also built around this
    // node and measured slower or equal -- world images packed by a kernel of
    // their own, the k-th heaviest world paired with the k-th lightest, the
    // leaf refit folded into the step's epilogue, persistent wavefronts with
    // look-ahead, the order blended over steps, every wavefront walking the
    // tables itself -- is out of the code since round 5; the measurements are
    // in profiles/r04_phys_variants.jsonl and DESIGN.md section 14.
    uint32_t *world_cost = nullptr;
    int32_t *world_order = nullptr;
    int32_t *fallback_list = nullptr;
    if (max_bodies != 0) {
        const uint64_t worlds = mwhip_num_worlds(exec);
        world_cost = (uint32_t *)mwhip_alloc_device(exec, worlds * 4u, 1);
        world_order = (int32_t *)mwhip_alloc_device(exec, worlds * 4u, 1);
        fallback_list = (int32_t *)mwhip_alloc_device(exec, (worlds + 1u) * 4u, 1);
        if (world_cost == nullptr || world_order == nullptr ||
                fallback_list == nullptr) {
            FATAL("madrona_amd physics: world order allocation failed: %s",
                  mwhip_last_error());
        }
    }
    auto params = builder.constructNodeData<PhysicsStepNode>(
        PhysicsStepNode {
            PhysicsStepParams { (int32_t)num_substeps,
                                (uint32_t)phys::detail::capacityHint(
                                    "MADRONA_MWHIP_PHYS_LDS_CONTACTS", 0),
                                world_cost, world_order, fallback_list },
            PhysicsFrame {} });

    if (world_order != nullptr) {
        mwhip_node_desc order {};
        order.kind = MWHIP_NODE_KERNEL;
        order.name = "physics:orderWorlds";
        order.kernel = order_stub();
        order.count_mode = MWHIP_COUNT_FIXED;
        order.fixed_count = 1;
        order.threads_per_invocation = 1024;
        cur_node = builder.addRuntimeNode(order, params.id, {cur_node});
    }

    mwhip_node_desc desc {};
    desc.kind = MWHIP_NODE_KERNEL;
    desc.name = max_bodies != 0 ? "physics:worldStep(LDS)" :
                                  "physics:worldStep";
    desc.kernel = step_stub(max_bodies, lanes_per_world);
    desc.count_mode = MWHIP_COUNT_PER_WORLD;
    desc.threads_per_invocation = 64;
    if (lanes_per_world == 32) {
        // one wavefront per PAIR of worlds
        desc.count_mode = MWHIP_COUNT_FIXED;
        desc.fixed_count = (mwhip_num_worlds(exec) + 1u) / 2u;
    }
    desc.arg0 = 0u;
    cur_node = builder.addRuntimeNode(desc, params.id, {cur_node});

    if (max_bodies != 0) {
        // the worlds the LDS kernel could not hold, out of HBM (usually none:
        // the launch then costs its floor, ~4 us; the reference has no cap on
        // a world's bodies or contacts, broadphase.cpp:892-1052)
        mwhip_node_desc fb {};
        fb.kind = MWHIP_NODE_KERNEL;
        fb.name = "physics:worldStep(fallback)";
        fb.kernel = fallback_stub();
        fb.count_mode = MWHIP_COUNT_FIXED;
        fb.fixed_count = 64u * 256u;    // 64 workgroups of four wavefronts
        fb.threads_per_invocation = 1;
        fb.arg0 = 1u;
        cur_node = builder.addRuntimeNode(fb, params.id, {cur_node});
    }
#else
    (void)num_substeps;
#endif

    return detail::setupPostIntegrationTasks(builder, {cur_node});
}

MADRONA_HOST_API inline TaskGraphNodeID setupCleanupTasks(
    TaskGraphBuilder &builder, Span<const TaskGraphNodeID> deps)
{
    return builder.addToGraph<ClearTmpNode<CollisionEventTemporary>>(deps);
}

MADRONA_HOST_API inline TaskGraphNodeID setupStandaloneBroadphaseOverlapTasks(
    TaskGraphBuilder &builder,
    Span<const TaskGraphNodeID> deps)
{
    return detail::setupCandidateTasks(builder, deps);
}

MADRONA_HOST_API inline TaskGraphNodeID setupStandaloneBroadphaseCleanupTasks(
    TaskGraphBuilder &builder,
    Span<const TaskGraphNodeID> deps)
{
    return builder.addToGraph<ClearTmpNode<CandidateTemporary>>(deps);
}

}

// Per-world broadphase: a 4-wide BVH over the world's rigid bodies.
//
// API contract: reference include/madrona/broadphase.hpp:14-115 +
// broadphase.inl (LeafID, BVH: reserveLeaf, getLeafAABB, findIntersecting,
// findLeafIntersecting, updateLeafPosition, expandLeaf, refitLeaf,
// rebuildOnUpdate, updateTree, clearLeaves).  Behaviour follows
// src/physics/broadphase.cpp: a top-down midpoint-split build that only runs
// after PhysicsSystem::reset (:47-285), leaf AABBs inflated by motion
// (:440-497) and a leaf-to-root refit that only ever grows boxes (:550-647).
//
// One BVH per world, stored as a singleton component; its arrays come from the
// executor's persistent region (rawAlloc).  Refit uses integer atomics on the
// float bit patterns (order independent => deterministic).
#pragma once

#include <madrona/components.hpp>
#include <madrona/context.hpp>
#include <madrona/math.hpp>
#include <madrona/memory.hpp>

namespace madrona::phys {

struct ObjectManager;

}

namespace madrona::phys::broadphase {

struct LeafID {
    int32_t id;
};

// LIFO of node indices for the traversals: the top 8 entries live in two
// 64-bit registers (16 bits each), older ones spill to a private array.  A
// plain int32_t stack[32] is indexed dynamically, which puts every push and pop
// in scratch memory on CDNA (hundreds of cycles each).
class NodeStack {
public:
    MADRONA_HD inline NodeStack() : lo_(0), hi_(0), num_reg_(0), num_spilled_(0) {}

    MADRONA_HD inline bool empty() const
    {
        return num_reg_ == 0 && num_spilled_ == 0;
    }

    MADRONA_HD inline void push(int32_t node)
    {
        if (num_reg_ == 8) {
            spilled_[num_spilled_++] = (uint16_t)(hi_ >> 48);
            num_reg_ = 7;
        }
        hi_ = (hi_ << 16) | (lo_ >> 48);
        lo_ = (lo_ << 16) | (uint64_t)(uint16_t)node;
        num_reg_++;
    }

    MADRONA_HD inline int32_t pop()
    {
        if (num_reg_ == 0) {
            return (int32_t)spilled_[--num_spilled_];
        }
        int32_t node = (int32_t)(lo_ & 0xFFFFu);
        lo_ = (lo_ >> 16) | (hi_ << 48);
        hi_ >>= 16;
        num_reg_--;
        return node;
    }

private:
    uint64_t lo_;
    uint64_t hi_;
    int32_t num_reg_;
    int32_t num_spilled_;
    uint16_t spilled_[32];
};

class BVH {
public:
    MADRONA_HD inline BVH(const ObjectManager *obj_mgr,
                          CountT max_leaves,
                          float leaf_velocity_expansion,
                          float leaf_accel_expansion);

    MADRONA_HD inline LeafID reserveLeaf(Entity e, base::ObjectID obj_id);

    // (this backend: reset systems that run 64 lanes per world) the leaf a
    // sequence of reserveLeaf calls would have given its idx-th caller, taken
    // by the lanes in parallel; one lane then sets the count
    MADRONA_HD inline LeafID reserveLeafAt(int32_t idx, Entity e,
                                           base::ObjectID obj_id)
    {
        leaf_entities_[idx] = e;
        leaf_obj_ids_[idx] = obj_id;
        return LeafID { idx };
    }
    MADRONA_HD inline void setNumLeaves(int32_t n) { num_leaves_ = n; }
    MADRONA_HD inline math::AABB getLeafAABB(LeafID leaf_id) const
    {
        return leaf_aabbs_[leaf_id.id];
    }

    template <typename Fn>
    MADRONA_HD inline void findIntersecting(const math::AABB &aabb,
                                            Fn &&fn) const;
    // (see broadphase.inl: for callers that confirm entities against their
    // geometry afterwards)
    template <typename Fn>
    MADRONA_HD inline void findIntersectingLeafBoxes(const math::AABB &aabb,
                                                     Fn &&fn) const;

    template <typename Fn>
    MADRONA_HD inline void findLeafIntersecting(LeafID leaf_id, Fn &&fn) const
    {
        findIntersecting(leaf_aabbs_[leaf_id.id], std::forward<Fn>(fn));
    }

    // Closest hit of the ray o + t * d, 0 <= t <= t_max, against the collision
    // primitives of every leaf (hulls and planes).  Returns Entity::none() on a
    // miss.  Reference src/physics/broadphase.cpp:658-871.
    // PRECONDITION (this backend): the leaf boxes are current, i.e.
    // PhysicsSystem::setupBroadphaseTasks ran after the last write to a body's
    // Position / Rotation / Scale.  The traversal culls by a leaf's OWN box
    // (leaf_aabbs_), which only the leaf update refreshes; the reference culls
    // by the boxes of the tree nodes, which are never smaller than a stale leaf
    // box and may therefore still reach a body that was teleported (pose written,
    // no leaf update) -- here such a body can be missed until the next leaf
    // update.  The same holds for PhysicsSystem::findFirstEntitiesWithinAABBsWave.
    // (Every simulator in sims/ runs the broadphase tasks before its ray systems.)
    MADRONA_HD inline Entity traceRay(math::Vector3 o,
                                      math::Vector3 d,
                                      float *out_hit_t,
                                      math::Vector3 *out_hit_normal,
                                      float t_max = float(INFINITY));

    // (this backend) traceRay for rays that share their origin: called by the
    // lanes of a 32-lane group together (an agent's lidar in a
    // CustomParallelForNode<..., 32, 1, ...>: lane = ray) with the group's LDS
    // scratch -- broadphase::rayGroupScratch() -- the ray-independent half of
    // every leaf test is computed once per leaf instead of once per (ray,
    // leaf).  Same hits, bit for bit.  scratch == nullptr, rays that turn out
    // not to share origin and tree, or fewer than 8 of them: plain traceRay.
    // Only kernels that call rayGroupScratch() carry its LDS (39 KB per
    // 256-thread workgroup); plain traceRay carries none.
    struct RayGroupScratch;
    MADRONA_HD inline Entity traceRayShared(RayGroupScratch *scratch,
                                            math::Vector3 o,
                                            math::Vector3 d,
                                            float *out_hit_t,
                                            math::Vector3 *out_hit_normal,
                                            float t_max = float(INFINITY));

    MADRONA_HD inline void updateLeafPosition(LeafID leaf_id,
                                              const math::Vector3 &pos,
                                              const math::Quat &rot,
                                              const math::Diag3x3 &scale,
                                              const math::Vector3 &linear_vel,
                                              const math::AABB &obj_aabb);

    MADRONA_HD inline math::AABB expandLeaf(LeafID leaf_id,
                                            const math::Vector3 &linear_vel);

    MADRONA_HD inline void updateLeafAndRefit(LeafID leaf_id,
                                              const math::Vector3 &pos,
                                              const math::Quat &rot,
                                              const math::Diag3x3 &scale,
                                              const math::Vector3 &linear_vel,
                                              const math::AABB &obj_aabb);
    MADRONA_HD inline void refitLeaf(LeafID leaf_id,
                                     const math::AABB &leaf_aabb);

    MADRONA_HD inline void rebuildOnUpdate() { force_rebuild_ = true; }
    MADRONA_HD inline void updateTree();

    MADRONA_HD inline void clearLeaves() { num_leaves_ = 0; }

    MADRONA_HD inline int32_t numLeaves() const { return num_leaves_; }

    // Leaves in the order a full traversal (findIntersecting with an all-
    // covering box) reports them.  Pruning only removes visits, so EVERY query
    // reports its leaves as a subsequence of this order; together with the leaf
    // boxes it lets a kernel enumerate a query's hits without walking the tree
    // (phys_impl/world_step.inl).  Valid from the last rebuild on.
    MADRONA_HD inline const int32_t *traversalOrder() const
    {
        return dfs_leaves_;
    }

    // the box the traversal tests for `leaf`: its slot in the parent node (grown
    // by every refit since the last rebuild, so a superset of getLeafAABB())
    MADRONA_HD inline math::AABB leafSlotBounds(int32_t leaf) const
    {
        uint32_t parent = leaf_parents_[leaf];
        return nodes_[parent >> 2].bounds((CountT)(parent & 3u));
    }

    MADRONA_HD inline Entity leafEntity(int32_t leaf) const
    {
        return leaf_entities_[leaf];
    }

    // What the fused physics step reads of the tree, as addresses: a kernel that
    // has them can fetch the boxes of all its leaves in one round of loads
    // instead of going through the object once per box (phys_impl/
    // world_step.inl, loadWorldFramed).
    struct StepView {
        const math::AABB *leafAABBs;
        const uint32_t *leafParents;
        const void *nodes;
        const int32_t *traversalOrder;
        int32_t numLeaves;
    };
    // (loads through the global address space: mwhip::loadGlobal)
    MADRONA_HD static inline StepView loadStepView(const BVH *tree)
    {
        return StepView {
            mwhip::loadGlobal(&tree->leaf_aabbs_),
            mwhip::loadGlobal(&tree->leaf_parents_),
            mwhip::loadGlobal(&tree->nodes_),
            mwhip::loadGlobal(&tree->dfs_leaves_),
            mwhip::loadGlobal(&tree->num_leaves_),
        };
    }
    // What updateLeafAndRefit reads of the object before it touches a leaf; a
    // caller that updates many leaves of one tree loads it once, and can put
    // the leaf's own loads (parent, slot) next to loads of its own
    // (physicsStepLdsKernel's epilogue).
    struct RefitView {
        uint32_t *leafParents;
        void *nodes;
        math::AABB *leafAABBs;
        void *leafTransforms;
        int32_t *sortedLeaves;
        float velocityExpansion;
        float accelExpansion;
        bool refit;         // false: a rebuild is pending, boxes are not grown
    };
    MADRONA_HD static inline RefitView loadRefitView(const BVH *tree)
    {
        return RefitView {
            mwhip::loadGlobal(&tree->leaf_parents_),
            mwhip::loadGlobal(&tree->nodes_),
            mwhip::loadGlobal(&tree->leaf_aabbs_),
            mwhip::loadGlobal(&tree->leaf_transforms_),
            mwhip::loadGlobal(&tree->sorted_leaves_),
            mwhip::loadGlobal(&tree->leaf_velocity_expansion_),
            mwhip::loadGlobal(&tree->leaf_accel_expansion_),
            !mwhip::loadGlobalBool(&tree->force_rebuild_),
        };
    }
    // ... plus what a wavefront that refreshes ALL leaves of the tree starts
    // from (physics.inl bvhRefreshKernel): who the leaves are
    struct RefreshView {
        RefitView refit;
        const Entity *leafEntities;
        const base::ObjectID *leafObjIDs;
        const ObjectManager *objMgr;
        int32_t numLeaves;
    };
    MADRONA_HD static inline RefreshView loadRefreshView(const BVH *tree)
    {
        return RefreshView {
            loadRefitView(tree),
            mwhip::loadGlobal(&tree->leaf_entities_),
            mwhip::loadGlobal(&tree->leaf_obj_ids_),
            mwhip::loadGlobal(&tree->obj_mgr_),
            mwhip::loadGlobal(&tree->num_leaves_),
        };
    }
    // leaf_parent = view.leafParents[leaf], slot = loadSlotBounds(view.nodes,
    // leaf_parent) (anything if !view.refit)
    MADRONA_HD static inline void applyLeafUpdate(
        const RefitView &view, int32_t leaf, uint32_t leaf_parent,
        math::AABB slot, const math::Vector3 &pos, const math::Quat &rot,
        const math::Diag3x3 &scale, const math::Vector3 &linear_vel,
        const math::AABB &obj_aabb);

    // == leafSlotBounds(leaf) with parent = view.leafParents[leaf]
    MADRONA_HD static inline math::AABB loadSlotBounds(const void *nodes,
                                                       uint32_t parent)
    {
        const Node *node = (const Node *)nodes + (parent >> 2);
        const uint32_t c = parent & 3u;
        return math::AABB {
            { mwhip::loadGlobal(&node->minX[c]), mwhip::loadGlobal(&node->minY[c]),
              mwhip::loadGlobal(&node->minZ[c]) },
            { mwhip::loadGlobal(&node->maxX[c]), mwhip::loadGlobal(&node->maxY[c]),
              mwhip::loadGlobal(&node->maxZ[c]) } };
    }

    // ---- staged rebuild (physics.inl bvhUpdateKernel) -------------------------
    // The top-down build is a long chain of dependent accesses to a few KB:
    // run from HBM it is pure latency.  A kernel copies the leaf boxes next to
    // the CU, runs rebuildStaged() on a *copy* of this object whose arrays point
    // at that storage, and copies nodes / leaf parents / orders back.
    static constexpr uint32_t nodeBytes = 116;

    MADRONA_HD inline bool needsRebuild() const { return force_rebuild_; }
    MADRONA_HD inline int32_t nodeCapacity() const
    {
        return (int32_t)num_allocated_nodes_;
    }
    MADRONA_HD inline void *rawNodes() const { return nodes_; }
    MADRONA_HD inline math::AABB *rawLeafAABBs() const { return leaf_aabbs_; }
    MADRONA_HD inline uint32_t *rawLeafParents() const { return leaf_parents_; }
    MADRONA_HD inline int32_t *rawSortedLeaves() const { return sorted_leaves_; }
    MADRONA_HD inline int32_t *rawTraversalOrder() const { return dfs_leaves_; }

    MADRONA_HD inline int32_t numNodes() const { return (int32_t)num_nodes_; }

    // a copy of this object whose tree arrays live somewhere else (LDS)
    // leaf_centers (optional): centroid of every leaf box, precomputed
    MADRONA_HD inline BVH rebased(void *nodes, math::AABB *leaf_aabbs,
                                  uint32_t *leaf_parents,
                                  int32_t *sorted_leaves,
                                  int32_t *traversal_order,
                                  math::Vector3 *leaf_centers = nullptr) const
    {
        BVH copy = *this;
        copy.leaf_centers_ = leaf_centers;
        copy.nodes_ = (Node *)nodes;
        copy.leaf_aabbs_ = leaf_aabbs;
        copy.leaf_parents_ = leaf_parents;
        copy.sorted_leaves_ = sorted_leaves;
        copy.dfs_leaves_ = traversal_order;
        return copy;
    }

    // call on a rebased copy; returns the number of nodes the tree may reference
    struct RebuildStackEntry {
        int32_t nodeID;
        int32_t parentID;
        int32_t offset;
        int32_t numObjs;
    };
    static constexpr int32_t rebuildStackSize = 64;

    MADRONA_HD inline int32_t rebuildStaged(RebuildStackEntry *stack)
    {
        rebuild(stack);
        return (int32_t)num_nodes_;
    }

#if defined(__HIPCC__)
    // Wave-cooperative flavour of the staged rebuild: all 64 lanes call it on
    // identical rebased copies (<= 64 leaves); builds the same tree, node for
    // node, as rebuild().  Needs leaf_centers_.
    __device__ inline int32_t rebuildStagedWave(uint32_t lane,
                                                RebuildStackEntry *stack);

    // The same tree again, built breadth first: every range of a level is
    // split at once (the lanes hold the leaves in sorted position; a range is a
    // segment of lanes), node ids, merged bounds and the traversal order are
    // derived afterwards from the range records.  A level costs two segmented
    // partitions whatever the number of ranges in it; the stack machine above
    // walks the ranges one by one (15 partitions, 16 leaf nodes and 20 merges in
    // a row for a 28-leaf tree).  Returns -1 if the tree needs more range
    // records than `scratch` holds (the caller then takes rebuildStagedWave).
    struct RebuildRange {
        int16_t lo;             // first sorted position
        int16_t n;              // leaves (0 .. 64)
        int16_t parent;         // range index, -1: root
        int16_t firstChild;     // range index of child 0, -1: a leaf node
        int16_t node;           // node id (pre-order, as rebuild() numbers them)
        int16_t subtreeNodes;
        int16_t leafStart;      // first traversal rank of the subtree's leaves
        int16_t slot;           // child index in the parent
    };
    static constexpr int32_t maxRebuildRanges = 96;
    struct SegmentedScratch {
        RebuildRange ranges[maxRebuildRanges];
        int32_t fromLeft[64];
        int32_t fromRight[64];
    };
    __device__ inline int32_t rebuildStagedSegmented(uint32_t lane,
                                                     SegmentedScratch *scratch);
#endif

    MADRONA_HD inline void finishRebuild(int32_t num_nodes)
    {
        num_nodes_ = num_nodes;
        num_tree_leaves_ = num_leaves_;
        force_rebuild_ = false;
    }

private:
    static constexpr int32_t sentinel_ = -1;
    static constexpr uint32_t leaf_bit_ = 0x80000000u;

    // SoA over the 4 children so one node's child boxes load as 6 x 16 B
    struct alignas(4) Node {
        float minX[4];
        float minY[4];
        float minZ[4];
        float maxX[4];
        float maxY[4];
        float maxZ[4];
        int32_t children[4];
        int32_t parentID;

        MADRONA_HD inline bool isLeaf(CountT c) const
        {
            return ((uint32_t)children[c] & leaf_bit_) != 0;
        }

        MADRONA_HD inline int32_t leafIDX(CountT c) const
        {
            return (int32_t)((uint32_t)children[c] & ~leaf_bit_);
        }

        MADRONA_HD inline void setLeaf(CountT c, int32_t idx)
        {
            children[c] = (int32_t)(leaf_bit_ | (uint32_t)idx);
        }

        MADRONA_HD inline bool hasChild(CountT c) const
        {
            return children[c] != sentinel_;
        }

        MADRONA_HD inline void setBounds(CountT c, const math::AABB &aabb)
        {
            minX[c] = aabb.pMin.x; minY[c] = aabb.pMin.y; minZ[c] = aabb.pMin.z;
            maxX[c] = aabb.pMax.x; maxY[c] = aabb.pMax.y; maxZ[c] = aabb.pMax.z;
        }

        MADRONA_HD inline math::AABB bounds(CountT c) const
        {
            return math::AABB { { minX[c], minY[c], minZ[c] },
                                { maxX[c], maxY[c], maxZ[c] } };
        }
    };

    static_assert(sizeof(Node) == nodeBytes);

    struct LeafTransform {
        math::Vector3 pos;
        math::Quat rot;
        math::Diag3x3 scale;
    };

    MADRONA_HD static inline CountT numInternalNodes(CountT num_leaves)
    {
        CountT third = utils::divideRoundUp(num_leaves - 1, CountT(3));
        return (third > 1 ? third : 1) + num_leaves;
    }

    MADRONA_HD inline math::Vector3 leafCenter(int32_t sorted_idx) const
    {
        if (leaf_centers_ != nullptr) {
            return leaf_centers_[sorted_leaves_[sorted_idx]];
        }
        math::AABB aabb = leaf_aabbs_[sorted_leaves_[sorted_idx]];
        return (aabb.pMin + aabb.pMax) / 2.f;
    }

    MADRONA_HD inline bool traceRayIntoLeaf(int32_t leaf_idx,
                                            math::Vector3 world_ray_o,
                                            math::Vector3 world_ray_d,
                                            float t_min,
                                            float t_max,
                                            float *hit_t,
                                            math::Vector3 *hit_normal);
    // the ray-independent part of a leaf test (physics.inl)
    struct RayLeaf;
    MADRONA_HD inline RayLeaf rayLeaf(int32_t leaf_idx,
                                      math::Vector3 world_ray_o) const;
    MADRONA_HD inline bool traceRayIntoLeaf(const RayLeaf &leaf,
                                            math::Vector3 world_ray_d,
                                            float t_min,
                                            float t_max,
                                            float *hit_t,
                                            math::Vector3 *hit_normal);

    template <bool SharedOrigin>
    MADRONA_HD inline Entity traceRayImpl(RayGroupScratch *scratch,
                                          math::Vector3 o,
                                          math::Vector3 d,
                                          float *out_hit_t,
                                          math::Vector3 *out_hit_normal,
                                          float t_max);

    MADRONA_HD inline int32_t midpointSplit(int32_t base, int32_t num_elems);
    MADRONA_HD inline void growAncestors(int32_t child_idx,
                                         const math::AABB &leaf_aabb);
    MADRONA_HD static inline void growAncestors(Node *nodes, int32_t child_idx,
                                                const math::AABB &leaf_aabb);
    MADRONA_HD inline void rebuild();
    MADRONA_HD inline void rebuild(RebuildStackEntry *stack);
#if defined(__HIPCC__)
    __device__ inline int32_t midpointSplitWave(uint32_t lane, int32_t base,
                                                int32_t num_elems);
#endif

    Node *nodes_;
    CountT num_nodes_;
    CountT num_allocated_nodes_;
    Entity *leaf_entities_;
    const ObjectManager *obj_mgr_;
    base::ObjectID *leaf_obj_ids_;
    math::AABB *leaf_aabbs_;
    LeafTransform *leaf_transforms_;
    uint32_t *leaf_parents_;
    int32_t *sorted_leaves_;
    int32_t *dfs_leaves_;
    math::Vector3 *leaf_centers_;   // only set on rebased copies
    int32_t num_leaves_;
    int32_t num_tree_leaves_;       // leaves the current tree was built over
    [[maybe_unused]] int32_t num_allocated_leaves_;     // device bounds check
    float leaf_velocity_expansion_;
    float leaf_accel_expansion_;
    bool force_rebuild_;
};

}

#include "broadphase.inl"

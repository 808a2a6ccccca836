#pragma once

namespace madrona::phys::broadphase {

BVH::BVH(const ObjectManager *obj_mgr,
         CountT max_leaves,
         float leaf_velocity_expansion,
         float leaf_accel_expansion)
    // (node indices travel through 16-bit stack slots: max_leaves < ~49 K)
    : nodes_((Node *)rawAlloc(sizeof(Node) * numInternalNodes(max_leaves))),
      num_nodes_(0),
      num_allocated_nodes_(numInternalNodes(max_leaves)),
      leaf_entities_((Entity *)rawAlloc(sizeof(Entity) * max_leaves)),
      obj_mgr_(obj_mgr),
      leaf_obj_ids_((base::ObjectID *)rawAlloc(
          sizeof(base::ObjectID) * max_leaves)),
      leaf_aabbs_((math::AABB *)rawAlloc(sizeof(math::AABB) * max_leaves)),
      leaf_transforms_((LeafTransform *)rawAlloc(
          sizeof(LeafTransform) * max_leaves)),
      leaf_parents_((uint32_t *)rawAlloc(sizeof(uint32_t) * max_leaves)),
      sorted_leaves_((int32_t *)rawAlloc(sizeof(int32_t) * max_leaves)),
      dfs_leaves_((int32_t *)rawAlloc(sizeof(int32_t) * max_leaves)),
      leaf_centers_(nullptr),
      num_leaves_(0),
      num_tree_leaves_(0),
      num_allocated_leaves_((int32_t)max_leaves),
      leaf_velocity_expansion_(leaf_velocity_expansion),
      leaf_accel_expansion_(leaf_accel_expansion),
      force_rebuild_(true)
{}

LeafID BVH::reserveLeaf(Entity e, base::ObjectID obj_id)
{
#if defined(__HIP_DEVICE_COMPILE__)
    int32_t leaf_idx = __hip_atomic_fetch_add(&num_leaves_, 1, __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT);
    if (leaf_idx >= num_allocated_leaves_) {
        mwhip::raiseError(mwGPU::getStateManager(), mwhip::kErrPhysics);
        leaf_idx = num_allocated_leaves_ - 1;
    }
#else
    int32_t leaf_idx = num_leaves_++;
#endif

    leaf_entities_[leaf_idx] = e;
    leaf_obj_ids_[leaf_idx] = obj_id;

    return LeafID { leaf_idx };
}

template <typename Fn>
void BVH::findIntersecting(const math::AABB &aabb, Fn &&fn) const
{
    // depth-first, children visited 0..3, deeper nodes pushed and popped LIFO:
    // the visit order defines the order candidates are emitted in
    NodeStack stack;
    stack.push(0);

    while (!stack.empty()) {
        const Node &node = nodes_[stack.pop()];
        for (CountT c = 0; c < 4; c++) {
            if (!node.hasChild(c)) {
                continue;
            }

            if (!aabb.overlaps(node.bounds(c))) {
                continue;
            }

            if (node.isLeaf(c)) {
                fn(leaf_entities_[node.leafIDX(c)]);
            } else {
                stack.push(node.children[c]);
            }
        }
    }
}

// The leaves whose OWN box (leaf_aabbs_) overlaps the query, in the order
// findIntersecting reports leaves.  For callers that confirm every reported
// entity against its actual geometry afterwards (findEntitiesWithinAABB): a
// body whose hull overlaps the query also overlaps it with its leaf box -- the
// hull plus a margin -- and a leaf is reached by the tree walk iff the query
// overlaps its slot box, a superset of the leaf box; so the confirmed entities,
// and their order, are those of the walk, without the dependent node -> child
// -> node chain (one lane per world walks it alone in a portable simulator's
// grab system: 169 us at 8192 worlds).  Same precondition as traceRay: the
// leaf boxes are current.  A pending rebuild takes the walk.
template <typename Fn>
void BVH::findIntersectingLeafBoxes(const math::AABB &aabb, Fn &&fn) const
{
    if (force_rebuild_) {
        findIntersecting(aabb, std::forward<Fn>(fn));
        return;
    }
    const int32_t n = num_tree_leaves_;
    constexpr int32_t group = 8;
    for (int32_t g = 0; g < n; g += group) {
        // (leaf index -> box: two dependent round trips, paid once per group)
        int32_t leaf[group];
MADRONA_UNROLL
        for (int32_t k = 0; k < group; k++) {
            leaf[k] = dfs_leaves_[g + k < n ? g + k : n - 1];
        }
        math::AABB box[group];
MADRONA_UNROLL
        for (int32_t k = 0; k < group; k++) {
            box[k] = leaf_aabbs_[leaf[k]];
        }
MADRONA_UNROLL
        for (int32_t k = 0; k < group; k++) {
            if (g + k < n && aabb.overlaps(box[k])) {
                fn(leaf_entities_[leaf[k]]);
            }
        }
    }
}

namespace detail {

// Leaf boxes are swept along the linear velocity and padded for acceleration
// so they stay valid across the whole step.
MADRONA_HD inline math::AABB expandAABBWithMotion(
    math::AABB aabb, const math::Vector3 &linear_velocity,
    float velocity_expansion, float accel_expansion)
{
MADRONA_UNROLL
    for (int32_t i = 0; i < 3; i++) {
        float pos_delta = velocity_expansion * linear_velocity[i];

        float min_delta = pos_delta - accel_expansion;
        float max_delta = pos_delta + accel_expansion;

        if (min_delta < 0.f) {
            aabb.pMin[i] += min_delta;
        }
        if (max_delta > 0.f) {
            aabb.pMax[i] += max_delta;
        }
    }

    return aabb;
}

// fetch-min / fetch-max on floats; returns the previous value
MADRONA_HD inline float fetchMinF(float *addr, float value, bool atomic)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (atomic) {
        if (!signbit(value)) {
            return __int_as_float(atomicMin((int *)addr, __float_as_int(value)));
        }
        return __uint_as_float(
            atomicMax((unsigned int *)addr, __float_as_uint(value)));
    }
#else
    (void)atomic;
#endif
    float old = *addr;
    if (value < old) {
        *addr = value;
    }
    return old;
}

MADRONA_HD inline float fetchMaxF(float *addr, float value, bool atomic)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (atomic) {
        if (!signbit(value)) {
            return __int_as_float(atomicMax((int *)addr, __float_as_int(value)));
        }
        return __uint_as_float(
            atomicMin((unsigned int *)addr, __float_as_uint(value)));
    }
#else
    (void)atomic;
#endif
    float old = *addr;
    if (value > old) {
        *addr = value;
    }
    return old;
}

}

void BVH::updateLeafPosition(LeafID leaf_id,
                             const math::Vector3 &pos,
                             const math::Quat &rot,
                             const math::Diag3x3 &scale,
                             const math::Vector3 &linear_vel,
                             const math::AABB &obj_aabb)
{
    math::AABB world_aabb = obj_aabb.applyTRS(pos, rot, scale);
    leaf_aabbs_[leaf_id.id] = detail::expandAABBWithMotion(
        world_aabb, linear_vel, leaf_velocity_expansion_, leaf_accel_expansion_);
    leaf_transforms_[leaf_id.id] = LeafTransform { pos, rot, scale };
    sorted_leaves_[leaf_id.id] = leaf_id.id;
}

math::AABB BVH::expandLeaf(LeafID leaf_id, const math::Vector3 &linear_vel)
{
    math::AABB expanded = detail::expandAABBWithMotion(
        leaf_aabbs_[leaf_id.id], linear_vel, leaf_velocity_expansion_,
        leaf_accel_expansion_);
    leaf_aabbs_[leaf_id.id] = expanded;
    return expanded;
}

// Grows the boxes on the path leaf -> root until nothing changes.  Several
// leaves of one world refit concurrently: upper levels use atomics.
void BVH::refitLeaf(LeafID leaf_id, const math::AABB &leaf_aabb)
{
    auto growChild = [&leaf_aabb](Node &node, int32_t c, bool atomic) {
        float x_min_prev = detail::fetchMinF(&node.minX[c], leaf_aabb.pMin.x, atomic);
        float y_min_prev = detail::fetchMinF(&node.minY[c], leaf_aabb.pMin.y, atomic);
        float z_min_prev = detail::fetchMinF(&node.minZ[c], leaf_aabb.pMin.z, atomic);
        float x_max_prev = detail::fetchMaxF(&node.maxX[c], leaf_aabb.pMax.x, atomic);
        float y_max_prev = detail::fetchMaxF(&node.maxY[c], leaf_aabb.pMax.y, atomic);
        float z_max_prev = detail::fetchMaxF(&node.maxZ[c], leaf_aabb.pMax.z, atomic);

        return leaf_aabb.pMin.x < x_min_prev || leaf_aabb.pMin.y < y_min_prev ||
               leaf_aabb.pMin.z < z_min_prev || leaf_aabb.pMax.x > x_max_prev ||
               leaf_aabb.pMax.y > y_max_prev || leaf_aabb.pMax.z > z_max_prev;
    };

    uint32_t leaf_parent = leaf_parents_[leaf_id.id];
    int32_t node_idx = (int32_t)(leaf_parent >> 2);
    int32_t sub_idx = (int32_t)(leaf_parent & 3u);

    // the leaf's own slot is touched by this thread only
    if (!growChild(nodes_[node_idx], sub_idx, false)) {
        return;
    }

    int32_t child_idx = node_idx;
    node_idx = nodes_[node_idx].parentID;

    while (node_idx != sentinel_) {
        Node &node = nodes_[node_idx];

        int32_t child_offset = 0;
        for (int32_t j = 0; j < 4; j++) {
            if (node.children[j] == child_idx) {
                child_offset = j;
                break;
            }
        }

        if (!growChild(node, child_offset, true)) {
            break;
        }

        child_idx = node_idx;
        node_idx = node.parentID;
    }
}

// updateLeafPosition + refitLeaf(leaf, its new box) in one call, ordered for
// the memory pipeline: everything the refit's first level reads (the leaf's
// parent slot) is fetched BEFORE this thread's first store -- on CDNA a load
// issued after a store waits for the store's acknowledgement, and the plain
// sequence (store leaf box, reload it, load the slot) pays that twice.  A tree
// that is about to be rebuilt is not refitted (the rebuild derives every box
// from the leaf boxes).
void BVH::updateLeafAndRefit(LeafID leaf_id,
                             const math::Vector3 &pos,
                             const math::Quat &rot,
                             const math::Diag3x3 &scale,
                             const math::Vector3 &linear_vel,
                             const math::AABB &obj_aabb)
{
    const int32_t leaf = leaf_id.id;

    // (loads through the global address space, a round at a time: what the
    // object holds; the leaf's parent; its slot in the parent node.  Through
    // `this` every member and every element was a round trip of its own.)
    const RefitView view = loadRefitView(this);
    uint32_t leaf_parent = 0;
    math::AABB slot = math::AABB::invalid();
    if (view.refit) {
        leaf_parent = mwhip::loadGlobal(view.leafParents + leaf);
        slot = loadSlotBounds(view.nodes, leaf_parent);
    }
    applyLeafUpdate(view, leaf, leaf_parent, slot, pos, rot, scale, linear_vel,
                    obj_aabb);
}

void BVH::applyLeafUpdate(const RefitView &view, int32_t leaf,
                          uint32_t leaf_parent, math::AABB slot,
                          const math::Vector3 &pos, const math::Quat &rot,
                          const math::Diag3x3 &scale,
                          const math::Vector3 &linear_vel,
                          const math::AABB &obj_aabb)
{
    using mwhip::storeGlobal;

    math::AABB world_aabb = obj_aabb.applyTRS(pos, rot, scale);
    math::AABB leaf_aabb = detail::expandAABBWithMotion(
        world_aabb, linear_vel, view.velocityExpansion, view.accelExpansion);

    storeGlobal(view.leafAABBs + leaf, leaf_aabb);
    storeGlobal((LeafTransform *)view.leafTransforms + leaf,
                LeafTransform { pos, rot, scale });
    storeGlobal(view.sortedLeaves + leaf, leaf);

    if (!view.refit) {
        return;
    }

    // the leaf's own slot is touched by this thread only: plain stores
    const int32_t node_idx = (int32_t)(leaf_parent >> 2);
    const int32_t sub_idx = (int32_t)(leaf_parent & 3u);
    Node *nodes = (Node *)view.nodes;
    Node &leaf_node = nodes[node_idx];
    bool grew = false;
    if (leaf_aabb.pMin.x < slot.pMin.x) {
        storeGlobal(&leaf_node.minX[sub_idx], leaf_aabb.pMin.x);
        grew = true;
    }
    if (leaf_aabb.pMin.y < slot.pMin.y) {
        storeGlobal(&leaf_node.minY[sub_idx], leaf_aabb.pMin.y);
        grew = true;
    }
    if (leaf_aabb.pMin.z < slot.pMin.z) {
        storeGlobal(&leaf_node.minZ[sub_idx], leaf_aabb.pMin.z);
        grew = true;
    }
    if (leaf_aabb.pMax.x > slot.pMax.x) {
        storeGlobal(&leaf_node.maxX[sub_idx], leaf_aabb.pMax.x);
        grew = true;
    }
    if (leaf_aabb.pMax.y > slot.pMax.y) {
        storeGlobal(&leaf_node.maxY[sub_idx], leaf_aabb.pMax.y);
        grew = true;
    }
    if (leaf_aabb.pMax.z > slot.pMax.z) {
        storeGlobal(&leaf_node.maxZ[sub_idx], leaf_aabb.pMax.z);
        grew = true;
    }
    if (!grew) {
        return;
    }

    growAncestors(nodes, node_idx, leaf_aabb);
}

// Upper levels of a refit: several leaves of a world grow them concurrently.
void BVH::growAncestors(int32_t child_idx, const math::AABB &leaf_aabb)
{
    growAncestors(mwhip::loadGlobal(&nodes_), child_idx, leaf_aabb);
}

void BVH::growAncestors(Node *nodes, int32_t child_idx,
                        const math::AABB &leaf_aabb)
{
    using mwhip::loadGlobal;
    int32_t node_idx = loadGlobal(&nodes[child_idx].parentID);

    while (node_idx != sentinel_) {
        Node &node = nodes[node_idx];

        // (which child this is, and where the walk goes next: one round)
        const int32_t children[4] = {
            loadGlobal(&node.children[0]), loadGlobal(&node.children[1]),
            loadGlobal(&node.children[2]), loadGlobal(&node.children[3]) };
        const int32_t parent = loadGlobal(&node.parentID);
        int32_t child_offset = 0;
        for (int32_t j = 3; j >= 0; j--) {
            if (children[j] == child_idx) {
                child_offset = j;
            }
        }

        const int32_t c = child_offset;
        float x_min_prev = detail::fetchMinF(&node.minX[c], leaf_aabb.pMin.x, true);
        float y_min_prev = detail::fetchMinF(&node.minY[c], leaf_aabb.pMin.y, true);
        float z_min_prev = detail::fetchMinF(&node.minZ[c], leaf_aabb.pMin.z, true);
        float x_max_prev = detail::fetchMaxF(&node.maxX[c], leaf_aabb.pMax.x, true);
        float y_max_prev = detail::fetchMaxF(&node.maxY[c], leaf_aabb.pMax.y, true);
        float z_max_prev = detail::fetchMaxF(&node.maxZ[c], leaf_aabb.pMax.z, true);

        const bool grew =
            leaf_aabb.pMin.x < x_min_prev || leaf_aabb.pMin.y < y_min_prev ||
            leaf_aabb.pMin.z < z_min_prev || leaf_aabb.pMax.x > x_max_prev ||
            leaf_aabb.pMax.y > y_max_prev || leaf_aabb.pMax.z > z_max_prev;
        if (!grew) {
            break;
        }

        child_idx = node_idx;
        node_idx = parent;
    }
}

// Partitions sorted_leaves_[base, base + num_elems) about the midpoint of the
// centroid bounds on the widest axis; returns the size of the lower half.
int32_t BVH::midpointSplit(int32_t base, int32_t num_elems)
{
    using math::Vector3;

    Vector3 center_min { FLT_MAX, FLT_MAX, FLT_MAX };
    Vector3 center_max { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (int32_t i = 0; i < num_elems; i++) {
        Vector3 center = leafCenter(base + i);
        center_min = Vector3::min(center_min, center);
        center_max = Vector3::max(center_max, center);
    }

    Vector3 center_diff = center_max - center_min;
    int32_t axis;
    if (center_diff.x > center_diff.y && center_diff.x > center_diff.z) {
        axis = 0;
    } else if (center_diff.y > center_diff.x && center_diff.y > center_diff.z) {
        axis = 1;
    } else {
        axis = 2;
    }

    float split_val = 0.5f * (center_min[axis] + center_max[axis]);

    int32_t start = 0;
    int32_t end = num_elems;
    while (start < end) {
        while (start < end && leafCenter(base + start)[axis] < split_val) {
            ++start;
        }

        while (start < end && leafCenter(base + end - 1)[axis] >= split_val) {
            --end;
        }

        if (start < end) {
            int32_t tmp = sorted_leaves_[base + start];
            sorted_leaves_[base + start] = sorted_leaves_[base + end - 1];
            sorted_leaves_[base + end - 1] = tmp;
            ++start;
            --end;
        }
    }

    if (start > 0 && start < num_elems) {
        return start;
    }
    return num_elems / 2;
}

// Top-down build, one thread per world, explicit stack.  An interior node
// splits its range in four (midpoint split, then each half again); its entry
// stays on the stack until its four sub-ranges have been built, then its
// merged bounds go into the first free child slot of its parent.
void BVH::rebuild()
{
    RebuildStackEntry stack[rebuildStackSize];
    rebuild(stack);
}

// `stack`: rebuildStackSize entries of working storage.  On the GPU a private
// array of this size lives in scratch memory, where every push / pop is a
// memory round trip; the staged rebuild passes LDS instead.
void BVH::rebuild(RebuildStackEntry *stack)
{
    using StackEntry = RebuildStackEntry;

    const int32_t num_leaves = num_leaves_;
    num_tree_leaves_ = num_leaves;
    num_nodes_ = numInternalNodes(num_leaves);

    stack[0] = StackEntry { sentinel_, sentinel_, 0, num_leaves };
    CountT stack_size = 1;

    int32_t cur_node_offset = 0;

    while (stack_size > 0) {
        StackEntry &entry = stack[stack_size - 1];
        int32_t node_id;

        if (entry.numObjs <= 4) {
            node_id = cur_node_offset++;
            Node &node = nodes_[node_id];
            node.parentID = entry.parentID;

            for (int32_t i = 0; i < 4; i++) {
                if (i < entry.numObjs) {
                    int32_t leaf_id = sorted_leaves_[entry.offset + i];
                    leaf_parents_[leaf_id] =
                        ((uint32_t)node_id << 2) | (uint32_t)i;
                    node.setLeaf(i, leaf_id);
                    node.setBounds(i, leaf_aabbs_[leaf_id]);
                } else {
                    node.children[i] = sentinel_;
                    node.setBounds(i, math::AABB::invalid());
                }
            }
        } else if (entry.nodeID == sentinel_) {
            node_id = cur_node_offset++;
            entry.nodeID = node_id;

            Node &node = nodes_[node_id];
            for (int32_t i = 0; i < 4; i++) {
                node.children[i] = sentinel_;
            }
            node.parentID = entry.parentID;

            const int32_t offset = entry.offset;
            const int32_t num_objs = entry.numObjs;

            int32_t second_split = midpointSplit(offset, num_objs);
            int32_t num_h1 = second_split;
            int32_t num_h2 = num_objs - second_split;

            int32_t first_split = midpointSplit(offset, num_h1);
            int32_t third_split = midpointSplit(offset + second_split, num_h2);

            // pushed in reverse so the quarters are built left to right
            stack[stack_size++] = StackEntry {
                sentinel_, node_id, offset + num_h1 + third_split,
                num_h2 - third_split };
            stack[stack_size++] = StackEntry {
                sentinel_, node_id, offset + num_h1, third_split };
            stack[stack_size++] = StackEntry {
                sentinel_, node_id, offset + first_split, num_h1 - first_split };
            stack[stack_size++] = StackEntry {
                sentinel_, node_id, offset, first_split };

            continue;
        } else {
            node_id = entry.nodeID;
        }

        stack_size -= 1;

        Node &node = nodes_[node_id];
        if (node.parentID == sentinel_) {
            continue;
        }

        math::AABB combined = math::AABB::invalid();
        for (int32_t i = 0; i < 4; i++) {
            if (!node.hasChild(i)) {
                break;
            }
            combined = math::AABB::merge(combined, node.bounds(i));
        }

        Node &parent = nodes_[node.parentID];
        int32_t child_offset = 0;
        while (parent.children[child_offset] != sentinel_) {
            child_offset++;
        }

        parent.children[child_offset] = node_id;
        parent.setBounds(child_offset, combined);
    }

    // record the order an unpruned traversal visits the leaves in
    {
        // (the build stack is free again: reuse it)
        static_assert(sizeof(RebuildStackEntry) * rebuildStackSize >=
                      32 * sizeof(int32_t));
        int32_t *visit = (int32_t *)stack;
        visit[0] = 0;
        CountT visit_size = 1;
        int32_t rank = 0;
        while (visit_size > 0 && num_leaves > 0) {
            const Node &node = nodes_[visit[--visit_size]];
            for (CountT c = 0; c < 4; c++) {
                if (!node.hasChild(c)) {
                    continue;
                }
                if (node.isLeaf(c)) {
                    dfs_leaves_[rank++] = node.leafIDX(c);
                } else {
                    visit[visit_size++] = node.children[c];
                }
            }
        }
    }
}

#if defined(__HIPCC__)
// ---------------------------------------------------------------------------
// The rebuild on a wavefront.  Run by one lane, rebuild() is a chain of a few
// thousand dependent LDS accesses (~100 us per world on MI355X); here the same
// state machine advances with all lanes: a range's leaves sit one per lane for
// the midpoint splits (bounds by butterfly reduction, the Hoare partition's
// swaps computed from two ballots), the four children of a node are written by
// four lanes, merged bounds by six (one per box component), stack entries move
// as 16-byte records.  The tree, the leaf order and the traversal order are
// those of rebuild(), exactly: every decision is an integer function of the
// same float comparisons.
// ---------------------------------------------------------------------------
namespace detail {

__device__ inline void waveFence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ inline uint64_t lanesBelow(uint32_t lane)
{
    return (1ull << lane) - 1ull;
}

}

// rebuild()'s midpointSplit: lanes [0, num_elems) hold the range's elements.
// The sequential partition swaps the k-th element >= split_val from the left
// with the k-th element < split_val from the right until the scans meet, i.e.
// exactly the elements on the wrong side of the final boundary, paired in
// that order.  dfs_leaves_ (unused until the end of the build) is the
// exchange buffer.
int32_t BVH::midpointSplitWave(uint32_t lane, int32_t base, int32_t num_elems)
{
    using math::Vector3;

    const bool active = (int32_t)lane < num_elems;
    int32_t leaf = 0;
    Vector3 center { 0.f, 0.f, 0.f };
    if (active) {
        leaf = sorted_leaves_[base + (int32_t)lane];
        center = leaf_centers_[leaf];
    }

    float lo[3], hi[3];
MADRONA_UNROLL
    for (int32_t a = 0; a < 3; a++) {
        lo[a] = active ? center[a] : FLT_MAX;
        hi[a] = active ? center[a] : -FLT_MAX;
    }
MADRONA_UNROLL
    for (uint32_t d = 32; d > 0; d >>= 1) {
MADRONA_UNROLL
        for (int32_t a = 0; a < 3; a++) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], d, 64));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], d, 64));
        }
    }

    Vector3 center_min { lo[0], lo[1], lo[2] };
    Vector3 center_max { hi[0], hi[1], hi[2] };
    Vector3 center_diff = center_max - center_min;
    int32_t axis;
    if (center_diff.x > center_diff.y && center_diff.x > center_diff.z) {
        axis = 0;
    } else if (center_diff.y > center_diff.x && center_diff.y > center_diff.z) {
        axis = 1;
    } else {
        axis = 2;
    }

    float split_val = 0.5f * (center_min[axis] + center_max[axis]);

    const float v = axis == 0 ? center.x : axis == 1 ? center.y : center.z;
    const bool below = active && v < split_val;
    const bool above = active && v >= split_val;
    const uint64_t below_mask = __builtin_amdgcn_ballot_w64(below);
    const uint64_t above_mask = __builtin_amdgcn_ballot_w64(above);

    const int32_t boundary = (int32_t)__builtin_popcountll(below_mask);
    const uint64_t left = boundary >= 64 ? ~0ull : ((1ull << boundary) - 1ull);
    const uint64_t wrong_above = above_mask & left;      // move right
    const uint64_t wrong_below = below_mask & ~left;     // move left

    if (wrong_above != 0) {
        int32_t *from_left = dfs_leaves_;
        int32_t *from_right = dfs_leaves_ + 32;
        const bool moves_right = ((wrong_above >> lane) & 1ull) != 0;
        const bool moves_left = ((wrong_below >> lane) & 1ull) != 0;
        uint32_t k = 0;
        if (moves_right) {          // k-th from the left
            k = (uint32_t)__builtin_popcountll(
                wrong_above & detail::lanesBelow(lane));
            from_left[k] = leaf;
        } else if (moves_left) {    // k-th from the right
            k = (uint32_t)__builtin_popcountll(
                wrong_below & ~(detail::lanesBelow(lane) | (1ull << lane)));
            from_right[k] = leaf;
        }
        detail::waveFence();
        if (moves_right) {
            sorted_leaves_[base + (int32_t)lane] = from_right[k];
        } else if (moves_left) {
            sorted_leaves_[base + (int32_t)lane] = from_left[k];
        }
        detail::waveFence();
    }

    if (boundary > 0 && boundary < num_elems) {
        return boundary;
    }
    return num_elems / 2;
}

int32_t BVH::rebuildStagedWave(uint32_t lane, RebuildStackEntry *stack)
{
    const int32_t num_leaves = num_leaves_;
    num_tree_leaves_ = num_leaves;
    num_nodes_ = numInternalNodes(num_leaves);

    if (lane == 0) {
        stack[0] = RebuildStackEntry { sentinel_, sentinel_, 0, num_leaves };
    }
    detail::waveFence();

    int32_t stack_size = 1;
    int32_t cur_node_offset = 0;

    while (stack_size > 0) {
        const RebuildStackEntry entry = stack[stack_size - 1];
        int32_t node_id;

        if (entry.numObjs <= 4) {
            node_id = cur_node_offset++;
            Node &node = nodes_[node_id];
            if (lane == 0) {
                node.parentID = entry.parentID;
            }
            if (lane < 4) {
                const int32_t i = (int32_t)lane;
                if (i < entry.numObjs) {
                    int32_t leaf_id = sorted_leaves_[entry.offset + i];
                    leaf_parents_[leaf_id] =
                        ((uint32_t)node_id << 2) | (uint32_t)i;
                    node.setLeaf(i, leaf_id);
                    node.setBounds(i, leaf_aabbs_[leaf_id]);
                } else {
                    node.children[i] = sentinel_;
                    node.setBounds(i, math::AABB::invalid());
                }
            }
            detail::waveFence();
        } else if (entry.nodeID == sentinel_) {
            node_id = cur_node_offset++;

            Node &node = nodes_[node_id];
            if (lane == 0) {
                stack[stack_size - 1].nodeID = node_id;
                node.parentID = entry.parentID;
            }
            if (lane < 4) {
                node.children[lane] = sentinel_;
            }

            const int32_t offset = entry.offset;
            const int32_t num_objs = entry.numObjs;

            int32_t second_split = midpointSplitWave(lane, offset, num_objs);
            int32_t num_h1 = second_split;
            int32_t num_h2 = num_objs - second_split;

            int32_t first_split = midpointSplitWave(lane, offset, num_h1);
            int32_t third_split =
                midpointSplitWave(lane, offset + second_split, num_h2);

            // pushed in reverse so the quarters are built left to right
            if (lane < 4) {
                RebuildStackEntry pushed;
                pushed.nodeID = sentinel_;
                pushed.parentID = node_id;
                if (lane == 0) {
                    pushed.offset = offset + num_h1 + third_split;
                    pushed.numObjs = num_h2 - third_split;
                } else if (lane == 1) {
                    pushed.offset = offset + num_h1;
                    pushed.numObjs = third_split;
                } else if (lane == 2) {
                    pushed.offset = offset + first_split;
                    pushed.numObjs = num_h1 - first_split;
                } else {
                    pushed.offset = offset;
                    pushed.numObjs = first_split;
                }
                stack[stack_size + (int32_t)lane] = pushed;
            }
            stack_size += 4;
            detail::waveFence();

            continue;
        } else {
            node_id = entry.nodeID;
        }

        stack_size -= 1;

        // (node.parentID is the entry's in every case)
        if (entry.parentID == sentinel_) {
            continue;
        }

        Node &node = nodes_[node_id];
        Node &parent = nodes_[entry.parentID];

        // first free child slot of the parent
        const uint64_t free_slots = __builtin_amdgcn_ballot_w64(
            lane < 4 && parent.children[lane < 4 ? lane : 0] == sentinel_);
        const int32_t child_offset = (int32_t)__builtin_ctzll(free_slots);

        // merged bounds of the node's children, one box component per lane
        // (the Node is six float[4] arrays: minX minY minZ maxX maxY maxZ)
        if (lane < 6) {
            const float *component = (const float *)&node + lane * 4;
            const bool is_min = lane < 3;
            float merged = is_min ? FLT_MAX : -FLT_MAX;
            for (int32_t i = 0; i < 4; i++) {
                if (!node.hasChild(i)) {
                    break;
                }
                merged = is_min ? fminf(merged, component[i]) :
                                  fmaxf(merged, component[i]);
            }
            ((float *)&parent)[lane * 4 + (uint32_t)child_offset] = merged;
        }
        if (lane == 0) {
            parent.children[child_offset] = node_id;
        }
        detail::waveFence();
    }

    // record the order an unpruned traversal visits the leaves in
    {
        int32_t *visit = (int32_t *)stack;
        if (lane == 0) {
            visit[0] = 0;
        }
        detail::waveFence();

        int32_t visit_size = 1;
        int32_t rank = 0;
        while (visit_size > 0 && num_leaves > 0) {
            const Node &node = nodes_[visit[--visit_size]];
            const int32_t child =
                lane < 4 ? node.children[lane] : sentinel_;
            const bool has_child = child != sentinel_;
            const bool is_leaf =
                has_child && ((uint32_t)child & leaf_bit_) != 0;
            const bool is_inner = has_child && !is_leaf;
            const uint64_t leaf_mask = __builtin_amdgcn_ballot_w64(is_leaf);
            const uint64_t inner_mask = __builtin_amdgcn_ballot_w64(is_inner);
            // the node index was read before this iteration's pushes
            detail::waveFence();
            if (is_leaf) {
                dfs_leaves_[rank + (int32_t)__builtin_popcountll(
                    leaf_mask & detail::lanesBelow(lane))] =
                        (int32_t)((uint32_t)child & ~leaf_bit_);
            }
            if (is_inner) {
                visit[visit_size + (int32_t)__builtin_popcountll(
                    inner_mask & detail::lanesBelow(lane))] = child;
            }
            rank += (int32_t)__builtin_popcountll(leaf_mask);
            visit_size += (int32_t)__builtin_popcountll(inner_mask);
            detail::waveFence();
        }
    }

    return (int32_t)num_nodes_;
}

// rebuild() breadth first (broadphase.hpp).  Lane p holds sorted position p.
// Equivalence with the stack machine: a midpoint split only reads the centres of
// its own range and only permutes sorted_leaves_ inside it, so splitting
// disjoint ranges in any order -- or at once -- gives the same permutation; node
// ids are rebuild()'s pre-order numbers (a node is numbered when it is first
// reached, children 0..3 in order), children fill their parent's slots in that
// order, merged bounds are min / max over the same boxes (exact, order free),
// and the traversal order is the one the unpruned stack walk produces: a node's
// inner children are pushed 0..3 and popped 3..0.
int32_t BVH::rebuildStagedSegmented(uint32_t lane, SegmentedScratch *scratch)
{
    using math::Vector3;
    const int32_t num_leaves = num_leaves_;
    num_tree_leaves_ = num_leaves;
    num_nodes_ = numInternalNodes(num_leaves);
    RebuildRange *ranges = scratch->ranges;

    if (lane == 0) {
        ranges[0] = RebuildRange { 0, (int16_t)num_leaves, -1, -1, 0, 1, 0, 0 };
    }
    detail::waveFence();

    const int32_t p = (int32_t)lane;
    // the range this lane's position is in while it is being split (-1: settled)
    int32_t my_range = p < num_leaves && num_leaves > 4 ? 0 : -1;
    int32_t num_ranges = 1;
    int32_t level_first = 0;        // ranges of the current level: [first, num_ranges)
    int32_t depth = 0;

    // one segmented midpoint split of [s_lo, s_lo + s_n) for every lane with
    // s_n > 0; returns the size of the lower part (midpointSplit's value)
    auto splitSegments = [&](int32_t s_lo, int32_t s_n) -> int32_t {
        const bool active = s_n > 0;
        int32_t leaf = 0;
        Vector3 center { 0.f, 0.f, 0.f };
        if (active) {
            leaf = sorted_leaves_[p];
            center = leaf_centers_[leaf];
        }
        const int32_t s_end = s_lo + s_n;
        float lo[3], hi[3];
MADRONA_UNROLL
        for (int32_t a = 0; a < 3; a++) {
            lo[a] = active ? center[a] : FLT_MAX;
            hi[a] = active ? center[a] : -FLT_MAX;
        }
        // (suffix reduction inside the segment, then everyone reads the head)
MADRONA_UNROLL
        for (int32_t d = 1; d < 64; d <<= 1) {
            const bool take = active && p + d < s_end;
MADRONA_UNROLL
            for (int32_t a = 0; a < 3; a++) {
                const float l = __shfl_down(lo[a], (uint32_t)d, 64);
                const float h = __shfl_down(hi[a], (uint32_t)d, 64);
                if (take) {
                    lo[a] = fminf(lo[a], l);
                    hi[a] = fmaxf(hi[a], h);
                }
            }
        }
MADRONA_UNROLL
        for (int32_t a = 0; a < 3; a++) {
            lo[a] = __shfl(lo[a], active ? s_lo : p, 64);
            hi[a] = __shfl(hi[a], active ? s_lo : p, 64);
        }

        Vector3 center_min { lo[0], lo[1], lo[2] };
        Vector3 center_max { hi[0], hi[1], hi[2] };
        Vector3 center_diff = center_max - center_min;
        int32_t axis;
        if (center_diff.x > center_diff.y && center_diff.x > center_diff.z) {
            axis = 0;
        } else if (center_diff.y > center_diff.x && center_diff.y > center_diff.z) {
            axis = 1;
        } else {
            axis = 2;
        }
        const float split_val = 0.5f * (center_min[axis] + center_max[axis]);
        const float v = axis == 0 ? center.x : axis == 1 ? center.y : center.z;
        const bool below = active && v < split_val;
        const bool above = active && v >= split_val;
        const uint64_t below_mask = __builtin_amdgcn_ballot_w64(below);
        const uint64_t above_mask = __builtin_amdgcn_ballot_w64(above);

        const uint64_t seg_mask = !active ? 0ull :
            (s_n >= 64 ? ~0ull : (((1ull << s_n) - 1ull) << s_lo));
        const int32_t boundary =
            (int32_t)__builtin_popcountll(below_mask & seg_mask);
        const uint64_t left = !active ? 0ull :
            (boundary >= 64 ? ~0ull : (((1ull << boundary) - 1ull) << s_lo));
        const uint64_t wrong_above = above_mask & seg_mask & left;      // move right
        const uint64_t wrong_below = below_mask & seg_mask & ~left;     // move left

        const bool moves_right = ((wrong_above >> lane) & 1ull) != 0;
        const bool moves_left = ((wrong_below >> lane) & 1ull) != 0;
        const bool any_moves =
            __builtin_amdgcn_ballot_w64(moves_right || moves_left) != 0ull;
        if (any_moves) {
            uint32_t k = 0;
            if (moves_right) {          // k-th of its segment from the left
                k = (uint32_t)__builtin_popcountll(
                    wrong_above & detail::lanesBelow(lane));
                scratch->fromLeft[s_lo + (int32_t)k] = leaf;
            } else if (moves_left) {    // k-th of its segment from the right
                k = (uint32_t)__builtin_popcountll(
                    wrong_below & ~(detail::lanesBelow(lane) | (1ull << lane)));
                scratch->fromRight[s_lo + (int32_t)k] = leaf;
            }
            detail::waveFence();
            if (moves_right) {
                sorted_leaves_[p] = scratch->fromRight[s_lo + (int32_t)k];
            } else if (moves_left) {
                sorted_leaves_[p] = scratch->fromLeft[s_lo + (int32_t)k];
            }
            detail::waveFence();
        }

        if (boundary > 0 && boundary < s_n) {
            return boundary;
        }
        return s_n / 2;
    };

    bool overflow = false;
    while (__builtin_amdgcn_ballot_w64(my_range >= 0) != 0ull) {
        depth++;
        // ---- the half split of every range of the level with more than 4 ----
        int32_t r_lo = 0, r_n = 0;
        if (my_range >= 0) {
            r_lo = ranges[my_range].lo;
            r_n = ranges[my_range].n;
        }
        const int32_t h1 = splitSegments(r_lo, r_n);
        // ---- both quarter splits ----
        const bool upper = my_range >= 0 && p >= r_lo + h1;
        const int32_t q_lo = my_range < 0 ? 0 : (upper ? r_lo + h1 : r_lo);
        const int32_t q_n = my_range < 0 ? 0 : (upper ? r_n - h1 : h1);
        const int32_t q = splitSegments(q_lo, q_n);
        // (the head of the range needs the split of BOTH halves: the upper
        // half's from its first lane; h1 is in [1, r_n - 1], so both exist)
        const int32_t first_split = __shfl(q, my_range >= 0 ? r_lo : p, 64);
        const int32_t third_split = __shfl(q, my_range >= 0 ? r_lo + h1 : p, 64);

        // ---- the four children of every split range: records ----
        const bool head = my_range >= 0 && p == r_lo;
        const uint64_t heads = __builtin_amdgcn_ballot_w64(head);
        const int32_t num_split = (int32_t)__builtin_popcountll(heads);
        const int32_t child_base = num_ranges + 4 * (int32_t)__builtin_popcountll(
            heads & detail::lanesBelow(lane));
        if (num_ranges + 4 * num_split > maxRebuildRanges) {
            overflow = true;
            break;
        }
        const int32_t c_lo[4] = { r_lo, r_lo + first_split, r_lo + h1,
                                  r_lo + h1 + third_split };
        const int32_t c_n[4] = { first_split, h1 - first_split, third_split,
                                 r_n - h1 - third_split };
        if (head) {
            ranges[my_range].firstChild = (int16_t)child_base;
MADRONA_UNROLL
            for (int32_t c = 0; c < 4; c++) {
                ranges[child_base + c] = RebuildRange {
                    (int16_t)c_lo[c], (int16_t)c_n[c], (int16_t)my_range, -1,
                    0, 1, 0, (int16_t)c };
            }
        }
        // where this lane's position went
        if (my_range >= 0) {
            const int32_t base = __shfl(child_base, r_lo, 64);
            int32_t c = p >= c_lo[3] ? 3 : (p >= c_lo[2] ? 2 : (p >= c_lo[1] ? 1 : 0));
            my_range = c_n[c] > 4 ? base + c : -1;
        }
        level_first = num_ranges;
        num_ranges += 4 * num_split;
        detail::waveFence();
    }
    (void)level_first;
    if (overflow) {
        return -1;
    }

    // ---- subtree sizes, bottom up (a child's index is above its parent's) ----
    for (int32_t it = 0; it < depth; it++) {
        for (int32_t r = p; r < num_ranges; r += 64) {
            const int32_t fc = ranges[r].firstChild;
            if (fc >= 0) {
                ranges[r].subtreeNodes = (int16_t)(1 + ranges[fc].subtreeNodes +
                    ranges[fc + 1].subtreeNodes + ranges[fc + 2].subtreeNodes +
                    ranges[fc + 3].subtreeNodes);
            }
        }
        detail::waveFence();
    }
    // ---- node ids (pre-order) and traversal ranks (inner children 3..0), top down
    for (int32_t it = 0; it < depth; it++) {
        for (int32_t r = p; r < num_ranges; r += 64) {
            const int32_t par = ranges[r].parent;
            if (par >= 0) {
                const int32_t fc = ranges[par].firstChild;
                const int32_t c = ranges[r].slot;
                int32_t node = ranges[par].node + 1;
                int32_t rank = ranges[par].leafStart;
                for (int32_t o = 0; o < 4; o++) {
                    if (o < c) node += ranges[fc + o].subtreeNodes;
                    if (o > c) rank += ranges[fc + o].n;
                }
                ranges[r].node = (int16_t)node;
                ranges[r].leafStart = (int16_t)rank;
            }
        }
        detail::waveFence();
    }

    // ---- the nodes ----
    for (int32_t r = p; r < num_ranges; r += 64) {
        const RebuildRange range = ranges[r];
        Node &node = nodes_[range.node];
        node.parentID = range.parent >= 0 ? (int32_t)ranges[range.parent].node :
                                           sentinel_;
        if (range.firstChild < 0) {
            for (int32_t i = 0; i < 4; i++) {
                if (i < range.n) {
                    const int32_t leaf_id = sorted_leaves_[range.lo + i];
                    leaf_parents_[leaf_id] =
                        ((uint32_t)range.node << 2) | (uint32_t)i;
                    node.setLeaf(i, leaf_id);
                    node.setBounds(i, leaf_aabbs_[leaf_id]);
                    dfs_leaves_[range.leafStart + i] = leaf_id;
                } else {
                    node.children[i] = sentinel_;
                    node.setBounds(i, math::AABB::invalid());
                }
            }
        } else {
            for (int32_t c = 0; c < 4; c++) {
                const RebuildRange child = ranges[range.firstChild + c];
                math::AABB merged = math::AABB::invalid();
                for (int32_t i = 0; i < child.n; i++) {
                    merged = math::AABB::merge(
                        merged, leaf_aabbs_[sorted_leaves_[child.lo + i]]);
                }
                node.children[c] = child.node;
                node.setBounds(c, merged);
            }
        }
    }
    detail::waveFence();
    return (int32_t)num_nodes_;
}
#endif

void BVH::updateTree()
{
    if (force_rebuild_) {
        force_rebuild_ = false;
        rebuild();
    }
}

}

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

void BVH::updateTree()
{
    if (force_rebuild_) {
        force_rebuild_ = false;
        rebuild();
    }
}

}

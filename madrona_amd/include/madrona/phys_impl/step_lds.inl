// Part of the fused rigid-body step (phys_impl/world_step.inl includes the parts
// in order, inside namespace madrona::phys::kernels): the step with the world resident in LDS (physicsStepLdsKernel): block layout, primitive staging, the per-launch frame, the kernel.

// ===========================================================================
// The same step with the world resident in LDS.
//
// The generic kernel above is bound by dependent HBM/L2 round trips (each
// constraint chases Loc -> column pointer -> row for both bodies; measured:
// ~870 K cycles per world-step, > 95 % of them waiting).  A world's rigid-body
// state is a few KB, so this variant loads it once (coalesced, all lanes), runs
// candidates + every substep against LDS, and stores positions / velocities /
// solver state back once.  HBM traffic per world-step: one read and one write
// of the body columns; everything else stays on the CU.
//
// Candidate pairs come from the BVH without walking it: the traversal reports a
// leaf iff the query box overlaps the leaf's slot box (ancestor boxes are
// supersets), in an order that does not depend on the query
// (BVH::traversalOrder), so lane a tests its box against the slot boxes in
// that order.
//
// MAXB bounds bodies per world (LDS capacity); the host picks the instantiation
// for the bulk of the worlds, and a world that exceeds it -- or the block's
// contact capacity -- is handed to physicsStepKernel (step_hbm.inl, fallback
// mode) in the same step.
// ===========================================================================
// candidate pair of the LDS step: body indices inside the world
// (4 bytes: MAXB <= 128 bodies, < 256 primitives per object)
struct WaveCandidate {
    uint8_t a;
    uint8_t b;
    uint8_t aPrim;
    uint8_t bPrim;
};

// A contact of the LDS step: the bodies are indices inside the world, so the two
// Locs and the point count of a ContactConstraint (96 B) fit one word (80 B).
struct PackedContact {
    math::Vector4 points[4];    // xyz = world position, w = penetration depth
    math::Vector3 normal;
    uint32_t meta;              // ref body | alt body << 8 | numPoints << 16

    __device__ static inline PackedContact pack(const ContactConstraint &c)
    {
        return PackedContact {
            { c.points[0], c.points[1], c.points[2], c.points[3] }, c.normal,
            (uint32_t)c.ref.row | ((uint32_t)c.alt.row << 8) |
                ((uint32_t)c.numPoints << 16),
        };
    }
    // what the solver reads of a contact (xpbd.hpp, ContactT): the points stay
    // where they are -- its loops over them index memory, not registers
    struct View {
        Loc ref, alt;
        const math::Vector4 *points;
        int32_t numPoints;
        math::Vector3 normal;
    };
    __device__ inline View view() const
    {
        return View {
            Loc { 0, (int32_t)(meta & 0xFFu) },
            Loc { 0, (int32_t)((meta >> 8) & 0xFFu) },
            points, (int32_t)(meta >> 16), normal,
        };
    }
    __device__ inline int32_t refBody() const { return (int32_t)(meta & 0xFFu); }
    __device__ inline int32_t altBody() const { return (int32_t)((meta >> 8) & 0xFFu); }
};
static_assert(sizeof(PackedContact) == 80);

// (what the narrowphase routines call to hand over their result, physics.inl:
// straight into the packed record -- found through its namespace)
__device__ inline void manifoldToContact(const narrowphase::Manifold &manifold,
                                         Loc ref_loc, Loc other_loc,
                                         PackedContact *out)
{
    for (int i = 0; i < 4; i++) {
        out->points[i] = math::Vector4::fromVec3W(manifold.contactPoints[i],
                                                  manifold.penetrationDepths[i]);
    }
    out->normal = manifold.normal;
    out->meta = (uint32_t)ref_loc.row | ((uint32_t)other_loc.row << 8) |
        ((uint32_t)manifold.numContactPoints << 16);
}

__device__ inline void sphereToContact(const narrowphase::SphereContact &sphere,
                                       Loc a_loc, Loc b_loc, PackedContact *out)
{
    out->points[0] = math::Vector4::fromVec3W(sphere.pt, sphere.depth);
    out->points[1] = math::Vector4::zero();
    out->points[2] = math::Vector4::zero();
    out->points[3] = math::Vector4::zero();
    out->normal = sphere.normal;
    out->meta = (uint32_t)b_loc.row | ((uint32_t)a_loc.row << 8) | (1u << 16);
}

// Narrowphase scratch of a world in the LDS step: clipping rows of the lanes that
// test a pair on their own, and the scratch of ONE cooperative hull-hull test
// (a world runs its hull-hull pairs one after the other, with all of its
// lanes).  One world per wavefront: both side by side (WaveScratch).  Two
// worlds per wavefront: the rows are dead when the hull-hull tests of a chunk
// start, the two share their storage; rows of four corners (what the faces of
// boxes and wedges need; a hull with larger faces takes the per-lane HBM path),
// 21 of them -- a chunk of the Escape Room holds ~13 hull-plane pairs.
// (Rounds 3-5 gave a 32-lane world two scratch blocks and ran two tests side by
// side, 16 lanes each.  With two small blocks -- HullScratchT<8, 12> -- in the
// same storage that is 372 spilled dwords instead of 301 under the 256-register
// cap and 593 us against 576: profiles/r06_hull_variants.jsonl.)
template <int LPW>
struct BlockScratch {
    static constexpr uint32_t polyVerts = lanePolyVerts;
    static constexpr uint32_t polyDwords = lanePolyDwords;
    static constexpr uint32_t polyRows = lanePolyRows;
    WaveScratch both;
    __device__ inline float *lanePoly() { return both.lanePoly; }
    __device__ inline HullScratch *hull() { return &both.hull; }
};

template <>
struct BlockScratch<32> {
    static constexpr uint32_t polyVerts = 4;
    static constexpr uint32_t polyDwords = polyVerts * 4 + 1;
    static constexpr uint32_t polyRows = 21;
    union alignas(16) {
        float rows[polyRows * polyDwords];
        HullScratch hullScratch;
    };
    static_assert(sizeof(float) * polyRows * polyDwords <= sizeof(HullScratch));
    __device__ inline float *lanePoly() { return rows; }
    __device__ inline HullScratch *hull() { return &hullScratch; }
};

// The object manager's primitives (and as many hull meshes as fit) copied next
// to the CU; hull pointers inside `prims` are rebased onto `arena`.  One copy
// per WORKGROUP: the worlds of a wavefront share it (an executor has one
// object manager; a world with a manager of its own reads it from HBM).
struct PrimBlock {
    static constexpr int maxPrims = (int)PrimImage::maxPrims;   // more: hull data stays in HBM
    static constexpr int arenaDwords = (int)PrimImage::arenaDwords;   // object-space hull meshes
    CollisionPrimitive prims[maxPrims];
    math::AABB primAABBs[maxPrims];
    const void *primMeshKey[maxPrims];      // HBM vertex array of each hull prim
    alignas(16) uint32_t arena[arenaDwords];
};

template <int MAXB, int LPW = 64>
struct WorldBlock {
    static constexpr int maxBodies = MAXB;
    // sized so that a 32-body block stays under 20 KB of LDS: eight
    // single-wave workgroups per CU, two per SIMD -- what the register cap of
    // the kernel admits.  Six candidates per body in LDS (a dense pile of n
    // bodies has up to n (n - 1) / 2 pairs); more spill to HBM.
    // (two worlds per wavefront: three per body and as many contacts as lanes --
    // the Escape Room holds ~21 candidates and ~14 contacts per world; a world
    // with more candidates spills them, one with more contacts takes the HBM
    // kernel for the step)
    static constexpr int maxCandidates = LPW == 32 ? MAXB * 3 : MAXB * 6;
    // (at least one per lane of the world: the narrowphase stages one contact
    // per lane)
    static constexpr int maxContacts = LPW == 32 ? LPW :
        (MAXB + MAXB / 4 > LPW ? MAXB + MAXB / 4 : LPW);
    // joints whose ROWS are kept in the block (more: their rows are read from
    // HBM in every substep) and joints whose body indices are (more: the world
    // takes the HBM kernel)
    static constexpr int maxJoints = 2;
    static constexpr int maxJointBodies = 8;

    // ---- the world image: what a step reads from the ECS tables -----------
    // The first imageBytes of this struct are exactly what physicsPackKernel
    // leaves per world in HBM (same layout), so that the step kernel starts
    // with ONE coalesced copy instead of a dozen dependent round trips
    // (row ranges -> column pointers -> body columns -> object metadata ->
    // leaf -> parent node slot), exposed at two waves per SIMD.
    // poses[0, MAXB): where the bodies are; behind them the substep's records of
    // the bodies the solver can change (solverSlot): where each was when the
    // substep began, then where the integration put it (same layout as
    // xpbd::SubstepPrevState / PreSolvePositional) -- one array, so that "the
    // record of body k" is an index and not a branch: an inert static body's
    // records ARE its pose (LdsBodyStore)
    struct Pose {
        math::Vector3 x;
        math::Quat q;
    };
    static_assert(sizeof(Pose) == sizeof(xpbd::SubstepPrevState) &&
                  sizeof(Pose) == sizeof(xpbd::PreSolvePositional));
    math::Diag3x3 scale[MAXB];
    Velocity vel[MAXB];
    // (what only the lane that owns a body ever looks at -- external force and
    // torque, where its row is, its entity id -- stays in that lane's
    // registers: LaneBodies)
    xpbd::BodyConstants constants[MAXB];    // zeroed for static bodies
    uint16_t primOffset[MAXB];
    uint8_t primCount[MAXB];
    uint8_t resp[MAXB];
    uint8_t solverKey[MAXB];                // ldsBodyKey: 0 = inert static body, else k + 1
    static_assert(MAXB < 255);

    // broadphase boxes are dead once the candidates exist: the contacts of the
    // substeps reuse their storage
    static constexpr size_t boxBytes =
        MAXB * (2 * sizeof(math::AABB) + 2 * sizeof(int32_t));
    static constexpr size_t contactBytes =
        maxContacts * sizeof(PackedContact);
    alignas(16) char shared[boxBytes > contactBytes ? boxBytes : contactBytes];
    // ---- end of the image (imageBytes below) ------------------------------

    // State of the substep for the bodies the solver can change (solverKey != 0;
    // slot = solverSlot[body]).  An inert static body -- most of an Escape
    // Room's: floor, borders, walls -- is where it was when the substep began
    // and has no pre-solve velocity: its records are made up from pos / rot
    // when they are asked for (LdsBodyStore) and when the step is stored.  Two
    // worlds per wavefront keep 20 slots for 32 bodies (a world with more
    // takes the HBM kernel; setupPhysicsStepTasks looks at the worlds before
    // it picks this layout).
    static constexpr int maxSolverBodies = LPW == 32 ? 20 : MAXB;
    static constexpr int prevBase = MAXB;
    static constexpr int prePosBase = MAXB + maxSolverBodies;
    Pose poses[MAXB + 2 * maxSolverBodies];
    xpbd::PreSolveVelocity preVel[maxSolverBodies];
    uint8_t solverSlot[MAXB];               // 0xFF: an inert static body
    static constexpr uint32_t noSlot = 0xFFu;
    WaveCandidate candidates[maxCandidates];
    float lambdas[maxContacts];

    JointConstraint joints[maxJoints];
    uint8_t jointBodies[maxJointBodies][2];
    PhysicsSystemState sys;
    BlockScratch<LPW> scratch;

    // by body: the box a body queries the tree with; by traversal rank: the
    // slot box the traversal tests and the entity id of that leaf's body
    __device__ inline math::AABB *queryBox() { return (math::AABB *)shared; }
    __device__ inline math::AABB *rankSlotBox()
    {
        return (math::AABB *)shared + MAXB;
    }
    // by traversal rank: { the leaf's entity id, packRankInfo(its body) } -- one
    // 8-byte read per box test gives everything the candidate pass asks of the
    // other body
    __device__ inline int2 *rankEntityInfo()
    {
        return (int2 *)((math::AABB *)shared + 2 * MAXB);
    }
    // body index | primitive count << 8 | (response type is Static) << 16
    __device__ static inline int32_t packRankInfo(int32_t body, uint32_t prim_count,
                                                  bool is_static)
    {
        return (int32_t)((uint32_t)body | (prim_count << 8) |
                         (is_static ? 1u << 16 : 0u));
    }
    static_assert(MAXB <= 256);
    __device__ inline PackedContact *contacts()
    {
        return (PackedContact *)shared;
    }
    // leaf id -> traversal rank, while the world is loaded (the candidate list
    // does not exist yet)
    __device__ inline uint16_t *leafRank() { return (uint16_t *)candidates; }
    static_assert(sizeof(WaveCandidate) * maxCandidates >= sizeof(uint16_t) * MAXB);

    // bytes of the world image (a multiple of 16)
    __host__ __device__ static constexpr size_t imageBytes()
    {
        return (__builtin_offsetof(WorldBlock, shared) + boxBytes + 15) &
            ~(size_t)15;
    }
};

// What only the lane that owns a body reads: kept in its registers from the
// load of the world to the store (chunk c of lane l = body c * LPW + l).
template <int CHUNKS>
struct LaneBodies {
    math::Vector3 extForce[CHUNKS];
    math::Vector3 extTorque[CHUNKS];
    Loc loc[CHUNKS];            // the body's row
    int32_t entityID[CHUNKS];
};

// Copies `count` dwords with `lanes` lanes (`lane` = this one's index among them).
__device__ inline void waveCopyDwords(uint32_t lane, uint32_t lanes, uint32_t *dst,
                                      const uint32_t *src, uint32_t count)
{
    // (batching eight loads ahead of the stores -- dst and src are generic
    // pointers -- was measured: 744 -> 753 us, the counts are a few hundred
    // dwords and the registers cost more than the L2 round trips)
    for (uint32_t i = lane; i < count; i += lanes) {
        dst[i] = src[i];
    }
}

// ... from HBM (global loads, four rounds of the lanes in flight: the copies
// of a primitive image are a few hundred dwords, one row of lanes at a time
// they were a dozen L2 round trips in a row)
__device__ inline void waveCopyDwordsGlobal(uint32_t lane, uint32_t lanes,
                                            uint32_t *dst, const uint32_t *src,
                                            uint32_t count)
{
    constexpr uint32_t batch = 4;
    for (uint32_t first = lane; first < count; first += lanes * batch) {
        uint32_t v[batch];
#pragma unroll
        for (uint32_t u = 0; u < batch; u++) {
            const uint32_t i = first + u * lanes;
            v[u] = mwhip::loadGlobal(src + (i < count ? i : first));
        }
#pragma unroll
        for (uint32_t u = 0; u < batch; u++) {
            const uint32_t i = first + u * lanes;
            if (i < count) {
                dst[i] = v[u];
            }
        }
    }
}

// Stages primitives [0, num_prims) of the object manager in LDS, with the
// `lanes` lanes that are in the call together (`lane` = this one's index among
// them: a world's group, or the whole wavefront when both of its worlds use
// the same manager).  Returns an ObjectManager whose primitive arrays point at
// the copies (or the original when they do not fit).
// image_only: only from the loader's ready-made image (what the copy holds then
// does not depend on num_prims: callers that share the block may each fill it).
__device__ inline ObjectManager stagePrimitives(uint32_t lane, uint32_t lanes,
                                                PrimBlock *pb,
                                                const ObjectManager &obj_mgr,
                                                uint32_t num_prims,
                                                bool image_only)
{
    if (num_prims > (uint32_t)PrimBlock::maxPrims) {
        return obj_mgr;
    }

    // the loader's ready-made image covers these primitives: three coalesced
    // copies and a pointer fix-up instead of the walk below
    const PrimImage *image = obj_mgr.primImage;
    if (image != nullptr) {
        const uint32_t image_prims = mwhip::loadGlobal(&image->numPrims);
        const uint32_t arena_used = mwhip::loadGlobal(&image->arenaUsed);
        if (image_prims >= num_prims && image_prims != 0) {
            // (the mesh offsets with the first batch of the copies)
            int32_t offsets[4] = { -1, -1, -1, -1 };
            if (lane < image_prims) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    offsets[i] = mwhip::loadGlobal(&image->meshOffset[lane][i]);
                }
            }
            waveCopyDwordsGlobal(lane, lanes, (uint32_t *)pb->prims,
                (const uint32_t *)image->prims,
                image_prims * (uint32_t)(sizeof(CollisionPrimitive) / 4));
            waveCopyDwordsGlobal(lane, lanes, (uint32_t *)pb->primAABBs,
                (const uint32_t *)image->primAABBs,
                image_prims * (uint32_t)(sizeof(math::AABB) / 4));
            waveCopyDwordsGlobal(lane, lanes, pb->arena, image->arena, arena_used);
            wave::phaseFence();
            if (lane < image_prims && offsets[0] >= 0) {
                geo::HalfEdgeMesh &staged = pb->prims[lane].hull.halfEdgeMesh;
                staged.facePlanes = (geo::Plane *)(pb->arena + offsets[0]);
                staged.halfEdges = (geo::HalfEdge *)(pb->arena + offsets[1]);
                staged.vertices = (math::Vector3 *)(pb->arena + offsets[2]);
                staged.faceBaseHalfEdges = pb->arena + offsets[3];
            }
            wave::phaseFence();

            ObjectManager staged = obj_mgr;
            staged.collisionPrimitives = pb->prims;
            staged.primitiveAABBs = pb->primAABBs;
            return staged;
        }
    }

    if (image_only) {
        return obj_mgr;
    }

    waveCopyDwords(lane, lanes, (uint32_t *)pb->prims,
        (const uint32_t *)obj_mgr.collisionPrimitives,
        num_prims * (uint32_t)(sizeof(CollisionPrimitive) / 4));
    waveCopyDwords(lane, lanes, (uint32_t *)pb->primAABBs,
        (const uint32_t *)obj_mgr.primitiveAABBs,
        num_prims * (uint32_t)(sizeof(math::AABB) / 4));
    wave::phaseFence();
    if (lane < num_prims) {
        pb->primMeshKey[lane] =
            pb->prims[lane].type == CollisionPrimitive::Type::Hull ?
                (const void *)pb->prims[lane].hull.halfEdgeMesh.vertices :
                nullptr;
    }
    wave::phaseFence();

    // every lane walks the (short) primitive list; meshes shared by several
    // primitives are staged once
    uint32_t arena_used = 0;
    for (uint32_t p = 0; p < num_prims; p++) {
        if (pb->prims[p].type != CollisionPrimitive::Type::Hull) {
            continue;
        }

        // still the HBM pointers: a primitive is only patched in its own turn
        const geo::HalfEdgeMesh src = pb->prims[p].hull.halfEdgeMesh;

        bool shared = false;
        for (uint32_t q = 0; q < p; q++) {
            if (pb->primMeshKey[q] == (const void *)src.vertices) {
                if (lane == 0) {
                    pb->prims[p].hull.halfEdgeMesh =
                        pb->prims[q].hull.halfEdgeMesh;
                }
                shared = true;
                break;
            }
        }
        if (shared) {
            wave::phaseFence();
            continue;
        }

        const uint32_t hedge_dw = src.numHalfEdges * 3;
        const uint32_t base_dw = src.numFaces;
        const uint32_t plane_dw = src.numFaces * 4;
        const uint32_t vert_dw = src.numVertices * 3;
        const uint32_t need = hedge_dw + base_dw + plane_dw + vert_dw;
        if (arena_used + need > (uint32_t)PrimBlock::arenaDwords) {
            continue;       // stays in HBM
        }

        uint32_t *dst = pb->arena + arena_used;
        waveCopyDwords(lane, lanes, dst, (const uint32_t *)src.facePlanes, plane_dw);
        waveCopyDwords(lane, lanes, dst + plane_dw,
                       (const uint32_t *)src.halfEdges, hedge_dw);
        waveCopyDwords(lane, lanes, dst + plane_dw + hedge_dw,
                       (const uint32_t *)src.vertices, vert_dw);
        waveCopyDwords(lane, lanes, dst + plane_dw + hedge_dw + vert_dw,
                       src.faceBaseHalfEdges, base_dw);
        if (lane == 0) {
            geo::HalfEdgeMesh &staged = pb->prims[p].hull.halfEdgeMesh;
            staged.facePlanes = (geo::Plane *)dst;
            staged.halfEdges = (geo::HalfEdge *)(dst + plane_dw);
            staged.vertices = (math::Vector3 *)(dst + plane_dw + hedge_dw);
            staged.faceBaseHalfEdges = dst + plane_dw + hedge_dw + vert_dw;
        }
        arena_used += need;
        wave::phaseFence();
    }
    wave::phaseFence();

    ObjectManager staged = obj_mgr;
    staged.collisionPrimitives = pb->prims;
    staged.primitiveAABBs = pb->primAABBs;
    return staged;
}

template <int MAXB, int LPW = 64>
struct LdsBodyStore {
    using Block = WorldBlock<MAXB, LPW>;
    Block *w;

    __device__ inline math::Vector3 &position(Loc l) { return w->poses[l.row].x; }
    __device__ inline math::Quat &rotation(Loc l) { return w->poses[l.row].q; }
    __device__ inline Velocity &velocity(Loc l) { return w->vel[l.row]; }
    // (an inert static body: see WorldBlock::poses.  No branches: a contact's
    // two bodies are of different kinds lane by lane)
    __device__ inline xpbd::SubstepPrevState prevState(Loc l)
    {
        const uint32_t slot = w->solverSlot[l.row];
        const auto p = w->poses[slot == Block::noSlot ? (uint32_t)l.row :
                                (uint32_t)Block::prevBase + slot];
        return xpbd::SubstepPrevState { p.x, p.q };
    }
    __device__ inline xpbd::PreSolvePositional presolvePositional(Loc l)
    {
        const uint32_t slot = w->solverSlot[l.row];
        const auto p = w->poses[slot == Block::noSlot ? (uint32_t)l.row :
                                (uint32_t)Block::prePosBase + slot];
        return xpbd::PreSolvePositional { p.x, p.q };
    }
    __device__ inline xpbd::PreSolveVelocity presolveVelocity(Loc l)
    {
        const uint32_t slot = w->solverSlot[l.row];
        const bool inert = slot == Block::noSlot;
        xpbd::PreSolveVelocity v = w->preVel[inert ? 0u : slot];
        v.v.x = inert ? 0.f : v.v.x;
        v.v.y = inert ? 0.f : v.v.y;
        v.v.z = inert ? 0.f : v.v.z;
        v.omega.x = inert ? 0.f : v.omega.x;
        v.omega.y = inert ? 0.f : v.omega.y;
        v.omega.z = inert ? 0.f : v.omega.z;
        return v;
    }
    __device__ inline xpbd::BodyConstants constants(Loc l)
    {
        return w->constants[l.row];
    }
};

// Key of body k for the dependency levels of the solver: 0 = a static body the
// solver cannot change (never orders constraints), else k + 1.  Whether a
// static body is inert is decided ONCE per step, when the world is loaded
// (solverKey): an inert one stays inert (the solver's writes to it are no-ops),
// and one that is not is treated as dynamic for the whole step -- conservative,
// so still the sequential result -- instead of normalising its rotation again
// for every contact of every solve (round 3: twice per contact and substep).
template <int MAXB, int LPW>
__device__ inline uint32_t ldsBodyKey(const WorldBlock<MAXB, LPW> *w, int32_t k)
{
    return (uint32_t)w->solverKey[k];
}

template <int MAXB, int LPW>
__device__ inline PairSetup ldsSetupPair(const WorldBlock<MAXB, LPW> *w,
                                         const ObjectManager &obj_mgr,
                                         const WaveCandidate &candidate)
{
    const int32_t ka = candidate.a;
    const int32_t kb = candidate.b;

    return setupPair(obj_mgr, Loc { 0, ka }, Loc { 0, kb },
        (uint32_t)w->primOffset[ka] + candidate.aPrim,
        (uint32_t)w->primOffset[kb] + candidate.bPrim,
        PrimitiveTransform { w->poses[ka].x, w->poses[ka].q, w->scale[ka] },
        PrimitiveTransform { w->poses[kb].x, w->poses[kb].q, w->scale[kb] });
}

// ---------------------------------------------------------------------------
// Loading a world through the frame (PhysicsFrame, physics.inl): five rounds
// of loads (rounds 1-3 walked the tables from every wavefront: twenty).
// Every round is issued in one go -- nothing is written, and nothing loaded is
// looked at, before the loads of the round are on their way:
//   1  (caller) the world's place in the order; the frame itself is hot
//   2  row ranges in every rigid-body table, the tree's arrays, the object
//      manager, the joint range, the solver parameters
//   3  (frame: L2 hits) the columns of this lane's body; the traversal order
//   4  the body's row in every column; the world's joints
//   5  what the row names: object metadata, primitive range, leaf box, leaf
//      parent; the entity slots of the joints' bodies
//   6  the leaf's slot box in its parent node
// ---------------------------------------------------------------------------
struct FramedWorld {
    // numBodies < 0: not stepped by this kernel -- unsteppable: tables await
    // their sort or the tree does not match them (kErrPhysics); tooManyBodies:
    // more than the instantiation's MAXB (the HBM kernel takes the world)
    static constexpr int32_t unsteppable = -1;
    static constexpr int32_t tooManyBodies = -2;    // (or joints)
    int32_t numBodies;
    int32_t jointBegin;
    int32_t numJoints;
    const ObjectManager *objMgr;    // the world's manager (HBM)
};

__device__ inline void fillPhysicsFrame(EcsState *S, const PhysicsScratch *ps,
                                        PhysicsFrame *F, uint32_t tid,
                                        uint32_t num_threads)
{
    StateManager *state_mgr = static_cast<StateManager *>(S);
    const uint32_t num_arch = ps->numBodyArchetypes;
    for (uint32_t i = tid; i < num_arch * PhysicsFrame::numColumns;
         i += num_threads) {
        const uint32_t a = i / PhysicsFrame::numColumns;
        const uint32_t c = i % PhysicsFrame::numColumns;
        const TableHdr &tbl = S->tables[ps->bodyArchetypes[a]];
        F->columns[a][c] = (int32_t)c < tbl.numColumns ? tbl.columns[c] : nullptr;
    }
    if (tid < PhysicsFrame::maxArchetypes) {
        // (entries behind the last archetype repeat the first: the step kernel
        // reads all of them without a branch and ignores what it got)
        const uint32_t id = ps->bodyArchetypes[tid < num_arch ? tid : 0u];
        const TableHdr &tbl = S->tables[id];
        F->archetype[tid] = id;
        F->worldOffsets[tid] = tbl.worldOffsets;
        F->worldCounts[tid] = tbl.worldCounts;
    }
    // what a replay can change: whether a table awaits its sort, and where the
    // joint rows are (the sort in front of this kernel swaps its buffers)
    if (tid == 1) {
        uint32_t unsorted = 0;
        for (uint32_t a = 0; a < num_arch; a++) {
            unsorted |= S->tables[ps->bodyArchetypes[a]].needsSort;
        }
        F->unsorted = unsorted;
        F->numArchetypes = num_arch;
    }
    // (the frame's column slots cover the body archetypes' Entity, WorldID,
    // RigidBody bundle and XPBD solver state, and the joint table's constraint
    // is its first component column)
    static_assert(PhysicsFrame::numColumns ==
                  (uint32_t)xpbd::XPBDCols::PreSolveVelocity + 1u);
    static_assert(RGDCols::JointConstraint == 2);
    if (tid == 2) {
        F->joints = (const JointConstraint *)
            S->tables[ps->jointArchetype].columns[RGDCols::JointConstraint];
    }
    // the singleton columns, the tables' world ranges, the entity store, the
    // object manager: a chain of six dependent loads by one thread, behind the
    // cost scan of the other wavefronts.  Followed on EVERY launch (round 4
    // kept the first replay's answers: stale if a simulator ever swaps its
    // object manager's arrays or the store moves).
    if (tid == 0) {
        F->systemStates = state_mgr->getSingletonColumn<PhysicsSystemState>();
        F->objectData = state_mgr->getSingletonColumn<ObjectData>();
        const ObjectManager *mgr = F->objectData[0].mgr;
        F->objMgr = mgr;
        F->objMgrCopy = *mgr;
        const TableHdr &joint_tbl = S->tables[ps->jointArchetype];
        F->jointOffsets = joint_tbl.worldOffsets;
        F->jointCounts = joint_tbl.worldCounts;
        F->entities = mwhip::entitiesOf(S);
        F->trees = state_mgr->getSingletonColumn<broadphase::BVH>();
    }
}

// nothing moves across: what was issued before stays before, what uses it after
__device__ __attribute__((always_inline)) inline void roundIssued()
{
    __builtin_amdgcn_sched_barrier(0);
}

template <int MAXB, int LPW>
__device__ __attribute__((always_inline)) inline FramedWorld loadWorldFramed(
    uint32_t lane, WorldBlock<MAXB, LPW> *w, const PhysicsFrame *F, int32_t world,
    LaneBodies<(MAXB + LPW - 1) / LPW> &mine)
{
    using Block = WorldBlock<MAXB, LPW>;
    using mwhip::loadGlobal;
    using mwhip::loadInvariant;
    constexpr uint32_t max_arch = PhysicsFrame::maxArchetypes;
    constexpr int32_t chunks = (MAXB + LPW - 1) / LPW;
    FramedWorld out {};

    // ---- round 2 (the frame's own words: scalar loads, hot) ---------------------
    const uint32_t num_arch = loadInvariant(&F->numArchetypes);
    const uint32_t tables_unsorted = loadInvariant(&F->unsorted);
    // (roundIssued(): the loads of a round are all on their way before the first
    // of them is looked at.  Left to itself the scheduler -- the kernel is at its
    // register limit -- pairs every load with its use: eight row counts became
    // eight round trips one after the other.)
    int32_t row_base[max_arch];
    int32_t rows[max_arch];
#pragma unroll
    for (uint32_t a = 0; a < max_arch; a++) {
        row_base[a] = loadGlobal(loadInvariant(&F->worldOffsets[a]) + world);
        rows[a] = loadGlobal(loadInvariant(&F->worldCounts[a]) + world);
    }
    const broadphase::BVH::StepView tree =
        broadphase::BVH::loadStepView(loadInvariant(&F->trees) + world);
    const ObjectManager *world_mgr =
        loadGlobal(&(loadInvariant(&F->objectData) + world)->mgr);
    const int32_t joint_begin = loadGlobal(loadInvariant(&F->jointOffsets) + world);
    const int32_t num_joints = loadGlobal(loadInvariant(&F->jointCounts) + world);
    const PhysicsSystemState sys_regs =
        loadGlobal(loadInvariant(&F->systemStates) + world);
    roundIssued();

    int32_t body_base[max_arch + 1];
    body_base[0] = 0;
#pragma unroll
    for (uint32_t a = 0; a < max_arch; a++) {
        rows[a] = a < num_arch ? rows[a] : 0;
        body_base[a + 1] = body_base[a] + rows[a];
    }
    const int32_t num_bodies = body_base[max_arch];
    out.numBodies = num_bodies;
    out.jointBegin = joint_begin;
    out.numJoints = num_joints;
    out.objMgr = world_mgr;
    if (tables_unsorted != 0u || tree.numLeaves != num_bodies) {
        out.numBodies = FramedWorld::unsteppable;
        return out;
    }
    if (num_bodies > MAXB || num_joints > Block::maxJointBodies) {
        out.numBodies = FramedWorld::tooManyBodies;
        return out;
    }
    if (lane == 0) {
        w->sys = sys_regs;
    }
    if (num_bodies == 0) {
        wave::phaseFence();
        return out;
    }
    // (the manager's arrays: the frame's copy unless this world has its own)
    const RigidBodyMetadata *metadata = loadInvariant(&F->objMgrCopy.metadata);
    const uint32_t *prim_offsets =
        loadInvariant(&F->objMgrCopy.rigidBodyPrimitiveOffsets);
    const uint32_t *prim_counts =
        loadInvariant(&F->objMgrCopy.rigidBodyPrimitiveCounts);
    if (world_mgr != loadInvariant(&F->objMgr)) {
        metadata = loadGlobal(&world_mgr->metadata);
        prim_offsets = loadGlobal(&world_mgr->rigidBodyPrimitiveOffsets);
        prim_counts = loadGlobal(&world_mgr->rigidBodyPrimitiveCounts);
    }

    // (the world's joints, when they fit the block: their rows go out first,
    // the entity slots of their bodies with round 5 of the first chunk)
    const bool my_joint = (int32_t)lane < num_joints;
    JointConstraint joint {};
    mwhip::EntitySlot joint_slots[2] {};
    if (my_joint) {
        joint = loadGlobal(loadInvariant(&F->joints) + joint_begin + (int32_t)lane);
    }

    // (the traversal order of every chunk first: a body's rank in it is written
    // by whichever lane holds that position of the order)
    int32_t order_leaf[chunks];
#pragma unroll
    for (int32_t c = 0; c < chunks; c++) {
        const int32_t k = c * LPW + (int32_t)lane;
        order_leaf[c] = loadGlobal(tree.traversalOrder + (k < num_bodies ? k : 0));
    }

    // Lanes without a body load the world's first body again (no branches
    // around the loads: a round stays one batch) and write nothing.
#pragma unroll
    for (int32_t c = 0; c < chunks; c++) {
        const int32_t k = c * LPW + (int32_t)lane;
        const bool active = k < num_bodies;
        const int32_t kc = active ? k : 0;

        // ---- round 3: where the body's rows are (frame: L2 hits) ------------------
        uint32_t slot = 0;
        int32_t row = row_base[0] + kc;
#pragma unroll
        for (uint32_t a = 1; a < max_arch; a++) {
            if (kc >= body_base[a] && rows[a] != 0) {
                slot = a;
                row = row_base[a] + (kc - body_base[a]);
            }
        }
        void *const *col = F->columns[slot];
        const uint32_t archetype = loadGlobal(&F->archetype[slot]);
        const Entity *col_entity = (const Entity *)loadGlobal(&col[0]);
        const base::Position *col_pos =
            (const base::Position *)loadGlobal(&col[RGDCols::Position]);
        const base::Rotation *col_rot =
            (const base::Rotation *)loadGlobal(&col[RGDCols::Rotation]);
        const base::Scale *col_scale =
            (const base::Scale *)loadGlobal(&col[RGDCols::Scale]);
        const base::ObjectID *col_obj =
            (const base::ObjectID *)loadGlobal(&col[RGDCols::ObjectID]);
        const ResponseType *col_resp =
            (const ResponseType *)loadGlobal(&col[RGDCols::ResponseType]);
        const broadphase::LeafID *col_leaf =
            (const broadphase::LeafID *)loadGlobal(&col[RGDCols::LeafID]);
        const Velocity *col_vel =
            (const Velocity *)loadGlobal(&col[RGDCols::Velocity]);
        const ExternalForce *col_force =
            (const ExternalForce *)loadGlobal(&col[RGDCols::ExternalForce]);
        const ExternalTorque *col_torque =
            (const ExternalTorque *)loadGlobal(&col[RGDCols::ExternalTorque]);
        roundIssued();

        // ---- round 4: the rows ---------------------------------------------------------
        const base::Position pos = loadGlobal(col_pos + row);
        const base::Rotation rot = loadGlobal(col_rot + row);
        const base::Scale scale = loadGlobal(col_scale + row);
        const Velocity vel = loadGlobal(col_vel + row);
        const ExternalForce force = loadGlobal(col_force + row);
        const ExternalTorque torque = loadGlobal(col_torque + row);
        const ResponseType resp = loadGlobal(col_resp + row);
        const Entity entity = loadGlobal(col_entity + row);
        const base::ObjectID obj_id = loadGlobal(col_obj + row);
        const int32_t leaf = loadGlobal(col_leaf + row).id;
        roundIssued();

        // ---- round 5: what the row names ----------------------------------------------
        const RigidBodyMetadata body_metadata = loadGlobal(metadata + obj_id.idx);
        const uint32_t prim_offset = loadGlobal(prim_offsets + obj_id.idx);
        const uint32_t prim_count = loadGlobal(prim_counts + obj_id.idx);
        const math::AABB query_box = loadGlobal(tree.leafAABBs + leaf);
        const uint32_t parent = loadGlobal(tree.leafParents + leaf);
        if (c == 0 && my_joint) {
            // (StateManager::getLoc reads the slot of any non-negative id)
            const mwhip::EntitySlot *entities = loadInvariant(&F->entities);
            joint_slots[0] = loadGlobal(entities + (joint.e1.id >= 0 ? joint.e1.id : 0));
            joint_slots[1] = loadGlobal(entities + (joint.e2.id >= 0 ? joint.e2.id : 0));
        }
        roundIssued();

        // ---- round 6 -------------------------------------------------------------------
        const math::AABB slot_box =
            broadphase::BVH::loadSlotBounds(tree.nodes, parent);
        roundIssued();

        // ---- into the block (writeBodyRow) ---------------------------------------------
        if (c == 0) {
#pragma unroll
            for (int32_t r = 0; r < chunks; r++) {
                if (r * LPW + (int32_t)lane < num_bodies) {
                    w->leafRank()[order_leaf[r]] = (uint16_t)(r * LPW + (int32_t)lane);
                }
            }
            wave::phaseFence();
        }
        mine.loc[c] = Loc { archetype, active ? row : -1 };
        mine.extForce[c] = force;
        mine.extTorque[c] = torque;
        mine.entityID[c] = entity.id;
        if (active) {
            const uint32_t rank = w->leafRank()[leaf];
            w->poses[k].x = pos;
            w->poses[k].q = rot;
            w->scale[k] = scale;
            w->vel[k] = vel;
            w->resp[k] = (uint8_t)resp;
            w->constants[k] = xpbd::bodyConstants(body_metadata, resp);
            w->primOffset[k] = (uint16_t)prim_offset;
            w->primCount[k] = (uint8_t)prim_count;
            w->queryBox()[k] = query_box;
            w->rankSlotBox()[rank] = slot_box;
            w->rankEntityInfo()[rank] = int2 { entity.id,
                Block::packRankInfo(k, prim_count,
                                    resp == ResponseType::Static) };
        }
    }
    // (the candidate list -- leafRank's storage -- is written next: the ranks
    // have been read by now)
    wave::phaseFence();

    // ---- the world's joints: which bodies they join ------------------------------
    if (num_joints > 0) {
        // (StateManager::getLoc: a stale or empty handle is nowhere)
        Loc end_loc[2] = { Loc::none(), Loc::none() };
        if (my_joint) {
            const Entity ends[2] = { joint.e1, joint.e2 };
#pragma unroll
            for (int32_t e = 0; e < 2; e++) {
                if (ends[e].id >= 0 && joint_slots[e].gen == ends[e].gen) {
                    end_loc[e] = Loc { joint_slots[e].loc.archetype,
                                       joint_slots[e].loc.row };
                }
            }
        }
        // body index of an end point = the lane (and chunk) whose row it is:
        // joint j's lane asks, every lane answers for its bodies (no match:
        // body 0, like the table walk of rounds 1-3)
        uint32_t index[2] = { 0u, 0u };
        for (int32_t j = 0; j < num_joints; j++) {
#pragma unroll
            for (int32_t e = 0; e < 2; e++) {
                const uint32_t want_arch = __shfl(end_loc[e].archetype, j, LPW);
                const int32_t want_row = __shfl(end_loc[e].row, j, LPW);
                uint32_t found = 0;
                bool any = false;
#pragma unroll
                for (int32_t c = chunks - 1; c >= 0; c--) {
                    const uint64_t match = wave::groupBallot<LPW>(
                        mine.loc[c].row >= 0 && mine.loc[c].archetype == want_arch &&
                        mine.loc[c].row == want_row);
                    if (match != 0) {
                        found = (uint32_t)(c * LPW) +
                            (uint32_t)__builtin_ctzll(match);
                        any = true;
                    }
                }
                if ((int32_t)lane == j && any) {
                    index[e] = found;
                }
            }
        }
        if (my_joint) {
            if ((int32_t)lane < Block::maxJoints) {
                w->joints[lane] = joint;
            }
            w->jointBodies[lane][0] = (uint8_t)index[0];
            w->jointBodies[lane][1] = (uint8_t)index[1];
        }
        wave::phaseFence();
    }
    return out;
}

// Wavefronts per SIMD.  The step is bound by instruction issue and by the
// latency of its LDS / HBM round trips (SQ counters, profiles/r06_phys_occupancy_
// counters.jsonl: one wavefront per SIMD issues 59 % of its cycles and waits
// 38 %), so what it wants is a second instruction stream per SIMD: 256
// registers per wavefront and 20 KB of LDS per workgroup.
//
// LPW = lanes per world.  64: one world per wavefront (MAXB <= 64: 19-26 KB
// blocks, two wavefronts per SIMD).  32: TWO worlds per wavefront, one per half
// (MAXB <= 32): a 28-body world with ~14 contacts and ~21 candidates keeps 45 %
// of 64 lanes busy -- the same instruction stream then advances two worlds.
// Every wave-level primitive works on the world's group of LPW lanes
// (wave::groupBallot, shuffles of width LPW); loop trip counts, early exits and
// the level loops of the solver are per group, the halves diverge where their
// worlds differ.  Rounds 3-5 ran this layout at ONE wavefront per SIMD with the
// whole register file (two 19 KB blocks = 37.9 KB per workgroup, four per CU);
// since round 6 a pair of blocks and the workgroup's primitive image are 20 256
// B -- eight workgroups per CU, two per SIMD (profiles/r06_diet*_variants.jsonl:
// 680 -> 575 us at 8192 Escape-Room worlds, 301 dwords spilled under the
// 256-register cap).  -DMADRONA_PHYS_LDS32_WAVES_PER_EU=1 keeps one (394
// registers) for measurements.
#ifndef MADRONA_PHYS_LDS_WAVES_PER_EU
#define MADRONA_PHYS_LDS_WAVES_PER_EU 2
#endif
#ifndef MADRONA_PHYS_LDS32_WAVES_PER_EU
#define MADRONA_PHYS_LDS32_WAVES_PER_EU 2
#endif
#ifdef MADRONA_PHYS_LDS_NUM_VGPR
// (measurement builds: a register cap without the occupancy to go with it --
// the compiler ignores amdgpu_waves_per_eu above what the LDS block admits)
#define MADRONA_PHYS_VGPR_CAP __attribute__((amdgpu_num_vgpr(MADRONA_PHYS_LDS_NUM_VGPR)))
#else
#define MADRONA_PHYS_VGPR_CAP
#endif
template <int MAXB, int LPW = 64>
__global__ void __launch_bounds__(64) MADRONA_PHYS_VGPR_CAP
__attribute__((amdgpu_waves_per_eu(
    MAXB <= 64 && LPW == 64 ? MADRONA_PHYS_LDS_WAVES_PER_EU :
    LPW == 32 ? MADRONA_PHYS_LDS32_WAVES_PER_EU : 1)))
physicsStepLdsKernel(EcsState *S, void *node_data, uint32_t, uint32_t)
{
    mwhip::TraceScope trace_scope(S);
    using Block = WorldBlock<MAXB, LPW>;
    static_assert(LPW == 64 || (LPW == 32 && MAXB <= 32));
    constexpr int worlds_per_wave = 64 / LPW;
    constexpr int32_t chunks = (MAXB + LPW - 1) / LPW;  // bodies per lane

    PhysicsScratch *ps = detail::scratch(S);
    const PhysicsStepParams params = *(const PhysicsStepParams *)node_data;

    // lane = index inside the world's group of LPW lanes
    const uint32_t lane = wave::laneID() & (uint32_t)(LPW - 1);
    const int32_t group = (int32_t)(wave::laneID() / (uint32_t)LPW);
    const int32_t num_worlds = S->numWorlds;

    __shared__ Block blocks[worlds_per_wave];
    __shared__ PrimBlock prim_block;
    Block *w = &blocks[group];
    LdsBodyStore<MAXB, LPW> store { w };

#ifdef MADRONA_PHYS_PROFILE_LDS_HH
#define MADRONA_PHYS_PROFILE_LDS 1
#endif
#ifdef MADRONA_PHYS_PROFILE_LDS
    // (the phase profile of a build that keeps its registers: the accumulators
    // sit in what the block leaves of the workgroup's LDS, one ds_add per mark)
    __shared__ uint32_t prof_lds[worlds_per_wave][16];
    if (lane < 16u) prof_lds[group][lane] = 0u;
    unsigned long long prof_t = __builtin_readcyclecounter();
#undef PHYS_PROF
#ifdef MADRONA_PHYS_PROFILE_LDS_HH
    // (the stages of the hull-hull tests instead of the phases: same slots)
#define PHYS_PROF(slot) do { (void)prof_t; } while (0)
#undef PHYS_HH_PROF
#define PHYS_HH_PROF() HullHullProf { &prof_lds[group][0], \
        (unsigned long long)__builtin_readcyclecounter() }
#else
#define PHYS_PROF(slot) do { unsigned long long now_ = __builtin_readcyclecounter(); \
        if (lane == 0u) atomicAdd(&prof_lds[group][slot], (uint32_t)(now_ - prof_t)); \
        prof_t = now_; } while (0)
#endif
#endif
#ifdef MADRONA_PHYS_PROFILE
    unsigned long long prof_t = __builtin_readcyclecounter();
    unsigned long long prof_acc[32] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                        0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    // stages of the cooperative hull-hull tests: slots 16.. (HullHullProf)
#ifdef MADRONA_PHYS_PROFILE_HH
#define PHYS_HH_PROF() HullHullProf { prof_acc, \
        (unsigned long long)__builtin_readcyclecounter() }
#else
#define PHYS_HH_PROF() HullHullProf {}
#endif
#elif !defined(MADRONA_PHYS_PROFILE_LDS_HH)
#define PHYS_HH_PROF() HullHullProf {}
#endif

    // Worlds are taken in the order physicsOrderKernel left: the ones that took
    // longest last step first, so that the launch does not end on a few heavy
    // worlds with most of the chip idle; the two worlds of a wavefront (LPW =
    // 32) are neighbours in that order and cost about the same.  One workgroup
    // per job (a world, or a pair of them): the hardware's dispatcher hands the
    // next job to a free slot within a microsecond.  (Measured and removed in
    // round 5, numbers in profiles/r04_phys_variants.jsonl and DESIGN.md: the
    // k-th heaviest paired with the k-th lightest, + 20 us; persistent
    // wavefronts taking jobs from a counter with the next world's header
    // fetched ahead, +- 0; world images packed by a kernel of their own,
    // + 21 us net; the leaf refit folded into the epilogue, + 6 us net; the
    // order blended over several steps, +- 0.  Round 6, profiles/r06_cost*_
    // variants.jsonl, r06_phys_wave_log.txt: the order from hull pairs and
    // contacts alone instead of the clock, +- 0; a cost that decays instead of
    // being replaced, - 1 % of the kernel, the step unchanged; with the times
    // the wavefronts took known beforehand the kernel would end 50-60 us
    // earlier -- last step's cost predicts this step's with r = 0.64; the k-th
    // heaviest with the k-th lightest again, now that the halves share their
    // hull-hull pairs out: + 12-15 us, r06_pairing_variants.jsonl.)
    const int32_t *world_order = params.worldOrder;
    const int32_t num_jobs = (num_worlds + worlds_per_wave - 1) / worlds_per_wave;

    // (the scratch block's words never change: invariant loads at the top of
    // the kernel, not two dependent round trips in front of the candidate pass)
    const uint32_t candidates_per_world =
        mwhip::loadInvariant(&ps->candidatesPerWorld);
    CandidateCollision *const world_candidates =
        mwhip::loadInvariant(&ps->worldCandidates);

    // the frame: where every world's rows are, resolved once per launch by
    // physicsOrderKernel (fillPhysicsFrame)
    const PhysicsFrame *frame = &((const PhysicsStepNode *)node_data)->frame;

    // A world this instantiation cannot hold -- more than MAXB bodies, more
    // contacts in a substep than the block has room for -- is left untouched
    // (nothing of it has been stored yet) and handed to the kernel that works
    // out of HBM, which runs right behind this one over the worlds listed here
    // (physicsStepKernel, fallback mode; reference: no cap, its tables grow).
    int32_t *const fallback_list = params.fallbackList;
    const uint32_t contact_cap = params.contactCap != 0u &&
        params.contactCap < (uint32_t)Block::maxContacts ?
            params.contactCap : (uint32_t)Block::maxContacts;
    auto toFallback = [&](int32_t world) {
        if (lane == 0) {
            const int32_t at = __hip_atomic_fetch_add(fallback_list, 1,
                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            fallback_list[1 + at] = world;
        }
    };

    for (int32_t job = (int32_t)blockIdx.x; job < num_jobs;
         job += (int32_t)gridDim.x) {
    // (raising the priority of the wavefronts at the head of the order -- the
    // heaviest worlds -- over the wavefront they share a SIMD with moves nothing:
    // 522-529 us for the heaviest half / quarter / eighth against 522-527,
    // profiles/r06_prio_variants.jsonl)
    do {
        const int32_t order_slot = job * worlds_per_wave + group;
        if (order_slot >= num_worlds) {
            continue;       // (odd world count: this half has no world)
        }
        const int32_t world = world_order[order_slot];
        const long long cost_t0 = (long long)wall_clock64();
        uint32_t cost_work = 1;    // (what the world asked of the wavefront, roughly)
#ifdef MADRONA_PHYS_WAVELOG
        // (measurement builds, profiles/tools/phys_wave_log.py: what the order
        // kernel believed of this world against what it took)
        const uint32_t wavelog_predicted = params.worldCost[world];
        uint32_t wavelog_hull_pairs = 0, wavelog_contacts = 0;
        uint32_t wavelog_iters_max = 0, wavelog_iters_shared = 0;
#endif

        // ---- the world through the frame: HBM -> LDS in five rounds ---------
        LaneBodies<chunks> mine;
        const FramedWorld framed =
            loadWorldFramed<MAXB, LPW>(lane, w, frame, world, mine);
        if (framed.numBodies < 0) {
            if (framed.numBodies == FramedWorld::tooManyBodies) {
                toFallback(world);
            } else {
                mwhip::raiseError(S, mwhip::kErrPhysics);
            }
            continue;
        }
        const int32_t num_bodies = framed.numBodies;
        const ObjectManager *hbm_mgr = framed.objMgr == frame->objMgr ?
            &frame->objMgrCopy : framed.objMgr;
        PHYS_PROF(8);
        const ObjectManager &hbm_obj_mgr = *hbm_mgr;

        // primitives referenced by this world's bodies
        uint32_t prim_end = 0;
        // ... and the bodies the solver can change, with their slots
        uint32_t solver_bodies = 0;
#pragma unroll
        for (int32_t c = 0; c < chunks; c++) {
            const int32_t k = c * LPW + (int32_t)lane;
            bool solved = false;
            if (k < num_bodies) {
                uint32_t end = (uint32_t)w->primOffset[k] + w->primCount[k];
                prim_end = end > prim_end ? end : prim_end;
                solved = !((uint32_t)w->resp[k] == (uint32_t)ResponseType::Static &&
                           staticBodyIsInert(w->poses[k].q));
                w->solverKey[k] = solved ? (uint8_t)(k + 1) : (uint8_t)0;
            }
            const uint64_t solved_mask = wave::groupBallot<LPW>(solved);
            if (k < num_bodies) {
                w->solverSlot[k] = solved ? (uint8_t)(solver_bodies +
                    wave::rankInGroup(solved_mask, lane)) : (uint8_t)Block::noSlot;
            }
            solver_bodies += (uint32_t)__builtin_popcountll(solved_mask);
        }
        if (solver_bodies > (uint32_t)Block::maxSolverBodies) {
            // (nothing of the world has been changed yet)
            toFallback(world);
            continue;
        }
        prim_end = wave::maxReduce<LPW>(prim_end);
        // One primitive image per workgroup.  Two worlds share it: it then only
        // ever holds the loader's image of the EXECUTOR's manager -- whichever
        // world fills it, and whenever, the bytes are the same --, a world with
        // a manager of its own (or a manager without an image) reads its
        // primitives from HBM.  Worlds that are here together copy with all
        // their lanes.
        ObjectManager obj_mgr = hbm_obj_mgr;
        if constexpr (worlds_per_wave == 1) {
            obj_mgr = stagePrimitives(lane, (uint32_t)LPW, &prim_block, hbm_obj_mgr,
                                      prim_end, false);
        } else if (framed.objMgr == frame->objMgr) {
            const uint64_t here = __builtin_amdgcn_ballot_w64(true);
            const bool both = (uint32_t)here != 0u && (uint32_t)(here >> 32) != 0u;
            obj_mgr = stagePrimitives(both ? wave::laneID() : lane,
                                      both ? 64u : (uint32_t)LPW, &prim_block,
                                      hbm_obj_mgr, prim_end, true);
        }
        PHYS_PROF(9);

        // ---- the other world of the wavefront (LPW = 32): a half whose world has
        // no hull-hull pair left takes one of the other world's (the narrowphase
        // below) -- only between worlds that read the same primitives ----------
        [[maybe_unused]] bool may_share = false;
        [[maybe_unused]] Block *const partner_w = &blocks[group ^ (worlds_per_wave - 1)];
        auto fromPartner64 = [](unsigned long long v) {
            return ((unsigned long long)(uint32_t)__shfl_xor((int32_t)(v >> 32), 32, 64)
                        << 32) |
                (unsigned long long)(uint32_t)__shfl_xor((int32_t)v, 32, 64);
        };
        if constexpr (worlds_per_wave == 2) {
            const uint64_t here = __builtin_amdgcn_ballot_w64(true);
            if ((uint32_t)here != 0u && (uint32_t)(here >> 32) != 0u) {
                may_share =
                    fromPartner64((unsigned long long)(uintptr_t)
                        obj_mgr.collisionPrimitives) ==
                        (unsigned long long)(uintptr_t)obj_mgr.collisionPrimitives &&
                    fromPartner64((unsigned long long)(uintptr_t)
                        obj_mgr.primitiveAABBs) ==
                        (unsigned long long)(uintptr_t)obj_mgr.primitiveAABBs;
            }
        }

        // ---- broadphase: candidate pairs in (body, traversal) order -----------
        // lane = body; one pass over the slot boxes in traversal order leaves a
        // bit mask of hits, the pairs are written from the mask.  Slot boxes
        // only grow between rebuilds, so a long-lived world with mobile bodies
        // collects many more candidates than contacts: what does not fit the
        // LDS list spills into the world's segment of the HBM candidate
        // scratch (4-byte records).
        WaveCandidate *spilled_candidates = (WaveCandidate *)(
            world_candidates + (size_t)world * candidates_per_world);
        const uint32_t candidate_capacity = (uint32_t)Block::maxCandidates +
            candidates_per_world *
                (uint32_t)(sizeof(CandidateCollision) / sizeof(WaveCandidate));
        uint32_t num_candidates = 0;
#pragma unroll
        for (int32_t cc = 0; cc < chunks; cc++) {
            const int32_t chunk = cc * LPW;
            if (chunk >= num_bodies) {
                break;
            }
            const int32_t k = chunk + (int32_t)lane;
            const bool active = k < num_bodies;

            constexpr int mask_words = (MAXB + 63) / 64;
            uint64_t hits[mask_words];
#pragma unroll
            for (int m = 0; m < mask_words; m++) {
                hits[m] = 0;
            }

            uint32_t n = 0;
            if (active) {
                const math::AABB query = w->queryBox()[k];
                const int32_t my_id = mine.entityID[cc];
                const bool my_static =
                    (uint32_t)w->resp[k] == (uint32_t)ResponseType::Static;
                const uint32_t a_prims = w->primCount[k];

#pragma unroll
                for (int m = 0; m < mask_words; m++) {
                    const int32_t r_end = num_bodies - m * 64 < 64 ?
                        num_bodies - m * 64 : 64;
                    // the box tests first, four at a time: the four slot boxes
                    // and entity ids are read out of LDS into registers before
                    // the first comparison (broadcast reads: every lane of the
                    // world asks for the same ones), and a test is twelve
                    // comparisons AND-ed as integers.  (AABB::overlaps is a
                    // chain of &&: the compiler made it three dependent LDS
                    // round trips and two branches per box -- the later
                    // coordinates were only read if the earlier ones overlapped.)
                    // ... and so is what used to be a second loop over the
                    // boxes that passed (orderBody -> resp / primCount: three
                    // dependent LDS round trips per hit, the lanes of the world
                    // waiting for the one with the most hits): a pair of two
                    // static bodies is no candidate, the others count their
                    // primitive pairs -- both from the word packed next to the
                    // entity id (rankEntityInfo).
                    uint64_t raw = 0;
                    for (int32_t j = 0; j < r_end; j += 4) {
                        float box[4][6];
                        int2 other[4];
#pragma unroll
                        for (int32_t u = 0; u < 4; u++) {
                            // (past the world's last body: its last one again)
                            const int32_t r = m * 64 +
                                (j + u < r_end ? j + u : r_end - 1);
                            const math::AABB slot = w->rankSlotBox()[r];
                            box[u][0] = slot.pMin.x;
                            box[u][1] = slot.pMin.y;
                            box[u][2] = slot.pMin.z;
                            box[u][3] = slot.pMax.x;
                            box[u][4] = slot.pMax.y;
                            box[u][5] = slot.pMax.z;
                            other[u] = w->rankEntityInfo()[r];
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int32_t u = 0; u < 4; u++) {
                            const uint32_t info = (uint32_t)other[u].y;
                            const uint32_t both_static =
                                (my_static ? 1u : 0u) & (info >> 16);
                            // == query.overlaps(slot) && my_id < other_id &&
                            //    !(both static)
                            const uint32_t hit =
                                (uint32_t)(query.pMin.x < box[u][3]) &
                                (uint32_t)(box[u][0] < query.pMax.x) &
                                (uint32_t)(query.pMin.y < box[u][4]) &
                                (uint32_t)(box[u][1] < query.pMax.y) &
                                (uint32_t)(query.pMin.z < box[u][5]) &
                                (uint32_t)(box[u][2] < query.pMax.z) &
                                (uint32_t)(my_id < other[u].x) &
                                (uint32_t)(j + u < r_end) &
                                (both_static ^ 1u);
                            raw |= (uint64_t)hit << (j + u);
                            n += hit * a_prims * ((info >> 8) & 0xFFu);
                        }
                    }
                    hits[m] = raw;
                }
            }

            uint32_t chunk_total;
            uint32_t out = num_candidates +
                wave::exclusiveScan<LPW>(n, lane, &chunk_total);

            if (active && n != 0 && out + n <= candidate_capacity) {
                const uint32_t a_prims = w->primCount[k];
#pragma unroll
                for (int m = 0; m < mask_words; m++) {
                    uint64_t pending = hits[m];
                    while (pending != 0) {
                        const int32_t r =
                            m * 64 + (int32_t)__builtin_ctzll(pending);
                        pending &= pending - 1;
                        const uint32_t info = (uint32_t)w->rankEntityInfo()[r].y;
                        const int32_t kb = (int32_t)(info & 0xFFu);
                        const uint32_t b_prims = (info >> 8) & 0xFFu;
                        const uint32_t total_checks = a_prims * b_prims;
                        uint32_t ca = 0, cb = 0;    // c / b_prims, c % b_prims
                        for (uint32_t c = 0; c < total_checks; c++) {
                            const WaveCandidate candidate {
                                (uint8_t)k, (uint8_t)kb, (uint8_t)ca, (uint8_t)cb,
                            };
                            cb++;
                            if (cb == b_prims) {
                                cb = 0;
                                ca++;
                            }
                            if (out < (uint32_t)Block::maxCandidates) {
                                w->candidates[out] = candidate;
                            } else {
                                spilled_candidates[
                                    out - (uint32_t)Block::maxCandidates] =
                                        candidate;
                            }
                            out++;
                        }
                    }
                }
            }
            num_candidates += chunk_total;
        }
        if (num_candidates > candidate_capacity) {
            mwhip::raiseError(S, mwhip::kErrTableOverflow);
            continue;
        }
        wave::phaseFence();
        PHYS_PROF(1);

        auto candidateAt = [&](uint32_t c) {
            return c < (uint32_t)Block::maxCandidates ? w->candidates[c] :
                spilled_candidates[c - (uint32_t)Block::maxCandidates];
        };

        // ---- the world's joints (table sorted by world just before) -----------
        const int32_t num_joints = framed.numJoints;
        const JointConstraint *joints =
            mwhip::loadInvariant(&frame->joints) + framed.jointBegin;

        PHYS_PROF(7);
        bool bailed = false;
        for (int32_t substep = 0; substep < params.numSubsteps; substep++) {
            // ---- integrate (xpbd.cpp substepRigidBodies) ------------------------
#pragma unroll
            for (int32_t c = 0; c < chunks; c++) {
                const int32_t k = c * LPW + (int32_t)lane;
                if (k >= num_bodies) {
                    continue;
                }
                if (w->solverKey[k] == 0) {
                    continue;       // (inert static body: no records)
                }
                const uint32_t slot = w->solverSlot[k];
                Vector3 x = w->poses[k].x;
                Quat q = w->poses[k].q;
                const uint32_t resp = (uint32_t)w->resp[k];

                w->poses[Block::prevBase + slot] = typename Block::Pose { x, q };

                if (resp == (uint32_t)ResponseType::Static) {
                    w->poses[Block::prePosBase + slot] = typename Block::Pose { x, q };
                    w->preVel[slot] = xpbd::PreSolveVelocity {
                        Vector3::zero(), Vector3::zero() };
                    continue;
                }

                xpbd::SubstepResult next = xpbd::integrateBody(
                    x, q, w->vel[k].linear, w->vel[k].angular,
                    w->constants[k].invMass, w->constants[k].invInertia,
                    mine.extForce[c], mine.extTorque[c], w->sys.g, w->sys.h,
                    resp == (uint32_t)ResponseType::Dynamic);

                w->poses[k].x = next.x;
                w->poses[k].q = next.q;
                w->poses[Block::prePosBase + slot] =
                    typename Block::Pose { next.x, next.q };
                w->preVel[slot] = xpbd::PreSolveVelocity { next.v, next.omega };
            }
            wave::phaseFence();
            PHYS_PROF(2);

            // ---- narrowphase: contacts in candidate order ------------------------
            // A lane's contact is staged in LDS at slot (num_contacts + lane)
            // and the chunk is compacted in place afterwards: nothing as wide
            // as a ContactConstraint stays live in registers across the
            // cooperative hull-hull tests.  A chunk is as wide as the free
            // slots allow.
            uint32_t num_contacts = 0;
            bool contacts_overflow = false;
            for (uint32_t chunk = 0; chunk < num_candidates; ) {
                const uint32_t free_slots = contact_cap - num_contacts;
                if (free_slots == 0) {
                    contacts_overflow = true;
                    break;
                }
                uint32_t width = num_candidates - chunk;
                width = width < (uint32_t)LPW ? width : (uint32_t)LPW;
                width = width < free_slots ? width : free_slots;

                PackedContact *stage = w->contacts() + num_contacts;
                bool has_contact = false;
                bool too_big = false;
                bool unsupported = false;

                uint32_t kind = 0;      // 1: this lane alone, 2: whole wave
                {
                    PairSetup pair;
                    if (lane < width) {
                        pair = ldsSetupPair(w, obj_mgr,
                                            candidateAt(chunk + lane));
                        if (pair.aabbOverlap) {
                            kind = pair.test == NarrowphaseTest::HullHull ?
                                2 : 1;
                        }
                    }
                    PHYS_PROF(0);

                    // hull against plane: two lanes a pair (hullPlaneContactTeam),
                    // the pairs in lane order over the world's teams.  The team
                    // reads the pair out of the registers of the lane that set it
                    // up and hands the outcome back to it.
                    if constexpr (LPW == 32) {
                        const bool plane_pair = kind == 1 &&
                            pair.test == NarrowphaseTest::HullPlane;
                        const uint64_t plane_pairs =
                            wave::groupBallot<LPW>(plane_pair);
                        const uint32_t num_plane_pairs =
                            (uint32_t)__builtin_popcountll(plane_pairs);
                        const uint32_t my_rank = wave::rankInGroup(plane_pairs, lane);
                        if (plane_pair) {
                            kind = 3;       // (not one of the lanes on their own)
                        }
                        for (uint32_t first = 0; first < num_plane_pairs;
                             first += LPW / 2) {
                            const uint32_t team = first + (lane >> 1);
                            const bool in_team = team < num_plane_pairs;
                            const int src = (int)wave::nthSetBit<LPW>(plane_pairs,
                                in_team ? team : 0u);
                            auto from = [&](float v) { return __shfl(v, src, LPW); };
                            const PrimitiveTransform a_txfm {
                                { from(pair.a.pos.x), from(pair.a.pos.y),
                                  from(pair.a.pos.z) },
                                { from(pair.a.rot.w), from(pair.a.rot.x),
                                  from(pair.a.rot.y), from(pair.a.rot.z) },
                                { from(pair.a.scale.d0), from(pair.a.scale.d1),
                                  from(pair.a.scale.d2) },
                            };
                            PrimitiveTransform plane_txfm = a_txfm;
                            plane_txfm.pos = { from(pair.b.pos.x), from(pair.b.pos.y),
                                               from(pair.b.pos.z) };
                            plane_txfm.rot = { from(pair.b.rot.w), from(pair.b.rot.x),
                                               from(pair.b.rot.y), from(pair.b.rot.z) };
                            const unsigned long long prim_bits =
                                (unsigned long long)(uintptr_t)pair.aPrim;
                            const CollisionPrimitive *a_prim =
                                (const CollisionPrimitive *)(uintptr_t)(
                                    ((unsigned long long)(uint32_t)__shfl(
                                        (int32_t)(prim_bits >> 32), src, LPW) << 32) |
                                    (unsigned long long)(uint32_t)__shfl(
                                        (int32_t)prim_bits, src, LPW));
                            const int32_t a_row = __shfl(pair.aLoc.row, src, LPW);
                            const int32_t b_row = __shfl(pair.bLoc.row, src, LPW);

                            uint32_t outcome = 0;
                            if (in_team) {
                                LazyHull a(a_prim->hull.halfEdgeMesh, a_txfm.pos,
                                           a_txfm.rot, a_txfm.scale, false);
                                outcome = hullPlaneContactTeam<BlockScratch<LPW>::polyVerts>(
                                    (lane & 1u) != 0u, a, plane_txfm,
                                    Loc { 0, a_row }, Loc { 0, b_row }, stage + src);
                            }
                            // (the lane whose pair it is learns the outcome)
                            const uint32_t mine = (uint32_t)__shfl((int32_t)outcome,
                                (int)((my_rank - first) * 2u) & (LPW - 1), LPW);
                            if (plane_pair && my_rank >= first &&
                                    my_rank < first + LPW / 2) {
                                has_contact = (mine & 1u) != 0u;
                                too_big = (mine & 2u) != 0u;
                            }
                        }
                    }

                    // lanes on their own, in rounds of the block's scratch rows
                    using Scratch = BlockScratch<LPW>;
                    uint64_t solo = wave::groupBallot<LPW>(kind == 1);
                    const uint32_t solo_rank = wave::rankInGroup(solo, lane);
                    const uint32_t solo_count =
                        (uint32_t)__builtin_popcountll(solo);
                    for (uint32_t first = 0; first < solo_count;
                         first += Scratch::polyRows) {
                        if (kind == 1 && solo_rank >= first &&
                                solo_rank < first + Scratch::polyRows) {
                            has_contact = collidePairLane(pair,
                                w->scratch.lanePoly() +
                                    (solo_rank - first) * Scratch::polyDwords,
                                stage + lane, &too_big, &unsupported,
                                Scratch::polyVerts);
                        }
                    }
                }
                // (the hull-hull scratch may be the rows' storage)
                wave::phaseFence();

                PHYS_PROF(3);
                uint64_t hull_pairs = wave::groupBallot<LPW>(kind == 2);
                cost_work += 24u * (uint32_t)__builtin_popcountll(hull_pairs);
#ifdef MADRONA_PHYS_WAVELOG
                wavelog_hull_pairs += (uint32_t)__builtin_popcountll(hull_pairs);
                {
                    // (hull-hull rounds of the wavefront: each half its own pairs,
                    // against the halves sharing them out)
                    const uint64_t all = __builtin_amdgcn_ballot_w64(kind == 2);
                    const uint32_t na = (uint32_t)__builtin_popcountll(all & 0xFFFFFFFFull);
                    const uint32_t nb = (uint32_t)__builtin_popcountll(all >> 32);
                    wavelog_iters_max += na > nb ? na : nb;
                    wavelog_iters_shared += (na + nb + 1u) / 2u;
                }
#endif
#ifdef MADRONA_PHYS_PROFILE_LDS
                {
                    const uint64_t solo_pairs =
                        wave::groupBallot<LPW>(kind == 1 || kind == 3);
                    if (lane == 0u) {
                        atomicAdd(&prof_lds[group][12],
                                  (uint32_t)__builtin_popcountll(hull_pairs));
                        atomicAdd(&prof_lds[group][14],
                                  (uint32_t)__builtin_popcountll(solo_pairs));
                        atomicAdd(&prof_lds[group][15], width);
                    }
                }
#endif
#ifdef MADRONA_PHYS_PROFILE
                // (event counts next to the cycle counters: lane 0 of a world)
                {
                    const uint64_t solo_pairs = wave::groupBallot<LPW>(kind == 1);
                    if (lane == 0) {
                        prof_acc[12] +=
                            (unsigned long long)__builtin_popcountll(hull_pairs);
                        prof_acc[14] +=
                            (unsigned long long)__builtin_popcountll(solo_pairs);
                        prof_acc[15] += width;
                    }
                }
#endif
                // Two worlds per wavefront: the halves share their pairs out.  A
                // half with pairs of its own takes the first of them; one that has
                // none left takes the LAST of the other world's while that world
                // still has two or more (it takes its first itself) -- both halves
                // derive who takes what from the same two masks, so nothing is
                // agreed on at run time.  The test runs on the taker's lanes and
                // in its scratch; the contact lands in the slot of the lane that
                // owns the candidate, which also learns the outcome.
                [[maybe_unused]] uint32_t other_pairs = 0;
                [[maybe_unused]] bool partner_takes = false;
                [[maybe_unused]] uint32_t partner_chunk = 0, partner_contacts = 0;
                [[maybe_unused]] WaveCandidate *partner_spilled = nullptr;
                if constexpr (worlds_per_wave == 2) {
                    const uint64_t all = __builtin_amdgcn_ballot_w64(kind == 2);
                    const uint64_t here = __builtin_amdgcn_ballot_w64(true);
                    const uint32_t partner_shift = group != 0 ? 0u : 32u;
                    partner_takes = may_share &&
                        (uint32_t)(here >> partner_shift) != 0u;
                    if (partner_takes) {
                        other_pairs = (uint32_t)(all >> partner_shift);
                    }
                    partner_chunk = (uint32_t)__shfl_xor((int32_t)chunk, 32, 64);
                    partner_contacts =
                        (uint32_t)__shfl_xor((int32_t)num_contacts, 32, 64);
                    partner_spilled = (WaveCandidate *)(uintptr_t)fromPartner64(
                        (unsigned long long)(uintptr_t)spilled_candidates);
                }
                while (hull_pairs != 0 ||
                       (worlds_per_wave == 2 &&
                        __builtin_popcount(other_pairs) >= 2)) {
                    // the next pair of the world, with all of its lanes -- or the
                    // other world's last
                    const bool steal = hull_pairs == 0;
                    uint32_t src;
                    if (!steal) {
                        src = (uint32_t)__builtin_ctzll(hull_pairs);
                    } else {
                        src = 31u - (uint32_t)__builtin_clz(other_pairs);
                    }
                    // (what the other half does in this round, from the same masks)
                    int32_t taken_from_me = -1;
                    if constexpr (worlds_per_wave == 2) {
                        if (partner_takes && other_pairs == 0 &&
                                __builtin_popcountll(hull_pairs) >= 2) {
                            taken_from_me = 63 - __builtin_clzll(hull_pairs);
                        }
                        if (steal) {
                            // (the other half takes its first, this one its last)
                            other_pairs &= other_pairs - 1;
                            other_pairs &= ~(1u << src);
                        } else if (other_pairs != 0) {
                            other_pairs &= other_pairs - 1;
                        }
                    }
                    if (!steal) {
                        hull_pairs &= hull_pairs - 1;
                    }
                    if (taken_from_me >= 0) {
                        hull_pairs &= ~(1ull << taken_from_me);
                    }

                    bool pair_too_big = false;
                    const Block *pair_world = steal ? partner_w : w;
                    const uint32_t c = (steal ? partner_chunk : chunk) + src;
                    const WaveCandidate candidate =
                        c < (uint32_t)Block::maxCandidates ? pair_world->candidates[c] :
                            (steal ? partner_spilled : spilled_candidates)[
                                c - (uint32_t)Block::maxCandidates];
                    PairSetup shared_pair = ldsSetupPair(pair_world, obj_mgr, candidate);
                    PackedContact *pair_out = steal ?
                        partner_w->contacts() + partner_contacts + src : stage + src;
                    // every lane writes the same contact to the owner's slot
                    const bool found = hullHullWave<LPW>(lane, shared_pair,
                        w->scratch.hull(), pair_out, &pair_too_big,
                        PHYS_HH_PROF());
                    // the lane that owns the candidate learns the outcome
                    const uint32_t outcome = __shfl(
                        (found ? 1u : 0u) | (pair_too_big ? 2u : 0u), 0, LPW);
                    if (!steal && lane == src) {
                        has_contact = (outcome & 1u) != 0u;
                        too_big = (outcome & 2u) != 0u;
                    }
                    if constexpr (worlds_per_wave == 2) {
                        const uint32_t theirs =
                            (uint32_t)__shfl_xor((int32_t)outcome, 32, 64);
                        if (taken_from_me >= 0 && (int32_t)lane == taken_from_me) {
                            has_contact = (theirs & 1u) != 0u;
                            too_big = (theirs & 2u) != 0u;
                        }
                    }
#ifdef MADRONA_PHYS_PROFILE
                    if (lane == 0) {
                        prof_acc[13] += outcome & 1u;
                    }
#endif
#ifdef MADRONA_PHYS_PROFILE_LDS
                    if (lane == 0u) {
                        atomicAdd(&prof_lds[group][13], outcome & 1u);
                    }
#endif
                }

                // Hulls whose faces outgrow the LDS scratch (rare: none in the
                // Escape Room or Hide-and-Seek shapes): the generic routine,
                // one lane at a time through the world's scratch in HBM -- no
                // per-lane arrays in private memory for a path that is almost
                // never taken.
                uint64_t big_pairs = wave::groupBallot<LPW>(too_big);
                while (big_pairs != 0) {
                    const uint32_t src = (uint32_t)__builtin_ctzll(big_pairs);
                    big_pairs &= big_pairs - 1;
                    if (lane == src) {
                        geo::Plane *tmp_faces = (geo::Plane *)(
                            ps->worldHullScratch +
                            (size_t)world * PhysicsScratch::hullScratchBytes);
                        static_assert(PhysicsScratch::hullScratchBytes >=
                            MADRONA_PHYS_MAX_HULL_ELEMS *
                                (sizeof(geo::Plane) + sizeof(math::Vector3)));
                        math::Vector3 *tmp_vertices = (math::Vector3 *)(
                            tmp_faces + MADRONA_PHYS_MAX_HULL_ELEMS);
                        PairSetup pair = ldsSetupPair(w, obj_mgr,
                                                      candidateAt(chunk + lane));
                        has_contact = collidePairStored(pair, tmp_vertices,
                            tmp_faces, MADRONA_PHYS_MAX_HULL_ELEMS, stage + lane,
                            &unsupported);
                    }
                    wave::phaseFence();
                }
                if (unsupported) {
                    mwhip::raiseError(S, mwhip::kErrPhysics);
                }
                wave::phaseFence();

                // compact in place: all reads, then all writes
                uint64_t mask = wave::groupBallot<LPW>(has_contact);
                const uint32_t rank = wave::rankInGroup(mask, lane);
                const bool moves = has_contact && rank != lane;
                PackedContact moved;
                if (moves) {
                    moved = stage[lane];
                }
                wave::phaseFence();
                if (moves) {
                    stage[rank] = moved;
                }
                num_contacts += (uint32_t)__builtin_popcountll(mask);
                chunk += width;
            }
            if (contacts_overflow) {
                // more contacts than the block holds: the world is stepped out
                // of HBM instead (nothing of this step has left LDS yet)
                toFallback(world);
                bailed = true;
                break;
            }
            cost_work += 8u * num_contacts;
#ifdef MADRONA_PHYS_WAVELOG
            wavelog_contacts += num_contacts;
#endif
            wave::phaseFence();
            PHYS_PROF(5);

            // ---- position solve: contacts, then joints, level by level ----------
            // (the levels of the first window of contacts -- all of them, in
            // every world seen so far -- serve the velocity solve as well: same
            // contacts, same bodies)
            uint32_t first_level = 0, first_max_level = 0;
            for (uint32_t base = 0; base < num_contacts; base += LPW) {
                const uint32_t n = num_contacts - base < (uint32_t)LPW ?
                    num_contacts - base : (uint32_t)LPW;
                const uint32_t i = base + lane;
                uint32_t key_a = 0, key_b = 0;
                if (lane < n) {
                    key_a = ldsBodyKey(w, w->contacts()[i].refBody());
                    key_b = ldsBodyKey(w, w->contacts()[i].altBody());
                }
                uint32_t level = constraintLevels<LPW>(lane, n, key_a, key_b);
                uint32_t max_level = wave::maxReduce<LPW>(lane < n ? level : 0);
                if (base == 0) {
                    first_level = level;
                    first_max_level = max_level;
                }

                for (uint32_t l = 0; l <= max_level; l++) {
                    // two lanes per contact, a body each (xpbd::paired): the
                    // level's contacts in index order over the world's teams
                    const uint64_t members =
                        wave::groupBallot<LPW>(lane < n && level == l);
                    const uint32_t count = (uint32_t)__builtin_popcountll(members);
                    for (uint32_t first = 0; first < count; first += LPW / 2) {
                        const uint32_t team = first + (lane >> 1);
                        if (team < count) {
                            const uint32_t ci =
                                base + wave::nthSetBit<LPW>(members, team);
                            const bool second = (lane & 1u) != 0u;
                            float lambda_n[4] { 0.f, 0.f, 0.f, 0.f };
                            xpbd::paired::handleContact(second, store,
                                w->contacts()[ci].view(), lambda_n);
                            if (!second) {
                                w->lambdas[ci] = lambda_n[0];
                            }
                        }
                    }
                    wave::phaseFence();
                }
            }

            // (at most maxJointBodies of them: one window)
            if (num_joints > 0) {
                const uint32_t n = (uint32_t)num_joints;
                Loc l1 { 0, 0 }, l2 { 0, 0 };
                uint32_t key_a = 0, key_b = 0;
                if (lane < n) {
                    l1 = Loc { 0, (int32_t)w->jointBodies[lane][0] };
                    l2 = Loc { 0, (int32_t)w->jointBodies[lane][1] };
                    key_a = ldsBodyKey(w, l1.row);
                    key_b = ldsBodyKey(w, l2.row);
                }
                uint32_t level = constraintLevels<LPW>(lane, n, key_a, key_b);
                uint32_t max_level = wave::maxReduce<LPW>(lane < n ? level : 0);

                for (uint32_t l = 0; l <= max_level; l++) {
                    // (two lanes per joint, an end each; at most maxJointBodies
                    // joints: one round of teams)
                    const uint64_t members =
                        wave::groupBallot<LPW>(lane < n && level == l);
                    const uint32_t team = lane >> 1;
                    if (team < (uint32_t)__builtin_popcountll(members)) {
                        const uint32_t j = wave::nthSetBit<LPW>(members, team);
                        const bool second = (lane & 1u) != 0u;
                        const Loc me { 0, (int32_t)w->jointBodies[j][second ? 1 : 0] };
                        // (rows beyond the block's: out of the sorted table)
                        xpbd::paired::handleJointConstraint(second, store, me,
                            j < (uint32_t)Block::maxJoints ? w->joints[j] : joints[j]);
                    }
                    wave::phaseFence();
                }
            }

            PHYS_PROF(4);
            // ---- velocities -----------------------------------------------------
            for (int32_t k = (int32_t)lane; k < num_bodies; k += LPW) {
                w->vel[k] = xpbd::deriveVelocity(w->poses[k].x, w->poses[k].q,
                    store.prevState(Loc { 0, k }), w->sys.h);
            }
            wave::phaseFence();
            PHYS_PROF(4);

            for (uint32_t base = 0; base < num_contacts; base += LPW) {
                const uint32_t n = num_contacts - base < (uint32_t)LPW ?
                    num_contacts - base : (uint32_t)LPW;
                const uint32_t i = base + lane;
                uint32_t level = first_level;
                uint32_t max_level = first_max_level;
                if (base != 0) {
                    uint32_t key_a = 0, key_b = 0;
                    if (lane < n) {
                        key_a = ldsBodyKey(w, w->contacts()[i].refBody());
                        key_b = ldsBodyKey(w, w->contacts()[i].altBody());
                    }
                    level = constraintLevels<LPW>(lane, n, key_a, key_b);
                    max_level = wave::maxReduce<LPW>(lane < n ? level : 0);
                }

                for (uint32_t l = 0; l <= max_level; l++) {
                    const uint64_t members =
                        wave::groupBallot<LPW>(lane < n && level == l);
                    const uint32_t count = (uint32_t)__builtin_popcountll(members);
                    for (uint32_t first = 0; first < count; first += LPW / 2) {
                        const uint32_t team = first + (lane >> 1);
                        if (team < count) {
                            const uint32_t ci =
                                base + wave::nthSetBit<LPW>(members, team);
                            float lambda_n[4] { w->lambdas[ci], 0.f, 0.f, 0.f };
                            xpbd::paired::solveVelocitiesForContact(
                                (lane & 1u) != 0u, store,
                                w->contacts()[ci].view(), lambda_n, w->sys.h,
                                w->sys.restitutionThreshold);
                        }
                    }
                    wave::phaseFence();
                }
            }
            PHYS_PROF(11);
        }

        PHYS_PROF(6);
        if (bailed) {
            continue;
        }
        // ---- store: LDS -> HBM --------------------------------------------------
        // The addresses of the body's six columns in ONE round of loads, then
        // the stores, none of them waited for.  (Through ctx.getDirect every
        // column was: load the column address, wait -- for the stores before
        // it as well, they share the counter --, store: 14 us per pair of
        // worlds in the phase profile, now 3.)
#pragma unroll
        for (int32_t c = 0; c < chunks; c++) {
            const int32_t k = c * LPW + (int32_t)lane;
            if (k >= num_bodies) {
                continue;
            }
            const Loc loc = mine.loc[c];
            const TableHdr &tbl = mwhip::tablesOf(S)[loc.archetype];
            base::Position *col_pos = (base::Position *)
                mwhip::loadGlobal(&tbl.columns[RGDCols::Position]);
            base::Rotation *col_rot = (base::Rotation *)
                mwhip::loadGlobal(&tbl.columns[RGDCols::Rotation]);
            Velocity *col_vel = (Velocity *)
                mwhip::loadGlobal(&tbl.columns[RGDCols::Velocity]);
            xpbd::SubstepPrevState *col_prev = (xpbd::SubstepPrevState *)
                mwhip::loadGlobal(&tbl.columns[xpbd::XPBDCols::SubstepPrevState]);
            xpbd::PreSolvePositional *col_pre_pos = (xpbd::PreSolvePositional *)
                mwhip::loadGlobal(&tbl.columns[xpbd::XPBDCols::PreSolvePositional]);
            xpbd::PreSolveVelocity *col_pre_vel = (xpbd::PreSolveVelocity *)
                mwhip::loadGlobal(&tbl.columns[xpbd::XPBDCols::PreSolveVelocity]);
            roundIssued();
            const base::Position pos = w->poses[k].x;
            const base::Rotation rot = w->poses[k].q;
            const Velocity vel = w->vel[k];
            const xpbd::SubstepPrevState prev = store.prevState(Loc { 0, k });
            const xpbd::PreSolvePositional pre_pos =
                store.presolvePositional(Loc { 0, k });
            const xpbd::PreSolveVelocity pre_vel =
                store.presolveVelocity(Loc { 0, k });
            mwhip::storeGlobal(col_pos + loc.row, pos);
            mwhip::storeGlobal(col_rot + loc.row, rot);
            mwhip::storeGlobal(col_vel + loc.row, vel);
            mwhip::storeGlobal(col_prev + loc.row, prev);
            mwhip::storeGlobal(col_pre_pos + loc.row, pre_pos);
            mwhip::storeGlobal(col_pre_vel + loc.row, pre_vel);
        }
        // (the stores are not waited for: nothing of this job reads them, the
        // end of the kernel covers them, and the block is only rewritten by
        // this wavefront's own later LDS writes)
        __builtin_amdgcn_wave_barrier();
        PHYS_PROF(7);

        // what this world cost: the wavefront's time, shared out between its two
        // worlds (they run in lock step: the clock alone cannot tell them apart)
        {
            uint32_t cost = (uint32_t)((long long)wall_clock64() - cost_t0);
            if (LPW == 32) {
                const uint32_t other = __shfl_xor(cost_work, 32, 64);
                const uint32_t most = cost_work > other ? cost_work : other;
                cost = (uint32_t)(((uint64_t)cost * cost_work) / most);
            }
            if (lane == 0) {
                params.worldCost[world] = cost;
            }
#ifdef MADRONA_PHYS_WAVELOG
            if (lane == 0 && S->moduleData[1] != nullptr) {
                uint32_t *rec = (uint32_t *)S->moduleData[1] + 8 * (size_t)world;
                rec[0] = (uint32_t)job;
                rec[1] = wavelog_predicted;
                rec[2] = wavelog_iters_max | (wavelog_iters_shared << 16);
                rec[3] = (uint32_t)((long long)wall_clock64() - cost_t0);
                rec[4] = (uint32_t)((unsigned long long)cost_t0 & 0xFFFFFFFFull);
                rec[5] = num_candidates | ((uint32_t)num_bodies << 16);
                rec[6] = wavelog_hull_pairs | (wavelog_contacts << 16);
                rec[7] = cost;
            }
#endif
        }
    } while (false);
    }

#ifdef MADRONA_PHYS_PROFILE_LDS
    wave::phaseFence();
    if (lane < 16u && S->moduleData[1] != nullptr) {
        atomicAdd(&((unsigned long long *)S->moduleData[1])[lane],
                  (unsigned long long)prof_lds[group][lane]);
    }
#undef PHYS_PROF
#define PHYS_PROF(slot) do {} while (0)
#endif
#ifdef MADRONA_PHYS_PROFILE
    if (lane == 0 && S->moduleData[1] != nullptr) {
        unsigned long long *dst = (unsigned long long *)S->moduleData[1];
#pragma unroll
        for (int i = 0; i < 32; i++) {
            atomicAdd(&dst[i], prof_acc[i]);
        }
    }
#endif
}

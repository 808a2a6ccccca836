// Part of the fused rigid-body step (phys_impl/world_step.inl includes the parts
// in order, inside namespace madrona::phys::kernels): the step out of HBM (physicsStepKernel: worlds beyond 128 bodies, and the fallback behind the LDS kernels) + what both kernels share.

struct WorldBodies {
    // Every loop over these arrays is fully unrolled with the bound below and
    // predicated on numArchetypes: indexed by a run-time value they live in
    // scratch memory, and filling them was a chain of scratch round trips at
    // the head of every world (8 % of the step kernel's cycles).
    static constexpr uint32_t maxArchetypes = PhysicsScratch::maxBodyArchetypes;

    uint32_t numArchetypes;
    uint32_t archetype[maxArchetypes];
    int32_t rowBase[maxArchetypes];
    int32_t bodyBase[maxArchetypes + 1];

    __device__ inline int32_t count() const
    {
        int32_t n = 0;
#pragma unroll
        for (uint32_t a = 0; a < maxArchetypes; a++) {
            if (a < numArchetypes) {
                n = bodyBase[a + 1];
            }
        }
        return n;
    }

    // k-th body of the world in the CPU backend's iteration order
    __device__ inline Loc loc(int32_t k) const
    {
        uint32_t arch = archetype[0];
        int32_t row = rowBase[0] + k;
#pragma unroll
        for (uint32_t a = 1; a < maxArchetypes; a++) {
            if (a < numArchetypes && k >= bodyBase[a]) {
                arch = archetype[a];
                row = rowBase[a] + (k - bodyBase[a]);
            }
        }
        return Loc { arch, row };
    }

    // row ranges of `world` in the rigid-body tables; false: a table is unsorted
    __device__ inline bool fill(mwhip::EcsState *S, const PhysicsScratch *ps,
                                int32_t world)
    {
        numArchetypes = ps->numBodyArchetypes;
        // (the pointers first, then what they lead to: two rounds of loads)
        const int32_t *offsets[maxArchetypes];
        const int32_t *counts[maxArchetypes];
        uint32_t unsorted = 0;
#pragma unroll
        for (uint32_t a = 0; a < maxArchetypes; a++) {
            offsets[a] = nullptr;
            counts[a] = nullptr;
            archetype[a] = 0;
            if (a < numArchetypes) {
                archetype[a] = ps->bodyArchetypes[a];
                const mwhip::TableHdr &tbl = S->tables[archetype[a]];
                offsets[a] = tbl.worldOffsets;
                counts[a] = tbl.worldCounts;
                unsorted |= tbl.needsSort;
            }
        }
        int32_t rows[maxArchetypes];
#pragma unroll
        for (uint32_t a = 0; a < maxArchetypes; a++) {
            rowBase[a] = 0;
            rows[a] = 0;
            if (a < numArchetypes) {
                rowBase[a] = offsets[a][world];
                rows[a] = counts[a][world];
            }
        }
        bodyBase[0] = 0;
#pragma unroll
        for (uint32_t a = 0; a < maxArchetypes; a++) {
            bodyBase[a + 1] = bodyBase[a] + rows[a];
        }
        return unsorted == 0;
    }
};

// Dependency levels for a window of <= 64 constraints held one per lane.
// key_a / key_b: the two bodies (0 = static / none, never conflicts).
template <int LPW = 64>
__device__ inline uint32_t constraintLevels(uint32_t lane, uint32_t n,
                                            uint64_t key_a, uint64_t key_b)
{
    uint32_t level = 0;
    for (uint32_t j = 0; j + 1 < n; j++) {
        uint64_t ja = __shfl(key_a, j, LPW);
        uint64_t jb = __shfl(key_b, j, LPW);
        uint32_t jl = __shfl(level, j, LPW);
        bool conflict =
            (ja != 0 && (ja == key_a || ja == key_b)) ||
            (jb != 0 && (jb == key_a || jb == key_b));
        if (lane > j && lane < n && conflict && jl + 1 > level) {
            level = jl + 1;
        }
    }
    return level;
}

// (32-bit keys: body indices inside an LDS-resident world)
// (Round 6 also built the levels without this chain of dependent shuffles --
// conflict masks in n independent rounds, then one ballot per level -- and
// measured no difference at two wavefronts per SIMD: 525-530 us against
// 524-526, profiles/r06_levels_variants.jsonl.  Not kept.  Constraint j's words
// through v_readlane instead of __shfl: step 0.802 -> 0.812 ms, ten more
// spilled SGPRs.  Not kept either.)
template <int LPW = 64>
__device__ inline uint32_t constraintLevels(uint32_t lane, uint32_t n,
                                            uint32_t key_a, uint32_t key_b)
{
    uint32_t level = 0;
    for (uint32_t j = 0; j + 1 < n; j++) {
        const uint32_t ja = __shfl(key_a, j, LPW);
        const uint32_t jb = __shfl(key_b, j, LPW);
        const uint32_t jl = __shfl(level, j, LPW);
        const bool conflict =
            (ja != 0u && (ja == key_a || ja == key_b)) ||
            (jb != 0u && (jb == key_a || jb == key_b));
        if (lane > j && lane < n && conflict && jl + 1 > level) {
            level = jl + 1;
        }
    }
    return level;
}

// A static body does not order the constraints that touch it as long as the
// solver's writes to it are no-ops.  Positions are (x += 0), but the reference
// renormalises the rotation in every positional update (xpbd.cpp
// applyPositionalUpdate), so that only holds while the rotation is a fixed
// point of normalize() -- e.g. not for a tilted body that was switched to
// Static mid-flight; such a body orders its constraints like a dynamic one.
__device__ inline bool staticBodyIsInert(math::Quat q)
{
    math::Quat n = q.normalize();
    return n.w == q.w && n.x == q.x && n.y == q.y && n.z == q.z;
}

__device__ inline uint64_t bodyKey(Context &ctx, Loc loc)
{
    if (ctx.getDirect<ResponseType>(RGDCols::ResponseType, loc) ==
            ResponseType::Static &&
        staticBodyIsInert(
            ctx.getDirect<base::Rotation>(RGDCols::Rotation, loc))) {
        return 0;
    }
    return ((uint64_t)(loc.archetype + 1) << 32) | (uint64_t)(uint32_t)loc.row;
}

#ifndef MADRONA_PHYS_WAVES_PER_EU
#define MADRONA_PHYS_WAVES_PER_EU 1
#endif
__global__ void __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(MADRONA_PHYS_WAVES_PER_EU)))
physicsStepKernel(EcsState *S, void *node_data, uint32_t fallback_mode, uint32_t)
{
    mwhip::TraceScope trace_scope(S);
    StateManager *state_mgr = static_cast<StateManager *>(S);
    PhysicsScratch *ps = detail::scratch(S);
    const PhysicsStepParams params = *(const PhysicsStepParams *)node_data;

    const uint32_t lane = wave::laneID();
    const int32_t waves_per_block = (int32_t)(blockDim.x / 64);
    const int32_t wave_in_block = __builtin_amdgcn_readfirstlane(
        (int32_t)(threadIdx.x / 64));
    const int32_t num_worlds = S->numWorlds;

    // generic fallback only (hulls whose faces outgrow the LDS scratch)
    constexpr int32_t max_elems = MADRONA_PHYS_MAX_HULL_ELEMS;
    geo::Plane tmp_faces[max_elems];
    math::Vector3 tmp_vertices[max_elems];

    __shared__ WaveScratch block_scratch[4];
    WaveScratch *scratch = &block_scratch[wave_in_block];

#ifdef MADRONA_PHYS_PROFILE
    // per-phase cycle counters (debug builds): moduleData[1] -> uint64[8]
    unsigned long long prof_t = __builtin_readcyclecounter();
    unsigned long long prof_acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#define PHYS_PROF(slot) do { unsigned long long now_ = __builtin_readcyclecounter(); \
        prof_acc[slot] += now_ - prof_t; prof_t = now_; } while (0)
#else
#define PHYS_PROF(slot) do {} while (0)
#endif

    const uint32_t cand_stride = ps->candidatesPerWorld;
    const uint32_t contact_stride = ps->contactsPerWorld;

    // fallback mode (launched behind an LDS step kernel): only the worlds that
    // kernel listed -- too many bodies or contacts for its block
    const int32_t *fallback_list = fallback_mode != 0u ?
        ((const PhysicsStepParams *)node_data)->fallbackList : nullptr;
    const int32_t num_jobs = fallback_list != nullptr ?
        __hip_atomic_load(fallback_list, __ATOMIC_RELAXED,
                          __HIP_MEMORY_SCOPE_AGENT) : num_worlds;

    for (int32_t job = (int32_t)blockIdx.x * waves_per_block + wave_in_block;
         job < num_jobs; job += (int32_t)gridDim.x * waves_per_block) {
        const int32_t world = fallback_list != nullptr ?
            __hip_atomic_load(fallback_list + 1 + job, __ATOMIC_RELAXED,
                              __HIP_MEMORY_SCOPE_AGENT) : job;
        Context ctx = TaskGraph::makeContext<Context>(
            state_mgr, WorldID { world }, true);
        const ObjectManager &obj_mgr = *ctx.singleton<ObjectData>().mgr;
        const PhysicsSystemState physics_sys =
            ctx.singleton<PhysicsSystemState>();

        // ---- the world's bodies ---------------------------------------------
        WorldBodies bodies;
        const bool unsorted = !bodies.fill(S, ps, world);
        if (unsorted) {
            mwhip::raiseError(S, mwhip::kErrPhysics);
            continue;
        }
        const int32_t num_bodies = bodies.count();
        PHYS_PROF(0);

        CandidateCollision *candidates =
            ps->worldCandidates + (uint64_t)world * cand_stride;
        ContactConstraint *contacts =
            ps->worldContacts + (uint64_t)world * contact_stride;
        float *lambdas = ps->worldLambdas + (uint64_t)world * contact_stride;

        // ---- broadphase: candidate pairs in (body, traversal) order -----------
        uint32_t num_candidates = 0;
        for (int32_t chunk = 0; chunk < num_bodies; chunk += 64) {
            const int32_t k = chunk + (int32_t)lane;
            const bool active = k < num_bodies;

            Loc a_loc = active ? bodies.loc(k) : Loc { 0, 0 };
            Entity e = Entity::none();
            broadphase::LeafID leaf_id { 0 };
            uint32_t n = 0;
            if (active) {
                e = ctx.getDirect<Entity>(0, a_loc);
                leaf_id = ctx.getDirect<broadphase::LeafID>(
                    RGDCols::LeafID, a_loc);
                detail::forEachCandidate(ctx, e, leaf_id, a_loc,
                    [&](Loc, CountT a_num_prims, CountT b_num_prims) {
                        n += (uint32_t)(a_num_prims * b_num_prims);
                    });
            }

            uint32_t chunk_total;
            uint32_t out = num_candidates +
                wave::exclusiveScan(n, lane, &chunk_total);

            if (active && n != 0 && out + n <= cand_stride) {
                detail::forEachCandidate(ctx, e, leaf_id, a_loc,
                    [&](Loc b_loc, CountT a_num_prims, CountT b_num_prims) {
                        CountT total_checks = a_num_prims * b_num_prims;
                        for (CountT c = 0; c < total_checks; c++) {
                            CandidateCollision &candidate = candidates[out++];
                            candidate.a = a_loc;
                            candidate.b = b_loc;
                            candidate.aPrim = (uint32_t)(c / b_num_prims);
                            candidate.bPrim = (uint32_t)(c % b_num_prims);
                        }
                    });
            }
            num_candidates += chunk_total;
        }
        if (num_candidates > cand_stride) {
            mwhip::raiseError(S, mwhip::kErrTableOverflow);
            continue;
        }
        wave::phaseFence();
        PHYS_PROF(1);

        // ---- the world's joints (table sorted by world just before) -----------
        const TableHdr &joint_tbl = S->tables[ps->jointArchetype];
        const int32_t joint_begin = joint_tbl.worldOffsets[world];
        const int32_t num_joints = joint_tbl.worldCounts[world];
        const JointConstraint *joints =
            (const JointConstraint *)joint_tbl.columns[2] + joint_begin;

        for (int32_t substep = 0; substep < params.numSubsteps; substep++) {
            // ---- integrate ------------------------------------------------------
            for (int32_t k = (int32_t)lane; k < num_bodies; k += 64) {
                Loc loc = bodies.loc(k);
                xpbd::substepRigidBodies(ctx,
                    ctx.getDirect<base::Position>(RGDCols::Position, loc),
                    ctx.getDirect<base::Rotation>(RGDCols::Rotation, loc),
                    ctx.getDirect<Velocity>(RGDCols::Velocity, loc),
                    ctx.getDirect<base::ObjectID>(RGDCols::ObjectID, loc),
                    ctx.getDirect<ResponseType>(RGDCols::ResponseType, loc),
                    ctx.getDirect<ExternalForce>(RGDCols::ExternalForce, loc),
                    ctx.getDirect<ExternalTorque>(RGDCols::ExternalTorque, loc),
                    ctx.getDirect<xpbd::SubstepPrevState>(
                        xpbd::XPBDCols::SubstepPrevState, loc),
                    ctx.getDirect<xpbd::PreSolvePositional>(
                        xpbd::XPBDCols::PreSolvePositional, loc),
                    ctx.getDirect<xpbd::PreSolveVelocity>(
                        xpbd::XPBDCols::PreSolveVelocity, loc));
            }
            wave::phaseFence();
            PHYS_PROF(2);

            // ---- narrowphase: contacts in candidate order ------------------------
            uint32_t num_contacts = 0;
            for (uint32_t chunk = 0; chunk < num_candidates; chunk += 64) {
                const uint32_t c = chunk + lane;
                ContactConstraint contact;
                bool has_contact = false;
                bool too_big = false;
                bool unsupported = false;

                // per lane: order the pair, reject by world AABBs, classify
                uint32_t kind = 0;      // 1: this lane alone, 2: whole wave
                PairSetup pair;
                if (c < num_candidates) {
                    pair = setupPair(ctx, obj_mgr, candidates[c]);
                    if (pair.aabbOverlap) {
                        kind = pair.test == NarrowphaseTest::HullHull ? 2 : 1;
                    }
                }

                // lanes on their own, in rounds of lanePolyRows scratch rows
                uint64_t solo = __builtin_amdgcn_ballot_w64(kind == 1);
                const uint32_t solo_rank = wave::rankInBallot(solo);
                const uint32_t solo_count = (uint32_t)__builtin_popcountll(solo);
                for (uint32_t first = 0; first < solo_count;
                     first += lanePolyRows) {
                    if (kind == 1 && solo_rank >= first &&
                            solo_rank < first + lanePolyRows) {
                        has_contact = collidePairLane(pair,
                            scratch->lanePoly +
                                (solo_rank - first) * lanePolyDwords,
                            &contact, &too_big, &unsupported);
                    }
                }

                // hull-hull pairs: one after the other, SAT loops over the lanes
                uint64_t hull_pairs = __builtin_amdgcn_ballot_w64(kind == 2);
                while (hull_pairs != 0) {
                    const uint32_t src = (uint32_t)__builtin_ctzll(hull_pairs);
                    hull_pairs &= hull_pairs - 1;

                    PairSetup shared_pair =
                        setupPair(ctx, obj_mgr, candidates[chunk + src]);
                    ContactConstraint shared_contact;
                    bool shared_too_big = false;
                    bool found = hullHullWave(lane, shared_pair, &scratch->hull,
                        &shared_contact, &shared_too_big);
                    if (lane == src) {
                        contact = shared_contact;
                        has_contact = found;
                        too_big = shared_too_big;
                    }
                }

                // rare: polygons larger than the LDS scratch -> generic path
                // with the hulls stored in the lane's private memory
                if (too_big) {
                    has_contact = collidePairStored(pair, tmp_vertices,
                        tmp_faces, max_elems, &contact, &unsupported);
                }
                if (unsupported) {
                    mwhip::raiseError(S, mwhip::kErrPhysics);
                }

                uint64_t mask = __builtin_amdgcn_ballot_w64(has_contact);
                uint32_t dst = num_contacts + wave::rankInBallot(mask);
                if (has_contact && dst < contact_stride) {
                    contacts[dst] = contact;
                }
                num_contacts += (uint32_t)__builtin_popcountll(mask);
            }
            if (num_contacts > contact_stride) {
                mwhip::raiseError(S, mwhip::kErrTableOverflow);
                num_contacts = contact_stride;
            }
            wave::phaseFence();
            PHYS_PROF(3);

            // ---- position solve: contacts, then joints, level by level ----------
            for (uint32_t base = 0; base < num_contacts; base += 64) {
                const uint32_t n = num_contacts - base < 64 ?
                    num_contacts - base : 64;
                const uint32_t i = base + lane;
                uint64_t key_a = 0, key_b = 0;
                if (lane < n) {
                    key_a = bodyKey(ctx, contacts[i].ref);
                    key_b = bodyKey(ctx, contacts[i].alt);
                }
                uint32_t level = constraintLevels(lane, n, key_a, key_b);
                uint32_t max_level = wave::maxReduce(lane < n ? level : 0);

                for (uint32_t l = 0; l <= max_level; l++) {
                    if (lane < n && level == l) {
                        float lambda_n[4] { 0.f, 0.f, 0.f, 0.f };
                        xpbd::handleContact(ctx, obj_mgr, contacts[i], lambda_n);
                        lambdas[i] = lambda_n[0];
                    }
                    wave::phaseFence();
                }
            }

            for (int32_t base = 0; base < num_joints; base += 64) {
                const uint32_t n = num_joints - base < 64 ?
                    (uint32_t)(num_joints - base) : 64u;
                const int32_t i = base + (int32_t)lane;
                uint64_t key_a = 0, key_b = 0;
                if (lane < n) {
                    key_a = bodyKey(ctx, ctx.loc(joints[i].e1));
                    key_b = bodyKey(ctx, ctx.loc(joints[i].e2));
                }
                uint32_t level = constraintLevels(lane, n, key_a, key_b);
                uint32_t max_level = wave::maxReduce(lane < n ? level : 0);

                for (uint32_t l = 0; l <= max_level; l++) {
                    if (lane < n && level == l) {
                        xpbd::handleJointConstraint(ctx, obj_mgr, joints[i]);
                    }
                    wave::phaseFence();
                }
            }

            PHYS_PROF(4);
            // ---- velocities -----------------------------------------------------
            for (int32_t k = (int32_t)lane; k < num_bodies; k += 64) {
                Loc loc = bodies.loc(k);
                xpbd::setVelocities(ctx,
                    ctx.getDirect<base::Position>(RGDCols::Position, loc),
                    ctx.getDirect<base::Rotation>(RGDCols::Rotation, loc),
                    ctx.getDirect<xpbd::SubstepPrevState>(
                        xpbd::XPBDCols::SubstepPrevState, loc),
                    ctx.getDirect<Velocity>(RGDCols::Velocity, loc));
            }
            wave::phaseFence();
            PHYS_PROF(5);

            for (uint32_t base = 0; base < num_contacts; base += 64) {
                const uint32_t n = num_contacts - base < 64 ?
                    num_contacts - base : 64;
                const uint32_t i = base + lane;
                uint64_t key_a = 0, key_b = 0;
                if (lane < n) {
                    key_a = bodyKey(ctx, contacts[i].ref);
                    key_b = bodyKey(ctx, contacts[i].alt);
                }
                uint32_t level = constraintLevels(lane, n, key_a, key_b);
                uint32_t max_level = wave::maxReduce(lane < n ? level : 0);

                for (uint32_t l = 0; l <= max_level; l++) {
                    if (lane < n && level == l) {
                        float lambda_n[4] { lambdas[i], 0.f, 0.f, 0.f };
                        xpbd::solveVelocitiesForContact(ctx, obj_mgr,
                            contacts[i], lambda_n, physics_sys.h,
                            physics_sys.restitutionThreshold);
                    }
                    wave::phaseFence();
                }
            }
            PHYS_PROF(6);
        }
    }

#ifdef MADRONA_PHYS_PROFILE
    if (lane == 0 && S->moduleData[1] != nullptr) {
        unsigned long long *dst = (unsigned long long *)S->moduleData[1];
        for (int i = 0; i < 8; i++) {
            atomicAdd(&dst[i], prof_acc[i]);
        }
    }
#endif
}

// Part of the fused rigid-body step (phys_impl/world_step.inl includes the parts
// in order, inside namespace madrona::phys::kernels): narrowphase on a wavefront: cooperative hull-hull SAT, clipping, manifolds.

// ---------------------------------------------------------------------------
// narrowphase on a wavefront
// ---------------------------------------------------------------------------
// Per-lane clipping scratch in LDS: lanePolyVerts points + depths per lane,
// rows padded to an odd dword count so lanes fall into different banks.
inline constexpr uint32_t lanePolyVerts = 8;
inline constexpr uint32_t lanePolyDwords = lanePolyVerts * 4 + 1;
// rows available per round; lanes that need one are served in rounds
inline constexpr uint32_t lanePolyRows = 16;
// clipping scratch of the wave-cooperative hull-hull path (2 polygons)
inline constexpr uint32_t wavePolyVerts = 24;

// world-space copies of the two hulls of a cooperative hull-hull test
inline constexpr uint32_t waveHullElems = 16;   // vertices, faces per hull

// scratch of ONE cooperative hull-hull test (a world runs as many at a time as
// it has lane groups for them: hullHullWave<G>)
// (ELEMS: vertices / faces per hull it holds in world space -- a larger hull is
// evaluated lazily; POLY: corners of the reference + the incident face it clips
// -- more: the pair takes the per-lane HBM path)
template <uint32_t ELEMS, uint32_t POLY>
struct alignas(16) HullScratchT {
    static constexpr uint32_t elems = ELEMS;
    static constexpr uint32_t poly = POLY;
    math::Vector3 clip[2][POLY];
    math::Vector3 hullVerts[2][ELEMS];
    geo::Plane hullPlanes[2][ELEMS];
};
using HullScratch = HullScratchT<waveHullElems, wavePolyVerts>;

struct alignas(16) WaveScratch {
    float lanePoly[lanePolyRows * lanePolyDwords];
    HullScratch hull;
};

// number of vertices of face `face_idx`
template <typename HullT>
__device__ inline uint32_t faceVertexCount(const HullT &h, uint32_t face_idx)
{
    uint32_t n = 0;
    uint32_t hedge_idx = h.faceBaseHedge(face_idx);
    const uint32_t start = hedge_idx;
    do {
        hedge_idx = h.hedge(hedge_idx).next;
        n++;
    } while (hedge_idx != start);
    return n;
}

template <int LPW>
__device__ inline Vector3 shflVec3(Vector3 v, int src);

// SAT face query with the faces of `a` spread over the lanes (sequential
// reference: narrowphase.hpp queryFaceDirections)
template <int LPW = 64, typename HullA, typename HullB>
__device__ inline FaceQuery queryFaceDirectionsWave(uint32_t lane,
                                                    const HullA &a,
                                                    const HullB &b)
{
    float best_sep = -FLT_MAX;
    uint32_t best_face = 0xFFFFFFFFu;

    const uint32_t num_a_faces = (uint32_t)a.numFaces();
    for (uint32_t f = lane; f < num_a_faces; f += LPW) {
        float face_dist = getHullDistanceFromPlane(a.plane(f), b);
        if (face_dist > best_sep) {
            best_sep = face_dist;
            best_face = f;
        }
    }
    wave::argMaxFirst<LPW>(best_sep, best_face);

    FaceQuery best;
    best.separation = best_sep;
    if (best_face == 0xFFFFFFFFu) {
        best.faceIdx = -1;
        best.plane = Plane { Vector3::zero(), 0.f };
    } else {
        best.faceIdx = (CountT)best_face;
        best.plane = a.plane(best_face);
    }
    return best;
}

// SAT edge query with the (edge of a, edge of b) pairs spread over the lanes
// (sequential reference: narrowphase.hpp queryEdgeDirections)
template <int LPW = 64, typename HullA, typename HullB>
__device__ inline EdgeQuery queryEdgeDirectionsWave(uint32_t lane,
                                                    const HullA &a,
                                                    const HullB &b)
{
    float best_sep = -FLT_MAX;
    uint32_t best_pair = 0xFFFFFFFFu;
    Vector3 best_normal = Vector3::zero();

    const uint32_t b_num_edges = (uint32_t)b.numEdges();
    const uint32_t num_pairs = (uint32_t)a.numEdges() * b_num_edges;
    // (two or three rounds per iteration -- #pragma unroll -- to have their loads
    // in flight together: no difference, profiles/r06_edge_unroll_variants.jsonl)
    for (uint32_t p = lane; p < num_pairs; p += LPW) {
        int32_t he_idx_a = (int32_t)((p / b_num_edges) * 2);
        int32_t he_idx_b = (int32_t)((p % b_num_edges) * 2);
        EdgeTestResult r = testEdgePair(a, b, he_idx_a, he_idx_b);
        if (r.separation > best_sep) {
            best_sep = r.separation;
            best_pair = p;
            best_normal = r.normal;
        }
    }
    wave::argMaxFirst<LPW>(best_sep, best_pair);

    EdgeQuery best;
    best.separation = best_sep;
    if (best_pair == 0xFFFFFFFFu) {
        best.normal = Vector3::zero();
        best.edgeIdxA = 0;
        best.edgeIdxB = 0;
    } else {
        best.edgeIdxA = (int32_t)((best_pair / b_num_edges) * 2);
        best.edgeIdxB = (int32_t)((best_pair % b_num_edges) * 2);
        // (the lane that tested the winning pair -- pair p went to lane p % LPW --
        // still holds its normal: no second test of the pair)
        best.normal = shflVec3<LPW>(best_normal, (int)(best_pair % (uint32_t)LPW));
    }
    return best;
}

// makeHullState with the vertices / planes spread over the G lanes of the test.
// The centroid is left out: only the edge query reads it, and only hull a's
// (narrowphase.hpp testEdgePair) -- hullCentroid() below, once both face
// queries have failed to separate the pair (half of the tests end before).
template <int G = 64>
__device__ inline HullState makeHullStateWave(uint32_t lane,
                                              const HalfEdgeMesh &mesh,
                                              const PrimitiveTransform &txfm,
                                              Vector3 *dst_vertices,
                                              Plane *dst_planes)
{
    LazyHull lazy(mesh, txfm.pos, txfm.rot, txfm.scale, false);
    for (uint32_t i = lane; i < mesh.numVertices; i += G) {
        dst_vertices[i] = lazy.vertex(i);
    }
    for (uint32_t i = lane; i < mesh.numFaces; i += G) {
        dst_planes[i] = lazy.plane(i);
    }

    HalfEdgeMesh world_mesh = mesh;
    world_mesh.facePlanes = dst_planes;
    world_mesh.vertices = dst_vertices;
    return HullState { world_mesh, Vector3::zero() };
}

// the centroid is a sequential sum (fp order): every lane adds it up
__device__ inline void hullCentroid(HullState &hull)
{
    Vector3 center = Vector3::zero();
    const CountT num_vertices = (CountT)hull.mesh.numVertices;
    for (CountT i = 0; i < num_vertices; i++) {
        center += hull.mesh.vertices[i];
    }
    center /= (float)num_vertices;
    hull.center = center;
}
__device__ inline void hullCentroid(LazyHull &) {}     // (has it already)

// Profile builds (-DMADRONA_PHYS_PROFILE): cycles and exit counts of the stages
// of a cooperative hull-hull test, accumulated in the calling kernel's own
// counters (slots 16.. of prof_acc: registers -- atomics per mark distort the
// very thing they measure; profiles/tools/phys_phase_cycles.py).  Its own
// switch (-DMADRONA_PHYS_PROFILE_HH on top of -DMADRONA_PHYS_PROFILE): the 16
// extra accumulators push the kernel into spilling, which inflates the phase
// figures of the same build; use it for the stage split and exit counts only.
struct HullHullProf {
#ifdef MADRONA_PHYS_PROFILE_LDS_HH
    // (accumulators in LDS, slots 0 .. 8 of the step kernel's prof_lds row: a
    // build that keeps its registers, profiles/tools/phys_phase_cycles.py HH)
    uint32_t *acc;
    unsigned long long t;
    __device__ inline void mark(uint32_t lane, int slot)
    {
        unsigned long long now = __builtin_readcyclecounter();
        if (lane == 0u) atomicAdd(&acc[slot], (uint32_t)(now - t));
        t = now;
    }
    __device__ inline void count(uint32_t lane, int slot)
    {
        if (lane == 0u) atomicAdd(&acc[slot], 1u);
    }
#elif defined(MADRONA_PHYS_PROFILE_HH)
    unsigned long long *acc;
    unsigned long long t;
    __device__ inline void mark(uint32_t, int slot)
    {
        unsigned long long now = __builtin_readcyclecounter();
        acc[16 + slot] += now - t;
        t = now;
    }
    __device__ inline void count(uint32_t, int slot)
    {
        acc[16 + slot] += 1ull;
    }
#else
    __device__ inline void mark(uint32_t, int) {}
    __device__ inline void count(uint32_t, int) {}
#endif
};

template <int LPW = 64, typename ScratchT, typename HullA, typename HullB,
          typename OutT>
__device__ inline bool hullHullWaveSAT(uint32_t lane, const PairSetup &pair,
                                       HullA &a, const HullB &b,
                                       ScratchT *scratch,
                                       OutT *out, bool *too_big,
                                       HullHullProf prof = HullHullProf {});

// Hull-hull pair handled by a group of LPW lanes (`pair` is uniform across the
// group; `lane` = index inside it).  Returns false with *too_big set when the
// clipped polygon may not fit the LDS scratch.
// (A template so that only the device pass instantiates it.  Keeping it out of
// line to confine its register footprint was measured: 1166 -> 1637 us.)
template <int LPW = 64, typename ScratchT = HullScratch,
          typename OutT = ContactConstraint>
__device__ inline bool
hullHullWave(uint32_t lane, const PairSetup &pair,
                                    ScratchT *scratch,
                                    OutT *out, bool *too_big,
                                    HullHullProf prof = HullHullProf {})
{
    const HalfEdgeMesh &a_mesh = pair.aPrim->hull.halfEdgeMesh;
    const HalfEdgeMesh &b_mesh = pair.bPrim->hull.halfEdgeMesh;

    if (a_mesh.numVertices <= ScratchT::elems &&
        a_mesh.numFaces <= ScratchT::elems &&
        b_mesh.numVertices <= ScratchT::elems &&
        b_mesh.numFaces <= ScratchT::elems) {
        // small hulls: transform once into LDS
        HullState a = makeHullStateWave<LPW>(lane, a_mesh, pair.a,
            scratch->hullVerts[0], scratch->hullPlanes[0]);
        HullState b = makeHullStateWave<LPW>(lane, b_mesh, pair.b,
            scratch->hullVerts[1], scratch->hullPlanes[1]);
        wave::phaseFence();
#ifdef MADRONA_PHYS_EAGER_CENTROID
        // (round 3, for A/B measurements: both centroids up front)
        hullCentroid(a);
        hullCentroid(b);
#endif
        prof.mark(lane, 0);     // hulls into LDS
        return hullHullWaveSAT<LPW>(lane, pair, a, b, scratch, out, too_big,
                                    prof);
    }

    LazyHull a(a_mesh, pair.a.pos, pair.a.rot, pair.a.scale);
    LazyHull b(b_mesh, pair.b.pos, pair.b.rot, pair.b.scale);
    return hullHullWaveSAT<LPW>(lane, pair, a, b, scratch, out, too_big, prof);
}

// ---------------------------------------------------------------------------
// Face contact with the polygon spread over the lanes (sequential reference:
// narrowphase.hpp createFaceContact + clipPolygon + buildFaceContactManifold).
// The sequential routines walk polygons of <= 8 vertices through LDS one vertex
// at a time -- a few hundred dependent LDS round trips per overlapping pair,
// which every lane of the group repeats in lock step: 20 K of the ~45 K cycles
// such a pair costs.  Here lane i HOLDS vertex i:
//   * Sutherland-Hodgman against side plane k: lane i looks at the edge
//     (vertex i - 1 -> vertex i), emits 0, 1 or 2 vertices, and finds its place
//     in the output with two ballots (the sequential loop's order);
//   * the reduction to four points: "first index that reaches the maximum, if it
//     beats the starting value" = wave::argMaxFirst over the lanes, three times.
// Same expressions on the same operands, so the same bits; every lane returns
// the same manifold.  Requires n_ref, n_inc and every clipped polygon <= LPW
// vertices (a clip adds at most one vertex per plane).
// ---------------------------------------------------------------------------
template <int LPW>
__device__ inline Vector3 shflVec3(Vector3 v, int src)
{
    return Vector3 { __shfl(v.x, src, LPW), __shfl(v.y, src, LPW),
                     __shfl(v.z, src, LPW) };
}

template <int LPW>
__device__ inline Manifold createFaceContactWave(uint32_t lane, Plane ref_plane,
                                                 int32_t ref_face_idx,
                                                 int32_t incident_face_idx,
                                                 const HullState &ref,
                                                 const HullState &other,
                                                 Vector3 *lds_a, float *lds_b)
{
    const uint64_t lane_lt = (1ull << lane) - 1ull;

    // lane i: vertex i of the incident face, vertex i of the reference face
    // (the rings are linked lists: every lane walks them, keeps its own)
    Vector3 v = Vector3::zero();
    int32_t n = 0;
    {
        uint32_t hedge_idx = other.faceBaseHedge(incident_face_idx);
        const uint32_t start_hedge_idx = hedge_idx;
        uint32_t my_root = 0;
        do {
            const HalfEdge cur_hedge = other.hedge(hedge_idx);
            hedge_idx = cur_hedge.next;
            if ((uint32_t)n == lane) my_root = cur_hedge.rootVertex;
            n++;
        } while (hedge_idx != start_hedge_idx);
        if ((int32_t)lane < n) v = other.vertex(my_root);
    }
    Plane side_plane { Vector3::zero(), 0.f };
    int32_t n_ref = 0;
    {
        uint32_t hedge_idx = ref.faceBaseHedge(ref_face_idx);
        const uint32_t start_hedge_idx = hedge_idx;
        uint32_t my_root = 0;
        do {
            const HalfEdge cur_hedge = ref.hedge(hedge_idx);
            hedge_idx = cur_hedge.next;
            if ((uint32_t)n_ref == lane) my_root = cur_hedge.rootVertex;
            n_ref++;
        } while (hedge_idx != start_hedge_idx);
        Vector3 cur_point = Vector3::zero();
        if ((int32_t)lane < n_ref) cur_point = ref.vertex(my_root);
        // side plane k runs from point k to point k + 1 (the ring closes)
        const int32_t next_lane = (int32_t)lane + 1 < n_ref ? (int32_t)lane + 1 : 0;
        const Vector3 next_point = shflVec3<LPW>(cur_point, next_lane);
        const Vector3 edge = next_point - cur_point;
        const Vector3 plane_normal = cross(edge, ref_plane.normal);
        side_plane = Plane { plane_normal, dot(plane_normal, cur_point) };
    }

    // ---- clip against every side plane, in ring order ----
    for (int32_t k = 0; k < n_ref; k++) {
        const Plane clip {
            shflVec3<LPW>(side_plane.normal, k), __shfl(side_plane.d, k, LPW) };
        const bool have = (int32_t)lane < n;
        const float d2 = getDistanceFromPlane(clip, v);
        const int32_t prev = lane == 0u ? n - 1 : (int32_t)lane - 1;
        const Vector3 v1 = shflVec3<LPW>(v, prev < 0 ? 0 : prev);
        const float d1 = __shfl(d2, prev < 0 ? 0 : prev, LPW);

        // what the sequential loop emits for the edge v1 -> v (in this order)
        const bool crossing = have && ((d1 <= 0.0f && d2 > 0.0f) ||
                                       (d2 <= 0.0f && d1 > 0.0f));
        const bool keep = have && d2 <= 0.0f;
        const uint64_t m_cross = wave::groupBallot<LPW>(crossing);
        const uint64_t m_keep = wave::groupBallot<LPW>(keep);
        const int32_t at = (int32_t)__builtin_popcountll(m_cross & lane_lt) +
            (int32_t)__builtin_popcountll(m_keep & lane_lt);
        if (crossing) {
            lds_a[at] = planeIntersection(clip, v1, v);
        }
        if (keep) {
            lds_a[at + (crossing ? 1 : 0)] = v;
        }
        n = (int32_t)__builtin_popcountll(m_cross) +
            (int32_t)__builtin_popcountll(m_keep);
        wave::phaseFence();
        v = (int32_t)lane < n ? lds_a[lane] : Vector3::zero();
        wave::phaseFence();
    }

    // ---- what lies below the reference plane, projected onto it ----
    float depth = 0.f;
    int32_t m = 0;
    {
        const float d = getDistanceFromPlane(ref_plane, v);
        const bool below = (int32_t)lane < n && d <= 0.0f;
        const uint64_t m_below = wave::groupBallot<LPW>(below);
        if (below) {
            const int32_t at = (int32_t)__builtin_popcountll(m_below & lane_lt);
            lds_a[at] = v - d * ref_plane.normal;
            lds_b[at] = -d;
        }
        m = (int32_t)__builtin_popcountll(m_below);
        wave::phaseFence();
        v = (int32_t)lane < m ? lds_a[lane] : Vector3::zero();
        depth = (int32_t)lane < m ? lds_b[lane] : 0.f;
        wave::phaseFence();
    }

    // ---- the <= 4 points that best preserve the polygon (lane i = contact i) ----
    Manifold manifold;
    for (int i = 0; i < 4; i++) {
        manifold.contactPoints[i] = Vector3::zero();
        manifold.penetrationDepths[i] = 0.f;
    }
    auto contactOf = [&](uint32_t src, int slot) {
        manifold.contactPoints[slot] = shflVec3<LPW>(v, (int)src);
        manifold.penetrationDepths[slot] = __shfl(depth, (int)src, LPW);
    };
    const Vector3 contact_normal = ref_plane.normal;
    if (m <= 4) {
        manifold.numContactPoints = m;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const Vector3 p = shflVec3<LPW>(v, i);
            const float dp = __shfl(depth, i, LPW);
            if (i < m) {
                manifold.contactPoints[i] = p;
                manifold.penetrationDepths[i] = dp;
            }
        }
    } else {
        manifold.numContactPoints = 4;
        contactOf(0u, 0);
        const bool candidate = lane >= 1u && (int32_t)lane < m;

        // farthest from the first point (first index wins, must beat 0)
        float max_dist_sq = 0.f;
        {
            float val = candidate ? manifold.contactPoints[0].distance2(v) : -FLT_MAX;
            uint32_t idx = lane;
            wave::argMaxFirst<LPW>(val, idx);
            if (val > 0.f) {
                max_dist_sq = val;
                contactOf(idx, 1);
            }
        }
        Vector3 ba = manifold.contactPoints[1] - manifold.contactPoints[0];

        // largest triangle with the first two
        float max_tri_area = 0.0f;
        {
            const Vector3 bc = v - manifold.contactPoints[1];
            const float signed_area = contact_normal.dot(cross(ba, bc));
            float val = candidate ? copysignf(signed_area, 1.f) : -FLT_MAX;
            uint32_t idx = lane;
            wave::argMaxFirst<LPW>(val, idx);
            if (val > 0.f) {
                max_tri_area = val;
                contactOf(idx, 2);
            }
            // (the reference keeps the winning sign in a bool that is never -1:
            // its swap of the first two points never happens, narrowphase.hpp)
        }

        const Vector3 cb = manifold.contactPoints[2] - manifold.contactPoints[1];
        const Vector3 ac = manifold.contactPoints[0] - manifold.contactPoints[2];

        // most outside that triangle
        float most_neg_area = 0.f;
        {
            const Vector3 aq = manifold.contactPoints[0] - v;
            const Vector3 qc = v - manifold.contactPoints[2];
            const float abq_area = contact_normal.dot(cross(ba, aq));
            const float bcq_area = contact_normal.dot(cross(cb, qc));
            const float caq_area = contact_normal.dot(cross(aq, ac));
            const float q_min_area = fminf(abq_area, fminf(bcq_area, caq_area));
            float val = candidate ? -q_min_area : -FLT_MAX;
            uint32_t idx = lane;
            wave::argMaxFirst<LPW>(val, idx);
            if (val > 0.f) {
                most_neg_area = -val;
                contactOf(idx, 3);
            }
        }

        if (max_dist_sq == 0.f || max_tri_area == 0.f || most_neg_area == 0.f) {
            manifold.numContactPoints = 0;
            manifold.normal = Vector3::zero();
            return manifold;
        }
    }

    // (the identity transform of the sequential routine: it turns -0 into +0)
    const Vector3 world_offset { 0, 0, 0 };
    const Quat to_world_frame { 1, 0, 0, 0 };
    for (int i = 0; i < 4; i++) {
        if (i < manifold.numContactPoints) {
            manifold.contactPoints[i] =
                to_world_frame.rotateVec(manifold.contactPoints[i]) + world_offset;
        }
    }
    manifold.normal = to_world_frame.rotateVec(contact_normal);
    return manifold;
}

// (only hulls staged in LDS take the lane-parallel manifold)
template <int LPW, typename ScratchT, typename HullA, typename HullB,
          typename OutT>
__device__ inline bool faceContactWave(uint32_t lane, const SATResult &sat,
                                       const HullA &a, const HullB &b,
                                       const PairSetup &pair, ScratchT *scratch,
                                       uint32_t n_ref, uint32_t n_inc,
                                       OutT *out, bool *found)
{
  if constexpr (!std::is_same_v<HullA, HullState> ||
                !std::is_same_v<HullB, HullState>) {
    return false;
  } else {
#ifdef MADRONA_PHYS_SEQUENTIAL_MANIFOLD
    return false;
#else
    // every polygon of the clipping fits the group: the incident face gains at
    // most one vertex per side plane
    if (n_ref + n_inc > (uint32_t)LPW) {
        return false;
    }
    const uint32_t ref_face = sat.contact.refFaceIdxOrEdgeIdxA & 0x7FFFFFFFu;
    const bool a_is_ref = ref_face == sat.contact.refFaceIdxOrEdgeIdxA;
    const Plane ref_plane { sat.contact.normal, sat.contact.planeDOrSeparation };
    const Manifold manifold = a_is_ref ?
        createFaceContactWave<LPW>(lane, ref_plane, (int32_t)ref_face,
            (int32_t)sat.contact.incidentFaceIdxOrEdgeIdxB, a, b,
            scratch->clip[0], (float *)scratch->clip[1]) :
        createFaceContactWave<LPW>(lane, ref_plane, (int32_t)ref_face,
            (int32_t)sat.contact.incidentFaceIdxOrEdgeIdxB, b, a,
            scratch->clip[0], (float *)scratch->clip[1]);
    // barely touching pairs can lose every clipped point to fp32
    *found = manifold.numContactPoints != 0;
    if (*found) {
        manifoldToContact(manifold, a_is_ref ? pair.aLoc : pair.bLoc,
                          a_is_ref ? pair.bLoc : pair.aLoc, out);
    }
    return true;
#endif
  }
}

template <int LPW, typename ScratchT, typename HullA, typename HullB,
          typename OutT>
__device__ inline bool hullHullWaveSAT(uint32_t lane, const PairSetup &pair,
                                       HullA &a, const HullB &b,
                                       ScratchT *scratch,
                                       OutT *out, bool *too_big,
                                       HullHullProf prof)
{
    FaceQuery face_query_a = queryFaceDirectionsWave<LPW>(lane, a, b);
    prof.mark(lane, 1);
    if (face_query_a.separation > 0.0f) {
        prof.count(lane, 5);
        return false;
    }

    FaceQuery face_query_b = queryFaceDirectionsWave<LPW>(lane, b, a);
    prof.mark(lane, 2);
    if (face_query_b.separation > 0.0f) {
        prof.count(lane, 6);
        return false;
    }

#ifndef MADRONA_PHYS_EAGER_CENTROID
    hullCentroid(a);
#endif
    EdgeQuery edge_query = queryEdgeDirectionsWave<LPW>(lane, a, b);
    prof.mark(lane, 3);
    if (edge_query.separation > 0.0f) {
        prof.count(lane, 7);
        return false;
    }
    prof.count(lane, 8);

    // from here on every lane computes the same thing (cheap, and it keeps the
    // wave converged); the clipping polygons live in LDS
    const SATResult sat =
        chooseSATContact(a, b, face_query_a, face_query_b, edge_query);

    if (sat.type == ContactType::SATFace) {
        uint32_t ref_face = sat.contact.refFaceIdxOrEdgeIdxA & 0x7FFFFFFFu;
        bool a_is_ref = ref_face == sat.contact.refFaceIdxOrEdgeIdxA;
        uint32_t inc_face = sat.contact.incidentFaceIdxOrEdgeIdxB;
        uint32_t n_ref = a_is_ref ? faceVertexCount(a, ref_face) :
                                    faceVertexCount(b, ref_face);
        uint32_t n_inc = a_is_ref ? faceVertexCount(b, inc_face) :
                                    faceVertexCount(a, inc_face);
        if (n_ref + n_inc > ScratchT::poly) {
            *too_big = true;
            return false;
        }
        bool found_wave = false;
        if (faceContactWave<LPW>(lane, sat, a, b, pair, scratch, n_ref, n_inc, out,
                                 &found_wave)) {
            prof.mark(lane, 4);
            return found_wave;
        }
    }

    const bool found = satToContact(sat, a, b, pair.aLoc, pair.bLoc,
                                    scratch->clip[0], scratch->clip[1], out);
    prof.mark(lane, 4);
    return found;
}

// ---------------------------------------------------------------------------
// Hull against plane by a TEAM of two lanes (the two-worlds-per-wavefront
// kernel; sequential reference: hullPlaneContact = doSATPlane +
// createFacePlaneContact + buildFaceContactManifold, physics.inl /
// narrowphase.hpp).  One lane per pair spends most of its ~1.2 K instructions
// on loops over the hull's vertices and faces -- every vertex against the plane,
// every face's normal for the incident face, the incident face's corners -- and
// the kernel is bound by the instructions it issues.  Here
// lane `second` = false takes the first half of every such range and its
// partner the second half; the halves meet in a minimum, an arg-min (the lower
// index wins ties: the first half's) and a corner count, exchanged inside the
// quad.  Faces of up to MaxCorners corners (more: *too big*, like a face that
// does not fit a lane's LDS row): the surviving corners stay in registers and go
// straight into the packed contact, in ring order -- no clipping scratch.
// Returns bit 0: a contact was written, bit 1: the face is too big.
// ---------------------------------------------------------------------------
__device__ inline float teamPartner(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(
        __builtin_bit_cast(int, v), 0xB1 /* quad_perm:[1,0,3,2] */, 0xF, 0xF,
        true));
}

__device__ inline int32_t teamPartner(int32_t v)
{
    return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);
}

// (OutT: the LDS step's PackedContact)
template <int MaxCorners = 4, typename OutT>
__device__ inline uint32_t hullPlaneContactTeam(bool second, const LazyHull &a_hull,
                                                const PrimitiveTransform &plane_txfm,
                                                Loc a_loc, Loc b_loc,
                                                OutT *out)
{
    static_assert(MaxCorners % 2 == 0);
    constexpr Vector3 base_normal = { 0, 0, 1 };
    Vector3 plane_normal = plane_txfm.rot.rotateVec(base_normal);

    Plane plane { plane_normal, dot(plane_normal, plane_txfm.pos) };

    // ---- doSATPlane: getHullDistanceFromPlane ----
    {
        const CountT num_verts = a_hull.numVertices();
        const CountT half = (num_verts + 1) / 2;
        const CountT end = second ? num_verts : half;
        float min_dot_n = FLT_MAX;
        for (CountT i = second ? half : 0; i < end; i++) {
            float cur_dot = dot(a_hull.vertex(i), plane.normal);
            if (cur_dot < min_dot_n) {
                min_dot_n = cur_dot;
            }
        }
        const float theirs = teamPartner(min_dot_n);
        const float of_first = second ? theirs : min_dot_n;
        const float of_second = second ? min_dot_n : theirs;
        min_dot_n = of_second < of_first ? of_second : of_first;

        float separation = min_dot_n - plane.d;
        if (separation > 0.0f) {
            return 0u;
        }
    }

    // ---- findIncidentFace ----
    int32_t incident_face_idx;
    {
        const CountT num_faces = a_hull.numFaces();
        const CountT half = (num_faces + 1) / 2;
        const CountT end = second ? num_faces : half;
        float min_dot = FLT_MAX;
        int32_t minimizing_face = -1;
        for (CountT face_idx = second ? half : 0; face_idx < end; face_idx++) {
            float face_dot_ref = dot(a_hull.plane(face_idx).normal, plane.normal);
            if (face_dot_ref < min_dot) {
                min_dot = face_dot_ref;
                minimizing_face = (int32_t)face_idx;
            }
        }
        const float their_dot = teamPartner(min_dot);
        const int32_t their_face = teamPartner(minimizing_face);
        const float first_dot = second ? their_dot : min_dot;
        const float second_dot = second ? min_dot : their_dot;
        const int32_t first_face = second ? their_face : minimizing_face;
        const int32_t second_face = second ? minimizing_face : their_face;
        incident_face_idx = second_dot < first_dot ? second_face : first_face;
    }

    // ---- the incident face's corners (every lane walks the ring, keeps the
    // roots of its half) ----
    constexpr int32_t mine_max = MaxCorners / 2;
    uint32_t roots[mine_max];
    int32_t num_corners = 0;
    {
#pragma unroll
        for (int32_t j = 0; j < mine_max; j++) roots[j] = 0;
        uint32_t hedge_idx = a_hull.faceBaseHedge(incident_face_idx);
        const uint32_t start_hedge_idx = hedge_idx;
        do {
            const HalfEdge cur_hedge = a_hull.hedge(hedge_idx);
            hedge_idx = cur_hedge.next;
            const int32_t slot = num_corners - (second ? mine_max : 0);
#pragma unroll
            for (int32_t j = 0; j < mine_max; j++) {
                if (slot == j) roots[j] = cur_hedge.rootVertex;
            }
            num_corners++;
        } while (hedge_idx != start_hedge_idx);
    }
    if (num_corners > MaxCorners) {
        return 2u;
    }

    // ---- createFacePlaneContact: the first lane holds corners [0, MaxCorners / 2),
    // its partner the rest; what lies on or below the plane, projected onto it ----
    Vector3 points[mine_max];
    float depths[mine_max];
    bool below[mine_max];
    int32_t my_count = 0;
#pragma unroll
    for (int32_t j = 0; j < mine_max; j++) {
        const int32_t corner = j + (second ? mine_max : 0);
        Vector3 vertex = a_hull.vertex(roots[j]);
        float d = getDistanceFromPlane(plane, vertex);
        below[j] = corner < num_corners && d <= 0.0f;
        points[j] = vertex - d * plane.normal;
        depths[j] = -d;
        my_count += below[j] ? 1 : 0;
    }
    const int32_t their_count = teamPartner(my_count);
    const int32_t num_contacts = my_count + their_count;

    // ---- buildFaceContactManifold (at most four points: all of them) +
    // manifoldToContact: the plane is always b and always the reference ----
    if (num_contacts == 0) {
        return 0u;
    }
    const Vector3 world_offset { 0, 0, 0 };
    const Quat to_world_frame { 1, 0, 0, 0 };
    int32_t at = second ? their_count : 0;
#pragma unroll
    for (int32_t j = 0; j < mine_max; j++) {
        if (below[j]) {
            out->points[at] = math::Vector4::fromVec3W(
                to_world_frame.rotateVec(points[j]) + world_offset, depths[j]);
            at++;
        }
    }
    if (!second) {
        for (int32_t i = num_contacts; i < 4; i++) {
            out->points[i] = math::Vector4::zero();
        }
        out->normal = to_world_frame.rotateVec(plane.normal);
        out->meta = (uint32_t)b_loc.row | ((uint32_t)a_loc.row << 8) |
            ((uint32_t)num_contacts << 16);
    }
    return 1u;
}

// Every other primitive pair: one lane, hull evaluated lazily, clipping
// scratch in the lane's LDS row.
// row: poly_verts points + poly_verts depths of clipping scratch (a hull whose
// face has more corners comes back with *too_big set)
template <typename OutT>
__device__ inline bool collidePairLane(const PairSetup &pair, float *row,
                                       OutT *out,
                                       bool *too_big, bool *unsupported,
                                       uint32_t poly_verts = lanePolyVerts)
{
    switch (pair.test) {
    case NarrowphaseTest::SphereSphere:
        return sphereSphereContact(pair, out);
    case NarrowphaseTest::SpherePlane:
        return spherePlaneContact(pair, out);
    case NarrowphaseTest::HullPlane: {
        LazyHull a(pair.aPrim->hull.halfEdgeMesh, pair.a.pos, pair.a.rot,
                   pair.a.scale, false);

        // the contact polygon is (part of) the face SAT picks: it must fit
        // the lane's LDS row
        return hullPlaneContact(a, pair.b, pair.aLoc, pair.bLoc,
                                row, row + poly_verts * 3, out,
                                (CountT)poly_verts, too_big);
    }
    case NarrowphaseTest::SphereHull: {
        // hull in the sphere's frame, evaluated lazily (no centroid needed)
        LazyHull b(pair.bPrim->hull.halfEdgeMesh, pair.b.pos - pair.a.pos,
                   pair.b.rot, pair.b.scale, false);
        return sphereHullContact(pair, b, out);
    }
    case NarrowphaseTest::PlanePlane:
    default:
        *unsupported = true;
        return false;
    }
}

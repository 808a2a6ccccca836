// Narrowphase geometry: sphere / convex hull / plane contact generation.
//
// Pure functions (no ECS access), host + device, so they can be unit-tested on
// the CPU against the reference.  Behaviour and floating-point evaluation
// order follow the reference's scalar path, which is what BOTH of its backends
// run (src/physics/narrowphase.cpp:85-88 disables the warp-cooperative
// variant): hull transform :151-223, SAT face queries :296-365, Minkowski-face
// pruned edge queries :367-566, incident face :569-652, Sutherland-Hodgman
// clipping :659-700, manifold reduction to <= 4 points :771-905, face / plane /
// edge contacts :907-1138, dispatch :1214-1514.
//
// Every routine is a template over the hull representation:
//   * HullState  -- world-space vertices / planes stored in caller scratch (the
//     reference's layout; used on the host and by the generic device path);
//   * LazyHull   -- object-space mesh + transform, world-space vertices and
//     planes recomputed on every access.  Same expressions, no fp contraction,
//     hence the same bits as the stored form -- but no per-lane scratch memory,
//     which on CDNA costs hundreds of cycles per access (phys_impl/
//     world_step.inl uses this form, and splits the hull-hull SAT loops over
//     the lanes of a wavefront).
#pragma once

#include <madrona/geo.hpp>
#include <madrona/math.hpp>
#include <madrona/ecs.hpp>

#ifndef MADRONA_PHYS_MAX_HULL_ELEMS
#define MADRONA_PHYS_MAX_HULL_ELEMS 128
#endif

namespace madrona::phys {

struct CollisionPrimitive;
struct ContactConstraint;

namespace narrowphase {

using math::Vector3;
using math::Vector4;
using math::Quat;
using math::Diag3x3;
using math::Mat3x3;
using geo::HalfEdge;
using geo::HalfEdgeMesh;
using geo::Plane;
using geo::Segment;

// raw CollisionPrimitive::Type values OR-ed together
enum class NarrowphaseTest : uint32_t {
    SphereSphere = 1,
    HullHull = 2,
    SphereHull = 3,
    PlanePlane = 4,
    SpherePlane = 5,
    HullPlane = 6,
};

struct HullState {
    HalfEdgeMesh mesh;      // world-space vertices and planes
    Vector3 center;

    MADRONA_HD inline CountT numVertices() const { return (CountT)mesh.numVertices; }
    MADRONA_HD inline CountT numFaces() const { return (CountT)mesh.numFaces; }
    MADRONA_HD inline CountT numEdges() const { return (CountT)mesh.numEdges(); }
    MADRONA_HD inline Vector3 vertex(CountT i) const { return mesh.vertices[i]; }
    MADRONA_HD inline Plane plane(CountT i) const { return mesh.facePlanes[i]; }
    MADRONA_HD inline HalfEdge hedge(CountT i) const { return mesh.halfEdges[i]; }
    MADRONA_HD inline uint32_t faceBaseHedge(CountT f) const
    {
        return mesh.faceBaseHalfEdges[f];
    }
};

// World-space view of an object-space mesh, evaluated on demand with the
// expressions makeHullState stores.
struct LazyHull {
    const HalfEdgeMesh *mesh;
    Mat3x3 vertexTxfm;
    Mat3x3 normalTxfm;
    Vector3 translation;
    Vector3 center;

    // need_center: only the hull-hull edge test uses the centroid
    MADRONA_HD inline LazyHull(const HalfEdgeMesh &obj_mesh, Vector3 t, Quat r,
                               Diag3x3 s, bool need_center = true)
        : mesh(&obj_mesh), translation(t)
    {
        Mat3x3 unscaled_rot = Mat3x3::fromQuat(r);
        vertexTxfm = unscaled_rot * s;
        normalTxfm = unscaled_rot * s.inv();

        center = Vector3::zero();
        if (!need_center) {
            return;
        }

        const CountT num_vertices = (CountT)obj_mesh.numVertices;
        for (CountT i = 0; i < num_vertices; i++) {
            center += vertex(i);
        }
        center /= (float)num_vertices;
    }

    MADRONA_HD inline CountT numVertices() const { return (CountT)mesh->numVertices; }
    MADRONA_HD inline CountT numFaces() const { return (CountT)mesh->numFaces; }
    MADRONA_HD inline CountT numEdges() const { return (CountT)mesh->numEdges(); }

    MADRONA_HD inline Vector3 vertex(CountT i) const
    {
        return vertexTxfm * mesh->vertices[i] + translation;
    }

    MADRONA_HD inline Plane plane(CountT i) const
    {
        Plane obj_plane = mesh->facePlanes[i];
        Vector3 plane_origin =
            vertexTxfm * (obj_plane.normal * obj_plane.d) + translation;

        Vector3 txfmed_normal = (normalTxfm * obj_plane.normal).normalize();
        return Plane { txfmed_normal, dot(txfmed_normal, plane_origin) };
    }

    MADRONA_HD inline HalfEdge hedge(CountT i) const { return mesh->halfEdges[i]; }
    MADRONA_HD inline uint32_t faceBaseHedge(CountT f) const
    {
        return mesh->faceBaseHalfEdges[f];
    }
};

struct Manifold {
    Vector3 contactPoints[4];
    float penetrationDepths[4];
    int32_t numContactPoints;
    Vector3 normal;
};

enum class ContactType : uint32_t {
    None,
    Sphere,
    SATPlane,
    SATFace,
    SATEdge,
};

struct SphereContact {
    Vector3 normal;
    Vector3 pt;
    float depth;
};

struct SATContact {
    Vector3 normal;
    float planeDOrSeparation;
    uint32_t refFaceIdxOrEdgeIdxA;      // bit 31 set: hull b owns the reference face
    uint32_t incidentFaceIdxOrEdgeIdxB;
};


// ---------------------------------------------------------------------------
// hull setup
// ---------------------------------------------------------------------------
MADRONA_HD inline HullState makeHullState(const HalfEdgeMesh &mesh,
                                          Vector3 translation,
                                          Quat rotation,
                                          Diag3x3 scale,
                                          Vector3 *dst_vertices,
                                          Plane *dst_planes)
{
    Mat3x3 unscaled_rot = Mat3x3::fromQuat(rotation);
    Mat3x3 vertex_txfm = unscaled_rot * scale;
    Mat3x3 normal_txfm = unscaled_rot * scale.inv();

    Vector3 center = Vector3::zero();
    const CountT num_vertices = (CountT)mesh.numVertices;
    for (CountT i = 0; i < num_vertices; i++) {
        Vector3 world_pos = vertex_txfm * mesh.vertices[i] + translation;
        dst_vertices[i] = world_pos;
        center += world_pos;
    }
    center /= (float)num_vertices;

    const CountT num_faces = (CountT)mesh.numFaces;
    for (CountT i = 0; i < num_faces; i++) {
        Plane obj_plane = mesh.facePlanes[i];
        Vector3 plane_origin =
            vertex_txfm * (obj_plane.normal * obj_plane.d) + translation;

        Vector3 txfmed_normal = (normal_txfm * obj_plane.normal).normalize();
        dst_planes[i] = Plane { txfmed_normal, dot(txfmed_normal, plane_origin) };
    }

    HalfEdgeMesh world_mesh = mesh;
    world_mesh.facePlanes = dst_planes;
    world_mesh.vertices = dst_vertices;

    return HullState { world_mesh, center };
}

MADRONA_HD inline float getDistanceFromPlane(const Plane &plane,
                                             const Vector3 &a)
{
    float adotn = dot(a, plane.normal);
    return adotn - plane.d;
}

// point where the line p1 -> p2 crosses the plane
MADRONA_HD inline Vector3 planeIntersection(const Plane &plane,
                                            const Vector3 &p1,
                                            const Vector3 &p2)
{
    float distance = getDistanceFromPlane(plane, p1);
    return p1 + (p2 - p1) * (-distance / plane.normal.dot(p2 - p1));
}

// signed distance of the hull's deepest vertex
template <typename HullT>
MADRONA_HD inline float getHullDistanceFromPlane(const Plane &plane,
                                                 const HullT &h)
{
    float min_dot_n = FLT_MAX;
    const CountT num_verts = h.numVertices();
    for (CountT i = 0; i < num_verts; i++) {
        float cur_dot = dot(h.vertex(i), plane.normal);
        if (cur_dot < min_dot_n) {
            min_dot_n = cur_dot;
        }
    }
    return min_dot_n - plane.d;
}

// ---------------------------------------------------------------------------
// SAT
// ---------------------------------------------------------------------------
struct FaceQuery {
    float separation;
    CountT faceIdx;
    Plane plane;
};

// face of a that b is least behind; stops at the first separating face
template <typename HullA, typename HullB>
MADRONA_HD inline FaceQuery queryFaceDirections(const HullA &a,
                                                const HullB &b)
{
    FaceQuery best;
    best.separation = -FLT_MAX;
    best.faceIdx = -1;
    best.plane = Plane { Vector3::zero(), 0.f };

    const CountT num_a_faces = a.numFaces();
    for (CountT face_idx = 0; face_idx < num_a_faces; face_idx++) {
        Plane plane = a.plane(face_idx);
        float face_dist = getHullDistanceFromPlane(plane, b);

        if (face_dist > best.separation) {
            best.separation = face_dist;
            best.faceIdx = face_idx;
            best.plane = plane;

            if (face_dist > 0) {
                break;
            }
        }
    }

    return best;
}

// do arcs (a,b) and (c,d) intersect on the Gauss map?
MADRONA_HD inline bool isMinkowskiFace(const Vector3 &a, const Vector3 &b,
                                       const Vector3 &c, const Vector3 &d)
{
    Vector3 bxa = b.cross(a);
    Vector3 dxc = d.cross(c);

    float cba = c.dot(bxa);
    float dba = d.dot(bxa);
    float adc = a.dot(dxc);
    float bdc = b.dot(dxc);

    return cba * dba < 0.0f && adc * bdc < 0.0f && cba * bdc > 0.0f;
}

template <typename HullT>
MADRONA_HD inline Segment getEdgeSegment(const HullT &h, HalfEdge start)
{
    return Segment { h.vertex(start.rootVertex),
                     h.vertex(h.hedge(start.next).rootVertex) };
}

struct EdgeTestResult {
    Vector3 normal;
    float separation;
};

template <typename HullA, typename HullB>
MADRONA_HD inline EdgeTestResult edgeDistance(const HullA &a,
                                              const HullB &b,
                                              HalfEdge hedge_a,
                                              HalfEdge hedge_b)
{
    Segment segment_a = getEdgeSegment(a, hedge_a);
    Segment segment_b = getEdgeSegment(b, hedge_b);

    Vector3 dir_a = segment_a.p2 - segment_a.p1;
    Vector3 dir_b = segment_b.p2 - segment_b.p1;

    Vector3 unnormalized_cross = dir_a.cross(dir_b);
    float normal_len2 = unnormalized_cross.length2();

    if (normal_len2 == 0) {
        return EdgeTestResult { Vector3::zero(), -FLT_MAX };
    }

    float inv_normal_len = 1.f / sqrtf(normal_len2);
    Vector3 normal = unnormalized_cross * inv_normal_len;

    // make it point away from a
    if (normal.dot(segment_a.p1 - a.center) < 0.0f) {
        normal = -normal;
    }

    float separation = normal.dot(segment_b.p1 - segment_a.p1);
    return EdgeTestResult { normal, separation };
}

struct EdgeQuery {
    float separation;
    Vector3 normal;
    int32_t edgeIdxA;
    int32_t edgeIdxB;
};

// One (edge of a, edge of b) pair: separation along the pair's axis, or
// -FLT_MAX when the pair does not span a face of the Minkowski difference.
template <typename HullA, typename HullB>
MADRONA_HD inline EdgeTestResult testEdgePair(const HullA &a, const HullB &b,
                                              int32_t he_idx_a,
                                              int32_t he_idx_b)
{
    HalfEdge cur_hedge_a = a.hedge(he_idx_a);
    HalfEdge twin_hedge_a = a.hedge(he_idx_a ^ 1);
    Vector3 a_normal1 = a.plane(cur_hedge_a.face).normal;
    Vector3 a_normal2 = a.plane(twin_hedge_a.face).normal;

    HalfEdge cur_hedge_b = b.hedge(he_idx_b);
    HalfEdge twin_hedge_b = b.hedge(he_idx_b ^ 1);
    Vector3 b_normal1 = b.plane(cur_hedge_b.face).normal;
    Vector3 b_normal2 = b.plane(twin_hedge_b.face).normal;

    if (isMinkowskiFace(a_normal1, a_normal2, -b_normal1, -b_normal2)) {
        return edgeDistance(a, b, cur_hedge_a, cur_hedge_b);
    }

    return EdgeTestResult { Vector3::zero(), -FLT_MAX };
}

// best over all edge pairs in (edge of a)-major order, first maximum wins;
// stops at the first separating pair
template <typename HullA, typename HullB>
MADRONA_HD inline EdgeQuery queryEdgeDirections(const HullA &a,
                                                const HullB &b)
{
    EdgeQuery best;
    best.separation = -FLT_MAX;
    best.normal = Vector3::zero();
    best.edgeIdxA = 0;
    best.edgeIdxB = 0;

    const CountT a_num_edges = a.numEdges();
    const CountT b_num_edges = b.numEdges();

    for (CountT edge_idx_a = 0; edge_idx_a < a_num_edges; edge_idx_a++) {
        int32_t he_idx_a = (int32_t)(edge_idx_a * 2);

        for (CountT edge_idx_b = 0; edge_idx_b < b_num_edges; edge_idx_b++) {
            int32_t he_idx_b = (int32_t)(edge_idx_b * 2);

            EdgeTestResult edge_cmp = testEdgePair(a, b, he_idx_a, he_idx_b);

            if (edge_cmp.separation > best.separation) {
                best.separation = edge_cmp.separation;
                best.normal = edge_cmp.normal;
                best.edgeIdxA = he_idx_a;
                best.edgeIdxB = he_idx_b;

                if (edge_cmp.separation > 0) {
                    return best;
                }
            }
        }
    }

    return best;
}

// face of h most anti-parallel to ref_normal
template <typename HullT>
MADRONA_HD inline CountT findIncidentFace(const HullT &h,
                                          Vector3 ref_normal)
{
    float min_dot = FLT_MAX;
    CountT minimizing_face = -1;

    const CountT num_faces = h.numFaces();
    for (CountT face_idx = 0; face_idx < num_faces; face_idx++) {
        float face_dot_ref = dot(h.plane(face_idx).normal, ref_normal);
        if (face_dot_ref < min_dot) {
            min_dot = face_dot_ref;
            minimizing_face = face_idx;
        }
    }

    return minimizing_face;
}

struct SATResult {
    ContactType type;
    SATContact contact;
};

// turns the three query results (all non-separating) into the contact feature
template <typename HullA, typename HullB>
MADRONA_HD inline SATResult chooseSATContact(const HullA &a, const HullB &b,
                                             const FaceQuery &face_query_a,
                                             const FaceQuery &face_query_b,
                                             const EdgeQuery &edge_query)
{
    SATResult result;
    result.contact = SATContact { Vector3::zero(), 0.f, 0u, 0u };

    bool is_face_contact_a = face_query_a.separation > edge_query.separation;
    bool is_face_contact_b = face_query_b.separation > edge_query.separation;

    if (is_face_contact_a || is_face_contact_b) {
        bool a_is_ref = face_query_a.separation >= face_query_b.separation;

        Plane ref_plane = a_is_ref ? face_query_a.plane : face_query_b.plane;
        CountT ref_face_idx =
            a_is_ref ? face_query_a.faceIdx : face_query_b.faceIdx;

        CountT incident_face_idx = a_is_ref ?
            findIncidentFace(b, ref_plane.normal) :
            findIncidentFace(a, ref_plane.normal);

        result.type = ContactType::SATFace;
        result.contact.normal = ref_plane.normal;
        result.contact.planeDOrSeparation = ref_plane.d;
        result.contact.refFaceIdxOrEdgeIdxA =
            (uint32_t)ref_face_idx | (a_is_ref ? 0u : (1u << 31));
        result.contact.incidentFaceIdxOrEdgeIdxB = (uint32_t)incident_face_idx;
    } else {
        result.type = ContactType::SATEdge;
        result.contact.normal = edge_query.normal;
        result.contact.planeDOrSeparation = edge_query.separation;
        result.contact.refFaceIdxOrEdgeIdxA = (uint32_t)edge_query.edgeIdxA;
        result.contact.incidentFaceIdxOrEdgeIdxB = (uint32_t)edge_query.edgeIdxB;
    }

    return result;
}

template <typename HullA, typename HullB>
MADRONA_HD inline SATResult doSAT(const HullA &a, const HullB &b)
{
    SATResult result;
    result.type = ContactType::None;
    result.contact = SATContact { Vector3::zero(), 0.f, 0u, 0u };

    FaceQuery face_query_a = queryFaceDirections(a, b);
    if (face_query_a.separation > 0.0f) {
        return result;
    }

    FaceQuery face_query_b = queryFaceDirections(b, a);
    if (face_query_b.separation > 0.0f) {
        return result;
    }

    EdgeQuery edge_query = queryEdgeDirections(a, b);
    if (edge_query.separation > 0.0f) {
        return result;
    }

    return chooseSATContact(a, b, face_query_a, face_query_b, edge_query);
}

template <typename HullT>
MADRONA_HD inline SATResult doSATPlane(const Plane &plane, const HullT &h)
{
    SATResult result;
    result.type = ContactType::None;
    result.contact = SATContact { Vector3::zero(), 0.f, 0u, 0u };

    float separation = getHullDistanceFromPlane(plane, h);
    if (separation > 0.0f) {
        return result;
    }

    CountT incident_face_idx = findIncidentFace(h, plane.normal);

    result.type = ContactType::SATPlane;
    result.contact.normal = plane.normal;
    result.contact.planeDOrSeparation = plane.d;
    result.contact.incidentFaceIdxOrEdgeIdxB = (uint32_t)incident_face_idx;
    return result;
}

// ---------------------------------------------------------------------------
// manifolds
// ---------------------------------------------------------------------------

// Sutherland-Hodgman: keeps the part of the polygon behind the plane
MADRONA_HD inline CountT clipPolygon(Vector3 *dst_vertices,
                                     Plane clipping_plane,
                                     const Vector3 *input_vertices,
                                     CountT num_input_vertices)
{
    CountT num_new_vertices = 0;

    Vector3 v1 = input_vertices[num_input_vertices - 1];
    float d1 = getDistanceFromPlane(clipping_plane, v1);

    for (CountT i = 0; i < num_input_vertices; ++i) {
        Vector3 v2 = input_vertices[i];
        float d2 = getDistanceFromPlane(clipping_plane, v2);

        if (d1 <= 0.0f && d2 <= 0.0f) {
            dst_vertices[num_new_vertices++] = v2;
        } else if (d1 <= 0.0f && d2 > 0.0f) {
            dst_vertices[num_new_vertices++] =
                planeIntersection(clipping_plane, v1, v2);
        } else if (d2 <= 0.0f && d1 > 0.0f) {
            dst_vertices[num_new_vertices++] =
                planeIntersection(clipping_plane, v1, v2);
            dst_vertices[num_new_vertices++] = v2;
        }

        v1 = v2;
        d1 = d2;
    }

    return num_new_vertices;
}

// Reduces a contact polygon to the <= 4 points that best preserve its area:
// first point, farthest from it, largest triangle, most outside that triangle.
MADRONA_HD inline Manifold buildFaceContactManifold(Vector3 contact_normal,
                                                    Vector3 *contacts,
                                                    float *penetration_depths,
                                                    CountT num_contacts,
                                                    Vector3 world_offset,
                                                    Quat to_world_frame)
{
    Manifold manifold;
    for (int i = 0; i < 4; i++) {
        manifold.contactPoints[i] = Vector3::zero();
        manifold.penetrationDepths[i] = 0.f;
    }

    // (loops over the manifold's four slots have constant bounds throughout:
    // a dynamically indexed private array lives in scratch memory on the GPU)
    if (num_contacts <= 4) {
        manifold.numContactPoints = (int32_t)num_contacts;
        for (int i = 0; i < 4; i++) {
            if ((CountT)i < num_contacts) {
                manifold.contactPoints[i] = contacts[i];
                manifold.penetrationDepths[i] = penetration_depths[i];
            }
        }
    } else {
        manifold.numContactPoints = 4;
        manifold.contactPoints[0] = contacts[0];
        manifold.penetrationDepths[0] = penetration_depths[0];

        float max_dist_sq = 0.f;
        for (CountT i = 1; i < num_contacts; i++) {
            Vector3 cur_contact = contacts[i];
            float dist_sq = manifold.contactPoints[0].distance2(cur_contact);
            if (dist_sq > max_dist_sq) {
                max_dist_sq = dist_sq;
                manifold.contactPoints[1] = cur_contact;
                manifold.penetrationDepths[1] = penetration_depths[i];
            }
        }

        Vector3 ba = manifold.contactPoints[1] - manifold.contactPoints[0];

        float max_tri_area = 0.0f;
        // NB: the reference keeps the winning sign in a bool
        // (narrowphase.cpp:826), so -1.f is stored as `true` and the swap
        // below never triggers; kept for identical results.
        bool max_tri_sign = false;
        for (CountT i = 1; i < num_contacts; i++) {
            Vector3 cur_contact = contacts[i];
            Vector3 bc = cur_contact - manifold.contactPoints[1];
            float signed_area = contact_normal.dot(cross(ba, bc));
            float area = copysignf(signed_area, 1.f);

            if (area > max_tri_area) {
                max_tri_area = area;
                max_tri_sign = copysignf(1.f, signed_area) != 0.f;

                manifold.contactPoints[2] = cur_contact;
                manifold.penetrationDepths[2] = penetration_depths[i];
            }
        }

        if ((float)max_tri_sign == -1.f) {
            ba = -ba;
            Vector3 tmp = manifold.contactPoints[0];
            manifold.contactPoints[0] = manifold.contactPoints[1];
            manifold.contactPoints[1] = tmp;
        }

        Vector3 cb = manifold.contactPoints[2] - manifold.contactPoints[1];
        Vector3 ac = manifold.contactPoints[0] - manifold.contactPoints[2];

        float most_neg_area = 0.f;
        for (CountT i = 1; i < num_contacts; i++) {
            Vector3 cur_contact = contacts[i];

            Vector3 aq = manifold.contactPoints[0] - cur_contact;
            Vector3 qc = cur_contact - manifold.contactPoints[2];

            float abq_area = contact_normal.dot(cross(ba, aq));
            float bcq_area = contact_normal.dot(cross(cb, qc));
            float caq_area = contact_normal.dot(cross(aq, ac));

            float q_min_area = fminf(abq_area, fminf(bcq_area, caq_area));
            if (q_min_area < most_neg_area) {
                most_neg_area = q_min_area;
                manifold.contactPoints[3] = cur_contact;
                manifold.penetrationDepths[3] = penetration_depths[i];
            }
        }

        if (max_dist_sq == 0.f || max_tri_area == 0.f || most_neg_area == 0.f) {
            manifold.numContactPoints = 0;
            manifold.normal = Vector3::zero();
            return manifold;
        }
    }

    for (int i = 0; i < 4; i++) {
        if (i < manifold.numContactPoints) {
            manifold.contactPoints[i] =
                to_world_frame.rotateVec(manifold.contactPoints[i]) +
                world_offset;
        }
    }

    manifold.normal = to_world_frame.rotateVec(contact_normal);
    return manifold;
}

// Clips the incident face against the side planes of the reference face and
// keeps what lies below the reference plane, projected onto it.
template <typename RefHull, typename OtherHull>
MADRONA_HD inline Manifold createFaceContact(Plane ref_plane,
                                             int32_t ref_face_idx,
                                             int32_t incident_face_idx,
                                             const RefHull &ref,
                                             const OtherHull &other,
                                             void *tmp_buf1, void *tmp_buf2,
                                             Vector3 world_offset,
                                             Quat to_world_frame)
{
    Vector3 *clipping_input = (Vector3 *)tmp_buf1;
    Vector3 *clipping_dst = (Vector3 *)tmp_buf2;

    CountT num_clipped_vertices = 0;
    {
        uint32_t hedge_idx = other.faceBaseHedge(incident_face_idx);
        const uint32_t start_hedge_idx = hedge_idx;
        do {
            const HalfEdge cur_hedge = other.hedge(hedge_idx);
            hedge_idx = cur_hedge.next;
            clipping_input[num_clipped_vertices++] =
                other.vertex(cur_hedge.rootVertex);
        } while (hedge_idx != start_hedge_idx);
    }

    {
        uint32_t hedge_idx = ref.faceBaseHedge(ref_face_idx);
        const uint32_t start_hedge_idx = hedge_idx;

        HalfEdge cur_hedge = ref.hedge(hedge_idx);
        Vector3 cur_point = ref.vertex(cur_hedge.rootVertex);
        do {
            hedge_idx = cur_hedge.next;
            cur_hedge = ref.hedge(hedge_idx);
            Vector3 next_point = ref.vertex(cur_hedge.rootVertex);

            Vector3 edge = next_point - cur_point;
            Vector3 plane_normal = cross(edge, ref_plane.normal);
            Plane side_plane { plane_normal, dot(plane_normal, cur_point) };
            cur_point = next_point;

            num_clipped_vertices = clipPolygon(clipping_dst, side_plane,
                clipping_input, num_clipped_vertices);

            Vector3 *tmp = clipping_dst;
            clipping_dst = clipping_input;
            clipping_input = tmp;
        } while (hedge_idx != start_hedge_idx);
    }

    // the free buffer now holds the depths
    float *penetration_depths = (float *)clipping_dst;

    CountT num_below_plane = 0;
    for (CountT i = 0; i < num_clipped_vertices; ++i) {
        Vector3 vertex = clipping_input[i];
        float d = getDistanceFromPlane(ref_plane, vertex);
        if (d <= 0.0f) {
            clipping_input[num_below_plane] = vertex - d * ref_plane.normal;
            penetration_depths[num_below_plane] = -d;
            num_below_plane += 1;
        }
    }

    return buildFaceContactManifold(ref_plane.normal, clipping_input,
        penetration_depths, num_below_plane, world_offset, to_world_frame);
}

template <typename HullT>
MADRONA_HD inline Manifold createFacePlaneContact(Plane plane,
                                                  int32_t incident_face_idx,
                                                  const HullT &h,
                                                  Vector3 *contacts_tmp,
                                                  float *penetration_depths_tmp,
                                                  Vector3 world_offset,
                                                  Quat to_world_frame)
{
    CountT num_incident_vertices = 0;

    uint32_t hedge_idx = h.faceBaseHedge(incident_face_idx);
    const uint32_t start_hedge_idx = hedge_idx;
    do {
        const HalfEdge cur_hedge = h.hedge(hedge_idx);
        hedge_idx = cur_hedge.next;
        Vector3 vertex = h.vertex(cur_hedge.rootVertex);

        float d = getDistanceFromPlane(plane, vertex);
        if (d <= 0.0f) {
            contacts_tmp[num_incident_vertices] = vertex - d * plane.normal;
            penetration_depths_tmp[num_incident_vertices] = -d;
            num_incident_vertices += 1;
        }
    } while (hedge_idx != start_hedge_idx);

    return buildFaceContactManifold(plane.normal, contacts_tmp,
        penetration_depths_tmp, num_incident_vertices, world_offset,
        to_world_frame);
}

MADRONA_HD inline Segment shortestSegmentBetween(const Segment &seg1,
                                                 const Segment &seg2)
{
    Vector3 v1 = seg1.p2 - seg1.p1;
    Vector3 v2 = seg2.p2 - seg2.p1;
    Vector3 v21 = seg2.p1 - seg1.p1;

    float dotv22 = v2.dot(v2);
    float dotv11 = v1.dot(v1);
    float dotv21 = v2.dot(v1);
    float dotv211 = v21.dot(v1);
    float dotv212 = v21.dot(v2);

    float denom = dotv21 * dotv21 - dotv22 * dotv11;

    float s, t;
    if (fabsf(denom) < 0.00001f) {
        s = 0.0f;
        t = (dotv11 * s - dotv211) / dotv21;
    } else {
        s = (dotv212 * dotv21 - dotv22 * dotv211) / denom;
        t = (-dotv211 * dotv21 + dotv11 * dotv212) / denom;
    }

    s = fmaxf(fminf(s, 1.0f), 0.0f);
    t = fmaxf(fminf(t, 1.0f), 0.0f);

    return Segment { seg1.p1 + s * v1, seg2.p1 + t * v2 };
}

template <typename HullA, typename HullB>
MADRONA_HD inline Manifold createEdgeContact(Vector3 normal,
                                             float separation,
                                             int32_t hedge_idx_a,
                                             int32_t hedge_idx_b,
                                             const HullA &a,
                                             const HullB &b,
                                             Vector3 world_offset,
                                             Quat to_world_frame)
{
    Segment seg_a = getEdgeSegment(a, a.hedge(hedge_idx_a));
    Segment seg_b = getEdgeSegment(b, b.hedge(hedge_idx_b));

    Segment s = shortestSegmentBetween(seg_a, seg_b);
    Vector3 contact = s.p1;

    Manifold manifold;
    for (int i = 0; i < 4; i++) {
        manifold.contactPoints[i] = Vector3::zero();
        manifold.penetrationDepths[i] = 0.f;
    }
    manifold.contactPoints[0] = to_world_frame.rotateVec(contact) + world_offset;
    manifold.penetrationDepths[0] = -separation;
    manifold.numContactPoints = 1;
    manifold.normal = to_world_frame.rotateVec(normal);
    return manifold;
}

}
}

// Host-side baking of collision assets into the arrays ObjectManager points at.
//
// API contract: reference include/madrona/physics_assets.hpp:12-65
// (SourceCollisionPrimitive, SourceCollisionObject, RigidBodyAssets::
// processRigidBodyAssets).  Arithmetic follows src/physics/physics_assets.cpp
// (Newell planes :211-253, half-edge construction :638-747, mass properties
// :956-1134, inertia diagonalisation :802-953) operation for operation: the
// baked floats feed the solver, so they must equal the CPU oracle's.
//
// Not built: the quickhull path (build_convex_hulls == true, :281-554) -- input
// meshes must already be convex with coplanar faces merged.
#pragma once

#include <madrona/physics.hpp>
#include <madrona/importer.hpp>
#include <madrona/stack_alloc.hpp>
#include <madrona/geo.hpp>

#include <cstdlib>
#include <cstring>
#include <cmath>
#include <unordered_map>

namespace madrona::phys {

struct SourceCollisionPrimitive {
    struct HullInput {
        uint32_t hullIDX;
    };

    CollisionPrimitive::Type type;
    union {
        CollisionPrimitive::Sphere sphere;
        CollisionPrimitive::Plane plane;
        HullInput hullInput;
    };
};

struct SourceCollisionObject {
    Span<const SourceCollisionPrimitive> prims;
    float invMass;
    RigidBodyFrictionData friction;
};

struct RigidBodyAssets {
    struct HullData {
        geo::HalfEdge *halfEdges;
        uint32_t *faceBaseHalfEdges;
        geo::Plane *facePlanes;
        math::Vector3 *vertices;

        uint32_t numHalfEdges;
        uint32_t numFaces;
        uint32_t numVerts;
    } hullData;

    // per primitive
    CollisionPrimitive *primitives;
    math::AABB *primitiveAABBs;

    // per object
    RigidBodyMetadata *metadatas;
    math::AABB *objAABBs;
    uint32_t *primOffsets;
    uint32_t *primCounts;

    uint32_t numConvexHulls;
    uint32_t totalNumPrimitives;
    uint32_t numObjs;

    // Returns the malloc'd buffer every pointer of *out_assets points into
    // (caller frees), nullptr on failure.
    static inline void *processRigidBodyAssets(
        Span<const imp::SourceMesh> convex_hull_meshes,
        Span<const SourceCollisionObject> collision_objs,
        bool build_convex_hulls,
        StackAlloc &tmp_alloc,
        RigidBodyAssets *out_assets,
        CountT *out_num_bytes);
};

namespace assets_impl {

using math::Vector3;
using math::Quat;
using math::Diag3x3;
using math::Mat3x3;
using math::Symmetric3x3;
using math::AABB;
using geo::HalfEdge;
using geo::HalfEdgeMesh;
using geo::Plane;

struct MassProperties {
    Diag3x3 inertiaTensor;
    Vector3 centerOfMass;
    Quat toDiagonal;
};

// Newell's method: normal from the projected areas, centroid as plane point
inline Plane newellPlane(const Vector3 *verts, const uint32_t *indices,
                         CountT num_indices)
{
    Vector3 centroid { 0, 0, 0 };
    Vector3 n { 0, 0, 0 };

    CountT num_verts = 0;
    for (CountT i = num_indices - 1, j = 0; j < num_indices; i = j, j++) {
        Vector3 vi = verts[indices[i]];
        Vector3 vj = verts[indices[j]];

        n.x += (vi.y - vj.y) * (vi.z + vj.z);
        n.y += (vi.z - vj.z) * (vi.x + vj.x);
        n.z += (vi.x - vj.x) * (vi.y + vj.y);

        centroid += vj;
        num_verts += 1;
    }

    centroid /= (float)num_verts;

    n = normalize(n);
    return Plane { n, dot(centroid, n) };
}

// Half-edge ids are handed out in pairs (edge, twin) in first-seen order.
inline HalfEdgeMesh buildHalfEdgeMesh(StackAlloc &tmp_alloc,
                                      const imp::SourceMesh &src_mesh)
{
    auto num_face_verts = [&src_mesh](CountT face_idx) -> uint32_t {
        return src_mesh.faceCounts == nullptr ?
            3u : src_mesh.faceCounts[face_idx];
    };

    uint32_t num_hedges = 0;
    for (CountT f = 0; f < (CountT)src_mesh.numFaces; f++) {
        num_hedges += num_face_verts(f);
    }

    HalfEdge *hedges_out = tmp_alloc.allocN<HalfEdge>(num_hedges);
    uint32_t *face_base_hedges_out =
        tmp_alloc.allocN<uint32_t>(src_mesh.numFaces);
    Plane *face_planes_out = tmp_alloc.allocN<Plane>(src_mesh.numFaces);

    std::unordered_map<uint64_t, uint32_t> edge_to_hedge;
    auto edge_id = [](uint32_t a, uint32_t b) {
        return ((uint64_t)a << 32) | (uint64_t)b;
    };

    uint32_t num_assigned = 0;
    const uint32_t *face_indices = src_mesh.indices;
    for (CountT face_idx = 0; face_idx < (CountT)src_mesh.numFaces;
         face_idx++) {
        CountT n = num_face_verts(face_idx);

        face_planes_out[face_idx] =
            newellPlane(src_mesh.positions, face_indices, n);

        for (CountT k = 0; k < n; k++) {
            uint32_t a_idx = face_indices[k];
            uint32_t b_idx = face_indices[(k + 1) % n];
            uint32_t c_idx = face_indices[(k + 2) % n];

            auto found = edge_to_hedge.find(edge_id(a_idx, b_idx));
            uint32_t hedge_idx;
            if (found == edge_to_hedge.end()) {
                hedge_idx = num_assigned;
                edge_to_hedge.emplace(edge_id(a_idx, b_idx), num_assigned);
                edge_to_hedge.emplace(edge_id(b_idx, a_idx), num_assigned + 1);
                num_assigned += 2;
            } else {
                hedge_idx = found->second;
            }

            if (k == 0) {
                face_base_hedges_out[face_idx] = hedge_idx;
            }

            // an edge not seen yet will receive the next free id
            auto next_found = edge_to_hedge.find(edge_id(b_idx, c_idx));
            uint32_t next_hedge_idx = next_found == edge_to_hedge.end() ?
                num_assigned : next_found->second;

            hedges_out[hedge_idx] =
                HalfEdge { next_hedge_idx, a_idx, (uint32_t)face_idx };
        }

        face_indices += n;
    }

    if (num_assigned != num_hedges) {
        FATAL("physics assets: hull mesh is not a closed manifold");
    }

    HalfEdgeMesh mesh;
    mesh.halfEdges = hedges_out;
    mesh.faceBaseHalfEdges = face_base_hedges_out;
    mesh.facePlanes = face_planes_out;
    mesh.vertices = src_mesh.positions;
    mesh.numHalfEdges = num_hedges;
    mesh.numFaces = src_mesh.numFaces;
    mesh.numVertices = src_mesh.numVertices;
    return mesh;
}

// ---- inertia tensor diagonalisation (McAdams et al. 2011, Jacobi sweeps with
// approximate Givens quaternions) ---------------------------------------------
struct GivensPair {
    float ch;
    float sh;
};

inline GivensPair approxGivensQuaternion(Symmetric3x3 m)
{
    constexpr float gamma = 5.82842712474619f;
    constexpr float c_star = 0.9238795325112867f;
    constexpr float s_star = 0.3826834323650898f;

    float a11 = m.diag[0], a12 = m.off[0], a22 = m.diag[1];

    float ch = 2.f * (a11 - a22);
    float sh = a12;

    float sh2 = sh * sh;

    // already diagonal: identity rotation
    if (sh2 < 1e-20f) {
        return { 1.f, 0.f };
    }

    float ch2 = ch * ch;

    bool b = (gamma * sh2) < ch2;

    float omega = math::rsqrtApprox(ch2 + sh2);

    ch = b ? (omega * ch) : c_star;
    sh = b ? (omega * sh) : s_star;

    return { ch, sh };
}

// Q^T m Q for the (unnormalised) quaternion (ch, 0, 0, sh)
inline Symmetric3x3 jacobiIterConjugation(Symmetric3x3 m, float ch, float sh)
{
    float ch2 = ch * ch;
    float sh2 = sh * sh;
    float q_scale = ch2 + sh2;

    float q11 = (ch2 - sh2) / q_scale;
    float q12 = (-2.f * sh * ch) / q_scale;
    float q21 = (2.f * sh * ch) / q_scale;
    float q22 = (ch2 - sh2) / q_scale;

    float m11 = m.diag.x, m22 = m.diag.y, m33 = m.diag.z;
    float m12 = m.off.x, m13 = m.off.y, m23 = m.off.z;

    float m11q11_m12q21 = m11 * q11 + m12 * q21;
    float m11q12_m12q22 = m11 * q12 + m12 * q22;

    float m12q11_m22q21 = m12 * q11 + m22 * q21;
    float m12q12_m22q22 = m12 * q12 + m22 * q22;

    Symmetric3x3 out;
    out.diag = Vector3 {
        q11 * m11q11_m12q21 + q21 * m12q11_m22q21,
        q12 * m11q12_m12q22 + q22 * m12q12_m22q22,
        m33,
    };
    out.off = Vector3 {
        q12 * m11q11_m12q21 + q22 * m12q11_m22q21,
        m13 * q11 + m23 * q21,
        m13 * q12 + m23 * q22,
    };
    return out;
}

inline void swapf(float &a, float &b) { float t = a; a = b; b = t; }

inline void diagonalizeInertiaTensor(const Symmetric3x3 &m,
                                     Diag3x3 *out_diag, Quat *out_rot)
{
    constexpr CountT num_jacobi_iters = 8;

    Symmetric3x3 cur_mat = m;
    Quat accumulated_rot { 1, 0, 0, 0 };
    for (CountT i = 0; i < num_jacobi_iters; i++) {
        // pairs (1,2), (1,3), (2,3); the matrix is permuted so the pair being
        // rotated always sits in the upper-left 2x2
        GivensPair g1 = approxGivensQuaternion(cur_mat);
        cur_mat = jacobiIterConjugation(cur_mat, g1.ch, g1.sh);

        swapf(cur_mat.diag[1], cur_mat.diag[2]);
        swapf(cur_mat.off[0], cur_mat.off[1]);

        GivensPair g2 = approxGivensQuaternion(cur_mat);
        cur_mat = jacobiIterConjugation(cur_mat, g2.ch, g2.sh);

        swapf(cur_mat.diag[0], cur_mat.diag[2]);
        swapf(cur_mat.off[0], cur_mat.off[2]);

        GivensPair g3 = approxGivensQuaternion(cur_mat);
        cur_mat = jacobiIterConjugation(cur_mat, g3.ch, g3.sh);

        Symmetric3x3 restored;
        restored.diag =
            Vector3 { cur_mat.diag[2], cur_mat.diag[0], cur_mat.diag[1] };
        restored.off =
            Vector3 { cur_mat.off[1], cur_mat.off[2], cur_mat.off[0] };
        cur_mat = restored;

        accumulated_rot = Quat { g1.ch, 0, 0, g1.sh } *
            Quat { g2.ch, 0, g2.sh, 0 } * Quat { g3.ch, g3.sh, 0, 0 } *
            accumulated_rot;
    }

    Quat final_rot = accumulated_rot.normalize();

    {
        Mat3x3 q = Mat3x3::fromQuat(final_rot);

        float m11 = m.diag.x, m22 = m.diag.y, m33 = m.diag.z;
        float m12 = m.off.x, m13 = m.off.y, m23 = m.off.z;

        float q11 = q[0].x, q21 = q[0].y, q31 = q[0].z;
        float q12 = q[1].x, q22 = q[1].y, q32 = q[1].z;
        float q13 = q[2].x, q23 = q[2].y, q33 = q[2].z;

        out_diag->d0 = q11 * (m11 * q11 + m12 * q21 + m13 * q31) +
                       q21 * (m12 * q11 + m22 * q21 + m23 * q31) +
                       q31 * (m13 * q11 + m23 * q21 + m33 * q31);

        out_diag->d1 = q12 * (m11 * q12 + m12 * q22 + m13 * q32) +
                       q22 * (m12 * q12 + m22 * q22 + m23 * q32) +
                       q32 * (m13 * q12 + m23 * q22 + m33 * q32);

        out_diag->d2 = q13 * (m11 * q13 + m12 * q23 + m13 * q33) +
                       q23 * (m12 * q13 + m22 * q23 + m23 * q33) +
                       q33 * (m13 * q13 + m23 * q23 + m33 * q33);
    }

    *out_rot = final_rot;
}

// Mass, centre of mass and inertia of a union of primitives by summing the
// covariance of the tetrahedra (origin, face fan) of every hull
// (Blow & Binstock, "How to find the inertia tensor").
inline MassProperties computeMassProperties(
    const HalfEdgeMesh *convex_hulls,
    const SourceCollisionObject &src_obj)
{
    Symmetric3x3 C_canonical;
    C_canonical.diag = Vector3 { 1.f / 60.f, 1.f / 60.f, 1.f / 60.f };
    C_canonical.off = Vector3 { 1.f / 120.f, 1.f / 120.f, 1.f / 120.f };
    constexpr float density = 1.f;

    Symmetric3x3 C_total;
    C_total.diag = Vector3::zero();
    C_total.off = Vector3::zero();

    float m_total = 0;
    Vector3 x_total = Vector3::zero();

    auto process_tet = [&](Vector3 e1, Vector3 e2, Vector3 e3) {
        Mat3x3 A {{ e1, e2, e3 }};
        float det_A = A.determinant();
        Symmetric3x3 C = det_A * Symmetric3x3::AXAT(A, C_canonical);

        float volume = 1.f / 6.f * det_A;
        float m = volume * density;

        Vector3 x = 0.25f * e1 + 0.25f * e2 + 0.25f * e3;

        float old_m_total = m_total;
        m_total += m;
        x_total = (x * m + x_total * old_m_total) / m_total;

        C_total += C;
    };

    for (const SourceCollisionPrimitive &prim : src_obj.prims) {
        if (prim.type == CollisionPrimitive::Type::Sphere) {
            m_total += 1.f;

            float r = prim.sphere.radius;

            // covariance, not inertia: half the textbook 2/5 r^2
            float v = 1.f / 5.f * r * r;
            Symmetric3x3 C_sphere;
            C_sphere.diag = Vector3 { v, v, v };
            C_sphere.off = Vector3::zero();
            C_total += C_sphere;
            continue;
        } else if (prim.type == CollisionPrimitive::Type::Plane) {
            // infinite mass and inertia for the whole object
            return MassProperties {
                Diag3x3::uniform(INFINITY),
                Vector3::zero(),
                Quat { 1, 0, 0, 0 },
            };
        }

        const HalfEdgeMesh &convex_hull = convex_hulls[prim.hullInput.hullIDX];

        for (CountT face_idx = 0; face_idx < (CountT)convex_hull.numFaces;
             face_idx++) {
            uint32_t root_hedge_idx = convex_hull.faceBaseHalfEdges[face_idx];
            HalfEdge root_hedge = convex_hull.halfEdges[root_hedge_idx];
            Vector3 v1 = convex_hull.vertices[root_hedge.rootVertex];
            uint32_t cur_hedge_idx = root_hedge.next;

            while (true) {
                HalfEdge cur_hedge = convex_hull.halfEdges[cur_hedge_idx];
                uint32_t next_hedge_idx = cur_hedge.next;
                if (next_hedge_idx == root_hedge_idx) {
                    break;
                }

                HalfEdge next_hedge = convex_hull.halfEdges[next_hedge_idx];

                Vector3 v2 = convex_hull.vertices[cur_hedge.rootVertex];
                Vector3 v3 = convex_hull.vertices[next_hedge.rootVertex];

                process_tet(v1, v2, v3);

                cur_hedge_idx = next_hedge_idx;
            }
        }
    }

    // move the covariance to the centre of mass
    {
        Vector3 x = x_total;
        Vector3 delta_x = -x_total;

        Symmetric3x3 cross_terms;
        cross_terms.diag = 2.f * Vector3 {
            x.x * delta_x.x,
            x.y * delta_x.y,
            x.z * delta_x.z,
        };
        cross_terms.off = Vector3 {
            x.x * delta_x.y + x.y * delta_x.x,
            x.x * delta_x.z + x.z * delta_x.x,
            x.y * delta_x.z + x.z * delta_x.y,
        };

        Symmetric3x3 delta_sq = Symmetric3x3::vvT(delta_x);
        C_total = C_total + m_total * (cross_terms + delta_sq);
    }

    float tr_C = C_total[0][0] + C_total[1][1] + C_total[2][2];
    Symmetric3x3 tr_C_diag;
    tr_C_diag.diag = Vector3 { tr_C, tr_C, tr_C };
    tr_C_diag.off = Vector3::zero();

    Symmetric3x3 inertia_tensor = tr_C_diag - C_total;

    float inv_mass = 1.f / m_total;
    inertia_tensor *= inv_mass;

    Diag3x3 diag_inertia;
    Quat rot_to_diag;
    diagonalizeInertiaTensor(inertia_tensor, &diag_inertia, &rot_to_diag);

    return MassProperties { diag_inertia, x_total, rot_to_diag };
}

inline RigidBodyMassData toMassData(const MassProperties &mass_props,
                                    float inv_m)
{
    Diag3x3 inv_inertia = inv_m / mass_props.inertiaTensor;

    RigidBodyMassData out;
    out.invMass = inv_m;
    out.invInertiaTensor =
        Vector3 { inv_inertia.d0, inv_inertia.d1, inv_inertia.d2 };
    out.toCenterOfMass = mass_props.centerOfMass;
    out.toInteriaFrame = mass_props.toDiagonal;
    return out;
}

inline void setupPrimitives(const HalfEdgeMesh *hull_meshes,
                            Span<const SourceCollisionObject> collision_objs,
                            CollisionPrimitive *out_prims,
                            AABB *out_prim_aabbs,
                            AABB *out_obj_aabbs,
                            uint32_t *out_prim_offsets,
                            uint32_t *out_prim_counts)
{
    using Type = CollisionPrimitive::Type;

    uint32_t cur_prim_offset = 0;
    for (CountT obj_idx = 0; obj_idx < collision_objs.size(); obj_idx++) {
        const SourceCollisionObject &collision_obj = collision_objs[obj_idx];
        CountT num_prims = collision_obj.prims.size();

        AABB obj_aabb = AABB::invalid();

        for (CountT prim_idx = 0; prim_idx < num_prims; prim_idx++) {
            const SourceCollisionPrimitive &src_prim =
                collision_obj.prims[prim_idx];

            CollisionPrimitive *out_prim =
                &out_prims[cur_prim_offset + prim_idx];
            memset((void *)out_prim, 0, sizeof(CollisionPrimitive));
            out_prim->type = src_prim.type;

            AABB prim_aabb;
            switch (src_prim.type) {
            case Type::Sphere: {
                out_prim->sphere = src_prim.sphere;
                const float r = src_prim.sphere.radius;
                prim_aabb = AABB { { -r, -r, -r }, { r, r, r } };
            } break;
            case Type::Plane: {
                out_prim->plane = CollisionPrimitive::Plane {};
                prim_aabb = AABB {
                    { -FLT_MAX, -FLT_MAX, -FLT_MAX },
                    { FLT_MAX, FLT_MAX, 0 },
                };
            } break;
            case Type::Hull: {
                const HalfEdgeMesh &hull_mesh =
                    hull_meshes[src_prim.hullInput.hullIDX];

                prim_aabb = AABB::point(hull_mesh.vertices[0]);
                for (CountT v = 1; v < (CountT)hull_mesh.numVertices; v++) {
                    prim_aabb.expand(hull_mesh.vertices[v]);
                }

                out_prim->hull.halfEdgeMesh = hull_mesh;
            } break;
            }

            out_prim_aabbs[cur_prim_offset + prim_idx] = prim_aabb;
            obj_aabb = AABB::merge(obj_aabb, prim_aabb);
        }

        out_obj_aabbs[obj_idx] = obj_aabb;
        out_prim_offsets[obj_idx] = cur_prim_offset;
        out_prim_counts[obj_idx] = (uint32_t)num_prims;

        cur_prim_offset += (uint32_t)num_prims;
    }
}

inline size_t alignUp64(size_t v) { return (v + 63) & ~size_t(63); }

}

inline void *RigidBodyAssets::processRigidBodyAssets(
    Span<const imp::SourceMesh> convex_hull_meshes,
    Span<const SourceCollisionObject> collision_objs,
    bool build_convex_hulls,
    StackAlloc &tmp_alloc,
    RigidBodyAssets *out_assets,
    CountT *out_num_bytes)
{
    using namespace assets_impl;

    if (build_convex_hulls) {
        FATAL("madrona_amd physics assets: quickhull is not available, pass "
              "convex meshes with merged coplanar faces and "
              "build_convex_hulls = false");
    }

    auto tmp_frame = tmp_alloc.push();

    const CountT num_hulls = convex_hull_meshes.size();
    const CountT num_objs = collision_objs.size();

    HalfEdgeMesh *built_hulls =
        tmp_alloc.allocN<HalfEdgeMesh>(num_hulls > 0 ? num_hulls : 1);
    for (CountT i = 0; i < num_hulls; i++) {
        built_hulls[i] = buildHalfEdgeMesh(tmp_alloc, convex_hull_meshes[i]);
    }

    size_t total_prims = 0;
    for (CountT i = 0; i < num_objs; i++) {
        total_prims += (size_t)collision_objs[i].prims.size();
    }

    size_t total_hedges = 0, total_faces = 0, total_verts = 0;
    for (CountT i = 0; i < num_hulls; i++) {
        total_hedges += built_hulls[i].numHalfEdges;
        total_faces += built_hulls[i].numFaces;
        total_verts += built_hulls[i].numVertices;
    }

    // one buffer, 64-byte aligned sections
    const size_t section_bytes[10] = {
        sizeof(HalfEdge) * total_hedges,
        sizeof(uint32_t) * total_faces,
        sizeof(Plane) * total_faces,
        sizeof(Vector3) * total_verts,
        sizeof(CollisionPrimitive) * total_prims,
        sizeof(AABB) * total_prims,
        sizeof(RigidBodyMetadata) * (size_t)num_objs,
        sizeof(AABB) * (size_t)num_objs,
        sizeof(uint32_t) * (size_t)num_objs,
        sizeof(uint32_t) * (size_t)num_objs,
    };
    size_t section_offsets[10];
    size_t num_buffer_bytes = 0;
    for (int i = 0; i < 10; i++) {
        section_offsets[i] = num_buffer_bytes;
        num_buffer_bytes = alignUp64(num_buffer_bytes + section_bytes[i]);
    }
    if (num_buffer_bytes == 0) {
        num_buffer_bytes = 64;
    }

    char *buffer = (char *)aligned_alloc(64, num_buffer_bytes);
    memset(buffer, 0, num_buffer_bytes);

    RigidBodyAssets assets;
    assets.hullData.halfEdges = (HalfEdge *)(buffer + section_offsets[0]);
    assets.hullData.faceBaseHalfEdges =
        (uint32_t *)(buffer + section_offsets[1]);
    assets.hullData.facePlanes = (Plane *)(buffer + section_offsets[2]);
    assets.hullData.vertices = (Vector3 *)(buffer + section_offsets[3]);
    assets.hullData.numHalfEdges = (uint32_t)total_hedges;
    assets.hullData.numFaces = (uint32_t)total_faces;
    assets.hullData.numVerts = (uint32_t)total_verts;
    assets.primitives = (CollisionPrimitive *)(buffer + section_offsets[4]);
    assets.primitiveAABBs = (AABB *)(buffer + section_offsets[5]);
    assets.metadatas = (RigidBodyMetadata *)(buffer + section_offsets[6]);
    assets.objAABBs = (AABB *)(buffer + section_offsets[7]);
    assets.primOffsets = (uint32_t *)(buffer + section_offsets[8]);
    assets.primCounts = (uint32_t *)(buffer + section_offsets[9]);
    assets.numConvexHulls = (uint32_t)num_hulls;
    assets.totalNumPrimitives = (uint32_t)total_prims;
    assets.numObjs = (uint32_t)num_objs;

    size_t hedge_offset = 0, face_offset = 0, vert_offset = 0;
    for (CountT i = 0; i < num_hulls; i++) {
        HalfEdgeMesh &hull_mesh = built_hulls[i];

        HalfEdge *he_out = &assets.hullData.halfEdges[hedge_offset];
        uint32_t *face_bases_out =
            &assets.hullData.faceBaseHalfEdges[face_offset];
        Plane *face_planes_out = &assets.hullData.facePlanes[face_offset];
        Vector3 *verts_out = &assets.hullData.vertices[vert_offset];

        memcpy(he_out, hull_mesh.halfEdges,
               sizeof(HalfEdge) * hull_mesh.numHalfEdges);
        memcpy(face_bases_out, hull_mesh.faceBaseHalfEdges,
               sizeof(uint32_t) * hull_mesh.numFaces);
        memcpy(face_planes_out, hull_mesh.facePlanes,
               sizeof(Plane) * hull_mesh.numFaces);
        memcpy((void *)verts_out, hull_mesh.vertices,
               sizeof(Vector3) * hull_mesh.numVertices);

        hull_mesh.halfEdges = he_out;
        hull_mesh.faceBaseHalfEdges = face_bases_out;
        hull_mesh.facePlanes = face_planes_out;
        hull_mesh.vertices = verts_out;

        hedge_offset += hull_mesh.numHalfEdges;
        face_offset += hull_mesh.numFaces;
        vert_offset += hull_mesh.numVertices;
    }

    setupPrimitives(built_hulls, collision_objs, assets.primitives,
                    assets.primitiveAABBs, assets.objAABBs,
                    assets.primOffsets, assets.primCounts);

    for (CountT i = 0; i < num_objs; i++) {
        const SourceCollisionObject &collision_obj = collision_objs[i];
        MassProperties mass_props =
            computeMassProperties(built_hulls, collision_obj);

        assets.metadatas[i].mass =
            toMassData(mass_props, collision_obj.invMass);
        assets.metadatas[i].friction = collision_obj.friction;
    }

    tmp_alloc.pop(tmp_frame);

    *out_assets = assets;
    *out_num_bytes = (CountT)num_buffer_bytes;
    return buffer;
}

}

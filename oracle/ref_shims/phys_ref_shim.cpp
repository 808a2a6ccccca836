// TEST INFRASTRUCTURE ONLY.  Exposes the reference's rigid-body asset baker and
// scalar narrowphase behind a C ABI so tests can compare madrona_amd's host /
// device-shared physics math with the reference's own code, function by
// function.  The narrowphase routines are file-static in the reference, so the
// translation unit is included where it lies (nothing is copied).
#include <madrona/physics.hpp>
#include <madrona/physics_assets.hpp>
#include <madrona/context.hpp>

#include "../../../reference/src/physics/physics_impl.hpp"

// runNarrowphase & friends are static: pull the TU in
#include "../../../reference/src/physics/narrowphase.cpp"

#include <cstring>
#include <vector>

using namespace madrona;
using namespace madrona::phys;

extern "C" {

#define API __attribute__((visibility("default")))

// Unit cube (quads) baked by the reference: returns the flat arrays tests diff.
// out_floats: [numObjs x (invMass, invInertia xyz, com xyz, rot wxyz, muS, muD)]
// then per primitive AABB (6), then hull planes (4 per face), then hull verts.
API int32_t ref_bake_objects(const float *verts, uint32_t num_verts,
                             const uint32_t *indices, const uint32_t *face_counts,
                             uint32_t num_faces,
                             const int32_t *obj_types,   // 0 hull, 1 plane, 2 sphere
                             const float *obj_inv_mass,
                             const float *obj_radius,
                             uint32_t num_objs,
                             float *out_floats, uint32_t max_floats,
                             uint32_t *out_hedges, uint32_t max_hedge_words)
{
    imp::SourceMesh mesh {};
    mesh.positions = (math::Vector3 *)verts;
    mesh.indices = (uint32_t *)indices;
    mesh.faceCounts = (uint32_t *)face_counts;
    mesh.numVertices = num_verts;
    mesh.numFaces = num_faces;

    std::vector<SourceCollisionPrimitive> prims(num_objs);
    std::vector<SourceCollisionObject> objs(num_objs);
    for (uint32_t i = 0; i < num_objs; i++) {
        if (obj_types[i] == 0) {
            prims[i].type = CollisionPrimitive::Type::Hull;
            prims[i].hullInput.hullIDX = 0;
        } else if (obj_types[i] == 1) {
            prims[i].type = CollisionPrimitive::Type::Plane;
        } else {
            prims[i].type = CollisionPrimitive::Type::Sphere;
            prims[i].sphere.radius = obj_radius[i];
        }
        objs[i] = SourceCollisionObject {
            Span<const SourceCollisionPrimitive>(&prims[i], 1),
            obj_inv_mass[i], { 0.5f, 0.75f },
        };
    }

    StackAlloc tmp_alloc;
    RigidBodyAssets assets;
    CountT num_bytes;
    void *buf = RigidBodyAssets::processRigidBodyAssets(
        Span<const imp::SourceMesh>(&mesh, 1),
        Span<const SourceCollisionObject>(objs.data(), (CountT)num_objs),
        false, tmp_alloc, &assets, &num_bytes);
    if (buf == nullptr) return -1;

    uint32_t n = 0;
    auto put = [&](float v) { if (n < max_floats) out_floats[n] = v; n++; };
    for (uint32_t i = 0; i < num_objs; i++) {
        const RigidBodyMetadata &m = assets.metadatas[i];
        put(m.mass.invMass);
        put(m.mass.invInertiaTensor.x); put(m.mass.invInertiaTensor.y);
        put(m.mass.invInertiaTensor.z);
        put(m.mass.toCenterOfMass.x); put(m.mass.toCenterOfMass.y);
        put(m.mass.toCenterOfMass.z);
        put(m.mass.toInteriaFrame.w); put(m.mass.toInteriaFrame.x);
        put(m.mass.toInteriaFrame.y); put(m.mass.toInteriaFrame.z);
        put(m.friction.muS); put(m.friction.muD);
    }
    for (uint32_t i = 0; i < assets.totalNumPrimitives; i++) {
        const math::AABB &a = assets.primitiveAABBs[i];
        put(a.pMin.x); put(a.pMin.y); put(a.pMin.z);
        put(a.pMax.x); put(a.pMax.y); put(a.pMax.z);
    }
    for (uint32_t i = 0; i < assets.hullData.numFaces; i++) {
        const geo::Plane &p = assets.hullData.facePlanes[i];
        put(p.normal.x); put(p.normal.y); put(p.normal.z); put(p.d);
    }
    for (uint32_t i = 0; i < assets.hullData.numVerts; i++) {
        const math::Vector3 &v = assets.hullData.vertices[i];
        put(v.x); put(v.y); put(v.z);
    }

    uint32_t h = 0;
    for (uint32_t i = 0; i < assets.hullData.numHalfEdges; i++) {
        const geo::HalfEdge &e = assets.hullData.halfEdges[i];
        if (h + 3 <= max_hedge_words) {
            out_hedges[h] = e.next; out_hedges[h + 1] = e.rootVertex;
            out_hedges[h + 2] = e.face;
        }
        h += 3;
    }
    for (uint32_t i = 0; i < assets.hullData.numFaces; i++) {
        if (h < max_hedge_words) out_hedges[h] = assets.hullData.faceBaseHalfEdges[i];
        h++;
    }

    free(buf);
    return (int32_t)n;
}

// One hull-hull / hull-plane pair through the reference's scalar narrowphase
// (narrowphaseDispatch + the manifold builders).  Transforms: pos(3) rot(wxyz)
// scale(3).  b_is_plane: b is the z-up plane primitive.  out: 27 floats:
// [has_contact, ref_is_a, numPoints, normal(3), points(4x4)] + pad.
API void ref_collide_pair(const float *verts, uint32_t num_verts,
                          const uint32_t *indices, const uint32_t *face_counts,
                          uint32_t num_faces,
                          const float *a_txfm, const float *b_txfm,
                          int32_t b_is_plane, float *out)
{
    using namespace narrowphase;

    imp::SourceMesh mesh {};
    mesh.positions = (math::Vector3 *)verts;
    mesh.indices = (uint32_t *)indices;
    mesh.faceCounts = (uint32_t *)face_counts;
    mesh.numVertices = num_verts;
    mesh.numFaces = num_faces;

    SourceCollisionPrimitive src_prims[3];
    src_prims[0].type = CollisionPrimitive::Type::Hull;
    src_prims[0].hullInput.hullIDX = 0;
    src_prims[1].type = CollisionPrimitive::Type::Plane;
    src_prims[2].type = CollisionPrimitive::Type::Sphere;
    src_prims[2].sphere.radius = 0.7f;
    SourceCollisionObject objs[3] = {
        { Span<const SourceCollisionPrimitive>(&src_prims[0], 1), 1.f, { 0.5f, 0.5f } },
        { Span<const SourceCollisionPrimitive>(&src_prims[1], 1), 0.f, { 0.5f, 0.5f } },
        { Span<const SourceCollisionPrimitive>(&src_prims[2], 1), 1.f, { 0.5f, 0.5f } },
    };

    StackAlloc tmp_alloc;
    RigidBodyAssets assets;
    CountT num_bytes;
    void *buf = RigidBodyAssets::processRigidBodyAssets(
        Span<const imp::SourceMesh>(&mesh, 1),
        Span<const SourceCollisionObject>(objs, 3),
        false, tmp_alloc, &assets, &num_bytes);

    // b_is_plane: 0 hull-hull, 1 hull-plane, 2 sphere (a) - hull (b)
    const CollisionPrimitive *a_prim = &assets.primitives[b_is_plane == 2 ? 2 : 0];
    const CollisionPrimitive *b_prim = &assets.primitives[b_is_plane == 1 ? 1 : 0];

    Vector3 a_pos { a_txfm[0], a_txfm[1], a_txfm[2] };
    Quat a_rot { a_txfm[3], a_txfm[4], a_txfm[5], a_txfm[6] };
    Diag3x3 a_scale { a_txfm[7], a_txfm[8], a_txfm[9] };
    Vector3 b_pos { b_txfm[0], b_txfm[1], b_txfm[2] };
    Quat b_rot { b_txfm[3], b_txfm[4], b_txfm[5], b_txfm[6] };
    Diag3x3 b_scale { b_txfm[7], b_txfm[8], b_txfm[9] };

    constexpr int32_t max_tmp = 512;
    static thread_local Plane tmp_faces[max_tmp];
    static thread_local Vector3 tmp_vertices[max_tmp];

    NarrowphaseTest test = b_is_plane == 1 ? NarrowphaseTest::HullPlane :
        b_is_plane == 2 ? NarrowphaseTest::SphereHull :
                          NarrowphaseTest::HullHull;
    NarrowphaseResult result = narrowphaseDispatch(
        test, a_pos, b_pos, a_rot, b_rot, a_scale, b_scale, a_prim, b_prim,
        max_tmp, max_tmp, tmp_vertices, tmp_faces);

    memset(out, 0, sizeof(float) * 28);
    void *tmp_a = tmp_faces;
    void *tmp_b = tmp_faces + max_tmp / 2;
    Manifold manifold {};
    bool has = false;
    float ref_is_a = 0.f;
    switch (result.type) {
    case ContactType::Sphere: {
        has = true;
        manifold.numContactPoints = 1;
        manifold.normal = result.sphere.normal;
        manifold.contactPoints[0] = result.sphere.pt;
        manifold.penetrationDepths[0] = result.sphere.depth;
    } break;
    case ContactType::SATPlane: {
        Plane plane { result.sat.normal, result.sat.planeDOrSeparation };
        manifold = createFacePlaneContact(plane,
            (int32_t)result.sat.incidentFaceIdxOrEdgeIdxB, result.aVertices,
            result.aHalfEdges, result.aFaceHedgeRoots, (Vector3 *)tmp_a,
            (float *)tmp_b, { 0, 0, 0 }, { 1, 0, 0, 0 });
        has = manifold.numContactPoints > 0;
    } break;
    case ContactType::SATFace: {
        uint32_t mask = result.sat.refFaceIdxOrEdgeIdxA;
        uint32_t ref_face = mask & 0x7FFFFFFFu;
        bool a_is_ref = ref_face == mask;
        Plane ref_plane { result.sat.normal, result.sat.planeDOrSeparation };
        manifold = a_is_ref ?
            createFaceContact(ref_plane, (int32_t)ref_face,
                (int32_t)result.sat.incidentFaceIdxOrEdgeIdxB,
                result.aVertices, result.bVertices, result.aHalfEdges,
                result.bHalfEdges, result.aFaceHedgeRoots,
                result.bFaceHedgeRoots, tmp_a, tmp_b,
                { 0, 0, 0 }, { 1, 0, 0, 0 }) :
            createFaceContact(ref_plane, (int32_t)ref_face,
                (int32_t)result.sat.incidentFaceIdxOrEdgeIdxB,
                result.bVertices, result.aVertices, result.bHalfEdges,
                result.aHalfEdges, result.bFaceHedgeRoots,
                result.aFaceHedgeRoots, tmp_a, tmp_b,
                { 0, 0, 0 }, { 1, 0, 0, 0 });
        has = manifold.numContactPoints > 0;
        ref_is_a = a_is_ref ? 1.f : 0.f;
    } break;
    case ContactType::SATEdge: {
        manifold = createEdgeContact(result.sat.normal,
            result.sat.planeDOrSeparation,
            (int32_t)result.sat.refFaceIdxOrEdgeIdxA,
            (int32_t)result.sat.incidentFaceIdxOrEdgeIdxB,
            result.aVertices, result.bVertices, result.aHalfEdges,
            result.bHalfEdges, { 0, 0, 0 }, { 1, 0, 0, 0 });
        has = true;
        ref_is_a = 1.f;
    } break;
    default: break;
    }

    if (has) {
        out[0] = 1.f;
        out[1] = ref_is_a;
        out[2] = (float)manifold.numContactPoints;
        out[3] = manifold.normal.x; out[4] = manifold.normal.y;
        out[5] = manifold.normal.z;
        for (int i = 0; i < 4; i++) {
            out[6 + i * 4 + 0] = manifold.contactPoints[i].x;
            out[6 + i * 4 + 1] = manifold.contactPoints[i].y;
            out[6 + i * 4 + 2] = manifold.contactPoints[i].z;
            out[6 + i * 4 + 3] = manifold.penetrationDepths[i];
        }
    }

    free(buf);
}

}

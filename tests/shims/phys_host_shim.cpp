// Host build of madrona_amd's physics math (the same headers the device code
// is compiled from) behind the C ABI of oracle/ref_shims/phys_ref_shim.cpp, so
// tests can diff the two function by function without a GPU.
#include <madrona/physics.hpp>
#include <madrona/physics_assets.hpp>

#include <cstring>
#include <vector>

using namespace madrona;
using namespace madrona::phys;

extern "C" {

#define API __attribute__((visibility("default")))

API int32_t amd_bake_objects(const float *verts, uint32_t num_verts,
                             const uint32_t *indices, const uint32_t *face_counts,
                             uint32_t num_faces,
                             const int32_t *obj_types,
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

// mode 0: hulls stored in scratch (collidePairStored); mode 1: LazyHull
API void amd_collide_pair(const float *verts, uint32_t num_verts,
                          const uint32_t *indices, const uint32_t *face_counts,
                          uint32_t num_faces,
                          const float *a_txfm, const float *b_txfm,
                          int32_t b_is_plane, int32_t mode, float *out)
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
    PairSetup pair {};
    pair.aLoc = Loc { 1, 0 };
    pair.bLoc = Loc { 2, 0 };
    pair.aPrim = &assets.primitives[b_is_plane == 2 ? 2 : 0];
    pair.bPrim = &assets.primitives[b_is_plane == 1 ? 1 : 0];
    pair.a = PrimitiveTransform {
        { a_txfm[0], a_txfm[1], a_txfm[2] },
        { a_txfm[3], a_txfm[4], a_txfm[5], a_txfm[6] },
        { a_txfm[7], a_txfm[8], a_txfm[9] } };
    pair.b = PrimitiveTransform {
        { b_txfm[0], b_txfm[1], b_txfm[2] },
        { b_txfm[3], b_txfm[4], b_txfm[5], b_txfm[6] },
        { b_txfm[7], b_txfm[8], b_txfm[9] } };
    pair.test = b_is_plane == 1 ? NarrowphaseTest::HullPlane :
        b_is_plane == 2 ? NarrowphaseTest::SphereHull :
                          NarrowphaseTest::HullHull;
    pair.aabbOverlap = true;

    constexpr int32_t max_tmp = 128;
    geo::Plane tmp_faces[max_tmp];
    math::Vector3 tmp_vertices[max_tmp];

    ContactConstraint contact {};
    bool unsupported = false;
    bool has;
    if (mode == 0) {
        has = collidePairStored(pair, tmp_vertices, tmp_faces, max_tmp,
                                &contact, &unsupported);
    } else if (b_is_plane == 2) {
        LazyHull b(pair.bPrim->hull.halfEdgeMesh, pair.b.pos - pair.a.pos,
                   pair.b.rot, pair.b.scale, false);
        has = sphereHullContact(pair, b, &contact);
    } else if (b_is_plane == 1) {
        LazyHull a(pair.aPrim->hull.halfEdgeMesh, pair.a.pos, pair.a.rot,
                   pair.a.scale, false);
        has = hullPlaneContact(a, pair.b, pair.aLoc, pair.bLoc, tmp_faces,
                               tmp_faces + max_tmp / 2, &contact);
    } else {
        LazyHull a(pair.aPrim->hull.halfEdgeMesh, pair.a.pos, pair.a.rot,
                   pair.a.scale);
        LazyHull b(pair.bPrim->hull.halfEdgeMesh, pair.b.pos, pair.b.rot,
                   pair.b.scale);
        SATResult sat = doSAT(a, b);
        has = satToContact(sat, a, b, pair.aLoc, pair.bLoc, tmp_faces,
                           tmp_faces + max_tmp / 2, &contact);
    }

    memset(out, 0, sizeof(float) * 28);
    if (has) {
        out[0] = 1.f;
        // hull-plane: the plane (b) is the reference; report as the ref shim does
        out[1] = b_is_plane != 0 ? 0.f :
            (contact.ref.archetype == 1 ? 1.f : 0.f);
        out[2] = (float)contact.numPoints;
        out[3] = contact.normal.x; out[4] = contact.normal.y;
        out[5] = contact.normal.z;
        for (int i = 0; i < 4; i++) {
            out[6 + i * 4 + 0] = contact.points[i].x;
            out[6 + i * 4 + 1] = contact.points[i].y;
            out[6 + i * 4 + 2] = contact.points[i].z;
            out[6 + i * 4 + 3] = contact.points[i].w;
        }
    }
    (void)unsupported;

    free(buf);
}

}

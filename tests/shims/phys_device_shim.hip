// TEST INFRASTRUCTURE.  madrona_amd's narrowphase code paths run ON THE DEVICE,
// pair by pair, behind the C ABI of the host shim (phys_host_shim.cpp), so the
// GPU tests can diff them function by function against the reference's
// narrowphase (oracle/_ref/libphys_ref.so):
//   mode 0  collidePairStored   one lane per pair, hulls in private scratch
//                               (the > 128-body kernel, and the fallback of
//                               the LDS kernel for big faces)
//   mode 1  collidePairLane     one lane per pair, LazyHull + an LDS row (the
//                               LDS kernel's plane / sphere path)
//   mode 2  hullHullWave        one wavefront per pair (the LDS kernel's
//                               cooperative hull-hull SAT)
//   mode 3  hullHullWave<32>    two pairs per wavefront, 32 lanes each
//   mode 4  hullHullWave<16>    four pairs per wavefront, 16 lanes each (what
//                               the two-worlds-per-wavefront kernel runs)
#include <madrona/mwhip/user_prelude.hpp>
// (the physics headers are written like simulator sources: unannotated
// definitions, made host + device by the wrapper's pragma -- madrona_amd/Makefile)
#pragma clang force_cuda_host_device begin
#include <madrona/physics.hpp>
#include <madrona/physics_assets.hpp>
#include <madrona/physics_loader.hpp>
#pragma clang force_cuda_host_device end

#include <vector>

using namespace madrona;
using namespace madrona::phys;
using namespace madrona::phys::narrowphase;

namespace {

struct PairIn {
    float a[10];    // pos xyz, rot wxyz, scale xyz
    float b[10];
};

__device__ inline PrimitiveTransform loadTxfm(const float *t)
{
    return PrimitiveTransform {
        { t[0], t[1], t[2] }, { t[3], t[4], t[5], t[6] }, { t[7], t[8], t[9] } };
}

__device__ inline void storeContact(bool has, const ContactConstraint &c,
                                    int32_t kind, float *out)
{
    // layout of phys_ref_shim.cpp: has, ref-is-a, numPoints, normal, points
    out[0] = has ? 1.f : 0.f;
    out[1] = has && kind == 0 && c.ref.archetype == 1 ? 1.f : 0.f;
    out[2] = has ? (float)c.numPoints : 0.f;
    out[3] = has ? c.normal.x : 0.f;
    out[4] = has ? c.normal.y : 0.f;
    out[5] = has ? c.normal.z : 0.f;
    for (int i = 0; i < 4; i++) {
        out[6 + 4 * i + 0] = has ? c.points[i].x : 0.f;
        out[6 + 4 * i + 1] = has ? c.points[i].y : 0.f;
        out[6 + 4 * i + 2] = has ? c.points[i].z : 0.f;
        out[6 + 4 * i + 3] = has ? c.points[i].w : 0.f;
    }
}

__global__ void __launch_bounds__(64)
collideKernel(const ObjectManager *obj_mgr, const PairIn *pairs,
              uint32_t num_pairs, int32_t kind, int32_t mode, float *out,
              int32_t *flags)
{
    __shared__ kernels::WaveScratch scratch;
    __shared__ kernels::HullScratch group_scratch[4];

    const uint32_t lane = threadIdx.x;
    // lanes per pair and pairs per wavefront of the cooperative modes
    const uint32_t group_lanes = mode == 3 ? 32u : (mode == 4 ? 16u : 64u);
    const uint32_t group = lane / group_lanes;
    const uint32_t p = mode == 2 ? blockIdx.x :
        (mode >= 3 ? blockIdx.x * (64u / group_lanes) + group :
                     blockIdx.x * 64 + lane);
    const bool active = p < num_pairs;

    PairSetup pair {};
    if (active) {
        pair.aLoc = Loc { 1, 0 };
        pair.bLoc = Loc { 2, 0 };
        pair.aPrim = &obj_mgr->collisionPrimitives[kind == 2 ? 2 : 0];
        pair.bPrim = &obj_mgr->collisionPrimitives[kind == 1 ? 1 : 0];
        pair.a = loadTxfm(pairs[p].a);
        pair.b = loadTxfm(pairs[p].b);
        pair.test = kind == 1 ? NarrowphaseTest::HullPlane :
            kind == 2 ? NarrowphaseTest::SphereHull :
                        NarrowphaseTest::HullHull;
        pair.aabbOverlap = true;
    }

    ContactConstraint contact {};
    bool has = false;
    bool too_big = false;
    bool unsupported = false;

    if (mode == 0) {
        constexpr int32_t max_elems = MADRONA_PHYS_MAX_HULL_ELEMS;
        geo::Plane tmp_faces[max_elems];
        math::Vector3 tmp_vertices[max_elems];
        if (active) {
            has = collidePairStored(pair, tmp_vertices, tmp_faces, max_elems,
                                    &contact, &unsupported);
        }
    } else if (mode == 1) {
        // lanes take turns on the LDS rows, as in the step kernel
        for (uint32_t first = 0; first < 64; first += kernels::lanePolyRows) {
            if (active && lane >= first &&
                    lane < first + kernels::lanePolyRows) {
                has = kernels::collidePairLane(pair,
                    scratch.lanePoly + (lane - first) * kernels::lanePolyDwords,
                    &contact, &too_big, &unsupported);
            }
        }
    } else if (mode == 2) {
        if (active) {   // wave-uniform
            has = kernels::hullHullWave(lane, pair, &scratch.hull, &contact,
                                        &too_big);
        }
    } else if (mode == 3) {
        if (active) {   // uniform per 32-lane group
            has = kernels::hullHullWave<32>(lane % 32u, pair,
                &group_scratch[group], &contact, &too_big);
        }
    } else {
        if (active) {   // uniform per 16-lane group
            has = kernels::hullHullWave<16>(lane % 16u, pair,
                &group_scratch[group], &contact, &too_big);
        }
    }

    if (active && (mode < 2 || lane % group_lanes == 0)) {
        storeContact(has, contact, kind, out + (size_t)p * 28);
        flags[p] = (too_big ? 1 : 0) | (unsupported ? 2 : 0);
    }
}

// The reference's own known-answer tests for code on this path, evaluated by
// the overlay's headers ON THE DEVICE: tests/math.cpp:23-48 (quaternions) and
// tests/gjk.cpp:19-48 (simplex sub-solvers of the sphere-hull GJK).
__global__ void katKernel(float *out)
{
    using math::Quat;
    using math::Vector3;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;

    Quat q1 = Quat::angleAxis(0, { 0, 1, 0 });
    Quat q2 = Quat::angleAxis(math::toRadians(45), { 0, 1, 0 });
    Quat q3 = Quat::angleAxis(math::toRadians(45), { 1, 0, 0 });
    Quat m1 = q2 * q3;
    const Quat qs[4] = { q1, q2, q3, m1 };
    for (int i = 0; i < 4; i++) {
        out[4 * i + 0] = qs[i].w;
        out[4 * i + 1] = qs[i].x;
        out[4 * i + 2] = qs[i].y;
        out[4 * i + 3] = qs[i].z;
    }

    {
        Vector3 Y[4] {
            { 0.814353108f, 0.195752025f, -0.698764443f },
            { -0.784147143f, 0.126484752f, 0.701235533f },
            { -0.784147143f, 0.126484752f, -0.698764443f },
            { -0.784147143f, 0.126484752f, 0.701235533f },
        };
        gjk::SimplexSolve s3 = gjk::solveTriangle(Y[0], Y[1], Y[2]);
        gjk::SimplexSolve s4 = gjk::solveTetrahedron(Y[0], Y[1], Y[2], Y[3]);
        out[16] = s3.vLen2;
        out[17] = s4.vLen2;
    }
    {
        Vector3 Y[4] {
            { 0.793287277f, 2.86326122f, -0.700307727f },
            { -0.794485092f, -0.542466521f, 0.699692249f },
            { 0.80550468f, -0.536717057f, -0.700307727f },
            { -0.794485092f, -0.542466521f, -0.700307727f },
        };
        gjk::SimplexSolve s = gjk::solveTetrahedron(Y[0], Y[1], Y[2], Y[3]);
        out[18] = s.v.x;
        out[19] = s.v.y;
        out[20] = s.v.z;
        out[21] = s.vLen2;
    }
}

}

extern "C" {

#define API __attribute__((visibility("default")))

// out[22]: q1, q2, q3, q2*q3 (w x y z each), |v|^2 of the 3- and 4-simplex
// solves with a duplicated point, v and |v|^2 of the simplex around the origin
API int32_t dev_reference_kats(float *out)
{
    float *d_out = nullptr;
    if (hipMalloc(&d_out, sizeof(float) * 22) != hipSuccess) return -2;
    hipLaunchKernelGGL(katKernel, dim3(1), dim3(64), 0, 0, d_out);
    int32_t rc = hipDeviceSynchronize() == hipSuccess ? 0 : -3;
    (void)hipMemcpy(out, d_out, sizeof(float) * 22, hipMemcpyDeviceToHost);
    (void)hipFree(d_out);
    return rc;
}

// kind: 0 hull-hull, 1 hull-plane, 2 sphere (a) - hull (b); see modes above.
// out: num_pairs x 28 floats, flags: num_pairs (bit 0: face too big for the
// LDS scratch -- the caller falls back to mode 0 --, bit 1: unsupported).
API int32_t dev_collide_pairs(const float *verts, uint32_t num_verts,
                              const uint32_t *indices,
                              const uint32_t *face_counts, uint32_t num_faces,
                              const float *pairs, uint32_t num_pairs,
                              int32_t kind, int32_t mode, float *out,
                              int32_t *flags)
{
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
    if (buf == nullptr) return -1;

    PhysicsLoader loader(ExecMode::CUDA, 4, 0);
    loader.loadRigidBodies(assets);
    free(buf);

    PairIn *d_pairs = nullptr;
    float *d_out = nullptr;
    int32_t *d_flags = nullptr;
    if (hipMalloc(&d_pairs, sizeof(PairIn) * num_pairs) != hipSuccess ||
        hipMalloc(&d_out, sizeof(float) * 28 * num_pairs) != hipSuccess ||
        hipMalloc(&d_flags, sizeof(int32_t) * num_pairs) != hipSuccess) {
        return -2;
    }
    (void)hipMemcpy(d_pairs, pairs, sizeof(PairIn) * num_pairs,
                    hipMemcpyHostToDevice);

    const uint32_t blocks = mode == 2 ? num_pairs :
        mode == 3 ? (num_pairs + 1) / 2 :
        mode == 4 ? (num_pairs + 3) / 4 : (num_pairs + 63) / 64;
    hipLaunchKernelGGL(collideKernel, dim3(blocks), dim3(64), 0, 0,
                       &loader.getObjectManager(), d_pairs, num_pairs, kind,
                       mode, d_out, d_flags);
    int32_t rc = hipDeviceSynchronize() == hipSuccess ? 0 : -3;

    (void)hipMemcpy(out, d_out, sizeof(float) * 28 * num_pairs,
                    hipMemcpyDeviceToHost);
    (void)hipMemcpy(flags, d_flags, sizeof(int32_t) * num_pairs,
                    hipMemcpyDeviceToHost);
    (void)hipFree(d_pairs);
    (void)hipFree(d_out);
    (void)hipFree(d_flags);
    return rc;
}

}

// Batch ray caster for gfx950 (BASELINE config 5): every view of
// every world rendered to res x res RGBA8 + depth, into the render-target
// columns of the ECS.
//
// What is reproduced from the reference (src/mw/device/bvh_raycast.cpp): the
// image.  Ray generation (:58-88), the object-space ray per instance with its
// t rescaling (:627-646, :747-766), Woop's watertight ray / triangle test with
// back-face culling, the same fmaf placement and the double-precision edge
// fallback (:318-447, :228-270), closest hit, material / override colour (:772-812), the light loop
// with spot cut-off and optional shadow rays (:840-930), pixel encoding
// (:816-835), depth 0 / black on a miss.  Results agree with the reference's
// own code compiled for the host (oracle/ref_shims/raycast_ref_shim.cpp) to
// float rounding: the only freedom a ray caster has is the ORDER in which it
// visits instances, and that moves t_max by an ulp per visited instance
// (t_max * t_scale / t_scale).
//
// What is not: the acceleration structures.  The reference builds 4-wide
// quantised nodes with Embree on the host (BLAS) and a ~1000-line multi-kernel
// LBVH + tree-rotation optimiser on the device (TLAS, bvh.cpp).  Here:
//  * TLAS: instances arrive grouped by world and Morton-sorted inside a world
//    with their world-space boxes (RenderingSystem, madrona/render/ecs.inl), so
//    ONE wavefront per world builds a Karras radix tree over them in LDS --
//    internal node i from the longest-common-prefix function of the codes,
//    boxes bottom-up with LDS arrival counters -- and writes <= n - 1 binary
//    nodes (the fp32 boxes of both children, 64 B) into the world's slice of a
//    node array;
//  * BLAS: a binary median-split BVH per object built once on the host, same
//    node shape (leaves of <= 4 triangles, triangles stored contiguously by
//    leaf);
//  * trace kernel: a 16 x 16 pixel tile of one view per workgroup; the world's
//    TLAS nodes, instances and lights are staged in LDS once per workgroup, the
//    traversal stack (one for both levels) lives in LDS too (a dynamically
//    indexed private array is a scratch array on CDNA).  Children are visited
//    near to far.  Rays of a tile are coherent: a wave walks mostly the same
//    nodes.
#include "runtime_internal.hpp"
#include "render_internal.hpp"

#include <madrona/math.hpp>

namespace madrona {
namespace mwhip {

namespace {

using math::Vector3;
using math::Quat;
using math::Diag3x3;
using math::AABB;

// layouts of the ECS components involved (madrona/render/ecs.hpp)
struct alignas(16) InstanceRec {
    Vector3 position;
    Quat rotation;
    Diag3x3 scale;
    int32_t matID;
    int32_t objectID;
    int32_t worldIDX;
    uint32_t color;
};
static_assert(sizeof(InstanceRec) == 64);

struct alignas(16) ViewRec {
    Vector3 position;
    Quat rotation;
    float xScale;
    float yScale;
    float zNear;
    int32_t worldIDX;
    uint32_t pad;
};
static_assert(sizeof(ViewRec) == 48);

struct LightRec {
    bool directional;       // LightDesc::Type: Directional = true
    bool castShadow;
    Vector3 position;
    Vector3 direction;
    float cutoff;
    float intensity;
    bool active;
};
static_assert(sizeof(LightRec) == 40);

struct alignas(16) LeafBox { AABB aabb; };
static_assert(sizeof(LeafBox) == 32);

constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kNoChild = 0xFFFFFFFFu;

__device__ inline uint32_t laneId()
{
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// ---------------------------------------------------------------------------
// TLAS: Karras radix tree over a world's Morton-sorted instances
// ---------------------------------------------------------------------------
// longest common prefix of the keys of leaves i and j (ties broken by index),
// -1 outside [0, n)
__device__ inline int32_t lcp(const uint32_t *codes, int32_t n, int32_t i,
                              int32_t j)
{
    if (j < 0 || j >= n) {
        return -1;
    }
    const uint32_t a = codes[i], b = codes[j];
    if (a != b) {
        return __builtin_clz(a ^ b);
    }
    return 32 + __builtin_clz((uint32_t)i ^ (uint32_t)j);
}

// Boxes that only steer a traversal (top- and bottom-level nodes, the staged
// leaf boxes) are grown by a hair: a ray running exactly along a face of the
// geometry (axis-aligned scenes seen through the exact centre row / column of
// an odd resolution) must still enter the box -- the watertight triangle test
// behind it counts a hit on that face's edge, and so does the reference
// (whose dequantised node boxes are conservative too).
__host__ __device__ inline AABB inflated(const AABB &b)
{
    auto grow = [](float v) { return 1e-5f * (v < 0.f ? -v : v) + 1e-6f; };
    return AABB {
        { b.pMin.x - grow(b.pMin.x), b.pMin.y - grow(b.pMin.y),
          b.pMin.z - grow(b.pMin.z) },
        { b.pMax.x + grow(b.pMax.x), b.pMax.y + grow(b.pMax.y),
          b.pMax.z + grow(b.pMax.z) },
    };
}

__device__ inline AABB merge(const AABB &a, const AABB &b)
{
    return AABB {
        { fminf(a.pMin.x, b.pMin.x), fminf(a.pMin.y, b.pMin.y),
          fminf(a.pMin.z, b.pMin.z) },
        { fmaxf(a.pMax.x, b.pMax.x), fmaxf(a.pMax.y, b.pMax.y),
          fmaxf(a.pMax.z, b.pMax.z) },
    };
}


// (inverse rotation / scale and the object's offsets are worked out once per
// instance by the TLAS build instead of once per ray that enters it)
static_assert(sizeof(PreparedInstance) == 64);

__device__ inline PreparedInstance prepareInstance(const InstanceRec &inst,
                                                   const RenderGeometryDev &geo)
{
    PreparedInstance p;
    p.invRotation = inst.rotation.inv();
    p.invScale = inst.scale.inv();
    p.position = inst.position;
    const bool has_volume = !(inst.scale.d0 == 0.f && inst.scale.d1 == 0.f &&
                              inst.scale.d2 == 0.f);
    const bool known = (uint32_t)inst.objectID < geo.numObjects;
    const uint32_t obj = known ? (uint32_t)inst.objectID : 0u;
    p.objectID = obj;
    p.isBox = geo.objectIsBox[obj];
    p.nodeOffset = geo.objectNodeOffset[obj];
    p.triangleOffset = geo.objectTriangleOffset[obj];
    p.valid = has_volume && known ? 1 : 0;
    p.pad = 0;
    return p;
}


template <int MAXN>
struct TlasLDS {
    uint32_t codes[MAXN];
    uint32_t left[MAXN];                // children of internal node i
    uint32_t right[MAXN];
    int32_t parent[2 * MAXN];           // [0, n-1): internal, [n-1 ..): leaves
    uint32_t arrivals[MAXN];
    AABB box[MAXN];                     // internal nodes
};

// worlds of up to this many instances are built by the small instantiation (3 KB
// of LDS: two dozen worlds per CU at a time instead of three), the rest by the
// large one; each launch skips the other's worlds
constexpr int32_t kSmallTlasLeaves = 64;

template <int MAXN>
__global__ void __launch_bounds__(64)
renderTlasBuild(EcsState *S, RenderParams params)
{
    TraceScope trace_scope(S);
    const int32_t world = (int32_t)blockIdx.x;
    const uint32_t lane = laneId();

    const TableHdr &tbl = S->tables[params.layout.renderable_archetype];
    const int32_t first = tbl.worldOffsets[world];
    const int32_t n = tbl.worldCounts[world];
    if (n <= 0) {
        return;
    }
    if (n > (int32_t)kMaxTlasLeaves || tbl.needsSort != 0u) {
        if (MAXN == (int)kMaxTlasLeaves) {
            raiseError(S, kErrRender);
        }
        return;
    }
    if ((MAXN == kSmallTlasLeaves) != (n <= kSmallTlasLeaves)) {
        return;     // the other instantiation's world
    }

    const uint32_t *codes_hbm =
        (const uint32_t *)tbl.columns[params.mortonColumn] + first;
    const LeafBox *leaf_boxes =
        (const LeafBox *)tbl.columns[params.tlbvhColumn] + first;
    BvhNode *nodes = params.tlasNodes + first;

    // the instances as the trace kernel wants them (world -> object matrices)
    {
        const InstanceRec *records =
            (const InstanceRec *)tbl.columns[params.instanceColumn] + first;
        for (int32_t i = (int32_t)lane; i < n; i += 64) {
            params.prepared[first + i] = prepareInstance(records[i], params.geometry);
        }
    }

    if (n == 1) {
        if (lane == 0) {
            BvhNode root {};
            root.box[0] = inflated(leaf_boxes[0].aabb);
            root.box[1] = root.box[0];
            root.child[0] = kLeafBit | 0u;
            root.child[1] = kNoChild;
            nodes[0] = root;
        }
        return;
    }

    __shared__ TlasLDS<MAXN> lds;

    for (int32_t i = (int32_t)lane; i < n; i += 64) {
        lds.codes[i] = codes_hbm[i];
        lds.arrivals[i] = 0;
    }
    if (lane == 0) {
        lds.parent[0] = -1;
    }
    __syncthreads();

    // ---- topology: one internal node per lane (Karras 2012, fig. 4) --------
    for (int32_t i = (int32_t)lane; i < n - 1; i += 64) {
        const int32_t d = (lcp(lds.codes, n, i, i + 1) -
                           lcp(lds.codes, n, i, i - 1)) >= 0 ? 1 : -1;
        const int32_t delta_min = lcp(lds.codes, n, i, i - d);

        int32_t l_max = 2;
        while (lcp(lds.codes, n, i, i + l_max * d) > delta_min) {
            l_max *= 2;
        }
        int32_t l = 0;
        for (int32_t t = l_max / 2; t >= 1; t /= 2) {
            if (lcp(lds.codes, n, i, i + (l + t) * d) > delta_min) {
                l += t;
            }
        }
        const int32_t j = i + l * d;

        const int32_t delta_node = lcp(lds.codes, n, i, j);
        int32_t s = 0;
        for (int32_t t = (l + 1) / 2; ; t = (t + 1) / 2) {
            if (lcp(lds.codes, n, i, i + (s + t) * d) > delta_node) {
                s += t;
            }
            if (t == 1) break;
        }
        const int32_t gamma = i + s * d + (d < 0 ? d : 0);

        const int32_t lo = i < j ? i : j;
        const int32_t hi = i < j ? j : i;
        const bool left_leaf = lo == gamma;
        const bool right_leaf = hi == gamma + 1;
        lds.left[i] = left_leaf ? (kLeafBit | (uint32_t)gamma) : (uint32_t)gamma;
        lds.right[i] = right_leaf ? (kLeafBit | (uint32_t)(gamma + 1)) :
                                    (uint32_t)(gamma + 1);
        lds.parent[left_leaf ? (n - 1 + gamma) : gamma] = i;
        lds.parent[right_leaf ? (n - 1 + gamma + 1) : (gamma + 1)] = i;
    }
    __syncthreads();

    // ---- boxes, bottom-up: the second child to arrive at a node merges --------
    for (int32_t leaf = (int32_t)lane; leaf < n; leaf += 64) {
        int32_t cur = lds.parent[n - 1 + leaf];
        while (cur >= 0) {
            if (__hip_atomic_fetch_add(&lds.arrivals[cur], 1u, __ATOMIC_ACQ_REL,
                                       __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) {
                break;      // the sibling subtree is not done yet
            }
            const uint32_t l = lds.left[cur], r = lds.right[cur];
            const AABB lb = (l & kLeafBit) != 0u ?
                inflated(leaf_boxes[l & ~kLeafBit].aabb) : lds.box[l];
            const AABB rb = (r & kLeafBit) != 0u ?
                inflated(leaf_boxes[r & ~kLeafBit].aabb) : lds.box[r];
            lds.box[cur] = merge(lb, rb);
            cur = lds.parent[cur];
        }
    }
    __syncthreads();

    for (int32_t i = (int32_t)lane; i < n - 1; i += 64) {
        const uint32_t l = lds.left[i], r = lds.right[i];
        BvhNode node {};
        node.box[0] = (l & kLeafBit) != 0u ?
            inflated(leaf_boxes[l & ~kLeafBit].aabb) : lds.box[l];
        node.box[1] = (r & kLeafBit) != 0u ?
            inflated(leaf_boxes[r & ~kLeafBit].aabb) : lds.box[r];
        node.child[0] = l;
        node.child[1] = r;
        nodes[i] = node;
    }
}

// ---------------------------------------------------------------------------
// tracing
// ---------------------------------------------------------------------------
struct RayIsect {
    int32_t kx, ky, kz;
    float Sx, Sy, Sz;
};

__device__ inline float comp(const Vector3 &v, int32_t k)
{
    return k == 0 ? v.x : (k == 1 ? v.y : v.z);
}

// watertight ray / triangle test WITH back-face culling: the reference compiles
// its test with MADRONA_MESHBVH_BACKFACE_CULLING (defined by mesh_bvh.inl:3,
// which bvh_raycast.cpp includes), i.e. bvh_raycast.cpp:318-447 with the
// culling branches -- same fmaf placement, same double-precision edge fallback,
// same epsilons.  A camera inside a closed mesh sees nothing of it.
__device__ inline bool rayTriangle(const Vector3 &ta, const Vector3 &tb,
                                   const Vector3 &tc, const RayIsect &r,
                                   const Vector3 &org, float t_max,
                                   float *t_out, Vector3 *bary_out = nullptr)
{
    const Vector3 A = ta - org, B = tb - org, C = tc - org;
    const float a_kz = comp(A, r.kz), a_kx = comp(A, r.kx), a_ky = comp(A, r.ky);
    const float b_kz = comp(B, r.kz), b_kx = comp(B, r.kx), b_ky = comp(B, r.ky);
    const float c_kz = comp(C, r.kz), c_kx = comp(C, r.kx), c_ky = comp(C, r.ky);

    const float Ax = fmaf(-r.Sx, a_kz, a_kx), Ay = fmaf(-r.Sy, a_kz, a_ky);
    const float Bx = fmaf(-r.Sx, b_kz, b_kx), By = fmaf(-r.Sy, b_kz, b_ky);
    const float Cx = fmaf(-r.Sx, c_kz, c_kx), Cy = fmaf(-r.Sy, c_kz, c_ky);

    float U = fmaf(Cx, By, -Cy * Bx);
    float V = fmaf(Ax, Cy, -Ay * Cx);
    float W = fmaf(Bx, Ay, -By * Ax);

    constexpr float eps_tol = 1e-7f;
    if (U > -eps_tol && U < eps_tol) U = 0.f;
    if (V > -eps_tol && V < eps_tol) V = 0.f;
    if (W > -eps_tol && W < eps_tol) W = 0.f;

    if (U < 0.f || V < 0.f || W < 0.f) {
        return false;
    }

    if (U == 0.f || V == 0.f || W == 0.f) {
        U = (float)((double)Cx * (double)By - (double)Cy * (double)Bx);
        V = (float)((double)Ax * (double)Cy - (double)Ay * (double)Cx);
        W = (float)((double)Bx * (double)Ay - (double)By * (double)Ax);
        if (U < 0.f || V < 0.f || W < 0.f) {
            return false;
        }
    }

    const float det = U + V + W;
    if (det == 0.f) {
        return false;
    }

    const float Az = r.Sz * a_kz, Bz = r.Sz * b_kz, Cz = r.Sz * c_kz;
    const float T = fmaf(U, Az, fmaf(V, Bz, W * Cz));
    if (T < 0.f || T > t_max * det) {
        return false;
    }

    const float rcp_det = 1.f / det;
    *t_out = T * rcp_det;
    if (bary_out != nullptr) {
        // (reference rayTriangleIntersection :441: Vector3{U,V,W} * rcpDet)
        *bary_out = Vector3 { U, V, W } * rcp_det;
    }
    return true;
}

// A ray prepared for slab tests: t = b * inv - o_inv per bound.  inv from
// v_rcp_f32 (1 ulp): the boxes only steer the traversal, and boxEntry() keeps a
// margin.  A direction component of exactly 0 would give inf - inf = NaN terms
// (and fmaxf(-inf, NaN) = -inf: a box the ray runs through rejected); such a
// component takes a huge finite reciprocal instead, so that t = (b - o) * 1e25
// has the right sign and is far outside [0, t_max] unless o sits on the bound
// -- which the hair the steering boxes are grown by (inflated()) excludes.
struct SlabRay {
    Vector3 inv;
    Vector3 oInv;
};

__device__ inline SlabRay slabRay(const Vector3 &o, const Vector3 &d)
{
    SlabRay r;
    r.inv = Vector3 {
        d.x == 0.f ? __builtin_copysignf(1e25f, d.x) : __builtin_amdgcn_rcpf(d.x),
        d.y == 0.f ? __builtin_copysignf(1e25f, d.y) : __builtin_amdgcn_rcpf(d.y),
        d.z == 0.f ? __builtin_copysignf(1e25f, d.z) : __builtin_amdgcn_rcpf(d.z) };
    r.oInv = Vector3 { o.x * r.inv.x, o.y * r.inv.y, o.z * r.inv.z };
    return r;
}

// entry distance of the ray into a box over [0, t_max], or +inf on a miss
__device__ inline float boxEntry(const AABB &b, const SlabRay &r, float t_max)
{
    const float tx0 = fmaf(b.pMin.x, r.inv.x, -r.oInv.x);
    const float tx1 = fmaf(b.pMax.x, r.inv.x, -r.oInv.x);
    const float ty0 = fmaf(b.pMin.y, r.inv.y, -r.oInv.y);
    const float ty1 = fmaf(b.pMax.y, r.inv.y, -r.oInv.y);
    const float tz0 = fmaf(b.pMin.z, r.inv.z, -r.oInv.z);
    const float tz1 = fmaf(b.pMax.z, r.inv.z, -r.oInv.z);
    const float t_near = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)),
                               fmaxf(fminf(tz0, tz1), 0.f));
    const float t_far = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)),
                              fminf(fmaxf(tz0, tz1), t_max));
    // a little slack: the boxes are exact, the triangle test is watertight and
    // may accept a hit an ulp outside the box
    return t_near <= fmaf(t_far, 1.00001f, 1e-6f) ? t_near : INFINITY;
}

constexpr uint32_t kStackDepth = 24;

// what shading a hit on an instance needs: its rotation (normals) and its colour
// -- material / override resolved once per instance (reference traceRay
// :772-812) unless the hit itself decides: a textured material (sampled at the
// hit's uv) or a mesh with per-triangle materials
constexpr int32_t kShadeFinal = -1;         // `color` is the colour
constexpr int32_t kShadePerTriangle = -3;   // material from the hit triangle

struct alignas(16) ShadeRec {
    Quat rotation;
    Vector3 color;
    int32_t material;   // kShadeFinal / kShadePerTriangle / a textured material
};

__device__ inline Vector3 materialColorOf(const RenderGeometryDev &geo, int32_t m)
{
    return Vector3 { geo.materialColor[3 * m], geo.materialColor[3 * m + 1],
                     geo.materialColor[3 * m + 2] };
}

__device__ inline bool materialIsTextured(const RenderGeometryDev &geo, int32_t m)
{
    return geo.materialTexture != nullptr && geo.materialTexture[m] >= 0;
}

__device__ inline ShadeRec shadeRecord(const InstanceRec &inst,
                                       const RenderGeometryDev &geo)
{
    ShadeRec rec;
    rec.rotation = inst.rotation;
    rec.color = Vector3 { 1.f, 1.f, 1.f };
    rec.material = kShadeFinal;

    if (inst.matID == -2) {             // MaterialOverride::UseOverrideColor
        rec.color = Vector3 { (float)((inst.color >> 16) & 0xFFu) / 255.f,
                              (float)((inst.color >> 8) & 0xFFu) / 255.f,
                              (float)(inst.color & 0xFFu) / 255.f };
        return rec;
    }
    int32_t material = inst.matID;
    if (material == -1) {               // UseDefaultMaterial: the mesh's own
        material = (uint32_t)inst.objectID < geo.numObjects ?
            geo.objectMaterial[inst.objectID] : -1;
        if (material == -1 && geo.triangleMaterial != nullptr &&
                (uint32_t)inst.objectID < geo.numObjects) {
            rec.material = kShadePerTriangle;
            return rec;
        }
    }
    if (material >= 0 && (uint32_t)material < geo.numMaterials) {
        rec.color = materialColorOf(geo, material);
        if (materialIsTextured(geo, material)) {
            rec.material = material;
        }
    }
    return rec;
}

// tex2D<float4>(tex, x, y) of a texture object over an RGBA8 array with wrap
// addressing, linear filtering, normalised coordinates and normalised-float
// reads (how the reference creates its textures, render/asset_processor.cpp:
// 312-345), as the CUDA programming guide defines it: xB = N x - 0.5,
// i = floor(xB), alpha = frac(xB) kept in 9-bit fixed point with 8 fractional
// bits, tex = (1 - a)(1 - b) T[i, j] + a (1 - b) T[i + 1, j] + (1 - a) b
// T[i, j + 1] + a b T[i + 1, j + 1], indices wrapped.  The same definition is
// the oracle's tex2D (oracle/ref_shims/raycast_ref_shim.cpp): there is no CUDA
// texture unit here to compare either of them with.
__device__ inline Vector3 sampleTexture(const RenderGeometryDev &geo, int32_t tex,
                                        float x, float y)
{
    const uint32_t first = geo.textureInfo[4 * tex];
    const int32_t width = (int32_t)geo.textureInfo[4 * tex + 1];
    const int32_t height = (int32_t)geo.textureInfo[4 * tex + 2];
    const float xb = x * (float)width - 0.5f;
    const float yb = y * (float)height - 0.5f;
    const float xf = floorf(xb), yf = floorf(yb);
    const float a = floorf((xb - xf) * 256.f + 0.5f) * (1.f / 256.f);
    const float b = floorf((yb - yf) * 256.f + 0.5f) * (1.f / 256.f);
    auto wrap = [](int32_t i, int32_t n) {
        i %= n;
        return i < 0 ? i + n : i;
    };
    const int32_t i0 = wrap((int32_t)xf, width), i1 = wrap((int32_t)xf + 1, width);
    const int32_t j0 = wrap((int32_t)yf, height), j1 = wrap((int32_t)yf + 1, height);
    auto texel = [&](int32_t i, int32_t j) {
        const uint32_t t = geo.texels[first + (uint32_t)(j * width + i)];
        return Vector3 { (float)(t & 0xFFu) / 255.f, (float)((t >> 8) & 0xFFu) / 255.f,
                         (float)((t >> 16) & 0xFFu) / 255.f };
    };
    const Vector3 t00 = texel(i0, j0), t10 = texel(i1, j0);
    const Vector3 t01 = texel(i0, j1), t11 = texel(i1, j1);
    return (1.f - a) * (1.f - b) * t00 + a * (1.f - b) * t10 +
           (1.f - a) * b * t01 + a * b * t11;
}

struct TraceLDS {
    static constexpr int maxInstances = 64;
    static constexpr int maxLights = 8;

    BvhNode nodes[maxInstances];
    PreparedInstance instances[maxInstances];
    LightRec lights[maxLights];
    // world-space boxes of the instances, and the ones the current tile's
    // frustum touches, near to far
    AABB leafBox[maxInstances];
    ShadeRec shade[maxInstances];
    uint32_t tileCount;
    uint8_t tileList[maxInstances];
    // traversal stack (both levels), one column per thread
    uint16_t stack[kStackDepth][256];
};

struct Hit {
    bool hit;
    float t;            // world-space distance along the normalised ray
    int32_t instance;   // index inside the world
    uint32_t triangle;  // index into the executor's triangle array
};

struct WorldView {
    const BvhNode *nodes;                  // HBM (worlds too big for the LDS slot)
    const PreparedInstance *prepared;
    int32_t numInstances;
};

struct GeoView {
    const BvhNode *nodes;
    const Vector3 *triangles;
    const float *bounds;        // 6 per object
    const uint32_t *boxFaces;   // 12 per object
};

// shear constants of Woop's test: exact division for the axis the hit distance
// is measured along (reference computeRayIsectInfo, :228-270)
__device__ inline RayIsect rayIsect(const Vector3 &d)
{
    const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    RayIsect isect;
    isect.kz = (ax > ay && ax > az) ? 0 : (ay > az ? 1 : 2);
    isect.kx = isect.kz + 1 == 3 ? 0 : isect.kz + 1;
    isect.ky = isect.kx + 1 == 3 ? 0 : isect.kx + 1;
    if (comp(d, isect.kz) < 0.f) {
        const int32_t t = isect.kx; isect.kx = isect.ky; isect.ky = t;
    }
    isect.Sz = 1.f / comp(d, isect.kz);
    isect.Sx = comp(d, isect.kx) * isect.Sz;
    isect.Sy = comp(d, isect.ky) * isect.Sz;
    return isect;
}

// closest hit of the ray against one instance's triangles: the ray goes to
// object space, t is rescaled on the way in and out (reference :627-646,
// :747-766).  The stack above `sp_base` is free.
template <bool AnyHit>
__device__ __forceinline__ void traceInstance(
    EcsState *S, const GeoView &geo, const PreparedInstance &inst,
    int32_t inst_idx, const Vector3 &world_o, const Vector3 &world_d,
    float &t_max, Hit &best, TraceLDS *lds, uint32_t sp_base,
    uint32_t tid)
{
    if (inst.valid == 0) {
        return;
    }
    const Vector3 o =
        inst.invScale * inst.invRotation.rotateVec(world_o - inst.position);
    Vector3 d = inst.invScale * inst.invRotation.rotateVec(world_d);
    const float t_scale = d.length();
    t_max *= t_scale;
    d /= t_scale;

    const Vector3 *tris = geo.triangles + 3u * (size_t)inst.triangleOffset;
    const SlabRay slab = slabRay(o, d);

    if (inst.isBox != 0u) {
        // The mesh is its own bounding box (12 outward triangles): a ray from
        // outside can only hit the face through which it enters the slabs (from
        // inside every face is a back face: culled).  The slabs pick the face,
        // its two triangles go through the reference's test -- the same t, bit
        // for bit -- instead of a tree walk over all twelve.
        const float *b = geo.bounds + 6u * inst.objectID;
        const float tx0 = fmaf(b[0], slab.inv.x, -slab.oInv.x);
        const float tx1 = fmaf(b[3], slab.inv.x, -slab.oInv.x);
        const float ty0 = fmaf(b[1], slab.inv.y, -slab.oInv.y);
        const float ty1 = fmaf(b[4], slab.inv.y, -slab.oInv.y);
        const float tz0 = fmaf(b[2], slab.inv.z, -slab.oInv.z);
        const float tz1 = fmaf(b[5], slab.inv.z, -slab.oInv.z);
        float nx = fminf(tx0, tx1), ny = fminf(ty0, ty1), nz = fminf(tz0, tz1);
        float fx = fmaxf(tx0, tx1), fy = fmaxf(ty0, ty1), fz = fmaxf(tz0, tz1);
        // A ray parallel to a pair of faces is inside that slab for every t or
        // for none -- bounds included: running exactly along a face it meets
        // the neighbouring faces on their shared edge, which the watertight
        // test (and so the reference) counts as a hit.
        bool outside = false;
        if (d.x == 0.f) {
            outside = outside || o.x < b[0] || o.x > b[3];
            nx = -INFINITY; fx = INFINITY;
        }
        if (d.y == 0.f) {
            outside = outside || o.y < b[1] || o.y > b[4];
            ny = -INFINITY; fy = INFINITY;
        }
        if (d.z == 0.f) {
            outside = outside || o.z < b[2] || o.z > b[5];
            nz = -INFINITY; fz = INFINITY;
        }
        if (outside) {
            t_max = t_max / t_scale;
            return;
        }
        const float t_near = fmaxf(fmaxf(nx, ny), nz);
        const float t_far = fminf(fminf(fx, fy), fminf(fz, t_max));
        if (t_near <= fmaf(t_far, 1.00001f, 1e-6f)) {
            const RayIsect isect = rayIsect(d);
            const uint32_t axis = nx >= ny && nx >= nz ? 0u : (ny >= nz ? 1u : 2u);
            const float dir = axis == 0u ? d.x : (axis == 1u ? d.y : d.z);
            const uint32_t face = axis * 2u + (dir > 0.f ? 0u : 1u);
            const uint32_t *face_tris =
                geo.boxFaces + 12u * inst.objectID + 2u * face;
            bool found = false;
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const uint32_t tri_idx = face_tris[k];
                const Vector3 *tri = tris + 3u * (size_t)tri_idx;
                float t;
                if (rayTriangle(tri[0], tri[1], tri[2], isect, o, t_max, &t)) {
                    t_max = t;
                    best.hit = true;
                    best.instance = inst_idx;
                    best.triangle = inst.triangleOffset + tri_idx;
                    found = true;
                }
            }
            // (a ray grazing an edge of the box: the slabs and the watertight
            // test may disagree about the face -- all twelve, then)
            if (!found && fabsf(t_near - t_far) <= 1e-4f * fabsf(t_far) + 1e-6f) {
                for (uint32_t tri_idx = 0; tri_idx < 12u; tri_idx++) {
                    const Vector3 *tri = tris + 3u * (size_t)tri_idx;
                    float t;
                    if (rayTriangle(tri[0], tri[1], tri[2], isect, o, t_max, &t)) {
                        t_max = t;
                        best.hit = true;
                        best.instance = inst_idx;
                        best.triangle = inst.triangleOffset + tri_idx;
                    }
                }
            }
        }
        t_max = t_max / t_scale;
        return;
    }

    const RayIsect isect = rayIsect(d);
    const BvhNode *nodes = geo.nodes + inst.nodeOffset;

    uint32_t sp = sp_base;
    uint32_t cur = 0;
    while (true) {
        const BvhNode node = nodes[cur];
        float entry[2];
#pragma unroll
        for (int c = 0; c < 2; c++) {
            entry[c] = node.child[c] == kNoChild ? INFINITY :
                boxEntry(node.box[c], slab, t_max);
        }
        const int near = entry[1] < entry[0] ? 1 : 0;
        uint32_t next = kNoChild;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int c = k == 0 ? near : 1 - near;
            if (!(entry[c] <= t_max)) {
                continue;
            }
            const uint32_t child = node.child[c];
            if ((child & kLeafBit) != 0u) {
                // leaf: bits 0..27 first triangle, 28..30 count - 1
                const uint32_t first_tri = child & 0x0FFFFFFFu;
                const uint32_t num_tris = ((child >> 28) & 7u) + 1u;
                for (uint32_t t_idx = 0; t_idx < num_tris; t_idx++) {
                    const Vector3 *tri = tris + 3u * (size_t)(first_tri + t_idx);
                    float t;
                    if (rayTriangle(tri[0], tri[1], tri[2], isect, o, t_max, &t)) {
                        t_max = t;
                        best.hit = true;
                        best.instance = inst_idx;
                        best.triangle = inst.triangleOffset + first_tri + t_idx;
                        if constexpr (AnyHit) {
                            t_max = t_max / t_scale;
                            return;
                        }
                    }
                }
            } else if (next == kNoChild) {
                next = child;
            } else if (sp < kStackDepth) {
                lds->stack[sp++][tid] = (uint16_t)child;
            } else {
                raiseError(S, kErrRender);
            }
        }
        if (next != kNoChild) {
            cur = next;
        } else {
            if (sp == sp_base) break;
            cur = lds->stack[--sp][tid];
        }
    }

    t_max = t_max / t_scale;
}

// Staged: the world's nodes and instances are the workgroup's LDS copies (known
// at compile time, so that they are read with LDS instructions, not through
// flat pointers)
template <bool AnyHit, bool Staged>
__device__ __forceinline__ Hit traceWorld(
    EcsState *S, const GeoView &geo, const WorldView &w, const Vector3 &o,
    const Vector3 &d, float t_max, TraceLDS *lds, uint32_t tid)
{
    const BvhNode *world_nodes = Staged ? lds->nodes : w.nodes;
    const PreparedInstance *world_instances = Staged ? lds->instances : w.prepared;

    Hit best;
    best.hit = false;
    best.t = 0.f;
    best.instance = -1;
    best.triangle = 0;

    if (w.numInstances <= 0) {
        return best;
    }
    const SlabRay slab = slabRay(o, d);

    uint32_t sp = 0;
    uint32_t cur = 0;       // internal node index
    while (true) {
        const BvhNode node = world_nodes[cur];
        float entry[2];
#pragma unroll
        for (int c = 0; c < 2; c++) {
            entry[c] = node.child[c] == kNoChild ? INFINITY :
                boxEntry(node.box[c], slab, t_max);
        }
        const int near = entry[1] < entry[0] ? 1 : 0;
        uint32_t next = kNoChild;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int c = k == 0 ? near : 1 - near;
            if (!(entry[c] <= t_max)) {
                continue;
            }
            const uint32_t child = node.child[c];
            if ((child & kLeafBit) != 0u) {
                const int32_t idx = (int32_t)(child & ~kLeafBit);
                traceInstance<AnyHit>(S, geo, world_instances[idx], idx, o, d, t_max,
                                      best, lds, sp, tid);
                if constexpr (AnyHit) {
                    if (best.hit) {
                        best.t = t_max;
                        return best;
                    }
                }
            } else if (next == kNoChild) {
                next = child;
            } else if (sp < kStackDepth) {
                lds->stack[sp++][tid] = (uint16_t)child;
            } else {
                raiseError(S, kErrRender);
            }
        }
        if (next != kNoChild) {
            cur = next;
        } else {
            if (sp == 0) break;
            cur = lds->stack[--sp][tid];
        }
    }

    best.t = t_max;
    return best;
}

// Primary rays of a tile whose world is staged: instead of a top-level walk per
// ray, the instances the tile's frustum touches (culled once per tile by one
// wavefront, ordered near to far) are tried in turn behind their own box test.
__device__ __forceinline__ Hit traceTileList(
    EcsState *S, const GeoView &geo, const Vector3 &o, const Vector3 &d,
    float t_max, TraceLDS *lds, uint32_t tid)
{
    Hit best;
    best.hit = false;
    best.t = 0.f;
    best.instance = -1;
    best.triangle = 0;

    const SlabRay slab = slabRay(o, d);
    const uint32_t n = lds->tileCount;
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t idx = lds->tileList[k];
        if (boxEntry(lds->leafBox[idx], slab, t_max) <= t_max) {
            traceInstance<false>(S, geo, lds->instances[idx], (int32_t)idx, o, d,
                                 t_max, best, lds, 0u, tid);
        }
    }
    best.t = t_max;
    return best;
}

// Workgroups stride over the 16 x 16 tiles of all views.
//
// PlainMaterials: the scene has neither per-triangle materials nor textures
// (geometry.triangleMaterial == nullptr and no material with a texture: then
// shadeRecord never leaves kShadeFinal), so the per-hit material / texture
// lookup is not even compiled in -- BASELINE config 5's scene; round 3 paid a
// run-time branch and the registers of the texture path for it (4.36 -> 4.69 ms).
//
// Occupancy (round 4, profiles/r04_raycast_variants.jsonl, config 5): the
// kernel is a chain of dependent node / triangle fetches per ray, so what it
// needs is wavefronts to switch between.  With the geometry in LDS the block is
// 53 KB: three workgroups per CU, three wavefronts per SIMD, 145 registers.
// With the geometry left in HBM (it is a few KB that every workgroup reads: L2
// and L1 hits) the block is 24 KB, and capped at 128 registers (13 spilled
// dwords) four wavefronts share a SIMD: 4.84 -> 4.45 ms at the same grid; 96
// registers for five spill 66 dwords and lose (4.28 ms against 3.42).  Hence:
// geometry in HBM and four wavefronts per SIMD (the LDS copy and its switch
// are out of the code since round 6).
#ifndef MADRONA_RAYCAST_WAVES
#define MADRONA_RAYCAST_WAVES 4
#endif
template <bool PlainMaterials>
__global__ void __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(MADRONA_RAYCAST_WAVES)))
renderRaycast(EcsState *S, RenderParams params)
{
    TraceScope trace_scope(S);
    __shared__ TraceLDS lds;

    const uint32_t res = params.resolution;
    const uint32_t tiles_per_side = (res + 15u) / 16u;
    const uint32_t tiles_per_view = tiles_per_side * tiles_per_side;
    const uint32_t tid = threadIdx.x;

    const TableHdr &cam_tbl = S->tables[params.layout.camera_archetype];
    const TableHdr &inst_tbl = S->tables[params.layout.renderable_archetype];
    const TableHdr &light_tbl = S->tables[params.layout.light_archetype];
    const TableHdr &out_tbl = S->tables[params.layout.output_archetype];
    const uint32_t num_views = (uint32_t)cam_tbl.numRows;
    const uint32_t total_tiles = num_views * tiles_per_view;

    const RenderGeometryDev geo_dev = params.geometry;
    // (bottom-level trees and triangles stay in HBM: a few KB every workgroup
    // reads, L2 / L1 hits; copied to LDS they made the block 53 KB and cost a
    // wavefront per SIMD -- round 4, profiles/r04_raycast_variants.jsonl)
    GeoView geo;
    geo.nodes = geo_dev.nodes;
    geo.triangles = geo_dev.triangleVertices;
    geo.bounds = geo_dev.objectBounds;
    geo.boxFaces = geo_dev.objectBoxFaces;

    uint8_t *rgb_out = (uint8_t *)out_tbl.columns[params.rgbColumn];
    float *depth_out = (float *)out_tbl.columns[params.depthColumn];
    const uint32_t pixels_per_view = res * res;
    int32_t staged_world = -1;

    // a contiguous run of tiles per workgroup: the tiles of a view, and the views
    // of a world, follow each other
    // (whole views where there are views enough to go round -- a workgroup that
    // takes half a view stages the same world as its neighbour: config 5 at one
    // workgroup per view 3.42 ms, at two 5.61 ms -- the grid is sized for the
    // camera table's capacity, the rows it holds decide here)
    uint32_t tiles_per_wg = (total_tiles + gridDim.x - 1u) / gridDim.x;
    if (num_views >= gridDim.x || num_views >= 2048u) {
        const uint32_t views_per_wg = (num_views + gridDim.x - 1u) / gridDim.x;
        tiles_per_wg = (views_per_wg > 0u ? views_per_wg : 1u) * tiles_per_view;
    }
    const uint32_t tile_begin = blockIdx.x * tiles_per_wg;
    const uint32_t tile_end = tile_begin + tiles_per_wg < total_tiles ?
        tile_begin + tiles_per_wg : total_tiles;

    for (uint32_t tile = tile_begin; tile < tile_end; tile++) {
        const uint32_t view_idx = tile / tiles_per_view;
        const uint32_t tile_in_view = tile % tiles_per_view;
        const ViewRec view =
            ((const ViewRec *)cam_tbl.columns[params.cameraColumn])[view_idx];
        const int32_t world = view.worldIDX;

        const int32_t inst_first = inst_tbl.worldOffsets[world];
        const int32_t num_inst = inst_tbl.worldCounts[world];
        const int32_t light_first = light_tbl.worldOffsets[world];
        const int32_t num_lights = light_tbl.worldCounts[world];
        const int32_t staged_lights =
            num_lights < lds.maxLights ? num_lights : lds.maxLights;
        const LightRec *lights_hbm = (const LightRec *)
            light_tbl.columns[params.lightColumn] + light_first;

        const InstanceRec *inst_hbm =
            (const InstanceRec *)inst_tbl.columns[params.instanceColumn] +
            inst_first;
        const BvhNode *nodes_hbm = params.tlasNodes + inst_first;
        const PreparedInstance *prepared_hbm = params.prepared + inst_first;
        const bool staged = num_inst <= lds.maxInstances;

        // ---- stage the world next to the CU (tiles of one world follow each
        // other: once per world, not per tile) -------------------------------
        if (world != staged_world) {
            __syncthreads();
            if (staged) {
                const uint32_t inst_dw = (uint32_t)num_inst * 16u;
                for (uint32_t i = tid; i < inst_dw; i += 256u) {
                    ((uint32_t *)lds.instances)[i] =
                        ((const uint32_t *)prepared_hbm)[i];
                }
                const uint32_t node_dw =
                    (uint32_t)(num_inst > 1 ? num_inst - 1 : num_inst) * 16u;
                for (uint32_t i = tid; i < node_dw; i += 256u) {
                    ((uint32_t *)lds.nodes)[i] = ((const uint32_t *)nodes_hbm)[i];
                }
                const LeafBox *boxes_hbm = (const LeafBox *)
                    inst_tbl.columns[params.tlbvhColumn] + inst_first;
                if ((int32_t)tid < num_inst) {
                    lds.leafBox[tid] = inflated(boxes_hbm[tid].aabb);
                    lds.shade[tid] = shadeRecord(inst_hbm[tid], geo_dev);
                }
            }
            const uint32_t light_dw = (uint32_t)staged_lights * 10u;
            const uint32_t *src = (const uint32_t *)((const LightRec *)
                light_tbl.columns[params.lightColumn] + light_first);
            for (uint32_t i = tid; i < light_dw; i += 256u) {
                ((uint32_t *)lds.lights)[i] = src[i];
            }
            staged_world = world;
            __syncthreads();
        }

        WorldView wv;
        wv.nodes = nodes_hbm;
        wv.prepared = prepared_hbm;
        wv.numInstances = num_inst;

        const uint32_t tile_x = (tile_in_view % tiles_per_side) * 16u;
        const uint32_t tile_y = (tile_in_view / tiles_per_side) * 16u;
        const uint32_t px = tile_x + (tid & 15u);
        const uint32_t py = tile_y + (tid >> 4);

        // ---- the view's rays (reference calculateOutRay, :58-88) --------------
        const Quat rot = view.rotation;
        const Vector3 ray_start = view.position;
        const Vector3 look_at = rot.inv().rotateVec(Vector3 { 0.f, 1.f, 0.f });
        const float h = 1.f / (-view.yScale);
        const float viewport = 2.f * h;
        const Vector3 forward = look_at.normalize();
        const Vector3 u = rot.inv().rotateVec(Vector3 { 1.f, 0.f, 0.f });
        const Vector3 v = math::cross(forward, u).normalize();
        const Vector3 horizontal = u * viewport;
        const Vector3 vertical = v * viewport;
        const Vector3 lower_left =
            ray_start - horizontal / 2.f - vertical / 2.f + forward;

        // ---- which instances can the tile's rays meet at all -------------------
        if (staged) {
            __syncthreads();        // the previous tile's list is not read any more
            if (tid < 64u) {
                // frustum through the outer edges of the tile's pixels
                const float u0 = (float)tile_x / (float)res;
                const float v0 = (float)tile_y / (float)res;
                const float u1 = (float)(tile_x + 16u < res ? tile_x + 16u : res) /
                                 (float)res;
                const float v1 = (float)(tile_y + 16u < res ? tile_y + 16u : res) /
                                 (float)res;
                const Vector3 base = lower_left - ray_start;
                const Vector3 corner[4] = {
                    base + u0 * horizontal + v0 * vertical,
                    base + u1 * horizontal + v0 * vertical,
                    base + u1 * horizontal + v1 * vertical,
                    base + u0 * horizontal + v1 * vertical,
                };
                const Vector3 centre = corner[0] + corner[2];

                bool inside = (int32_t)tid < num_inst &&
                              lds.instances[tid < 64u ? tid : 0u].valid != 0;
                const AABB box = lds.leafBox[tid < (uint32_t)lds.maxInstances ?
                                             tid : 0u];
                const Vector3 lo = box.pMin - ray_start;
                const Vector3 hi = box.pMax - ray_start;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    Vector3 n = math::cross(corner[k], corner[(k + 1) & 3]);
                    if (n.dot(centre) < 0.f) {
                        n = -n;
                    }
                    // the box corner furthest along the inward normal
                    const float reach = n.x * (n.x > 0.f ? hi.x : lo.x) +
                                        n.y * (n.y > 0.f ? hi.y : lo.y) +
                                        n.z * (n.z > 0.f ? hi.z : lo.z);
                    // (a margin: culling must never lose an instance a ray hits)
                    inside = inside && reach >= -1e-4f * (fabsf(n.x) + fabsf(n.y) +
                                                          fabsf(n.z));
                }
                const float ahead = forward.x * (forward.x > 0.f ? hi.x : lo.x) +
                                    forward.y * (forward.y > 0.f ? hi.y : lo.y) +
                                    forward.z * (forward.z > 0.f ? hi.z : lo.z);
                inside = inside && ahead >= 0.f;
                // near to far by where the box starts along the view direction
                const float key = forward.x * (forward.x > 0.f ? lo.x : hi.x) +
                                  forward.y * (forward.y > 0.f ? lo.y : hi.y) +
                                  forward.z * (forward.z > 0.f ? lo.z : hi.z);

                const uint64_t mask = __builtin_amdgcn_ballot_w64(inside);
                uint32_t rank = 0;
                for (uint64_t m = mask; m != 0; m &= m - 1) {
                    const uint32_t j = (uint32_t)__builtin_ctzll(m);
                    const float other = __shfl(key, (int)j);
                    rank += (other < key || (other == key && j < tid)) ? 1u : 0u;
                }
                if (inside) {
                    lds.tileList[rank] = (uint8_t)tid;
                }
                if (tid == 0u) {
                    lds.tileCount = (uint32_t)__builtin_popcountll(mask);
                }
            }
            __syncthreads();
        }

        if (px >= res || py >= res) {
            continue;
        }

        const float pixel_u = ((float)px + 0.5f) / (float)res;
        const float pixel_v = ((float)py + 0.5f) / (float)res;
        const Vector3 ray_dir = (lower_left + pixel_u * horizontal +
                                 pixel_v * vertical - ray_start).normalize();

        const Hit first = staged ?
            traceTileList(S, geo, ray_start, ray_dir, 10000.f, &lds, tid) :
            traceWorld<false, false>(S, geo, wv, ray_start, ray_dir, 10000.f,
                                     &lds, tid);

        const uint32_t pixel = px + py * res;
        float depth = 0.f;
        Vector3 shaded { 0.f, 0.f, 0.f };
        if (first.hit) {
            depth = first.t;
            if (params.rgbd != 0u) {
                const ShadeRec shade = staged ? lds.shade[first.instance] :
                    shadeRecord(inst_hbm[first.instance], geo_dev);
                Vector3 color = shade.color;
                // geometric normal of the hit triangle (reference :443-446)
                const Vector3 *tri = geo.triangles + 3u * (size_t)first.triangle;
                if (!PlainMaterials && shade.material != kShadeFinal) {
                    // the hit decides: the triangle's own material and / or a
                    // texture sampled at the hit's uv (reference :772-800)
                    int32_t material = shade.material;
                    if (material == kShadePerTriangle) {
                        material = geo_dev.triangleMaterial[first.triangle];
                        color = Vector3 { 1.f, 1.f, 1.f };
                        if (material >= 0 && (uint32_t)material < geo_dev.numMaterials) {
                            color = materialColorOf(geo_dev, material);
                        } else {
                            material = -1;
                        }
                    }
                    if (material >= 0 && materialIsTextured(geo_dev, material) &&
                            geo_dev.triangleUV != nullptr) {
                        // barycentrics of the hit: the triangle test once more,
                        // on the object-space ray of this instance (:469-478)
                        const PreparedInstance &pi = staged ?
                            lds.instances[first.instance] :
                            prepared_hbm[first.instance];
                        const Vector3 obj_o = pi.invScale *
                            pi.invRotation.rotateVec(ray_start - pi.position);
                        Vector3 obj_d =
                            pi.invScale * pi.invRotation.rotateVec(ray_dir);
                        obj_d /= obj_d.length();
                        float t_again;
                        Vector3 bary { 1.f, 0.f, 0.f };
                        (void)rayTriangle(tri[0], tri[1], tri[2], rayIsect(obj_d),
                                          obj_o, INFINITY, &t_again, &bary);
                        const float *uv = geo_dev.triangleUV + 6u * (size_t)first.triangle;
                        const float u = uv[0] * bary.x + uv[2] * bary.y + uv[4] * bary.z;
                        const float v = uv[1] * bary.x + uv[3] * bary.y + uv[5] * bary.z;
                        const Vector3 texel = sampleTexture(geo_dev,
                            geo_dev.materialTexture[material], u, 1.f - v);
                        const Vector3 base = materialColorOf(geo_dev, material);
                        color = Vector3 { texel.x * base.x, texel.y * base.y,
                                          texel.z * base.z };
                    }
                }
                const Vector3 obj_normal =
                    math::cross(tri[1] - tri[0], tri[2] - tri[0]).normalize();
                const Vector3 normal = shade.rotation.rotateVec(obj_normal);
                const Vector3 hit_pos = ray_start + first.t * ray_dir;

                // ---- lights (reference computeFragment, :840-930) -------------
                float light_contrib = 0.f;
                for (int32_t i = 0; i < num_lights; i++) {
                    // (the first few of a world's lights are in LDS)
                    const LightRec light = i < lds.maxLights ? lds.lights[i] :
                                                               lights_hbm[i];
                    Vector3 light_dir = -light.direction;
                    if (!light.directional) {
                        light_dir = (light.position - hit_pos).normalize();
                        if (light.cutoff != -1.f) {
                            float c = (-light_dir).dot(light.direction);
                            c /= light_dir.length() * light.direction.length();
                            const float angle = acosf(c);
                            if (fabsf(angle) > fabsf(light.cutoff)) {
                                continue;
                            }
                        }
                    }
                    if (light.castShadow) {
                        if (light_dir.dot(normal) > 0.f) {
                            // (any hit will do: the reference looks for the
                            // closest one and only asks whether there is one)
                            const Hit shadow = staged ?
                                traceWorld<true, true>(S, geo, wv, hit_pos,
                                    light_dir, 10000.f, &lds, tid) :
                                traceWorld<true, false>(S, geo, wv, hit_pos,
                                    light_dir, 10000.f, &lds, tid);
                            if (!shadow.hit) {
                                light_contrib += fminf(fmaxf(
                                    normal.dot(light_dir), 0.f), 1.f);
                            }
                        }
                    } else {
                        light_contrib += fminf(fmaxf(normal.dot(light_dir), 0.f),
                                               1.f);
                    }
                }
                shaded = fmaxf(0.2f, light_contrib) * color;
                shaded.x = fminf(1.f, shaded.x);
                shaded.y = fminf(1.f, shaded.y);
                shaded.z = fminf(1.f, shaded.z);
            }
        }

        depth_out[(size_t)view_idx * pixels_per_view + pixel] = depth;
        if (params.rgbd != 0u) {
            uint32_t *dst = (uint32_t *)rgb_out +
                            (size_t)view_idx * pixels_per_view + pixel;
            *dst = (uint32_t)(uint8_t)(shaded.x * 255.f) |
                   ((uint32_t)(uint8_t)(shaded.y * 255.f) << 8) |
                   ((uint32_t)(uint8_t)(shaded.z * 255.f) << 16) | (255u << 24);
        }
    }
}

}

// ---------------------------------------------------------------------------
// host: bottom-level BVHs, launches
// ---------------------------------------------------------------------------
namespace {

struct BuildTri {
    float v[9];
    float c[3];
    float uv[6];        // travels with the triangle into leaf order
    int32_t material;
};

AABB boxOf(const std::vector<BuildTri> &tris, uint32_t first, uint32_t count)
{
    AABB b { { 3.4e38f, 3.4e38f, 3.4e38f }, { -3.4e38f, -3.4e38f, -3.4e38f } };
    for (uint32_t i = first; i < first + count; i++) {
        for (int k = 0; k < 3; k++) {
            const float *p = tris[i].v + 3 * k;
            b.pMin.x = std::min(b.pMin.x, p[0]); b.pMax.x = std::max(b.pMax.x, p[0]);
            b.pMin.y = std::min(b.pMin.y, p[1]); b.pMax.y = std::max(b.pMax.y, p[1]);
            b.pMin.z = std::min(b.pMin.z, p[2]); b.pMax.z = std::max(b.pMax.z, p[2]);
        }
    }
    return b;
}

constexpr uint32_t kLeafTris = 4;

// Is the mesh exactly its own axis-aligned bounding box, seen from outside:
// twelve triangles, each flat on one face with its normal pointing out, the
// two of a face covering it?  (What the ray caster may then intersect as slabs.)
bool meshIsItsBounds(const std::vector<BuildTri> &tris, uint32_t *face_tris)
{
    if (tris.size() != 12) {
        return false;
    }
    const AABB box = boxOf(tris, 0, 12);
    const float lo[3] = { box.pMin.x, box.pMin.y, box.pMin.z };
    const float hi[3] = { box.pMax.x, box.pMax.y, box.pMax.z };
    for (int a = 0; a < 3; a++) {
        if (!(hi[a] > lo[a])) return false;
    }
    double face_area[6] = { 0, 0, 0, 0, 0, 0 };
    int face_count[6] = { 0, 0, 0, 0, 0, 0 };
    for (size_t tri_idx = 0; tri_idx < tris.size(); tri_idx++) {
        const BuildTri &t = tris[tri_idx];
        int face = -1;
        for (int a = 0; a < 3 && face < 0; a++) {
            if (t.v[a] == lo[a] && t.v[3 + a] == lo[a] && t.v[6 + a] == lo[a]) {
                face = 2 * a;
            } else if (t.v[a] == hi[a] && t.v[3 + a] == hi[a] &&
                       t.v[6 + a] == hi[a]) {
                face = 2 * a + 1;
            }
        }
        if (face < 0) return false;
        // every vertex a corner of the box
        for (int k = 0; k < 9; k++) {
            if (t.v[k] != lo[k % 3] && t.v[k] != hi[k % 3]) return false;
        }
        const double e1[3] = { (double)t.v[3] - t.v[0], (double)t.v[4] - t.v[1],
                               (double)t.v[5] - t.v[2] };
        const double e2[3] = { (double)t.v[6] - t.v[0], (double)t.v[7] - t.v[1],
                               (double)t.v[8] - t.v[2] };
        const double n[3] = { e1[1] * e2[2] - e1[2] * e2[1],
                              e1[2] * e2[0] - e1[0] * e2[2],
                              e1[0] * e2[1] - e1[1] * e2[0] };
        const int a = face / 2;
        const double outward = (face & 1) != 0 ? n[a] : -n[a];
        if (!(outward > 0.0) || face_count[face] >= 2) return false;
        face_area[face] += 0.5 * outward;
        face_tris[2 * face + face_count[face]++] = (uint32_t)tri_idx;
    }
    for (int a = 0; a < 3; a++) {
        const double want = ((double)hi[(a + 1) % 3] - lo[(a + 1) % 3]) *
                            ((double)hi[(a + 2) % 3] - lo[(a + 2) % 3]);
        for (int side = 0; side < 2; side++) {
            if (std::fabs(face_area[2 * a + side] - want) > 1e-6 * want) {
                return false;
            }
        }
    }
    return true;
}

// node `node_idx` covers triangles [first, first + count), count > kLeafTris:
// median split along the widest axis of the centroids
void buildBlas(std::vector<BuildTri> &tris, uint32_t first, uint32_t count,
               std::vector<BvhNode> &nodes, uint32_t node_idx)
{
    float lo[3] = { 3.4e38f, 3.4e38f, 3.4e38f };
    float hi[3] = { -3.4e38f, -3.4e38f, -3.4e38f };
    for (uint32_t i = first; i < first + count; i++) {
        for (int a = 0; a < 3; a++) {
            lo[a] = std::min(lo[a], tris[i].c[a]);
            hi[a] = std::max(hi[a], tris[i].c[a]);
        }
    }
    int axis = 0;
    if (hi[1] - lo[1] > hi[axis] - lo[axis]) axis = 1;
    if (hi[2] - lo[2] > hi[axis] - lo[axis]) axis = 2;
    const uint32_t half = count / 2;
    std::nth_element(tris.begin() + first, tris.begin() + first + half,
                     tris.begin() + first + count,
                     [axis](const BuildTri &a, const BuildTri &b) {
                         return a.c[axis] < b.c[axis];
                     });

    const uint32_t part_first[2] = { first, first + half };
    const uint32_t part_count[2] = { half, count - half };
    for (int c = 0; c < 2; c++) {
        nodes[node_idx].box[c] = inflated(boxOf(tris, part_first[c], part_count[c]));
        if (part_count[c] <= kLeafTris) {
            nodes[node_idx].child[c] =
                kLeafBit | ((part_count[c] - 1u) << 28) | part_first[c];
        } else {
            const uint32_t child = (uint32_t)nodes.size();
            nodes.push_back(BvhNode {});
            nodes[node_idx].child[c] = child;
            buildBlas(tris, part_first[c], part_count[c], nodes, child);
        }
    }
}

}

int buildRenderGeometry(const mwhip_render_geometry &src, RenderGeometryHost &out,
                        std::string &error)
{
    out.numObjects = src.num_objects;
    out.numMaterials = src.num_materials;
    out.objectNodeOffset.assign(src.num_objects + 1, 0);
    out.objectTriangleOffset.assign(src.num_objects + 1, 0);
    out.objectRootBox.assign((size_t)src.num_objects * 6, 0.f);
    out.objectMaterial.assign(src.num_objects, -1);
    out.objectIsBox.assign(src.num_objects, 0u);
    out.objectBoxFaces.assign((size_t)src.num_objects * 12, 0u);
    if (src.object_material != nullptr) {
        out.objectMaterial.assign(src.object_material,
                                  src.object_material + src.num_objects);
    }
    if (src.num_materials != 0) {
        out.materialColor.assign(src.material_color,
                                 src.material_color + 3 * (size_t)src.num_materials);
    }
    const bool have_uv = src.vertex_uv != nullptr;
    const bool have_tri_mat = src.triangle_material != nullptr;
    if (src.material_texture != nullptr && src.num_materials != 0) {
        out.materialTexture.assign(src.material_texture,
                                   src.material_texture + src.num_materials);
        for (uint32_t t = 0; t < src.num_textures; t++) {
            const mwhip_texture &tex = src.textures[t];
            if (tex.width == 0 || tex.height == 0 || tex.rgba8 == nullptr) {
                error = "render geometry: texture " + std::to_string(t) + " is empty";
                return -1;
            }
            out.textureInfo.push_back((uint32_t)out.texels.size());
            out.textureInfo.push_back(tex.width);
            out.textureInfo.push_back(tex.height);
            out.textureInfo.push_back(0u);
            const uint32_t *px = (const uint32_t *)tex.rgba8;
            out.texels.insert(out.texels.end(), px,
                              px + (size_t)tex.width * tex.height);
        }
        for (int32_t t : out.materialTexture) {
            if (t >= (int32_t)src.num_textures) {
                error = "render geometry: a material names texture " +
                        std::to_string(t) + " of " + std::to_string(src.num_textures);
                return -1;
            }
        }
    }

    for (uint32_t obj = 0; obj < src.num_objects; obj++) {
        const uint32_t tri_first = src.object_triangle_offset[obj];
        const uint32_t tri_count = src.object_triangle_offset[obj + 1] - tri_first;
        const uint32_t vert_first = src.object_vertex_offset[obj];
        const uint32_t vert_count = src.object_vertex_offset[obj + 1] - vert_first;
        const float *verts = src.vertices + 3 * (size_t)vert_first;

        std::vector<BuildTri> tris(tri_count);
        for (uint32_t t = 0; t < tri_count; t++) {
            for (int k = 0; k < 3; k++) {
                const uint32_t vi = src.indices[3 * (size_t)(tri_first + t) + k];
                if (vi >= vert_count) {
                    error = "render geometry: object " + std::to_string(obj) +
                            " indexes vertex " + std::to_string(vi) + " of " +
                            std::to_string(vert_count);
                    return -1;
                }
                for (int a = 0; a < 3; a++) {
                    tris[t].v[3 * k + a] = verts[3 * (size_t)vi + a];
                }
                for (int a = 0; a < 2; a++) {
                    tris[t].uv[2 * k + a] = have_uv ?
                        src.vertex_uv[2 * ((size_t)vert_first + vi) + a] : 0.f;
                }
            }
            tris[t].material = have_tri_mat ?
                src.triangle_material[tri_first + t] : -1;
            for (int a = 0; a < 3; a++) {
                tris[t].c[a] =
                    (tris[t].v[a] + tris[t].v[3 + a] + tris[t].v[6 + a]) / 3.f;
            }
        }

        std::vector<BvhNode> nodes(1);
        nodes[0].child[0] = nodes[0].child[1] = kNoChild;
        if (tri_count > kLeafTris) {
            buildBlas(tris, 0, tri_count, nodes, 0);
        } else if (tri_count > 0) {
            nodes[0].box[0] = inflated(boxOf(tris, 0, tri_count));
            nodes[0].child[0] = kLeafBit | ((tri_count - 1u) << 28);
        }
        if (nodes.size() > kMaxBlasNodes) {
            error = "render geometry: object " + std::to_string(obj) + " needs " +
                    std::to_string(nodes.size()) + " BVH nodes (limit " +
                    std::to_string(kMaxBlasNodes) + ")";
            return -1;
        }

        const uint32_t tri_base = (uint32_t)(out.triangleVertices.size() / 9);
        if ((uint64_t)tri_base + tri_count > 0x0FFFFFFFull) {
            error = "render geometry: more than 2^28 triangles";
            return -1;
        }
        out.objectNodeOffset[obj] = (uint32_t)out.nodes.size();
        out.objectTriangleOffset[obj] = tri_base;
        out.nodes.insert(out.nodes.end(), nodes.begin(), nodes.end());
        for (const BuildTri &t : tris) {
            out.triangleVertices.insert(out.triangleVertices.end(), t.v, t.v + 9);
            if (have_uv) {
                out.triangleUV.insert(out.triangleUV.end(), t.uv, t.uv + 6);
            }
            if (have_tri_mat) {
                out.triangleMaterial.push_back(t.material);
            }
        }
        if (tri_count > 0) {
            // (tris are in leaf order by now: ids as the trace kernel sees them)
            out.objectIsBox[obj] = meshIsItsBounds(
                tris, out.objectBoxFaces.data() + 12 * (size_t)obj) ? 1u : 0u;
            const AABB root = boxOf(tris, 0, tri_count);
            float *rb = out.objectRootBox.data() + 6 * (size_t)obj;
            rb[0] = root.pMin.x; rb[1] = root.pMin.y; rb[2] = root.pMin.z;
            rb[3] = root.pMax.x; rb[4] = root.pMax.y; rb[5] = root.pMax.z;
        }
    }
    out.objectNodeOffset[src.num_objects] = (uint32_t)out.nodes.size();
    out.objectTriangleOffset[src.num_objects] =
        (uint32_t)(out.triangleVertices.size() / 9);
    return 0;
}

void buildRenderLaunches(EcsState *state_dev, const RenderParams &params,
                         uint32_t num_worlds, uint32_t view_capacity,
                         uint32_t max_workgroups, std::vector<KernelLaunch> &out)
{
    {
        KernelLaunch k;
        k.fn = (const void *)&renderTlasBuild<kSmallTlasLeaves>;
        k.grid = dim3(num_worlds, 1, 1);
        k.block = dim3(64, 1, 1);
        k.setArgs(state_dev, params);
        k.name = "render";
        k.role = "tlas.build";
        out.push_back(k);
        k.fn = (const void *)&renderTlasBuild<(int)kMaxTlasLeaves>;
        k.role = "tlas.build.large";
        out.push_back(k);
    }
    {
        KernelLaunch k;
        // (what the geometry can ask of the shading, decided here once)
        // (materialTexture is only uploaded for scenes that have textures)
        const bool plain = params.geometry.triangleMaterial == nullptr &&
            params.geometry.materialTexture == nullptr;
        k.fn = plain ? (const void *)&renderRaycast<true> :
                       (const void *)&renderRaycast<false>;
        const uint32_t tiles_per_side = (params.resolution + 15u) / 16u;
        const uint64_t tiles =
            (uint64_t)view_capacity * tiles_per_side * tiles_per_side;
        // Workgroups walk contiguous runs of tiles and stage a world's
        // instances when they enter it.  The grid that works best is ONE
        // WORKGROUP PER VIEW (its run = the view's tiles, one staging per
        // workgroup; config 5: 1536 workgroups 4.43 ms, 4096 3.68, 8192 3.56,
        // 16384 = the views 3.42; 32768 -- half a view each, every world staged
        // twice as often -- 5.61): as many workgroups as views, within
        // [max_workgroups, 65536].
        const uint64_t want = std::min<uint64_t>(
            std::max<uint64_t>(view_capacity, max_workgroups), 65536);
        k.grid = dim3((uint32_t)std::min<uint64_t>(std::max<uint64_t>(tiles, 1),
                                                  want), 1, 1);
        k.block = dim3(256, 1, 1);
        k.setArgs(state_dev, params);
        k.name = "render";
        k.role = "raycast";
        out.push_back(k);
    }
}

}
}

// TEST INFRASTRUCTURE ONLY -- never linked into or called by the product.
//
// The reference's batch ray caster (src/mw/device/bvh_raycast.cpp: ray
// generation, two-level traversal, watertight triangle test, shading, pixel
// writes) is CUDA device code.  Nothing in it needs a GPU: this translation
// unit defines the CUDA keywords away, provides threadIdx & co as thread-local
// variables, #includes the file where it lies, and calls its kernel entry
// (bvhRaycastEntry) once per pixel.  The acceleration structures it walks are
// built here with the reference's own BVHNodeQuantized::construct
// (include/madrona/mesh_bvh.hpp:72-133) from plain triangles -- the reference
// gets them from Embree (absent) and from its device LBVH kernels; which valid
// BVH is walked changes which boxes are tested, not which triangles are hit.
//
// Behind a C ABI for tests/test_raycast_gpu.py, which feeds it the instance /
// view / light rows of the HIP backend's tables and compares images.
#include <madrona/math.hpp>
#include <madrona/sync.hpp>
#include <madrona/render/ecs.hpp>
#include <madrona/components.hpp>
#include <madrona/mesh_bvh.hpp>

#include <algorithm>
#include <bit>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <utility>
#include <vector>

// ---- CUDA, as far as bvh_raycast.cpp uses it -----------------------------------
#define __device__
#define __global__
#define __constant__
#define __shared__
#define __syncwarp()

namespace {
struct Dim3 { uint32_t x, y, z; };
}
static thread_local Dim3 threadIdx, blockIdx, blockDim, gridDim;

// A texture object = the address of one of these.  tex2D<float4> over an RGBA8
// array with wrap addressing, linear filtering, normalised coordinates and
// normalised-float reads (how the reference creates its textures,
// src/render/asset_processor.cpp:312-345), as the CUDA programming guide
// defines the fetch: xB = N x - 0.5, i = floor(xB), alpha = frac(xB) in 9-bit
// fixed point with 8 fractional bits, bilinear blend of the four wrapped
// texels.  (No CUDA texture unit exists here to pin this against: the
// definition is the oracle.)
typedef unsigned long long cudaTextureObject_t;
struct float4 { float x, y, z, w; };
struct ShimTexture {
    uint32_t width, height;
    const uint8_t *rgba8;
};
template <typename T>
static inline T tex2D(cudaTextureObject_t handle, float x, float y)
{
    const ShimTexture &tex = *(const ShimTexture *)(uintptr_t)handle;
    const int32_t width = (int32_t)tex.width, height = (int32_t)tex.height;
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
    auto texel = [&](int32_t i, int32_t j, int c) {
        return (float)tex.rgba8[4 * ((size_t)j * width + i) + c] / 255.f;
    };
    float out[3];
    for (int c = 0; c < 3; c++) {
        out[c] = (1.f - a) * (1.f - b) * texel(i0, j0, c) +
                 a * (1.f - b) * texel(i1, j0, c) +
                 (1.f - a) * b * texel(i0, j1, c) + a * b * texel(i1, j1, c);
    }
    return T { out[0], out[1], out[2], 1.f };
}

static inline float __uint_as_float(uint32_t v) { return std::bit_cast<float>(v); }
static inline uint32_t __float_as_uint(float v) { return std::bit_cast<uint32_t>(v); }
static inline int __ffs(uint32_t v) { return __builtin_ffs((int)v); }

#include REF_RAYCAST_CPP

extern "C" { BVHParams bvhParams; }

// ---- acceleration structures -------------------------------------------------------
namespace {

struct BuildItem {
    AABB box;
    int32_t leafRef;    // what a leaf child refers to (first triangle / instance)
    uint8_t triSize;    // bottom level: triangles in the leaf
};

Vector3 centre(const AABB &b) { return (b.pMin + b.pMax) * 0.5f; }

// splits [first, first + count) at the median along the widest centroid axis
uint32_t split(BuildItem *items, uint32_t first, uint32_t count)
{
    Vector3 lo = centre(items[first].box), hi = lo;
    for (uint32_t i = first + 1; i < first + count; i++) {
        Vector3 c = centre(items[i].box);
        lo = Vector3::min(lo, c);
        hi = Vector3::max(hi, c);
    }
    Vector3 ext = hi - lo;
    int axis = ext.y > ext.x ? 1 : 0;
    if (ext.z > ext[axis]) axis = 2;
    uint32_t half = count / 2;
    std::nth_element(items + first, items + first + half, items + first + count,
        [axis](const BuildItem &a, const BuildItem &b) {
            return centre(a.box)[axis] < centre(b.box)[axis];
        });
    return half;
}

// 4-wide node over items [first, first + count), count >= 1; returns its index
uint32_t buildNode(std::vector<QBVHNode> &nodes, BuildItem *items,
                   uint32_t first, uint32_t count)
{
    const uint32_t self = (uint32_t)nodes.size();
    nodes.push_back(QBVHNode {});

    // up to four groups: halves of halves
    uint32_t group_first[4], group_count[4], num_groups = 0;
    if (count <= 4) {
        for (uint32_t i = 0; i < count; i++) {
            group_first[num_groups] = first + i;
            group_count[num_groups++] = 1;
        }
    } else {
        uint32_t half = split(items, first, count);
        uint32_t parts[2][2] = { { first, half }, { first + half, count - half } };
        for (auto &p : parts) {
            uint32_t q = split(items, p[0], p[1]);
            group_first[num_groups] = p[0];
            group_count[num_groups++] = q;
            group_first[num_groups] = p[0] + q;
            group_count[num_groups++] = p[1] - q;
        }
    }

    AABB child_boxes[4];
    int32_t child_refs[4];
    uint8_t tri_sizes[4] = { 0, 0, 0, 0 };
    for (uint32_t g = 0; g < num_groups; g++) {
        AABB box = items[group_first[g]].box;
        for (uint32_t i = 1; i < group_count[g]; i++) {
            box = AABB::merge(box, items[group_first[g] + i].box);
        }
        child_boxes[g] = box;
        if (group_count[g] == 1) {
            child_refs[g] = -(items[group_first[g]].leafRef + 1);
            tri_sizes[g] = items[group_first[g]].triSize;
        } else {
            uint32_t child = buildNode(nodes, items, group_first[g], group_count[g]);
            child_refs[g] = (int32_t)child + 1;
        }
    }

    if (getenv("RAYCAST_REF_DEBUG") != nullptr) {
        for (uint32_t g = 0; g < num_groups; g++) {
            fprintf(stderr, "node %u group %u: first %u count %u ref %d box [%f %f %f | %f %f %f]\n",
                    self, g, group_first[g], group_count[g], child_refs[g],
                    child_boxes[g].pMin.x, child_boxes[g].pMin.y, child_boxes[g].pMin.z,
                    child_boxes[g].pMax.x, child_boxes[g].pMax.y, child_boxes[g].pMax.z);
        }
    }
    QBVHNode node = QBVHNode::construct(num_groups, child_boxes, child_refs);
    for (int i = 0; i < 4; i++) node.triSize[i] = tri_sizes[i];
    nodes[self] = node;
    return self;
}

// (a box flat along an axis has no exponent to quantise with)
AABB inflate(AABB b)
{
    const Vector3 eps { 1e-3f, 1e-3f, 1e-3f };
    return AABB { b.pMin - eps, b.pMax + eps };
}

// self-check: every child box, as the traversal will
// dequantise it, contains the triangles below it
bool validate(const std::vector<QBVHNode> &nodes, uint32_t idx,
              const std::vector<MeshBVH::BVHVertex> &verts,
              std::vector<uint32_t> &tris_out)
{
    const QBVHNode &node = nodes[idx];
    bool ok = true;
    for (int c = 0; c < 4; c++) {
        if (!node.hasChild(c)) continue;
        std::vector<uint32_t> below;
        if (node.isLeaf(c)) {
            uint32_t first = node.childrenIdx[c] & ~0x8000'0000u;
            for (uint32_t t = 0; t < node.triSize[c]; t++) below.push_back(first + t);
        } else {
            ok = validate(nodes, node.childrenIdx[c], verts, below) && ok;
        }
        AABB box = node.convertToAABB(c);
        for (uint32_t t : below) {
            for (int k = 0; k < 3; k++) {
                Vector3 p = verts[3 * t + k].pos;
                if (p.x < box.pMin.x || p.y < box.pMin.y || p.z < box.pMin.z ||
                        p.x > box.pMax.x || p.y > box.pMax.y || p.z > box.pMax.z) {
                    fprintf(stderr, "node %u child %d: triangle %u outside "
                            "(%f %f %f) not in [%f %f %f | %f %f %f] exp %d %d %d\n",
                            idx, c, t, p.x, p.y, p.z, box.pMin.x, box.pMin.y,
                            box.pMin.z, box.pMax.x, box.pMax.y, box.pMax.z,
                            node.expX, node.expY, node.expZ);
                    ok = false;
                }
            }
            tris_out.push_back(t);
        }
    }
    return ok;
}

struct ObjectBVH {
    std::vector<QBVHNode> nodes;
    std::vector<MeshBVH::BVHVertex> vertices;
    std::vector<MeshBVH::LeafMaterial> leafMats;
    AABB root;
};

}

extern "C" __attribute__((visibility("default")))
int raycast_ref_render(
    uint32_t num_objects, const float *vertices, const uint32_t *indices,
    const uint32_t *vertex_offsets, const uint32_t *triangle_offsets,
    const int32_t *object_materials, uint32_t num_materials,
    const float *material_colors,
    uint32_t num_worlds,
    const void *instances, const int32_t *instance_offsets,
    const int32_t *instance_counts,
    const void *views, uint32_t num_views,
    const void *lights, const int32_t *light_offsets, const int32_t *light_counts,
    uint32_t resolution, uint32_t rgbd, uint32_t num_threads,
    uint8_t *rgb_out, float *depth_out,
    // optional (NULL / 0): uv per vertex, material per triangle (for objects
    // whose object material is -1), texture id per material, RGBA8 textures
    const float *vertex_uvs, const int32_t *triangle_materials,
    const int32_t *material_textures, uint32_t num_textures,
    const uint32_t *texture_dims, const uint8_t *texels)
{
    static_assert(sizeof(InstanceData) == 64 &&
                  sizeof(PerspectiveCameraData) == 48 && sizeof(LightDesc) == 40);

    // ---- bottom level: leaves of <= 2 consecutive triangles -----------------------
    std::vector<ObjectBVH> objects(num_objects);
    std::vector<MeshBVH> bvhs(num_objects);
    for (uint32_t o = 0; o < num_objects; o++) {
        ObjectBVH &obj = objects[o];
        const float *verts = vertices + 3 * (size_t)vertex_offsets[o];
        const uint32_t tri_first = triangle_offsets[o];
        const uint32_t num_tris = triangle_offsets[o + 1] - tri_first;
        if (num_tris == 0) return -1;

        std::vector<AABB> tri_boxes(num_tris);
        for (uint32_t t = 0; t < num_tris; t++) {
            AABB box {};
            for (int k = 0; k < 3; k++) {
                const uint32_t vi = indices[3 * (size_t)(tri_first + t) + k];
                const float *p = verts + 3 * (size_t)vi;
                Vector3 pos { p[0], p[1], p[2] };
                Vector2 uv { 0.f, 0.f };
                if (vertex_uvs != nullptr) {
                    const float *q = vertex_uvs + 2 * ((size_t)vertex_offsets[o] + vi);
                    uv = Vector2 { q[0], q[1] };
                }
                obj.vertices.push_back({ pos, uv });
                // (AABB::expand on AABB::invalid() only ever moves one bound
                // per axis and point: start from the first vertex)
                box = k == 0 ? AABB::point(pos) : AABB::merge(box, AABB::point(pos));
            }
            tri_boxes[t] = inflate(box);
            obj.leafMats.push_back({ { { triangle_materials != nullptr ?
                triangle_materials[tri_first + t] : -1 } } });
        }
        std::vector<BuildItem> items;
        for (uint32_t t = 0; t < num_tris; t += 2) {
            BuildItem item;
            item.box = tri_boxes[t];
            item.leafRef = (int32_t)t;
            item.triSize = 1;
            if (t + 1 < num_tris) {
                item.box = AABB::merge(item.box, tri_boxes[t + 1]);
                item.triSize = 2;
            }
            items.push_back(item);
        }
        obj.root = items[0].box;
        for (const BuildItem &it : items) obj.root = AABB::merge(obj.root, it.box);
        buildNode(obj.nodes, items.data(), 0, (uint32_t)items.size());
        {
            std::vector<uint32_t> all;
            if (!validate(obj.nodes, 0, obj.vertices, all) || all.size() != num_tris) {
                fprintf(stderr, "object %u: bad BVH (%zu of %u triangles)\n", o,
                        all.size(), num_tris);
                return -5;
            }
        }

        MeshBVH &bvh = bvhs[o];
        bvh.nodes = obj.nodes.data();
        bvh.leafMats = obj.leafMats.data();
        bvh.vertices = obj.vertices.data();
        bvh.rootAABB = obj.root;
        bvh.numNodes = (uint32_t)obj.nodes.size();
        bvh.numLeaves = num_tris;
        bvh.numVerts = (uint32_t)obj.vertices.size();
        bvh.materialIDX = object_materials != nullptr ? object_materials[o] : -1;
        bvh.magic = 0;
    }

    std::vector<Material> materials(num_materials + 1);
    for (uint32_t m = 0; m < num_materials; m++) {
        materials[m].color = Vector4 { material_colors[3 * m],
            material_colors[3 * m + 1], material_colors[3 * m + 2], 1.f };
        materials[m].textureIdx =
            material_textures != nullptr ? material_textures[m] : -1;
        materials[m].roughness = 0.f;
        materials[m].metalness = 0.f;
    }
    std::vector<ShimTexture> shim_textures(num_textures);
    std::vector<cudaTextureObject_t> texture_objects(num_textures);
    {
        size_t at = 0;
        for (uint32_t t = 0; t < num_textures; t++) {
            shim_textures[t] = { texture_dims[2 * t], texture_dims[2 * t + 1],
                                 texels + at };
            at += (size_t)texture_dims[2 * t] * texture_dims[2 * t + 1] * 4;
            texture_objects[t] = (cudaTextureObject_t)(uintptr_t)&shim_textures[t];
        }
    }

    // ---- top level: one tree per world, in the world's slice of the node array --------
    const InstanceData *inst = (const InstanceData *)instances;
    uint32_t total_instances = 0;
    for (uint32_t w = 0; w < num_worlds; w++) {
        if (instance_counts[w] <= 0) return -2;     // (the reference reads a root)
        total_instances = std::max(total_instances,
            (uint32_t)(instance_offsets[w] + instance_counts[w]));
    }
    std::vector<QBVHNode> tlas(total_instances);
    for (uint32_t w = 0; w < num_worlds; w++) {
        const uint32_t first = (uint32_t)instance_offsets[w];
        const uint32_t n = (uint32_t)instance_counts[w];
        std::vector<BuildItem> items(n);
        for (uint32_t i = 0; i < n; i++) {
            const InstanceData &d = inst[first + i];
            if (d.objectID < 0 || (uint32_t)d.objectID >= num_objects) return -3;
            items[i].box = inflate(objects[d.objectID].root.applyTRS(
                d.position, d.rotation, d.scale));
            items[i].leafRef = (int32_t)i;
            items[i].triSize = 0;
        }
        std::vector<QBVHNode> nodes;
        buildNode(nodes, items.data(), 0, n);
        if (nodes.size() > n) return -4;
        std::copy(nodes.begin(), nodes.end(), tlas.begin() + first);
    }

    // (its atomics have no default constructor: zeroed storage)
    alignas(BVHInternalData) static thread_local unsigned char
        internal_storage[sizeof(BVHInternalData)];
    memset(internal_storage, 0, sizeof(internal_storage));
    BVHInternalData &internal = *(BVHInternalData *)internal_storage;
    internal.traversalNodes = tlas.data();
    internal.numViews = num_views;

    bvhParams = BVHParams {};
    bvhParams.numWorlds = num_worlds;
    bvhParams.instances = (InstanceData *)instances;
    bvhParams.views = (PerspectiveCameraData *)views;
    bvhParams.lights = (LightDesc *)lights;
    bvhParams.instanceOffsets = (int32_t *)instance_offsets;
    bvhParams.instanceCounts = (int32_t *)instance_counts;
    bvhParams.lightOffsets = (int32_t *)light_offsets;
    bvhParams.lightCounts = (int32_t *)light_counts;
    bvhParams.internalData = &internal;
    bvhParams.bvhs = bvhs.data();
    bvhParams.rgbOutput = rgb_out;
    bvhParams.depthOutput = depth_out;
    bvhParams.renderOutputResolution = resolution;
    bvhParams.raycastRGBD = rgbd;
    bvhParams.materials = materials.data();
    bvhParams.textures = texture_objects.data();
    bvhParams.nearSphere = 0.f;

    // ---- the kernel: one "thread" per pixel, each looping over all views ---------
    // (launch shape of cuda_exec.cpp:2664-2683 collapsed to blockDim = 1:
    // pixel = blockIdx.y/z * blockDim.x + threadIdx.x/y; gridDim.x = resident
    // view slots)
    num_threads = std::max(num_threads, 1u);
    std::vector<std::thread> workers;
    for (uint32_t t = 0; t < num_threads; t++) {
        workers.emplace_back([=]() {
            blockDim = { 1, 1, 1 };
            gridDim = { 1, resolution, resolution };
            threadIdx = { 0, 0, 0 };
            for (uint32_t py = t; py < resolution; py += num_threads) {
                for (uint32_t px = 0; px < resolution; px++) {
                    blockIdx = { 0, px, py };
                    bvhRaycastEntry();
                }
            }
        });
    }
    for (auto &w : workers) w.join();
    return 0;
}

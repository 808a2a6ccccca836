// Batch ray caster: what runtime.hip and render_raycast.hip share.
#pragma once

#include "runtime_internal.hpp"

#include <madrona/math.hpp>

#include <vector>

namespace madrona {
namespace mwhip {

// most instances a world may hold for the ray caster (LDS budget of the
// one-wavefront TLAS build: 48 B per leaf)
constexpr uint32_t kMaxTlasLeaves = 1024;
// bottom-level nodes per object (traversal stack entries are 16 bit)
constexpr uint32_t kMaxBlasNodes = 65536;

// Binary BVH node of both levels: the boxes of BOTH children, so that a leaf is
// only entered after its own box was hit.  child = index of an internal node of
// the same tree, or 0x80000000 | leaf, or 0xFFFFFFFF: no such child (a tree
// with one leaf).  Top level: leaf = instance index inside the world.  Bottom
// level: leaf = first triangle (bits 0..27) | (triangles - 1) << 28.
struct alignas(16) BvhNode {
    math::AABB box[2];
    uint32_t child[2];
    uint32_t pad[2];
};
static_assert(sizeof(BvhNode) == 64);

// An instance as the trace loop wants it: the inverse rotation and inverse
// scale of the world -> object map (applied in the reference's order, so that
// object-space rays agree with its to the bit), and where the object's
// bottom-level tree lives.  Written by the TLAS build (one per row of the
// renderable table).
struct alignas(16) PreparedInstance {
    math::Quat invRotation;
    math::Diag3x3 invScale;
    math::Vector3 position;
    uint32_t objectID;
    uint32_t isBox;             // the object is an axis-aligned box (objectBounds)
    uint32_t nodeOffset;        // first node of the object's tree
    uint32_t triangleOffset;    // first triangle of the object
    int32_t valid;              // 0: no volume (all-zero scale) or no such object
    uint32_t pad;
};

struct RenderGeometryDev {
    uint32_t numObjects;
    uint32_t numMaterials;
    const BvhNode *nodes;
    const math::Vector3 *triangleVertices;  // 3 per triangle, leaf order
    const uint32_t *objectNodeOffset;       // [numObjects + 1]
    const uint32_t *objectTriangleOffset;   // [numObjects + 1]
    const int32_t *objectMaterial;          // [numObjects], -1: none (white)
    const float *objectBounds;              // [numObjects][6] object-space min, max
    const uint32_t *objectIsBox;            // [numObjects] mesh == its bounds, faces outward
    const uint32_t *objectBoxFaces;         // [numObjects][12] the two triangles of face
                                            // axis * 2 + (max side), object-local ids
    const float *materialColor;             // rgb per material
    // per-triangle materials / textures (all may be null: untextured objects
    // with one material each take none of this)
    const float *triangleUV;                // 6 per triangle, leaf order
    const int32_t *triangleMaterial;        // per triangle, leaf order
    const int32_t *materialTexture;         // per material, -1: none
    const uint32_t *textureInfo;            // [numTextures][4] first texel, width, height, -
    const uint32_t *texels;                 // RGBA8
    uint32_t numTextures;
    uint32_t pad_;
};

struct RenderGeometryHost {
    uint32_t numObjects = 0;
    uint32_t numMaterials = 0;
    std::vector<BvhNode> nodes;
    std::vector<float> triangleVertices;
    std::vector<uint32_t> objectNodeOffset;
    std::vector<uint32_t> objectTriangleOffset;
    std::vector<int32_t> objectMaterial;
    std::vector<float> materialColor;
    std::vector<float> objectRootBox;       // 6 per object: what TLBVHNode uses
    std::vector<uint32_t> objectIsBox;      // the mesh IS that box (12 outward triangles)
    std::vector<uint32_t> objectBoxFaces;   // 12 per object (leaf-order triangle ids)
    std::vector<float> triangleUV;          // empty: no uvs given
    std::vector<int32_t> triangleMaterial;  // empty: no per-triangle materials
    std::vector<int32_t> materialTexture;   // empty: no textures
    std::vector<uint32_t> textureInfo;
    std::vector<uint32_t> texels;
};

struct RenderParams {
    mwhip_render_layout layout;
    uint32_t instanceColumn, mortonColumn, tlbvhColumn;
    uint32_t cameraColumn, lightColumn, rgbColumn, depthColumn;
    uint32_t resolution;
    uint32_t rgbd;
    uint32_t pad_[3];
    BvhNode *tlasNodes;         // one slot per row of the renderable table
    PreparedInstance *prepared; // likewise
    RenderGeometryDev geometry;
};

// host: one median-split BVH per object; -1 + mwhip_last_error on bad input
int buildRenderGeometry(const mwhip_render_geometry &src, RenderGeometryHost &out,
                        std::string &error);

// the launches of one render pass: TLAS build (a wavefront per world), then the
// ray caster (a 16 x 16 tile of one view per workgroup)
void buildRenderLaunches(EcsState *state_dev, const RenderParams &params,
                         uint32_t num_worlds, uint32_t view_capacity,
                         uint32_t max_workgroups, std::vector<KernelLaunch> &out);

}
}

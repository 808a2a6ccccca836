// What a renderer's asset processor hands to the executor for the batch ray
// caster (reference include/madrona/render/cuda_batch_render_assets.hpp:8-28),
// field for field.  On this backend the buffers may live in host or device
// memory (the executor copies them while it is constructed), and a texture
// object is the address of a render::TextureRGBA8 describing an RGBA8 image in
// host memory -- what the reference uploads into a cudaArray of uchar4 and
// samples with wrap addressing, linear filtering and normalised coordinates
// (src/render/asset_processor.cpp:312-345).
#pragma once

#include <madrona/mesh_bvh.hpp>

namespace madrona::render {

struct TextureRGBA8 {
    uint32_t width;
    uint32_t height;
    const uint8_t *pixels;      // width * height * 4 bytes, row 0 first
};

}

// (the two CUDA handle types the reference's struct names)
using cudaTextureObject_t = unsigned long long;
using cudaArray_t = void *;

namespace madrona::render {

struct MeshBVHData {
    QBVHNode *nodes;
    uint64_t numNodes;

    MeshBVH::LeafMaterial *leafMaterial;
    uint64_t numLeaves;

    MeshBVH::BVHVertex *vertices;
    uint64_t numVerts;

    MeshBVH *meshBVHs;
    uint64_t numBVHs;
};

struct MaterialData {
    // array of texture objects: (cudaTextureObject_t)(uintptr_t)&TextureRGBA8
    cudaTextureObject_t *textures;
    uint32_t numTextureBuffers;
    cudaArray_t *textureBuffers;    // unused here
    Material *materials;
};

}

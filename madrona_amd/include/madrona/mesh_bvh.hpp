// Data layout of the bottom-level mesh BVHs a renderer hands to the batch ray
// caster (reference include/madrona/mesh_bvh.hpp:20-47, 146-178, 294-307): the
// 4-wide quantised node, the per-triangle material record, the de-indexed
// vertex with its uv, the Material record, and the MeshBVH header.  Layouts
// only -- this backend reads them as raw memory (madrona/mw_gpu.hpp turns them
// into the triangles its own bottom-level trees are built from); the
// reference's traversal / construction methods are not part of it.
#pragma once

#include <madrona/types.hpp>
#include <madrona/math.hpp>
#include <madrona/geo.hpp>

#define MADRONA_BVH_WIDTH 4

#ifndef MADRONA_BLAS_LEAF_WIDTH
#define MADRONA_BLAS_LEAF_WIDTH 2
#endif

namespace madrona {

template <typename NodeIndex, int Width>
struct BVHNodeQuantized {
    using NodeIndexT = NodeIndex;
    static constexpr int NodeWidth = Width;

    math::Vector3 minPoint;
    int8_t expX, expY, expZ;
    uint8_t numChildren;
    // bottom level: triangles of each leaf child
    uint8_t triSize[Width];
    // child boxes, quantised to 8 bits against minPoint / 2^exp
    uint8_t qMinX[Width], qMinY[Width], qMinZ[Width];
    uint8_t qMaxX[Width], qMaxY[Width], qMaxZ[Width];
    // internal child: its node index; leaf: 0x80000000 | first triangle;
    // 0xFFFFFFFF: no such child
    NodeIndex childrenIdx[Width];

    bool hasChild(uint32_t i) const { return childrenIdx[i] != 0xFFFF'FFFFu; }
    bool isLeaf(uint32_t i) const { return (childrenIdx[i] & 0x8000'0000u) != 0u; }
    uint32_t leafIDX(uint32_t i) const { return childrenIdx[i] & ~0x8000'0000u; }
};

using QBVHNode = BVHNodeQuantized<uint32_t, MADRONA_BVH_WIDTH>;
static_assert(sizeof(QBVHNode) == 12 + 4 + 4 + 24 + 16);

struct Material {
    math::Vector4 color;
    int32_t textureIdx;     // -1: untextured
    float roughness;
    float metalness;
};

struct MeshBVH {
    static constexpr inline CountT numTrisPerLeaf = MADRONA_BLAS_LEAF_WIDTH;
    static constexpr inline CountT nodeWidth = MADRONA_BVH_WIDTH;

    struct BVHMaterial {
        int32_t matIDX;
    };

    struct LeafMaterial {
        BVHMaterial material[1];
    };

    struct BVHVertex {
        math::Vector3 pos;
        math::Vector2 uv;
    };

    QBVHNode *nodes;
    LeafMaterial *leafMats;     // per triangle, read when materialIDX == -1
    BVHVertex *vertices;        // 3 per triangle, leaf order

    math::AABB rootAABB;
    uint32_t numNodes;
    uint32_t numLeaves;
    uint32_t numVerts;

    int32_t materialIDX;        // the whole mesh's material, or -1: per triangle

    uint32_t magic;
};

}

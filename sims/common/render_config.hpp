// HIP backend: a simulator's MeshSet as the executor's render configuration.
#pragma once

#include "mesh_set.hpp"

#include <madrona/mw_gpu.hpp>

#include <cstdint>
#include <vector>

namespace simmgr {

// geometry + materials of a MeshSet as the executor's render configuration
static inline void meshesToRenderConfig(const simmesh::MeshSet &m,
                                        madrona::CudaBatchRenderConfig &cfg)
{
    using madrona::CountT;
    using madrona::Span;
    using madrona::math::Vector3;
    using madrona::math::Vector2;
    cfg.geoTriangles.vertices = Span<const Vector3>(
        (const Vector3 *)m.vertices.data(), (CountT)(m.vertices.size() / 3));
    cfg.geoTriangles.indices =
        Span<const uint32_t>(m.indices.data(), (CountT)m.indices.size());
    cfg.geoTriangles.objectVertexOffsets = Span<const uint32_t>(
        m.vertexOffsets.data(), (CountT)m.vertexOffsets.size());
    cfg.geoTriangles.objectTriangleOffsets = Span<const uint32_t>(
        m.triangleOffsets.data(), (CountT)m.triangleOffsets.size());
    cfg.geoTriangles.vertexUVs = Span<const Vector2>(
        (const Vector2 *)m.vertexUVs.data(), (CountT)(m.vertexUVs.size() / 2));
    cfg.geoTriangles.triangleMaterials = Span<const int32_t>(
        m.triangleMaterials.data(), (CountT)m.triangleMaterials.size());
    cfg.triangleMaterials.materialColors = Span<const Vector3>(
        (const Vector3 *)m.materialColors.data(),
        (CountT)(m.materialColors.size() / 3));
    cfg.triangleMaterials.objectMaterials = Span<const int32_t>(
        m.objectMaterials.data(), (CountT)m.objectMaterials.size());
    static thread_local std::vector<madrona::render::TextureRGBA8> textures;
    textures.clear();
    for (const simmesh::MeshSet::Texture &t : m.textures) {
        textures.push_back({ t.width, t.height, t.rgba8.data() });
    }
    if (!textures.empty()) {
        cfg.triangleMaterials.materialTextures = Span<const int32_t>(
            m.materialTextures.data(), (CountT)m.materialTextures.size());
        cfg.triangleMaterials.textures =
            Span<const madrona::render::TextureRGBA8>(
                textures.data(), (CountT)textures.size());
    }
}

// The same MeshSet in the form a renderer's asset processor hands over
// (reference render/cuda_batch_render_assets.hpp): 4-wide quantised mesh BVHs
// over de-indexed vertices, per-triangle materials, Material records, texture
// objects.  The trees built here are valid but naive (chains of nodes with up
// to four leaves of MADRONA_BLAS_LEAF_WIDTH triangles, every child box the
// whole mesh): the executor only reads their leaves.  Storage lives as long as
// the returned object.
struct ReferenceFormAssets {
    std::vector<madrona::QBVHNode> nodes;
    std::vector<madrona::MeshBVH::LeafMaterial> leafMaterials;
    std::vector<madrona::MeshBVH::BVHVertex> vertices;
    std::vector<madrona::MeshBVH> meshes;
    std::vector<madrona::Material> materials;
    std::vector<madrona::render::TextureRGBA8> textureDescs;
    std::vector<cudaTextureObject_t> textureObjects;
};

static inline ReferenceFormAssets *meshesToReferenceAssets(
    const simmesh::MeshSet &m, madrona::CudaBatchRenderConfig &cfg)
{
    using namespace madrona;
    auto *out = new ReferenceFormAssets();
    const uint32_t num_objects = m.numObjects();
    constexpr uint32_t leaf_tris = (uint32_t)MeshBVH::numTrisPerLeaf;

    // first pass: sizes (the arrays must not move once meshes point into them)
    size_t total_tris = m.indices.size() / 3, total_nodes = 0;
    for (uint32_t o = 0; o < num_objects; o++) {
        const uint32_t tris = m.triangleOffsets[o + 1] - m.triangleOffsets[o];
        const uint32_t leaves = (tris + leaf_tris - 1) / leaf_tris;
        // a chain: every node holds three leaves and a link, the last one four
        total_nodes += leaves <= 4 ? 1 : (leaves - 4 + 2) / 3 + 1;
    }
    out->nodes.reserve(total_nodes);
    out->vertices.reserve(total_tris * 3);
    out->leafMaterials.reserve(total_tris);
    out->meshes.resize(num_objects);

    for (uint32_t o = 0; o < num_objects; o++) {
        const uint32_t tri_first = m.triangleOffsets[o];
        const uint32_t tris = m.triangleOffsets[o + 1] - tri_first;
        const uint32_t vert_first = m.vertexOffsets[o];
        const size_t node_first = out->nodes.size();
        const size_t corner_first = out->vertices.size();
        const size_t mat_first = out->leafMaterials.size();

        for (uint32_t t = 0; t < tris; t++) {
            for (int k = 0; k < 3; k++) {
                const uint32_t vi = vert_first + m.indices[3 * (size_t)(tri_first + t) + k];
                out->vertices.push_back({
                    { m.vertices[3 * vi], m.vertices[3 * vi + 1], m.vertices[3 * vi + 2] },
                    { m.vertexUVs[2 * vi], m.vertexUVs[2 * vi + 1] } });
            }
            out->leafMaterials.push_back({ { { m.triangleMaterials[tri_first + t] } } });
        }

        const float *rb = m.rootAABBs.data() + 6 * (size_t)o;
        const uint32_t leaves = (tris + leaf_tris - 1) / leaf_tris;
        uint32_t leaf = 0;
        do {
            QBVHNode node {};
            node.minPoint = { rb[0], rb[1], rb[2] };
            node.expX = node.expY = node.expZ = 0;
            const uint32_t left = leaves - leaf;
            const uint32_t here = left <= 4 ? left : 3;
            uint32_t c = 0;
            for (; c < here; c++, leaf++) {
                const uint32_t first_tri = leaf * leaf_tris;
                node.childrenIdx[c] = 0x8000'0000u | first_tri;
                node.triSize[c] = (uint8_t)(tris - first_tri < leaf_tris ?
                                            tris - first_tri : leaf_tris);
                node.qMaxX[c] = node.qMaxY[c] = node.qMaxZ[c] = 255;
            }
            if (left > 4) {
                node.childrenIdx[c] =
                    (uint32_t)(out->nodes.size() - node_first) + 1u;   // next in chain
                node.qMaxX[c] = node.qMaxY[c] = node.qMaxZ[c] = 255;
                c++;
            }
            node.numChildren = (uint8_t)c;
            for (; c < 4; c++) node.childrenIdx[c] = 0xFFFF'FFFFu;
            out->nodes.push_back(node);
        } while (leaf < leaves);

        MeshBVH &mesh = out->meshes[o];
        mesh.nodes = out->nodes.data() + node_first;
        mesh.leafMats = out->leafMaterials.data() + mat_first;
        mesh.vertices = out->vertices.data() + corner_first;
        mesh.rootAABB = { { rb[0], rb[1], rb[2] }, { rb[3], rb[4], rb[5] } };
        mesh.numNodes = (uint32_t)(out->nodes.size() - node_first);
        mesh.numLeaves = leaves;
        mesh.numVerts = tris * 3;
        mesh.materialIDX = m.objectMaterials[o];
        mesh.magic = 0;
    }

    for (size_t i = 0; i < m.materialColors.size() / 3; i++) {
        out->materials.push_back(Material {
            { m.materialColors[3 * i], m.materialColors[3 * i + 1],
              m.materialColors[3 * i + 2], 1.f },
            m.materialTextures[i], 0.5f, 0.f });
    }
    out->textureDescs.reserve(m.textures.size());
    for (const simmesh::MeshSet::Texture &t : m.textures) {
        out->textureDescs.push_back({ t.width, t.height, t.rgba8.data() });
        out->textureObjects.push_back(
            (cudaTextureObject_t)(uintptr_t)&out->textureDescs.back());
    }

    cfg.geoBVHData.nodes = out->nodes.data();
    cfg.geoBVHData.numNodes = out->nodes.size();
    cfg.geoBVHData.leafMaterial = out->leafMaterials.data();
    cfg.geoBVHData.numLeaves = out->leafMaterials.size();
    cfg.geoBVHData.vertices = out->vertices.data();
    cfg.geoBVHData.numVerts = out->vertices.size();
    cfg.geoBVHData.meshBVHs = out->meshes.data();
    cfg.geoBVHData.numBVHs = out->meshes.size();
    cfg.materialData.textures = out->textureObjects.data();
    cfg.materialData.numTextureBuffers = (uint32_t)out->textureObjects.size();
    cfg.materialData.textureBuffers = nullptr;
    cfg.materialData.materials = out->materials.data();
    cfg.numMaterials = (uint32_t)out->materials.size();
    return out;
}

}

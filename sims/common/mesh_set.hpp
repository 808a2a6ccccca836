// Triangle meshes + materials a synthetic simulator hands to the batch ray
// caster (no asset importers here: shapes are generated).  One entry per
// object id, in order.
#pragma once

#include <cmath>
#include <cstdint>
#include <vector>

namespace simmesh {

struct MeshSet {
    std::vector<float> vertices;                    // xyz, all objects
    std::vector<uint32_t> indices;                  // object-local, 3 per triangle
    std::vector<uint32_t> vertexOffsets { 0 };      // [objects + 1]
    std::vector<uint32_t> triangleOffsets { 0 };    // [objects + 1]
    std::vector<float> materialColors;              // rgb per material
    std::vector<int32_t> objectMaterials;           // per object, -1: per triangle / none
    std::vector<float> rootAABBs;                   // 6 per object
    // per-triangle materials and textures (reference MeshBVH::leafMats,
    // BVHVertex::uv, Material::textureIdx): uvs per vertex (0, 0 unless set with
    // uv()), one material per triangle (read for objects whose material is -1)
    std::vector<float> vertexUVs;                   // 2 per vertex
    std::vector<int32_t> triangleMaterials;         // per triangle
    std::vector<int32_t> materialTextures;          // per material, -1: none
    struct Texture {
        uint32_t width, height;
        std::vector<uint8_t> rgba8;
    };
    std::vector<Texture> textures;
    int32_t currentTriangleMaterial = -1;           // what tri() tags triangles with

    uint32_t numObjects() const { return (uint32_t)triangleOffsets.size() - 1u; }

    uint32_t vert(float x, float y, float z)
    {
        vertices.insert(vertices.end(), { x, y, z });
        vertexUVs.insert(vertexUVs.end(), { 0.f, 0.f });
        return (uint32_t)(vertices.size() / 3) - vertexOffsets.back() - 1u;
    }
    // uv of the vertex added last
    void uv(float u, float v)
    {
        vertexUVs[vertexUVs.size() - 2] = u;
        vertexUVs[vertexUVs.size() - 1] = v;
    }
    void tri(uint32_t a, uint32_t b, uint32_t c)
    {
        indices.insert(indices.end(), { a, b, c });
        triangleMaterials.push_back(currentTriangleMaterial);
    }
    void quad(uint32_t a, uint32_t b, uint32_t c, uint32_t d)
    {
        tri(a, b, c);
        tri(a, c, d);
    }

    // closes the object started after the previous endObject()
    void endObject(int32_t material)
    {
        float lo[3] = { 3e38f, 3e38f, 3e38f }, hi[3] = { -3e38f, -3e38f, -3e38f };
        for (size_t v = vertexOffsets.back(); v < vertices.size() / 3; v++) {
            for (int a = 0; a < 3; a++) {
                lo[a] = std::fmin(lo[a], vertices[3 * v + a]);
                hi[a] = std::fmax(hi[a], vertices[3 * v + a]);
            }
        }
        rootAABBs.insert(rootAABBs.end(), { lo[0], lo[1], lo[2], hi[0], hi[1], hi[2] });
        vertexOffsets.push_back((uint32_t)(vertices.size() / 3));
        triangleOffsets.push_back((uint32_t)(indices.size() / 3));
        objectMaterials.push_back(material);
    }

    int32_t material(float r, float g, float b, int32_t texture = -1)
    {
        materialColors.insert(materialColors.end(), { r, g, b });
        materialTextures.push_back(texture);
        return (int32_t)(materialColors.size() / 3) - 1;
    }

    // a width x height checker of two colours with a gradient over it (so that
    // neighbouring texels differ and filtering shows)
    int32_t checkerTexture(uint32_t width, uint32_t height, uint32_t cell)
    {
        Texture t { width, height, std::vector<uint8_t>((size_t)width * height * 4) };
        for (uint32_t y = 0; y < height; y++) {
            for (uint32_t x = 0; x < width; x++) {
                const bool on = ((x / cell) + (y / cell)) % 2u == 0u;
                uint8_t *px = t.rgba8.data() + 4 * ((size_t)y * width + x);
                px[0] = (uint8_t)(on ? 230 : 40 + (200 * x) / width);
                px[1] = (uint8_t)(on ? 60 + (150 * y) / height : 220);
                px[2] = (uint8_t)(on ? 30 : 120);
                px[3] = 255;
            }
        }
        textures.push_back(std::move(t));
        return (int32_t)textures.size() - 1;
    }

    void box(float x0, float y0, float z0, float x1, float y1, float z1)
    {
        const uint32_t base = vert(x0, y0, z0);
        vert(x1, y0, z0); vert(x0, y1, z0); vert(x1, y1, z0);
        vert(x0, y0, z1); vert(x1, y0, z1); vert(x0, y1, z1); vert(x1, y1, z1);
        const uint32_t q[6][4] = { { 0, 2, 3, 1 }, { 4, 5, 7, 6 }, { 0, 1, 5, 4 },
                                   { 2, 6, 7, 3 }, { 0, 4, 6, 2 }, { 1, 3, 7, 5 } };
        for (auto &f : q) quad(base + f[0], base + f[1], base + f[2], base + f[3]);
    }

    void ellipsoid(float cx, float cy, float cz, float rx, float ry, float rz,
                   int slices, int stacks)
    {
        const uint32_t south = vert(cx, cy, cz - rz);
        for (int st = 1; st < stacks; st++) {
            const float phi = 3.14159265f * (float)st / (float)stacks;
            for (int sl = 0; sl < slices; sl++) {
                const float th = 6.2831853f * (float)sl / (float)slices;
                vert(cx + rx * sinf(phi) * cosf(th), cy + ry * sinf(phi) * sinf(th),
                     cz - rz * cosf(phi));
            }
        }
        const uint32_t north = vert(cx, cy, cz + rz);
        auto ring = [&](int st, int sl) {
            return south + 1u + (uint32_t)((st - 1) * slices + sl % slices);
        };
        for (int sl = 0; sl < slices; sl++) {
            tri(south, ring(1, sl + 1), ring(1, sl));
            tri(north, ring(stacks - 1, sl), ring(stacks - 1, sl + 1));
            for (int st = 1; st < stacks - 1; st++) {
                quad(ring(st, sl), ring(st, sl + 1), ring(st + 1, sl + 1),
                     ring(st + 1, sl));
            }
        }
    }

    void cylinder(float radius, float z0, float z1, int segs)
    {
        const uint32_t lo = vert(0.f, 0.f, z0), hi = vert(0.f, 0.f, z1);
        for (int i = 0; i < segs; i++) {
            const float th = 6.2831853f * (float)i / (float)segs;
            vert(radius * cosf(th), radius * sinf(th), z0);
            vert(radius * cosf(th), radius * sinf(th), z1);
        }
        for (int i = 0; i < segs; i++) {
            const uint32_t a = hi + 1u + 2u * (uint32_t)i;
            const uint32_t b = hi + 1u + 2u * (uint32_t)((i + 1) % segs);
            tri(lo, b, a);
            tri(hi, a + 1, b + 1);
            quad(a, b, b + 1, a + 1);
        }
    }
};

}

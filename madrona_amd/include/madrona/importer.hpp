// Only the mesh description the physics asset baker consumes.
// API contract: reference include/madrona/importer.hpp:12-25 (SourceMesh).
// File importers (glTF/OBJ/USD) are out of scope for this backend.
#pragma once

#include <madrona/math.hpp>
#include <madrona/span.hpp>

namespace madrona::imp {

struct SourceMesh {
    math::Vector3 *positions;
    math::Vector3 *normals;
    math::Vector4 *tangentAndSigns;
    math::Vector2 *uvs;

    uint32_t *indices;
    uint32_t *faceCounts;     // nullptr: triangles
    uint32_t *faceMaterials;

    uint32_t numVertices;
    uint32_t numFaces;
    uint32_t materialIDX;
};

struct SourceObject {
    Span<SourceMesh> meshes;
};

}

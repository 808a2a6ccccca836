// HIP backend: a simulator's MeshSet as the executor's render configuration.
#pragma once

#include "mesh_set.hpp"

#include <madrona/mw_gpu.hpp>

namespace simmgr {

// geometry + materials of a MeshSet as the executor's render configuration
static inline void meshesToRenderConfig(const simmesh::MeshSet &m,
                                        madrona::CudaBatchRenderConfig &cfg)
{
    using madrona::CountT;
    using madrona::Span;
    using madrona::math::Vector3;
    cfg.geoBVHData.vertices = Span<const Vector3>(
        (const Vector3 *)m.vertices.data(), (CountT)(m.vertices.size() / 3));
    cfg.geoBVHData.indices =
        Span<const uint32_t>(m.indices.data(), (CountT)m.indices.size());
    cfg.geoBVHData.objectVertexOffsets = Span<const uint32_t>(
        m.vertexOffsets.data(), (CountT)m.vertexOffsets.size());
    cfg.geoBVHData.objectTriangleOffsets = Span<const uint32_t>(
        m.triangleOffsets.data(), (CountT)m.triangleOffsets.size());
    cfg.materialData.materialColors = Span<const Vector3>(
        (const Vector3 *)m.materialColors.data(),
        (CountT)(m.materialColors.size() / 3));
    cfg.materialData.objectMaterials = Span<const int32_t>(
        m.objectMaterials.data(), (CountT)m.objectMaterials.size());
}

}

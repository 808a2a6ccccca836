// Convex-hull geometry types shared by asset baking (host) and the narrowphase
// (device).  API contract: reference include/madrona/geo.hpp:7-78, geo.inl.
#pragma once

#include <madrona/math.hpp>

namespace madrona::geo {

struct HalfEdge {
    uint32_t next;
    uint32_t rootVertex;
    uint32_t face;
};

struct Plane {
    math::Vector3 normal; // potentially unnormalized
    float d;
};

struct Segment {
    math::Vector3 p1;
    math::Vector3 p2;
};

// Half edges are stored in twin pairs: (2k, 2k + 1).
struct HalfEdgeMesh {
    template <typename Fn>
    MADRONA_HD inline void iterateFaceIndices(uint32_t face, Fn &&fn) const
    {
        const uint32_t start = faceBaseHalfEdges[face];
        uint32_t hedge_idx = start;
        do {
            const HalfEdge &hedge = halfEdges[hedge_idx];
            fn(hedge.rootVertex);
            hedge_idx = hedge.next;
        } while (hedge_idx != start);
    }

    MADRONA_HD inline uint32_t twinIDX(uint32_t half_edge_id) const
    {
        return half_edge_id ^ 1u;
    }

    MADRONA_HD inline uint32_t numEdges() const { return numHalfEdges / 2; }

    MADRONA_HD inline uint32_t edgeToHalfEdge(uint32_t edge_id) const
    {
        return edge_id * 2;
    }

    HalfEdge *halfEdges;
    uint32_t *faceBaseHalfEdges;
    Plane *facePlanes;
    math::Vector3 *vertices;

    uint32_t numHalfEdges;
    uint32_t numFaces;
    uint32_t numVertices;
};

// Sphere at the origin, ray_d normalized.  Numerically careful form (Ray
// Tracing Gems I, ch. 7), same evaluation order as reference geo.inl:36-70.
MADRONA_HD inline float intersectRayOriginSphere(math::Vector3 ray_o,
                                                 math::Vector3 ray_d,
                                                 float r)
{
    float r2 = math::sqr(r);
    float c = ray_o.length2() - r2;
    if (c <= 0.f) {
        return 0.f;
    }

    float b_prime = -dot(ray_o, ray_d);
    if (b_prime < 0.f) {
        return FLT_MAX;
    }

    float l2 = (ray_o + b_prime * ray_d).length2();
    float delta = r2 - l2;
    if (delta < 0.f) {
        return FLT_MAX;
    }

    float q = b_prime + sqrtf(delta);
    return c / q;
}

// Non-unit triangle normal from the two shorter edges (reference geo.inl:
// 157-172).
MADRONA_HD inline math::Vector3 computeTriangleGeoNormal(math::Vector3 ab,
                                                         math::Vector3 ac,
                                                         math::Vector3 bc)
{
    math::Vector3 normal_bc = cross(ab, bc);
    math::Vector3 normal_ac = cross(ab, ac);
    return bc.length2() < ac.length2() ? normal_bc : normal_ac;
}

}

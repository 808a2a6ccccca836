// Distance from the origin to a convex hull with GJK; the simplex sub-problems
// are solved with the signed-volumes method (Montanari, Petrinic, Barbieri,
// "Improving the GJK Algorithm for Faster and More Reliable Distance Queries
// Between Convex Objects", ToG 2017), top down: tetrahedron -> faces -> edges.
//
// Used by the sphere-vs-hull narrowphase test.  Arithmetic and the order in
// which sub-simplices are tried follow the reference (src/physics/gjk.hpp:
// 1-simplex :176-184, 2-simplex :186-255, 3-simplex :258-393, 4-simplex
// :395-515, iteration :517-695; src/physics/geo.cpp:12-61 for the hull
// support function), including its deviations from the paper (a sub-simplex is
// searched whenever the sign test is *not strictly* passed), so results match
// the CPU oracle bit for bit.
#pragma once

#include <madrona/math.hpp>

namespace madrona::phys::gjk {

using math::Vector2;
using math::Vector3;

// closest point of a simplex to the origin: v = sum lambdas[i] * Y[i]
struct SimplexSolve {
    Vector3 v;
    float vLen2;
    float lambdas[4];
};

MADRONA_HD inline bool sameSign(float a, float b)
{
    return (a > 0 && b > 0) || (a < 0 && b < 0);
}

MADRONA_HD inline SimplexSolve makeSolve(Vector3 v, float l0, float l1,
                                         float l2, float l3)
{
    return SimplexSolve { v, v.length2(), { l0, l1, l2, l3 } };
}

MADRONA_HD inline SimplexSolve solvePoint(Vector3 Y0)
{
    return makeSolve(Y0, 1.f, 0.f, 0.f, 0.f);
}

// segment: the newest point is Y1 (s1 of the paper)
MADRONA_HD inline SimplexSolve solveSegment(Vector3 Y0, Vector3 Y1)
{
    Vector3 s1 = Y1;
    Vector3 s2 = Y0;

    Vector3 t = s2 - s1;
    float t_len2 = t.length2();

    // project on the coordinate axis along which the segment is longest
    float mu_max = s1.x - s2.x;
    float s1_I = s1.x;
    float s2_I = s2.x;

    float mu_y = s1.y - s2.y;
    if (fabsf(mu_y) > fabsf(mu_max)) {
        mu_max = mu_y;
        s1_I = s1.y;
        s2_I = s2.y;
    }

    float mu_z = s1.z - s2.z;
    if (fabsf(mu_z) > fabsf(mu_max)) {
        mu_max = mu_z;
        s1_I = s1.z;
        s2_I = s2.z;
    }

    // that coordinate of the origin's projection on the line
    float pO_I = (dot(s2, t) / t_len2) * (s1_I - s2_I) + s2_I;

    float C1 = pO_I - s2_I;
    float C2 = s1_I - pO_I;

    if (sameSign(mu_max, C1) && sameSign(mu_max, C2)) {
        float lambda2 = C2 / mu_max;
        Vector3 v = s1 + t * lambda2;

        float lambda1 = 1.f - lambda2;
        return makeSolve(v, lambda2, lambda1, 0.f, 0.f);
    }

    return makeSolve(s1, 0.f, 1.f, 0.f, 0.f);
}

// triangle: the newest point is Y2
MADRONA_HD inline SimplexSolve solveTriangle(Vector3 Y0, Vector3 Y1, Vector3 Y2)
{
    Vector3 s1 = Y2;
    Vector3 s2 = Y1;
    Vector3 s3 = Y0;

    Vector3 n = cross(s2 - s1, s3 - s1);
    float n_len2 = n.length2();
    Vector3 pO = dot(s1, n) * n / n_len2;

    // drop the coordinate along which the triangle's projected area is largest
    float M_14 = s2.y * s3.z - s3.y * s2.z
               - s1.y * s3.z + s3.y * s1.z
               + s1.y * s2.z - s2.y * s1.z;

    float M_24 = s2.x * s3.z - s3.x * s2.z
               - s1.x * s3.z + s3.x * s1.z
               + s1.x * s2.z - s2.x * s1.z;

    float M_34 = s2.x * s3.y - s3.x * s2.y
               - s1.x * s3.y + s3.x * s1.y
               + s1.x * s2.y - s2.x * s1.y;

    float M_14_abs = fabsf(M_14);
    float M_24_abs = fabsf(M_24);
    float M_34_abs = fabsf(M_34);

    float mu_max;
    Vector2 s1_2D, s2_2D, s3_2D, pO_2D;
    if (M_14_abs >= M_24_abs && M_14_abs >= M_34_abs) {
        mu_max = M_14;
        s1_2D = { s1.y, s1.z };
        s2_2D = { s2.y, s2.z };
        s3_2D = { s3.y, s3.z };
        pO_2D = { pO.y, pO.z };
    } else if (M_24_abs >= M_34_abs) {
        mu_max = M_24;
        s1_2D = { s1.x, s1.z };
        s2_2D = { s2.x, s2.z };
        s3_2D = { s3.x, s3.z };
        pO_2D = { pO.x, pO.z };
    } else {
        mu_max = M_34;
        s1_2D = { s1.x, s1.y };
        s2_2D = { s2.x, s2.y };
        s3_2D = { s3.x, s3.y };
        pO_2D = { pO.x, pO.y };
    }

    float C1 = pO_2D.x * s2_2D.y + pO_2D.y * s3_2D.x + s2_2D.x * s3_2D.y
             - pO_2D.x * s3_2D.y - pO_2D.y * s2_2D.x - s3_2D.x * s2_2D.y;

    float C2 = pO_2D.x * s3_2D.y + pO_2D.y * s1_2D.x + s3_2D.x * s1_2D.y
             - pO_2D.x * s1_2D.y - pO_2D.y * s3_2D.x - s1_2D.x * s3_2D.y;

    float C3 = pO_2D.x * s1_2D.y + pO_2D.y * s2_2D.x + s1_2D.x * s2_2D.y
             - pO_2D.x * s2_2D.y - pO_2D.y * s1_2D.x - s2_2D.x * s1_2D.y;

    bool inside1 = sameSign(mu_max, C1);
    bool inside2 = sameSign(mu_max, C2);
    bool inside3 = sameSign(mu_max, C3);

    if (inside1 && inside2 && inside3) {
        float lambda2 = C2 / mu_max;
        float lambda3 = C3 / mu_max;
        float lambda1 = 1.f - lambda2 - lambda3;

        Vector3 v = s1 * lambda1 + s2 * lambda2 + s3 * lambda3;
        return makeSolve(v, lambda3, lambda2, lambda1, 0.f);
    }

    // best of the edges whose test failed, tried in this order
    SimplexSolve res;
    res.vLen2 = FLT_MAX;
    if (!inside2) {
        SimplexSolve sub = solveSegment(Y0, Y2);
        res = makeSolve(sub.v, sub.lambdas[0], 0.f, sub.lambdas[1], 0.f);
        res.vLen2 = sub.vLen2;
    }

    if (!inside3) {
        SimplexSolve sub = solveSegment(Y1, Y2);
        if (sub.vLen2 < res.vLen2) {
            res = makeSolve(sub.v, 0.f, sub.lambdas[0], sub.lambdas[1], 0.f);
            res.vLen2 = sub.vLen2;
        }
    }

    if (!inside1) {
        SimplexSolve sub = solveSegment(Y0, Y1);
        if (sub.vLen2 < res.vLen2) {
            res = makeSolve(sub.v, sub.lambdas[0], sub.lambdas[1], 0.f, 0.f);
            res.vLen2 = sub.vLen2;
        }
    }

    return res;
}

// tetrahedron: the newest point is Y3
MADRONA_HD inline SimplexSolve solveTetrahedron(Vector3 Y0, Vector3 Y1,
                                                Vector3 Y2, Vector3 Y3)
{
    Vector3 s1 = Y3;
    Vector3 s2 = Y2;
    Vector3 s3 = Y1;
    Vector3 s4 = Y0;

    // cofactors of the bottom row of [s1 s2 s3 s4; 1 1 1 1]
    auto det = [](Vector3 a, Vector3 b, Vector3 c) {
        return dot(a, cross(b, c));
    };

    float C_41 = -det(s2, s3, s4);
    float C_42 = det(s1, s3, s4);
    float C_43 = -det(s1, s2, s4);
    float C_44 = det(s1, s2, s3);

    float det_M = C_41 + C_42 + C_43 + C_44;

    bool inside1 = sameSign(det_M, C_41);
    bool inside2 = sameSign(det_M, C_42);
    bool inside3 = sameSign(det_M, C_43);
    bool inside4 = sameSign(det_M, C_44);

    if (inside1 && inside2 && inside3 && inside4) {
        float lambda1 = C_41 / det_M;
        float lambda2 = C_42 / det_M;
        float lambda3 = C_43 / det_M;
        float lambda4 = 1.f - lambda1 - lambda2 - lambda3;

        Vector3 v = s1 * lambda1 + s2 * lambda2 + s3 * lambda3 + s4 * lambda4;
        return makeSolve(v, lambda4, lambda3, lambda2, lambda1);
    }

    // best of the faces whose test failed (a zero determinant fails it too, so
    // degenerate tetrahedra still search their faces), tried in this order
    SimplexSolve res;
    res.vLen2 = FLT_MAX;
    if (!inside2) {
        SimplexSolve sub = solveTriangle(Y0, Y1, Y3);
        res = makeSolve(sub.v, sub.lambdas[0], sub.lambdas[1], 0.f,
                        sub.lambdas[2]);
        res.vLen2 = sub.vLen2;
    }

    if (!inside3) {
        SimplexSolve sub = solveTriangle(Y0, Y2, Y3);
        if (sub.vLen2 < res.vLen2) {
            res = makeSolve(sub.v, sub.lambdas[0], 0.f, sub.lambdas[1],
                            sub.lambdas[2]);
            res.vLen2 = sub.vLen2;
        }
    }

    if (!inside4) {
        SimplexSolve sub = solveTriangle(Y1, Y2, Y3);
        if (sub.vLen2 < res.vLen2) {
            res = makeSolve(sub.v, 0.f, sub.lambdas[0], sub.lambdas[1],
                            sub.lambdas[2]);
            res.vLen2 = sub.vLen2;
        }
    }

    if (!inside1) {
        SimplexSolve sub = solveTriangle(Y0, Y1, Y2);
        if (sub.vLen2 < res.vLen2) {
            res = makeSolve(sub.v, sub.lambdas[0], sub.lambdas[1],
                            sub.lambdas[2], 0.f);
            res.vLen2 = sub.vLen2;
        }
    }

    return res;
}

// farthest hull vertex along v (first one wins among equals)
template <typename HullT>
MADRONA_HD inline Vector3 hullSupport(const HullT &hull, Vector3 v)
{
    float max_dot = -FLT_MAX;
    Vector3 support = Vector3::zero();
    const CountT num_verts = hull.numVertices();
    for (CountT i = 0; i < num_verts; i++) {
        Vector3 w = hull.vertex(i);

        float w_dot_v = dot(w, v);
        if (w_dot_v > max_dot) {
            max_dot = w_dot_v;
            support = w;
        }
    }

    return support;
}

// Squared distance from the origin to the hull, 0 when the origin is inside or
// within tolerance (then *closest_point is meaningless); otherwise
// *closest_point is the hull point nearest the origin.
template <typename HullT>
MADRONA_HD inline float hullClosestPointToOrigin(const HullT &hull,
                                                 float err_tolerance2,
                                                 Vector3 *closest_point)
{
    Vector3 v = -hull.vertex(0);
    Vector3 Y[4] = { Vector3::zero(), Vector3::zero(), Vector3::zero(),
                     Vector3::zero() };
    CountT nY = 0;

    float v_len2 = 0.f;
    float prev_v_len2 = FLT_MAX;

    while (true) {
        Vector3 w = hullSupport(hull, v);

        SimplexSolve solve;
        if (nY == 0) {
            Y[0] = w;
            solve = solvePoint(Y[0]);
        } else if (nY == 1) {
            Y[1] = w;
            solve = solveSegment(Y[0], Y[1]);
        } else if (nY == 2) {
            Y[2] = w;
            solve = solveTriangle(Y[0], Y[1], Y[2]);
        } else {
            Y[3] = w;
            solve = solveTetrahedron(Y[0], Y[1], Y[2], Y[3]);
        }

        // keep the points that support the new closest point
        nY = 0;
MADRONA_UNROLL
        for (CountT i = 0; i < 4; i++) {
            if (solve.lambdas[i] != 0.f) {
                // Y[nY] = Y[i] with constant indices only (a dynamically
                // indexed private array would live in scratch on the GPU)
                const Vector3 kept = Y[i];
MADRONA_UNROLL
                for (CountT j = 0; j < 4; j++) {
                    if (j == nY) {
                        Y[j] = kept;
                    }
                }
                nY += 1;
            }
        }

        // four supporting points: the origin is inside the tetrahedron
        if (nY == 4) {
            *closest_point = -v;
            return 0.f;
        }

        if (solve.vLen2 <= err_tolerance2) {
            *closest_point = -v;
            return 0.f;
        }

        {
            float max_Y_len2 = Y[0].length2();
MADRONA_UNROLL
            for (CountT i = 1; i < 4; i++) {
                float Y_len2 = Y[i].length2();
                if (i < nY && Y_len2 > max_Y_len2) {
                    max_Y_len2 = Y_len2;
                }
            }

            // v vanishes relative to the simplex: direction meaningless
            if (solve.vLen2 <= FLT_EPSILON * max_Y_len2) {
                *closest_point = -v;
                return 0.f;
            }
        }

        v_len2 = solve.vLen2;
        v = -solve.v;

        // no longer improving within fp32
        if (prev_v_len2 - v_len2 <= FLT_EPSILON * prev_v_len2) {
            break;
        }

        prev_v_len2 = v_len2;
    }

    *closest_point = -v;
    return v_len2;
}

}

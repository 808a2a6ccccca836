// Small-vector math used by simulators and the physics nodes, host + device.
//
// API contract: reference include/madrona/math.hpp:17-385 (type names, member
// names, layouts -- components are stored in exported tensors, so
// sizeof/field order are ABI).  The implementation is written for this
// backend: everything is defined in-class, marked host+device, and keeps the
// reference CPU backend's *evaluation order* (math.inl) so that fp32 results
// are bit-identical when both sides are built with -ffp-contract=off.  In
// particular inverse lengths are IEEE 1/sqrt (the CPU branch of
// math.inl:53-63, 233-241), never the fast rsqrt of the reference GPU build.
#pragma once

#include <madrona/macros.hpp>
#include <madrona/types.hpp>

#include <cfloat>
#include <cmath>
#include <cstdint>

namespace madrona {
namespace math {

struct Vector2;
struct Vector3;
struct Vector4;
struct Quat;
struct Diag3x3;
struct Mat3x3;
struct Mat3x4;

constexpr inline float pi { 3.14159265358979323846264338327950288f };
constexpr inline float pi_d2 { pi / 2.f };
constexpr inline float pi_m2 { pi * 2.f };

MADRONA_HD inline constexpr float toRadians(float degrees)
{
    constexpr float mult = pi / 180.f;
    return mult * degrees;
}

MADRONA_HD inline float sqr(float x) { return x * x; }

// Bit-trick reciprocal square root (one Newton step), same constants as the
// reference (math.inl:23-33) so results match bit for bit.
MADRONA_HD inline float rsqrtApprox(float x)
{
    float y = __builtin_bit_cast(float,
        0x5F1FFFF9u - (__builtin_bit_cast(uint32_t, x) >> 1));
    return y * (0.703952253f * (2.38924456f - x * y * y));
}

// a * x^2 + b * x + c = 0, a != 0
MADRONA_HD inline bool solveQuadraticUnsafe(float a, float b, float c,
                                            float *t1, float *t2)
{
    float det = b * b - 4.f * a * c;
    if (det < 0.f) {
        return false;
    }

    float sqrt_det = sqrtf(det);
    float rcp_2a = 1.f / (2.f * a);
    *t1 = (-b - sqrt_det) * rcp_2a;
    *t2 = (-b + sqrt_det) * rcp_2a;
    return true;
}

struct Vector2 {
    float x;
    float y;

    MADRONA_HD inline float dot(const Vector2 &o) const { return x * o.x + y * o.y; }
    MADRONA_HD inline float length2() const { return x * x + y * y; }
    MADRONA_HD inline float length() const { return sqrtf(length2()); }
    MADRONA_HD inline float invLength() const { return 1.f / length(); }

    MADRONA_HD inline float &operator[](CountT i) { return i == 0 ? x : y; }
    MADRONA_HD inline float operator[](CountT i) const { return i == 0 ? x : y; }

    MADRONA_HD constexpr inline Vector2 &operator+=(const Vector2 &o) { x += o.x; y += o.y; return *this; }
    MADRONA_HD constexpr inline Vector2 &operator-=(const Vector2 &o) { x -= o.x; y -= o.y; return *this; }
    MADRONA_HD constexpr inline Vector2 &operator+=(float o) { x += o; y += o; return *this; }
    MADRONA_HD constexpr inline Vector2 &operator-=(float o) { x -= o; y -= o; return *this; }
    MADRONA_HD constexpr inline Vector2 &operator*=(float o) { x *= o; y *= o; return *this; }
    MADRONA_HD constexpr inline Vector2 &operator/=(float o) { float inv = 1.f / o; return *this *= inv; }

    MADRONA_HD constexpr friend inline Vector2 operator+(Vector2 v) { return v; }
    MADRONA_HD constexpr friend inline Vector2 operator-(Vector2 v) { return Vector2 { -v.x, -v.y }; }
    MADRONA_HD constexpr friend inline Vector2 operator+(Vector2 a, const Vector2 &b) { a += b; return a; }
    MADRONA_HD constexpr friend inline Vector2 operator-(Vector2 a, const Vector2 &b) { a -= b; return a; }
    MADRONA_HD constexpr friend inline Vector2 operator+(Vector2 a, float b) { a += b; return a; }
    MADRONA_HD constexpr friend inline Vector2 operator-(Vector2 a, float b) { a -= b; return a; }
    MADRONA_HD constexpr friend inline Vector2 operator*(Vector2 a, float b) { a *= b; return a; }
    MADRONA_HD constexpr friend inline Vector2 operator/(Vector2 a, float b) { a /= b; return a; }
    MADRONA_HD constexpr friend inline Vector2 operator+(float a, Vector2 b) { return b + a; }
    MADRONA_HD constexpr friend inline Vector2 operator-(float a, Vector2 b) { return -b + a; }
    MADRONA_HD constexpr friend inline Vector2 operator*(float a, Vector2 b) { return b * a; }
    MADRONA_HD constexpr friend inline Vector2 operator/(float a, Vector2 b) { return Vector2 { a / b.x, a / b.y }; }

    MADRONA_HD static inline Vector2 min(Vector2 a, Vector2 b) { return Vector2 { fminf(a.x, b.x), fminf(a.y, b.y) }; }
    MADRONA_HD static inline Vector2 max(Vector2 a, Vector2 b) { return Vector2 { fmaxf(a.x, b.x), fmaxf(a.y, b.y) }; }
};

struct Vector3 {
    float x;
    float y;
    float z;

    MADRONA_HD inline float dot(const Vector3 &o) const
    {
        return x * o.x + y * o.y + z * o.z;
    }

    MADRONA_HD inline Vector3 cross(const Vector3 &o) const
    {
        return Vector3 {
            y * o.z - z * o.y,
            z * o.x - x * o.z,
            x * o.y - y * o.x,
        };
    }

    // two vectors perpendicular to *this (which must be normalized)
    MADRONA_HD inline void frame(Vector3 *a, Vector3 *b) const
    {
        Vector3 arbitrary = fabsf(x) < 0.8 ?
            Vector3 { 1, 0, 0 } : Vector3 { 0, 1, 0 };
        *a = cross(arbitrary);
        *b = cross(*a);
    }

    MADRONA_HD inline float length2() const { return x * x + y * y + z * z; }
    MADRONA_HD inline float length() const { return sqrtf(length2()); }
    MADRONA_HD inline float invLength() const { return 1.f / length(); }

    MADRONA_HD inline float distance(const Vector3 &o) const { return (*this - o).length(); }
    MADRONA_HD inline float distance2(const Vector3 &o) const { return (*this - o).length2(); }

    [[nodiscard]] MADRONA_HD inline Vector3 normalize() const { return *this * invLength(); }

    MADRONA_HD constexpr inline Vector2 xy() const { return Vector2 { x, y }; }
    MADRONA_HD constexpr inline Vector2 yz() const { return Vector2 { y, z }; }
    MADRONA_HD constexpr inline Vector2 xz() const { return Vector2 { x, z }; }
    MADRONA_HD constexpr inline Vector2 yx() const { return Vector2 { y, x }; }
    MADRONA_HD constexpr inline Vector2 zy() const { return Vector2 { z, y }; }
    MADRONA_HD constexpr inline Vector2 zx() const { return Vector2 { z, x }; }

    MADRONA_HD inline float &operator[](CountT i) { return i == 0 ? x : (i == 1 ? y : z); }
    MADRONA_HD inline float operator[](CountT i) const { return i == 0 ? x : (i == 1 ? y : z); }

    MADRONA_HD constexpr inline Vector3 &operator+=(const Vector3 &o) { x += o.x; y += o.y; z += o.z; return *this; }
    MADRONA_HD constexpr inline Vector3 &operator-=(const Vector3 &o) { x -= o.x; y -= o.y; z -= o.z; return *this; }
    MADRONA_HD constexpr inline Vector3 &operator+=(float o) { x += o; y += o; z += o; return *this; }
    MADRONA_HD constexpr inline Vector3 &operator-=(float o) { x -= o; y -= o; z -= o; return *this; }
    MADRONA_HD constexpr inline Vector3 &operator*=(float o) { x *= o; y *= o; z *= o; return *this; }
    MADRONA_HD constexpr inline Vector3 &operator/=(float o) { float inv = 1.f / o; return *this *= inv; }

    MADRONA_HD constexpr friend inline Vector3 operator-(Vector3 v) { return Vector3 { -v.x, -v.y, -v.z }; }
    MADRONA_HD constexpr friend inline Vector3 operator+(Vector3 a, const Vector3 &b) { a += b; return a; }
    MADRONA_HD constexpr friend inline Vector3 operator-(Vector3 a, const Vector3 &b) { a -= b; return a; }
    MADRONA_HD constexpr friend inline Vector3 operator+(Vector3 a, float b) { a += b; return a; }
    MADRONA_HD constexpr friend inline Vector3 operator-(Vector3 a, float b) { a -= b; return a; }
    MADRONA_HD constexpr friend inline Vector3 operator*(Vector3 a, float b) { a *= b; return a; }
    MADRONA_HD constexpr friend inline Vector3 operator/(Vector3 a, float b) { a /= b; return a; }
    MADRONA_HD constexpr friend inline Vector3 operator+(float a, Vector3 b) { return b + a; }
    MADRONA_HD constexpr friend inline Vector3 operator-(float a, Vector3 b) { return -b + a; }
    MADRONA_HD constexpr friend inline Vector3 operator*(float a, Vector3 b) { return b * a; }
    MADRONA_HD constexpr friend inline Vector3 operator/(float a, Vector3 b) { return Vector3 { a / b.x, a / b.y, a / b.z }; }

    MADRONA_HD static inline Vector3 min(Vector3 a, Vector3 b)
    {
        return Vector3 { fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z) };
    }

    MADRONA_HD static inline Vector3 max(Vector3 a, Vector3 b)
    {
        return Vector3 { fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z) };
    }

    MADRONA_HD static constexpr inline Vector3 zero() { return Vector3 { 0, 0, 0 }; }
    MADRONA_HD static constexpr inline Vector3 one() { return Vector3 { 1, 1, 1 }; }
    MADRONA_HD static constexpr inline Vector3 all(float v) { return Vector3 { v, v, v }; }
};

MADRONA_HD inline float dot(Vector2 a, Vector2 b) { return a.dot(b); }
MADRONA_HD inline float dot(Vector3 a, Vector3 b) { return a.dot(b); }
MADRONA_HD inline Vector3 cross(Vector3 a, Vector3 b) { return a.cross(b); }
MADRONA_HD inline Vector3 normalize(Vector3 v) { return v.normalize(); }

// reflects `direction` over `normal`
MADRONA_HD inline Vector3 reflect(Vector3 direction, Vector3 normal)
{
    return direction - (2.f * direction.dot(normal) * normal) /
        normal.dot(normal);
}

struct Vector4 {
    float x;
    float y;
    float z;
    float w;

    MADRONA_HD inline Vector3 xyz() const { return Vector3 { x, y, z }; }

    MADRONA_HD inline float &operator[](CountT i)
    {
        return i == 1 ? y : (i == 2 ? z : (i == 3 ? w : x));
    }

    MADRONA_HD inline float operator[](CountT i) const
    {
        return i == 1 ? y : (i == 2 ? z : (i == 3 ? w : x));
    }

    MADRONA_HD static inline Vector4 fromVec3W(Vector3 v, float w) { return Vector4 { v.x, v.y, v.z, w }; }

    MADRONA_HD inline Vector4 operator*(float s) const { return Vector4 { x * s, y * s, z * s, w * s }; }
    MADRONA_HD inline Vector4 operator+(const Vector4 &o) const { return Vector4 { x + o.x, y + o.y, z + o.z, w + o.w }; }

    MADRONA_HD static constexpr inline Vector4 zero() { return Vector4 { 0, 0, 0, 0 }; }
    MADRONA_HD static constexpr inline Vector4 one() { return Vector4 { 1, 1, 1, 1 }; }
};

struct Quat {
    float w;
    float x;
    float y;
    float z;

    MADRONA_HD inline float length2() const { return w * w + x * x + y * y + z * z; }
    MADRONA_HD inline float length() const { return sqrtf(length2()); }
    MADRONA_HD inline float invLength() const { return 1.f / sqrtf(length2()); }

    [[nodiscard]] MADRONA_HD inline Quat normalize() const
    {
        float inv_length = invLength();
        return Quat { w * inv_length, x * inv_length, y * inv_length,
                      z * inv_length };
    }

    [[nodiscard]] MADRONA_HD inline Quat inv() const { return Quat { w, -x, -y, -z }; }

    MADRONA_HD inline Vector3 rotateVec(Vector3 v) const
    {
        Vector3 pure { x, y, z };
        float scalar = w;

        Vector3 pure_x_v = cross(pure, v);
        Vector3 pure_x_pure_x_v = cross(pure, pure_x_v);

        return v + 2.f * ((pure_x_v * scalar) + pure_x_pure_x_v);
    }

    MADRONA_HD static inline Quat angleAxis(float angle, Vector3 normal)
    {
        float coshalf = cosf(angle / 2.f);
        float sinhalf = sinf(angle / 2.f);

        return Quat { coshalf, normal.x * sinhalf, normal.y * sinhalf,
                      normal.z * sinhalf };
    }

    MADRONA_HD static inline Quat fromAngularVec(Vector3 v) { return Quat { 0, v.x, v.y, v.z }; }

    // rotation whose matrix has columns a, b, c
    MADRONA_HD static inline Quat fromBasis(Vector3 a, Vector3 b, Vector3 c)
    {
        // pick the largest of 4w^2-1, 4x^2-1, 4y^2-1, 4z^2-1 for stability
        float fx = a.x - b.y - c.z;
        float fy = b.y - a.x - c.z;
        float fz = c.z - a.x - b.y;
        float fw = a.x + b.y + c.z;

        int which = 0;
        float biggest = fw;
        if (fx > biggest) { biggest = fx; which = 1; }
        if (fy > biggest) { biggest = fy; which = 2; }
        if (fz > biggest) { biggest = fz; which = 3; }

        float biggest_val = sqrtf(biggest + 1.f) * 0.5f;
        float mult = 0.25f / biggest_val;

        switch (which) {
        case 0:
            return { biggest_val, (b.z - c.y) * mult, (c.x - a.z) * mult,
                     (a.y - b.x) * mult };
        case 1:
            return { (b.z - c.y) * mult, biggest_val, (a.y + b.x) * mult,
                     (c.x + a.z) * mult };
        case 2:
            return { (c.x - a.z) * mult, (a.y + b.x) * mult, biggest_val,
                     (b.z + c.y) * mult };
        default:
            return { (a.y - b.x) * mult, (c.x + a.z) * mult,
                     (b.z + c.y) * mult, biggest_val };
        }
    }

    MADRONA_HD static constexpr inline Quat id() { return Quat { 1.f, 0.f, 0.f, 0.f }; }

    MADRONA_HD inline Quat &operator+=(Quat o) { w += o.w; x += o.x; y += o.y; z += o.z; return *this; }
    MADRONA_HD inline Quat &operator-=(Quat o) { w -= o.w; x -= o.x; y -= o.y; z -= o.z; return *this; }
    MADRONA_HD inline Quat &operator*=(Quat o) { return *this = (*this * o); }
    MADRONA_HD inline Quat &operator*=(float f) { w *= f; x *= f; y *= f; z *= f; return *this; }

    MADRONA_HD friend inline Quat operator+(Quat a, Quat b) { return a += b; }
    MADRONA_HD friend inline Quat operator-(Quat a, Quat b) { return a -= b; }

    MADRONA_HD friend inline Quat operator*(Quat a, Quat b)
    {
        return Quat {
            (a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z),
            (a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y),
            (a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x),
            (a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w),
        };
    }

    MADRONA_HD friend inline Quat operator*(Quat a, float b) { a *= b; return a; }
    MADRONA_HD friend inline Quat operator*(float b, Quat a) { a *= b; return a; }
};

struct Diag3x3 {
    float d0;
    float d1;
    float d2;

    MADRONA_HD inline Diag3x3 inv() const { return Diag3x3 { 1.f / d0, 1.f / d1, 1.f / d2 }; }

    MADRONA_HD static inline Diag3x3 fromVec(Vector3 v) { return Diag3x3 { v.x, v.y, v.z }; }
    MADRONA_HD static constexpr inline Diag3x3 uniform(float scale) { return Diag3x3 { scale, scale, scale }; }
    MADRONA_HD static constexpr inline Diag3x3 id() { return uniform(1.f); }

    MADRONA_HD inline Diag3x3 &operator*=(Diag3x3 o) { d0 *= o.d0; d1 *= o.d1; d2 *= o.d2; return *this; }
    MADRONA_HD inline Diag3x3 &operator*=(float o) { d0 *= o; d1 *= o; d2 *= o; return *this; }
    MADRONA_HD inline Diag3x3 &operator/=(float o) { float inv = 1.f / o; return *this *= inv; }

    MADRONA_HD inline float &operator[](CountT i) { return i == 0 ? d0 : (i == 1 ? d1 : d2); }
    MADRONA_HD inline float operator[](CountT i) const { return i == 0 ? d0 : (i == 1 ? d1 : d2); }

    MADRONA_HD friend inline Diag3x3 operator*(Diag3x3 a, Diag3x3 b) { a *= b; return a; }
    MADRONA_HD friend inline Diag3x3 operator*(Diag3x3 a, float b) { a *= b; return a; }
    MADRONA_HD friend inline Diag3x3 operator*(float a, Diag3x3 b) { b *= a; return b; }
    MADRONA_HD friend inline Vector3 operator*(Diag3x3 d, Vector3 v) { return Vector3 { d.d0 * v.x, d.d1 * v.y, d.d2 * v.z }; }
    MADRONA_HD friend inline Diag3x3 operator/(Diag3x3 a, float b) { a /= b; return a; }
    MADRONA_HD friend inline Diag3x3 operator/(float a, Diag3x3 b) { return Diag3x3 { a / b.d0, a / b.d1, a / b.d2 }; }
};

namespace detail {

// Column-major rotation(+scale) matrix entries shared by Mat3x3::fromQuat,
// Mat3x3::fromRS and Mat3x4::fromTRS.
MADRONA_HD inline void rotScaleColumns(Quat r, Diag3x3 s, Vector3 *c0,
                                       Vector3 *c1, Vector3 *c2)
{
    float x2 = r.x * r.x;
    float y2 = r.y * r.y;
    float z2 = r.z * r.z;
    float xz = r.x * r.z;
    float xy = r.x * r.y;
    float yz = r.y * r.z;
    float wx = r.w * r.x;
    float wy = r.w * r.y;
    float wz = r.w * r.z;

    Diag3x3 ds = 2.f * s;

    *c0 = Vector3 { s.d0 - ds.d0 * (y2 + z2), ds.d0 * (xy + wz), ds.d0 * (xz - wy) };
    *c1 = Vector3 { ds.d1 * (xy - wz), s.d1 - ds.d1 * (x2 + z2), ds.d1 * (yz + wx) };
    *c2 = Vector3 { ds.d2 * (xz + wy), ds.d2 * (yz - wx), s.d2 - ds.d2 * (x2 + y2) };
}

}

struct Mat3x3 {
    struct Transpose {
        const Mat3x3 *src;

        MADRONA_HD inline Vector3 operator[](CountT i) const
        {
            return Vector3 { src->cols[0][i], src->cols[1][i], src->cols[2][i] };
        }

        MADRONA_HD friend inline Vector3 operator*(Transpose t, Vector3 v)
        {
            return Vector3 { dot(t.src->cols[0], v), dot(t.src->cols[1], v),
                             dot(t.src->cols[2], v) };
        }
    };

    Vector3 cols[3];

    MADRONA_HD inline float determinant() const
    {
        Vector3 c0 = cols[0];
        Vector3 c1 = cols[1];
        Vector3 c2 = cols[2];

        return c0.x * (c1.y * c2.z - c2.y * c1.z) -
               c0.y * (c1.x * c2.z - c2.x * c1.z) +
               c0.z * (c1.x * c2.y - c2.x * c1.y);
    }

    MADRONA_HD inline Transpose transpose() const { return Transpose { this }; }

    MADRONA_HD static inline Mat3x3 fromQuat(Quat r)
    {
        float x2 = r.x * r.x;
        float y2 = r.y * r.y;
        float z2 = r.z * r.z;
        float xz = r.x * r.z;
        float xy = r.x * r.y;
        float yz = r.y * r.z;
        float wx = r.w * r.x;
        float wy = r.w * r.y;
        float wz = r.w * r.z;

        return {{
            { 1.f - 2.f * (y2 + z2), 2.f * (xy + wz), 2.f * (xz - wy) },
            { 2.f * (xy - wz), 1.f - 2.f * (x2 + z2), 2.f * (yz + wx) },
            { 2.f * (xz + wy), 2.f * (yz - wx), 1.f - 2.f * (x2 + y2) },
        }};
    }

    MADRONA_HD static inline Mat3x3 fromRS(Quat r, Diag3x3 s)
    {
        Mat3x3 m;
        detail::rotScaleColumns(r, s, &m.cols[0], &m.cols[1], &m.cols[2]);
        return m;
    }

    MADRONA_HD inline Vector3 &operator[](CountT i) { return cols[i]; }
    MADRONA_HD inline Vector3 operator[](CountT i) const { return cols[i]; }

    MADRONA_HD inline Mat3x3 &operator+=(const Mat3x3 &o) { cols[0] += o[0]; cols[1] += o[1]; cols[2] += o[2]; return *this; }
    MADRONA_HD inline Mat3x3 &operator-=(const Mat3x3 &o) { cols[0] -= o[0]; cols[1] -= o[1]; cols[2] -= o[2]; return *this; }

    MADRONA_HD inline Vector3 operator*(Vector3 v) const
    {
        return cols[0] * v.x + cols[1] * v.y + cols[2] * v.z;
    }

    MADRONA_HD inline Mat3x3 operator*(const Mat3x3 &o) const
    {
        return Mat3x3 {{ *this * o[0], *this * o[1], *this * o[2] }};
    }

    MADRONA_HD inline Mat3x3 &operator*=(const Mat3x3 &o) { return *this = (*this * o); }
    MADRONA_HD inline Mat3x3 &operator*=(float s) { cols[0] *= s; cols[1] *= s; cols[2] *= s; return *this; }

    MADRONA_HD friend inline Mat3x3 operator+(Mat3x3 a, const Mat3x3 &b) { return (a += b); }
    MADRONA_HD friend inline Mat3x3 operator-(Mat3x3 a, const Mat3x3 &b) { return (a -= b); }

    MADRONA_HD friend inline Mat3x3 operator*(const Mat3x3 &m, Diag3x3 d)
    {
        return Mat3x3 {{ m[0] * d.d0, m[1] * d.d1, m[2] * d.d2 }};
    }

    MADRONA_HD friend inline Mat3x3 operator*(Diag3x3 d, const Mat3x3 &m)
    {
        return Mat3x3 {{ d * m[0], d * m[1], d * m[2] }};
    }

    MADRONA_HD friend inline Mat3x3 operator*(Mat3x3 a, Mat3x3::Transpose b)
    {
        return Mat3x3 {{ a * b[0], a * b[1], a * b[2] }};
    }

    MADRONA_HD friend inline Mat3x3 operator*(Mat3x3::Transpose a, Mat3x3 b)
    {
        return Mat3x3 {{ a * b[0], a * b[1], a * b[2] }};
    }

    MADRONA_HD friend inline Mat3x3 operator*(float s, const Mat3x3 &m)
    {
        return Mat3x3 {{ s * m[0], s * m[1], s * m[2] }};
    }

    MADRONA_HD friend inline Mat3x3 operator*(const Mat3x3 &m, float s) { return s * m; }

    MADRONA_HD friend inline Mat3x3 operator/(const Mat3x3 &m, float s)
    {
        return Mat3x3 {{ m[0] / s, m[1] / s, m[2] / s }};
    }
};

MADRONA_HD inline Mat3x3 outerProduct(Vector3 a, Vector3 b)
{
    return Mat3x3 {{ a * b.x, a * b.y, a * b.z }};
}

struct Symmetric3x3 {
    Vector3 diag;   // m11 m22 m33
    Vector3 off;    // m12 m13 m23

    // A * A^T
    MADRONA_HD static inline Symmetric3x3 AAT(Mat3x3 A)
    {
        Vector3 r0 { A[0].x, A[1].x, A[2].x };
        Vector3 r1 { A[0].y, A[1].y, A[2].y };
        Vector3 r2 { A[0].z, A[1].z, A[2].z };

        return Symmetric3x3 {
            { dot(r0, r0), dot(r1, r1), dot(r2, r2) },
            { dot(r0, r1), dot(r0, r2), dot(r1, r2) },
        };
    }

    // A * X * A^T
    MADRONA_HD static inline Symmetric3x3 AXAT(Mat3x3 A, Symmetric3x3 X)
    {
        Vector3 r0 { A[0].x, A[1].x, A[2].x };
        Vector3 r1 { A[0].y, A[1].y, A[2].y };
        Vector3 r2 { A[0].z, A[1].z, A[2].z };

        Vector3 xc0 = X[0];
        Vector3 xc1 = X[1];
        Vector3 xc2 = X[2];

        // rows of A * X
        Vector3 ax0 { dot(r0, xc0), dot(r0, xc1), dot(r0, xc2) };
        Vector3 ax1 { dot(r1, xc0), dot(r1, xc1), dot(r1, xc2) };
        Vector3 ax2 { dot(r2, xc0), dot(r2, xc1), dot(r2, xc2) };

        return Symmetric3x3 {
            { dot(r0, ax0), dot(r1, ax1), dot(r2, ax2) },
            { dot(r1, ax0), dot(r2, ax0), dot(r2, ax1) },
        };
    }

    // v * v^T
    MADRONA_HD static inline Symmetric3x3 vvT(Vector3 v)
    {
        return Symmetric3x3 {
            { v.x * v.x, v.y * v.y, v.z * v.z },
            { v.x * v.y, v.x * v.z, v.y * v.z },
        };
    }

    MADRONA_HD inline Vector3 operator[](CountT i) const
    {
        return i == 0 ? Vector3 { diag.x, off.x, off.y } :
              (i == 1 ? Vector3 { off.x, diag.y, off.z } :
                        Vector3 { off.y, off.z, diag.z });
    }

    MADRONA_HD inline Symmetric3x3 &operator+=(const Symmetric3x3 &o) { diag += o.diag; off += o.off; return *this; }
    MADRONA_HD inline Symmetric3x3 &operator-=(const Symmetric3x3 &o) { diag -= o.diag; off -= o.off; return *this; }
    MADRONA_HD inline Symmetric3x3 &operator*=(const Symmetric3x3 &o) { return *this = (*this * o); }
    MADRONA_HD inline Symmetric3x3 &operator*=(float s) { diag *= s; off *= s; return *this; }

    MADRONA_HD friend inline Symmetric3x3 operator+(Symmetric3x3 a, Symmetric3x3 b) { a += b; return a; }
    MADRONA_HD friend inline Symmetric3x3 operator-(Symmetric3x3 a, Symmetric3x3 b) { a -= b; return a; }

    // NOTE: like the reference (math.inl, Symmetric3x3 operator*), the
    // off-diagonal terms of the right operand are taken from `a`; kept so
    // results stay identical to the reference.
    MADRONA_HD friend inline Symmetric3x3 operator*(Symmetric3x3 a, Symmetric3x3 b)
    {
        float a11 = a.diag.x, a22 = a.diag.y, a33 = a.diag.z;
        float a12 = a.off.x, a13 = a.off.y, a23 = a.off.z;
        float b11 = b.diag.x, b22 = b.diag.y, b33 = b.diag.z;
        float b12 = a.off.x, b13 = a.off.y, b23 = a.off.z;

        return Symmetric3x3 {
            { a11 * b11 + a12 * b12 + a13 * b13,
              a12 * b12 + a22 * b22 + a23 * b23,
              a13 * b13 + a23 * b23 + a33 * b33 },
            { a11 * b12 + a12 * b22 + a13 * b23,
              a11 * b13 + a12 * b23 + a13 * b33,
              a12 * b13 + a22 * b23 + a23 * b33 },
        };
    }

    MADRONA_HD friend inline Symmetric3x3 operator*(Symmetric3x3 a, float b) { a *= b; return a; }
    MADRONA_HD friend inline Symmetric3x3 operator*(float a, Symmetric3x3 b) { b *= a; return b; }
};

struct Mat3x4 {
    Vector3 cols[4];

    MADRONA_HD inline Vector3 txfmPoint(Vector3 p) const
    {
        return cols[0] * p.x + cols[1] * p.y + cols[2] * p.z + cols[3];
    }

    MADRONA_HD inline Vector3 txfmDir(Vector3 p) const
    {
        return cols[0] * p.x + cols[1] * p.y + cols[2] * p.z;
    }

    MADRONA_HD inline Mat3x4 compose(const Mat3x4 &o) const
    {
        return Mat3x4 {{
            txfmDir(o.cols[0]), txfmDir(o.cols[1]), txfmDir(o.cols[2]),
            txfmPoint(o.cols[3]),
        }};
    }

    MADRONA_HD inline void decompose(Vector3 *translation, Quat *rotation,
                                     Diag3x3 *scale) const
    {
        Diag3x3 s { cols[0].length(), cols[1].length(), cols[2].length() };

        if (dot(cross(cols[0], cols[1]), cols[2]) < 0.f) {
            s.d0 *= -1.f;
        }

        // Gram-Schmidt on the normalized axes
        Vector3 v1 = cols[0] / s.d0;
        Vector3 v2 = cols[1] / s.d1;
        Vector3 v3 = cols[2] / s.d2;

        v2 = normalize(v2 - dot(v2, v1) * v1);
        v3 = v3 - dot(v3, v1) * v1;
        v3 -= dot(v3, v2) * v2;
        v3 = normalize(v3);

        *translation = cols[3];
        *rotation = Quat::fromBasis(v1, v2, v3);
        *scale = s;
    }

    MADRONA_HD static inline Mat3x4 fromRows(Vector4 row0, Vector4 row1, Vector4 row2)
    {
        return {{
            { row0.x, row1.x, row2.x },
            { row0.y, row1.y, row2.y },
            { row0.z, row1.z, row2.z },
            { row0.w, row1.w, row2.w },
        }};
    }

    MADRONA_HD static inline Mat3x4 fromTRS(Vector3 t, Quat r,
                                            Diag3x3 s = { 1.f, 1.f, 1.f })
    {
        Mat3x4 m;
        detail::rotScaleColumns(r, s, &m.cols[0], &m.cols[1], &m.cols[2]);
        m.cols[3] = t;
        return m;
    }

    MADRONA_HD static constexpr inline Mat3x4 identity()
    {
        return Mat3x4 {{ { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 }, { 0, 0, 0 } }};
    }
};

struct Mat4x4 {
    Vector4 cols[4];

    MADRONA_HD inline Vector4 txfmPoint(Vector4 p) const
    {
        return cols[0] * p.x + cols[1] * p.y + cols[2] * p.z + cols[3] * p.w;
    }

    MADRONA_HD inline Mat4x4 compose(const Mat4x4 &o) const
    {
        return Mat4x4 {{ txfmPoint(o.cols[0]), txfmPoint(o.cols[1]),
                         txfmPoint(o.cols[2]), txfmPoint(o.cols[3]) }};
    }

    MADRONA_HD static constexpr inline Mat4x4 identity()
    {
        return Mat4x4 {{ { 1, 0, 0, 0 }, { 0, 1, 0, 0 }, { 0, 0, 1, 0 },
                         { 0, 0, 0, 1 } }};
    }
};

struct AABB {
    Vector3 pMin;
    Vector3 pMax;

    MADRONA_HD inline float surfaceArea() const
    {
        Vector3 d = pMax - pMin;
        return 2.f * (d.x * d.y + d.x * d.z + d.y * d.z);
    }

    MADRONA_HD inline float distance2(const AABB &o) const
    {
        float dist2 = 0.f;
MADRONA_UNROLL
        for (CountT i = 0; i < 3; i++) {
            float isect_min = fmaxf(pMin[i], o.pMin[i]);
            float isect_max = fminf(pMax[i], o.pMax[i]);
            float diff = isect_min - isect_max;
            if (diff > 0) {
                dist2 += diff * diff;
            }
        }
        return dist2;
    }

    MADRONA_HD inline Vector3 centroid() const { return 0.5f * (pMin + pMax); }

    MADRONA_HD inline int maxDimension() const
    {
        Vector3 d = pMax - pMin;
        if (d.x > d.y && d.x > d.z) {
            return 0;
        }
        return d.y > d.z ? 1 : 2;
    }

    MADRONA_HD inline Vector3 offset(const Vector3 &p) const
    {
        Vector3 o = p - pMin;
        if (pMax.x > pMin.x) o.x /= pMax.x - pMin.x;
        if (pMax.y > pMin.y) o.y /= pMax.y - pMin.y;
        if (pMax.z > pMin.z) o.z /= pMax.z - pMin.z;
        return o;
    }

    MADRONA_HD inline bool overlaps(const AABB &o) const
    {
        return pMin.x < o.pMax.x && o.pMin.x < pMax.x &&
               pMin.y < o.pMax.y && o.pMin.y < pMax.y &&
               pMin.z < o.pMax.z && o.pMin.z < pMax.z;
    }

    // true when overlapping or exactly touching
    MADRONA_HD inline bool intersects(const AABB &o) const
    {
        bool x_sep = pMax.x < o.pMin.x || pMin.x > o.pMax.x;
        bool y_sep = pMax.y < o.pMin.y || pMin.y > o.pMax.y;
        bool z_sep = pMax.z < o.pMin.z || pMin.z > o.pMax.z;
        return !x_sep && !y_sep && !z_sep;
    }

    MADRONA_HD inline bool contains(const AABB &o) const
    {
        return pMin.x <= o.pMin.x && pMin.y <= o.pMin.y && pMin.z <= o.pMin.z &&
               pMax.x >= o.pMax.x && pMax.y >= o.pMax.y && pMax.z >= o.pMax.z;
    }

    MADRONA_HD inline bool contains(const Vector3 &p) const
    {
        return pMin.x <= p.x && pMin.y <= p.y && pMin.z <= p.z &&
               pMax.x >= p.x && pMax.y >= p.y && pMax.z >= p.z;
    }

    MADRONA_HD inline void expand(const Vector3 &p)
    {
        if (p.x < pMin.x) pMin.x = p.x; else if (p.x > pMax.x) pMax.x = p.x;
        if (p.y < pMin.y) pMin.y = p.y; else if (p.y > pMax.y) pMax.y = p.y;
        if (p.z < pMin.z) pMin.z = p.z; else if (p.z > pMax.z) pMax.z = p.z;
    }

    // slab test: max of the per-axis entry distances vs min of the exits
    MADRONA_HD inline bool rayIntersects(Vector3 ray_o, Diag3x3 inv_ray_d,
                                         float ray_t_min, float ray_t_max,
                                         float &hit_t, float &far_t)
    {
        Vector3 t_lower = inv_ray_d * (pMin - ray_o);
        Vector3 t_upper = inv_ray_d * (pMax - ray_o);

        Vector3 t_near = Vector3::min(t_lower, t_upper);
        Vector3 t_far = Vector3::max(t_lower, t_upper);

        float t_box_min =
            fmaxf(t_near.x, fmaxf(t_near.y, fmaxf(t_near.z, ray_t_min)));
        float t_box_max =
            fminf(t_far.x, fminf(t_far.y, fminf(t_far.z, ray_t_max)));

        if (t_box_min <= t_box_max) {
            hit_t = t_box_min;
            far_t = t_box_max;
            return true;
        }

        hit_t = FLT_MAX;
        far_t = FLT_MAX;
        return false;
    }

    MADRONA_HD inline bool rayIntersects(Vector3 ray_o, Diag3x3 inv_ray_d,
                                         float ray_t_min, float ray_t_max)
    {
        float hit_t, far_t;
        return rayIntersects(ray_o, inv_ray_d, ray_t_min, ray_t_max,
                             hit_t, far_t);
    }

    // AABB of this box after scale, rotation, translation (Arvo's method)
    [[nodiscard]] MADRONA_HD inline AABB applyTRS(
        const Vector3 &translation, const Quat &rotation,
        const Diag3x3 &scale = { 1, 1, 1 }) const
    {
        Mat3x3 rot_mat = Mat3x3::fromRS(rotation, scale);

        AABB txfmed;
MADRONA_UNROLL
        for (CountT i = 0; i < 3; i++) {
            txfmed.pMin[i] = txfmed.pMax[i] = translation[i];
MADRONA_UNROLL
            for (CountT j = 0; j < 3; j++) {
                // rot_mat is column major
                float e = rot_mat[j][i] * pMin[j];
                float f = rot_mat[j][i] * pMax[j];

                if (e < f) {
                    txfmed.pMin[i] += e;
                    txfmed.pMax[i] += f;
                } else {
                    txfmed.pMin[i] += f;
                    txfmed.pMax[i] += e;
                }
            }
        }

        return txfmed;
    }

    MADRONA_HD inline float operator[](CountT i) const
    {
        return i < 3 ? pMin[i] : pMax[i - 3];
    }

    MADRONA_HD static inline AABB invalid()
    {
        return AABB { Vector3 { FLT_MAX, FLT_MAX, FLT_MAX },
                      Vector3 { -FLT_MAX, -FLT_MAX, -FLT_MAX } };
    }

    MADRONA_HD static inline AABB point(const Vector3 &p) { return AABB { p, p }; }

    MADRONA_HD static inline AABB merge(const AABB &a, const AABB &b)
    {
        return AABB { Vector3::min(a.pMin, b.pMin), Vector3::max(a.pMax, b.pMax) };
    }
};

struct AABB2D {
    Vector2 pMin;
    Vector2 pMax;

    MADRONA_HD inline Vector2 centroid() const { return 0.5f * (pMin + pMax); }

    MADRONA_HD inline float area() const
    {
        Vector2 diff = pMax - pMin;
        return diff.x * diff.y;
    }
};

constexpr inline Vector3 up { 0, 0, 1 };
constexpr inline Vector3 fwd { 0, 1, 0 };
constexpr inline Vector3 right { 1, 0, 0 };

}

constexpr inline math::Vector3 worldUp = math::up;
constexpr inline math::Vector3 worldFwd = math::fwd;
constexpr inline math::Vector3 worldRight = math::right;

}

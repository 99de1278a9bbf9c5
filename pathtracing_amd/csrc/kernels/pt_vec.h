// f32 vector / matrix helpers shared by the host set-up code and the gfx950
// kernels.  Every function keeps the association of the reference's math crate
// (math/src/lib.rs) so that, with FP contraction off, results are bit-identical
// to the Rust worker's: dot is x*x' + y*y' + z*z' left to right (:247-249),
// cross per :251-257, Mat3*Vec3 is three dots of the rows (:453-458), Mat4*Vec4
// divides by the w dot (:651-659).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PT_HD __host__ __device__ __forceinline__
#else
#define PT_HD inline
#endif

namespace ptk {

constexpr float kEpsilon = 0.0001f;                 // math::EPSILON (lib.rs:4)
constexpr float kPi = 3.14159274101257324219f;      // math::PI
constexpr float kInvPi = 0.318309873342514038086f;  // math::INV_PI
constexpr float kF32Max = 3.40282346638528859812e+38f;

PT_HD float f32_inf() { return __builtin_inff(); }
PT_HD float sqrt_rn(float x) { return __builtin_sqrtf(x); }
PT_HD float abs_f(float x) { return __builtin_fabsf(x); }

struct V2 {
    float x, y;
};
struct V3 {
    float x, y, z;
};

PT_HD V3 mk3(float x, float y, float z) {
    V3 v;
    v.x = x;
    v.y = y;
    v.z = z;
    return v;
}
PT_HD V2 mk2(float x, float y) {
    V2 v;
    v.x = x;
    v.y = y;
    return v;
}
PT_HD V3 splat3(float s) { return mk3(s, s, s); }
PT_HD V3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
PT_HD V3 operator+(V3 a, V3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_HD V3 operator-(V3 a, V3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_HD V3 operator*(V3 a, V3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_HD V3 operator*(V3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
PT_HD V3 operator*(float s, V3 a) { return mk3(a.x * s, a.y * s, a.z * s); }
PT_HD V3 operator/(V3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
PT_HD V2 operator+(V2 a, V2 b) { return mk2(a.x + b.x, a.y + b.y); }
PT_HD V2 operator*(V2 a, float s) { return mk2(a.x * s, a.y * s); }
PT_HD V2 operator*(V2 a, V2 b) { return mk2(a.x * b.x, a.y * b.y); }

PT_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PT_HD V3 cross(V3 a, V3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
PT_HD float length(V3 a) { return sqrt_rn(dot(a, a)); }
PT_HD V3 normalized(V3 a) { return a / length(a); }
// lib.rs:259-261
PT_HD V3 reflect(V3 self, V3 relative) { return self - (2.0f * dot(self, relative)) * relative; }

struct M3 {
    V3 row0, row1, row2;
};
PT_HD V3 mul(const M3& a, V3 rhs) { return mk3(dot(rhs, a.row0), dot(rhs, a.row1), dot(rhs, a.row2)); }

// Row-major 4x4; m[r][c].
struct M4 {
    float m[4][4];
};
PT_HD float dot4(const float* row, float x, float y, float z, float w) {
    return x * row[0] + y * row[1] + z * row[2] + w * row[3];
}
// &Mat4 * Vec4::from_vec3(p) (lib.rs:519-526, :651-659)
PT_HD V3 mul_point(const M4& a, V3 p) {
    float w = dot4(a.m[3], p.x, p.y, p.z, 1.0f);
    V3 r = mk3(dot4(a.m[0], p.x, p.y, p.z, 1.0f), dot4(a.m[1], p.x, p.y, p.z, 1.0f),
               dot4(a.m[2], p.x, p.y, p.z, 1.0f));
    return r / w;
}

// worker/src/ray.rs:3-9
struct Ray {
    V3 source, direction;
    float min, max;
};

// Saturating f32 -> u32 (Rust `as u32`): NaN and negatives give 0.
PT_HD uint32_t f32_as_u32(float v) {
    if (!(v > 0.0f)) return 0u;
    if (v >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)v;
}
// f32::clamp: NaN stays NaN.
PT_HD float clamp_f(float v, float lo, float hi) {
    if (v < lo) return lo;
    if (v > hi) return hi;
    return v;
}
// f32::signum: +-1 by sign bit, NaN stays NaN.
PT_HD float signum_f(float x) {
    if (x != x) return x;
    return __builtin_signbit(x) ? -1.0f : 1.0f;
}

} // namespace ptk

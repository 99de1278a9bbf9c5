// Portable single-precision sin / cos / asin / acos / atan2.
//
// Every function here is built from IEEE-754 f32 add, subtract, multiply,
// divide, square root, comparisons and int<->float conversions only, so a
// translation unit compiled with FP contraction off produces the same bits on
// an x86 host (g++) and on gfx950 (hipcc).  That is what lets a GPU render be
// compared bit for bit with a CPU render: the reference worker calls the
// platform libm for these (math/src/lib.rs:272-300, worker/src/camera.rs:24,
// worker/src/scene/hierarchy/sphere.rs:93-94), and libm and the device math
// library differ in the last place.
//
// Method: Cody-Waite argument reduction by multiples of pi/4 followed by
// minimax polynomials on [-pi/4, pi/4] (sin, cos); polynomial on [0, 0.5] plus
// the half-angle identity (asin, acos); two-step range reduction plus an odd
// polynomial (atan).  Accuracy is about 1-2 ulp against a correctly rounded
// result over the argument ranges the renderer uses (|x| <= 8192 for sin/cos).
#ifndef PT_PORTABLE_MATH_H
#define PT_PORTABLE_MATH_H

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PT_HD __host__ __device__ inline
#else
#define PT_HD inline
#endif

namespace ptm {

constexpr float kPi = 3.14159274101257324219f;       // f32 nearest to pi
constexpr float kPiOver2 = 1.57079637050628662109f;  // f32 nearest to pi/2
constexpr float kPiOver4 = 0.785398185253143310547f; // f32 nearest to pi/4
constexpr float kFourOverPi = 1.27323949337005615234f;

// pi/4 split in three parts; the first two have few significant bits so that
// j * part is exact for the small integers j that occur.
constexpr float kDP1 = 0.78515625f;
constexpr float kDP2 = 2.4187564849853515625e-4f;
constexpr float kDP3 = 3.77489497744594108e-8f;

// Correctly rounded on the host, and on the device as long as the translation
// unit is built with -fhip-fp32-correctly-rounded-divide-sqrt (hipcc's default).
PT_HD float sqrt_f32(float x) { return __builtin_sqrtf(x); }

PT_HD float abs_f32(float x) { return __builtin_fabsf(x); }

// sin and cos polynomial kernels on |r| <= pi/4, z = r*r.
PT_HD float sin_poly(float r, float z) {
    float y = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r;
    return y + r;
}
PT_HD float cos_poly(float z) {
    float y = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z;
    y = y - 0.5f * z;
    return y + 1.0f;
}

// Reduces |x| to r in [-pi/4, pi/4] and the octant j (even, 0..6 after masking).
PT_HD float reduce_pio4(float ax, int* octant) {
    int j = (int)(kFourOverPi * ax);
    float y = (float)j;
    if (j & 1) {
        j += 1;
        y += 1.0f;
    }
    *octant = j & 7;
    return ((ax - y * kDP1) - y * kDP2) - y * kDP3;
}

// sin and cos of one argument from one reduction and one evaluation of the two
// polynomials.  sin_f32 / cos_f32 are this function with one result dropped, so
// calling it is interchangeable with calling them, bit for bit.
PT_HD void sincos_f32(float x, float* s, float* c) {
    bool neg = x < 0.0f;
    float ax = neg ? -x : x;
    if (!(ax <= 8192.0f)) { // out of the supported range (also NaN)
        *s = 0.0f;
        *c = 0.0f;
        return;
    }
    int j;
    float r = reduce_pio4(ax, &j);
    bool half_turn = j > 3; // octants 4..7: both functions change sign
    if (half_turn) j -= 4;
    bool swap = j == 2;     // octants 2,3 (6,7): sin takes the cos polynomial and vice versa
    float z = r * r;
    float sp = sin_poly(r, z), cp = cos_poly(z);
    float ys = swap ? cp : sp;
    float yc = swap ? sp : cp;
    bool sin_neg = neg != half_turn;
    bool cos_neg = half_turn != swap;
    *s = sin_neg ? -ys : ys;
    *c = cos_neg ? -yc : yc;
}

PT_HD float sin_f32(float x) {
    float s, c;
    sincos_f32(x, &s, &c);
    return s;
}

PT_HD float cos_f32(float x) {
    float s, c;
    sincos_f32(x, &s, &c);
    return c;
}

PT_HD float asin_f32(float x) {
    float a = abs_f32(x);
    if (!(a <= 1.0f)) return __builtin_nanf("");
    float z, w;
    bool big = a > 0.5f;
    if (a < 1.0e-4f) return x;
    if (big) {
        z = 0.5f * (1.0f - a);
        w = sqrt_f32(z);
    } else {
        w = a;
        z = w * w;
    }
    float p = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z +
                7.4953002686e-2f) * z + 1.6666752422e-1f) * z * w + w;
    if (big) {
        p = p + p;
        p = kPiOver2 - p;
    }
    return x < 0.0f ? -p : p;
}

// acos(x) = pi - 2 asin(sqrt((1+x)/2)) below -0.5, 2 asin(sqrt((1-x)/2)) above 0.5,
// pi/2 - asin(x) between; in all three the asin argument is at most 0.5 in
// magnitude, so one evaluation of asin's small-argument form serves them all.
PT_HD float acos_f32(float x) {
    if (!(abs_f32(x) <= 1.0f)) return __builtin_nanf("");
    bool low = x < -0.5f, high = x > 0.5f;
    float arg = x;
    if (low || high) arg = sqrt_f32(0.5f * (1.0f + (low ? x : -x)));
    float w = abs_f32(arg);
    float r = arg;
    if (!(w < 1.0e-4f)) {
        float z = w * w;
        float p = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z +
                    7.4953002686e-2f) * z + 1.6666752422e-1f) * z * w + w;
        r = arg < 0.0f ? -p : p;
    }
    if (low) return kPi - 2.0f * r;
    if (high) return 2.0f * r;
    return kPiOver2 - r;
}

PT_HD float atan_f32(float x) {
    float a = abs_f32(x);
    float y;
    if (a > 2.414213562373095f) { // tan(3 pi / 8)
        y = kPiOver2;
        a = -(1.0f / a);
    } else if (a > 0.4142135623730950f) { // tan(pi / 8)
        y = kPiOver4;
        a = (a - 1.0f) / (a + 1.0f);
    } else {
        y = 0.0f;
    }
    float z = a * a;
    y = y + ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z -
              3.33329491539e-1f) * z * a + a);
    return x < 0.0f ? -y : y;
}

// atan2(y, x) with the usual quadrant conventions; signed zeros of y are not
// distinguished on the negative x axis (returns +pi there).
PT_HD float atan2_f32(float y, float x) {
    if (x != x || y != y) return __builtin_nanf("");
    if (x == 0.0f) {
        if (y > 0.0f) return kPiOver2;
        if (y < 0.0f) return -kPiOver2;
        return 0.0f;
    }
    if (y == 0.0f) return x > 0.0f ? 0.0f : kPi;
    float base = atan_f32(y / x);
    if (x > 0.0f) return base;
    return y < 0.0f ? base - kPi : base + kPi;
}

} // namespace ptm
#endif

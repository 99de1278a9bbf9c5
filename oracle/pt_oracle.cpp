// pt_oracle.cpp -- CPU oracle for the path-tracing hot path.
//
// TEST INFRASTRUCTURE ONLY.  This file restates, on the CPU, the algorithm of
// the mertwole/PathTracing `worker` render loop so that the HIP implementation
// can be checked against it.  Nothing in the shipped library may include,
// link or call it: only tests/, __graft_entry__.smoke() and the cpu_baseline
// leg of bench.py do.
//
// Each function cites the reference lines it follows (paths relative to the
// reference checkout).  Arithmetic is IEEE f32 in the reference's exact
// association; build with -O2 -ffp-contract=off -fno-fast-math (Rust never
// contracts to FMA).
//
// Pinning: the reference ships no tests, no golden vectors and cannot be
// compiled here (Rust; no toolchain).  The oracle is pinned by known answers
// derived by hand from the cited source (tests/test_oracle_kat.py).  Three
// third-party behaviours are "parity unpinned": the rand 0.8 generator (the
// reference never seeds it, so only the distribution and the per-event draw
// order are restated; the generator here is this project's own counter-based
// one), tobj's triangulation order and `image`'s PNG decode (restated in
// oracle/loaders.py, which uses oracle_png_unfilter below).
//
// Two switchable deviations from a literal restatement, both off by default:
//   accel = 1   meshes are searched through a BVH instead of the linear scan of
//               mesh/mod.rs:57-62; the result is required to be identical.
//   math  = 1   sin/cos/acos/asin/atan2 come from pt_portable_math.h (the same
//               f32-only routines the GPU kernels use) instead of libm, which
//               makes a CPU render comparable bit for bit with a GPU render.
#include "../include/pathtracing_amd.h"
#include "../pathtracing_amd/csrc/pt_portable_math.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------- math crate
// math/src/lib.rs:4-6
constexpr float EPSILON = 0.0001f;
constexpr float PI = 3.14159274101257324219f;
constexpr float INV_PI = 0.318309873342514038086f;
constexpr float F32_MAX = std::numeric_limits<float>::max();
constexpr float F32_INF = std::numeric_limits<float>::infinity();

struct Trig {
    float (*sin)(float);
    float (*cos)(float);
    float (*acos)(float);
    float (*asin)(float);
    float (*atan2)(float, float);
};
float libm_sin(float x) { return sinf(x); }
float libm_cos(float x) { return cosf(x); }
float libm_acos(float x) { return acosf(x); }
float libm_asin(float x) { return asinf(x); }
float libm_atan2(float y, float x) { return atan2f(y, x); }
const Trig kLibm = {libm_sin, libm_cos, libm_acos, libm_asin, libm_atan2};
const Trig kPortable = {ptm::sin_f32, ptm::cos_f32, ptm::acos_f32, ptm::asin_f32, ptm::atan2_f32};

struct V2 {
    float x = 0, y = 0;
};
// math/src/lib.rs:147-194
inline V2 operator+(V2 a, V2 b) { return {a.x + b.x, a.y + b.y}; }
inline V2 operator*(V2 a, float s) { return {a.x * s, a.y * s}; }
inline V2 operator*(V2 a, V2 b) { return {a.x * b.x, a.y * b.y}; }

struct V3 {
    float x = 0, y = 0, z = 0;
};
// math/src/lib.rs:320-367
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return a * s; } // lib.rs:348-353: rhs * self
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline V3 splat(float v) { return {v, v, v}; }
// lib.rs:247-249
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// lib.rs:251-257
inline V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float length(V3 a) { return std::sqrt(dot(a, a)); }   // lib.rs:235-241
inline V3 normalized(V3 a) { return a / length(a); }         // lib.rs:243-245
// lib.rs:259-261
inline V3 reflect(V3 self, V3 relative) { return self - 2.0f * dot(self, relative) * relative; }
// lib.rs:263-270
inline bool refract(V3 self, V3 normal, float eta, V3* out) {
    float d = dot(normal, self);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return false;
    *out = eta * self - normal * (eta * d + std::sqrt(k));
    return true;
}
// lib.rs:272-280
inline V3 random_on_unit_sphere(const Trig& m, float rand_0, float rand_1) {
    float theta = rand_0 * PI * 2.0f;
    float phi = m.acos((2.0f * rand_1) - 1.0f);
    float x = m.sin(phi) * m.cos(theta);
    float y = m.sin(phi) * m.sin(theta);
    float z = m.cos(phi);
    return {x, y, z};
}
// lib.rs:282-289
inline V3 random_on_hemisphere(const Trig& m, float rand_0, float rand_1, V3 normal) {
    V3 r = random_on_unit_sphere(m, rand_0, rand_1);
    if (dot(r, normal) < 0.0f) return r * -1.0f;
    return r;
}
// lib.rs:196-212 (Basis) and :291-300
inline V3 cosine_weighted_random_on_hemisphere(const Trig& m, float rand_0, float rand_1, V3 normal) {
    float r0 = m.cos(PI * 0.5f * (1.0f - rand_0));
    float theta = rand_1 * PI * 2.0f;
    V3 j = normal;
    V3 i = (std::fabs(j.y) + std::fabs(j.z) < EPSILON) ? V3{-j.y, j.x, 0.0f} : V3{0.0f, j.z, -j.y};
    V3 k = cross(j, i);
    float r = std::sqrt(r0);
    V3 rnd = j * (1.0f - r0 + EPSILON);
    return rnd + r * (i * m.cos(theta) + k * m.sin(theta));
}

struct M3 {
    V3 row0, row1, row2;
};
// lib.rs:426-430
inline float det(const M3& a) {
    return a.row0.x * (a.row1.y * a.row2.z - a.row2.y * a.row1.z) -
           a.row0.y * (a.row1.x * a.row2.z - a.row1.z * a.row2.x) +
           a.row0.z * (a.row1.x * a.row2.y - a.row1.y * a.row2.x);
}
// lib.rs:432-450
inline M3 transponsed_inverse(const M3& s) {
    M3 c;
    c.row0 = {s.row1.y * s.row2.z - s.row2.y * s.row1.z, s.row1.z * s.row2.x - s.row1.x * s.row2.z,
              s.row1.x * s.row2.y - s.row2.x * s.row1.y};
    c.row1 = {s.row0.z * s.row2.y - s.row0.y * s.row2.z, s.row0.x * s.row2.z - s.row0.z * s.row2.x,
              s.row2.x * s.row0.y - s.row0.x * s.row2.y};
    c.row2 = {s.row0.y * s.row1.z - s.row0.z * s.row1.y, s.row1.x * s.row0.z - s.row0.x * s.row1.z,
              s.row0.x * s.row1.y - s.row1.x * s.row0.y};
    float f = 1.0f / det(s);
    return {c.row0 * f, c.row1 * f, c.row2 * f}; // lib.rs:460-465
}
// lib.rs:453-458
inline V3 mul(const M3& a, V3 rhs) { return {dot(rhs, a.row0), dot(rhs, a.row1), dot(rhs, a.row2)}; }

struct V4 {
    float x = 0, y = 0, z = 0, w = 0;
};
inline V4 from_vec3(V3 v) { return {v.x, v.y, v.z, 1.0f}; }                 // lib.rs:519-526
inline V3 to_vec3(V4 v) { return {v.x, v.y, v.z}; }                        // lib.rs:528-530
inline float dot(V4 a, V4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; } // :532-534
inline V4 operator*(V4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }

struct M4 {
    V4 row0, row1, row2, row3;
};
inline const V4& row(const M4& a, int r) { return r == 0 ? a.row0 : r == 1 ? a.row1 : r == 2 ? a.row2 : a.row3; }
// lib.rs:554-562
inline M3 normal_matrix(const M4& a) {
    M3 ul{to_vec3(a.row0), to_vec3(a.row1), to_vec3(a.row2)};
    return transponsed_inverse(ul);
}
// lib.rs:564-584
inline float get_minor(const M4& a, int r, int c) {
    V3 rows[3];
    int n = 0;
    for (int i = 0; i < 4; i++) {
        if (i == r) continue;
        const V4& q = row(a, i);
        switch (c) {
        case 0: rows[n] = {q.y, q.z, q.w}; break;
        case 1: rows[n] = {q.x, q.z, q.w}; break;
        case 2: rows[n] = {q.x, q.y, q.w}; break;
        default: rows[n] = {q.x, q.y, q.z}; break;
        }
        n++;
    }
    return det(M3{rows[0], rows[1], rows[2]});
}
// lib.rs:586-615
inline M4 transponsed_inverse(const M4& a) {
    M4 c;
    c.row0 = {get_minor(a, 0, 0), -get_minor(a, 0, 1), get_minor(a, 0, 2), -get_minor(a, 0, 3)};
    c.row1 = {-get_minor(a, 1, 0), get_minor(a, 1, 1), -get_minor(a, 1, 2), get_minor(a, 1, 3)};
    c.row2 = {get_minor(a, 2, 0), -get_minor(a, 2, 1), get_minor(a, 2, 2), -get_minor(a, 2, 3)};
    c.row3 = {-get_minor(a, 3, 0), get_minor(a, 3, 1), -get_minor(a, 3, 2), get_minor(a, 3, 3)};
    float d = dot(a.row0, c.row0);
    float f = 1.0f / d;
    return {c.row0 * f, c.row1 * f, c.row2 * f, c.row3 * f}; // lib.rs:661-671
}
// lib.rs:617-644
inline M4 transponse(const M4& a) {
    return {{a.row0.x, a.row1.x, a.row2.x, a.row3.x},
            {a.row0.y, a.row1.y, a.row2.y, a.row3.y},
            {a.row0.z, a.row1.z, a.row2.z, a.row3.z},
            {a.row0.w, a.row1.w, a.row2.w, a.row3.w}};
}
inline M4 inverse(const M4& a) { return transponse(transponsed_inverse(a)); } // lib.rs:646-648
// lib.rs:651-659
inline V3 mul(const M4& a, V4 rhs) {
    float w = dot(rhs, a.row3);
    return V3{dot(rhs, a.row0), dot(rhs, a.row1), dot(rhs, a.row2)} / w;
}

inline V3 v3(const PtVec3& p) { return {p.x, p.y, p.z}; }
inline PtVec3 pv3(V3 p) { return {p.x, p.y, p.z}; }
inline M4 m4(const PtMat4& p) {
    M4 a;
    a.row0 = {p.m[0][0], p.m[0][1], p.m[0][2], p.m[0][3]};
    a.row1 = {p.m[1][0], p.m[1][1], p.m[1][2], p.m[1][3]};
    a.row2 = {p.m[2][0], p.m[2][1], p.m[2][2], p.m[2][3]};
    a.row3 = {p.m[3][0], p.m[3][1], p.m[3][2], p.m[3][3]};
    return a;
}

// ---------------------------------------------------------------------- rays
// worker/src/ray.rs:3-9
struct Ray {
    V3 source, direction;
    float min, max;
};

// worker/src/renderer/cpu_renderer/mod.rs:15-39 (+ two bookkeeping fields the
// reference does not have: which leaf node and which triangle produced the hit)
struct Result {
    bool hit = false;
    bool hit_inside = false;
    V3 point;
    V3 normal;
    V2 uv;
    float t = 0.0f;
    uint32_t material_id = 0;
    uint32_t node_index = PT_NO_INDEX;
    uint32_t tri_index = PT_NO_INDEX;
};

// ------------------------------------------------------------------- RNG
// The reference draws from rand 0.8 thread_rng() and is never seeded
// (camera.rs:21,28,52; material/base.rs:94,114,133-134; material/pbr.rs:109-112),
// so there is no stream to reproduce.  This generator is this project's own:
// draw c of stream s under seed k is the high half of
//   mix64(key + (c + 1) * G),  key = mix64(mix64(k + G) ^ s),  G = 0x9E3779B97F4A7C15
// with mix64 the SplitMix64 finaliser.  It keeps rand's output lattice:
// gen_range(lo..hi) on f32 is  u * (hi - lo) + lo  with u = (draw >> 9) * 2^-23
// (rand 0.8 UniformFloat::sample_single; its re-draw branch `res >= hi` cannot
// be taken for the three ranges the reference uses: -0.5..0.5, 0..1, 0..2pi),
// and gen_bool(0.5) is one draw compared with half the integer range.
constexpr uint64_t kGolden = 0x9E3779B97F4A7C15ull;
inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
struct Rng {
    uint64_t key;
    uint32_t counter;
    Rng(uint64_t seed, uint64_t stream) : key(mix64(mix64(seed + kGolden) ^ stream)), counter(0) {}
    uint32_t next_u32() {
        counter++;
        return (uint32_t)(mix64(key + (uint64_t)counter * kGolden) >> 32);
    }
    float gen_range(float lo, float hi) {
        float scale = hi - lo;
        float u = (float)(next_u32() >> 9) * 1.1920928955078125e-7f; // exact: k * 2^-23
        return u * scale + lo;
    }
    bool gen_bool_half() { return (next_u32() >> 31) == 0; }
};
inline uint64_t stream_id(uint64_t sample, uint64_t pixel_linear) { return (sample << 32) | pixel_linear; }

// ----------------------------------------------------------------- resources
// mesh/triangle/vertex.rs:8-15, triangle/mod.rs:19-24
struct Tri {
    V3 p[3];
    V2 uv[3];
    V3 n[3];
    V3 true_normal;
};

struct BvhNode {
    float bmin[3], bmax[3];
    int left, right; // inner: child node indices; leaf: left = -1 - first, right = count
};
struct Bvh {
    std::vector<BvhNode> nodes;
    std::vector<uint32_t> order; // leaf order -> triangle index
};

struct Mesh {
    std::vector<Tri> tris;
    Bvh bvh; // built lazily (accel mode only)
    bool bvh_built = false;
};

struct Image {
    uint32_t w = 0, h = 0;
    std::vector<uint8_t> rgba;
};

struct Material {
    PtMaterialDesc d;
    V3 f0;               // pbr.rs:93
    float roughness_sqr; // pbr.rs:94
};

struct Node {
    uint32_t type = 0;
    uint32_t index = 0; // position in the pre-order description
    std::vector<Node> children;
    // transform.rs:33-40
    M4 matrix, matrix_inverse;
    M3 normal_mat;
    M3 inverse_normal_mat; // normal_matrix() of matrix_inverse, recomputed per ray at ray.rs:29
    uint32_t mesh_id = 0, material_id = 0;
    V3 center;
    float radius = 0, radius_sqr = 0;
    V3 point, normal, tangent, bitangent;
};

struct Stats {
    uint64_t rays = 0, box_tests = 0, tri_tests = 0, hits = 0, tex_fetches = 0, paths = 0;
};

} // namespace

struct OracleScene {
    Node root;
    std::vector<Mesh> meshes;
    std::vector<Material> materials;
    std::vector<Image> images;
    std::string error;
};

namespace {

thread_local std::string g_error;

// triangle/mod.rs:65-84
Tri triangle_init(const float* pos, const float* nrm, bool nrm_valid, const float* uv) {
    Tri t;
    for (int i = 0; i < 3; i++) t.p[i] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
    V3 side_0 = t.p[1] - t.p[0];
    V3 side_1 = t.p[1] - t.p[2];
    t.true_normal = normalized(cross(side_0, side_1));
    for (int i = 0; i < 3; i++) {
        t.uv[i] = uv ? V2{uv[2 * i], uv[2 * i + 1]} : V2{0.0f, 0.0f}; // unwrap_or_default, :42
        t.n[i] = (nrm && nrm_valid) ? V3{nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]} : t.true_normal; // :77
    }
    return t;
}

// triangle/mod.rs:90-127 + :130-168
inline Result triangle_trace(const Tri& tr, const Ray& ray) {
    Result result;
    V3 edge0 = tr.p[1] - tr.p[0];
    V3 edge1 = tr.p[2] - tr.p[0];
    V3 pvec = cross(ray.direction, edge1);
    float determinant = dot(edge0, pvec);
    if (determinant < EPSILON && determinant > -EPSILON) return result;
    float inv_determinant = 1.0f / determinant;
    V3 tvec = ray.source - tr.p[0];
    float u = dot(tvec, pvec) * inv_determinant;
    if (!(u >= 0.0f && u <= 1.0f)) return result; // !(0.0..=1.0).contains(&u)
    V3 qvec = cross(tvec, edge0);
    float v = dot(ray.direction, qvec) * inv_determinant;
    if (v < 0.0f || u + v > 1.0f) return result;
    result.t = dot(edge1, qvec) * inv_determinant;
    if (result.t < ray.min || result.t > ray.max) return result;

    result.point = ray.source + ray.direction * result.t;
    result.hit = true;

    // get_barycentric_coords, :91-114
    V3 to0 = tr.p[0] - result.point, to1 = tr.p[1] - result.point, to2 = tr.p[2] - result.point;
    float b0 = std::fabs(dot(cross(to1, to2), tr.true_normal));
    float b1 = std::fabs(dot(cross(to0, to2), tr.true_normal));
    float b2 = std::fabs(dot(cross(to0, to1), tr.true_normal));
    float sum = b0 + b1 + b2; // iter().sum(): ((0 + b0) + b1) + b2
    float c0 = b0 / sum, c1 = b1 / sum, c2 = b2 / sum;
    // get_uv :116-120, get_normal :122-126
    result.uv = tr.uv[0] * c0 + tr.uv[1] * c1 + tr.uv[2] * c2;
    result.normal = tr.n[0] * c0 + tr.n[1] * c1 + tr.n[2] * c2;
    return result;
}

// mesh/mod.rs:53-65
Result mesh_trace_linear(const Mesh& mesh, const Ray& ray, Stats* st) {
    Result result;
    result.t = F32_INF;
    const size_t n = mesh.tris.size();
    for (size_t i = 0; i < n; i++) {
        Result r = triangle_trace(mesh.tris[i], ray);
        if (r.hit && r.t < result.t) {
            result = r;
            result.tri_index = (uint32_t)i;
        }
    }
    if (st) st->tri_tests += n;
    return result;
}

// ---- canonical BVH2 (binned SAH, <= 4 triangles per leaf) used by accel mode
struct BuildPrim {
    float bmin[3], bmax[3], c[3];
    uint32_t idx;
};

void tri_bounds(const Tri& t, float* mn, float* mx) {
    for (int a = 0; a < 3; a++) {
        float v0 = a == 0 ? t.p[0].x : a == 1 ? t.p[0].y : t.p[0].z;
        float v1 = a == 0 ? t.p[1].x : a == 1 ? t.p[1].y : t.p[1].z;
        float v2 = a == 0 ? t.p[2].x : a == 1 ? t.p[2].y : t.p[2].z;
        float lo = std::min(v0, std::min(v1, v2)), hi = std::max(v0, std::max(v1, v2));
        // Padding keeps the box a superset of the points the f32 Moller-Trumbore
        // test can accept near the triangle's border.
        float pad = 1e-3f * (hi - lo) + 1e-5f * std::max(std::fabs(lo), std::fabs(hi)) + 1e-6f;
        mn[a] = lo - pad;
        mx[a] = hi + pad;
    }
}

float half_area(const float* mn, const float* mx) {
    float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    return dx * dy + dy * dz + dz * dx;
}

int bvh_build_rec(Bvh& bvh, std::vector<BuildPrim>& prims, size_t lo, size_t hi, int depth) {
    int me = (int)bvh.nodes.size();
    bvh.nodes.push_back({});
    float mn[3] = {F32_INF, F32_INF, F32_INF}, mx[3] = {-F32_INF, -F32_INF, -F32_INF};
    float cmn[3] = {F32_INF, F32_INF, F32_INF}, cmx[3] = {-F32_INF, -F32_INF, -F32_INF};
    for (size_t i = lo; i < hi; i++)
        for (int a = 0; a < 3; a++) {
            mn[a] = std::min(mn[a], prims[i].bmin[a]);
            mx[a] = std::max(mx[a], prims[i].bmax[a]);
            cmn[a] = std::min(cmn[a], prims[i].c[a]);
            cmx[a] = std::max(cmx[a], prims[i].c[a]);
        }
    for (int a = 0; a < 3; a++) {
        bvh.nodes[me].bmin[a] = mn[a];
        bvh.nodes[me].bmax[a] = mx[a];
    }
    size_t n = hi - lo;
    if (n <= 4) {
        bvh.nodes[me].left = -1 - (int)bvh.order.size();
        bvh.nodes[me].right = (int)n;
        for (size_t i = lo; i < hi; i++) bvh.order.push_back(prims[i].idx);
        return me;
    }
    constexpr int kBins = 16;
    int best_axis = -1, best_split = -1;
    float best_cost = F32_INF;
    for (int a = 0; a < 3; a++) {
        float ext = cmx[a] - cmn[a];
        if (!(ext > 0.0f)) continue;
        float bmn[kBins][3], bmx[kBins][3];
        int cnt[kBins];
        for (int b = 0; b < kBins; b++) {
            cnt[b] = 0;
            for (int k = 0; k < 3; k++) {
                bmn[b][k] = F32_INF;
                bmx[b][k] = -F32_INF;
            }
        }
        float scale = (float)kBins / ext;
        for (size_t i = lo; i < hi; i++) {
            int b = (int)((prims[i].c[a] - cmn[a]) * scale);
            b = std::max(0, std::min(kBins - 1, b));
            cnt[b]++;
            for (int k = 0; k < 3; k++) {
                bmn[b][k] = std::min(bmn[b][k], prims[i].bmin[k]);
                bmx[b][k] = std::max(bmx[b][k], prims[i].bmax[k]);
            }
        }
        float rarea[kBins];
        int rcnt[kBins];
        float amn[3] = {F32_INF, F32_INF, F32_INF}, amx[3] = {-F32_INF, -F32_INF, -F32_INF};
        int c = 0;
        for (int b = kBins - 1; b >= 1; b--) {
            for (int k = 0; k < 3; k++) {
                amn[k] = std::min(amn[k], bmn[b][k]);
                amx[k] = std::max(amx[k], bmx[b][k]);
            }
            c += cnt[b];
            rarea[b] = c ? half_area(amn, amx) : 0.0f;
            rcnt[b] = c;
        }
        for (int k = 0; k < 3; k++) {
            amn[k] = F32_INF;
            amx[k] = -F32_INF;
        }
        c = 0;
        for (int b = 0; b < kBins - 1; b++) {
            for (int k = 0; k < 3; k++) {
                amn[k] = std::min(amn[k], bmn[b][k]);
                amx[k] = std::max(amx[k], bmx[b][k]);
            }
            c += cnt[b];
            if (c == 0 || rcnt[b + 1] == 0) continue;
            float cost = half_area(amn, amx) * (float)c + rarea[b + 1] * (float)rcnt[b + 1];
            if (cost < best_cost) {
                best_cost = cost;
                best_axis = a;
                best_split = b;
            }
        }
    }
    size_t mid;
    if (best_axis < 0 || depth > 48) {
        // all centroids coincide (or the tree is getting too deep): median split
        mid = lo + n / 2;
        int a = 0;
        float e = -1.0f;
        for (int k = 0; k < 3; k++)
            if (cmx[k] - cmn[k] > e) {
                e = cmx[k] - cmn[k];
                a = k;
            }
        std::nth_element(prims.begin() + lo, prims.begin() + mid, prims.begin() + hi,
                         [a](const BuildPrim& p, const BuildPrim& q) {
                             return p.c[a] < q.c[a] || (p.c[a] == q.c[a] && p.idx < q.idx);
                         });
    } else {
        float ext = cmx[best_axis] - cmn[best_axis];
        float scale = (float)kBins / ext;
        float base = cmn[best_axis];
        int a = best_axis, sp = best_split;
        auto it = std::partition(prims.begin() + lo, prims.begin() + hi, [=](const BuildPrim& p) {
            int b = (int)((p.c[a] - base) * scale);
            b = std::max(0, std::min(kBins - 1, b));
            return b <= sp;
        });
        mid = (size_t)(it - prims.begin());
        if (mid == lo || mid == hi) mid = lo + n / 2;
    }
    int l = bvh_build_rec(bvh, prims, lo, mid, depth + 1);
    int r = bvh_build_rec(bvh, prims, mid, hi, depth + 1);
    bvh.nodes[me].left = l;
    bvh.nodes[me].right = r;
    return me;
}

void bvh_build(Mesh& mesh) {
    std::vector<BuildPrim> prims(mesh.tris.size());
    for (size_t i = 0; i < prims.size(); i++) {
        tri_bounds(mesh.tris[i], prims[i].bmin, prims[i].bmax);
        for (int a = 0; a < 3; a++) prims[i].c[a] = 0.5f * (prims[i].bmin[a] + prims[i].bmax[a]);
        prims[i].idx = (uint32_t)i;
    }
    mesh.bvh.nodes.clear();
    mesh.bvh.order.clear();
    mesh.bvh.nodes.reserve(prims.size());
    mesh.bvh.order.reserve(prims.size());
    if (!prims.empty()) bvh_build_rec(mesh.bvh, prims, 0, prims.size(), 0);
    mesh.bvh_built = true;
}

// Conservative slab test: returns false only when the ray misses the box on
// the interval [0, limit]; NaNs (0 * inf) never cull.
inline bool box_hit(const BvhNode& nd, const Ray& ray, const float* inv, float limit) {
    float t0 = 0.0f, t1 = limit;
    const float o[3] = {ray.source.x, ray.source.y, ray.source.z};
    for (int a = 0; a < 3; a++) {
        float ta = (nd.bmin[a] - o[a]) * inv[a];
        float tb = (nd.bmax[a] - o[a]) * inv[a];
        if (ta != ta || tb != tb) continue;
        float lo = ta < tb ? ta : tb, hi = ta < tb ? tb : ta;
        // widen by a few ulps: the slab arithmetic itself rounds
        lo = lo - std::fabs(lo) * 4e-7f;
        hi = hi + std::fabs(hi) * 4e-7f;
        if (lo > t0) t0 = lo;
        if (hi < t1) t1 = hi;
    }
    return t0 <= t1;
}

// Same answer as mesh_trace_linear, found through the BVH.
Result mesh_trace_bvh(const Mesh& mesh, const Ray& ray, Stats* st) {
    Result result;
    result.t = F32_INF;
    if (mesh.bvh.nodes.empty()) return result;
    // Non-finite rays miss every triangle (u comes out infinite or NaN in
    // triangle_trace), and NaN must not cull in the slab test: answer directly
    // instead of walking the whole tree.
    if (!std::isfinite(ray.source.x) || !std::isfinite(ray.source.y) || !std::isfinite(ray.source.z) ||
        !std::isfinite(ray.direction.x) || !std::isfinite(ray.direction.y) || !std::isfinite(ray.direction.z))
        return result;
    float inv[3] = {1.0f / ray.direction.x, 1.0f / ray.direction.y, 1.0f / ray.direction.z};
    int stack[128];
    int sp = 0;
    stack[sp++] = 0;
    uint64_t boxes = 0, tris = 0;
    while (sp) {
        const BvhNode& nd = mesh.bvh.nodes[stack[--sp]];
        boxes++;
        float limit = result.t; // +inf until the first hit
        if (ray.max == ray.max && ray.max < limit) limit = ray.max;
        if (!box_hit(nd, ray, inv, limit)) continue;
        if (nd.left < 0) {
            int first = -1 - nd.left;
            for (int k = 0; k < nd.right; k++) {
                uint32_t idx = mesh.bvh.order[first + k];
                tris++;
                Result r = triangle_trace(mesh.tris[idx], ray);
                if (r.hit && (r.t < result.t || (result.hit && r.t == result.t && idx < result.tri_index))) {
                    result = r;
                    result.tri_index = idx;
                }
            }
        } else {
            if (sp + 2 > 128) continue; // cannot happen: depth is bounded by the builder
            stack[sp++] = nd.right;
            stack[sp++] = nd.left;
        }
    }
    if (st) {
        st->box_tests += boxes;
        st->tri_tests += tris;
    }
    return result;
}

// ray.rs:21-37
inline Ray apply_transform(const Ray& r, const M4& transform, const M3& transform_normal_matrix) {
    V4 source = from_vec3(r.source);
    V4 min_point = from_vec3(r.min * r.direction + r.source);
    V4 max_point = from_vec3(r.max * r.direction + r.source);
    V3 s = mul(transform, source);
    V3 mn = mul(transform, min_point);
    V3 mx = mul(transform, max_point);
    V3 direction = mul(transform_normal_matrix, r.direction);
    return {s, direction, length(mn - s), length(mx - s)};
}

struct Ctx {
    const OracleScene* scene;
    const Trig* trig;
    bool accel;
    Stats* st;
};

// sphere.rs:59-101
Result sphere_trace(const Node& n, const Ray& ray, const Trig& m) {
    Result result;
    V3 a = n.center - ray.source;
    float half_second_k = -dot(a, ray.direction);
    float discriminant = 4.0f * (half_second_k * half_second_k - (dot(a, a) - n.radius_sqr));
    if (discriminant < 0.0f) return result;
    float d_sqrt = std::sqrt(discriminant);
    float t1 = -half_second_k + d_sqrt / 2.0f;
    float t2 = -half_second_k - d_sqrt / 2.0f;
    if (t2 >= ray.min && t2 <= ray.max) {
        result.t = t2;
    } else if (t1 >= ray.min && t1 <= ray.max) {
        result.t = t1;
        result.hit_inside = true;
    } else {
        return result;
    }
    result.point = result.t * ray.direction + ray.source;
    float normal_facing_outside = result.hit_inside ? -1.0f : 1.0f;
    result.normal = (result.point - n.center) / (n.radius * normal_facing_outside);
    float u = m.atan2(result.normal.x, result.normal.z) / (2.0f * PI) + 0.5f;
    float v = m.asin(result.normal.y) / PI + 0.5f;
    result.uv = {u, v};
    result.hit = true;
    result.material_id = n.material_id;
    result.node_index = n.index;
    return result;
}

// f32::signum: 1.0 for +0.0 and positives, -1.0 for -0.0 and negatives, NaN for NaN
inline float rust_signum(float x) {
    if (x != x) return x;
    return std::signbit(x) ? -1.0f : 1.0f;
}

// plane.rs:72-103
Result plane_trace(const Node& n, const Ray& ray) {
    Result result;
    float t = dot(n.normal, n.point - ray.source) / dot(n.normal, ray.direction);
    if (t < ray.min || t > ray.max) return result;
    result.hit = true;
    result.point = ray.source + ray.direction * t;
    float normal_facing_dir = -rust_signum(dot(ray.direction, n.normal));
    result.hit_inside = normal_facing_dir < 0.0f;
    result.normal = n.normal * normal_facing_dir;
    V3 uv = result.point - n.point;
    result.uv = {dot(uv, n.bitangent), dot(uv, n.tangent)};
    result.t = t;
    result.material_id = n.material_id;
    result.node_index = n.index;
    return result;
}

Result node_trace(const Ctx& c, const Node& n, const Ray& ray) {
    switch (n.type) {
    case PT_NODE_COLLECTION: { // node_collection.rs:43-55
        Result result;
        result.t = F32_INF;
        for (const Node& child : n.children) {
            Result r = node_trace(c, child, ray);
            if (r.hit && r.t < result.t) result = r;
        }
        return result;
    }
    case PT_NODE_TRANSFORM: { // transform.rs:45-66
        Ray new_ray = apply_transform(ray, n.matrix_inverse, n.inverse_normal_mat);
        Result r = node_trace(c, n.children[0], new_ray);
        if (r.hit) {
            V3 point = mul(n.matrix, from_vec3(r.point));
            float t = length(point - ray.source);
            r.point = point;
            r.normal = mul(n.normal_mat, r.normal);
            r.t = t;
        }
        return r;
    }
    case PT_NODE_KD_TREE: // kd_tree.rs:47-51 is todo!(); defined here as its child
        return node_trace(c, n.children[0], ray);
    case PT_NODE_MESH: { // hierarchy/mesh.rs:61-66
        const Mesh& mesh = c.scene->meshes[n.mesh_id];
        Result r = c.accel ? mesh_trace_bvh(mesh, ray, c.st) : mesh_trace_linear(mesh, ray, c.st);
        r.material_id = n.material_id;
        r.node_index = n.index;
        return r;
    }
    case PT_NODE_SPHERE: return sphere_trace(n, ray, *c.trig);
    case PT_NODE_PLANE: return plane_trace(n, ray);
    }
    return Result();
}

// f32 as u32: saturating, NaN -> 0
inline uint32_t f32_as_u32(float v) {
    if (!(v > 0.0f)) return 0;
    if (v >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)v;
}
inline float rust_clamp(float v, float lo, float hi) { // f32::clamp, NaN stays NaN
    if (v < lo) return lo;
    if (v > hi) return hi;
    return v;
}

// material_input/mod.rs:49-56, texture.rs:53-66, image.rs:39-46
V3 sample_color(const OracleScene& s, const Material& mat, V2 uv, Stats* st) {
    if (!mat.d.color_is_texture) return v3(mat.d.color);
    if (mat.d.uv_mode == PT_UV_CLAMP)
        uv = {rust_clamp(uv.x, 0.0f, 1.0f), rust_clamp(uv.y, 0.0f, 1.0f)};
    else
        uv = {uv.x - std::floor(uv.x), uv.y - std::floor(uv.y)};
    uv.y = 1.0f - uv.y;
    const Image& img = s.images[mat.d.image_id];
    V2 coords = uv * V2{(float)(img.w - 1), (float)(img.h - 1)};
    uint32_t x = f32_as_u32(coords.x), y = f32_as_u32(coords.y);
    if (x >= img.w) x = img.w - 1; // unreachable for finite uv; `image` would panic
    if (y >= img.h) y = img.h - 1;
    const uint8_t* px = &img.rgba[((size_t)y * img.w + x) * 4];
    if (st) st->tex_fetches++;
    return {(float)px[0] / 256.0f, (float)px[1] / 256.0f, (float)px[2] / 256.0f};
}

inline float powi2(float x) { return x * x; }
inline float powi5(float x) { // square-and-multiply, as llvm.powi lowers it
    float x2 = x * x;
    float x4 = x2 * x2;
    return x * x4;
}

// base.rs:51-63
float fresnel_reflection(float theta_cos, float refraction) {
    float refr_sqr = refraction * refraction;
    float c = theta_cos * refraction;
    float g = std::sqrt(1.0f + c * c - refr_sqr);
    float a = (g - c) / (g + c);
    float b_nom = c * (g + c) - refr_sqr;
    float b_den = c * (g - c) + refr_sqr;
    float b = b_nom / b_den;
    return 0.5f * a * a * (1.0f + b * b);
}

struct ColorResult {
    bool is_color; // GetColorResult::Color vs NextRayColorMultiplierAndDirection
    V3 a;          // colour, or multiplier
    V3 dir;
};

// base.rs:87-145
ColorResult base_get_color(const Ctx& c, const Material& mat, V3 dir, const Result& tr, Rng& rng) {
    const PtMaterialDesc& d = mat.d;
    float random_num = rng.gen_range(0.0f, 1.0f);
    if (random_num < d.reflective) {
        return {false, splat(1.0f), reflect(dir, tr.normal)};
    } else if (random_num < d.reflective + d.emissive) {
        return {true, v3(d.emission), V3{}};
    } else if (random_num < d.reflective + d.emissive + d.refractive) {
        float cosv = dot(V3{} - dir, tr.normal);
        float refraction = tr.hit_inside ? d.refraction : 1.0f / d.refraction;
        float fresnel = fresnel_reflection(cosv, refraction);
        float rnd = rng.gen_range(0.0f, 1.0f);
        V3 new_dir;
        if (rnd < fresnel) {
            new_dir = reflect(dir, tr.normal);
        } else if (!refract(dir, tr.normal, refraction, &new_dir)) {
            V3 reflected = reflect(dir, tr.normal);
            new_dir = reflected * (tr.hit_inside ? -1.0f : 1.0f);
        }
        return {false, splat(1.0f), new_dir};
    } else {
        float r0 = rng.gen_range(0.0f, 1.0f);
        float r1 = rng.gen_range(0.0f, 1.0f);
        V3 new_direction = random_on_unit_sphere(*c.trig, r0, r1);
        if (dot(new_direction, tr.normal) < 0.0f) new_direction = V3{} - new_direction;
        V3 color = sample_color(*c.scene, mat, tr.uv, c.st);
        return {false, color, new_direction};
    }
}

// pbr.rs:48-87
float pbr_ndf(const Material& m, float nh) {
    nh = rust_clamp(nh, 0.0001f, 0.9999f);
    float r4 = m.roughness_sqr * m.roughness_sqr;
    float denom_sqrt = nh * nh * (r4 - 1.0f) + 1.0f;
    return r4 / (denom_sqrt * denom_sqrt * PI);
}
float pbr_geometry(const Material& m, float angle_cos) {
    float k = powi2(1.0f + m.d.roughness) / 8.0f;
    return angle_cos / (angle_cos * (1.0f - k) + k);
}
V3 pbr_fresnel(const Material& m, float hi) { return m.f0 + (splat(1.0f) - m.f0) * powi5(1.0f - hi); }
V3 pbr_brdf_diffuse(const Material& m, V3 input_dir, V3 output_dir) {
    V3 h = normalized(output_dir + input_dir);
    V3 specular_k = pbr_fresnel(m, dot(h, input_dir));
    V3 diffuse_k = (splat(1.0f) - specular_k) * (1.0f - m.d.metallic);
    V3 diffuse = INV_PI * v3(m.d.albedo);
    return diffuse_k * diffuse;
}
V3 pbr_brdf_specular(const Material& m, V3 normal, V3 input_dir, V3 output_dir) {
    float ni = dot(normal, input_dir);
    float no = dot(normal, output_dir);
    V3 h = normalized(output_dir + input_dir);
    float geometry = pbr_geometry(m, ni) * pbr_geometry(m, no);
    float ndf = pbr_ndf(m, dot(normal, h));
    V3 specular_k = pbr_fresnel(m, dot(h, input_dir));
    float specular = geometry * ndf / (4.0f * ni * no);
    return specular_k * specular;
}

// pbr.rs:105-136
ColorResult pbr_get_color(const Ctx& c, const Material& mat, V3 dir, const Result& tr, Rng& rng) {
    V3 input_dir = dir * -1.0f;
    float rand_0 = rng.gen_range(0.0f, 1.0f);
    float rand_1 = rng.gen_range(0.0f, 1.0f);
    V3 mulv, output_dir;
    float selection_probability;
    if (rng.gen_bool_half()) {
        output_dir = cosine_weighted_random_on_hemisphere(*c.trig, rand_0, rand_1, tr.normal);
        selection_probability = dot(output_dir, tr.normal);
        mulv = pbr_brdf_diffuse(mat, input_dir, output_dir);
    } else {
        output_dir = random_on_hemisphere(*c.trig, rand_0, rand_1, tr.normal);
        selection_probability = 1.0f;
        mulv = pbr_brdf_specular(mat, tr.normal, input_dir, output_dir);
    }
    mulv = mulv * (dot(output_dir, tr.normal) / selection_probability * PI * 1.0f);
    return {false, mulv, output_dir};
}

ColorResult get_color_result(const Ctx& c, uint32_t material_id, V3 dir, const Result& tr, Rng& rng) {
    const Material& mat = c.scene->materials[material_id];
    return mat.d.type == PT_MATERIAL_PBR ? pbr_get_color(c, mat, dir, tr, rng)
                                         : base_get_color(c, mat, dir, tr, rng);
}

// camera.rs:16-33 and :51-83.  `tan_half_fov` is tanf(fov * 0.5) (camera.rs:56).
Ray camera_get_ray(const PtCamera& cam, const Trig& m, float tan_half_fov, uint64_t px, uint64_t py, Rng& rng) {
    float x_offset = rng.gen_range(-0.5f, 0.5f);
    float y_offset = rng.gen_range(-0.5f, 0.5f);
    float res_x = (float)cam.resolution_x, res_y = (float)cam.resolution_y;
    V2 viewport{cam.focal_length * tan_half_fov * 2.0f, 0.0f};
    viewport.y = viewport.x * (res_y / res_x);
    V3 watch_dot = v3(cam.position);
    watch_dot.x += (((float)px + x_offset) / res_x - 0.5f) * viewport.x;
    watch_dot.y += (((float)py + y_offset) / res_y - 0.5f) * viewport.y;
    watch_dot.z -= cam.focal_length;
    V3 point_on_objective = v3(cam.position);
    V2 sample{0.0f, 0.0f};
    if (cam.bokeh_shape == PT_BOKEH_CIRCLE) {
        float phi = rng.gen_range(0.0f, 2.0f * PI);
        float r = std::sqrt(rng.gen_range(0.0f, 1.0f));
        sample = V2{m.cos(phi), m.sin(phi)} * r;
    } else if (cam.bokeh_shape == PT_BOKEH_SQUARE) {
        float sx = rng.gen_range(-0.5f, 0.5f);
        float sy = rng.gen_range(-0.5f, 0.5f);
        sample = {sx, sy};
    }
    V2 objective_sample = sample * cam.bokeh_size;
    point_on_objective.x += objective_sample.x;
    point_on_objective.y += objective_sample.y;
    M3 rot{v3(cam.rotation.row0), v3(cam.rotation.row1), v3(cam.rotation.row2)};
    V3 direction = mul(rot, watch_dot - point_on_objective);
    float near_plane_dist = length(direction) / cam.focal_length * cam.near_plane;
    return {point_on_objective, normalized(direction), near_plane_dist, F32_MAX};
}

// work_group.rs:26-51
V3 path_get_color(const Ctx& c, Ray ray, uint64_t max_depth, Rng& rng) {
    V3 multiplier = splat(1.0f);
    for (uint64_t d = 0; d < max_depth; d++) {
        if (c.st) c.st->rays++;
        Result tr = node_trace(c, c.scene->root, ray);
        if (!tr.hit) return V3{};
        if (c.st) c.st->hits++;
        ColorResult cr = get_color_result(c, tr.material_id, ray.direction, tr, rng);
        if (cr.is_color) return multiplier * cr.a;
        V3 ray_start = tr.point + cr.dir * EPSILON;
        ray = {ray_start, cr.dir, EPSILON, F32_MAX};
        multiplier = multiplier * cr.a;
    }
    return V3{};
}

bool build_node(const PtSceneDesc* d, uint32_t* cursor, Node* out, int depth, std::string* err) {
    if (*cursor >= d->node_count) {
        *err = "node array ends inside a subtree";
        return false;
    }
    if (depth > 64) {
        *err = "scene graph too deep";
        return false;
    }
    const PtNodeDesc& nd = d->nodes[*cursor];
    out->type = nd.type;
    out->index = *cursor;
    (*cursor)++;
    switch (nd.type) {
    case PT_NODE_COLLECTION:
        out->children.resize(nd.child_count);
        for (uint32_t i = 0; i < nd.child_count; i++)
            if (!build_node(d, cursor, &out->children[i], depth + 1, err)) return false;
        return true;
    case PT_NODE_TRANSFORM: // transform.rs:33-40
        out->matrix = m4(nd.matrix);
        out->matrix_inverse = inverse(out->matrix);
        out->normal_mat = normal_matrix(out->matrix);
        out->inverse_normal_mat = normal_matrix(out->matrix_inverse);
        out->children.resize(1);
        return build_node(d, cursor, &out->children[0], depth + 1, err);
    case PT_NODE_KD_TREE:
        out->children.resize(1);
        return build_node(d, cursor, &out->children[0], depth + 1, err);
    case PT_NODE_MESH:
        if (nd.mesh_id >= d->mesh_count || nd.material_id >= d->material_count) {
            *err = "mesh node references a missing mesh or material";
            return false;
        }
        out->mesh_id = nd.mesh_id;
        out->material_id = nd.material_id;
        return true;
    case PT_NODE_SPHERE: // sphere.rs:42-54
        if (nd.material_id >= d->material_count) {
            *err = "sphere references a missing material";
            return false;
        }
        out->center = v3(nd.center);
        out->radius = nd.radius;
        out->radius_sqr = nd.radius * nd.radius;
        out->material_id = nd.material_id;
        return true;
    case PT_NODE_PLANE: { // plane.rs:41-67
        if (nd.material_id >= d->material_count) {
            *err = "plane references a missing material";
            return false;
        }
        out->point = v3(nd.point);
        out->normal = normalized(v3(nd.normal));
        out->tangent = normalized(v3(nd.tangent));
        out->bitangent = normalized(v3(nd.bitangent));
        bool ok = std::fabs(dot(out->normal, out->tangent)) < 0.001f &&
                  std::fabs(dot(out->normal, out->bitangent)) < 0.001f &&
                  std::fabs(dot(out->tangent, out->bitangent)) < 0.001f;
        if (!ok) {
            *err = "plane axes are not orthogonal";
            return false;
        }
        out->material_id = nd.material_id;
        return true;
    }
    }
    *err = "unknown node type";
    return false;
}

void fill_hit(const Result& r, PtHit* h) {
    h->hit = r.hit ? 1 : 0;
    h->hit_inside = r.hit_inside ? 1 : 0;
    h->node_index = r.hit ? r.node_index : PT_NO_INDEX;
    h->tri_index = r.hit ? r.tri_index : PT_NO_INDEX;
    h->material_id = r.material_id;
    h->t = r.t;
    h->point = pv3(r.point);
    h->normal = pv3(r.normal);
    h->uv = {r.uv.x, r.uv.y};
}

void ensure_bvh(OracleScene* s) {
    for (Mesh& m : s->meshes)
        if (!m.bvh_built) bvh_build(m);
}

template <class F> void parallel_for(size_t n, int threads, F f) {
    if (threads <= 1 || n <= 1) {
        for (size_t i = 0; i < n; i++) f(i, 0);
        return;
    }
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++)
        pool.emplace_back([&, t] {
            for (;;) {
                size_t i = next.fetch_add(1);
                if (i >= n) break;
                f(i, t);
            }
        });
    for (auto& th : pool) th.join();
}

} // namespace

extern "C" {

struct OracleStats {
    uint64_t rays, box_tests, tri_tests, hits, tex_fetches, paths;
    double seconds;
};

const char* oracle_last_error() { return g_error.c_str(); }

int oracle_scene_create(const PtSceneDesc* d, OracleScene** out) {
    if (!d || !out || !d->nodes || d->node_count == 0) {
        g_error = "null or empty scene description";
        return PT_ERR_INVALID_ARGUMENT;
    }
    auto s = std::make_unique<OracleScene>();
    s->meshes.resize(d->mesh_count);
    for (uint32_t i = 0; i < d->mesh_count; i++) { // mesh/mod.rs:42-50
        const PtMeshDesc& m = d->meshes[i];
        s->meshes[i].tris.resize(m.triangle_count);
        for (uint64_t k = 0; k < m.triangle_count; k++) {
            bool valid = m.normals && (!m.normal_valid || m.normal_valid[k]);
            s->meshes[i].tris[k] = triangle_init(m.positions + 9 * k, m.normals ? m.normals + 9 * k : nullptr,
                                                 valid, m.uvs ? m.uvs + 6 * k : nullptr);
        }
    }
    s->images.resize(d->image_count);
    for (uint32_t i = 0; i < d->image_count; i++) {
        s->images[i].w = d->images[i].width;
        s->images[i].h = d->images[i].height;
        s->images[i].rgba.assign(d->images[i].rgba8,
                                 d->images[i].rgba8 + (size_t)d->images[i].width * d->images[i].height * 4);
    }
    s->materials.resize(d->material_count);
    for (uint32_t i = 0; i < d->material_count; i++) {
        Material& m = s->materials[i];
        m.d = d->materials[i];
        if (m.d.type == PT_MATERIAL_BASE && m.d.color_is_texture && m.d.image_id >= d->image_count) {
            g_error = "material references a missing image";
            return PT_ERR_SCENE;
        }
        // pbr.rs:92-96
        m.f0 = v3(m.d.albedo) * m.d.metallic + splat(0.04f) * (1.0f - m.d.metallic);
        m.roughness_sqr = powi2(m.d.roughness);
    }
    uint32_t cursor = 0;
    std::string err;
    if (!build_node(d, &cursor, &s->root, 0, &err)) {
        g_error = err;
        return PT_ERR_SCENE;
    }
    if (cursor != d->node_count) {
        g_error = "node array has entries after the root subtree";
        return PT_ERR_SCENE;
    }
    *out = s.release();
    return PT_OK;
}

void oracle_scene_destroy(OracleScene* s) { delete s; }

// scene.hierarchy.trace_ray for n rays.  accel: 0 linear scan, 1 BVH.  math: 0 libm, 1 portable.
int oracle_trace(OracleScene* s, const PtRay* rays, size_t n, int accel, int math, int threads, PtHit* out,
                 OracleStats* stats) {
    if (!s || !rays || !out) return PT_ERR_INVALID_ARGUMENT;
    if (accel) ensure_bvh(s);
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    std::vector<Stats> st(threads > 0 ? threads : 1);
    auto t0 = std::chrono::steady_clock::now();
    const size_t chunk = 256;
    size_t chunks = (n + chunk - 1) / chunk;
    parallel_for(chunks, threads, [&](size_t ci, int tid) {
        Ctx c{s, math ? &kPortable : &kLibm, accel != 0, &st[tid]};
        size_t hi = std::min(n, (ci + 1) * chunk);
        for (size_t i = ci * chunk; i < hi; i++) {
            Ray r{v3(rays[i].source), v3(rays[i].direction), rays[i].min, rays[i].max};
            st[tid].rays++;
            fill_hit(node_trace(c, s->root, r), &out[i]);
        }
    });
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        for (auto& q : st) {
            stats->rays += q.rays;
            stats->box_tests += q.box_tests;
            stats->tri_tests += q.tri_tests;
        }
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return PT_OK;
}

int oracle_generate_rays(const PtCamera* cam, uint64_t seed, const uint32_t* pixel_xy, const uint64_t* sample_index,
                         size_t n, int math, PtRay* out) {
    if (!cam || !pixel_xy || !sample_index || !out) return PT_ERR_INVALID_ARGUMENT;
    const Trig& m = math ? kPortable : kLibm;
    float tan_half_fov = tanf(cam->fov * 0.5f);
    for (size_t i = 0; i < n; i++) {
        uint64_t x = pixel_xy[2 * i], y = pixel_xy[2 * i + 1];
        Rng rng(seed, stream_id(sample_index[i], x + y * cam->resolution_x));
        Ray r = camera_get_ray(*cam, m, tan_half_fov, x, y, rng);
        out[i] = {pv3(r.source), pv3(r.direction), r.min, r.max};
    }
    return PT_OK;
}

int oracle_shade(OracleScene* s, const PtVec3* directions, const PtHit* hits, const uint64_t* stream, uint64_t seed,
                 size_t n, int math, uint32_t* out_kind, PtVec3* out_a, PtVec3* out_dir) {
    if (!s || !directions || !hits || !stream) return PT_ERR_INVALID_ARGUMENT;
    Ctx c{s, math ? &kPortable : &kLibm, false, nullptr};
    for (size_t i = 0; i < n; i++) {
        if (hits[i].material_id >= s->materials.size()) return PT_ERR_INVALID_ARGUMENT;
        Result tr;
        tr.hit = hits[i].hit != 0;
        tr.hit_inside = hits[i].hit_inside != 0;
        tr.point = v3(hits[i].point);
        tr.normal = v3(hits[i].normal);
        tr.uv = {hits[i].uv.x, hits[i].uv.y};
        tr.t = hits[i].t;
        tr.material_id = hits[i].material_id;
        Rng rng(seed, stream[i]);
        ColorResult cr = get_color_result(c, tr.material_id, v3(directions[i]), tr, rng);
        out_kind[i] = cr.is_color ? 0 : 1;
        out_a[i] = pv3(cr.a);
        out_dir[i] = pv3(cr.dir);
    }
    return PT_OK;
}

// CPURenderer::render (cpu_renderer/mod.rs:190-194): 32x32 tiles (:100-141, with
// the edge tiles clipped as the dead branches at :118,:125 intend), one path per
// pixel per iteration (work_group.rs:53-65), image = sum * (1/iterations)
// (work_group.rs:67-69, image_buffer.rs:26-40), rows flipped (mod.rs:143-175).
int oracle_render(OracleScene* s, const PtRenderTask* task, const PtRenderOpts* opts, int accel, int math,
                  int threads, float* out_rgb, OracleStats* stats) {
    if (!s || !task || !out_rgb) return PT_ERR_INVALID_ARGUMENT;
    PtRenderOpts o;
    memset(&o, 0, sizeof(o));
    if (opts) o = *opts;
    if (o.flags & PT_FLAG_BRUTE_FORCE) accel = 0;
    if (accel) ensure_bvh(s);
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    if (threads <= 0) threads = 1;
    const PtCamera& cam = task->camera;
    const uint64_t W = cam.resolution_x, H = cam.resolution_y;
    if (W == 0 || H == 0 || W * H > 0xFFFFFFFFull) return PT_ERR_INVALID_ARGUMENT;
    uint64_t x0 = 0, y0 = 0, x1 = W, y1 = H;
    if (o.tile_w && o.tile_h) {
        x0 = o.tile_x0;
        y0 = o.tile_y0;
        x1 = std::min<uint64_t>(W, x0 + o.tile_w);
        y1 = std::min<uint64_t>(H, y0 + o.tile_h);
    }
    const Trig& m = math ? kPortable : kLibm;
    float tan_half_fov = tanf(cam.fov * 0.5f);
    std::vector<V3> accum(W * H);
    const uint64_t ts = 32;
    uint64_t tiles_x = (W + ts - 1) / ts, tiles_y = (H + ts - 1) / ts;
    std::vector<Stats> st(threads);
    auto t0 = std::chrono::steady_clock::now();
    parallel_for(tiles_x * tiles_y, threads, [&](size_t ti, int tid) {
        Ctx c{s, &m, accel != 0, &st[tid]};
        uint64_t x_off = (ti % tiles_x) * ts, y_off = (ti / tiles_x) * ts;
        uint64_t w = std::min(ts, W - x_off), h = std::min(ts, H - y_off);
        for (uint64_t it = 0; it < task->iterations; it++)
            for (uint64_t x = 0; x < w; x++)
                for (uint64_t y = 0; y < h; y++) {
                    uint64_t gx = x_off + x, gy = y_off + y;
                    if (gx < x0 || gx >= x1 || gy < y0 || gy >= y1) continue;
                    Rng rng(o.seed, stream_id(o.sample_offset + it, gx + gy * W));
                    Ray ray = camera_get_ray(cam, m, tan_half_fov, gx, gy, rng);
                    st[tid].paths++;
                    V3 color = path_get_color(c, ray, task->trace_depth, rng);
                    V3& px = accum[gx + gy * W];
                    px = px + color;
                }
    });
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    float color_multiplier = 1.0f / (float)task->iterations;
    for (uint64_t y = 0; y < H; y++)
        for (uint64_t x = 0; x < W; x++) {
            V3 p = accum[x + y * W] * color_multiplier;
            uint64_t addr = x + (H - 1 - y) * W;
            out_rgb[addr * 3] = p.x;
            out_rgb[addr * 3 + 1] = p.y;
            out_rgb[addr * 3 + 2] = p.z;
        }
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        for (auto& q : st) {
            stats->rays += q.rays;
            stats->box_tests += q.box_tests;
            stats->tri_tests += q.tri_tests;
            stats->hits += q.hits;
            stats->tex_fetches += q.tex_fetches;
            stats->paths += q.paths;
        }
        stats->seconds = secs;
    }
    return PT_OK;
}

// Raw access to the random stream, for tests of the generator itself.
uint32_t oracle_rng_u32(uint64_t seed, uint64_t stream, uint32_t draw_index) {
    Rng r(seed, stream);
    uint32_t v = 0;
    for (uint32_t i = 0; i <= draw_index; i++) v = r.next_u32();
    return v;
}
float oracle_rng_range(uint64_t seed, uint64_t stream, uint32_t draw_index, float lo, float hi) {
    Rng r(seed, stream);
    for (uint32_t i = 0; i < draw_index; i++) r.next_u32();
    return r.gen_range(lo, hi);
}

// The portable routines and libm side by side, for accuracy tests.
float oracle_portable_math(int fn, float x, float y) {
    switch (fn) {
    case 0: return ptm::sin_f32(x);
    case 1: return ptm::cos_f32(x);
    case 2: return ptm::acos_f32(x);
    case 3: return ptm::asin_f32(x);
    case 4: return ptm::atan2_f32(x, y);
    }
    return 0.0f;
}
void oracle_portable_math_array(int fn, const float* x, const float* y, size_t n, float* out) {
    for (size_t i = 0; i < n; i++) out[i] = oracle_portable_math(fn, x[i], y ? y[i] : 0.0f);
}

// Mat4::inverse / normal_matrix of both, for known-answer tests of transform.rs:33-40.
void oracle_transform_matrices(const PtMat4* m, float* inverse16, float* normal9, float* inverse_normal9) {
    M4 a = m4(*m);
    M4 inv = inverse(a);
    M3 n = normal_matrix(a), ni = normal_matrix(inv);
    const V4* rows[4] = {&inv.row0, &inv.row1, &inv.row2, &inv.row3};
    for (int r = 0; r < 4; r++) {
        inverse16[4 * r] = rows[r]->x;
        inverse16[4 * r + 1] = rows[r]->y;
        inverse16[4 * r + 2] = rows[r]->z;
        inverse16[4 * r + 3] = rows[r]->w;
    }
    const V3* nr[3] = {&n.row0, &n.row1, &n.row2};
    const V3* nir[3] = {&ni.row0, &ni.row1, &ni.row2};
    for (int r = 0; r < 3; r++) {
        normal9[3 * r] = nr[r]->x;
        normal9[3 * r + 1] = nr[r]->y;
        normal9[3 * r + 2] = nr[r]->z;
        inverse_normal9[3 * r] = nir[r]->x;
        inverse_normal9[3 * r + 1] = nir[r]->y;
        inverse_normal9[3 * r + 2] = nir[r]->z;
    }
}

// The client's frame (client/src/frame.rs): RenderSum::add_render :61-71 adds a
// render to the running sum, into_image :73-83 divides by max(count, 1), and
// gamma_correction :86-103 maps each channel with (c / (1 + c) * 255) as u8 —
// Rust's float-to-u8 cast truncates toward zero, saturates at 0 and 255 and sends
// NaN to 0 — with alpha 255.  `sum` holds pixels * 3 floats.
void oracle_frame_add_render(float* sum, const float* render, size_t pixels) {
    for (size_t i = 0; i < pixels * 3; i++) sum[i] = sum[i] + render[i];
}

void oracle_frame_image(const float* sum, size_t pixels, uint64_t count, float* out_rgb, uint8_t* out_rgba) {
    const float divisor = (float)(count > 1 ? count : 1);
    for (size_t i = 0; i < pixels; i++) {
        for (int ch = 0; ch < 3; ch++) {
            float c = sum[3 * i + ch] / divisor;
            if (out_rgb) out_rgb[3 * i + ch] = c;
            if (out_rgba) {
                float v = c / (1.0f + c) * 255.0f;
                uint8_t b;
                if (v != v) b = 0;
                else if (v <= 0.0f) b = 0;
                else if (v >= 255.0f) b = 255;
                else b = (uint8_t)(int)v;
                out_rgba[4 * i + ch] = b;
            }
        }
        if (out_rgba) out_rgba[4 * i + 3] = 255;
    }
}

// PNG scanline reconstruction (PNG specification, "Filtering": types 0 None, 1 Sub, 2 Up,
// 3 Average, 4 Paeth) for oracle/loaders.py, which restates the host loaders that feed the
// path (image.rs:13-19 decodes through the `image` crate).  `filtered` holds `height` rows of
// 1 + stride bytes (filter type, then the row), `bpp` = bytes per complete pixel (at least 1);
// `out` receives height * stride bytes.  Returns 0, or 1 on an unknown filter type.
int oracle_png_unfilter(const uint8_t* filtered, uint32_t height, uint32_t stride, uint32_t bpp, uint8_t* out) {
    for (uint32_t y = 0; y < height; y++) {
        const uint8_t* in = filtered + (size_t)y * (stride + 1);
        const uint8_t type = in[0];
        in++;
        uint8_t* row = out + (size_t)y * stride;
        const uint8_t* up = y ? row - stride : nullptr;
        for (uint32_t i = 0; i < stride; i++) {
            const int a = i >= bpp ? row[i - bpp] : 0;
            const int b = up ? up[i] : 0;
            const int c = (up && i >= bpp) ? up[i - bpp] : 0;
            int pred;
            switch (type) {
            case 0: pred = 0; break;
            case 1: pred = a; break;
            case 2: pred = b; break;
            case 3: pred = (a + b) >> 1; break;
            case 4: {
                const int p = a + b - c;
                const int pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                break;
            }
            default: return 1;
            }
            row[i] = (uint8_t)(in[i] + pred);
        }
    }
    return 0;
}

} // extern "C"

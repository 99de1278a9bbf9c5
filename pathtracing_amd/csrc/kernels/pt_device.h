// Device functions of the path-tracing hot path: random stream, camera ray,
// primitive intersection, BVH traversal, scene-graph walk and material sampling.
// Each function cites the reference code whose arithmetic it reproduces; the
// translation unit must be built with -ffp-contract=off.
#pragma once
#include "../pt_portable_math.h"
#include "pt_scene_device.h"

namespace ptk {

// ------------------------------------------------------------------ random stream
// The reference draws from an unseeded rand 0.8 thread_rng() (camera.rs:21,28,52;
// material/base.rs:94,114,133-134; material/pbr.rs:109-112).  This stream is
// counter based instead: draw c (1-based) of stream s under seed k is the high
// 32 bits of mix64(key + c*G), key = mix64(mix64(k + G) ^ s), mix64 = SplitMix64
// finaliser.  gen_range keeps rand's f32 lattice: u = (draw >> 9) * 2^-23,
// value = u*(hi-lo) + lo.
constexpr uint64_t kGolden = 0x9E3779B97F4A7C15ull;
PT_HD uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
PT_HD uint64_t stream_key(uint64_t seed, uint64_t stream) { return mix64(mix64(seed + kGolden) ^ stream); }
PT_HD uint64_t stream_id(uint64_t sample, uint64_t pixel_linear) { return (sample << 32) | pixel_linear; }

struct Rng {
    uint64_t key;
    uint32_t counter;
    PT_HD uint32_t next_u32() {
        counter++;
        return (uint32_t)(mix64(key + (uint64_t)counter * kGolden) >> 32);
    }
    PT_HD float gen_range(float lo, float hi) {
        float scale = hi - lo;
        float u = (float)(next_u32() >> 9) * 1.1920928955078125e-7f;
        return u * scale + lo;
    }
    PT_HD bool gen_bool_half() { return (next_u32() >> 31) == 0u; }
};

// ------------------------------------------------------------------ camera
// camera.rs:16-33 (BokehShape::sample) and :51-83 (Camera::get_ray)
PT_HD Ray camera_get_ray(const DCamera& cam, uint32_t px, uint32_t py, Rng& rng) {
    float x_offset = rng.gen_range(-0.5f, 0.5f);
    float y_offset = rng.gen_range(-0.5f, 0.5f);
    V3 watch_dot = cam.position;
    watch_dot.x += (((float)px + x_offset) / cam.res_x - 0.5f) * cam.viewport_x;
    watch_dot.y += (((float)py + y_offset) / cam.res_y - 0.5f) * cam.viewport_y;
    watch_dot.z -= cam.focal_length;
    V3 point_on_objective = cam.position;
    V2 sample = mk2(0.0f, 0.0f);
    if (cam.bokeh_shape == 1u) { // circle
        float phi = rng.gen_range(0.0f, 2.0f * kPi);
        float r = sqrt_rn(rng.gen_range(0.0f, 1.0f));
        float sphi, cphi;
        ptm::sincos_f32(phi, &sphi, &cphi);
        sample = mk2(cphi, sphi) * r;
    } else if (cam.bokeh_shape == 2u) { // square
        float sx = rng.gen_range(-0.5f, 0.5f);
        float sy = rng.gen_range(-0.5f, 0.5f);
        sample = mk2(sx, sy);
    }
    V2 objective_sample = sample * cam.bokeh_size;
    point_on_objective.x += objective_sample.x;
    point_on_objective.y += objective_sample.y;
    V3 direction = mul(cam.rotation, watch_dot - point_on_objective);
    float near_plane_dist = length(direction) / cam.focal_length * cam.near_plane;
    Ray r;
    r.source = point_on_objective;
    r.direction = normalized(direction);
    r.min = near_plane_dist;
    r.max = kF32Max;
    return r;
}

// ------------------------------------------------------------------ triangle
// Moller-Trumbore exactly as triangle/mod.rs:130-157.  Returns true and t when
// the reference would report a hit.
PT_HD bool triangle_hit(V3 v0, V3 v1, V3 v2, const Ray& ray, float* t_out) {
    V3 edge0 = v1 - v0;
    V3 edge1 = v2 - v0;
    V3 pvec = cross(ray.direction, edge1);
    float determinant = dot(edge0, pvec);
    if (determinant < kEpsilon && determinant > -kEpsilon) return false;
    float inv_determinant = 1.0f / determinant;
    V3 tvec = ray.source - v0;
    float u = dot(tvec, pvec) * inv_determinant;
    if (!(u >= 0.0f && u <= 1.0f)) return false;
    V3 qvec = cross(tvec, edge0);
    float v = dot(ray.direction, qvec) * inv_determinant;
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = dot(edge1, qvec) * inv_determinant;
    if (t < ray.min || t > ray.max) return false;
    *t_out = t;
    return true;
}

// Point, uv and normal of a triangle hit: triangle/mod.rs:159-164 with
// get_barycentric_coords :91-114, get_uv :116-120, get_normal :122-126.
PT_HD void triangle_finish(const LeafTri& tri, const ShadeTri& sh, const Ray& ray, float t, Hit* h) {
    V3 p0 = ld3(tri.v0), p1 = ld3(tri.v1), p2 = ld3(tri.v2);
    V3 point = ray.source + ray.direction * t;
    V3 tn = ld3(sh.true_normal);
    V3 to0 = p0 - point, to1 = p1 - point, to2 = p2 - point;
    float b0 = abs_f(dot(cross(to1, to2), tn));
    float b1 = abs_f(dot(cross(to0, to2), tn));
    float b2 = abs_f(dot(cross(to0, to1), tn));
    float sum = b0 + b1 + b2;
    float c0 = b0 / sum, c1 = b1 / sum, c2 = b2 / sum;
    h->point = point;
    h->uv = mk2(sh.uv0[0], sh.uv0[1]) * c0 + mk2(sh.uv1[0], sh.uv1[1]) * c1 + mk2(sh.uv2[0], sh.uv2[1]) * c2;
    h->normal = ld3(sh.n0) * c0 + ld3(sh.n1) * c1 + ld3(sh.n2) * c2;
}

// ------------------------------------------------------------------ sphere / plane
// sphere.rs:59-101
// `want_uv` = false leaves uv at (0, 0): the caller knows nothing will read it
// (the sphere's material has no texture), and atan2 + asin are the bulk of the cost.
PT_HD void sphere_trace(const DNode& n, uint32_t node_index, const Ray& ray, Hit* out, bool want_uv = true) {
    out->hit = 0u;
    out->hit_inside = 0u;
    out->t = 0.0f;
    V3 a = n.center - ray.source;
    float half_second_k = -dot(a, ray.direction);
    float discriminant = 4.0f * (half_second_k * half_second_k - (dot(a, a) - n.radius_sqr));
    if (discriminant < 0.0f) return;
    float d_sqrt = sqrt_rn(discriminant);
    float t1 = -half_second_k + d_sqrt / 2.0f;
    float t2 = -half_second_k - d_sqrt / 2.0f;
    float t;
    uint32_t inside = 0u;
    if (t2 >= ray.min && t2 <= ray.max) {
        t = t2;
    } else if (t1 >= ray.min && t1 <= ray.max) {
        t = t1;
        inside = 1u;
    } else {
        return;
    }
    V3 point = t * ray.direction + ray.source;
    float normal_facing_outside = inside ? -1.0f : 1.0f;
    V3 normal = (point - n.center) / (n.radius * normal_facing_outside);
    float u = 0.0f, v = 0.0f;
    if (want_uv) {
        u = ptm::atan2_f32(normal.x, normal.z) / (2.0f * kPi) + 0.5f;
        v = ptm::asin_f32(normal.y) / kPi + 0.5f;
    }
    out->hit = 1u;
    out->hit_inside = inside;
    out->t = t;
    out->point = point;
    out->normal = normal;
    out->uv = mk2(u, v);
    out->material_id = n.material_id;
    out->node_index = node_index;
    out->tri_index = 0xFFFFFFFFu;
}

// plane.rs:72-103
PT_HD void plane_trace(const DNode& n, uint32_t node_index, const Ray& ray, Hit* out) {
    out->hit = 0u;
    out->hit_inside = 0u;
    out->t = 0.0f;
    float t = dot(n.normal, n.point - ray.source) / dot(n.normal, ray.direction);
    if (t < ray.min || t > ray.max) return;
    V3 point = ray.source + ray.direction * t;
    float normal_facing_dir = -signum_f(dot(ray.direction, n.normal));
    V3 uv = point - n.point;
    out->hit = 1u;
    out->hit_inside = normal_facing_dir < 0.0f ? 1u : 0u;
    out->t = t;
    out->point = point;
    out->normal = n.normal * normal_facing_dir;
    out->uv = mk2(dot(uv, n.bitangent), dot(uv, n.tangent));
    out->material_id = n.material_id;
    out->node_index = node_index;
    out->tri_index = 0xFFFFFFFFu;
}

#if defined(__HIPCC__)
// ------------------------------------------------------------------ BVH traversal
// The search must return what the linear scan of mesh/mod.rs:57-62 returns: the
// smallest t, the lowest triangle index among equal t.  Boxes are padded at
// build time and the slab test is widened, so a subtree is skipped only when no
// triangle inside it can report a hit at t <= the best so far.

constexpr int kLdsStack = 16;   // traversal-stack entries per lane kept in LDS
constexpr int kSpillStack = 112; // further entries in the lane's global spill column

// Per-lane spill column in global memory (the kernels run a fixed, persistent
// grid, so the area is sized once).  Dword k of lane `tid` lives at
// base[k * stride + tid]: lanes of a wave touching the same k coalesce.
// Layout of a column: [0, kSpillStack) traversal-stack overflow, then one
// 23-dword frame per scene-graph nesting level (saved ray, saved running best).
constexpr int kFrameDwords = 23;
constexpr int kSpillDwords = kSpillStack + kMaxGraphDepth * kFrameDwords;

// Both columns carry their address space in the pointer type, so pushes and pops
// are LDS and global instructions (a generic pointer would make them flat ones).
typedef __attribute__((address_space(3))) int32_t* LdsI32Ptr;
typedef __attribute__((address_space(1))) uint32_t* GlobalU32Ptr;
struct TravStack {
    LdsI32Ptr lds; // this lane's column: lds[k * stride]
    int stride;
    GlobalU32Ptr spill; // this lane's column: spill[k * spill_stride]
    uint32_t spill_stride;
    int sp;
    __device__ __forceinline__ void push(int32_t v) {
        if (sp < kLdsStack) lds[sp * stride] = v;
        else if (sp < kLdsStack + kSpillStack) spill[(size_t)(sp - kLdsStack) * spill_stride] = (uint32_t)v;
        sp++;
    }
    __device__ __forceinline__ int32_t pop() {
        if (sp == 0) return kBvhSentinel;
        sp--;
        return sp < kLdsStack ? lds[sp * stride] : (int32_t)spill[(size_t)(sp - kLdsStack) * spill_stride];
    }
    __device__ __forceinline__ GlobalU32Ptr frame(int depth) const {
        return spill + (size_t)(kSpillStack + depth * kFrameDwords) * spill_stride;
    }
};

__device__ __forceinline__ void spill_store_ray(const TravStack& st, int depth, const Ray& r) {
    GlobalU32Ptr f = st.frame(depth);
    const size_t s = st.spill_stride;
    f[0 * s] = __float_as_uint(r.source.x);
    f[1 * s] = __float_as_uint(r.source.y);
    f[2 * s] = __float_as_uint(r.source.z);
    f[3 * s] = __float_as_uint(r.direction.x);
    f[4 * s] = __float_as_uint(r.direction.y);
    f[5 * s] = __float_as_uint(r.direction.z);
    f[6 * s] = __float_as_uint(r.min);
    f[7 * s] = __float_as_uint(r.max);
}
__device__ __forceinline__ Ray spill_load_ray(const TravStack& st, int depth) {
    GlobalU32Ptr f = st.frame(depth);
    const size_t s = st.spill_stride;
    Ray r;
    r.source = mk3(__uint_as_float(f[0 * s]), __uint_as_float(f[1 * s]), __uint_as_float(f[2 * s]));
    r.direction = mk3(__uint_as_float(f[3 * s]), __uint_as_float(f[4 * s]), __uint_as_float(f[5 * s]));
    r.min = __uint_as_float(f[6 * s]);
    r.max = __uint_as_float(f[7 * s]);
    return r;
}
__device__ __forceinline__ void spill_store_hit(const TravStack& st, int depth, const Hit& h) {
    GlobalU32Ptr f = st.frame(depth) + (size_t)8 * st.spill_stride;
    const size_t s = st.spill_stride;
    f[0 * s] = __float_as_uint(h.t);
    f[1 * s] = h.hit | (h.hit_inside << 1);
    f[2 * s] = h.node_index;
    f[3 * s] = h.tri_index;
    f[4 * s] = h.material_id;
    f[5 * s] = __float_as_uint(h.point.x);
    f[6 * s] = __float_as_uint(h.point.y);
    f[7 * s] = __float_as_uint(h.point.z);
    f[8 * s] = __float_as_uint(h.normal.x);
    f[9 * s] = __float_as_uint(h.normal.y);
    f[10 * s] = __float_as_uint(h.normal.z);
    f[11 * s] = __float_as_uint(h.uv.x);
    f[12 * s] = __float_as_uint(h.uv.y);
}
__device__ __forceinline__ Hit spill_load_hit(const TravStack& st, int depth) {
    GlobalU32Ptr f = st.frame(depth) + (size_t)8 * st.spill_stride;
    const size_t s = st.spill_stride;
    Hit h;
    h.t = __uint_as_float(f[0 * s]);
    uint32_t flags = f[1 * s];
    h.hit = flags & 1u;
    h.hit_inside = (flags >> 1) & 1u;
    h.node_index = f[2 * s];
    h.tri_index = f[3 * s];
    h.material_id = f[4 * s];
    h.point = mk3(__uint_as_float(f[5 * s]), __uint_as_float(f[6 * s]), __uint_as_float(f[7 * s]));
    h.normal = mk3(__uint_as_float(f[8 * s]), __uint_as_float(f[9 * s]), __uint_as_float(f[10 * s]));
    h.uv = mk2(__uint_as_float(f[11 * s]), __uint_as_float(f[12 * s]));
    return h;
}

// Conservative slab test of a padded box against the ray interval [0, limit].
// The computed interval is widened by 4e-7 relative (the slab arithmetic rounds
// with less than 2e-7).  A NaN term arises only as 0 * inf, i.e. for a ray lying
// exactly in one of the box's boundary planes; min/max drop it and the other
// plane's +-inf then reports a miss, which is right: boxes are padded by more
// than the distance at which the f32 triangle test can still accept a ray, so
// such a ray cannot be accepted by any triangle inside.
__device__ __forceinline__ bool box_test(const float* bmin, const float* bmax, V3 o, V3 inv, float limit, float* entry) {
    float ax = (bmin[0] - o.x) * inv.x, bx = (bmax[0] - o.x) * inv.x;
    float ay = (bmin[1] - o.y) * inv.y, by = (bmax[1] - o.y) * inv.y;
    float az = (bmin[2] - o.z) * inv.z, bz = (bmax[2] - o.z) * inv.z;
    float lo = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
    float hi = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), limit));
    lo = lo * 0.9999996f;
    hi = hi * 1.0000004f;
    *entry = lo;
    return lo <= hi;
}

// Scene arrays are reached through pointers stored in device memory, which the
// compiler must take for generic addresses (flat loads); they always point into
// global memory, so say so and get global loads.
typedef float F4V __attribute__((ext_vector_type(4)));
struct GlobalF4Ptr {
    const __attribute__((address_space(1))) F4V* p;
    __device__ __forceinline__ float4 operator[](int i) const {
        F4V v = p[i];
        return make_float4(v.x, v.y, v.z, v.w);
    }
};
__device__ __forceinline__ GlobalF4Ptr global_f4(const void* p) {
    GlobalF4Ptr g;
    g.p = (const __attribute__((address_space(1))) F4V*)(unsigned long long)p;
    return g;
}

// A load that writes LDS directly (no register holds the data): every lane names its own 16 (4) bytes of global
// memory, and the wave's 64 pieces land next to each other from `wave_dst`, lane 0's first.  The load is counted by
// vmcnt like any other; the wave that issued it may read the LDS copy after waiting for it.
__device__ __forceinline__ void lds_dma16(const void* src, void* wave_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(unsigned long long)src,
                                     (__attribute__((address_space(3))) void*)wave_dst, 16, 0, 0);
}
__device__ __forceinline__ void lds_dma4(const void* src, void* wave_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(unsigned long long)src,
                                     (__attribute__((address_space(3))) void*)wave_dst, 4, 0, 0);
}

__device__ __forceinline__ LeafTri load_tri(const LeafTri* p) {
    GlobalF4Ptr q = global_f4(p);
    float4 a = q[0], b = q[1], c = q[2];
    LeafTri t;
    t.v0[0] = a.x; t.v0[1] = a.y; t.v0[2] = a.z; t.orig_index = __float_as_uint(a.w);
    t.v1[0] = b.x; t.v1[1] = b.y; t.v1[2] = b.z; t.pad1 = 0;
    t.v2[0] = c.x; t.v2[1] = c.y; t.v2[2] = c.z; t.pad2 = 0;
    return t;
}

__device__ __forceinline__ bool all_finite(V3 v) {
    const uint32_t e = 0x7F800000u;
    return ((__float_as_uint(v.x) & e) != e) && ((__float_as_uint(v.y) & e) != e) && ((__float_as_uint(v.z) & e) != e);
}

// State of one lane's search through a mesh's BVH.  The search must return what
// the linear scan of mesh/mod.rs:57-62 returns: the smallest t, the lowest
// triangle index among equal t.  It is cut into steps so that a wave can stop,
// hand finished lanes new rays and go on (see search_kernel).
struct TravState {
    const WideNode* nodes;
    const LeafTri* tris;
    V3 inv;
    float ray_limit; // ray.max, or +inf when it is NaN (a NaN max never rejects, triangle/mod.rs:155)
    int32_t cur;     // node to visit next: >= 0 inner, < 0 leaf, kBvhSentinel finished
    float tbest;
    uint32_t best_slot, best_orig, found;
};

// False: the ray cannot hit any triangle of the mesh.
__device__ __forceinline__ bool trav_reaches(const DMesh& mesh, const Ray& ray) {
    // A ray with a NaN or infinite origin or direction component cannot hit any
    // triangle: in triangle_hit the non-finite value reaches u = (tvec.pvec)/det
    // as an infinity or a NaN and fails 0 <= u <= 1 (the linear scan would test
    // every triangle and reject every one).  Such rays do occur: the reference's
    // cosine-weighted sampler takes sqrt(cos(f32 pi/2)) = sqrt(-4.4e-8) when its
    // random number is exactly 0.
    if (!all_finite(ray.source) || !all_finite(ray.direction)) return false;
    V3 inv = mk3(1.0f / ray.direction.x, 1.0f / ray.direction.y, 1.0f / ray.direction.z);
    float limit = (ray.max == ray.max) ? ray.max : f32_inf();
    float entry;
    if (!box_test(mesh.bmin, mesh.bmax, ray.source, inv, limit, &entry)) return false;
    return mesh.root_ref != kBvhSentinel;
}

// 1 / direction for the box tests of a search, kept finite: on the node grids the
// slab distances are sums of products with it (trav_inner), and an infinite factor
// would turn them into NaNs, i.e. into "this axis does not constrain the ray" -- a
// ray running exactly along a coordinate plane (a handful per million: mirror
// bounces off axis-aligned walls, sample directions that are exactly 0) would then
// open every box in a whole slab of the tree.  With +-1e28 in place of +-inf such a
// ray is off by at least 1e8 on that axis unless it starts within 1e-20 of the
// slab, which is as good as the exact answer and errs on the side of opening.
__device__ __forceinline__ V3 search_inverse_direction(V3 d) {
    const float big = 1.0e28f;
    return mk3(clamp_f(1.0f / d.x, -big, big), clamp_f(1.0f / d.y, -big, big), clamp_f(1.0f / d.z, -big, big));
}

// Starts a search.  False: the ray cannot hit the mesh at all; `ts` then
// describes a finished search without a hit.
__device__ __forceinline__ bool trav_begin(const DMesh& mesh, const Ray& ray, TravStack& st, TravState& ts) {
    ts.nodes = mesh.nodes;
    ts.tris = mesh.tris;
    ts.cur = kBvhSentinel; // until the checks below pass: a finished search without a hit
    ts.tbest = f32_inf();
    ts.best_slot = 0u;
    ts.best_orig = 0xFFFFFFFFu;
    ts.found = 0u;
    st.sp = 0;
    if (!trav_reaches(mesh, ray)) return false;
    ts.inv = search_inverse_direction(ray.direction);
    ts.ray_limit = (ray.max == ray.max) ? ray.max : f32_inf();
    ts.cur = mesh.root_ref;
    return true;
}

// Byte `c` of a packed dword as a float (one v_cvt_f32_ubyteN).
template <int C> __device__ __forceinline__ float byte_to_float(uint32_t packed) {
    return (float)((packed >> (8 * C)) & 0xFFu);
}

// Entry distance of child C of a node into `*t`, or +inf when the ray misses the
// child's box within [0, limit] (or the slot is empty).  near_q / far_q hold the
// quantised plane coordinates the ray meets first / last on each axis; a = grid
// step * 1/direction, bn / bf = (grid origin - ray origin) * 1/direction moved
// down / up by the rounding slack (see trav_inner).  A NaN term (0 * inf: a ray
// lying in a plane of the grid) is dropped by max3 / min3, which only makes the
// test more permissive.
template <int C>
__device__ __forceinline__ float child_entry(const uint32_t* near_q, const uint32_t* far_q, V3 a, V3 bn, V3 bf,
                                             float limit, int32_t ref) {
    float nx = __builtin_fmaf(byte_to_float<C>(near_q[0]), a.x, bn.x), fx = __builtin_fmaf(byte_to_float<C>(far_q[0]), a.x, bf.x);
    float ny = __builtin_fmaf(byte_to_float<C>(near_q[1]), a.y, bn.y), fy = __builtin_fmaf(byte_to_float<C>(far_q[1]), a.y, bf.y);
    float nz = __builtin_fmaf(byte_to_float<C>(near_q[2]), a.z, bn.z), fz = __builtin_fmaf(byte_to_float<C>(far_q[2]), a.z, bf.z);
    float lo = fmaxf(fmaxf(nx, ny), fmaxf(nz, 0.0f));
    float hi = fminf(fminf(fx, fy), fminf(fz, limit));
    return (lo <= hi && ref != kBvhSentinel) ? lo : f32_inf();
}

// Orders two (distance, reference) pairs by distance.
__device__ __forceinline__ void order_pair(float& ta, int32_t& ra, float& tb, int32_t& rb) {
    bool swap = tb < ta;
    float t0 = swap ? tb : ta, t1 = swap ? ta : tb;
    int32_t r0 = swap ? rb : ra, r1 = swap ? ra : rb;
    ta = t0;
    tb = t1;
    ra = r0;
    rb = r1;
}

// One inner node: test its (up to four) children, descend into the nearest, defer
// the others nearest on top.
//
// The slab distances are evaluated on the node's grid: plane coordinate q (0..255)
// of axis k lies at t = q * A + B with A = step_k / d_k, B = (origin_k - o_k) / d_k.
// In f32 that is fl(q * A' + B') with A' = fl(step * inv) (exact: step is a power
// of two), B' = fl(fl(origin - o) * inv), inv = fl(1 / d): against the exact value
// the result is off by at most eps * (2 * 255 |A| + 4 |B|), eps = 2^-24.  Entry
// distances are lowered and exit distances raised by twice that, so a box is
// skipped only when the ray misses it outright; the triangle boxes below carry the
// padding that makes "misses the box" imply "rejected by the f32 triangle test".
// (Fused multiply-adds are used here on purpose: nothing of a hit record depends on
// these values, only which boxes are opened.)
__device__ __forceinline__ void trav_inner(TravState& ts, const Ray& ray, TravStack& st) {
    GlobalF4Ptr q = global_f4(ts.nodes + ts.cur);
    float4 n0 = q[0], n1 = q[1], n2 = q[2], n3 = q[3];
    V3 step = mk3(n0.w, n2.z, n2.w);
    V3 a = mk3(step.x * ts.inv.x, step.y * ts.inv.y, step.z * ts.inv.z);
    V3 b = mk3((n0.x - ray.source.x) * ts.inv.x, (n0.y - ray.source.y) * ts.inv.y, (n0.z - ray.source.z) * ts.inv.z);
    const float k = 1.1920929e-7f; // 2 eps
    V3 slack = mk3(__builtin_fmaf(abs_f(a.x), 512.0f * k, abs_f(b.x) * (4.0f * k)),
                   __builtin_fmaf(abs_f(a.y), 512.0f * k, abs_f(b.y) * (4.0f * k)),
                   __builtin_fmaf(abs_f(a.z), 512.0f * k, abs_f(b.z) * (4.0f * k)));
    V3 bn = mk3(b.x - slack.x, b.y - slack.y, b.z - slack.z);
    V3 bf = mk3(b.x + slack.x, b.y + slack.y, b.z + slack.z);
    const uint32_t qlo[3] = {__float_as_uint(n1.x), __float_as_uint(n1.y), __float_as_uint(n1.z)};
    const uint32_t qhi[3] = {__float_as_uint(n1.w), __float_as_uint(n2.x), __float_as_uint(n2.y)};
    // a ray running towards smaller coordinates meets the high plane first
    const bool back[3] = {ts.inv.x < 0.0f, ts.inv.y < 0.0f, ts.inv.z < 0.0f};
    uint32_t near_q[3], far_q[3];
    for (int k3 = 0; k3 < 3; k3++) {
        near_q[k3] = back[k3] ? qhi[k3] : qlo[k3];
        far_q[k3] = back[k3] ? qlo[k3] : qhi[k3];
    }
    const float limit = fminf(ts.tbest, ts.ray_limit);
    int32_t r0 = __float_as_int(n3.x), r1 = __float_as_int(n3.y), r2 = __float_as_int(n3.z), r3 = __float_as_int(n3.w);
    float t0 = child_entry<0>(near_q, far_q, a, bn, bf, limit, r0);
    float t1 = child_entry<1>(near_q, far_q, a, bn, bf, limit, r1);
    float t2 = child_entry<2>(near_q, far_q, a, bn, bf, limit, r2);
    float t3 = child_entry<3>(near_q, far_q, a, bn, bf, limit, r3);
    // sort the four by entry distance (misses, at +inf, end up last); finding only the nearest and deferring
    // the others in slot order saves two comparators and opens 2 % more nodes: no faster
    order_pair(t0, r0, t1, r1);
    order_pair(t2, r2, t3, r3);
    order_pair(t0, r0, t2, r2);
    order_pair(t1, r1, t3, r3);
    order_pair(t1, r1, t2, r2);
    const float inf = f32_inf();
    if (t0 == inf) {
        ts.cur = st.pop();
        return;
    }
    if (t3 != inf) st.push(r3);
    if (t2 != inf) st.push(r2);
    if (t1 != inf) st.push(r1);
    ts.cur = r0;
}

// One leaf: the reference's triangle test on each of its triangles.
__device__ __forceinline__ void trav_leaf(TravState& ts, const Ray& ray, TravStack& st) {
    uint32_t ref = (uint32_t)(~ts.cur);
    uint32_t first = ref >> 3, count = (ref & 7u) + 1u;
    for (uint32_t k = 0; k < count; k++) {
        LeafTri tri = load_tri(ts.tris + first + k);
        float t;
        if (triangle_hit(ld3(tri.v0), ld3(tri.v1), ld3(tri.v2), ray, &t)) {
            if (t < ts.tbest || (ts.found && t == ts.tbest && tri.orig_index < ts.best_orig)) {
                ts.tbest = t;
                ts.best_orig = tri.orig_index;
                ts.best_slot = first + k;
                ts.found = 1u;
            }
        }
    }
    ts.cur = st.pop();
}

__device__ __forceinline__ bool trav_is_inner(const TravState& ts) { return ts.cur >= 0 && ts.cur != kBvhSentinel; }
__device__ __forceinline__ bool trav_done(const TravState& ts) { return ts.cur == kBvhSentinel; }

// mesh/mod.rs:53-65 literally: every triangle, strict <, first wins.
__device__ __forceinline__ bool mesh_closest_linear(const DMesh& mesh, const Ray& ray, float* best_t, uint32_t* slot) {
    float tbest = f32_inf();
    uint32_t best = 0u;
    bool found = false;
    for (uint32_t i = 0; i < mesh.tri_count; i++) {
        LeafTri tri = load_tri(mesh.tris_orig + i);
        float t;
        if (triangle_hit(ld3(tri.v0), ld3(tri.v1), ld3(tri.v2), ray, &t) && t < tbest) {
            tbest = t;
            best = i;
            found = true;
        }
    }
    *best_t = tbest;
    *slot = best;
    return found;
}

// The RayTraceResult of a mesh node (hierarchy/mesh.rs:61-66) once the closest
// triangle is known: `tri` is the winning triangle's record, `orig` its index.
__device__ __forceinline__ void mesh_result(const DMesh& mesh, const DNode& n, uint32_t node_index, const Ray& ray,
                                            bool found, float t, const LeafTri* tri, uint32_t orig, Hit* out) {
    out->hit = found ? 1u : 0u;
    out->hit_inside = 0u;
    out->t = found ? t : f32_inf(); // the reference returns its running t = +inf on a miss
    out->material_id = n.material_id;
    out->node_index = node_index;
    out->tri_index = found ? orig : 0xFFFFFFFFu;
    if (found) {
        LeafTri tr = load_tri(tri);
        GlobalF4Ptr q = global_f4(mesh.shade + orig); // 80 bytes = five rows
        float4 s0 = q[0], s1 = q[1], s2 = q[2], s3 = q[3], s4 = q[4];
        ShadeTri sh;
        sh.true_normal[0] = s0.x; sh.true_normal[1] = s0.y; sh.true_normal[2] = s0.z;
        sh.n0[0] = s0.w; sh.n0[1] = s1.x; sh.n0[2] = s1.y;
        sh.n1[0] = s1.z; sh.n1[1] = s1.w; sh.n1[2] = s2.x;
        sh.n2[0] = s2.y; sh.n2[1] = s2.z; sh.n2[2] = s2.w;
        sh.uv0[0] = s3.x; sh.uv0[1] = s3.y; sh.uv1[0] = s3.z; sh.uv1[1] = s3.w;
        sh.uv2[0] = s4.x; sh.uv2[1] = s4.y;
        triangle_finish(tr, sh, ray, t, out);
    }
}

// ------------------------------------------------------------------ scene graph
// Evaluates scene.hierarchy.trace_ray(ray) over the pre-order node array without
// recursion.  Semantics per node type:
//   collection (node_collection.rs:43-55): running best starts at t = +inf, a
//     child replaces it when child.hit && child.t < best.t;
//   transform (transform.rs:45-66): child sees apply_transform(ray, M^-1); a hit
//     comes back as point = M*point', t = |point - ray.source|, normal = N*normal';
//   kd_tree: its child.
// The walk is resumable: at a mesh node (unless `brute`) it returns kWalkMesh with
// w.ray = the ray in the mesh's space; the caller supplies the closest triangle
// (from a search it runs, or one that ran earlier), stores the node's result in
// `cur` with walk_mesh_result and calls walk() again with resume = true.  Which
// node comes next never depends on the ray, so all lanes of a wave stop at the
// same nodes.  The saved state of the open internal
// nodes (outer ray of a transform, running best of an enclosing collection)
// lives in the lane's spill column, indexed by nesting depth.
struct WalkState {
    Ray ray;       // the ray in the space of node i
    Hit best;      // running best of the innermost open collection
    uint32_t i;    // node to evaluate next
    uint32_t open; // innermost open collection / transform, or 0xFFFFFFFF
    int depth;     // number of open collections / transforms
};
constexpr int kWalkDone = 0, kWalkMesh = 1;

__device__ __forceinline__ void walk_begin(WalkState& w, const Ray& ray) {
    w.ray = ray;
    w.best.hit = 0u;
    w.best.hit_inside = 0u;
    w.best.t = f32_inf();
    w.best.material_id = 0u;
    w.best.node_index = 0xFFFFFFFFu;
    w.best.tri_index = 0xFFFFFFFFu;
    w.best.point = mk3(0, 0, 0);
    w.best.normal = mk3(0, 0, 0);
    w.best.uv = mk2(0, 0);
    w.i = 0u;
    w.open = 0xFFFFFFFFu;
    w.depth = 0;
}

// Node records are read through the constant address space with a wave-uniform
// index, so they arrive by scalar loads into SGPRs (one fetch per wave, no vector
// registers): every lane of a wave is always at the same node, because which node
// comes next never depends on the ray.
typedef const __attribute__((address_space(4))) DNode* ConstNodePtr;
__device__ __forceinline__ ConstNodePtr const_nodes(const DScene& sc) {
    return (ConstNodePtr)(unsigned long long)(sc.nodes);
}
// The description of mesh `id`, the same for every lane (a mesh node's, a search target's): read with scalar
// loads like the node array, so a wave does not wait a trip to memory through the vector path for a
// pointer it shares.  Only the fields the caller goes on to use are loaded.
__device__ __forceinline__ DMesh uniform_mesh(const DScene& sc, uint32_t id) {
    typedef const __attribute__((address_space(4))) DMesh* ConstMeshPtr;
    ConstMeshPtr m = (ConstMeshPtr)(unsigned long long)(sc.meshes) + id;
    DMesh d;
    d.nodes = m->nodes;
    d.tris = m->tris;
    d.tris_orig = m->tris_orig;
    d.shade = m->shade;
    d.tri_count = m->tri_count;
    d.node_count = m->node_count;
    d.root_ref = m->root_ref;
    d.pad = 0u;
    for (int k = 0; k < 3; k++) {
        d.bmin[k] = m->bmin[k];
        d.bmax[k] = m->bmax[k];
    }
    return d;
}
__device__ __forceinline__ V3 cv3(const __attribute__((address_space(4))) V3& v) { return mk3(v.x, v.y, v.z); }

// &Mat4 * Vec4::from_vec3(p) (lib.rs:519-526, :651-659) with the matrix in scalar registers
__device__ __forceinline__ V3 mul_point_c(const __attribute__((address_space(4))) M4& a, V3 p) {
    float w = p.x * a.m[3][0] + p.y * a.m[3][1] + p.z * a.m[3][2] + 1.0f * a.m[3][3];
    V3 r = mk3(p.x * a.m[0][0] + p.y * a.m[0][1] + p.z * a.m[0][2] + 1.0f * a.m[0][3],
               p.x * a.m[1][0] + p.y * a.m[1][1] + p.z * a.m[1][2] + 1.0f * a.m[1][3],
               p.x * a.m[2][0] + p.y * a.m[2][1] + p.z * a.m[2][2] + 1.0f * a.m[2][3]);
    // x / 1.0 is x for every x, so the three divisions are skipped when w is exactly 1 (affine
    // matrices and finite points; a non-finite point makes w NaN and takes the division)
    if (w != 1.0f) r = r / w;
    return r;
}
__device__ __forceinline__ V3 mul_c(const __attribute__((address_space(4))) M3& a, V3 rhs) {
    return mk3(dot(rhs, cv3(a.row0)), dot(rhs, cv3(a.row1)), dot(rhs, cv3(a.row2)));
}

// Ray::apply_transform (ray.rs:21-37) for a transform node read through scalar
// loads, with normal_matrix() of the inverse precomputed.
__device__ __forceinline__ Ray transform_ray_c(const __attribute__((address_space(4))) DNode& node, const Ray& r) {
    V3 min_point = r.min * r.direction + r.source;
    V3 max_point = r.max * r.direction + r.source;
    V3 s = mul_point_c(node.matrix_inverse, r.source);
    V3 mn = mul_point_c(node.matrix_inverse, min_point);
    V3 mx = mul_point_c(node.matrix_inverse, max_point);
    Ray o;
    o.source = s;
    o.direction = mul_c(node.inverse_normal_matrix, r.direction);
    o.min = length(mn - s);
    o.max = length(mx - s);
    return o;
}

// ---- instance run (DRun) -----------------------------------------------------
// Cursor over the children of the run whose world-space box the ray pierces.
struct RunCursor {
    V3 inv;
    float limit;
    uint32_t i; // next node of the box tree to look at
};
__device__ __forceinline__ void run_begin(const DRun& run, const Ray& world, bool active, RunCursor& c) {
    c.inv = mk3(1.0f / world.direction.x, 1.0f / world.direction.y, 1.0f / world.direction.z);
    // The whole half-line: a child tests t against the ray's range in its own space, where a
    // scaled child measures min and max as distances but t along a direction that is not of unit
    // length (ray.rs:21-37), so hits beyond the world-space max can be accepted.
    c.limit = f32_inf();
    // rays with a non-finite component hit no triangle (trav_reaches)
    c.i = (active && all_finite(world.source) && all_finite(world.direction)) ? 0u : run.tlas_count;
}
// The next such child (index in child order, any order of visit), or kRunNone.
// Ends: `skip` of a node is larger than the node's own index.
__device__ __forceinline__ uint32_t run_next(const DRun& run, const Ray& world, RunCursor& c) {
    GlobalF4Ptr nodes = global_f4(run.tlas);
    while (c.i < run.tlas_count) {
        float4 a = nodes[2 * (int)c.i], b = nodes[2 * (int)c.i + 1];
        float bmin[3] = {a.x, a.y, a.z}, bmax[3] = {b.x, b.y, b.z};
        const uint32_t skip = __float_as_uint(a.w), child = __float_as_uint(b.w);
        float entry;
        const bool pierced = box_test(bmin, bmax, world.source, c.inv, c.limit, &entry);
        if (pierced && child != kTlasInner) {
            c.i = skip;
            return child;
        }
        c.i = pierced ? c.i + 1u : skip;
    }
    return kRunNone;
}

// One node of the box tree: the child it holds if the ray pierces it and it is a leaf, else kRunNone;
// the cursor moves on either way (callers loop while run_live).
__device__ __forceinline__ bool run_live(const DRun& run, const RunCursor& c) { return c.i < run.tlas_count; }
__device__ __forceinline__ uint32_t run_step(const DRun& run, const Ray& world, RunCursor& c) {
    GlobalF4Ptr nodes = global_f4(run.tlas);
    float4 a = nodes[2 * (int)c.i], b = nodes[2 * (int)c.i + 1];
    float bmin[3] = {a.x, a.y, a.z}, bmax[3] = {b.x, b.y, b.z};
    const uint32_t skip = __float_as_uint(a.w), child = __float_as_uint(b.w);
    float entry;
    const bool pierced = box_test(bmin, bmax, world.source, c.inv, c.limit, &entry);
    const bool leaf = child != kTlasInner;
    c.i = (pierced && !leaf) ? c.i + 1u : skip;
    return (pierced && leaf) ? child : kRunNone;
}

// A child's table entry: ids, and the ray of `world` in the child's space
// (Ray::apply_transform, ray.rs:21-37, on the packed copy of the transform node).
struct RunChildIds {
    uint32_t transform_node, mesh_node, mesh_id, material_id;
};
__device__ __forceinline__ void unpack_rows(GlobalF4Ptr rows, int first, M4* m, M3* n) {
    for (int r = 0; r < 4; r++) {
        float4 v = rows[first + r];
        m->m[r][0] = v.x;
        m->m[r][1] = v.y;
        m->m[r][2] = v.z;
        m->m[r][3] = v.w;
    }
    float4 a = rows[first + 4], b = rows[first + 5], c = rows[first + 6];
    n->row0 = mk3(a.x, a.y, a.z);
    n->row1 = mk3(b.x, b.y, b.z);
    n->row2 = mk3(c.x, c.y, c.z);
}
__device__ __forceinline__ RunChildIds run_child_ids(const DRun& run, uint32_t k) {
    float4 v = global_f4(run.children + k)[0];
    RunChildIds ids;
    ids.transform_node = __float_as_uint(v.x);
    ids.mesh_node = __float_as_uint(v.y);
    ids.mesh_id = __float_as_uint(v.z);
    ids.material_id = __float_as_uint(v.w);
    return ids;
}
__device__ __forceinline__ Ray run_child_ray(const DRun& run, uint32_t k, const RunChildIds& ids, const Ray& r) {
    // the rows are asked for whether or not the child has a transform (every record holds them): the answer to
    // that is in the record too, and waiting for it first would be a second trip to memory
    M4 inv;
    M3 ninv;
    unpack_rows(global_f4(run.children + k), 1, &inv, &ninv);
    V3 min_point = r.min * r.direction + r.source;
    V3 max_point = r.max * r.direction + r.source;
    V3 s = mul_point(inv, r.source);
    V3 mn = mul_point(inv, min_point);
    V3 mx = mul_point(inv, max_point);
    Ray o;
    o.source = s;
    o.direction = mul(ninv, r.direction);
    o.min = length(mn - s);
    o.max = length(mx - s);
    return ids.transform_node == kRunNoTransform ? r : o;
}

// What the run's child k returns for `world` (transform.rs:45-66 around
// hierarchy/mesh.rs:61-66), given the closest triangle of its mesh for the ray in
// the mesh's space.
__device__ __forceinline__ void run_child_hit(const DScene& sc, uint32_t k, const Ray& world, float t, uint32_t slot,
                                              uint32_t orig, Hit* h) {
    const RunChildIds ids = run_child_ids(sc.run, k);
    Ray local = run_child_ray(sc.run, k, ids, world);
    const DMesh& mesh = sc.meshes[ids.mesh_id];
    DNode nd;
    nd.material_id = ids.material_id;
    nd.mesh_id = ids.mesh_id;
    mesh_result(mesh, nd, ids.mesh_node, local, true, t, mesh.tris + slot, orig, h);
    if (h->hit && ids.transform_node != kRunNoTransform) {
        M4 fwd;
        M3 nfwd;
        unpack_rows(global_f4(sc.run.children + k), 8, &fwd, &nfwd);
        V3 point = mul_point(fwd, h->point);
        h->t = length(point - world.source);
        h->point = point;
        h->normal = mul(nfwd, h->normal);
    }
}

// The distance at which the run's child k reports the triangle hit at `t` in its mesh's space: the t of
// run_child_hit's record (transform.rs:57-58 on the point of triangle_finish) without the rest of it.
__device__ __forceinline__ float run_child_distance(const DScene& sc, uint32_t k, const Ray& world, float t) {
    // ids and rows in one trip (run_child_ray)
    const RunChildIds ids = run_child_ids(sc.run, k);
    M4 inv, fwd;
    M3 ninv, nfwd;
    unpack_rows(global_f4(sc.run.children + k), 1, &inv, &ninv);
    unpack_rows(global_f4(sc.run.children + k), 8, &fwd, &nfwd);
    V3 source = mul_point(inv, world.source);
    V3 direction = mul(ninv, world.direction);
    V3 point = mul_point(fwd, source + direction * t);
    return ids.transform_node == kRunNoTransform ? t : length(point - world.source);
}

// node_collection.rs:48-52 over the children of the run: `h` of child k replaces the
// running best when it is closer, or as close and from an earlier child (the
// sequential loop keeps the first of equals; the candidates arrive in any order).
__device__ __forceinline__ void run_take(Hit& best, uint32_t& best_k, const Hit& h, uint32_t k) {
    if (h.hit && (h.t < best.t || (best.hit && h.t == best.t && k < best_k))) {
        best = h;
        best_k = k;
    }
}

// Where a lane finds the searches that were run ahead for its ray within the run.
struct RunLookup {
    uint32_t head;         // the ray's last job (index within the region), kRunNone, or kRunInPlace
    const float4* job_c;   // the region's jobs: (.., .., flag | child, previous job of the ray)
    const float4* run_res; // the region's results per job: (t, found, leaf slot, triangle index)
};

__device__ __forceinline__ void search_in_place(const DMesh& mesh, const Ray& ray, TravStack& st, TravState& ts);

// The run's result for `world`: min over its children as the reference's loop
// would find it.  With `look` the searches were run ahead; without (or when the
// job list was full) every candidate child is searched here.
__device__ __forceinline__ void walk_run(const DScene& sc, const Ray& world, const RunLookup* look, TravStack& st, Hit* out) {
    Hit best;
    best.hit = 0u;
    best.hit_inside = 0u;
    best.t = f32_inf();
    best.material_id = 0u;
    best.node_index = 0xFFFFFFFFu;
    best.tri_index = 0xFFFFFFFFu;
    best.point = mk3(0, 0, 0);
    best.normal = mk3(0, 0, 0);
    best.uv = mk2(0, 0);
    uint32_t best_k = 0xFFFFFFFFu;
    if (look != nullptr && look->head != kRunInPlace) {
        // which child wins needs only each candidate's distance; the winner's record is built once
        GlobalF4Ptr jobs = global_f4(look->job_c), results = global_f4(look->run_res);
        uint32_t j = look->head;
        float win_t = f32_inf(), win_local_t = 0.0f;
        uint32_t win_slot = 0u, win_orig = 0u;
        while (j != kRunNone) {
            const float4 jc = jobs[(int)j], res = results[(int)j];
            if (__float_as_uint(res.y) != 0u) {
                const uint32_t k = __float_as_uint(jc.z) & ~kRunJobFlag;
                const float t = run_child_distance(sc, k, world, res.x);
                if (t < win_t || (best_k != 0xFFFFFFFFu && t == win_t && k < best_k)) {
                    win_t = t;
                    best_k = k;
                    win_local_t = res.x;
                    win_slot = __float_as_uint(res.z);
                    win_orig = __float_as_uint(res.w);
                }
            }
            j = __float_as_uint(jc.w);
        }
        if (best_k != 0xFFFFFFFFu) run_child_hit(sc, best_k, world, win_local_t, win_slot, win_orig, &best);
    } else {
        RunCursor c;
        run_begin(sc.run, world, true, c);
        for (uint32_t k = run_next(sc.run, world, c); k != kRunNone; k = run_next(sc.run, world, c)) {
            const RunChildIds ids = run_child_ids(sc.run, k);
            Ray local = run_child_ray(sc.run, k, ids, world);
            TravState ts;
            search_in_place(sc.meshes[ids.mesh_id], local, st, ts);
            if (ts.found != 0u) {
                Hit h;
                run_child_hit(sc, k, world, ts.tbest, ts.best_slot, ts.best_orig, &h);
                run_take(best, best_k, h, k);
            }
        }
    }
    *out = best;
}

// `all_uv` = false: texture coordinates are computed only where the node's material samples a texture.
// `live_meshes` (wave-uniform): bit m clear = no lane of the wave has a triangle in the scene's m-th
// mesh node; a transform whose whole subtree is that mesh node is then passed over as the miss it is
// for every lane (instances the wave's rays do not touch cost nothing).
// kRun: the scene's instance run (if any) is taken as a whole; `run`: the searches run ahead for
// this lane's ray inside it (null: none were).  Without kRun the run's children are walked like
// any other node (kernels for scenes without a run carry none of its code).
template <bool kRun = true>
__device__ __forceinline__ int walk(const DScene& sc, WalkState& w, Hit& cur, bool resume, bool brute, TravStack& st,
                                    bool all_uv = true, uint32_t live_meshes = 0xFFFFFFFFu, const RunLookup* run = nullptr) {
    const uint32_t n = sc.node_count;
    ConstNodePtr nodes = const_nodes(sc);
    // wave-uniform cursor (see above)
    uint32_t i = __builtin_amdgcn_readfirstlane(w.i);
    uint32_t open = __builtin_amdgcn_readfirstlane(w.open);
    int depth = (int)__builtin_amdgcn_readfirstlane((uint32_t)w.depth);
    bool open_is_collection = open != 0xFFFFFFFFu && nodes[open].type == 0u;
    if (resume) {
        i++;
        if (open_is_collection && cur.hit && cur.t < w.best.t) w.best = cur;
    }
    int status = kWalkDone;
    for (;;) {
        while (open != 0xFFFFFFFFu && i == nodes[open].end) {
            depth--;
            if (nodes[open].type == 0u) {
                cur = w.best;
                if (depth > 0) w.best = spill_load_hit(st, depth);
            } else {
                w.ray = spill_load_ray(st, depth);
                if (cur.hit) {
                    V3 point = mul_point_c(nodes[open].matrix, cur.point);
                    cur.t = length(point - w.ray.source);
                    cur.point = point;
                    cur.normal = mul_c(nodes[open].normal_matrix, cur.normal);
                }
            }
            open = nodes[open].parent;
            open_is_collection = open != 0xFFFFFFFFu && nodes[open].type == 0u;
            if (open_is_collection && cur.hit && cur.t < w.best.t) w.best = cur;
        }
        if (i >= n) break;
        if (kRun && !brute && sc.run.child_count != 0u && i == sc.run.first_node) {
            // the instance run: all its children at once (the linear-scan mode visits them one by one below)
            walk_run(sc, w.ray, run, st, &cur);
            i = sc.run.end_node;
            if (open_is_collection && cur.hit && cur.t < w.best.t) w.best = cur;
            continue;
        }
        uint32_t type = nodes[i].type;
        if (type == 0u) { // collection
            if (depth > 0) spill_store_hit(st, depth, w.best);
            depth++;
            open = i;
            open_is_collection = true;
            w.best.hit = 0u;
            w.best.hit_inside = 0u;
            w.best.t = f32_inf();
            i++;
            continue;
        }
        if (type == 1u) { // transform: ray.rs:21-37 with normal_matrix() of the inverse precomputed
            const uint32_t only = nodes[i].only_mesh;
            if (only < 32u && ((live_meshes >> only) & 1u) == 0u) {
                cur.hit = 0u;
                cur.hit_inside = 0u;
                cur.t = f32_inf();
                i = nodes[i].end;
                continue;
            }
            spill_store_ray(st, depth, w.ray);
            depth++;
            open = i;
            open_is_collection = false;
            w.ray = transform_ray_c(nodes[i], w.ray);
            i++;
            continue;
        }
        if (type == 5u) { // kd_tree
            i++;
            continue;
        }
        if (type == 2u) {
            DNode nd;
            nd.material_id = nodes[i].material_id;
            nd.mesh_id = nodes[i].mesh_id;
            const DMesh& mesh = sc.meshes[nd.mesh_id];
            if (brute) {
                float t;
                uint32_t slot;
                bool found = mesh_closest_linear(mesh, w.ray, &t, &slot);
                mesh_result(mesh, nd, i, w.ray, found, t, mesh.tris_orig + slot, slot, &cur);
            } else {
                // every lane stops here together, whether or not its ray can reach the mesh
                status = kWalkMesh;
                break;
            }
        } else if (type == 3u) {
            DNode nd;
            nd.center = cv3(nodes[i].center);
            nd.radius = nodes[i].radius;
            nd.radius_sqr = nodes[i].radius_sqr;
            nd.material_id = nodes[i].material_id;
            typedef const __attribute__((address_space(4))) DMaterial* ConstMaterialPtr;
            bool want_uv = all_uv || ((ConstMaterialPtr)(unsigned long long)sc.materials)[nd.material_id].color_is_texture != 0u;
            sphere_trace(nd, i, w.ray, &cur, want_uv);
        } else {
            DNode nd;
            nd.point = cv3(nodes[i].point);
            nd.normal = cv3(nodes[i].normal);
            nd.tangent = cv3(nodes[i].tangent);
            nd.bitangent = cv3(nodes[i].bitangent);
            nd.material_id = nodes[i].material_id;
            plane_trace(nd, i, w.ray, &cur);
        }
        i++;
        if (open_is_collection && cur.hit && cur.t < w.best.t) w.best = cur;
    }
    w.i = i;
    w.open = open;
    w.depth = depth;
    return status;
}

// scene.hierarchy.trace_ray for a scene whose root collection holds only leaves and transforms
// over one mesh (DScene::flat_root): one wave-uniform loop over the children, the ray and the
// running best in registers, nothing saved or resumed.  Arithmetic and comparisons are those of
// walk(): node_collection.rs:43-55 around sphere.rs / plane.rs / transform.rs + hierarchy/mesh.rs.
// `res[m][index]` = the closest triangle of the scene's m-th mesh node for this ray where bit m of
// `listed` is set (m below `deferred`; further mesh nodes are searched here); `live_meshes`, `run` as in walk().
// `all_uv` as in walk().
template <bool kRun>
__device__ __forceinline__ void walk_flat(const DScene& sc, const Ray& world, float4* const* res, size_t index,
                                          uint32_t deferred, uint32_t listed, uint32_t live_meshes, const RunLookup* run,
                                          TravStack& st, Hit* out, bool all_uv = false, const float4* first_res = nullptr) {
    ConstNodePtr nodes = const_nodes(sc);
    Hit best;
    best.hit = 0u;
    best.hit_inside = 0u;
    best.t = f32_inf();
    best.material_id = 0u;
    best.node_index = 0xFFFFFFFFu;
    best.tri_index = 0xFFFFFFFFu;
    best.point = mk3(0, 0, 0);
    best.normal = mk3(0, 0, 0);
    best.uv = mk2(0, 0);
    const uint32_t n = sc.node_count;
    for (uint32_t i = 1; i < n;) {
        Hit cur;
        cur.hit = 0u;
        cur.hit_inside = 0u;
        cur.t = f32_inf();
        if (kRun && sc.run.child_count != 0u && i == sc.run.first_node) {
            walk_run(sc, world, run, st, &cur);
            i = sc.run.end_node;
        } else {
            const uint32_t type = nodes[i].type;
            const uint32_t end = nodes[i].end;
            if (type == 3u) {
                DNode nd;
                nd.center = cv3(nodes[i].center);
                nd.radius = nodes[i].radius;
                nd.radius_sqr = nodes[i].radius_sqr;
                nd.material_id = nodes[i].material_id;
                typedef const __attribute__((address_space(4))) DMaterial* ConstMaterialPtr;
                bool want_uv = all_uv || ((ConstMaterialPtr)(unsigned long long)sc.materials)[nd.material_id].color_is_texture != 0u;
                sphere_trace(nd, i, world, &cur, want_uv);
            } else if (type == 4u) {
                DNode nd;
                nd.point = cv3(nodes[i].point);
                nd.normal = cv3(nodes[i].normal);
                nd.tangent = cv3(nodes[i].tangent);
                nd.bitangent = cv3(nodes[i].bitangent);
                nd.material_id = nodes[i].material_id;
                plane_trace(nd, i, world, &cur);
            } else {
                // (transform ->) (kd_tree ->) mesh: the mesh node is the last node of the subtree
                const uint32_t j = end - 1u;
                const uint32_t m = nodes[j].mesh_ordinal;
                if (m >= 32u || ((live_meshes >> m) & 1u) != 0u) {
                    const bool transformed = type == 1u;
                    Ray local = transformed ? transform_ray_c(nodes[i], world) : world;
                    DNode nd;
                    nd.material_id = nodes[j].material_id;
                    nd.mesh_id = nodes[j].mesh_id;
                    const DMesh mesh = uniform_mesh(sc, nd.mesh_id);
                    TravState ts;
                    if (m < deferred) {
                        float4 r = make_float4(f32_inf(), __uint_as_float(0u), __uint_as_float(0u), __uint_as_float(0xFFFFFFFFu));
                        if ((listed >> m) & 1u) r = (first_res != nullptr && m == 0u) ? *first_res : res[m][index];
                        ts.tbest = r.x;
                        ts.found = __float_as_uint(r.y);
                        ts.best_slot = __float_as_uint(r.z);
                        ts.best_orig = __float_as_uint(r.w);
                    } else {
                        search_in_place(mesh, local, st, ts);
                    }
                    mesh_result(mesh, nd, j, local, ts.found != 0u, ts.tbest, mesh.tris + ts.best_slot, ts.best_orig, &cur);
                    if (transformed && cur.hit) {
                        V3 point = mul_point_c(nodes[i].matrix, cur.point);
                        cur.t = length(point - world.source);
                        cur.point = point;
                        cur.normal = mul_c(nodes[i].normal_matrix, cur.normal);
                    }
                }
            }
            i = end;
        }
        if (cur.hit && cur.t < best.t) best = cur;
    }
    *out = best;
}

// The mesh of the node the walk stopped at.
__device__ __forceinline__ const DMesh& walk_mesh(const DScene& sc, const WalkState& w) {
    return sc.meshes[const_nodes(sc)[__builtin_amdgcn_readfirstlane(w.i)].mesh_id];
}

// A whole search, begun and run to its end by this lane.
// (A search of a valid tree opens every node at most once; the step budget only
// makes sure that a lane leaves these loops whatever the node array holds.)
__device__ __forceinline__ void search_in_place(const DMesh& mesh, const Ray& ray, TravStack& st, TravState& ts) {
    if (!trav_begin(mesh, ray, st, ts)) return;
    uint32_t budget = 8u * mesh.node_count + 64u;
    while (!trav_done(ts) && budget != 0u) {
        while (trav_is_inner(ts) && budget != 0u) {
            trav_inner(ts, ray, st);
            budget--;
        }
        if (!trav_done(ts) && !trav_is_inner(ts)) {
            trav_leaf(ts, ray, st);
            budget--;
        }
    }
}

// The result of the mesh node the walk stopped at, from a finished search.
__device__ __forceinline__ void walk_mesh_result(const DScene& sc, const WalkState& w, const TravState& ts, Hit* cur) {
    ConstNodePtr nodes = const_nodes(sc);
    uint32_t i = __builtin_amdgcn_readfirstlane(w.i);
    DNode nd;
    nd.material_id = nodes[i].material_id;
    nd.mesh_id = nodes[i].mesh_id;
    const DMesh& mesh = sc.meshes[nd.mesh_id];
    mesh_result(mesh, nd, i, w.ray, ts.found != 0u, ts.tbest, mesh.tris + ts.best_slot, ts.best_orig, cur);
}

// scene.hierarchy.trace_ray for one ray, every search run to its end in place.
__device__ __forceinline__ void scene_trace(const DScene& sc, Ray ray, bool brute, TravStack& st, Hit* result) {
    WalkState w;
    TravState ts;
    Hit cur;
    cur.hit = 0u;
    cur.hit_inside = 0u;
    cur.t = 0.0f;
    cur.material_id = 0u;
    cur.node_index = 0xFFFFFFFFu;
    cur.tri_index = 0xFFFFFFFFu;
    cur.point = mk3(0, 0, 0);
    cur.normal = mk3(0, 0, 0);
    cur.uv = mk2(0, 0);
    walk_begin(w, ray);
    bool resume = false;
    while (walk(sc, w, cur, resume, brute, st) == kWalkMesh) {
        search_in_place(walk_mesh(sc, w), w.ray, st, ts);
        walk_mesh_result(sc, w, ts, &cur);
        resume = true;
    }
    *result = cur;
}

// ------------------------------------------------------------------ materials
// material_input/mod.rs:49-56, texture.rs:53-66, image.rs:39-46
// In two steps, so that a caller can start the fetch of the texel, do other work, and take the colour later.
__device__ __forceinline__ uint32_t sample_color_fetch(const DMaterial& m, V2 uv) {
    if (!m.color_is_texture) return 0u;
    if (m.uv_mode == 0u) uv = mk2(clamp_f(uv.x, 0.0f, 1.0f), clamp_f(uv.y, 0.0f, 1.0f));
    else uv = mk2(uv.x - floorf(uv.x), uv.y - floorf(uv.y));
    uv.y = 1.0f - uv.y;
    V2 coords = uv * mk2((float)(m.tex_w - 1u), (float)(m.tex_h - 1u));
    uint32_t x = f32_as_u32(coords.x), y = f32_as_u32(coords.y);
    if (x >= m.tex_w) x = m.tex_w - 1u;
    if (y >= m.tex_h) y = m.tex_h - 1u;
    return ((const __attribute__((address_space(1))) uint32_t*)(unsigned long long)m.texels)[(size_t)y * m.tex_w + x];
}
__device__ __forceinline__ V3 sample_color_finish(const DMaterial& m, uint32_t texel) {
    if (!m.color_is_texture) return m.color;
    return mk3((float)(texel & 0xFFu) / 256.0f, (float)((texel >> 8) & 0xFFu) / 256.0f,
               (float)((texel >> 16) & 0xFFu) / 256.0f);
}
__device__ __forceinline__ V3 sample_color(const DMaterial& m, V2 uv) { return sample_color_finish(m, sample_color_fetch(m, uv)); }

// math/src/lib.rs:272-280
__device__ __forceinline__ V3 random_on_unit_sphere(float rand_0, float rand_1) {
    float theta = rand_0 * kPi * 2.0f;
    float phi = ptm::acos_f32((2.0f * rand_1) - 1.0f);
    float sp, cp, st, ct;
    ptm::sincos_f32(phi, &sp, &cp);
    ptm::sincos_f32(theta, &st, &ct);
    float x = sp * ct;
    float y = sp * st;
    return mk3(x, y, cp);
}

// base.rs:51-63
__device__ __forceinline__ float fresnel_reflection(float theta_cos, float refraction) {
    float refr_sqr = refraction * refraction;
    float c = theta_cos * refraction;
    float g = sqrt_rn(1.0f + c * c - refr_sqr);
    float a = (g - c) / (g + c);
    float b_nom = c * (g + c) - refr_sqr;
    float b_den = c * (g - c) + refr_sqr;
    float b = b_nom / b_den;
    return 0.5f * a * a * (1.0f + b * b);
}

__device__ __forceinline__ float powi5(float x) {
    float x2 = x * x;
    float x4 = x2 * x2;
    return x * x4;
}

// Material::get_color.  Returns true for GetColorResult::Color (value in *a),
// false for NextRayColorMultiplierAndDirection (*a multiplier, *dir direction).
//
// Lanes of a wave meet different materials and branches, and the expensive part
// of three of the branches is the same function of two random numbers (a point
// on the unit sphere: base.rs:126-128 and pbr.rs:119 via lib.rs:272-289).  The
// function therefore runs in three steps: draw and classify (cheap, divergent),
// one shared evaluation of the sphere point for every lane that needs it, then
// the branch-specific rest.  Each lane still executes exactly the arithmetic of
// its own branch of the reference, in the reference's order of random draws.
__device__ __forceinline__ bool material_get_color(const DMaterial& m, V3 dir, const Hit& tr, Rng& rng, V3* a,
                                                   V3* out_dir) {
    enum { kReflect, kEmit, kRefract, kDiffuse, kPbrCosine, kPbrUniform };
    int kind;
    float u0 = 0.0f, u1 = 0.0f;
    if (m.type == 0u) { // base.rs:87-145
        float random_num = rng.gen_range(0.0f, 1.0f);
        if (random_num < m.reflective) kind = kReflect;
        else if (random_num < m.reflective + m.emissive) kind = kEmit;
        else if (random_num < m.reflective + m.emissive + m.refractive) kind = kRefract;
        else {
            kind = kDiffuse;
            u0 = rng.gen_range(0.0f, 1.0f);
            u1 = rng.gen_range(0.0f, 1.0f);
        }
    } else { // pbr.rs:105-110
        u0 = rng.gen_range(0.0f, 1.0f);
        u1 = rng.gen_range(0.0f, 1.0f);
        kind = rng.gen_bool_half() ? kPbrCosine : kPbrUniform;
    }

    // the texel of a diffuse bounce is on its way while the sphere point is worked out
    uint32_t texel = 0u;
    if (kind == kDiffuse) texel = sample_color_fetch(m, tr.uv);
    V3 sphere_point = mk3(0, 0, 0);
    if (kind == kDiffuse || kind == kPbrUniform) sphere_point = random_on_unit_sphere(u0, u1);

    if (kind == kReflect) {
        *a = splat3(1.0f);
        *out_dir = reflect(dir, tr.normal);
        return false;
    }
    if (kind == kEmit) {
        *a = m.emission;
        return true;
    }
    if (kind == kRefract) {
        float cosv = dot(mk3(0, 0, 0) - dir, tr.normal);
        float refraction = tr.hit_inside ? m.refraction : 1.0f / m.refraction;
        float fresnel = fresnel_reflection(cosv, refraction);
        float rnd = rng.gen_range(0.0f, 1.0f);
        V3 new_dir;
        if (rnd < fresnel) {
            new_dir = reflect(dir, tr.normal);
        } else {
            // lib.rs:263-270
            float d = dot(tr.normal, dir);
            float k = 1.0f - refraction * refraction * (1.0f - d * d);
            if (k < 0.0f) {
                V3 reflected = reflect(dir, tr.normal);
                new_dir = reflected * (tr.hit_inside ? -1.0f : 1.0f);
            } else {
                new_dir = refraction * dir - tr.normal * (refraction * d + sqrt_rn(k));
            }
        }
        *a = splat3(1.0f);
        *out_dir = new_dir;
        return false;
    }
    if (kind == kDiffuse) {
        V3 new_direction = sphere_point;
        if (dot(new_direction, tr.normal) < 0.0f) new_direction = mk3(0, 0, 0) - new_direction;
        *a = sample_color_finish(m, texel);
        *out_dir = new_direction;
        return false;
    }
    // pbr.rs:111-136
    V3 input_dir = dir * -1.0f;
    V3 normal = tr.normal;
    V3 output_dir;
    float selection_probability;
    if (kind == kPbrCosine) {
        // lib.rs:196-212, :291-300
        float r0 = ptm::cos_f32(kPi * 0.5f * (1.0f - u0));
        float theta = u1 * kPi * 2.0f;
        V3 j = normal;
        V3 i = (abs_f(j.y) + abs_f(j.z) < kEpsilon) ? mk3(-j.y, j.x, 0.0f) : mk3(0.0f, j.z, -j.y);
        V3 k = cross(j, i);
        float r = sqrt_rn(r0);
        V3 rnd = j * (1.0f - r0 + kEpsilon);
        float st, ct;
        ptm::sincos_f32(theta, &st, &ct);
        output_dir = rnd + r * (i * ct + k * st);
        selection_probability = dot(output_dir, normal);
    } else {
        // lib.rs:282-289
        V3 rnd = sphere_point;
        output_dir = (dot(rnd, normal) < 0.0f) ? rnd * -1.0f : rnd;
        selection_probability = 1.0f;
    }
    // fresnel, pbr.rs:60-62, evaluated by both brdf_diffuse (:64-73) and brdf_specular (:75-87)
    V3 h = normalized(output_dir + input_dir);
    V3 specular_k = m.f0 + (splat3(1.0f) - m.f0) * powi5(1.0f - dot(h, input_dir));
    V3 mulv;
    if (kind == kPbrCosine) {
        V3 diffuse_k = (splat3(1.0f) - specular_k) * (1.0f - m.metallic);
        V3 diffuse = kInvPi * m.albedo;
        mulv = diffuse_k * diffuse;
    } else {
        // brdf_specular with ndf :48-53, geometry :55-58
        float ni = dot(normal, input_dir);
        float no = dot(normal, output_dir);
        float kk = ((1.0f + m.roughness) * (1.0f + m.roughness)) / 8.0f;
        float g_i = ni / (ni * (1.0f - kk) + kk);
        float g_o = no / (no * (1.0f - kk) + kk);
        float geometry = g_i * g_o;
        float nh = clamp_f(dot(normal, h), 0.0001f, 0.9999f);
        float r4 = m.roughness_sqr * m.roughness_sqr;
        float denom_sqrt = nh * nh * (r4 - 1.0f) + 1.0f;
        float ndf = r4 / (denom_sqrt * denom_sqrt * kPi);
        float specular = geometry * ndf / (4.0f * ni * no);
        mulv = specular_k * specular;
    }
    mulv = mulv * (dot(output_dir, normal) / selection_probability * kPi * 1.0f);
    *a = mulv;
    *out_dir = output_dir;
    return false;
}
#endif // __HIPCC__

} // namespace ptk

// gfx950 kernels of the wavefront path tracer.
//
// One batch = some consecutive samples of every pixel of a rectangle.  Per batch:
//   segment_kernel x D   one path segment per live path (the first launch also
//                        generates the camera rays): scene trace + material
//                        sampling; survivors go to the wave's region of the other
//                        queue, finished paths store their colour into the
//                        contribution buffer
//   accumulate_kernel    per pixel, adds the batch's contributions in sample order
// (work_group.rs:26-69 is the per-pixel loop this replaces).
#include <hip/hip_runtime.h>

#include "pt_device.h"
#include "pt_kernels.h"

#include <algorithm>

namespace ptk {

constexpr int kBlock = 256;

__device__ __forceinline__ void path_slot_to_pixel(const BatchGeometry& g, uint32_t q, uint32_t* x, uint32_t* y,
                                                   bool* valid) {
    uint32_t block = q >> 6, within = q & 63u;
    uint32_t bx = block % g.nbx, by = block / g.nbx;
    uint32_t lx = bx * 8u + (within & 7u), ly = by * 8u + (within >> 3);
    *valid = lx < g.w && ly < g.h;
    *x = g.x0 + lx;
    *y = g.y0 + ly;
}

struct RenderParams {
    DScene scene;
    DCamera cam;
    BatchGeometry geo;
    uint64_t seed;
    uint64_t sample_first;
    uint32_t sample_count;
    uint32_t trace_depth;
    uint32_t brute;
};

__device__ __forceinline__ TravStack make_stack(int32_t* lds_stack, uint32_t* spill, uint32_t spill_threads) {
    TravStack st;
    st.lds = &lds_stack[threadIdx.x];
    st.stride = kBlock;
    st.spill = spill + (blockIdx.x * kBlock + threadIdx.x);
    st.spill_stride = spill_threads;
    st.sp = 0;
    return st;
}

#ifndef PT_SEGMENT_OCCUPANCY
#define PT_SEGMENT_OCCUPANCY 4 // waves per SIMD the segment kernel is compiled for = resident 256-thread blocks per CU
#endif
#ifndef PT_REFILL
#define PT_REFILL 16
#endif
uint32_t segment_blocks_per_cu() { return PT_SEGMENT_OCCUPANCY; }
constexpr uint32_t kRefill = PT_REFILL; // go back for new rays once this many lanes of a wave have finished their search

// One path segment (work_group.rs:29-47) for every live path of the batch.
//
// Persistent grid; every wave owns a fixed region of the ray queues (region w =
// records [w * cap, w * cap + count_w)), reads its rays from the input region and
// appends surviving paths to its output region, so queues need no atomics and no
// global compaction pass.  On the first segment (FIRST) there is no input queue:
// the wave generates camera rays for path slots taken in 64-slot chunks w, w + W,
// w + 2W, ... (W = waves in the grid), which spreads every wave's share over the
// whole frame and keeps the regions' loads balanced in later segments.
//
// Inside a wave, lanes do not wait for the slowest ray: each lane runs a small
// state machine (idle -> walking the scene graph -> searching a mesh BVH ->
// walking -> shading -> idle) and the wave refills idle lanes from its region
// whenever kRefill searches have finished.  BVH search lengths vary by an order
// of magnitude between neighbouring rays, so this is what keeps lanes busy.
template <bool FIRST>
__global__ void __launch_bounds__(kBlock, PT_SEGMENT_OCCUPANCY)
segment_kernel(RenderParams p, uint32_t depth, uint32_t cap, const float4* __restrict__ ia, const float4* __restrict__ ib,
               const float4* __restrict__ ic, float4* oa, float4* ob, float4* oc, const uint32_t* __restrict__ counts_in,
               uint32_t* counts_out, float* contrib, unsigned long long* ray_total, uint32_t* spill,
               uint32_t spill_threads) {
    __shared__ int32_t lds_stack[kLdsStack * kBlock];
    const uint32_t tid = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;
    const uint32_t waves = (gridDim.x * kBlock) >> 6;
    const uint64_t total = (uint64_t)p.sample_count * p.geo.npp;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    const size_t region = (size_t)wave * cap;

    uint32_t nin; // records this wave will take
    if (FIRST) {
        uint32_t chunks_total = (uint32_t)((total + 63u) >> 6);
        nin = wave < chunks_total ? ((chunks_total - wave + waves - 1u) / waves) * 64u : 0u;
    } else {
        nin = counts_in[wave];
    }
    uint32_t next = 0, out_pos = 0, traced = 0;

    TravStack st = make_stack(lds_stack, spill, spill_threads);
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
    ts.cur = kBvhSentinel;
    uint32_t phase = 0; // 0 idle, 1 searching a mesh, 2 walking the scene graph
    bool fresh = false;
    uint32_t path = 0, rng_counter = 0;
    V3 mult = mk3(0, 0, 0);
    const bool brute = p.brute != 0u;

    for (;;) {
        // ---- hand new rays to idle lanes
        unsigned long long idle_mask = __ballot(phase == 0u);
        uint32_t n_idle = (uint32_t)__popcll(idle_mask);
        uint32_t avail = nin - next;
        if (n_idle != 0u && avail != 0u) {
            uint32_t rank = (uint32_t)__popcll(idle_mask & lane_lt);
            if (phase == 0u && rank < avail) {
                uint32_t v = next + rank;
                Ray ray;
                bool have = false;
                if (FIRST) {
                    uint64_t slot = ((uint64_t)(v >> 6) * waves + wave) * 64u + (v & 63u);
                    if (slot < total) {
                        uint32_t s_local = (uint32_t)(slot / p.geo.npp);
                        uint32_t q = (uint32_t)(slot % p.geo.npp);
                        uint32_t x, y;
                        bool valid;
                        path_slot_to_pixel(p.geo, q, &x, &y, &valid);
                        if (valid) {
                            Rng rng;
                            rng.key = stream_key(p.seed, stream_id(p.sample_first + s_local,
                                                                   (uint64_t)x + (uint64_t)y * p.cam.width));
                            rng.counter = 0;
                            ray = camera_get_ray(p.cam, x, y, rng);
                            rng_counter = rng.counter;
                            path = (uint32_t)slot;
                            mult = mk3(1.0f, 1.0f, 1.0f);
                            have = true;
                        }
                    }
                } else {
                    float4 a = ia[region + v], b = ib[region + v], c = ic[region + v];
                    ray.source = mk3(a.x, a.y, a.z);
                    ray.min = a.w;
                    ray.direction = mk3(b.x, b.y, b.z);
                    ray.max = kF32Max;
                    path = __float_as_uint(b.w);
                    mult = mk3(c.x, c.y, c.z);
                    rng_counter = __float_as_uint(c.w);
                    have = true;
                }
                if (have) {
                    walk_begin(w, ray);
                    phase = 2u;
                    fresh = true;
                }
            }
            next += n_idle < avail ? n_idle : avail;
        }

        // ---- scene-graph walk for new rays and for rays back from a finished search
        bool emit = false;
        Ray nray;
        V3 nmult = mk3(0, 0, 0);
        if (phase == 2u) {
            int r = walk(p.scene, w, cur, !fresh, brute, st, ts);
            fresh = false;
            if (r == kWalkMesh) {
                phase = 1u;
            } else {
                // the segment's trace is complete: work_group.rs:31-47
                traced++;
                phase = 0u;
                V3 color = mk3(0, 0, 0);
                bool finished = true;
                if (cur.hit) {
                    uint32_t s_local = path / p.geo.npp, q = path % p.geo.npp;
                    uint32_t x, y;
                    bool valid;
                    path_slot_to_pixel(p.geo, q, &x, &y, &valid);
                    Rng rng;
                    rng.key = stream_key(p.seed, stream_id(p.sample_first + s_local, (uint64_t)x + (uint64_t)y * p.cam.width));
                    rng.counter = rng_counter;
                    V3 av, dir;
                    bool is_color =
                        material_get_color(p.scene.materials[cur.material_id], w.ray.direction, cur, rng, &av, &dir);
                    rng_counter = rng.counter;
                    if (is_color) {
                        color = mult * av;
                    } else if (depth + 1u < p.trace_depth) {
                        nray.source = cur.point + dir * kEpsilon;
                        nray.direction = dir;
                        nray.min = kEpsilon;
                        nray.max = kF32Max;
                        nmult = mult * av;
                        finished = false;
                        emit = true;
                    }
                }
                if (finished) {
                    float* dst = contrib + (size_t)path * 3;
                    dst[0] = color.x;
                    dst[1] = color.y;
                    dst[2] = color.z;
                }
            }
        }
        // ---- surviving paths go to this wave's region of the output queue
        unsigned long long emit_mask = __ballot(emit);
        if (emit_mask != 0ull) {
            if (emit) {
                size_t slot = region + out_pos + (uint32_t)__popcll(emit_mask & lane_lt);
                oa[slot] = make_float4(nray.source.x, nray.source.y, nray.source.z, nray.min);
                ob[slot] = make_float4(nray.direction.x, nray.direction.y, nray.direction.z, __uint_as_float(path));
                oc[slot] = make_float4(nmult.x, nmult.y, nmult.z, __uint_as_float(rng_counter));
            }
            out_pos += (uint32_t)__popcll(emit_mask);
        }

        // ---- BVH searches
        unsigned long long search_mask = __ballot(phase == 1u);
        if (search_mask == 0ull) {
            if (next >= nin && __ballot(phase != 0u) == 0ull) break;
            continue;
        }
        const uint32_t n_search = (uint32_t)__popcll(search_mask);
        for (;;) {
            while (phase == 1u && trav_is_inner(ts)) trav_inner(ts, w.ray, st);
            if (phase == 1u && !trav_done(ts)) trav_leaf(ts, w.ray, st);
            unsigned long long unfinished = __ballot(phase == 1u && !trav_done(ts));
            uint32_t n_unfinished = (uint32_t)__popcll(unfinished);
            if (n_unfinished == 0u || n_unfinished + kRefill <= n_search) break;
        }
        if (phase == 1u && trav_done(ts)) {
            walk_mesh_result(p.scene, w, ts, &cur);
            phase = 2u;
        }
    }
    // ---- wave epilogue: queue length of the region, rays traced
    unsigned long long traced_total = 0;
    {
        uint32_t v = traced;
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        traced_total = v;
    }
    if (lane == 0u) {
        if (counts_out) counts_out[wave] = out_pos;
        if (traced_total) atomicAdd(ray_total, traced_total);
    }
}

// work_group.rs:60-61: pixel += colour, one sample after the other.
__global__ void __launch_bounds__(kBlock) accumulate_kernel(BatchGeometry g, uint32_t width, uint32_t sample_count,
                                                            const float* __restrict__ contrib, float* accum) {
    uint32_t q = blockIdx.x * kBlock + threadIdx.x;
    if (q >= g.npp) return;
    uint32_t x, y;
    bool valid;
    path_slot_to_pixel(g, q, &x, &y, &valid);
    if (!valid) return;
    float* px = accum + ((size_t)x + (size_t)y * width) * 3;
    float r = px[0], gch = px[1], b = px[2];
    for (uint32_t s = 0; s < sample_count; s++) {
        const float* c = contrib + ((size_t)s * g.npp + q) * 3;
        r = r + c[0];
        gch = gch + c[1];
        b = b + c[2];
    }
    px[0] = r;
    px[1] = gch;
    px[2] = b;
}

// cpu_renderer/mod.rs:143-175 with the per-tile scale of work_group.rs:67-69.
__global__ void __launch_bounds__(kBlock) resolve_kernel(const float* __restrict__ accum, uint32_t width,
                                                         uint32_t height, float scale, float* out) {
    uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    uint64_t n = (uint64_t)width * height;
    if (i >= n) return;
    uint32_t x = (uint32_t)(i % width), y = (uint32_t)(i / width);
    const float* src = accum + i * 3;
    float* dst = out + ((uint64_t)x + (uint64_t)(height - 1u - y) * width) * 3;
    dst[0] = src[0] * scale;
    dst[1] = src[1] * scale;
    dst[2] = src[2] * scale;
}

__global__ void __launch_bounds__(kBlock, 4) trace_closest_kernel(DScene scene, const PtRay* __restrict__ rays, size_t n,
                                                               uint32_t brute, PtHit* hits, uint32_t* spill,
                                                               uint32_t spill_threads) {
    __shared__ int32_t lds_stack[kLdsStack * kBlock];
    TravStack st = make_stack(lds_stack, spill, spill_threads);
    const size_t grid_threads = (size_t)gridDim.x * kBlock;
    for (size_t gid = (size_t)blockIdx.x * kBlock + threadIdx.x; gid < n; gid += grid_threads) {
        PtRay r = rays[gid];
        Ray ray;
        ray.source = mk3(r.source.x, r.source.y, r.source.z);
        ray.direction = mk3(r.direction.x, r.direction.y, r.direction.z);
        ray.min = r.min;
        ray.max = r.max;
        Hit h;
        scene_trace(scene, ray, brute != 0u, st, &h);
        PtHit o;
        o.hit = h.hit;
        o.hit_inside = h.hit_inside;
        o.node_index = h.hit ? h.node_index : 0xFFFFFFFFu;
        o.tri_index = h.hit ? h.tri_index : 0xFFFFFFFFu;
        o.material_id = h.material_id;
        o.t = h.t;
        o.point = {h.point.x, h.point.y, h.point.z};
        o.normal = {h.normal.x, h.normal.y, h.normal.z};
        o.uv = {h.uv.x, h.uv.y};
        hits[gid] = o;
    }
}

__global__ void __launch_bounds__(kBlock) generate_rays_kernel(DCamera cam, uint64_t seed,
                                                               const uint32_t* __restrict__ pixel_xy,
                                                               const uint64_t* __restrict__ sample, size_t n,
                                                               PtRay* rays) {
    size_t gid = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= n) return;
    uint32_t x = pixel_xy[2 * gid], y = pixel_xy[2 * gid + 1];
    Rng rng;
    rng.key = stream_key(seed, stream_id(sample[gid], (uint64_t)x + (uint64_t)y * cam.width));
    rng.counter = 0;
    Ray r = camera_get_ray(cam, x, y, rng);
    PtRay o;
    o.source = {r.source.x, r.source.y, r.source.z};
    o.direction = {r.direction.x, r.direction.y, r.direction.z};
    o.min = r.min;
    o.max = r.max;
    rays[gid] = o;
}

__global__ void __launch_bounds__(kBlock) shade_kernel(DScene scene, const PtVec3* __restrict__ dirs,
                                                       const PtHit* __restrict__ hits,
                                                       const uint64_t* __restrict__ stream, uint64_t seed, size_t n,
                                                       uint32_t* kind, PtVec3* out_a, PtVec3* out_dir) {
    size_t gid = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= n) return;
    PtHit ph = hits[gid];
    Hit h;
    h.hit = ph.hit;
    h.hit_inside = ph.hit_inside;
    h.t = ph.t;
    h.material_id = ph.material_id;
    h.node_index = ph.node_index;
    h.tri_index = ph.tri_index;
    h.point = mk3(ph.point.x, ph.point.y, ph.point.z);
    h.normal = mk3(ph.normal.x, ph.normal.y, ph.normal.z);
    h.uv = mk2(ph.uv.x, ph.uv.y);
    Rng rng;
    rng.key = stream_key(seed, stream[gid]);
    rng.counter = 0;
    V3 a = mk3(0, 0, 0), d = mk3(0, 0, 0);
    PtVec3 dv = dirs[gid];
    bool is_color = material_get_color(scene.materials[h.material_id], mk3(dv.x, dv.y, dv.z), h, rng, &a, &d);
    kind[gid] = is_color ? 0u : 1u;
    out_a[gid] = {a.x, a.y, a.z};
    out_dir[gid] = {d.x, d.y, d.z};
}

// ------------------------------------------------------------------ launchers

size_t spill_dwords_per_thread() { return (size_t)kSpillDwords; }
uint32_t queue_region_capacity(uint64_t paths, uint32_t waves) {
    uint64_t chunks = (paths + 63) / 64;
    uint64_t per_wave = (chunks + waves - 1) / waves;
    return (uint32_t)(per_wave * 64);
}
static_assert(kTraceBlock == (uint32_t)kBlock, "block size mismatch");

static inline uint32_t blocks_for(uint64_t n) { return (uint32_t)((n + kBlock - 1) / kBlock); }

hipError_t render_batch(const DScene& scene, const DCamera& cam, const BatchGeometry& geo, uint64_t seed,
                        uint64_t sample_first, uint32_t sample_count, uint32_t trace_depth, bool brute,
                        Workspace& ws, float* d_accum, hipStream_t stream,
                        std::vector<hipEvent_t>* segment_events) {
    RenderParams p;
    p.scene = scene;
    p.cam = cam;
    p.geo = geo;
    p.seed = seed;
    p.sample_first = sample_first;
    p.sample_count = sample_count;
    p.trace_depth = trace_depth;
    p.brute = brute ? 1u : 0u;
    uint64_t total = (uint64_t)sample_count * geo.npp;
    if (total == 0 || trace_depth == 0) return hipSuccess;
    if (!ws.spill || ws.persistent_blocks == 0 || total > 0xFFFFFFFFull) return hipErrorInvalidValue;
    const uint32_t waves = ws.persistent_blocks * (kBlock / 64);
    const uint32_t cap = queue_region_capacity(total, waves);
    if ((uint64_t)cap * waves > ws.capacity_paths || total > ws.contrib_paths) return hipErrorInvalidValue;
    hipError_t e;
    for (uint32_t d = 0; d < trace_depth; d++) {
        int in = d & 1, out = in ^ 1;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (segment_events) {
            if ((e = hipEventCreate(&e0)) != hipSuccess) return e;
            if ((e = hipEventCreate(&e1)) != hipSuccess) return e;
            segment_events->push_back(e0);
            segment_events->push_back(e1);
            if ((e = hipEventRecord(e0, stream)) != hipSuccess) return e;
        }
        uint32_t* counts_out = (d + 1 < trace_depth) ? ws.wave_counts[out] : nullptr;
        unsigned long long* rt = reinterpret_cast<unsigned long long*>(ws.ray_total);
        if (d == 0)
            segment_kernel<true><<<ws.persistent_blocks, kBlock, 0, stream>>>(
                p, d, cap, nullptr, nullptr, nullptr, ws.queue[out][0], ws.queue[out][1], ws.queue[out][2], nullptr,
                counts_out, ws.contrib, rt, ws.spill, ws.persistent_blocks * kBlock);
        else
            segment_kernel<false><<<ws.persistent_blocks, kBlock, 0, stream>>>(
                p, d, cap, ws.queue[in][0], ws.queue[in][1], ws.queue[in][2], ws.queue[out][0], ws.queue[out][1],
                ws.queue[out][2], ws.wave_counts[in], counts_out, ws.contrib, rt, ws.spill,
                ws.persistent_blocks * kBlock);
        if (segment_events && (e = hipEventRecord(e1, stream)) != hipSuccess) return e;
    }
    accumulate_kernel<<<blocks_for(geo.npp), kBlock, 0, stream>>>(geo, cam.width, sample_count, ws.contrib, d_accum);
    return hipGetLastError();
}

hipError_t resolve_image(const float* d_accum, uint32_t width, uint32_t height, float scale, float* d_out,
                         hipStream_t stream) {
    uint64_t n = (uint64_t)width * height;
    if (n == 0) return hipSuccess;
    resolve_kernel<<<blocks_for(n), kBlock, 0, stream>>>(d_accum, width, height, scale, d_out);
    return hipGetLastError();
}

hipError_t trace_closest(const DScene& scene, const PtRay* d_rays, size_t n, bool brute, PtHit* d_hits,
                         const Workspace& ws, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    trace_closest_kernel<<<std::min(blocks_for(n), ws.persistent_blocks), kBlock, 0, stream>>>(
        scene, d_rays, n, brute ? 1u : 0u, d_hits, ws.spill, ws.persistent_blocks * kBlock);
    return hipGetLastError();
}

hipError_t generate_rays(const DCamera& cam, uint64_t seed, const uint32_t* d_pixel_xy, const uint64_t* d_sample,
                         size_t n, PtRay* d_rays, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    generate_rays_kernel<<<blocks_for(n), kBlock, 0, stream>>>(cam, seed, d_pixel_xy, d_sample, n, d_rays);
    return hipGetLastError();
}

hipError_t shade(const DScene& scene, const PtVec3* d_dirs, const PtHit* d_hits, const uint64_t* d_stream,
                 uint64_t seed, size_t n, uint32_t* d_kind, PtVec3* d_a, PtVec3* d_dir, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    shade_kernel<<<blocks_for(n), kBlock, 0, stream>>>(scene, d_dirs, d_hits, d_stream, seed, n, d_kind, d_a, d_dir);
    return hipGetLastError();
}

} // namespace ptk

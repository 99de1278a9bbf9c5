// gfx950 kernels of the wavefront path tracer.
//
// One batch = some consecutive samples of every pixel of a rectangle.  Per batch:
//   raygen_kernel        camera rays for every path slot -> queue 0
//   bounce_kernel  x D   one path segment per live path: scene trace + material
//                        sampling; survivors are compacted into the other queue
//                        (wave ballot + one atomic per wave), finished paths
//                        store their colour into the contribution buffer
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

// Appends the calling lane's record (when `alive`) to a queue; returns its slot.
__device__ __forceinline__ uint32_t queue_reserve(bool alive, uint32_t* counter) {
    unsigned long long mask = __ballot(alive);
    uint32_t lane = __lane_id();
    uint32_t base = 0;
    if (mask != 0ull) {
        uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
        if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
        base = __shfl(base, (int)leader);
    }
    unsigned long long below = mask & ((1ull << lane) - 1ull);
    return base + (uint32_t)__popcll(below);
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

__global__ void __launch_bounds__(kBlock) raygen_kernel(RenderParams p, float4* qa, float4* qb, float4* qc,
                                                        uint32_t* counters, float* contrib) {
    uint64_t gid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    uint64_t total = (uint64_t)p.sample_count * p.geo.npp;
    bool alive = false;
    Ray ray;
    uint32_t rng_counter = 0;
    if (gid < total) {
        uint32_t s_local = (uint32_t)(gid / p.geo.npp);
        uint32_t q = (uint32_t)(gid % p.geo.npp);
        uint32_t x, y;
        bool valid;
        path_slot_to_pixel(p.geo, q, &x, &y, &valid);
        if (valid) {
            Rng rng;
            rng.key = stream_key(p.seed, stream_id(p.sample_first + s_local, (uint64_t)x + (uint64_t)y * p.cam.width));
            rng.counter = 0;
            ray = camera_get_ray(p.cam, x, y, rng);
            rng_counter = rng.counter;
            alive = true;
        }
    }
    uint32_t slot = queue_reserve(alive, &counters[0]);
    if (alive) {
        qa[slot] = make_float4(ray.source.x, ray.source.y, ray.source.z, ray.min);
        qb[slot] = make_float4(ray.direction.x, ray.direction.y, ray.direction.z, __uint_as_float((uint32_t)gid));
        qc[slot] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(rng_counter));
    }
    (void)contrib;
}

__device__ __forceinline__ TravStack make_stack(int32_t* lds_stack, uint32_t* spill, uint32_t spill_threads) {
    TravStack st;
    st.lds = &lds_stack[threadIdx.x];
    st.stride = kBlock;
    st.spill = spill + (blockIdx.x * kBlock + threadIdx.x);
    st.spill_stride = spill_threads;
    st.sp = 0;
    return st;
}

// Persistent grid: wave w of the launch takes queue chunks w, w + W, w + 2W, ...
// of 64 consecutive records (W = waves in the grid), so neighbouring chunks --
// neighbouring pixels on the first segment -- are in flight together.
__global__ void __launch_bounds__(kBlock, 4) bounce_kernel(RenderParams p, uint32_t depth, const float4* __restrict__ ia,
                                                        const float4* __restrict__ ib, const float4* __restrict__ ic,
                                                        float4* oa, float4* ob, float4* oc, uint32_t* counters,
                                                        float* contrib, unsigned long long* ray_total,
                                                        uint32_t* spill, uint32_t spill_threads) {
    __shared__ int32_t lds_stack[kLdsStack * kBlock];
    const uint32_t count = counters[depth];
    const uint32_t tid = blockIdx.x * kBlock + threadIdx.x;
    if (tid == 0) atomicAdd(ray_total, (unsigned long long)count);
    const uint32_t grid_threads = gridDim.x * kBlock;
    TravStack st = make_stack(lds_stack, spill, spill_threads);
    for (uint32_t base = tid & ~63u; base < count; base += grid_threads) {
        const uint32_t gid = base + (tid & 63u);
        bool active = gid < count;
        bool next_alive = false;
        Ray nray;
        V3 nmult = mk3(0, 0, 0);
        uint32_t path = 0, rng_counter = 0;
        if (active) {
            float4 a = ia[gid], b = ib[gid], c = ic[gid];
            Ray ray;
            ray.source = mk3(a.x, a.y, a.z);
            ray.min = a.w;
            ray.direction = mk3(b.x, b.y, b.z);
            ray.max = kF32Max;
            path = __float_as_uint(b.w);
            V3 mult = mk3(c.x, c.y, c.z);
            rng_counter = __float_as_uint(c.w);

            Hit hit;
            scene_trace(p.scene, ray, p.brute != 0u, st, &hit);

            V3 color = mk3(0, 0, 0);
            bool finished = true;
            if (hit.hit) {
                uint32_t s_local = path / p.geo.npp, q = path % p.geo.npp;
                uint32_t x, y;
                bool valid;
                path_slot_to_pixel(p.geo, q, &x, &y, &valid);
                Rng rng;
                rng.key = stream_key(p.seed, stream_id(p.sample_first + s_local, (uint64_t)x + (uint64_t)y * p.cam.width));
                rng.counter = rng_counter;
                V3 av, dir;
                bool is_color =
                    material_get_color(p.scene.materials[hit.material_id], ray.direction, hit, rng, &av, &dir);
                rng_counter = rng.counter;
                if (is_color) {
                    color = mult * av; // work_group.rs:39-41
                } else if (depth + 1u < p.trace_depth) {
                    // work_group.rs:42-46
                    nray.source = hit.point + dir * kEpsilon;
                    nray.direction = dir;
                    nray.min = kEpsilon;
                    nray.max = kF32Max;
                    nmult = mult * av;
                    finished = false;
                    next_alive = true;
                }
            }
            if (finished) {
                float* dst = contrib + (size_t)path * 3;
                dst[0] = color.x;
                dst[1] = color.y;
                dst[2] = color.z;
            }
        }
        uint32_t slot = queue_reserve(next_alive, &counters[depth + 1u]);
        if (next_alive) {
            oa[slot] = make_float4(nray.source.x, nray.source.y, nray.source.z, nray.min);
            ob[slot] = make_float4(nray.direction.x, nray.direction.y, nray.direction.z, __uint_as_float(path));
            oc[slot] = make_float4(nmult.x, nmult.y, nmult.z, __uint_as_float(rng_counter));
        }
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
    if (total > ws.capacity_paths || total > 0xFFFFFFFFull || trace_depth + 1u > kMaxDepthCounters || !ws.spill)
        return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(ws.counters, 0, sizeof(uint32_t) * (trace_depth + 1u), stream);
    if (e != hipSuccess) return e;
    // slots of pixels outside the rectangle never run a path: their contribution is never read
    raygen_kernel<<<blocks_for(total), kBlock, 0, stream>>>(p, ws.queue[0][0], ws.queue[0][1], ws.queue[0][2],
                                                             ws.counters, ws.contrib);
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
        // the queue cannot be longer than the batch; the kernel reads the real length
        bounce_kernel<<<std::min(blocks_for(total), ws.persistent_blocks), kBlock, 0, stream>>>(
            p, d, ws.queue[in][0], ws.queue[in][1], ws.queue[in][2], ws.queue[out][0], ws.queue[out][1],
            ws.queue[out][2], ws.counters, ws.contrib, reinterpret_cast<unsigned long long*>(ws.ray_total), ws.spill,
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

// gfx950 kernels of the wavefront path tracer.
//
// One batch = some consecutive samples of every pixel of a rectangle.  Per batch:
//   camera_kernel        the batch's camera rays
//   per path segment:    search_kernel (closest triangle per ray and mesh node; the
//                        searches were listed by the kernel that queued the rays:
//                        camera_kernel or the previous bounce), then the walk
//                        (scene-graph walk: the segment's hit) and the bounce
//                        (material sampling; survivors go to the wave's region of
//                        the other queue, finished paths store their colour into
//                        the contribution buffer) -- walk_bounce_kernel where the
//                        scene's shape allows it, walk_kernel + bounce_kernel otherwise
//   accumulate_kernel    per pixel, adds the batch's contributions in sample order
// (work_group.rs:26-69 is the per-pixel loop this replaces).
#include <hip/hip_runtime.h>

#include "pt_device.h"
#include "pt_kernels.h"

#include <algorithm>
#include <cstring>
#include <mutex>

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

// Division of path ids by the batch's (wave-uniform) block and slot counts, once per ray and bounce: the
// quotient estimate mulhi(n, floor(2^32 / d)) is the quotient or one less (n * floor(2^32 / d) / 2^32 lies in
// (n / d - 1, n / d]), so one correction makes it exact; the reciprocal comes from the host, in a scalar register.
struct Divisor {
    uint32_t d, inv;
};
inline Divisor make_divisor(uint32_t d) {
    Divisor v;
    v.d = d;
    v.inv = d <= 1u ? 0xFFFFFFFFu : (uint32_t)((1ull << 32) / d);
    return v;
}
__device__ __forceinline__ uint32_t divide(uint32_t n, Divisor v, uint32_t* remainder) {
    uint32_t q = __umulhi(n, v.inv);
    uint32_t r = n - q * v.d;
    if (r >= v.d) {
        q++;
        r -= v.d;
    }
    *remainder = r;
    return q;
}

struct RenderParams {
    DScene scene;
    DCamera cam;
    BatchGeometry geo;
    Divisor by_npp, by_nbx; // geo.npp, geo.nbx
    uint64_t seed;
    uint64_t sample_first;
    uint32_t sample_count;
    uint32_t trace_depth;
    uint32_t brute;
};

// Sample (within the batch) and pixel of a path id.
__device__ __forceinline__ void path_to_pixel(const RenderParams& p, uint32_t path, uint32_t* s_local, uint32_t* x, uint32_t* y) {
    uint32_t q, bx;
    *s_local = divide(path, p.by_npp, &q);
    const uint32_t by = divide(q >> 6, p.by_nbx, &bx), within = q & 63u;
    *x = p.geo.x0 + bx * 8u + (within & 7u);
    *y = p.geo.y0 + by * 8u + (within >> 3);
}

__device__ __forceinline__ TravStack make_stack(int32_t* lds_stack, uint32_t* spill, uint32_t spill_threads) {
    TravStack st;
    st.lds = (LdsI32Ptr)&lds_stack[threadIdx.x];
    st.stride = kBlock;
    st.spill = (GlobalU32Ptr)(spill + (blockIdx.x * kBlock + threadIdx.x));
    st.spill_stride = spill_threads;
    st.sp = 0;
    return st;
}

#ifndef PT_SEARCH_OCCUPANCY
#define PT_SEARCH_OCCUPANCY 8 // waves per SIMD the BVH search kernel is compiled for
#endif
#ifndef PT_REFILL
#define PT_REFILL 16
#endif
#ifndef PT_INNER_MIN
#define PT_INNER_MIN 16
#endif
constexpr uint32_t kInnerMin = PT_INNER_MIN; // node steps go on while at least this many lanes take part
constexpr uint32_t kRefill = PT_REFILL; // refill a wave once this many of its lanes have finished their search
constexpr uint32_t kGridBlocksPerCu = PT_SEARCH_OCCUPANCY;
uint32_t segment_blocks_per_cu() { return kGridBlocksPerCu; }

// Device pointers of one path segment.  Every array is cut into `regions` regions
// of `cap` records; region r is owned by one wave per kernel (wave w of a launch
// takes regions w, w + W, ...), so nothing here needs atomics:
//   rays    this segment's rays            (source,min) (direction,path id) (multiplier,draw counter)
//   next    rays of the next segment, appended per region, lengths in next_counts
//   res[m]  closest triangle of the scene's m-th mesh node for each ray that has a job for it, indexed like
//           `rays`: (t, found, leaf slot, triangle index)
struct SegmentBuffers {
    float4 *ray_a, *ray_b, *ray_c;
    uint32_t* ray_counts;
    float4 *next_a, *next_b, *next_c;
    uint32_t* next_counts;
    float4* res[kMaxDeferred];
    float4 *hit_a, *hit_b; // the segment's hits, written by walk_kernel for bounce_kernel
    float* hit_c;
    // per ray: bit m set = the ray has a job for the scene's m-th mesh node, i.e. res[m] will hold a result for
    // it; clear = the ray cannot reach that mesh and nothing is stored ("no triangle" is implied)
    uint32_t *listed, *next_listed;
    // instance run: per ray the head of its chain of jobs, the job rows that hold the chain, one result per job
    uint32_t* run_head;
    const float4* run_job_c;
    float4* run_res;
    uint32_t job_cap; // job records per region
};

// Where the kernels that queue rays list the BVH searches of those rays: one job per
// (ray, mesh node whose bounds the ray reaches), appended to the wave's region of the
// job list; job_cap records per region (a ray can reach every listed mesh node).
//   job = (object-space source,min) (direction,max) (ray's record in the region, mesh id, ordinal of the mesh node, -)
struct SearchList {
    float4 *job_a, *job_b, *job_c;
    uint32_t* job_counts;
    unsigned long long* search_total;
    uint32_t count;   // mesh nodes listed: the scene's first `count` outside the instance run
    uint32_t run;     // 1: the children of the instance run are listed too
    uint32_t run_cap; // how many of a region's job records the run's children may take
    uint32_t job_cap; // job records per region
    uint32_t* run_head; // per ray (indexed like the queue): head of its chain of run jobs
};

typedef const __attribute__((address_space(4))) MeshTarget* ConstTargetPtr;

// Lists the searches of `world` (queued at record `item` of the wave's region r,
// `queued` = this lane queued a ray): per mesh node, the ray goes through the
// transform nodes above the mesh (the arithmetic the walk kernel repeats later,
// transform.rs:45-48); if it can reach the mesh it becomes a job.  Returns the lane's
// mask of listed mesh nodes.  Wave-uniform control flow; `job_pos` counts the region's jobs.
template <bool kRun>
__device__ __forceinline__ uint32_t list_searches(const DScene& scene, const SearchList& sl, bool queued,
                                                  const Ray& world, uint32_t r, uint32_t cap, uint32_t item, uint32_t& job_pos,
                                                  uint32_t& run_used, unsigned long long lane_lt) {
    uint32_t listed = 0u; // bit m: this lane's ray got a job for mesh node m
    ConstNodePtr nodes = const_nodes(scene);
    ConstTargetPtr targets = (ConstTargetPtr)(unsigned long long)scene.targets;
    const size_t job_base = (size_t)r * sl.job_cap;
    for (uint32_t m = 0; m < sl.count; m++) {
        bool need = false;
        Ray ray = world;
        const uint32_t mesh_id = targets[m].mesh_id;
        const DMesh mesh = uniform_mesh(scene, mesh_id); // asked for before the transforms, used after them
        if (queued) {
            const uint32_t chain_len = targets[m].chain_len;
            for (uint32_t c = 0; c < chain_len; c++) ray = transform_ray_c(nodes[targets[m].chain[c]], ray);
            need = trav_reaches(mesh, ray);
            if (need) listed |= 1u << m;
        }
        const unsigned long long mask = __ballot(need);
        if (need) {
            size_t j = job_base + job_pos + (uint32_t)__popcll(mask & lane_lt);
            sl.job_a[j] = make_float4(ray.source.x, ray.source.y, ray.source.z, ray.min);
            sl.job_b[j] = make_float4(ray.direction.x, ray.direction.y, ray.direction.z, ray.max);
            sl.job_c[j] = make_float4(__uint_as_float(item), __uint_as_float(mesh_id), __uint_as_float(m), 0.0f);
        }
        job_pos += (uint32_t)__popcll(mask);
    }
    if (!kRun || sl.run == 0u) return listed;
    // The instance run: one job per child whose world-space box the ray pierces and whose mesh
    // bounds the ray reaches in the child's space.  A ray's jobs are chained through job_c.w.
    // When the region's job list is full the ray is marked and searched while walking instead.
    //
    // Rays pierce different numbers of boxes, and taking a ray into a child's space is the expensive part
    // (~190 instructions), so the two are separated: first every lane walks the box tree and puts its
    // (lane, child) pairs into the wave's list in LDS; then the pairs are worked off 64 at a time, every lane
    // busy, each lane fetching the ray of the pair's lane with cross-lane reads.  A ray's chain is linked by
    // an exchange on its head in LDS, so its jobs are chained in no particular order (walk_run takes the
    // minimum over the chain).
    constexpr uint32_t kPairs = 512, kLaneShift = 26;
    __shared__ uint32_t run_pairs[kBlock / 64][kPairs];
    __shared__ uint32_t run_heads[kBlock / 64][64];
    const uint32_t lane = (uint32_t)__popcll(lane_lt), w = threadIdx.x >> 6;
    // where every child places the same mesh, its bounds are read once, by scalar loads, instead of per pair
    const bool one_mesh = scene.run.only_mesh != 0xFFFFFFFFu;
    const DMesh run_mesh = uniform_mesh(scene, one_mesh ? scene.run.only_mesh : 0u);
    RunCursor cursor;
    run_begin(scene.run, world, queued, cursor);
    run_heads[w][lane] = kRunNone;
    uint32_t n_pairs = 0u;  // wave-uniform
    bool pierces = false;   // this lane's ray has a pair
    bool overflow = scene.run.child_count > (1u << kLaneShift); // wave-uniform: the region's list cannot take the chunk's jobs
    for (bool more = true; more && !overflow;) {
        // one node of the box tree per lane and pass, every lane at its own place in the tree
        const bool live = run_live(scene.run, cursor);
        const uint32_t k = live ? run_step(scene.run, world, cursor) : kRunNone;
        more = __ballot(live) != 0ull;
        const unsigned long long found = __ballot(k != kRunNone);
        if (k != kRunNone) {
            run_pairs[w][n_pairs + (uint32_t)__popcll(found & lane_lt)] = (lane << kLaneShift) | k;
            pierces = true;
        }
        n_pairs += (uint32_t)__popcll(found);
        if (more && n_pairs + 64u <= kPairs) continue; // room for another pass
        // work the list off: every lane takes a pair
        for (uint32_t first = 0u; first < n_pairs && !overflow; first += 64u) {
            const bool have = first + lane < n_pairs;
            const uint32_t pair = have ? run_pairs[w][first + lane] : 0u;
            const uint32_t from = pair >> kLaneShift, child = pair & ((1u << kLaneShift) - 1u);
            Ray theirs;
            theirs.source = mk3(__shfl(world.source.x, (int)from), __shfl(world.source.y, (int)from), __shfl(world.source.z, (int)from));
            theirs.direction = mk3(__shfl(world.direction.x, (int)from), __shfl(world.direction.y, (int)from),
                                   __shfl(world.direction.z, (int)from));
            theirs.min = __shfl(world.min, (int)from);
            theirs.max = __shfl(world.max, (int)from);
            const uint32_t their_item = (uint32_t)__shfl((int)item, (int)from);
            bool need = false;
            Ray ray = theirs;
            uint32_t mesh_id = 0u;
            if (have) {
                const RunChildIds ids = run_child_ids(scene.run, child);
                mesh_id = ids.mesh_id;
                ray = run_child_ray(scene.run, child, ids, theirs);
                need = one_mesh ? trav_reaches(run_mesh, ray) : trav_reaches(scene.meshes[mesh_id], ray);
            }
            const unsigned long long mask = __ballot(need);
            const uint32_t wanted = (uint32_t)__popcll(mask);
            if (run_used + wanted > sl.run_cap) { // the other mesh nodes' share of the list stays theirs
                overflow = true;
                break;
            }
            if (need) {
                const uint32_t at = job_pos + (uint32_t)__popcll(mask & lane_lt);
                const size_t j = job_base + at;
                const uint32_t previous = atomicExch(&run_heads[w][from], at);
                sl.job_a[j] = make_float4(ray.source.x, ray.source.y, ray.source.z, ray.min);
                sl.job_b[j] = make_float4(ray.direction.x, ray.direction.y, ray.direction.z, ray.max);
                sl.job_c[j] = make_float4(__uint_as_float(their_item), __uint_as_float(mesh_id),
                                          __uint_as_float(kRunJobFlag | child), __uint_as_float(previous));
            }
            job_pos += wanted;
            run_used += wanted;
        }
        n_pairs = 0u;
    }
    uint32_t head = run_heads[w][lane];
    if (overflow) {
        // the rays of this chunk that pierce a box at all are searched while walking (jobs already listed for
        // them are searched for nothing); a lane that has not met its first box yet looks for it now
        if (!pierces) pierces = run_next(scene.run, world, cursor) != kRunNone;
        head = pierces ? kRunInPlace : kRunNone;
    }
    if (queued) sl.run_head[(size_t)r * cap + item] = head;
    return listed;
}

// The hit record of the closest-hit contract (PtHit) from the tracer's own.
__device__ __forceinline__ PtHit to_pt_hit(const Hit& h) {
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
    return o;
}

// -------------------------------------------------------------- camera kernel
// Camera::get_ray (camera.rs:51-83) for every path of the batch: path slot ->
// (sample, pixel), the first draws of the path's random stream, the ray into the
// wave's region of the first segment's queue.
//
// Which wave gets which 64-slot chunk (8x8 pixels of one sample) decides what the
// chip's eight L2 caches see for the rest of the path, because survivors stay in
// their wave's region.  Workgroups b and b + 8 share an XCD and its L2, so the
// waves of one XCD take whole rows of 8x8 blocks (row y goes to XCD y mod 8), in
// row order, all samples of a block next to each other: at any moment the waves
// of an XCD work on one short strip of the image, whose rays walk the same part
// of each BVH.
constexpr uint32_t kWavesPerBlock = kBlock / 64;
// 8 when block rows are dealt to XCDs, 1 (no distinction) for grids that do not split
// evenly and for images too low to give every XCD a few rows.
__host__ __device__ inline uint32_t camera_xcd_classes(uint32_t regions, uint32_t nby) {
    return (regions % (8u * kWavesPerBlock) == 0u && nby >= 32u) ? 8u : 1u;
}

template <bool kRun>
__global__ void __launch_bounds__(kBlock)
camera_kernel(RenderParams p, SegmentBuffers b, SearchList sl, uint32_t cap, uint32_t regions) {
    const uint32_t tid = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;
    const uint32_t waves = (gridDim.x * kBlock) >> 6;
    const uint64_t total = (uint64_t)p.sample_count * p.geo.npp;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    const uint32_t xcds = camera_xcd_classes(regions, p.geo.nby);
    const uint32_t regions_per_xcd = regions / xcds;
    unsigned long long listed = 0;
    for (uint32_t r = wave; r < regions; r += waves) {
        const size_t base = (size_t)r * cap;
        uint32_t out_pos = 0u, job_pos = 0u, run_used = 0u;
        // region r belongs to workgroup r / 4, whose XCD class is (r / 4) mod 8
        const uint32_t xcd = (r / kWavesPerBlock) % xcds;
        const uint32_t index_in_xcd = (r / (kWavesPerBlock * xcds)) * kWavesPerBlock + (r % kWavesPerBlock);
        const uint32_t rows = xcd < p.geo.nby ? (p.geo.nby - xcd + xcds - 1u) / xcds : 0u; // block rows of this XCD
        const uint64_t work = (uint64_t)rows * p.geo.nbx * p.sample_count;                   // chunks of this XCD
        for (uint64_t v = index_in_xcd; v < work; v += regions_per_xcd) {
            const uint32_t s_local = (uint32_t)(v % p.sample_count);
            const uint32_t k = (uint32_t)(v / p.sample_count);
            const uint32_t bx = k % p.geo.nbx, by = (k / p.geo.nbx) * xcds + xcd;
            const uint32_t q = (by * p.geo.nbx + bx) * 64u + lane;
            const uint64_t slot = (uint64_t)s_local * p.geo.npp + q;
            bool have = false;
            Ray ray;
            uint32_t rng_counter = 0;
            if (slot < total) {
                uint32_t x, y;
                path_slot_to_pixel(p.geo, q, &x, &y, &have);
                if (have) {
                    Rng rng;
                    rng.key = stream_key(p.seed, stream_id(p.sample_first + s_local, (uint64_t)x + (uint64_t)y * p.cam.width));
                    rng.counter = 0;
                    ray = camera_get_ray(p.cam, x, y, rng);
                    rng_counter = rng.counter;
                }
            }
            unsigned long long mask = __ballot(have);
            const uint32_t item = out_pos + (uint32_t)__popcll(mask & lane_lt);
            if (have) {
                size_t o = base + item;
                b.ray_a[o] = make_float4(ray.source.x, ray.source.y, ray.source.z, ray.min);
                b.ray_b[o] = make_float4(ray.direction.x, ray.direction.y, ray.direction.z, __uint_as_float((uint32_t)slot));
                b.ray_c[o] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(rng_counter));
            }
            if (sl.count | sl.run) {
                const uint32_t listed = list_searches<kRun>(p.scene, sl, have, ray, r, cap, item, job_pos, run_used, lane_lt);
                if (have && sl.count) b.listed[base + item] = listed;
            }
            out_pos += (uint32_t)__popcll(mask);
        }
        if (lane == 0u) {
            b.ray_counts[r] = out_pos;
            if (sl.count | sl.run) sl.job_counts[r] = job_pos;
        }
        listed += job_pos;
    }
    if (lane == 0u && listed) atomicAdd(sl.search_total, listed);
}

// -------------------------------------------------------------- inject kernel
// camera_kernel's place for rays that come from the caller (pt_trace_closest on the
// wavefront route): chunk c of 64 rays goes to region c mod regions, the path id of a
// ray is its index, and the rays' searches are listed exactly as for camera rays.
template <bool kRun>
__global__ void __launch_bounds__(kBlock)
inject_kernel(DScene scene, const PtRay* __restrict__ rays, uint32_t n, SegmentBuffers b, SearchList sl, uint32_t cap,
              uint32_t regions) {
    const uint32_t tid = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;
    const uint32_t waves = (gridDim.x * kBlock) >> 6;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    const uint32_t chunks = (n + 63u) / 64u;
    unsigned long long listed = 0;
    for (uint32_t r = wave; r < regions; r += waves) {
        const size_t base = (size_t)r * cap;
        uint32_t out_pos = 0u, job_pos = 0u, run_used = 0u;
        for (uint32_t c = r; c < chunks; c += regions) {
            const uint32_t i = c * 64u + lane;
            const bool have = i < n;
            Ray ray;
            ray.source = mk3(0, 0, 0);
            ray.direction = mk3(0, 0, 0);
            ray.min = 0.0f;
            ray.max = kF32Max;
            if (have) {
                PtRay in = rays[i];
                ray.source = mk3(in.source.x, in.source.y, in.source.z);
                ray.direction = mk3(in.direction.x, in.direction.y, in.direction.z);
                ray.min = in.min;
            }
            unsigned long long mask = __ballot(have);
            const uint32_t item = out_pos + (uint32_t)__popcll(mask & lane_lt);
            if (have) {
                size_t o = base + item;
                b.ray_a[o] = make_float4(ray.source.x, ray.source.y, ray.source.z, ray.min);
                b.ray_b[o] = make_float4(ray.direction.x, ray.direction.y, ray.direction.z, __uint_as_float(i));
                b.ray_c[o] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(0u));
            }
            if (sl.count | sl.run) {
                const uint32_t mine = list_searches<kRun>(scene, sl, have, ray, r, cap, item, job_pos, run_used, lane_lt);
                if (have && sl.count) b.listed[base + item] = mine;
            }
            out_pos += (uint32_t)__popcll(mask);
        }
        if (lane == 0u) {
            b.ray_counts[r] = out_pos;
            if (sl.count | sl.run) sl.job_counts[r] = job_pos;
        }
        listed += job_pos;
    }
    if (lane == 0u && listed) atomicAdd(sl.search_total, listed);
}

// ---------------------------------------------------------------- walk kernel
// hierarchy.trace_ray of a path segment (work_group.rs:30) with the BVH searches
// already done: the scene-graph walk, with the closest triangle of every mesh node
// read from res[].  Writes the segment's hit, indexed like the ray:
//   hit_a (point.xyz, material id | hit_inside << 31; 0xFFFFFFFF = nothing hit)
//   hit_b (normal.xyz, uv.x)      hit_c uv.y
// Scenes with more than kMaxDeferred mesh nodes, and the linear-scan mode, search
// the remaining meshes in place.  kFlat: the scene has a flat root (walk_flat);
// otherwise the general resumable walk.
// Waves per SIMD the kernel is compiled for: four without the instance-run code (128 registers:
// the loop over a flat root's children and the general walk both fit, and so does the flat loop with the
// run code); three for the general walk with the run code.
// kFull: the whole hit record of every ray goes to full_hits[path id] as well (the closest-hit
// contract on the wavefront route; a render needs only what the bounce reads).
template <bool kRun, bool kFlat, bool kFull = false>
__global__ void __launch_bounds__(kBlock, kRun && !kFlat ? 3 : 4)
walk_kernel(RenderParams p, SegmentBuffers b, uint32_t cap, uint32_t regions, unsigned long long* ray_total,
            uint32_t* spill, uint32_t spill_threads, PtHit* full_hits = nullptr) {
    __shared__ int32_t lds_stack[kLdsStack * kBlock];
    const uint32_t tid = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;
    const uint32_t waves = (gridDim.x * kBlock) >> 6;
    const bool brute = p.brute != 0u;
    TravStack st = make_stack(lds_stack, spill, spill_threads);
    unsigned long long traced = 0;

    for (uint32_t r = wave; r < regions; r += waves) {
        const size_t base = (size_t)r * cap;
        const uint32_t n_items = b.ray_counts[r];
        traced += n_items;
        for (uint32_t first = 0; first < n_items; first += 64u) {
            const uint32_t item = first + lane;
            if (item >= n_items) continue;
            float4 ra = b.ray_a[base + item], rb = b.ray_b[base + item];
            Ray ray;
            ray.source = mk3(ra.x, ra.y, ra.z);
            ray.min = ra.w;
            ray.direction = mk3(rb.x, rb.y, rb.z);
            ray.max = kF32Max;
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
            // which of the searched mesh nodes hold a triangle for any ray of this wave
            uint32_t live = 0xFFFFFFFFu;
            uint32_t listed = 0u; // which of them this ray has a result for
            if (!brute) {
                const uint32_t searched = p.scene.target_count < (uint32_t)kMaxDeferred ? p.scene.target_count : (uint32_t)kMaxDeferred;
                if (searched) listed = b.listed[base + item];
                for (uint32_t m = 0; m < searched; m++) {
                    uint32_t found = 0u;
                    if ((listed >> m) & 1u) found = __float_as_uint(b.res[m][base + item].y);
                    if (__ballot(found != 0u) == 0ull) live &= ~(1u << m);
                }
            }
            // the searches run ahead inside the instance run, if the scene has one
            RunLookup look;
            const bool have_run = kRun && !brute && p.scene.run.child_count != 0u;
            if (have_run) {
                look.head = b.run_head[base + item];
                look.job_c = b.run_job_c + (size_t)r * b.job_cap;
                look.run_res = b.run_res + (size_t)r * b.job_cap;
            }
            if (kFlat) {
                walk_flat<kRun>(p.scene, ray, b.res, base + item, (uint32_t)kMaxDeferred, listed, live, have_run ? &look : nullptr, st, &cur, kFull);
                if (kFull) full_hits[__float_as_uint(rb.w)] = to_pt_hit(cur);
                uint32_t tag = cur.hit ? (cur.material_id | (cur.hit_inside << 31)) : 0xFFFFFFFFu;
                b.hit_a[base + item] = make_float4(cur.point.x, cur.point.y, cur.point.z, __uint_as_float(tag));
                b.hit_b[base + item] = make_float4(cur.normal.x, cur.normal.y, cur.normal.z, cur.uv.x);
                b.hit_c[base + item] = cur.uv.y;
                continue;
            }
            WalkState w;
            TravState ts;
            walk_begin(w, ray);
            bool resume = false;
            while (walk<kRun>(p.scene, w, cur, resume, brute, st, kFull, live, have_run ? &look : nullptr) == kWalkMesh) {
                const uint32_t m = const_nodes(p.scene)[__builtin_amdgcn_readfirstlane(w.i)].mesh_ordinal;
                if (m < (uint32_t)kMaxDeferred) {
                    float4 res = make_float4(f32_inf(), __uint_as_float(0u), __uint_as_float(0u), __uint_as_float(0xFFFFFFFFu));
                    if ((listed >> m) & 1u) res = b.res[m][base + item];
                    ts.tbest = res.x;
                    ts.found = __float_as_uint(res.y);
                    ts.best_slot = __float_as_uint(res.z);
                    ts.best_orig = __float_as_uint(res.w);
                } else {
                    search_in_place(walk_mesh(p.scene, w), w.ray, st, ts);
                }
                walk_mesh_result(p.scene, w, ts, &cur);
                resume = true;
            }
            if (kFull) full_hits[__float_as_uint(rb.w)] = to_pt_hit(cur);
            uint32_t tag = cur.hit ? (cur.material_id | (cur.hit_inside << 31)) : 0xFFFFFFFFu;
            b.hit_a[base + item] = make_float4(cur.point.x, cur.point.y, cur.point.z, __uint_as_float(tag));
            b.hit_b[base + item] = make_float4(cur.normal.x, cur.normal.y, cur.normal.z, cur.uv.x);
            b.hit_c[base + item] = cur.uv.y;
        }
    }
    if (lane == 0u && traced) atomicAdd(ray_total, traced);
}

// -------------------------------------------------------------- bounce kernel
// The rest of a path segment (work_group.rs:31-47): Material::get_color on the
// hit the walk kernel found, texel fetch, then the next segment's ray into the
// wave's region of the other queue, or the path's final colour into the
// contribution buffer.
#ifndef PT_BOUNCE_OCCUPANCY
#define PT_BOUNCE_OCCUPANCY 5
#endif
// With the instance-run listing the kernel is asked for three waves per SIMD: the compiler then takes 124 registers
// and spills none (four waves fit after all); asked for five it takes 96 and spills 15, asked for four, 110 and none:
// the 115-instance frame 64.0 / 63.6 / 62.0 ms at 5 / 4 / 3.
#ifndef PT_BOUNCE_RUN_OCCUPANCY
#define PT_BOUNCE_RUN_OCCUPANCY 3
#endif
template <bool kRun>
__global__ void __launch_bounds__(kBlock, kRun ? PT_BOUNCE_RUN_OCCUPANCY : PT_BOUNCE_OCCUPANCY)
bounce_kernel(RenderParams p, uint32_t depth, SegmentBuffers b, SearchList sl, uint32_t cap, uint32_t regions,
              float* contrib) {
    const uint32_t tid = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;
    const uint32_t waves = (gridDim.x * kBlock) >> 6;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    unsigned long long listed = 0;
    // the first kLdsMaterials materials are read from a copy in LDS: one dependent trip to memory less per hit
    constexpr uint32_t kLdsMaterials = 32;
    __shared__ DMaterial lds_materials[kLdsMaterials];
    const uint32_t n_lds = p.scene.material_count < kLdsMaterials ? p.scene.material_count : kLdsMaterials;
    static_assert(sizeof(DMaterial) % 4 == 0, "DMaterial is copied by dwords");
    for (uint32_t i = threadIdx.x; i < n_lds * (uint32_t)(sizeof(DMaterial) / 4); i += kBlock)
        reinterpret_cast<uint32_t*>(lds_materials)[i] = reinterpret_cast<const uint32_t*>(p.scene.materials)[i];
    __syncthreads();

    for (uint32_t r = wave; r < regions; r += waves) {
        const size_t base = (size_t)r * cap;
        const uint32_t n_items = b.ray_counts[r];
        uint32_t out_pos = 0u, job_pos = 0u, run_used = 0u;
        // The records of a chunk are fetched while the chunk before it is worked on: the material and
        // the texel of a hit are two more dependent fetches, and a wave has nothing else to wait with.
        struct Records {
            float4 rb, rc, ha, hb;
            float hc;
        };
        auto fetch = [&](uint32_t first) {
            Records rec;
            const uint32_t item = first + lane;
            if (item < n_items) {
                rec.rb = b.ray_b[base + item];
                rec.rc = b.ray_c[base + item];
                rec.ha = b.hit_a[base + item];
                rec.hb = b.hit_b[base + item];
                rec.hc = b.hit_c[base + item];
            }
            return rec;
        };
        Records upcoming = fetch(0u);
        for (uint32_t first = 0; first < n_items; first += 64u) {
            const uint32_t item = first + lane;
            const Records rec = upcoming;
            if (first + 64u < n_items) upcoming = fetch(first + 64u);
            bool emit = false;
            Ray nray;
            V3 nmult = mk3(0, 0, 0);
            uint32_t path = 0, rng_counter = 0;
            if (item < n_items) {
                const float4 rb = rec.rb, rc = rec.rc, ha = rec.ha;
                path = __float_as_uint(rb.w);
                rng_counter = __float_as_uint(rc.w);
                const uint32_t tag = __float_as_uint(ha.w);
                V3 color = mk3(0, 0, 0);
                bool finished = true;
                if (tag != 0xFFFFFFFFu) {
                    const float4 hb = rec.hb;
                    Hit cur;
                    cur.hit = 1u;
                    cur.hit_inside = tag >> 31;
                    cur.material_id = tag & 0x7FFFFFFFu;
                    cur.t = 0.0f;
                    cur.node_index = 0u;
                    cur.tri_index = 0u;
                    cur.point = mk3(ha.x, ha.y, ha.z);
                    cur.normal = mk3(hb.x, hb.y, hb.z);
                    cur.uv = mk2(hb.w, rec.hc);
                    V3 mult = mk3(rc.x, rc.y, rc.z);
                    uint32_t s_local, x, y;
                    path_to_pixel(p, path, &s_local, &x, &y);
                    Rng rng;
                    rng.key = stream_key(p.seed, stream_id(p.sample_first + s_local, (uint64_t)x + (uint64_t)y * p.cam.width));
                    rng.counter = rng_counter;
                    V3 av, dir;
                    const DMaterial* material = cur.material_id < n_lds ? &lds_materials[cur.material_id] : &p.scene.materials[cur.material_id];
                    bool is_color = material_get_color(*material, mk3(rb.x, rb.y, rb.z), cur, rng, &av, &dir);
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
            // surviving paths go to this region of the next segment's queue
            unsigned long long emit_mask = __ballot(emit);
            const uint32_t next_item = out_pos + (uint32_t)__popcll(emit_mask & lane_lt);
            if (emit) {
                size_t slot = base + next_item;
                b.next_a[slot] = make_float4(nray.source.x, nray.source.y, nray.source.z, nray.min);
                b.next_b[slot] = make_float4(nray.direction.x, nray.direction.y, nray.direction.z, __uint_as_float(path));
                b.next_c[slot] = make_float4(nmult.x, nmult.y, nmult.z, __uint_as_float(rng_counter));
            }
            if ((sl.count | sl.run) && emit_mask != 0ull) {
                const uint32_t listed = list_searches<kRun>(p.scene, sl, emit, nray, r, cap, next_item, job_pos, run_used, lane_lt);
                if (emit && sl.count) b.next_listed[base + next_item] = listed;
            }
            out_pos += (uint32_t)__popcll(emit_mask);
        }
        if (lane == 0u) {
            b.next_counts[r] = out_pos;
            if (sl.count | sl.run) sl.job_counts[r] = job_pos;
        }
        listed += job_pos;
    }
    if (lane == 0u && listed) atomicAdd(sl.search_total, listed);
}

// ------------------------------------------------------- walk + bounce in one
// Scenes with a flat root and no instance run: walk_flat and the bounce in one kernel, the hit
// staying in registers (the two kernels hand each other 36 bytes per ray through memory, and the
// bounce is bound by its memory traffic).  Four waves per SIMD: built without the SLP vectoriser (its
// packed two-float operations want aligned register pairs and the moves to form them) the kernel fits
// 128 registers.  With four waves there is little to cover a trip to memory with, so the kernel keeps such
// trips off its critical path: the next chunk's records come into LDS by direct loads while this chunk is
// worked on (registers could not hold them), what all lanes share (mesh descriptions, scene nodes) comes by
// scalar loads, the materials are read from LDS, and the texel of a diffuse bounce is fetched before the
// arithmetic that does not need it.  Scenes with an instance run keep the two kernels: the
// walk reads the segment's job list while the bounce overwrites it with the next segment's.
// kFull: as in walk_kernel.
#ifndef PT_WALK_BOUNCE_OCCUPANCY
#define PT_WALK_BOUNCE_OCCUPANCY 4
#endif
constexpr uint32_t kWalkBounceMaterials = 32; // materials a scene may have to take this kernel
template <bool kFull = false>
__global__ void __launch_bounds__(kBlock, PT_WALK_BOUNCE_OCCUPANCY)
walk_bounce_kernel(RenderParams p, uint32_t depth, SegmentBuffers b, SearchList sl, uint32_t cap, uint32_t regions,
                   float* contrib, unsigned long long* ray_total, uint32_t* spill, uint32_t spill_threads,
                   PtHit* full_hits = nullptr) {
    __shared__ int32_t lds_stack[kLdsStack * kBlock];
    const uint32_t tid = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;
    const uint32_t waves = (gridDim.x * kBlock) >> 6;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    TravStack st = make_stack(lds_stack, spill, spill_threads);
    unsigned long long traced = 0, jobs_listed = 0;
    // every material of the scene is read from a copy in LDS (the launcher keeps scenes with more than
    // kWalkBounceMaterials on the two-kernel route): LDS reads, not flat ones that could be either
    __shared__ DMaterial lds_materials[kWalkBounceMaterials];
    const uint32_t n_lds = p.scene.material_count < kWalkBounceMaterials ? p.scene.material_count : kWalkBounceMaterials;
    for (uint32_t i = threadIdx.x; i < n_lds * (uint32_t)(sizeof(DMaterial) / 4); i += kBlock)
        reinterpret_cast<uint32_t*>(lds_materials)[i] = reinterpret_cast<const uint32_t*>(p.scene.materials)[i];
    __syncthreads();
    const uint32_t searched = p.scene.target_count < (uint32_t)kMaxDeferred ? p.scene.target_count : (uint32_t)kMaxDeferred;
    // The records of the next chunk (ray, listed searches, the first mesh node's result) are brought into LDS
    // by loads that write LDS directly while this chunk is worked on: no registers are held for them.
    __shared__ float4 ahead[kBlock / 64][4][64];
    __shared__ uint32_t ahead_listed[kBlock / 64][64];
    const uint32_t w = threadIdx.x >> 6;

    for (uint32_t r = wave; r < regions; r += waves) {
        const size_t base = (size_t)r * cap;
        const uint32_t n_items = b.ray_counts[r];
        traced += n_items;
        uint32_t out_pos = 0u, job_pos = 0u, run_used = 0u;
        auto fetch_ahead = [&](uint32_t first) { // n_items != 0; lanes past the end fetch the last record again
            const uint32_t it = first + lane < n_items ? first + lane : n_items - 1u;
            const size_t at = base + it;
            lds_dma16(b.ray_a + at, &ahead[w][0][0]);
            lds_dma16(b.ray_b + at, &ahead[w][1][0]);
            lds_dma16(b.ray_c + at, &ahead[w][2][0]);
            if (searched) {
                lds_dma4(b.listed + at, &ahead_listed[w][0]);
                lds_dma16(b.res[0] + at, &ahead[w][3][0]);
            }
        };
        if (n_items) fetch_ahead(0u);
        for (uint32_t first = 0; first < n_items; first += 64u) {
            const uint32_t item = first + lane;
            const bool active = item < n_items;
            bool emit = false;
            Ray nray;
            V3 nmult = mk3(0, 0, 0);
            uint32_t path = 0, rng_counter = 0;
            // ---- hierarchy.trace_ray (work_group.rs:30)
            float4 ra = make_float4(0, 0, 0, 0), rb = ra, rc = ra;
            uint32_t listed = 0u;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the records have landed
            ra = ahead[w][0][lane];
            rb = ahead[w][1][lane];
            rc = ahead[w][2][lane];
            if (searched) listed = active ? ahead_listed[w][lane] : 0u;
            uint32_t live = 0xFFFFFFFFu;
            for (uint32_t m = 0; m < searched; m++) {
                uint32_t found = 0u;
                if ((listed >> m) & 1u) found = __float_as_uint(m == 0u ? ahead[w][3][lane].y : b.res[m][base + item].y);
                if (__ballot(found != 0u) == 0ull) live &= ~(1u << m);
            }
            Ray ray;
            ray.source = mk3(ra.x, ra.y, ra.z);
            ray.min = ra.w;
            ray.direction = mk3(rb.x, rb.y, rb.z);
            ray.max = kF32Max;
            Hit cur;
            cur.hit = 0u;
            if (active)
                walk_flat<false>(p.scene, ray, b.res, base + item, (uint32_t)kMaxDeferred, listed, live, nullptr, st, &cur, kFull,
                                 &ahead[w][3][lane]);
            // this chunk's copy has been read: the next chunk's records may overwrite it
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (first + 64u < n_items) fetch_ahead(first + 64u);
            if (active) {
                // ---- the rest of the segment (work_group.rs:31-47)
                path = __float_as_uint(rb.w);
                if (kFull) full_hits[path] = to_pt_hit(cur);
                rng_counter = __float_as_uint(rc.w);
                V3 color = mk3(0, 0, 0);
                bool finished = true;
                if (cur.hit) {
                    V3 mult = mk3(rc.x, rc.y, rc.z);
                    uint32_t s_local, x, y;
                    path_to_pixel(p, path, &s_local, &x, &y);
                    Rng rng;
                    rng.key = stream_key(p.seed, stream_id(p.sample_first + s_local, (uint64_t)x + (uint64_t)y * p.cam.width));
                    rng.counter = rng_counter;
                    V3 av, dir;
                    const DMaterial* material = &lds_materials[cur.material_id < n_lds ? cur.material_id : 0u];
                    bool is_color = material_get_color(*material, ray.direction, cur, rng, &av, &dir);
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
            unsigned long long emit_mask = __ballot(emit);
            const uint32_t next_item = out_pos + (uint32_t)__popcll(emit_mask & lane_lt);
            if (emit) {
                size_t slot = base + next_item;
                b.next_a[slot] = make_float4(nray.source.x, nray.source.y, nray.source.z, nray.min);
                b.next_b[slot] = make_float4(nray.direction.x, nray.direction.y, nray.direction.z, __uint_as_float(path));
                b.next_c[slot] = make_float4(nmult.x, nmult.y, nmult.z, __uint_as_float(rng_counter));
            }
            if (sl.count && emit_mask != 0ull) {
                const uint32_t next_listed = list_searches<false>(p.scene, sl, emit, nray, r, cap, next_item, job_pos, run_used, lane_lt);
                if (emit) b.next_listed[base + next_item] = next_listed;
            }
            out_pos += (uint32_t)__popcll(emit_mask);
        }
        if (lane == 0u) {
            b.next_counts[r] = out_pos;
            if (sl.count) sl.job_counts[r] = job_pos;
        }
        jobs_listed += job_pos;
    }
    if (lane == 0u && traced) atomicAdd(ray_total, traced);
    if (lane == 0u && jobs_listed) atomicAdd(sl.search_total, jobs_listed);
}

// -------------------------------------------------------------- search kernel
struct SearchResults {
    float4* res[kMaxDeferred]; // per listed mesh node: closest triangle per ray, indexed like the ray queue
    float4* run_res;           // jobs of the instance run: closest triangle per job, indexed like the job list
};

// The BVH searches listed by the kernels that queued the rays (list_searches), for
// all mesh nodes in one launch.  This is where the time goes, so the
// kernel carries nothing but the search: ~64 registers per lane, 8 waves per
// SIMD to cover the latency of the dependent node fetches.  Lanes do not wait for
// the slowest search of their wave: each lane is a small state machine and the
// wave hands idle lanes new jobs whenever kRefill searches have finished (search
// lengths differ by an order of magnitude between neighbouring rays).
// kOneMesh: the scene has a single mesh, `one`; its node and triangle arrays are then kernel
// arguments (scalar registers) instead of per-lane state read from the mesh table at every refill.
// kCount: node records and triangle records fetched are counted into visit_totals[0] / [1] (the
// measured side of the kernel's byte count; two more registers, so timed runs use the other variant).
template <bool kOneMesh, bool kCount = false>
__global__ void __launch_bounds__(kBlock, PT_SEARCH_OCCUPANCY)
search_kernel(const DMesh* __restrict__ meshes, DMesh one, const float4* __restrict__ job_a, const float4* __restrict__ job_b,
              const float4* __restrict__ job_c, const uint32_t* __restrict__ job_counts, SearchResults out, uint32_t cap,
              uint32_t job_cap, uint32_t regions, uint32_t* spill, uint32_t spill_threads,
              unsigned long long* visit_totals = nullptr) {
    __shared__ int32_t lds_stack[kLdsStack * kBlock];
    const uint32_t tid = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;
    const uint32_t waves = (gridDim.x * kBlock) >> 6;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    TravStack st = make_stack(lds_stack, spill, spill_threads);

    uint32_t region = wave;                                   // region being drained (wave-uniform)
    uint32_t n_jobs = region < regions ? job_counts[region] : 0u;
    uint32_t next = 0u;
    TravState ts;
    ts.cur = kBvhSentinel;
    if (kOneMesh) {
        ts.nodes = one.nodes;
        ts.tris = one.tris;
    }
    Ray ray;
    float4* res_out = nullptr;
    bool busy = false;
    uint32_t nodes_fetched = 0u, tris_fetched = 0u;
    uint32_t inner_steps = 0u, leaf_steps = 0u, leaf_rounds = 0u; // wave-level: steps taken, triangle rounds inside leaf steps

    // Every pass of the loops below retires at least one step of some lane's search, and
    // a search ends: inner children have larger indices than their parents (WideNode),
    // so the nodes a lane opens cannot repeat.
    for (;;) {
        // ---- hand jobs to idle lanes
        unsigned long long idle_mask = __ballot(!busy);
        while (idle_mask != 0ull && region < regions) {
            uint32_t avail = n_jobs - next;
            if (avail == 0u) {
                region += waves;
                next = 0u;
                n_jobs = region < regions ? job_counts[region] : 0u;
                continue;
            }
            uint32_t n_idle = (uint32_t)__popcll(idle_mask);
            uint32_t rank = (uint32_t)__popcll(idle_mask & lane_lt);
            if (!busy && rank < avail) {
                size_t j = (size_t)region * job_cap + next + rank;
                float4 ja = job_a[j], jb = job_b[j], jc = job_c[j];
                ray.source = mk3(ja.x, ja.y, ja.z);
                ray.min = ja.w;
                ray.direction = mk3(jb.x, jb.y, jb.z);
                ray.max = jb.w;
                uint32_t v = __float_as_uint(jc.x), mesh_id = __float_as_uint(jc.y), m = __float_as_uint(jc.z);
                res_out = (m & kRunJobFlag) ? out.run_res + j : out.res[m] + ((size_t)region * cap + v);
                // whoever listed the job has already checked the ray against the mesh bounds
                ts.inv = search_inverse_direction(ray.direction);
                ts.ray_limit = (ray.max == ray.max) ? ray.max : f32_inf();
                if (kOneMesh) {
                    ts.cur = one.root_ref;
                } else {
                    const DMesh& mesh = meshes[mesh_id];
                    ts.nodes = mesh.nodes;
                    ts.tris = mesh.tris;
                    ts.cur = mesh.root_ref;
                }
                ts.tbest = f32_inf();
                ts.best_slot = 0u;
                ts.best_orig = 0xFFFFFFFFu;
                ts.found = 0u;
                st.sp = 0;
                busy = true;
            }
            next += n_idle < avail ? n_idle : avail;
            idle_mask = __ballot(!busy);
        }
        unsigned long long busy_mask = __ballot(busy);
        if (busy_mask == 0ull) break; // no job left for this wave
        const uint32_t n_busy = (uint32_t)__popcll(busy_mask);
        const bool more = region < regions;
        // ---- search until enough lanes are done to make a refill worthwhile
        for (;;) {
            // node steps while enough lanes take part, then the leaves that have piled up
            // (other schedules -- strict alternation, whichever kind more lanes wait for, one
            // triangle per step -- came out within 2 % of this one)
            for (;;) {
                bool at_inner = busy && trav_is_inner(ts);
                uint32_t n_inner = (uint32_t)__popcll(__ballot(at_inner));
                if (n_inner == 0u) break;
                if (n_inner < kInnerMin) {
                    uint32_t n_leaf = (uint32_t)__popcll(__ballot(busy && !trav_done(ts) && !trav_is_inner(ts)));
                    if (n_leaf != 0u) break;
                }
                if (kCount) inner_steps++;
                if (at_inner) {
                    if (kCount) nodes_fetched++;
                    trav_inner(ts, ray, st);
                }
            }
            const bool at_leaf = busy && !trav_done(ts) && !trav_is_inner(ts);
            if (kCount) {
                // a leaf step lasts as many triangle rounds as its fullest leaf has triangles
                uint32_t mine = at_leaf ? ((uint32_t)(~ts.cur) & 7u) + 1u : 0u, rounds = 0u;
                for (uint32_t c = 1; c <= 4u; c++) rounds += __ballot(mine >= c) != 0ull ? 1u : 0u;
                leaf_steps += rounds ? 1u : 0u;
                leaf_rounds += rounds;
            }
            if (at_leaf) {
                if (kCount) tris_fetched += ((uint32_t)(~ts.cur) & 7u) + 1u;
                trav_leaf(ts, ray, st);
            }
            uint32_t n_unfinished = (uint32_t)__popcll(__ballot(busy && !trav_done(ts)));
            if (n_unfinished == 0u || (more && n_unfinished + kRefill <= n_busy)) break;
        }
        if (busy && trav_done(ts)) {
            *res_out = make_float4(ts.tbest, __uint_as_float(ts.found), __uint_as_float(ts.best_slot),
                                   __uint_as_float(ts.best_orig));
            busy = false;
        }
    }
    if (kCount) {
        unsigned long long n = nodes_fetched, t = tris_fetched;
        for (int off = 32; off > 0; off >>= 1) {
            n += __shfl_down(n, off, 64);
            t += __shfl_down(t, off, 64);
        }
        if (lane == 0u) {
            atomicAdd(visit_totals, n);
            atomicAdd(visit_totals + 1, t);
            atomicAdd(visit_totals + 2, (unsigned long long)inner_steps);
            atomicAdd(visit_totals + 3, (unsigned long long)leaf_steps);
            atomicAdd(visit_totals + 4, (unsigned long long)leaf_rounds);
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

// client/src/frame.rs:61-71 (RenderSum::add_render): sum += render, element by
// element.  kVec: both buffers are 16-byte aligned, four floats per thread.
template <bool kVec>
__global__ void __launch_bounds__(kBlock) frame_add_kernel(float* sum, const float* __restrict__ render,
                                                           uint64_t floats) {
    uint64_t i = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) * 4;
    if (i >= floats) return;
    if (kVec && i + 4 <= floats) {
        float4 a = *reinterpret_cast<const float4*>(sum + i);
        float4 b = *reinterpret_cast<const float4*>(render + i);
        a.x = a.x + b.x;
        a.y = a.y + b.y;
        a.z = a.z + b.z;
        a.w = a.w + b.w;
        *reinterpret_cast<float4*>(sum + i) = a;
        return;
    }
    for (uint64_t k = i; k < i + 4 && k < floats; k++) sum[k] = sum[k] + render[k];
}

// client/src/frame.rs:91-93: (c / (1 + c) * 255) as u8.  Rust's cast truncates
// toward zero, saturates at both ends and maps NaN to 0.
__device__ inline uint32_t tone_u8(float c) {
    float v = c / (1.0f + c) * 255.0f;
    if (!(v > 0.0f)) return 0u;
    if (v >= 255.0f) return 255u;
    return (uint32_t)v;
}

// client/src/frame.rs:73-103 (RenderSum::into_image, gamma_correction): the mean
// image sum / divisor and its RGBA8 preview; either output may be null.  Four
// pixels per thread: 48 bytes in, 48 + 16 bytes out.
template <bool kVec>
__global__ void __launch_bounds__(kBlock) frame_image_kernel(const float* __restrict__ sum, uint64_t pixels,
                                                             float divisor, float* out_rgb, uint32_t* out_rgba) {
    uint64_t p0 = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) * 4;
    if (p0 >= pixels) return;
    float c[12];
    const bool whole = p0 + 4 <= pixels;
    if (kVec && whole) {
        const float4* src = reinterpret_cast<const float4*>(sum + p0 * 3);
        for (int k = 0; k < 3; k++) {
            float4 v = src[k];
            c[4 * k] = v.x;
            c[4 * k + 1] = v.y;
            c[4 * k + 2] = v.z;
            c[4 * k + 3] = v.w;
        }
    } else {
        for (int k = 0; k < 12; k++) c[k] = p0 * 3 + k < pixels * 3 ? sum[p0 * 3 + k] : 0.0f;
    }
    for (int k = 0; k < 12; k++) c[k] = c[k] / divisor;
    uint32_t packed[4];
    for (int q = 0; q < 4; q++)
        packed[q] = tone_u8(c[3 * q]) | (tone_u8(c[3 * q + 1]) << 8) | (tone_u8(c[3 * q + 2]) << 16) | 0xFF000000u;
    if (kVec && whole) {
        if (out_rgb) {
            float4* dst = reinterpret_cast<float4*>(out_rgb + p0 * 3);
            for (int k = 0; k < 3; k++) dst[k] = make_float4(c[4 * k], c[4 * k + 1], c[4 * k + 2], c[4 * k + 3]);
        }
        if (out_rgba) *reinterpret_cast<uint4*>(out_rgba + p0) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
        return;
    }
    for (int q = 0; q < 4 && p0 + q < pixels; q++) {
        if (out_rgb)
            for (int ch = 0; ch < 3; ch++) out_rgb[(p0 + q) * 3 + ch] = c[3 * q + ch];
        if (out_rgba) out_rgba[p0 + q] = packed[q];
    }
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
        hits[gid] = to_pt_hit(h);
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
uint32_t traversal_stack_capacity() { return (uint32_t)(kLdsStack + kSpillStack); }
uint32_t queue_region_capacity(uint32_t nbx, uint32_t nby, uint32_t samples, uint32_t waves) {
    // the XCD with the most block rows sets the size (see camera_kernel)
    uint32_t xcds = camera_xcd_classes(waves, nby);
    uint64_t chunks = (uint64_t)((nby + xcds - 1) / xcds) * nbx * samples;
    uint64_t waves_per_xcd = waves / xcds;
    uint64_t per_wave = (chunks + waves_per_xcd - 1) / waves_per_xcd;
    return (uint32_t)(per_wave * 64);
}
static_assert(kTraceBlock == (uint32_t)kBlock, "block size mismatch");

static inline uint32_t blocks_for(uint64_t n) { return (uint32_t)((n + kBlock - 1) / kBlock); }

uint32_t injected_region_capacity(size_t n, uint32_t waves) {
    uint64_t chunks = ((uint64_t)n + 63) / 64;
    uint64_t per_wave = waves ? (chunks + waves - 1) / waves : chunks;
    return (uint32_t)(std::max<uint64_t>(per_wave, 1) * 64);
}

namespace {

// The device pointers of a launch sequence over a workspace whose queues are cut into
// `regions` regions of `cap` records.
struct SegmentPlan {
    uint32_t cap = 0, regions = 0;
    uint32_t searched = 0;  // mesh nodes searched ahead of the walk
    uint32_t run_slots = 0; // job slots per ray for the children of the instance run
    uint32_t spill_threads = 0;
    SearchList sl;
    SearchResults sr;
    unsigned long long* totals = nullptr;
    const Workspace* ws = nullptr;

    SegmentBuffers buffers(int in) const {
        const Workspace& w = *ws;
        int out = in ^ 1;
        SegmentBuffers b;
        b.ray_a = w.queue[in][0];
        b.ray_b = w.queue[in][1];
        b.ray_c = w.queue[in][2];
        b.ray_counts = w.wave_counts[in];
        b.next_a = w.queue[out][0];
        b.next_b = w.queue[out][1];
        b.next_c = w.queue[out][2];
        b.next_counts = w.wave_counts[out];
        for (int k = 0; k < kMaxDeferred; k++) b.res[k] = w.results[k];
        b.hit_a = w.hits[0];
        b.hit_b = w.hits[1];
        b.hit_c = w.hit_uv_y;
        b.listed = w.listed[in];
        b.next_listed = w.listed[out];
        b.run_head = w.run_head;
        b.run_job_c = w.jobs[2];
        b.run_res = w.run_res;
        b.job_cap = cap * (searched + run_slots);
        return b;
    }
};

// Checks that the workspace holds what the launches will touch (a kernel handed a null or short
// array would fault the GPU) and fills the plan.
hipError_t make_plan(const DScene& scene, size_t target_count, uint32_t run_slots_cfg, bool brute, const Workspace& ws,
                     uint32_t cap, uint64_t paths, SegmentPlan* plan) {
    if (!ws.spill || !ws.wave_counts[0] || !ws.wave_counts[1] || !ws.job_counts || !ws.ray_total || !ws.hits[0] ||
        !ws.hits[1] || !ws.hit_uv_y || !ws.contrib || ws.persistent_blocks == 0)
        return hipErrorInvalidValue;
    for (int q = 0; q < 2; q++)
        for (int k = 0; k < 3; k++)
            if (!ws.queue[q][k]) return hipErrorInvalidValue;
    SegmentPlan& pl = *plan;
    pl.ws = &ws;
    pl.regions = ws.persistent_blocks * (kBlock / 64);
    pl.cap = cap;
    if ((uint64_t)cap * pl.regions > ws.capacity_paths || paths > ws.contrib_paths) return hipErrorInvalidValue;
    // mesh nodes searched ahead of the walk; the linear-scan mode searches nothing ahead
    pl.searched = brute ? 0u : std::min<uint32_t>((uint32_t)target_count, (uint32_t)kMaxDeferred);
    if (pl.searched > ws.result_arrays) return hipErrorInvalidValue;
    for (uint32_t m = 0; m < pl.searched; m++)
        if (!ws.results[m]) return hipErrorInvalidValue;
    // job slots per ray for the children of the instance run (more candidates than that are searched while walking)
    pl.run_slots = (!brute && scene.run.child_count != 0u) ? run_slots_cfg : 0u;
    if (pl.run_slots && (!ws.run_head || !ws.run_res)) return hipErrorInvalidValue;
    if (pl.searched && (!ws.listed[0] || !ws.listed[1])) return hipErrorInvalidValue;
    pl.spill_threads = ws.persistent_blocks * kBlock;
    pl.totals = reinterpret_cast<unsigned long long*>(ws.ray_total);
    memset(&pl.sl, 0, sizeof(pl.sl));
    for (int k = 0; k < kMaxDeferred; k++) pl.sr.res[k] = ws.results[k];
    pl.sr.run_res = ws.run_res;
    if (pl.searched || pl.run_slots) {
        if (!ws.jobs[0] || !ws.jobs[1] || !ws.jobs[2] ||
            (uint64_t)cap * (pl.searched + pl.run_slots) * pl.regions > ws.job_records)
            return hipErrorInvalidValue;
        // the kernels that queue rays list the searches of those rays (every mesh node searched ahead of the walk)
        pl.sl.job_a = ws.jobs[0];
        pl.sl.job_b = ws.jobs[1];
        pl.sl.job_c = ws.jobs[2];
        pl.sl.job_counts = ws.job_counts;
        pl.sl.search_total = pl.totals + 1;
        pl.sl.count = pl.searched;
        pl.sl.run = pl.run_slots ? 1u : 0u;
        pl.sl.run_cap = cap * pl.run_slots;
        pl.sl.job_cap = cap * (pl.searched + pl.run_slots);
        pl.sl.run_head = ws.run_head;
    }
    return hipSuccess;
}

// The grid of a kernel that loops over the regions: as many blocks as the device keeps resident for THAT kernel
// (its registers and LDS decide), never more than the workspace's persistent grid, which is sized for the
// kernel with the most (the search: segment_blocks_per_cu() a compute unit).  Blocks beyond that only queue
// behind the resident ones, and with two batches in flight they queue in front of the other batch's
// launches too: the 871k-triangle 1080p x 64 spp frame takes 82.5 ms with walk_bounce_kernel on the 1024 blocks
// it can keep resident and 85 ms on the search's 2048.
template <class Kernel> uint32_t resident_grid(Kernel kernel, const Workspace& ws) {
    static std::mutex mu;
    static std::vector<std::pair<const void*, int>> known; // blocks per compute unit, by kernel
    const void* key = reinterpret_cast<const void*>(kernel);
    int per_cu = 0;
    {
        std::lock_guard<std::mutex> lock(mu);
        for (const auto& k : known)
            if (k.first == key) per_cu = k.second;
    }
    if (per_cu == 0) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, 0) != hipSuccess || per_cu <= 0) {
            (void)hipGetLastError();
            per_cu = (int)kGridBlocksPerCu;
        }
        std::lock_guard<std::mutex> lock(mu);
        known.emplace_back(key, per_cu);
    }
    const uint32_t cus = std::max(1u, ws.persistent_blocks / kGridBlocksPerCu);
    return std::min(ws.persistent_blocks, cus * (uint32_t)per_cu);
}

void launch_search(const DScene& scene, const SegmentPlan& pl, hipStream_t stream, bool count = false) {
    const Workspace& ws = *pl.ws;
    if (count) {
        unsigned long long* visits = pl.totals + 2;
        if (scene.mesh_count == 1u)
            search_kernel<true, true><<<ws.persistent_blocks, kBlock, 0, stream>>>(
                scene.meshes, scene.mesh0, ws.jobs[0], ws.jobs[1], ws.jobs[2], ws.job_counts, pl.sr, pl.cap, pl.sl.job_cap,
                pl.regions, ws.spill, pl.spill_threads, visits);
        else
            search_kernel<false, true><<<ws.persistent_blocks, kBlock, 0, stream>>>(
                scene.meshes, scene.mesh0, ws.jobs[0], ws.jobs[1], ws.jobs[2], ws.job_counts, pl.sr, pl.cap, pl.sl.job_cap,
                pl.regions, ws.spill, pl.spill_threads, visits);
        return;
    }
    if (scene.mesh_count == 1u)
        search_kernel<true><<<ws.persistent_blocks, kBlock, 0, stream>>>(scene.meshes, scene.mesh0, ws.jobs[0], ws.jobs[1],
                                                                        ws.jobs[2], ws.job_counts, pl.sr, pl.cap, pl.sl.job_cap,
                                                                        pl.regions, ws.spill, pl.spill_threads);
    else
        search_kernel<false><<<ws.persistent_blocks, kBlock, 0, stream>>>(scene.meshes, scene.mesh0, ws.jobs[0], ws.jobs[1],
                                                                         ws.jobs[2], ws.job_counts, pl.sr, pl.cap, pl.sl.job_cap,
                                                                         pl.regions, ws.spill, pl.spill_threads);
}

} // namespace

hipError_t render_batch(const DScene& scene, const DCamera& cam, const BatchGeometry& geo, uint64_t seed,
                        uint64_t sample_first, uint32_t sample_count, uint32_t trace_depth, bool brute,
                        const std::vector<MeshTarget>& targets, uint32_t run_slots_cfg, Workspace& ws, hipStream_t stream,
                        std::vector<hipEvent_t>* search_events, bool count_visits) {
    RenderParams p;
    p.scene = scene;
    p.cam = cam;
    p.geo = geo;
    p.by_npp = make_divisor(geo.npp);
    p.by_nbx = make_divisor(geo.nbx);
    p.seed = seed;
    p.sample_first = sample_first;
    p.sample_count = sample_count;
    p.trace_depth = trace_depth;
    p.brute = brute ? 1u : 0u;
    uint64_t total = (uint64_t)sample_count * geo.npp;
    if (total == 0) return hipSuccess;
    if (total > 0xFFFFFFFFull || ws.persistent_blocks == 0) return hipErrorInvalidValue;
    if (trace_depth == 0) { // every path is black (work_group.rs:26-51 with max_depth 0)
        if (!ws.contrib || total > ws.contrib_paths) return hipErrorInvalidValue;
        return hipMemsetAsync(ws.contrib, 0, total * 3 * sizeof(float), stream);
    }
    SegmentPlan pl;
    hipError_t e = make_plan(scene, targets.size(), run_slots_cfg, brute, ws,
                             queue_region_capacity(geo.nbx, geo.nby, sample_count, ws.persistent_blocks * (kBlock / 64)), total,
                             &pl);
    if (e != hipSuccess) return e;
    const uint32_t cap = pl.cap, regions = pl.regions, spill_threads = pl.spill_threads;
    const SearchList& sl = pl.sl;
    unsigned long long* rt = pl.totals;
    // scenes with an instance run get the kernels that know about it; the others carry none of its code
    const bool with_run = pl.run_slots != 0u;
    if (with_run) camera_kernel<true><<<resident_grid(camera_kernel<true>, ws), kBlock, 0, stream>>>(p, pl.buffers(0), sl, cap, regions);
    else camera_kernel<false><<<resident_grid(camera_kernel<false>, ws), kBlock, 0, stream>>>(p, pl.buffers(0), sl, cap, regions);
    for (uint32_t d = 0; d < trace_depth; d++) {
        SegmentBuffers b = pl.buffers((int)(d & 1u));
        if (pl.searched || pl.run_slots) {
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (search_events) {
                if ((e = hipEventCreate(&e0)) != hipSuccess) return e;
                search_events->push_back(e0);
                if ((e = hipEventCreate(&e1)) != hipSuccess) return e;
                search_events->push_back(e1);
                if ((e = hipEventRecord(e0, stream)) != hipSuccess) return e;
            }
            launch_search(scene, pl, stream, count_visits);
            if (search_events && (e = hipEventRecord(e1, stream)) != hipSuccess) return e;
        }
        // one loop over the root's children where the scene has that shape (not in the linear-scan mode,
        // whose searches happen inside the general walk)
        const bool flat = scene.flat_root != 0u && !brute;
        if (flat && !with_run && scene.material_count <= kWalkBounceMaterials) {
            walk_bounce_kernel<false><<<resident_grid(walk_bounce_kernel<false>, ws), kBlock, 0, stream>>>(p, d, b, sl, cap, regions, ws.contrib, rt,
                                                                                   ws.spill, spill_threads);
            continue;
        }
        if (with_run) {
            if (flat) walk_kernel<true, true><<<resident_grid(walk_kernel<true, true>, ws), kBlock, 0, stream>>>(p, b, cap, regions, rt, ws.spill, spill_threads);
            else walk_kernel<true, false><<<resident_grid(walk_kernel<true, false>, ws), kBlock, 0, stream>>>(p, b, cap, regions, rt, ws.spill, spill_threads);
            bounce_kernel<true><<<resident_grid(bounce_kernel<true>, ws), kBlock, 0, stream>>>(p, d, b, sl, cap, regions, ws.contrib);
        } else {
            if (flat) walk_kernel<false, true><<<resident_grid(walk_kernel<false, true>, ws), kBlock, 0, stream>>>(p, b, cap, regions, rt, ws.spill, spill_threads);
            else walk_kernel<false, false><<<resident_grid(walk_kernel<false, false>, ws), kBlock, 0, stream>>>(p, b, cap, regions, rt, ws.spill, spill_threads);
            bounce_kernel<false><<<resident_grid(bounce_kernel<false>, ws), kBlock, 0, stream>>>(p, d, b, sl, cap, regions, ws.contrib);
        }
    }
    return hipGetLastError();
}

hipError_t accumulate_batch(const BatchGeometry& geo, uint32_t width, uint32_t sample_count, const Workspace& ws,
                            float* d_accum, hipStream_t stream) {
    if (geo.npp == 0 || sample_count == 0) return hipSuccess;
    if (!ws.contrib || (uint64_t)sample_count * geo.npp > ws.contrib_paths) return hipErrorInvalidValue;
    accumulate_kernel<<<blocks_for(geo.npp), kBlock, 0, stream>>>(geo, width, sample_count, ws.contrib, d_accum);
    return hipGetLastError();
}

hipError_t trace_closest_wavefront(const DScene& scene, const std::vector<MeshTarget>& targets, uint32_t run_slots_cfg,
                                   const PtRay* d_rays, size_t n, PtHit* d_hits, bool split, Workspace& ws,
                                   hipStream_t stream) {
    if (n == 0) return hipSuccess;
    if (n > 0x7FFFFFFFull || ws.persistent_blocks == 0) return hipErrorInvalidValue;
    SegmentPlan pl;
    const uint32_t cap = injected_region_capacity(n, ws.persistent_blocks * (kBlock / 64));
    hipError_t e = make_plan(scene, targets.size(), run_slots_cfg, false, ws, cap, n, &pl);
    if (e != hipSuccess) return e;
    // one path segment of a "render" whose path ids are the ray indices; the pixel a path id maps to only
    // selects the random stream of the bounce that walk_bounce_kernel carries along, whose colour nobody reads
    RenderParams p;
    memset(&p, 0, sizeof(p));
    p.scene = scene;
    p.cam.width = 1u;
    p.cam.height = 1u;
    p.geo.w = 8u;
    p.geo.h = 0xFFFFFFF8u;
    p.geo.nbx = 1u;
    p.geo.nby = 0x1FFFFFFFu;
    p.geo.npp = 0xFFFFFFFFu;
    p.by_npp = make_divisor(p.geo.npp);
    p.by_nbx = make_divisor(p.geo.nbx);
    p.sample_count = 1u;
    p.trace_depth = 1u;
    const uint32_t regions = pl.regions, spill_threads = pl.spill_threads;
    const bool with_run = pl.run_slots != 0u;
    SegmentBuffers b = pl.buffers(0);
    if (with_run) inject_kernel<true><<<ws.persistent_blocks, kBlock, 0, stream>>>(scene, d_rays, (uint32_t)n, b, pl.sl, cap, regions);
    else inject_kernel<false><<<ws.persistent_blocks, kBlock, 0, stream>>>(scene, d_rays, (uint32_t)n, b, pl.sl, cap, regions);
    if (pl.searched || pl.run_slots) launch_search(scene, pl, stream);
    const bool flat = scene.flat_root != 0u;
    unsigned long long* rt = pl.totals;
    if (flat && !with_run && !split && scene.material_count <= kWalkBounceMaterials) {
        walk_bounce_kernel<true><<<ws.persistent_blocks, kBlock, 0, stream>>>(p, 0u, b, pl.sl, cap, regions, ws.contrib, rt,
                                                                              ws.spill, spill_threads, d_hits);
    } else if (with_run) {
        if (flat) walk_kernel<true, true, true><<<ws.persistent_blocks, kBlock, 0, stream>>>(p, b, cap, regions, rt, ws.spill, spill_threads, d_hits);
        else walk_kernel<true, false, true><<<ws.persistent_blocks, kBlock, 0, stream>>>(p, b, cap, regions, rt, ws.spill, spill_threads, d_hits);
    } else {
        if (flat) walk_kernel<false, true, true><<<ws.persistent_blocks, kBlock, 0, stream>>>(p, b, cap, regions, rt, ws.spill, spill_threads, d_hits);
        else walk_kernel<false, false, true><<<ws.persistent_blocks, kBlock, 0, stream>>>(p, b, cap, regions, rt, ws.spill, spill_threads, d_hits);
    }
    return hipGetLastError();
}

hipError_t resolve_image(const float* d_accum, uint32_t width, uint32_t height, float scale, float* d_out,
                         hipStream_t stream) {
    uint64_t n = (uint64_t)width * height;
    if (n == 0) return hipSuccess;
    resolve_kernel<<<blocks_for(n), kBlock, 0, stream>>>(d_accum, width, height, scale, d_out);
    return hipGetLastError();
}

static bool aligned16(const void* a, const void* b = nullptr, const void* c = nullptr) {
    return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15u) == 0;
}

hipError_t frame_add(float* d_sum, const float* d_render, uint64_t floats, hipStream_t stream) {
    if (floats == 0) return hipSuccess;
    uint32_t blocks = blocks_for((floats + 3) / 4);
    if (aligned16(d_sum, d_render)) frame_add_kernel<true><<<blocks, kBlock, 0, stream>>>(d_sum, d_render, floats);
    else frame_add_kernel<false><<<blocks, kBlock, 0, stream>>>(d_sum, d_render, floats);
    return hipGetLastError();
}

hipError_t frame_image(const float* d_sum, uint64_t pixels, float divisor, float* d_rgb, uint8_t* d_rgba,
                       hipStream_t stream) {
    if (pixels == 0 || (!d_rgb && !d_rgba)) return hipSuccess;
    uint32_t blocks = blocks_for((pixels + 3) / 4);
    uint32_t* rgba = reinterpret_cast<uint32_t*>(d_rgba);
    if (aligned16(d_sum, d_rgb, d_rgba)) frame_image_kernel<true><<<blocks, kBlock, 0, stream>>>(d_sum, pixels, divisor, d_rgb, rgba);
    else frame_image_kernel<false><<<blocks, kBlock, 0, stream>>>(d_sum, pixels, divisor, d_rgb, rgba);
    return hipGetLastError();
}

hipError_t trace_closest(const DScene& scene, const PtRay* d_rays, size_t n, bool brute, PtHit* d_hits,
                         const Workspace& ws, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    trace_closest_kernel<<<std::min(blocks_for(n), ws.persistent_blocks), kBlock, 0, stream>>>(
        scene, d_rays, n, brute ? 1u : 0u, d_hits, ws.spill, ws.persistent_blocks * kBlock);
    return hipGetLastError();
}

// ---- closest hits without any allocation
// The spill columns (traversal-stack overflow and the saved frames of nested scene-graph
// nodes, see TravStack) of a grid of kStaticTraceBlocks blocks, 19.4 MB, held by every GPU
// the library runs on until the process ends.  Nothing in it has to start at zero: a lane
// writes an entry before it reads it.
constexpr size_t kStaticScratchDwords = (size_t)kStaticTraceBlocks * kBlock * kSpillDwords;
static __device__ __align__(256) uint32_t scratch[kStaticScratchDwords];

namespace {
constexpr int kMaxStaticDevices = 64;
uint32_t* g_scratch_ptr[kMaxStaticDevices]; // per device: where that device's copy of `scratch` lives
std::mutex g_scratch_mu;
} // namespace

size_t static_scratch_bytes() { return kStaticScratchDwords * sizeof(uint32_t); }

hipError_t static_scratch_init(int device) {
    if (device < 0 || device >= kMaxStaticDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (g_scratch_ptr[device]) return hipSuccess;
    uint32_t* ptr = nullptr;
    hipError_t e = hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!ptr) return hipErrorInvalidSymbol;
    g_scratch_ptr[device] = ptr;
    return hipSuccess;
}

hipError_t trace_closest_static(int device, const DScene& scene, const PtRay* d_rays, size_t n, bool brute,
                                PtHit* d_hits, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    if (device < 0 || device >= kMaxStaticDevices) return hipErrorInvalidDevice;
    uint32_t* ptr;
    {
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        ptr = g_scratch_ptr[device];
    }
    if (!ptr) return hipErrorNotInitialized;
    const uint32_t blocks = std::min(blocks_for(n), kStaticTraceBlocks);
    trace_closest_kernel<<<blocks, kBlock, 0, stream>>>(scene, d_rays, n, brute ? 1u : 0u, d_hits, ptr,
                                                        kStaticTraceBlocks * kBlock);
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

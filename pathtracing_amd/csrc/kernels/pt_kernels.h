// Host-callable entry points of the gfx950 kernels.
#pragma once
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>

#include <vector>

#include "../../../include/pathtracing_amd.h"
#include "pt_scene_device.h"

namespace ptk {

// Scratch memory of the wavefront pipeline, owned by the scene handle and grown
// on demand.  Queue records are three 16-byte rows per path (structure of
// float4 arrays, so every lane's load is 16 bytes and a wave's is 1 KiB):
//   a = (source.xyz, ray.min)   b = (direction.xyz, bits(path id))
//   c = (multiplier.rgb, bits(random-stream draw counter))
// A queue is cut into one region per wave of the persistent grid (region w starts
// at w * queue_region_capacity()); wave_counts[q][w] is the number of records in
// region w of queue q.
struct Workspace {
    float4* queue[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    uint32_t* wave_counts[2] = {nullptr, nullptr};
    // BVH search requests of the segment (three rows per request; one region per wave,
    // sized for every ray reaching every mesh node that is searched ahead of the walk)
    // with their per-region counts; job_records = records each array holds.
    // results[m] = closest triangle of the scene's m-th mesh node for each ray, indexed
    // like a queue
    float4* jobs[3] = {nullptr, nullptr, nullptr};
    uint32_t* job_counts = nullptr;
    uint64_t job_records = 0;
    float4* results[16] = {};
    // per queue, per ray: which of the searched mesh nodes the ray has a job (and so a result) for
    uint32_t* listed[2] = {nullptr, nullptr};
    // instance run: per ray the head of its chain of jobs (indexed like a queue), per job its
    // result (indexed like `jobs`)
    uint32_t* run_head = nullptr;
    float4* run_res = nullptr;
    // the segment's hit per ray, indexed like a queue: (point, material | inside << 31) (normal, uv.x), uv.y
    float4* hits[2] = {nullptr, nullptr};
    float* hit_uv_y = nullptr;
    float* contrib = nullptr;    // 3 floats per path of the batch: its final colour
    uint64_t capacity_paths = 0; // records each queue array holds
    uint64_t contrib_paths = 0;  // paths the contribution buffer holds
    uint32_t result_arrays = 0;  // how many of results[] are allocated
    // device-side running sums: [0] traced rays, [1] BVH searches, [2] node records and [3] triangle
    // records the search kernel fetched (counted only on request)
    uint64_t* ray_total = nullptr;
    // Trace kernels run a persistent grid of `persistent_blocks` blocks; `spill`
    // holds one column of spill_dwords() dwords per thread of that grid.
    uint32_t* spill = nullptr;
    uint32_t persistent_blocks = 0;
};
size_t spill_dwords_per_thread();
uint32_t traversal_stack_capacity(); // pending entries a lane's BVH search can hold
uint32_t segment_blocks_per_cu(); // resident blocks per compute unit the trace kernels are built for
// Records per wave region for a batch of `samples` samples of nbx x nby pixel blocks on a grid of `waves` waves.
uint32_t queue_region_capacity(uint32_t nbx, uint32_t nby, uint32_t samples, uint32_t waves);
constexpr uint32_t kTraceBlock = 256;

constexpr int kMaxDeferred = 16; // mesh nodes of a scene whose searches go through the search kernel
static_assert(sizeof(Workspace::results) / sizeof(float4*) == kMaxDeferred, "one result array per searched mesh node");

struct BatchGeometry {
    uint32_t x0, y0, w, h; // pixel rectangle, camera space
    uint32_t nbx, nby;     // 8x8 pixel blocks covering it
    uint32_t npp;          // path slots per sample = nbx * nby * 64
};

// Traces samples [sample_first, sample_first + sample_count) of every pixel of the
// rectangle: every path's final colour ends up in ws.contrib (accumulate_batch adds
// them to the image).
// `targets`: the scene graph's mesh nodes in array order.  `run_slots`: job slots per
// ray set aside for the children of the scene's instance run (0: the scene has none).
// `search_events`: when non-null, a pair of events (start, stop) is created and
// recorded around every search_kernel launch and appended to the vector.
// `count_visits`: the search kernel counts the node and triangle records it fetches.
hipError_t render_batch(const DScene& scene, const DCamera& cam, const BatchGeometry& geo, uint64_t seed,
                        uint64_t sample_first, uint32_t sample_count, uint32_t trace_depth, bool brute,
                        const std::vector<MeshTarget>& targets, uint32_t run_slots, Workspace& ws, hipStream_t stream,
                        std::vector<hipEvent_t>* search_events, bool count_visits = false);

// Adds a traced batch's colours, per pixel in sample order, into d_accum
// (camera-space layout, (x + y*W)*3).
hipError_t accumulate_batch(const BatchGeometry& geo, uint32_t width, uint32_t sample_count, const Workspace& ws,
                            float* d_accum, hipStream_t stream);

// Records per wave region that `n` rays handed in through trace_closest_wavefront need.
uint32_t injected_region_capacity(size_t n, uint32_t waves);

// scene.hierarchy.trace_ray for n rays (ray.max = f32::MAX, as every ray of a render
// has) through the kernels a render uses: the rays go into the wave regions of a
// queue, their BVH searches are listed, search_kernel runs them, the walk reads the
// results.  `split`: walk_kernel even where a render would use walk_bounce_kernel.
hipError_t trace_closest_wavefront(const DScene& scene, const std::vector<MeshTarget>& targets, uint32_t run_slots,
                                   const PtRay* d_rays, size_t n, PtHit* d_hits, bool split, Workspace& ws,
                                   hipStream_t stream);

hipError_t resolve_image(const float* d_accum, uint32_t width, uint32_t height, float scale, float* d_out,
                         hipStream_t stream);

// The client's frame (client/src/frame.rs:54-103) on the device: d_sum += d_render
// over `floats` elements, and the mean image d_sum / divisor with its RGBA8
// preview (either output may be null; d_rgba must be 4-byte aligned).
hipError_t frame_add(float* d_sum, const float* d_render, uint64_t floats, hipStream_t stream);
hipError_t frame_image(const float* d_sum, uint64_t pixels, float divisor, float* d_rgb, uint8_t* d_rgba,
                       hipStream_t stream);

hipError_t trace_closest(const DScene& scene, const PtRay* d_rays, size_t n, bool brute, PtHit* d_hits,
                         const Workspace& ws, hipStream_t stream);

// trace_closest on a fixed grid of kStaticTraceBlocks blocks whose spill columns live in a
// static device array: nothing is allocated, by this call or before it.  Every device
// holds its own copy of the array; static_scratch_init looks its address up on the
// current device (once per device, at set-up), and trace_closest_static fails with
// hipErrorNotInitialized on a device where that has not happened.  All calls on one
// device share the array: calls that overlap there race.
constexpr uint32_t kStaticTraceBlocks = 64;
hipError_t static_scratch_init(int device);
size_t static_scratch_bytes();
hipError_t trace_closest_static(int device, const DScene& scene, const PtRay* d_rays, size_t n, bool brute,
                                PtHit* d_hits, hipStream_t stream);

hipError_t generate_rays(const DCamera& cam, uint64_t seed, const uint32_t* d_pixel_xy, const uint64_t* d_sample,
                         size_t n, PtRay* d_rays, hipStream_t stream);

hipError_t shade(const DScene& scene, const PtVec3* d_dirs, const PtHit* d_hits, const uint64_t* d_stream,
                 uint64_t seed, size_t n, uint32_t* d_kind, PtVec3* d_a, PtVec3* d_dir, hipStream_t stream);

} // namespace ptk

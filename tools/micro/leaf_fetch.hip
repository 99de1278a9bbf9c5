// Micro-benchmark: dependent fetches of a leaf's three triangles at random places of an array, every lane its own leaf:
//   mode 0  48-byte triangles (three aligned 16-byte rows each), leaves at multiples of 48 bytes
//   mode 1  36-byte triangles (nine floats, read as 16 + 16 + 4 bytes at 4-byte alignment), leaves at multiples of 36 bytes
//   mode 2  36-byte triangles, every leaf at the start of a 128-byte line (108 bytes: two 64-byte sectors)
// Build: hipcc --offload-arch=gfx950 -O3 -o leaf_fetch leaf_fetch.hip ; run: ./leaf_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float F4 __attribute__((ext_vector_type(4)));
typedef float F4U __attribute__((ext_vector_type(4), aligned(4)));

template <int kMode>
__global__ void fetch(const float* __restrict__ base, const uint32_t* __restrict__ idx, float* out, int steps, uint32_t mask) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t i = idx[tid] & mask;
    float acc = 0.f;
    for (int s = 0; s < steps; s++) {
        uint32_t next = 0;
        if (kMode == 0) {
            const F4* p = (const F4*)(base + (size_t)i * 12);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                F4 a = p[3 * k], b = p[3 * k + 1], c = p[3 * k + 2];
                acc += a.x + b.y;
                next ^= __float_as_uint(c.z);
            }
        } else {
            const float* q = base + (kMode == 1 ? (size_t)i * 9 : (size_t)i * 32);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                F4U a = *(const F4U*)(q + 9 * k), b = *(const F4U*)(q + 9 * k + 4);
                float c = q[9 * k + 8];
                acc += a.x + b.y;
                next ^= __float_as_uint(c);
            }
        }
        i = (next + tid) & mask;
    }
    out[tid] = acc;
}

int main() {
    const size_t max_bytes = 1ull << 30;
    const int threads = 256 * 2048 * 4, steps = 60;
    std::vector<uint32_t> h(max_bytes / 4);
    uint64_t s = 88172645463325252ull;
    for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)(s >> 16); }
    std::vector<uint32_t> hi(threads);
    for (auto& v : hi) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)(s >> 16); }
    float* d_base; uint32_t* d_idx; float* d_out;
    if (hipMalloc(&d_base, max_bytes + 4096) != hipSuccess || hipMalloc(&d_idx, hi.size() * 4) != hipSuccess ||
        hipMalloc(&d_out, (size_t)threads * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemcpy(d_base, h.data(), max_bytes, hipMemcpyHostToDevice);
    hipMemcpy(d_idx, hi.data(), hi.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // leaf start positions: index i in [0, 2^lr); the three triangles follow it
    const int log_leaves[] = {15, 18, 20, 22};
    const char* names[] = {"48-B triangles", "36-B triangles, packed", "36-B triangles, leaf per line"};
    const int unit[] = {48, 36, 128};
    for (int lr : log_leaves)
        for (int mode = 0; mode < 3; mode++) {
            const uint32_t n = 1u << lr;
            if ((size_t)n * unit[mode] + 512 > max_bytes) continue;
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                if (mode == 0) fetch<0><<<threads / 256, 256>>>(d_base, d_idx, d_out, steps, n - 1);
                if (mode == 1) fetch<1><<<threads / 256, 256>>>(d_base, d_idx, d_out, steps, n - 1);
                if (mode == 2) fetch<2><<<threads / 256, 256>>>(d_base, d_idx, d_out, steps, n - 1);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("2^%d leaf positions, span %7.1f MB, %-32s: %8.3f ms, %6.2f G leaves/s\n", lr, (double)n * unit[mode] / 1e6,
                   names[mode], best, (double)threads * steps / best / 1e6);
            fflush(stdout);
        }
    return 0;
}

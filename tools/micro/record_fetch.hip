// Micro-benchmark: dependent fetches of small records at random places, every lane its own record,
// for record sizes of 2, 3 and 4 sixteen-byte rows and arrays from L2-sized to HBM-sized.
// Build: hipcc --offload-arch=gfx950 -O3 -o record_fetch record_fetch.hip ; run: ./record_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float F4 __attribute__((ext_vector_type(4)));

template <int kRows>
__global__ void fetch(const F4* __restrict__ rows, const uint32_t* __restrict__ idx, float* out, int steps, uint32_t mask) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t i = idx[tid] & mask;
    float acc = 0.f;
    for (int s = 0; s < steps; s++) {
        const F4* p = rows + (size_t)i * kRows;
        F4 r[kRows];
#pragma unroll
        for (int k = 0; k < kRows; k++) r[k] = p[k];
#pragma unroll
        for (int k = 0; k < kRows - 1; k++) acc += r[k].x;
        i = (__float_as_uint(r[kRows - 1].w) + tid) & mask; // dependent: the next record comes from the data
    }
    out[tid] = acc;
}

int main() {
    const size_t max_bytes = 1ull << 30;
    const int threads = 256 * 2048 * 4, steps = 100;
    std::vector<uint32_t> h(max_bytes / 4);
    uint64_t s = 88172645463325252ull;
    for (auto& v : h) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        v = (uint32_t)(s >> 16);
    }
    std::vector<uint32_t> hi(threads);
    for (auto& v : hi) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)(s >> 16); }
    F4* d_rows; uint32_t* d_idx; float* d_out;
    if (hipMalloc(&d_rows, max_bytes) != hipSuccess || hipMalloc(&d_idx, hi.size() * 4) != hipSuccess ||
        hipMalloc(&d_out, (size_t)threads * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemcpy(d_rows, h.data(), max_bytes, hipMemcpyHostToDevice);
    hipMemcpy(d_idx, hi.data(), hi.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int log_records[] = {15, 17, 19, 20, 22, 24}; // records in the array (power of two for the mask)
    for (int lr : log_records)
        for (int rows = 2; rows <= 4; rows++) {
            const uint32_t n = 1u << lr;
            if ((size_t)n * rows * 16 > max_bytes) continue;
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                if (rows == 2) fetch<2><<<threads / 256, 256>>>(d_rows, d_idx, d_out, steps, n - 1);
                if (rows == 3) fetch<3><<<threads / 256, 256>>>(d_rows, d_idx, d_out, steps, n - 1);
                if (rows == 4) fetch<4><<<threads / 256, 256>>>(d_rows, d_idx, d_out, steps, n - 1);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("records 2^%d x %d B = %7.1f MB: %8.3f ms, %7.2f G records/s, %6.2f TB/s of record bytes\n", lr, rows * 16,
                   (double)n * rows * 16 / 1e6, best, (double)threads * steps / best / 1e6,
                   (double)threads * steps * rows * 16 / best / 1e9);
            fflush(stdout);
        }
    return 0;
}

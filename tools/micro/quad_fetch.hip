// Micro-benchmark: 64-byte records at random places, fetched (a) by each lane alone, four 16-byte loads, or
// (b) by the four lanes of a quad together, one 16-byte row each, one record per load instruction.
// Build: hipcc --offload-arch=gfx950 -O3 -o quad_fetch quad_fetch.hip ; run: ./quad_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float F4 __attribute__((ext_vector_type(4)));

__global__ void own(const F4* __restrict__ nodes, const uint32_t* __restrict__ idx, float* out, int steps, uint32_t mask) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t i = idx[tid];
    float acc = 0.f;
    for (int s = 0; s < steps; s++) {
        const F4* p = nodes + (size_t)i * 4;
        F4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += a.x + b.y + c.z;
        i = (__float_as_uint(d.w) + tid) & mask; // dependent: next record from the data
    }
    out[tid] = acc;
}

__global__ void quad(const F4* __restrict__ nodes, const uint32_t* __restrict__ idx, float* out, int steps, uint32_t mask) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t lane4 = tid & 3u;
    uint32_t i = idx[tid];
    float acc = 0.f;
    __shared__ F4 stage[256 * 4];
    F4* mine = stage + (threadIdx.x & ~3u) * 4; // the quad's 4 records x 4 rows
    for (int s = 0; s < steps; s++) {
        // member k's index through LDS-free quad broadcast
        uint32_t i0 = __builtin_amdgcn_mov_dpp(i, 0x00, 0xF, 0xF, true); // quad_perm [0,0,0,0]
        uint32_t i1 = __builtin_amdgcn_mov_dpp(i, 0x55, 0xF, 0xF, true); // [1,1,1,1]
        uint32_t i2 = __builtin_amdgcn_mov_dpp(i, 0xAA, 0xF, 0xF, true); // [2,2,2,2]
        uint32_t i3 = __builtin_amdgcn_mov_dpp(i, 0xFF, 0xF, 0xF, true); // [3,3,3,3]
        F4 r0 = nodes[(size_t)i0 * 4 + lane4];
        F4 r1 = nodes[(size_t)i1 * 4 + lane4];
        F4 r2 = nodes[(size_t)i2 * 4 + lane4];
        F4 r3 = nodes[(size_t)i3 * 4 + lane4];
        mine[0 * 4 + lane4] = r0;
        mine[1 * 4 + lane4] = r1;
        mine[2 * 4 + lane4] = r2;
        mine[3 * 4 + lane4] = r3;
        F4 a = mine[lane4 * 4 + 0], b = mine[lane4 * 4 + 1], c = mine[lane4 * 4 + 2], d = mine[lane4 * 4 + 3];
        acc += a.x + b.y + c.z;
        i = (__float_as_uint(d.w) + tid) & mask;
    }
    out[tid] = acc;
}

int main() {
    const uint32_t n_nodes = 1u << 17; // 8 MB of records: L2-resident like the dragon's nodes
    const int threads = 256 * 2048 * 4, steps = 200;
    std::vector<float> h((size_t)n_nodes * 16);
    srand(1);
    for (size_t k = 0; k < h.size(); k++) {
        uint32_t v = (uint32_t)rand() * 2654435761u;
        h[k] = *reinterpret_cast<float*>(&v);
    }
    std::vector<uint32_t> hi(threads);
    for (auto& v : hi) v = (uint32_t)rand() & (n_nodes - 1);
    F4* d_nodes; uint32_t* d_idx; float* d_out;
    hipMalloc(&d_nodes, h.size() * 4); hipMalloc(&d_idx, hi.size() * 4); hipMalloc(&d_out, (size_t)threads * 4);
    hipMemcpy(d_nodes, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_idx, hi.data(), hi.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; mode++)
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (mode == 0) own<<<threads / 256, 256>>>(d_nodes, d_idx, d_out, steps, n_nodes - 1);
            else quad<<<threads / 256, 256>>>(d_nodes, d_idx, d_out, steps, n_nodes - 1);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s: %.3f ms, %.2f G records/s\n", mode ? "quad" : "own ", ms, (double)threads * steps / ms / 1e6);
        }
    return 0;
}

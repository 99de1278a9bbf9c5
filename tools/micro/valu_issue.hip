// Micro-benchmark: how many wave64 vector instructions one SIMD of gfx950 issues per cycle, by the number of
// waves that share the SIMD.  Every wave runs a long unrolled stream of INDEPENDENT register-only instructions
// (sixteen accumulators, so no instruction waits for the one before it) and reads the shader clock (s_memtime)
// before and after; the hardware id of every wave (HW_REG_HW_ID, HW_REG_XCC_ID) says which SIMD it ran on, so
// the rate is worked out per SIMD from the waves that really shared it: instructions of all its waves divided
// by (last end - first start).  Chip-wide instructions per second come from HIP events around the launch.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip ; run: ./valu_issue
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#define CHECK(x)                                                                              \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

// sixteen independent copies of one instruction; %16 / %17 are read-only operands
#define STREAM16(OP3)                                                                                                  \
    asm volatile(OP3(0) OP3(1) OP3(2) OP3(3) OP3(4) OP3(5) OP3(6) OP3(7) OP3(8) OP3(9) OP3(10) OP3(11) OP3(12) OP3(13) \
                     OP3(14) OP3(15)                                                                                   \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),     \
                   "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]),            \
                   "+v"(a[15])                                                                                         \
                 : "v"(x), "v"(y), "s"(m)                                                                              \
                 : "vcc", "s40", "s41")

#define I_FMA(k) "v_fma_f32 %" #k ", %" #k ", %16, %17\n"
#define I_ADD(k) "v_add_f32 %" #k ", %" #k ", %16\n"
#define I_MUL(k) "v_mul_f32 %" #k ", %" #k ", %16\n"
#define I_CND(k) "v_cndmask_b32 %" #k ", %" #k ", %16, vcc\n"
#define I_CVT(k) "v_cvt_f32_ubyte0 %" #k ", %" #k "\n"
#define I_MAX3(k) "v_max3_f32 %" #k ", %" #k ", %16, %17\n"
#define I_MIN(k) "v_min_f32 %" #k ", %" #k ", %16\n"
#define I_CMP(k) "v_cmp_lt_f32 vcc, %" #k ", %16\n"
#define I_MOV(k) "v_mov_b32 %" #k ", %16\n"
#define I_AND(k) "v_and_b32 %" #k ", %" #k ", %16\n"
#define I_LSHL(k) "v_lshlrev_b32 %" #k ", 1, %" #k "\n"
#define I_ADDU(k) "v_add_u32 %" #k ", %" #k ", %16\n"
#define I_SUB(k) "v_sub_f32 %" #k ", %" #k ", %16\n"
#define I_MAX(k) "v_max_f32 %" #k ", %" #k ", %16\n"
#define I_MIN3(k) "v_min3_f32 %" #k ", %" #k ", %16, %17\n"
#define I_MED3(k) "v_med3_f32 %" #k ", %" #k ", %16, %17\n"
#define I_FMAC(k) "v_fmac_f32 %" #k ", %16, %17\n"
#define I_MULLO(k) "v_mul_lo_u32 %" #k ", %" #k ", %16\n"
#define I_BFE(k) "v_bfe_u32 %" #k ", %" #k ", 8, 8\n"
#define I_CNDS(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %16, %18\n"
#define I_CMPS(k) "v_cmp_lt_f32_e64 s[40:41], %" #k ", %16\n"
#define I_CVTU(k) "v_cvt_f32_u32 %" #k ", %" #k "\n"
#define I_RCP(k) "v_rcp_f32 %" #k ", %" #k "\n"
#define I_FMAABS(k) "v_fma_f32 %" #k ", |%" #k "|, %16, -%17\n"
#define I_CMPCND(k) "v_cmp_lt_f32 vcc, %" #k ", %16\nv_cndmask_b32 %" #k ", %" #k ", %17, vcc\n"
#define I_RDL(k) "v_readlane_b32 s40, %" #k ", 3\n"
#define I_BFI(k) "v_bfi_b32 %" #k ", %16, %" #k ", %17\n"
#define I_ANDOR(k) "v_and_or_b32 %" #k ", %" #k ", %16, %17\n"
#define I_MINU(k) "v_min_u32 %" #k ", %" #k ", %16\n"
#define I_MAXU(k) "v_max_u32 %" #k ", %" #k ", %16\n"
#define I_MIN3U(k) "v_min3_u32 %" #k ", %" #k ", %16, %17\n"
#define I_PERM(k) "v_perm_b32 %" #k ", %" #k ", %16, %17\n"
#define I_LSHLADD(k) "v_lshl_add_u32 %" #k ", %" #k ", 2, %16\n"
// one select in every two / four / eight instructions, the rest multiply-adds
#define I_CND2(k) I_CND2_##k
#define I_CND4(k) I_CND4_##k
#define I_CND8(k) I_CND8_##k
#define CNDI(k) "v_cndmask_b32 %" #k ", %" #k ", %16, vcc\n"
#define FMAI(k) "v_fma_f32 %" #k ", %" #k ", %16, %17\n"
#define I_CND2_0 CNDI(0)
#define I_CND2_1 FMAI(1)
#define I_CND2_2 CNDI(2)
#define I_CND2_3 FMAI(3)
#define I_CND2_4 CNDI(4)
#define I_CND2_5 FMAI(5)
#define I_CND2_6 CNDI(6)
#define I_CND2_7 FMAI(7)
#define I_CND2_8 CNDI(8)
#define I_CND2_9 FMAI(9)
#define I_CND2_10 CNDI(10)
#define I_CND2_11 FMAI(11)
#define I_CND2_12 CNDI(12)
#define I_CND2_13 FMAI(13)
#define I_CND2_14 CNDI(14)
#define I_CND2_15 FMAI(15)
#define I_CND4_0 CNDI(0)
#define I_CND4_1 FMAI(1)
#define I_CND4_2 FMAI(2)
#define I_CND4_3 FMAI(3)
#define I_CND4_4 CNDI(4)
#define I_CND4_5 FMAI(5)
#define I_CND4_6 FMAI(6)
#define I_CND4_7 FMAI(7)
#define I_CND4_8 CNDI(8)
#define I_CND4_9 FMAI(9)
#define I_CND4_10 FMAI(10)
#define I_CND4_11 FMAI(11)
#define I_CND4_12 CNDI(12)
#define I_CND4_13 FMAI(13)
#define I_CND4_14 FMAI(14)
#define I_CND4_15 FMAI(15)
#define I_CND8_0 CNDI(0)
#define I_CND8_1 FMAI(1)
#define I_CND8_2 FMAI(2)
#define I_CND8_3 FMAI(3)
#define I_CND8_4 FMAI(4)
#define I_CND8_5 FMAI(5)
#define I_CND8_6 FMAI(6)
#define I_CND8_7 FMAI(7)
#define I_CND8_8 CNDI(8)
#define I_CND8_9 FMAI(9)
#define I_CND8_10 FMAI(10)
#define I_CND8_11 FMAI(11)
#define I_CND8_12 FMAI(12)
#define I_CND8_13 FMAI(13)
#define I_CND8_14 FMAI(14)
#define I_CND8_15 FMAI(15)
// the proportions of a node step of the search kernel: byte decodes, multiply-adds, three-way min / max, selects
#define I_MIX(k) I_MIX_##k
#define I_MIX_0 "v_cvt_f32_ubyte0 %0, %0\n"
#define I_MIX_1 "v_cvt_f32_ubyte1 %1, %1\n"
#define I_MIX_2 "v_cvt_f32_ubyte2 %2, %2\n"
#define I_MIX_3 "v_cvt_f32_ubyte3 %3, %3\n"
#define I_MIX_4 "v_fma_f32 %4, %4, %16, %17\n"
#define I_MIX_5 "v_fma_f32 %5, %5, %16, %17\n"
#define I_MIX_6 "v_fma_f32 %6, %6, %16, %17\n"
#define I_MIX_7 "v_fma_f32 %7, %7, %16, %17\n"
#define I_MIX_8 "v_fma_f32 %8, %8, %16, %17\n"
#define I_MIX_9 "v_fma_f32 %9, %9, %16, %17\n"
#define I_MIX_10 "v_max3_f32 %10, %10, %16, %17\n"
#define I_MIX_11 "v_min3_f32 %11, %11, %16, %17\n"
#define I_MIX_12 "v_cmp_lt_f32 vcc, %12, %16\n"
#define I_MIX_13 "v_cndmask_b32 %13, %13, %16, vcc\n"
#define I_MIX_14 "v_cndmask_b32 %14, %14, %17, vcc\n"
#define I_MIX_15 "v_add_f32 %15, %15, %16\n"

struct WaveRecord {
    unsigned long long t0, t1;
    uint32_t hw_id, xcc_id;
};

enum Kind { kFma, kAdd, kMul, kCnd, kCvt, kMax3, kMin, kCmp, kMix, kPkFma, kChain, kMov, kAnd, kLshl, kAddU, kSub, kMax, kMin3, kMed3, kFmac, kMulLo, kBfe, kCndS, kCmpS, kCvtU, kRcp, kFmaAbs, kCmpCnd, kRdl, kBfi, kAndOr, kMinU, kMaxU, kMin3U, kPerm, kLshlAdd, kCnd2, kCnd4, kCnd8, kKinds };
static const char* kind_name[kKinds] = {"v_fma_f32",        "v_add_f32",  "v_mul_f32", "v_cndmask_b32",
                                        "v_cvt_f32_ubyte0", "v_max3_f32", "v_min_f32", "v_cmp_lt_f32",
                                        "node-step mix",    "v_pk_fma_f32", "v_fma_f32 dependent chain",
                                        "v_mov_b32", "v_and_b32", "v_lshlrev_b32", "v_add_u32", "v_sub_f32", "v_max_f32", "v_min3_f32",
                                        "v_med3_f32", "v_fmac_f32", "v_mul_lo_u32", "v_bfe_u32", "v_cndmask_b32 (sgpr pair)",
                                        "v_cmp_lt_f32 (sgpr pair)", "v_cvt_f32_u32", "v_rcp_f32", "v_fma_f32 |a| -c", "v_cmp + v_cndmask (x2)",
                                        "v_readlane_b32", "v_bfi_b32", "v_and_or_b32", "v_min_u32", "v_max_u32",
                                        "v_min3_u32", "v_perm_b32", "v_lshl_add_u32", "1 v_cndmask : 1 v_fma", "1 v_cndmask : 3 v_fma",
                                        "1 v_cndmask : 7 v_fma"};

constexpr int kUnroll = 8; // blocks of 16 instructions per loop pass

template <int K> __global__ void __launch_bounds__(256) stream_kernel(WaveRecord* rec, float* sink, int passes, float x, float y, unsigned long long m) {
    float a[16];
    for (int k = 0; k < 16; k++) a[k] = (float)(threadIdx.x + k);
    typedef float F2 __attribute__((ext_vector_type(2)));
    F2 p[8], px = {x, y}, py = {y, x};
    for (int k = 0; k < 8; k++) p[k] = F2{(float)threadIdx.x, (float)k};
    asm volatile("s_mov_b64 vcc, %0" ::"s"(m) : "vcc");
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < passes; it++) {
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
            if (K == kFma) STREAM16(I_FMA);
            if (K == kAdd) STREAM16(I_ADD);
            if (K == kMul) STREAM16(I_MUL);
            if (K == kCnd) STREAM16(I_CND);
            if (K == kCvt) STREAM16(I_CVT);
            if (K == kMax3) STREAM16(I_MAX3);
            if (K == kMin) STREAM16(I_MIN);
            if (K == kCmp) STREAM16(I_CMP);
            if (K == kMix) STREAM16(I_MIX);
            if (K == kMov) STREAM16(I_MOV);
            if (K == kAnd) STREAM16(I_AND);
            if (K == kLshl) STREAM16(I_LSHL);
            if (K == kAddU) STREAM16(I_ADDU);
            if (K == kSub) STREAM16(I_SUB);
            if (K == kMax) STREAM16(I_MAX);
            if (K == kMin3) STREAM16(I_MIN3);
            if (K == kMed3) STREAM16(I_MED3);
            if (K == kFmac) STREAM16(I_FMAC);
            if (K == kMulLo) STREAM16(I_MULLO);
            if (K == kBfe) STREAM16(I_BFE);
            if (K == kCndS) STREAM16(I_CNDS);
            if (K == kCmpS) STREAM16(I_CMPS);
            if (K == kCvtU) STREAM16(I_CVTU);
            if (K == kRcp) STREAM16(I_RCP);
            if (K == kFmaAbs) STREAM16(I_FMAABS);
            if (K == kCmpCnd) STREAM16(I_CMPCND);
            if (K == kRdl) STREAM16(I_RDL);
            if (K == kBfi) STREAM16(I_BFI);
            if (K == kAndOr) STREAM16(I_ANDOR);
            if (K == kMinU) STREAM16(I_MINU);
            if (K == kMaxU) STREAM16(I_MAXU);
            if (K == kMin3U) STREAM16(I_MIN3U);
            if (K == kPerm) STREAM16(I_PERM);
            if (K == kLshlAdd) STREAM16(I_LSHLADD);
            if (K == kCnd2) STREAM16(I_CND2);
            if (K == kCnd4) STREAM16(I_CND4);
            if (K == kCnd8) STREAM16(I_CND8);
            if (K == kPkFma) {
                // sixteen packed multiply-adds (two f32 operations per lane each) on eight register pairs
                for (int r = 0; r < 2; r++)
                    asm volatile("v_pk_fma_f32 %0, %0, %8, %9\nv_pk_fma_f32 %1, %1, %8, %9\nv_pk_fma_f32 %2, %2, %8, %9\n"
                                 "v_pk_fma_f32 %3, %3, %8, %9\nv_pk_fma_f32 %4, %4, %8, %9\nv_pk_fma_f32 %5, %5, %8, %9\n"
                                 "v_pk_fma_f32 %6, %6, %8, %9\nv_pk_fma_f32 %7, %7, %8, %9\n"
                                 : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7])
                                 : "v"(px), "v"(py));
            }
            if (K == kChain) {
                // every instruction reads the result of the one before it
                asm volatile("v_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\n"
                             "v_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\n"
                             "v_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\n"
                             "v_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\n"
                             : "+v"(a[0])
                             : "v"(x), "v"(y));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int k = 0; k < 16; k++) s += a[k];
    for (int k = 0; k < 8; k++) s += p[k].x + p[k].y;
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    sink[tid] = s;
    if ((threadIdx.x & 63u) == 0u) {
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        WaveRecord r;
        r.t0 = t0;
        r.t1 = t1;
        r.hw_id = hw;
        r.xcc_id = xcc;
        rec[tid >> 6] = r;
    }
}

typedef void (*KernelFn)(WaveRecord*, float*, int, float, float, unsigned long long);
static KernelFn kernels[kKinds] = {stream_kernel<kFma>, stream_kernel<kAdd>,  stream_kernel<kMul>, stream_kernel<kCnd>,
                                   stream_kernel<kCvt>, stream_kernel<kMax3>, stream_kernel<kMin>, stream_kernel<kCmp>,
                                   stream_kernel<kMix>, stream_kernel<kPkFma>, stream_kernel<kChain>, stream_kernel<kMov>,
                                   stream_kernel<kAnd>, stream_kernel<kLshl>, stream_kernel<kAddU>, stream_kernel<kSub>,
                                   stream_kernel<kMax>, stream_kernel<kMin3>, stream_kernel<kMed3>, stream_kernel<kFmac>,
                                   stream_kernel<kMulLo>, stream_kernel<kBfe>, stream_kernel<kCndS>, stream_kernel<kCmpS>,
                                   stream_kernel<kCvtU>, stream_kernel<kRcp>, stream_kernel<kFmaAbs>, stream_kernel<kCmpCnd>,
                                   stream_kernel<kRdl>, stream_kernel<kBfi>,
                                   stream_kernel<kAndOr>, stream_kernel<kMinU>, stream_kernel<kMaxU>, stream_kernel<kMin3U>,
                                   stream_kernel<kPerm>, stream_kernel<kLshlAdd>, stream_kernel<kCnd2>, stream_kernel<kCnd4>,
                                   stream_kernel<kCnd8>};

int main(int argc, char** argv) {
    const int first_kind = argc > 1 ? atoi(argv[1]) : 0, last_kind = argc > 2 ? atoi(argv[2]) : kKinds - 1;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device: %s, %d CUs, %d SIMDs, nominal clock %.0f MHz\n", prop.gcnArchName, cus, cus * 4, prop.clockRate / 1000.0);
    printf("one wave-instruction = one wave64 vector instruction; rates per SIMD from s_memtime and the waves' hardware ids\n");
    const int passes = 4000;
    const double per_wave_base = (double)passes * kUnroll * 16;
    const int max_waves = cus * 4 * 8;
    WaveRecord* d_rec;
    float* d_sink;
    CHECK(hipMalloc(&d_rec, sizeof(WaveRecord) * max_waves));
    CHECK(hipMalloc(&d_sink, sizeof(float) * 64 * max_waves));
    std::vector<WaveRecord> rec(max_waves);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("%-26s %5s | %9s %9s %9s | %10s %8s | %s\n", "instruction", "waves", "cyc/inst", "cyc/inst", "cyc/inst", "chip G/s",
           "GHz", "waves per SIMD seen (count of SIMDs)");
    printf("%-26s %5s | %9s %9s %9s | %10s %8s |\n", "", "/SIMD", "median", "best", "worst", "(events)", "(implied)");
    for (int kind = first_kind; kind <= last_kind && kind < kKinds; kind++) {
        const double per_wave = per_wave_base * (kind == kCmpCnd ? 2 : 1); // that stream has two instructions per slot
        const int ws[4] = {1, 2, 4, 8};
        for (int wi = 0; wi < 4; wi++) {
            const int w = ws[wi];
            const int blocks = cus * w; // 256 threads = four waves, one per SIMD of a CU
            const int waves = blocks * 4;
            float ms = 0.f;
            for (int rep = 0; rep < 2; rep++) { // the first launch of each shape warms up
                CHECK(hipEventRecord(e0));
                kernels[kind]<<<blocks, 256>>>(d_rec, d_sink, passes, 1.0000001f, 1e-9f, 0x5555AAAA3333CCCCull);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            CHECK(hipMemcpy(rec.data(), d_rec, sizeof(WaveRecord) * waves, hipMemcpyDeviceToHost));
            // group the waves by the SIMD they ran on: xcc | se | sh | cu | simd
            struct Simd {
                unsigned long long first = ~0ull, last = 0;
                int n = 0;
            };
            std::map<uint32_t, Simd> simds;
            for (int i = 0; i < waves; i++) {
                uint32_t hw = rec[i].hw_id;
                uint32_t key = ((rec[i].xcc_id & 0xFu) << 16) | (((hw >> 13) & 7u) << 12) | (((hw >> 12) & 1u) << 11) |
                               (((hw >> 8) & 0xFu) << 4) | ((hw >> 4) & 3u);
                Simd& s = simds[key];
                s.first = std::min(s.first, rec[i].t0);
                s.last = std::max(s.last, rec[i].t1);
                s.n++;
            }
            std::vector<double> cpi; // cycles per wave-instruction on SIMDs that held exactly w waves
            std::map<int, int> histogram;
            double cycles_sum = 0;
            for (auto& kv : simds) {
                histogram[kv.second.n]++;
                if (kv.second.n != w) continue;
                double cyc = (double)(kv.second.last - kv.second.first);
                cpi.push_back(cyc / (per_wave * w));
                cycles_sum += cyc;
            }
            double med = 0, best = 0, worst = 0, ghz = 0;
            if (!cpi.empty()) {
                std::sort(cpi.begin(), cpi.end());
                med = cpi[cpi.size() / 2];
                best = cpi.front();
                worst = cpi.back();
                ghz = cycles_sum / cpi.size() / (ms * 1e-3) / 1e9; // a SIMD's busy span against the launch's wall time
            }
            printf("%-26s %5d | %9.3f %9.3f %9.3f | %10.1f %8.2f |", kind_name[kind], w, med, best, worst,
                   per_wave * waves / (ms * 1e-3) / 1e9, ghz);
            for (auto& h : histogram) printf(" %d:%d", h.first, h.second);
            printf("\n");
            fflush(stdout);
        }
    }
    printf("(v_pk_fma_f32 does two multiply-adds per lane; the dependent chain shows the latency of one instruction)\n");
    return 0;
}

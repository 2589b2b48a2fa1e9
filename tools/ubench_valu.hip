// Measures the issue rate of the VALU instructions the 64-bit Shoup butterfly is made of, on gfx950.
// For each instruction: 8 independent dependency chains per lane, W waves per SIMD (1, 2, 4), all 256
// CUs busy; reports shader cycles per wave-instruction per SIMD (s_memtime) and the implied chip-wide
// rate.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CHAINS 8
#define UNROLL 4

enum { OP_MAD64 = 0, OP_MULLO, OP_MULHI, OP_ADD32, OP_ADD64, OP_FMA64, OP_MUL24, OP_MADU24, OP_CNDMASK, OP_FMA32, NOPS };
static const char *NAMES[NOPS] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_add_u32", "v_lshl_add_u64",
                                  "v_fma_f64", "v_mul_u32_u24", "v_mad_u32_u24", "v_cndmask_b32", "v_fma_f32"};

template <int OP> __global__ void bench(uint64_t *sink, uint64_t *cycles, int iters) {
    uint32_t a = threadIdx.x * 2654435761u + 12345u, b = blockIdx.x * 40503u + 977u;
    uint64_t acc[CHAINS];
    uint32_t acc32[CHAINS];
    double accd[CHAINS];
    float accf[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
        acc[c] = a + c;
        acc32[c] = b + c;
        accd[c] = 1.0 + c;
        accf[c] = 1.0f + c;
    }
    double da = 1.0000001, db = 0.9999999;
    float fa = 1.0000001f, fb = 0.9999999f;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if constexpr (OP == OP_MAD64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc");
                if constexpr (OP == OP_MULLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(acc32[c]) : "v"(a));
                if constexpr (OP == OP_MULHI) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(acc32[c]) : "v"(a));
                if constexpr (OP == OP_ADD32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc32[c]) : "v"(a));
                if constexpr (OP == OP_ADD64) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[c]) : "v"(acc[(c + 1) % CHAINS]));
                if constexpr (OP == OP_FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(accd[c]) : "v"(da), "v"(db));
                if constexpr (OP == OP_MUL24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(acc32[c]) : "v"(a));
                if constexpr (OP == OP_MADU24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(acc32[c]) : "v"(a), "v"(b));
                if constexpr (OP == OP_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc32[c]) : "v"(a));
                if constexpr (OP == OP_FMA32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(accf[c]) : "v"(fa), "v"(fb));
            }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint64_t s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += acc[c] + acc32[c] + (uint64_t)accd[c] + (uint64_t)accf[c];
    sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int OP> static void run(uint64_t *sink, uint64_t *cyc, int iters) {
    for (int wps : {1, 2, 4}) {
        const int threads = 256 * wps, blocks = 256;  // one block per CU, wps waves per SIMD
        const int nwaves = blocks * threads / 64;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(threads), 0, 0, sink, cyc, 16);  // warm-up
        hipEventRecord(e0);
        hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(threads), 0, 0, sink, cyc, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<uint64_t> h(nwaves);
        hipMemcpy(h.data(), cyc, nwaves * sizeof(uint64_t), hipMemcpyDeviceToHost);
        double avg = 0;
        for (auto v : h) avg += (double)v;
        avg /= nwaves;
        const double instr_per_wave = (double)iters * UNROLL * CHAINS;
        const double cyc_per_instr_simd = avg / (instr_per_wave * wps);  // SIMD-level: wps waves interleave
        const double chip_rate = (double)nwaves * instr_per_wave * 64 / (ms * 1e-3) / 1e12;
        printf("%-16s waves/SIMD=%d  cycles/wave-instr (per SIMD)=%6.2f  wall=%7.3f ms  chip lane-ops=%6.2f T/s\n",
               NAMES[OP], wps, cyc_per_instr_simd, ms, chip_rate);
        hipEventDestroy(e0);
        hipEventDestroy(e1);
    }
}

int main() {
    uint64_t *sink, *cyc;
    hipMalloc(&sink, 256 * 1024 * sizeof(uint64_t));
    hipMalloc(&cyc, 256 * 16 * sizeof(uint64_t));
    const int iters = 4096;
    run<OP_MAD64>(sink, cyc, iters);
    run<OP_MULLO>(sink, cyc, iters);
    run<OP_MULHI>(sink, cyc, iters);
    run<OP_ADD32>(sink, cyc, iters);
    run<OP_ADD64>(sink, cyc, iters);
    run<OP_CNDMASK>(sink, cyc, iters);
    run<OP_MUL24>(sink, cyc, iters);
    run<OP_MADU24>(sink, cyc, iters);
    run<OP_FMA32>(sink, cyc, iters);
    run<OP_FMA64>(sink, cyc, iters);
    hipFree(sink);
    hipFree(cyc);
    return 0;
}

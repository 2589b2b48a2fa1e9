// Issue cost of a few VALU forms against OCCUPANCY (1 ... 8 waves per SIMD), 8 independent chains per lane, shader-cycle stamps:
// what one SIMD sustains (cycles per wave-instruction) and what ONE wave sustains (its own issue interval).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench5.hip -o tools/ubench5
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CHAINS 8
#define UNROLL 8
template <int OP> __global__ __launch_bounds__(256) void bench(uint64_t *sink, uint64_t *cycles, int iters, uint32_t sarg) {
    extern __shared__ uint32_t pad[];
    uint32_t a = threadIdx.x * 2654435761u + 12345u, b = blockIdx.x * 40503u + 977u;
    uint64_t acc[CHAINS]; uint32_t acc32[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) { acc[c] = 0x3ff0000000000000ull + a + c; acc32[c] = b + c; }
    uint64_t w64 = 0x3ff0000000000001ull + a;
    uint32_t s = __builtin_amdgcn_readfirstlane(sarg);
    if (sarg == 0xdeadbeef) pad[threadIdx.x] = a;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if constexpr (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc32[c]) : "v"(a));
                if constexpr (OP == 1) asm volatile("v_add_u32 %0, %1, %0" : "+v"(acc32[c]) : "s"(s));
                if constexpr (OP == 2) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc");
                if constexpr (OP == 3) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[c]) : "v"(w64));
                if constexpr (OP == 4) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(acc32[c]) : "v"(a));
                if constexpr (OP == 5) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc32[c]) : "v"(a), "v"(b));
                if constexpr (OP == 6) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(acc[c]) : "v"(w64));
                if constexpr (OP == 7) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_add_u32 %3, %3, %1" : "+v"(acc[c]), "+v"(acc32[c]) : "v"(a), "v"(b) : "vcc");
                if constexpr (OP == 8) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %2, %1, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc");   // dependent pair
                if constexpr (OP == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc32[c]) : "v"(a) : );
                if constexpr (OP == 10) asm volatile("v_sub_co_u32 %0, vcc, %0, %1\n\ts_nop 1\n\tv_subb_co_u32 %2, vcc, %2, %1, vcc" : "+v"(acc32[c]), "+v"(a) : "v"(b) : "vcc");
            }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint64_t r = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) r += acc[c] + acc32[c];
    sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r + a;
    if ((threadIdx.x & 63) == 0) cycles[((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}
static const char *NAMES[] = {"v_add_u32 v,v,v", "v_add_u32 v,s,v", "v_mad_u64_u32 vvv (indep)", "v_lshl_add_u64", "v_mul_hi_u32", "v_bitop3_b32",
                              "v_fma_f64", "mad64 ; add vvv (per instr)", "mad64 -> mad64 dependent pair (per instr)", "v_cndmask_b32 vcc",
                              "sub_co ; nop ; subb_co (per instr, 2)"};
static const int PER[] = {1, 1, 1, 1, 1, 1, 1, 2, 2, 1, 2};
template <int OP> static void run(uint64_t *sink, uint64_t *cyc) {
    (void)hipFuncSetAttribute((const void *)bench<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int w : {1, 2, 3, 4, 6, 8}) {
        const int blocks = 256 * w, threads = 256;
        const size_t lds = (size_t)(160 * 1024 / w) - 512;   // exactly w workgroups fit a CU
        const int nwaves = blocks * threads / 64, iters = 1024;
        for (int k = 0; k < 30; ++k) hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(threads), lds, 0, sink, cyc, iters, 12345u);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        const int reps = 30;
        for (int k = 0; k < reps; ++k) hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(threads), lds, 0, sink, cyc, iters, 12345u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        std::vector<uint64_t> h(nwaves);
        hipMemcpy(h.data(), cyc, nwaves * sizeof(uint64_t), hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += (double)v; avg /= nwaves;
        const double ipw = (double)iters * UNROLL * CHAINS * PER[OP];
        printf("%-44s waves/SIMD=%d  one wave issues every %6.2f cyc   SIMD: %5.2f cyc/instr   chip %6.2f T lane-ops/s  %s\n", NAMES[OP], w, avg / ipw,
               avg / (ipw * w), (double)nwaves * ipw * 64 / (ms * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
        fflush(stdout);
    }
}
int main() {
    uint64_t *sink, *cyc;
    hipMalloc(&sink, (size_t)256 * 8 * 256 * sizeof(uint64_t));
    hipMalloc(&cyc, (size_t)256 * 8 * 4 * sizeof(uint64_t));
    run<0>(sink, cyc); run<1>(sink, cyc); run<2>(sink, cyc); run<3>(sink, cyc); run<4>(sink, cyc); run<5>(sink, cyc); run<6>(sink, cyc);
    run<7>(sink, cyc); run<8>(sink, cyc); run<9>(sink, cyc); run<10>(sink, cyc);
    return 0;
}

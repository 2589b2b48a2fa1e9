# generates tools/ubench3.hip: issue rate of many VALU instruction forms at 4 waves/SIMD (and 2)
ops = [
 # name, asm, kind (32: acc32 chain; 64: acc64 chain; d: double chain)
 ("v_add_u32 v,v,v", "v_add_u32 %0, %0, %1", 32),
 ("v_add_u32 v,s,v", "v_add_u32 %0, %3, %0", 32),
 ("v_add_u32 v,imm,v", "v_add_u32 %0, 17, %0", 32),
 ("v_add_u32 v,lit,v", "v_add_u32 %0, 0x12345678, %0", 32),
 ("v_sub_u32 v,v,v", "v_sub_u32 %0, %0, %1", 32),
 ("v_subrev_u32 v,s,v", "v_subrev_u32 %0, %3, %0", 32),
 ("v_subrev_u32 v,v,v", "v_subrev_u32 %0, %1, %0", 32),
 ("v_min_u32 v,v,v", "v_min_u32 %0, %0, %1", 32),
 ("v_max_u32 v,v,v", "v_max_u32 %0, %0, %1", 32),
 ("v_and_b32 v,v,v", "v_and_b32 %0, %0, %1", 32),
 ("v_or_b32 v,v,v", "v_or_b32 %0, %0, %1", 32),
 ("v_xor_b32 v,v,v", "v_xor_b32 %0, %0, %1", 32),
 ("v_lshlrev_b32 v,imm,v", "v_lshlrev_b32 %0, 3, %0", 32),
 ("v_lshrrev_b32 v,imm,v", "v_lshrrev_b32 %0, 3, %0", 32),
 ("v_mov_b32 v,v", "v_mov_b32 %0, %1", 32),
 ("v_cndmask_b32 e32 v,v,vcc", "v_cndmask_b32 %0, %0, %1, vcc", 32),
 ("v_add3_u32", "v_add3_u32 %0, %0, %1, %2", 32),
 ("v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 1, %1", 32),
 ("v_lshl_add_u32 s", "v_lshl_add_u32 %0, %0, 1, %3", 32),
 ("v_and_or_b32", "v_and_or_b32 %0, %0, %1, %2", 32),
 ("v_bitop3_b32", "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96", 32),
 ("v_alignbit_b32", "v_alignbit_b32 %0, %0, %1, 28", 32),
 ("v_bfe_u32", "v_bfe_u32 %0, %0, 3, 9", 32),
 ("v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %1", 32),
 ("v_mul_hi_u32", "v_mul_hi_u32 %0, %0, %1", 32),
 ("v_mul_hi_u32 s", "v_mul_hi_u32 %0, %0, %3", 32),
 ("v_mul_u32_u24", "v_mul_u32_u24 %0, %0, %1", 32),
 ("v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %1, %2", 32),
 ("v_mul_hi_u32_u24", "v_mul_hi_u32_u24 %0, %0, %1", 32),
 ("v_mad_u64_u32 vvv", "v_mad_u64_u32 %0, vcc, %1, %2, %0", 64),
 ("v_mad_u64_u32 v,s,v", "v_mad_u64_u32 %0, vcc, %1, %3, %0", 64),
 ("v_mad_u64_u32 v,v,0", "v_mad_u64_u32 %0, vcc, %1, %2, 0", 64),
 ("v_mad_i64_i32", "v_mad_i64_i32 %0, vcc, %1, %2, %0", 64),
 ("v_lshl_add_u64", "v_lshl_add_u64 %0, %0, 0, %4", 64),
 ("v_add_co_u32", "v_add_co_u32 %0, vcc, %0, %1", 32),
 ("v_addc_co_u32", "v_addc_co_u32 %0, vcc, %0, %1, vcc", 32),
 ("v_cmp_lt_u32", "v_cmp_lt_u32 vcc, %0, %1", 32),
 ("v_pk_add_u16", "v_pk_add_u16 %0, %0, %1", 32),
 ("v_pk_mul_lo_u16", "v_pk_mul_lo_u16 %0, %0, %1", 32),
 ("v_pk_mad_u16", "v_pk_mad_u16 %0, %0, %1, %2", 32),
 ("v_pk_add_f32", "v_pk_add_f32 %0, %0, %4", 64),
 ("v_pk_fma_f32", "v_pk_fma_f32 %0, %0, %4, %4", 64),
 ("v_pk_mul_f32", "v_pk_mul_f32 %0, %0, %4", 64),
 ("v_fma_f32", "v_fma_f32 %0, %0, %1, %2", 32),
 ("v_add_f32", "v_add_f32 %0, %0, %1", 32),
 ("v_mul_f32", "v_mul_f32 %0, %0, %1", 32),
 ("v_fma_f64", "v_fma_f64 %0, %0, %4, %4", 64),
 ("v_add_f64", "v_add_f64 %0, %0, %4", 64),
 ("v_mul_f64", "v_mul_f64 %0, %0, %4", 64),
 ("v_cvt_f64_u32", "v_cvt_f64_u32 %0, %1", 64),
 ("v_cvt_f32_u32", "v_cvt_f32_u32 %0, %0", 32),
 ("v_cvt_u32_f32", "v_cvt_u32_f32 %0, %0", 32),
 ("v_rndne_f64", "v_rndne_f64 %0, %0", 64),
 ("v_mul_legacy_f32", "v_mul_legacy_f32 %0, %0, %1", 32),
 ("v_dot4_u32_u8", "v_dot4_u32_u8 %0, %1, %2, %0", 32),
 ("v_sad_u32", "v_sad_u32 %0, %0, %1, %2", 32),
 ("v_perm_b32", "v_perm_b32 %0, %0, %1, %2", 32),
 ("v_mov_b32 dpp", "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", 32),
 ("v_add_u32 dpp", "v_add_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", 32),
 ("v_mov_b64", "v_mov_b64 %0, %4", 64),
 ("v_lshlrev_b64", "v_lshlrev_b64 %0, 3, %0", 64),
]
import sys
if len(sys.argv) > 1 and sys.argv[1] == "mixed":   # mixed streams: do the double-rate forms keep their rate next to multiplies?
    ops = [
     ("mad64 ; add vvv (2 instr)", "v_mad_u64_u32 %5, vcc, %1, %2, %5\n\tv_add_u32 %0, %0, %1", 32),
     ("mad64 ; add svv (2 instr)", "v_mad_u64_u32 %5, vcc, %1, %2, %5\n\tv_add_u32 %0, %3, %0", 32),
     ("mulhi ; add vvv (2 instr)", "v_mul_hi_u32 %6, %6, %1\n\tv_add_u32 %0, %0, %1", 32),
     ("mulhi ; add svv (2 instr)", "v_mul_hi_u32 %6, %6, %1\n\tv_add_u32 %0, %3, %0", 32),
     ("mulhi ; add vvv ; sub vvv (3 instr)", "v_mul_hi_u32 %6, %6, %1\n\tv_add_u32 %0, %0, %1\n\tv_sub_u32 %0, %0, %2", 32),
     ("min ; sub vvv (2 instr)", "v_min_u32 %6, %6, %1\n\tv_sub_u32 %0, %0, %1", 32),
     ("min ; subrev svv (2 instr)", "v_min_u32 %6, %6, %1\n\tv_subrev_u32 %0, %3, %0", 32),
     ("add vvv ; add vvv (2 instr)", "v_add_u32 %6, %6, %1\n\tv_add_u32 %0, %0, %1", 32),
     ("cmp_lt_u64 e32 vcc ; 2 cndmask e32 vcc (3 instr)", "v_cmp_lt_u64_e32 vcc, %5, %4\n\tv_cndmask_b32_e32 %0, %0, %1, vcc\n\tv_cndmask_b32_e32 %6, %6, %2, vcc", 32),
     ("cmp_lt_u64 e64 s[] ; 2 cndmask e64 s[] (3 instr + nop)", "v_cmp_lt_u64_e64 s[10:11], %5, %4\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %0, %1, s[10:11]\n\tv_cndmask_b32_e64 %6, %6, %2, s[10:11]", 32),
     ("u32 fwd bfly sgpr 2p (7 instr)", "v_subrev_u32 %7, %3, %0\n\tv_min_u32 %0, %0, %7\n\tv_mul_hi_u32 %7, %6, %2\n\tv_mad_u64_u32 %5, vcc, %6, %1, %5\n\tv_mad_u64_u32 %5, vcc, %7, %3, %5\n\tv_lshl_add_u32 %7, %0, 1, %3\n\tv_sub_u32 %6, %7, %0", 32),
     ("u32 fwd bfly vgpr 2p (7 instr)", "v_sub_u32 %7, %0, %2\n\tv_min_u32 %0, %0, %7\n\tv_mul_hi_u32 %7, %6, %2\n\tv_mad_u64_u32 %5, vcc, %6, %1, %5\n\tv_mad_u64_u32 %5, vcc, %7, %3, %5\n\tv_lshl_add_u32 %7, %0, 1, %2\n\tv_sub_u32 %6, %7, %0", 32),
    ]
src = r'''// Issue rates of VALU instruction forms on gfx950 (generated by tools/gen_ubench3.py): 8 independent chains per lane,
// 256 CUs busy, W waves per SIMD.  Reports chip-wide lane-ops/s over a ~second of steady-state load (power-capped clocks)
// and shader cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CHAINS 8
#define UNROLL 4
template <int OP> __global__ void bench(uint64_t *sink, uint64_t *cycles, int iters, uint32_t sarg) {
    uint32_t a = threadIdx.x * 2654435761u + 12345u, b = blockIdx.x * 40503u + 977u;
    uint64_t acc[CHAINS]; uint32_t acc32[CHAINS], x32[CHAINS], y32[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) { acc[c] = 0x3ff0000000000000ull + a + c; acc32[c] = b + c; x32[c] = a ^ c; y32[c] = b ^ c; }
    uint64_t w64 = 0x3ff0000000000001ull + a;
    uint32_t s = __builtin_amdgcn_readfirstlane(sarg);
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
OPS
            }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint64_t r = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) r += acc[c] + acc32[c];
    sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cycles[((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}
static const char *NAMES[] = {NAMELIST};
template <int OP> static void run(uint64_t *sink, uint64_t *cyc) {
    for (int wps : {2, 4}) {
        const int threads = 256 * wps, blocks = 256;
        const int nwaves = blocks * threads / 64;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 2048;
        float ms = 0, warm = 0;
        hipEventRecord(e0);
        while (warm < 400.f) {  // steady-state clocks
            for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(threads), 0, 0, sink, cyc, iters, 12345u);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&warm, e0, e1);
        }
        hipEventRecord(e0);
        const int reps = 50;
        for (int k = 0; k < reps; ++k) hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(threads), 0, 0, sink, cyc, iters, 12345u);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        ms /= reps;
        std::vector<uint64_t> h(nwaves);
        hipMemcpy(h.data(), cyc, nwaves * sizeof(uint64_t), hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += (double)v; avg /= nwaves;
        const double ipw = (double)iters * UNROLL * CHAINS;
        printf("%-28s waves/SIMD=%d  cycles/wave-instr/SIMD=%6.2f  chip lane-ops=%6.2f T/s\n", NAMES[OP], wps, avg / (ipw * wps),
               (double)nwaves * ipw * 64 / (ms * 1e-3) / 1e12);
        fflush(stdout);
    }
}
int main() {
    uint64_t *sink, *cyc;
    hipMalloc(&sink, 256 * 1024 * sizeof(uint64_t));
    hipMalloc(&cyc, 256 * 16 * sizeof(uint64_t));
RUNS
    return 0;
}
'''
lines=[]
for i,(name,asm,kind) in enumerate(ops):
    asm = asm.replace("\n", "\\n").replace("\t", "\\t")
    accv = "acc32[c]" if kind==32 else "acc[c]"
    lines.append('                if constexpr (OP == %d) asm volatile("%s" : "+v"(%s) : "v"(a), "v"(b), "s"(s), "v"(w64), "v"(acc[c]), "v"(x32[c]), "v"(y32[c]) : "vcc", "s10", "s11");' % (i, asm, accv) if len(sys.argv) > 1 else
                 '                if constexpr (OP == %d) asm volatile("%s" : "+v"(%s) : "v"(a), "v"(b), "s"(s), "v"(w64) : "vcc");' % (i, asm, accv))
src=src.replace("OPS","\n".join(lines)).replace("NAMELIST", ", ".join('"%s"'%o[0] for o in ops)).replace("RUNS","\n".join("    run<%d>(sink, cyc);"%i for i in range(len(ops))))
open("/root/repo/tools/ubench3%s.hip" % ("m" if len(sys.argv) > 1 else ""),"w").write(src)

#!/usr/bin/env python3
"""Instruction census of ONE butterfly of every arithmetic class (gfx950 ISA as hipcc emits it from csrc/ntt_arith.hpp):
a straight-line kernel with NB dependent butterflies against thread-dependent (VGPR) twiddles is compiled twice (NB = 4
and NB = 12) and the per-opcode difference divided by 8 -- loads, stores and address arithmetic cancel.

    python tools/bfly_census.py > profiles/r02_butterfly_census.txt        (no GPU needed: compile only)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

SRC = r'''
#include "%(root)s/concrete-ntt_amd/csrc/ntt_arith.hpp"
using namespace cntt;
template <class T, int CLS, bool INV, int NB> __global__ void k(T *o, const T *in, const TwPair<T> *tw, ModParams<T> P) {
    const uint32_t t = threadIdx.x;
#if %(BOX)s
    // the form the kernels use for 32-bit lazy residues: values held in 64-bit "boxes" (BoxOps, ntt_arith.hpp)
    uint64_t x = BoxOps<CLS>::box(in[t], P), y = BoxOps<CLS>::box(in[t + 256], P);
#else
    T x = in[t], y = in[t + 256];
#endif
    TwPair<T> w[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) w[i] = tw[t + 64 * i];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
#if %(BOX)s
        if constexpr (INV) BoxOps<CLS>::template inv<false>(x, y, w[i].w, w[i].ws, P);
        else BoxOps<CLS>::template fwd<false>(x, y, w[i].w, w[i].ws, P);
#else
        if constexpr (INV) Bfly<T, CLS>::template inv<false>(x, y, w[i].w, w[i].ws, P);
        else Bfly<T, CLS>::template fwd<false>(x, y, w[i].w, w[i].ws, P);
#endif
        auto s = x; x = y; y = s;   // alternate the roles so that both outputs stay live
    }
#if %(BOX)s
    o[t] = BoxOps<CLS>::unbox(x, P); o[t + 256] = BoxOps<CLS>::unbox(y, P);
#else
    o[t] = (T)x; o[t + 256] = (T)y;
#endif
}
template __global__ void k<%(T)s, %(CLS)s, %(INV)s, 4>(%(T)s *, const %(T)s *, const TwPair<%(T)s> *, ModParams<%(T)s>);
template __global__ void k<%(T)s, %(CLS)s, %(INV)s, 12>(%(T)s *, const %(T)s *, const TwPair<%(T)s> *, ModParams<%(T)s>);
'''


def census(T, cls, inv, box=False):
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "c.hip")
        open(src, "w").write(SRC % {"root": ROOT, "T": T, "CLS": cls, "INV": "true" if inv else "false",
                                   "BOX": "1" if box else "0"})
        out = os.path.join(tmp, "c.s")
        subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", src, "-o", out],
                       check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    counts = []
    for nb in (4, 12):
        m = re.search(r"^_Z1kI\w*Li%dEEvPT_\w*:[^\n]*\n(.*?)s_endpgm" % nb, text, re.S | re.M)
        body = m.group(1)
        c = collections.Counter()
        for ln in body.split("\n"):
            ln = ln.strip()
            if not ln or ln.startswith((".", ";", "//")) or ln.endswith(":"):
                continue
            op = ln.split()[0]
            if op.startswith(("v_", "s_nop")):
                c[re.sub(r"_e(32|64)$", "", op)] += 1
        counts.append(c)
    diff = collections.Counter()
    for op in set(counts[0]) | set(counts[1]):
        d = (counts[1][op] - counts[0][op]) / 8.0
        if abs(d) > 1e-9:
            diff[op] = d
    return diff


CASES = [("uint64_t", "CLS_LAZY", "u64 lazy (p < 2^62): Harvey butterfly, Shoup product in one asm block"),
         ("uint64_t", "CLS_STRICT", "u64 strict (p < 2^63)"),
         ("uint64_t", "CLS_GENERIC", "u64 generic (Montgomery, any odd p)"),
         ("uint64_t", "CLS_PM64", "u64 p = 2^64 - c (fold by c)"),
         ("uint64_t", "CLS_FP", "u64 p < 2^50 (double-precision FMA; range reductions not included: 3 per element every 5th fwd / 2nd inv stage)"),
         ("uint64_t", "CLS_FP51", "u64 p < 2^51 (double-precision FMA; reductions every 3rd fwd stage / every inv stage not included)"),
         ("uint32_t", "CLS_LAZY", "u32 lazy (p < 2^30), plain 32-bit registers (not used by the kernels)"),
         ("uint32_t", "CLS_LAZY", "u32 lazy (p < 2^30), boxed: x + y w - q p as two v_mad_u64_u32 on 64-bit boxes (what the kernels run)", True),
         ("uint32_t", "CLS_GENERIC", "u32 generic (Montgomery): p >= 2^31 beyond the LDS-resident sizes"),
         ("uint32_t", "CLS_FPW", "u32 p >= 2^31 on doubles in 64-bit boxes (CLS_FPW; per pass boundary and element: +1 to box, +4 to unbox)", True)]

if __name__ == "__main__":
    print(__doc__.split("\n\n")[0])
    print("s_nop = wait states hipcc inserts (two between a VALU write of VCC / an SGPR and its VALU read on gfx950; one after an")
    print("inline-asm block whose result the next instruction reads); they occupy the wave, not the VALU.")
    for case in CASES:
        T, cls, title = case[:3]
        for inv in (False, True):
            d = census(T, cls, inv, len(case) > 3)
            valu = sum(v for k, v in d.items() if k.startswith("v_"))
            mul = sum(v for k, v in d.items() if re.match(r"v_(mad_u64_u32|mul_hi_u32|mul_lo_u32|mul_f64|fma_f64|fmac_f64)", k))
            print("\n%s, %s: %.2f VALU per butterfly (%.2f multiplies), %.2f s_nop" % (
                title, "inverse (Gentleman-Sande)" if inv else "forward (Cooley-Tukey)", valu, mul, d.get("s_nop", 0)))
            for op, v in sorted(d.items(), key=lambda kv: -kv[1]):
                print("    %-24s %6.2f" % (op, v))

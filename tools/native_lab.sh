#!/bin/bash
# Builds the timing-only ablation variants of the whole-product kernel lab (tools/native_lab.hip) from PATCHED COPIES of the
# kernel headers (the product headers carry no lab switches):  tools/native_lab.sh [variant ...]
#   base | nobar (no workgroup barriers) | unitw (uniform twiddle addresses) | nolds (no exchange) | all | fam1 (XOR-swizzled layout)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
cp $R/concrete-ntt_amd/csrc/*.hpp $R/concrete-ntt_amd/csrc/*.inc $T/
python3 - $T/ntt_kernel.hpp <<'PY'
import sys
p = sys.argv[1]
s = open(p).read()
def rep(a, b):
    global s
    assert a in s, a
    s = s.replace(a, b)
rep('''            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");''', '''#ifdef LAB_NOBAR
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#else
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#endif
            asm volatile("" ::: "memory");''')
rep('''        if constexpr (SUB) toff += qpre >> (b + 1);
        if constexpr (NORM && INV && b == LOGN - 1) {''', '''        if constexpr (SUB) toff += qpre >> (b + 1);
#ifdef LAB_UNITW
        toff = (1u << (LOGN - 1 - b));
#endif
        if constexpr (NORM && INV && b == LOGN - 1) {''')
rep('''        if constexpr (K > 0) B::template gather<RM>(r, (const T *)lds, ebase, true);
        B::template stages<K, 0, IMG, NORM, TWC>(r, ebase, 0u, 0u, tw, P, tid, img);
#endif
        if constexpr (K < NPASS - 1) {
            if constexpr (K > 0) wsync();
            B::template scatter<RM>(r, lds, ebase, true);
            wsync();''', '''#ifndef LAB_NOLDS
        if constexpr (K > 0) B::template gather<RM>(r, (const T *)lds, ebase, true);
#endif
        B::template stages<K, 0, IMG, NORM, TWC>(r, ebase, 0u, 0u, tw, P, tid, img);
#endif
        if constexpr (K < NPASS - 1) {
            if constexpr (K > 0) wsync();
#ifndef LAB_NOLDS
            B::template scatter<RM>(r, lds, ebase, true);
#endif
            wsync();''')
open(p, 'w').write(s)
PY
sed -i 's/^constexpr int ACC_FAM = 3;/#ifndef LAB_FAM\n#define LAB_FAM 3\n#endif\nconstexpr int ACC_FAM = LAB_FAM;/' $T/native_fused.hpp
for v in ${@:-base}; do
  case $v in
    base) fl="" ;; nobar) fl="-DLAB_NOBAR" ;; unitw) fl="-DLAB_UNITW" ;; nolds) fl="-DLAB_NOLDS" ;;
    all) fl="-DLAB_NOBAR -DLAB_UNITW -DLAB_NOLDS" ;; fam1) fl="-DLAB_FAM=1" ;;
    roll) fl="-DLAB_OPT=16" ;; w3roll) fl="-DLAB_WPS=3 -DLAB_OPT=16" ;; w2roll) fl="-DLAB_WPS=2 -DLAB_OPT=16" ;; w2) fl="-DLAB_WPS=2" ;;
    w3rroll) fl="-DLAB_WPS=3 -DLAB_OPT=20" ;; w2lrroll) fl="-DLAB_WPS=2 -DLAB_OPT=22" ;; w3lroll) fl="-DLAB_WPS=3 -DLAB_OPT=18" ;;
    w3lr) fl="-DLAB_WPS=3 -DLAB_OPT=6" ;; w3r) fl="-DLAB_WPS=3 -DLAB_OPT=4" ;; w3) fl="-DLAB_WPS=3" ;; w2lr) fl="-DLAB_WPS=2 -DLAB_OPT=6" ;;
    *) echo "unknown variant $v"; exit 1 ;;
  esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I$T $fl $R/tools/native_lab.hip -o $R/tools/native_lab_$v &
done
wait
rm -rf $T

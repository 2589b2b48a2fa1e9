#!/bin/bash
# Builds the timing labs against SCRATCH COPIES of the kernel headers with the lab-only switches patched in -- the product
# headers under concrete-ntt_amd/csrc carry no ablation code.
#   tools/lab_build.sh blk [-DCNTT_BLK_LAB=<bits>] [-DLAB_STAMPS] ...   -> tools/blk_lab_<suffix>   (tools/blk_lab.patch)
#   tools/lab_build.sh xlane [flags]                                     -> tools/ntt_lab_xlane      (tools/xlane_lab.patch)
#   tools/lab_build.sh ntt [flags]                                       -> tools/ntt_lab            (unpatched headers)
# The patches are diffs against the headers of the commit that introduced this script; after a change to ntt_blk.hpp /
# ntt_kernel.hpp `patch` may need --fuzz or a refreshed diff.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
what=$1; shift
T=$(mktemp -d)
cp $R/concrete-ntt_amd/csrc/*.hpp $R/concrete-ntt_amd/csrc/*.inc $T/
case $what in
  blk)   (cd $T && patch -s -p1 < $R/tools/blk_lab.patch); suffix=$(echo "$*" | tr -cd '0-9A-Za-z_=' | tr '=' '_'); out=$R/tools/blk_lab_${suffix:-0}; src=$R/tools/blk_lab.hip ;;
  xlane) (cd $T && patch -s -p1 < $R/tools/xlane_lab.patch); out=$R/tools/ntt_lab_xlane; src=$R/tools/ntt_lab.hip; set -- -DCNTT_LAB_XLANE "$@" ;;
  ntt)   out=$R/tools/ntt_lab; src=$R/tools/ntt_lab.hip ;;
  *) echo "usage: $0 blk|xlane|ntt [flags]"; exit 1 ;;
esac
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I$T "$@" $src -o $out
rm -rf $T
echo $out

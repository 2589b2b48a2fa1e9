#!/bin/bash
# Same-box A/B of two builds of the library: runs the given bench command with the in-tree library ("new"), then with
# tools/_ctrl/libcntt_hip.so copied over it ("old").  Output: gpurun_out/ab_<name>_{new,old}.jsonl
#   tools/ab_lib.sh <name> <bench command ...>      (the command gets --tag new|old appended)
set -e
name=$1; shift
mkdir -p gpurun_out
"$@" --tag new > gpurun_out/ab_${name}_new.jsonl
cp concrete-ntt_amd/libcntt_hip.so /tmp/libcntt_new.so
# whatever happens to the control run (failure, interrupt), the tree gets the NEW library back
trap 'cp /tmp/libcntt_new.so concrete-ntt_amd/libcntt_hip.so' EXIT
cp tools/_ctrl/libcntt_hip.so concrete-ntt_amd/libcntt_hip.so
"$@" --tag old > gpurun_out/ab_${name}_old.jsonl

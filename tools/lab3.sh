#!/bin/bash
mkdir -p gpurun_out
for m in 0 1 2; do echo "=== CNTT_MULLO2_MODE=$m"; timeout -k 5 150 ./tools/ntt_lab_m$m 2>&1 | grep -E "baseline|ALU only|check|wp 256"; done | tee gpurun_out/lab3.log

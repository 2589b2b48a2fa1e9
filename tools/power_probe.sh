#!/bin/bash
mkdir -p gpurun_out
( for i in $(seq 1 120); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/smi_load.txt 2>&1 &
SMI=$!
timeout -k 5 200 ./tools/ntt_lab > gpurun_out/lab.log 2>&1
kill $SMI 2>/dev/null
cat gpurun_out/lab.log
echo ---; awk 'NR%4==0' gpurun_out/smi_load.txt | head -40

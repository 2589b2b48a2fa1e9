#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python tools/bench_configs.py > gpurun_out/configs.log 2>&1; echo "configs rc=$?" | tee gpurun_out/progress.log
grep -v amdgpu.ids gpurun_out/configs.log | cut -c1-260

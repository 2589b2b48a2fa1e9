#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x --timeout=500 > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee gpurun_out/progress.log
tail -15 gpurun_out/pytest_gpu.log
[ $rc -eq 0 ] && timeout -k 10 400 python tools/bench_configs.py p32 c4 c3 c5 > gpurun_out/configs.log 2>&1; grep -v amdgpu.ids gpurun_out/configs.log | cut -c1-230

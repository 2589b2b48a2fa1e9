#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout=500 > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee gpurun_out/progress.log
tail -5 gpurun_out/pytest_gpu.log
[ $rc -eq 0 ] && bash tools/profile.sh

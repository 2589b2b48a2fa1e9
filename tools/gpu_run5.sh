#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout=500 > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee gpurun_out/progress.log
tail -5 gpurun_out/pytest_gpu.log
timeout -k 10 300 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?" | tee -a gpurun_out/progress.log
tail -2 gpurun_out/bench.log

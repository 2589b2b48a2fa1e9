#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 120 ./tools/ubench2 > gpurun_out/ubench2.log 2>&1; echo "ubench2 rc=$?" | tee gpurun_out/progress.log
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout=400 > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a gpurun_out/progress.log
tail -40 gpurun_out/pytest_gpu.log

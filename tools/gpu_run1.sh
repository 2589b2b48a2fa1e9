#!/bin/bash
# first GPU session: instruction-rate microbench, parity tests, smoke, bench, kernel trace
set -o pipefail
mkdir -p gpurun_out
echo "== ubench" | tee gpurun_out/progress.log
timeout -k 10 120 ./tools/ubench_valu > gpurun_out/ubench.log 2>&1; echo "ubench rc=$?" | tee -a gpurun_out/progress.log
echo "== smoke" | tee -a gpurun_out/progress.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc" | tee -a gpurun_out/progress.log
tail -5 gpurun_out/smoke.log
echo "== pytest gpu" | tee -a gpurun_out/progress.log
timeout -k 10 640 python -m pytest tests -m gpu -q -x --timeout=300 > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a gpurun_out/progress.log
tail -30 gpurun_out/pytest_gpu.log
echo "== bench" | tee -a gpurun_out/progress.log
timeout -k 10 300 python bench.py --steps 10 --warmup 2 > gpurun_out/bench.log 2>&1; echo "bench rc=$?" | tee -a gpurun_out/progress.log
tail -3 gpurun_out/bench.log

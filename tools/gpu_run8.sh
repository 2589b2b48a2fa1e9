#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?" | tee gpurun_out/progress.log
tail -1 gpurun_out/bench.log | cut -c1-600
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 1 --batch 16384 --dist-backend gloo --ramp-seconds 0.5 > gpurun_out/bench2.log 2>&1; echo "bench2 rc=$?" | tee -a gpurun_out/progress.log
grep -E "^\{|Error|error" gpurun_out/bench2.log | cut -c1-400

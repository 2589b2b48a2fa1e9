#!/bin/bash
# rocprofv3 evidence: kernel trace + stats of the driver's bench command, then PMC passes (counters in their own
# runs, kernel-trace / stats only beside them) over the kernel zoo of tools/prof_workload.py.
# usage: tools/profile.sh [outdir-name]   (run from the repo root on the GPU box)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# --no-cpu-baseline: the CPU leg builds the oracle through make -> sh -> cc; no such exec hops (and no 12 s of all-core
# CPU timing) inside a profiler-initialised process tree.  Build the checker BEFORE profiling, outside the profiler.
ARGS="$R/bench.py --gpus 1 --steps 200 --warmup 5 --no-cpu-baseline"
ZOO="$R/tools/prof_workload.py 4"
echo "== kernel trace" | tee -a $R/gpurun_out/progress.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1; echo "trace rc=$?" | tee -a $R/gpurun_out/progress.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/zoo_trace -- python3 $ZOO > $OUT/zoo_trace.log 2>&1; echo "zoo trace rc=$?" | tee -a $R/gpurun_out/progress.log
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  echo "== pmc $pass" | tee -a $R/gpurun_out/progress.log
  timeout -k 10 300 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$name -- python3 $ZOO > $OUT/pmc_$name.log 2>&1; echo "pmc rc=$?" | tee -a $R/gpurun_out/progress.log
done
cd $R
# the stamp is the hash compiled into the LIBRARY THE PASSES LOADED (cntt_version(): concrete_ntt_amd.build_info()), not a hash of the
# source tree next to it: bench.py compares it with the library it runs and flags a stale profile
STAMP=$(python3 -c "import concrete_ntt_amd as c; print(c.build_info()['csrc_hash'])" 2>/dev/null | tail -1)
python3 tools/trace_gaps.py $OUT/trace > $OUT/unfused_step_gaps.txt 2>&1
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1; python3 tools/summarize_prof.py $OUT --json --stamp "$STAMP" > $OUT/pmc_traffic.json 2>/dev/null; tail -40 $OUT/summary.txt; cat $OUT/unfused_step_gaps.txt

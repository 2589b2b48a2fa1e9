#!/bin/bash
# rocprofv3 kernel trace + PMC passes (each in its own run) over ONE workload script.
# usage: tools/profile_one.sh <outdir-name> <script.py> [args...]      (from the repo root on the GPU box)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1
shift
W="$R/$1"
shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/zoo_trace -- python3 $W "$@" > $OUT/zoo_trace.log 2>&1; echo "trace rc=$?"
# PMC_PASSES: ';'-separated groups of counters, one rocprofv3 run per group
DEFAULT_PASSES="FETCH_SIZE;WRITE_SIZE;SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES;SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
IFS=';' read -ra PASSES <<< "${PMC_PASSES:-$DEFAULT_PASSES}"
for pass in "${PASSES[@]}"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$name -- python3 $W "$@" > $OUT/pmc_$name.log 2>&1; echo "pmc $name rc=$?"
done
cd $R && python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1; tail -60 $OUT/summary.txt

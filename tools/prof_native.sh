#!/bin/bash
# PMC passes over the whole-product kernels (tools/prof_native.py) for the current environment (CNTT_SWITCHES=native_acc=0,
# CNTT_ACC_VARIANT are inherited).  usage: tools/prof_native.sh outdir-name   (repo root, GPU box)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-prof_native}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
W="$R/tools/prof_native.py 4"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/zoo_trace -- python3 $W > $OUT/trace.log 2>&1; echo "trace rc=$?"
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
  "SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_MFMA_MOPS_I8" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
  "TA_BUSY_avr TA_TA_BUSY_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$i -- python3 $W > $OUT/pmc_$i.log 2>&1; echo "pmc $i ($pass) rc=$?"
done
cd $R && python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1; grep -A40 "PMC" $OUT/summary.txt | head -80

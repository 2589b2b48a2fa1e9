#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/prof9
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof9 -- python3 $R/tools/bench_configs.py c3 c5 > $R/gpurun_out/prof9.log 2>&1
cd $R; f=$(ls -t gpurun_out/prof9/*/*kernel_stats.csv | head -1); python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-90s calls=%5s avg_us=%9.1f pct=%5s" % (r["Name"].replace("cntt::","").replace("unsigned int","u32").replace("unsigned long","u64")[:90], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY

#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -rf $R/gpurun_out/prof; mkdir -p $R/gpurun_out/prof
bash tools/profile.sh > $R/gpurun_out/profile_run.log 2>&1
tail -5 $R/gpurun_out/profile_run.log

#!/bin/bash
# rocprofv3 kernel stats of one shard-shaped rk_p3_prove (tools/bench_p3.py); usage on the GPU box: bash tools/prof_p3.sh r03 [lookups]
TAG=${1:-r03}
LK=${2:-0}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/${TAG}_p3_prof_lk$LK -o $TAG -- python3 $R/tools/bench_p3.py --shape 20x256 --jit --reps 4 --no-verify --lookups $LK > $R/gpurun_out/${TAG}_p3_prof_lk$LK.log 2>&1 || exit 1
head -40 $R/gpurun_out/${TAG}_p3_prof_lk$LK/${TAG}_kernel_stats.csv

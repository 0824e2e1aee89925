#!/bin/bash
# per-kernel device time of the PCS steps (tools/bench_pcs.py under rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_pcs -o pcs -- python3 $R/tools/bench_pcs.py > $R/gpurun_out/prof_pcs.log 2>&1 || exit 1

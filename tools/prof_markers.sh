#!/bin/bash
# roctx ranges of the proof stages (RK_ROCTX=1) next to the kernel trace
cd /tmp && export TMPDIR=/tmp
export RK_ROCTX=1
R=$GRAFT_REPO_ROOT
rocprofv3 --marker-trace --kernel-trace --stats -f csv -d $R/gpurun_out/prof_markers -o mk -- python3 $R/bench.py --steps 6 --warmup 1 --inflight 1 --no-cpu --no-h2d --no-small > $R/gpurun_out/prof_markers.log 2>&1

#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_po2_18 -o p18 -- python3 $R/bench.py --po2 18 --steps 48 --warmup 3 --no-cpu --no-h2d > $R/gpurun_out/prof_po2_18.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_po2_18_if1 -o p18 -- python3 $R/bench.py --po2 18 --steps 48 --warmup 3 --inflight 1 --no-cpu --no-h2d > $R/gpurun_out/prof_po2_18_if1.log 2>&1 || exit 1

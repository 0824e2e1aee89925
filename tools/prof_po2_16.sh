#!/bin/bash
# kernel trace of serial 2^16-cycle proofs (one context): what a small segment's 5 ms are made of
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_po2_16_if1 -o p16 -- python3 $R/bench.py --po2 16 --steps 24 --warmup 3 --inflight 1 --no-cpu --no-h2d --no-verify > $R/gpurun_out/prof_po2_16_if1.log 2>&1 || exit 1

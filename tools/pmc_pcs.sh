#!/bin/bash
# SQ counters (LDS conflicts, VALU activity) of the PCS kernels: tools/bench_pcs.py under rocprofv3 --pmc (own pass, no trace domains)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -f csv -d $R/gpurun_out/pmc_pcs -o pcs -- python3 $R/tools/bench_pcs.py > $R/gpurun_out/pmc_pcs.log 2>&1 || exit 1

#!/bin/bash
# SQ counters of the constraint-list evaluator (program_kernel): two passes, counters only
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -f csv -d $R/gpurun_out/pmc_prog_a -o a -- python3 $R/tools/bench_program.py --po2 20 --sizes 10000 --reps 1 --only-local > $R/gpurun_out/pmc_prog_a.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVES -f csv -d $R/gpurun_out/pmc_prog_b -o b -- python3 $R/tools/bench_program.py --po2 20 --sizes 10000 --reps 1 --only-local > $R/gpurun_out/pmc_prog_b.log 2>&1 || exit 1

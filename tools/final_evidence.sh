#!/bin/bash
# The round's judged evidence in one GPU call: rocprofv3 kernel stats (serial and three in flight), the HBM
# counters of the dominant kernel (separate --pmc passes), the SQ LDS / VALU counters, the contract bench line.
# usage (on the GPU box): bash tools/final_evidence.sh r03 ; then here: python tools/collect_profiles.py r03
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
G=$R/gpurun_out
B=$R/bench.py
rocprofv3 --kernel-trace --stats -f csv -d $G/${TAG}_final_if1 -o $TAG -- python3 $B --no-cpu --no-h2d --no-small --circuit '' --inflight 1 --steps 8 --warmup 2 > $G/${TAG}_final_if1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats -f csv -d $G/${TAG}_final_if3 -o $TAG -- python3 $B --no-cpu --no-h2d --no-small --circuit '' --steps 24 --warmup 3 > $G/${TAG}_final_if3.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE -f csv -d $G/${TAG}_final_pmc_fetch -o $TAG -- python3 $B --no-cpu --no-h2d --no-small --circuit '' --no-verify --inflight 1 --steps 1 --warmup 0 > $G/${TAG}_final_pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE -f csv -d $G/${TAG}_final_pmc_write -o $TAG -- python3 $B --no-cpu --no-h2d --no-small --circuit '' --no-verify --inflight 1 --steps 1 --warmup 0 > $G/${TAG}_final_pmc_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -f csv -d $G/${TAG}_final_pmc_sq -o $TAG -- python3 $B --no-cpu --no-h2d --no-small --circuit '' --no-verify --inflight 1 --steps 1 --warmup 0 > $G/${TAG}_final_pmc_sq.log 2>&1 || exit 1
cd $R && python3 bench.py > $G/${TAG}_final_bench.json 2> $G/${TAG}_final_bench.err || exit 1
cat $G/${TAG}_final_bench.json | cut -c1-300
# the other column counts SURVEY 8(d) asks for (W = 64, W = 512), same entry point, no CPU leg
: > $G/${TAG}_other_shapes.jsonl
for WD in 16,16,32 16,16,480; do
  python3 bench.py --widths $WD --no-cpu --no-small --no-h2d --circuit '' --steps 24 --warmup 2 >> $G/${TAG}_other_shapes.jsonl 2>> $G/${TAG}_final_bench.err || exit 1
done
# SP1's side: a shard with lookups, the Poseidon2 chip, the hash part of a compress step
python3 bench.py --preset sp1-p3 --p3-jit --p3-lookups 8 --steps 12 > $G/${TAG}_bench_sp1_p3_lookups.json 2>> $G/${TAG}_final_bench.err || exit 1
python3 tools/bench_p2_chip.py > $G/${TAG}_bench_p2_chip.jsonl 2>> $G/${TAG}_final_bench.err || exit 1
python3 tools/bench_compress_hashes.py > $G/${TAG}_bench_compress_hashes.json 2>> $G/${TAG}_final_bench.err || exit 1

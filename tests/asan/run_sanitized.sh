#!/bin/bash
# The library's HOST code under AddressSanitizer + UndefinedBehaviorSanitizer (device code compiled as usual:
# -fno-gpu-sanitize; GPU sanitizers are not available on this pool), exercised by the CPU test files that drive host
# code with hostile input: the seal verifier and the uni-stark verifier on ~2 000 mutated proofs, the mixed-matrix
# verifier, the constraint-list compiler / host evaluator, the AIR front end, the executor on random instruction
# streams, the parameter blob.  ~9 minutes to build (-O1 -g, 15 translation units), ~1 minute to run; not part of the
# default test run.  usage: bash tests/asan/run_sanitized.sh [build dir]   (no GPU needed)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
B=${1:-$ROOT/tests/asan/_build/host}
mkdir -p "$B"
SRC=$ROOT/raiko_amd/csrc
FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer"
for f in context kernels_ntt kernels_hash kernels_poly kernels_scan circuit_program circuit_jit mmcs kernels_pcs p3_air p3 comm prover verify session; do
  [ "$B/$f.o" -nt "$SRC/$f.hip" ] || /opt/rocm/bin/hipcc $FLAGS -c "$SRC/$f.hip" -o "$B/$f.o" &
  if (( $(jobs -r | wc -l) >= 4 )); then wait -n; fi
done
[ "$B/executor.o" -nt "$SRC/executor.cpp" ] || /opt/rocm/bin/hipcc -x hip $FLAGS -c "$SRC/executor.cpp" -o "$B/executor.o" &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -fno-gpu-sanitize "$B"/*.o -o "$B/libraiko_hip_asan.so" -lhiprtc -ldl
RT=$(find /opt/rocm/lib/llvm -name 'libclang_rt.asan-x86_64.so' | head -1)
cd "$ROOT"
fail=0
for t in test_p3 test_p2_chip test_verifier_fuzz test_verify test_mmcs test_program test_executor test_params test_abi test_prover_api test_pcs; do
  RAIKO_HIP_LIB="$B/libraiko_hip_asan.so" LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 \
    python -m pytest tests/$t.py -x -q -m "not gpu" -p no:cacheprovider > "$B/$t.log" 2>&1 || fail=1
  n=$(grep -c 'runtime error\|ERROR: AddressSanitizer' "$B/$t.log" || true)
  echo "$t: $(tail -1 "$B/$t.log") -- sanitizer reports: $n"
  [ "$n" = "0" ] || fail=1
done
exit $fail

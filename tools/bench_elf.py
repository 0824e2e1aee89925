#!/usr/bin/env python3
"""ELF -> receipt on one MI355X through the stand-in trace circuit: a hand-assembled loop of ~4 M cycles executed and
cut into 2^20-cycle segments (rk_exec_elf with record_trace), witness columns (rk_exec_witness), every segment
proven with the constraint list behind eval_check and verified inside rk_prove_session.  One JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rv32_asm as A  # noqa: E402
from raiko_amd import executor as X  # noqa: E402
from raiko_amd.hal import HipHal, prove_session  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 800000
    prog = A.li("a2", iters) + ["loop:", ("addi", "a3", "a3", 3), ("xor", "a4", "a4", "a3"), ("slli", "a5", "a4", 1),
                                ("addi", "a2", "a2", -1), ("bne", "a2", "zero", "loop")] + A.li("t0", 0) + [("ecall",)]
    image = A.elf(A.assemble(prog)[0])
    t0 = time.perf_counter()
    ex = X.execute(image, segment_limit_po2=20, record_trace=True)
    t1 = time.perf_counter()
    segs = X.trace_segments(ex)
    t2 = time.perf_counter()
    hal = HipHal(0)
    segs[0].program.compile(hal)                      # hiprtc: the generated eval_check kernel
    t3 = time.perf_counter()
    prove_session(segs[:1], inflight=1, verify=False, program=segs[0].program)   # warm-up: contexts, tables
    t4 = time.perf_counter()
    seals = prove_session(segs, inflight=3, verify=True, program=segs[0].program)
    t5 = time.perf_counter()
    X.execute_and_prove(image, segment_limit_po2=20, circuit="trace", pipeline=True)    # warm: witness context, stream pool
    t6 = time.perf_counter()
    X.execute_and_prove(image, segment_limit_po2=20, circuit="trace", pipeline=True)    # executor and prover overlapped, witness on the GPU
    t7 = time.perf_counter()
    X.execute_and_prove(image, segment_limit_po2=20, circuit="trace", pipeline=True, device_witness=False)
    t8 = time.perf_counter()
    # the same execution as SP1-style shards: one uni-stark proof per 2^20-cycle shard (rk_p3_prove_shards, 16-column trace AIR,
    # SP1's parameter set: blow-up 2, 100 queries, 16 proof-of-work bits), every proof verified inside
    from raiko_amd import p3
    from raiko_amd.hal import make_params
    blob = make_params(1)
    shards = X.p3_shards(ex)
    for tb, _ in shards:
        tb[0].air.compile(hal)
    p3.prove_shards(shards[:1], blob, batch=1, verify=False)              # warm: contexts, tables
    t9 = time.perf_counter()
    proofs = p3.prove_shards(shards, blob, batch=3, verify=True)
    t10 = time.perf_counter()
    p3_part = {"shards": len(shards), "prove_and_verify_s": round(t10 - t9, 3), "proven_cycles_per_s": round(ex.total_cycles / (t10 - t9), 1),
               "proof_words": [int(pf.size) for pf in proofs]}
    # ... and with the tables tied by lookups: cpu (16 columns, 11 interactions) + program + 2^16-row range table per shard
    t11 = time.perf_counter()
    lk = X.p3_shards(ex, lookups=True)
    t12 = time.perf_counter()
    for tb, _ in lk[:1]:
        for t in tb:
            t.air.compile(hal)
    p3.prove_shards(lk[:1], blob, batch=1, verify=False)
    t13 = time.perf_counter()
    lproofs = p3.prove_shards(lk, blob, batch=3, verify=True)
    t14 = time.perf_counter()
    p3_lk = {"shards": len(lk), "tables_per_shard": [[t.trace.shape[0], t.air.width, t.air.perm_width] for t in lk[0][0]],
             "host_multiplicities_s": round(t12 - t11, 3), "prove_and_verify_s": round(t14 - t13, 3),
             "proven_cycles_per_s": round(ex.total_cycles / (t14 - t13), 1), "proof_words": [int(pf.size) for pf in lproofs]}
    # ... and the whole ELF -> verified shard proofs route overlapped: executor | cpu table written on the GPU | prover | verifier
    pipe = X.P3Pipeline(blob)                                  # two contexts and the three compiled AIRs, kept across programs
    pipe.run(image, shard_po2=20)                              # warm: first-touch allocations
    t15 = time.perf_counter()
    exp, pproofs, _ = pipe.run(image, shard_po2=20)
    t16 = time.perf_counter()
    pipe.close()
    p3_lk["pipelined_elf_to_verified_proofs_s"] = round(t16 - t15, 3)
    p3_lk["cycles_per_s_pipelined"] = round(exp.total_cycles / (t16 - t15), 1)
    p3_lk["pipelined_proofs_identical"] = all(np.array_equal(a, b) for a, b in zip(pproofs, lproofs))
    print(json.dumps({"what": "ELF -> receipt through the stand-in trace circuit (4 + 2 + 16 columns)", "cycles": ex.total_cycles,
                      "as_uni_stark_shards": p3_part, "as_uni_stark_shards_with_lookups": p3_lk,
                      "pipelined_execute_to_receipt_s": round(t7 - t6, 3),
                      "cycles_per_s_pipelined": round(ex.total_cycles / (t7 - t6), 1),
                      "pipelined_host_witness_s": round(t8 - t7, 3),
                      "cycles_per_s_pipelined_host_witness": round(ex.total_cycles / (t8 - t7), 1),
                      "segments": len(segs), "execute_and_witness_s": round(t1 - t0, 3), "segments_and_program_s": round(t2 - t1, 3),
                      "hiprtc_compile_s": round(t3 - t2, 3), "prove_and_verify_s": round(t5 - t4, 3),
                      "proven_cycles_per_s_prove_only": round(ex.total_cycles / (t5 - t4), 1),
                      "cycles_per_s_execute_to_receipt": round(ex.total_cycles / ((t1 - t0) + (t5 - t4)), 1),
                      "seal_words": [int(s.size) for s in seals]}))


if __name__ == "__main__":
    main()

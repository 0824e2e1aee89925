#!/usr/bin/env python3
"""Throughput of the constraint-list evaluator (rk_program_eval_check) on one MI355X: random lists of a
given size over a W = 16/16/224 LDE at 2^po2 cycles, and the toy circuit's list run three ways
(interpreter, generated straight-line kernel, hand-written kernel).  One JSON line per case.

    python tools/bench_program.py --po2 20 > gpurun_out/r02_bench_program.jsonl
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from raiko_amd import _lib, circuit_program as cp, toy_circuit  # noqa: E402
from raiko_amd.hal import HipHal  # noqa: E402
from raiko_amd.segment import P, synthetic_tapset  # noqa: E402


def view_for(hal, po2, bufs, widths, globals_, mix):
    v = _lib.RkCircuitView()
    v.ctx = hal._ctx
    v.po2 = po2
    for g in range(3):
        v.group_size[g] = widths[g]
        v.d_lde[g] = bufs[g].ptr
    v.globals, v.n_globals = globals_.ctypes.data_as(_lib.u32p), globals_.size
    v.mix, v.n_mix = mix.ctypes.data_as(_lib.u32p), mix.size
    return v


def timed(hal, fn, reps):
    fn()
    hal.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    hal.sync()
    return (time.perf_counter() - t0) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--po2", type=int, default=20)
    ap.add_argument("--sizes", default="1000,10000,40000")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--jit", action="store_true", help="also rk_program_compile the expression-tree lists (hiprtc) and time the generated kernel")
    ap.add_argument("--only-local", action="store_true", help="only the expression-tree list (for counter collection)")
    ap.add_argument("--gen", help="directory with <name>.npz (steps, ret) and lib<name>.so built from tools/circuit_gen.py output: "
                                  "the list through the interpreter and through its generated code, outputs compared")
    ap.add_argument("--gen-name", default="big10000")
    args = ap.parse_args()
    from program_util import random_program
    import torch
    lib = _lib.load()
    ts = torch.cuda.Stream()                       # a stream whose handle the hand-written hooks can be given
    hal = HipHal(0, stream=ts.cuda_stream)
    rng = np.random.default_rng(1)
    po2, d = args.po2, 4 << args.po2
    widths = (16, 16, 224)
    taps = synthetic_tapset(*widths)
    bufs = [hal.copy_from_elem(rng.integers(0, P, size=(w, d), dtype=np.uint32)) for w in widths]
    globals_ = rng.integers(0, P, size=32, dtype=np.uint32)
    mix = rng.integers(0, P, size=40, dtype=np.uint32)
    pm = rng.integers(0, P, size=4, dtype=np.uint32)
    out = hal.alloc_elem(4 * d)
    v = view_for(hal, po2, bufs, widths, globals_, mix)
    v.stream = ts.cuda_stream
    for n_ops, local in [(int(x), loc) for x in args.sizes.split(",") for loc in ((True,) if args.only_local else (True, False))]:
        steps, ret = random_program(rng, taps, 32, 40, n_fp_ops=n_ops, n_live=0, depth=2, n_constraints=max(8, n_ops // 10),
                                    local=local)
        prog = cp.Program(steps, ret, taps)
        info = prog.info()
        run = lambda: _lib.check(hal._ctx, lib.rk_program_eval_check(prog.handle, C.byref(v), pm.ctypes.data_as(_lib.u32p), out.ptr))
        t = timed(hal, run, args.reps)
        shape = "expression trees (few values alive)" if local else "uniform operands (everything stays alive: spill-bound)"
        extra = {}
        if local and args.jit:
            t0 = time.perf_counter()
            prog.compile(hal)
            extra["hiprtc_compile_s"] = round(time.perf_counter() - t0, 2)
            extra["ms_runtime_compiled"] = round(timed(hal, run, args.reps) * 1e3, 3)
        print(json.dumps({**extra, "what": "rk_program_eval_check, random list, " + shape, "po2": po2, "points": d, "widths": widths,
                          "steps": info["n_steps"], "ops": info["n_ops"], "fp_slots": info["n_fp_slots"],
                          "mix_slots": info["n_mix_slots"], "ms": round(t * 1e3, 3),
                          "G_point_ops_per_s": round(info["n_ops"] * d / t / 1e9, 2)}), flush=True)
        prog.close()
    if args.only_local:
        return
    if args.gen:
        z = np.load(os.path.join(args.gen, args.gen_name + ".npz"))
        prog = cp.Program(z["steps"], int(z["ret"]), taps)
        so = C.CDLL(os.path.join(args.gen, "lib%s.so" % args.gen_name))
        gen_fn = C.cast(getattr(so, args.gen_name + "_eval_check"), _lib.EVAL_CHECK_FN)
        pmp = pm.ctypes.data_as(_lib.u32p)
        out2 = hal.alloc_elem(4 * d)
        t_int = timed(hal, lambda: _lib.check(hal._ctx, lib.rk_program_eval_check(prog.handle, C.byref(v), pmp, out.ptr)), args.reps)

        def run_gen():
            assert gen_fn(None, C.byref(v), pmp, out2.ptr) == 0
        t_gen = timed(hal, run_gen, args.reps)
        same = bool(np.array_equal(out.to_host(), out2.to_host()))
        info = prog.info()
        print(json.dumps({"what": "one list, interpreter vs generated code (tools/circuit_gen.py)", "po2": po2, "points": d,
                          "ops": info["n_ops"], "fp_slots": info["n_fp_slots"], "interpreter_ms": round(t_int * 1e3, 3),
                          "generated_ms": round(t_gen * 1e3, 3), "outputs_identical": same,
                          "generated_G_point_ops_per_s": round(info["n_ops"] * d / t_gen / 1e9, 1)}), flush=True)
        assert same
    # the toy circuit three ways (widths 8/4/8)
    toy_circuit.load()
    tw = toy_circuit.GEN_WIDTHS
    ttaps = synthetic_tapset(*tw)
    tbufs = [hal.copy_from_elem(rng.integers(0, P, size=(w, d), dtype=np.uint32)) for w in tw]
    tmix = rng.integers(0, P, size=toy_circuit.GEN_N_MIX, dtype=np.uint32)
    tv = view_for(hal, po2, tbufs, tw, globals_, tmix)
    steps, ret = cp.toy_program(ttaps, toy_circuit.GEN_N_MIX)
    prog = cp.Program(steps, ret, ttaps)
    pmp = pm.ctypes.data_as(_lib.u32p)
    hand = C.cast(C.c_void_p(toy_circuit.hooks_ptr()), C.POINTER(_lib.RkCircuitHooks)).contents.eval_check
    gen = C.cast(C.c_void_p(toy_circuit.gen_hooks_ptr()), C.POINTER(_lib.RkCircuitHooks)).contents.eval_check
    tv.stream = ts.cuda_stream
    cases = {"interpreter": lambda: lib.rk_program_eval_check(prog.handle, C.byref(tv), pmp, out.ptr),
             "generated": lambda: gen(None, C.byref(tv), pmp, out.ptr),
             "hand-written (other mixing)": lambda: hand(None, C.byref(tv), pmp, out.ptr)}
    for name, fn in cases.items():
        def run():
            assert fn() == 0
        t = timed(hal, run, args.reps)
        print(json.dumps({"what": "toy circuit eval_check: " + name, "po2": po2, "points": d, "ops": prog.info()["n_ops"],
                          "ms": round(t * 1e3, 3)}), flush=True)


if __name__ == "__main__":
    main()

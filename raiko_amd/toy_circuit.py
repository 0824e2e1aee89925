"""Host side of the toy circuit (examples/toy_circuit): witness generation and the ctypes
binding of libtoy_circuit.so, whose `rk_circuit_hooks` run CircuitHal::accumulate / eval_check
on the GPU inside a segment proof and whose `poly_ext` lets the verifier check the constraint
identity.  It plays the part risc0-circuit-rv32im 1.0.1 plays behind `session.prove()`
(reference provers/risc0/driver/src/bonsai.rs:271) -- that crate is not in the reference tree,
so this is the same interface on a circuit small enough to read (see toy_circuit.h), not rv32im.
"""
import ctypes as C
import os
import subprocess
from typing import Tuple

import numpy as np

from . import _lib
from .segment import P, Segment, synthetic_tapset

_HERE = os.path.dirname(os.path.abspath(__file__))
TOY_DIR = os.path.join(os.path.dirname(_HERE), "examples", "toy_circuit")
TOY_SO = os.path.join(TOY_DIR, "_build", "libtoy_circuit.so")
# the same circuit's constraint list (circuit_program.toy_program at widths 8/4/8, 8 mix elements) turned
# into straight-line HIP by tools/circuit_gen.py
GEN_SO = os.path.join(TOY_DIR, "_build", "libtoy_gen.so")
GEN_WIDTHS, GEN_N_MIX = (8, 4, 8), 8
_toy = None
_gen = None
_gen_hooks = None


def build(force: bool = False) -> str:
    root = os.path.dirname(_HERE)
    srcs = [os.path.join(TOY_DIR, f) for f in ("toy_circuit.hip", "toy_circuit.h", "Makefile")] + [
        _lib.LIB_PATH, os.path.join(root, "tools", "circuit_gen.py"), os.path.join(_HERE, "circuit_program.py")]
    outs = (TOY_SO, GEN_SO)
    if force or not all(os.path.exists(x) for x in outs) or any(
            os.path.getmtime(s) > min(os.path.getmtime(x) for x in outs) for s in srcs):
        r = subprocess.run(["make", "-C", TOY_DIR], capture_output=True, text=True)
        if r.returncode != 0:
            raise _lib.HipLibraryError("building libtoy_circuit.so failed:\n" + r.stderr[-2000:])
    return TOY_SO


def load():
    """libtoy_circuit.so (links libraiko_hip.so); no CPU fallback."""
    global _toy
    if _toy is None:
        _lib.load()
        if not os.path.exists(TOY_SO):
            raise _lib.HipLibraryError(f"{TOY_SO} not found: run `make -C examples/toy_circuit`")
        lib = C.CDLL(TOY_SO)
        lib.toy_circuit_hooks.restype = C.c_void_p
        lib.toy_circuit_hooks.argtypes = []
        _toy = lib
    return _toy


def hooks_ptr() -> int:
    return int(load().toy_circuit_hooks())


def load_gen():
    """libtoy_gen.so: eval_check / poly_ext generated from the toy circuit's constraint list"""
    global _gen
    if _gen is None:
        load()
        if not os.path.exists(GEN_SO):
            raise _lib.HipLibraryError(f"{GEN_SO} not found: run `make -C examples/toy_circuit`")
        _gen = C.CDLL(GEN_SO)
    return _gen


def gen_hooks_ptr() -> int:
    """rk_circuit_hooks {accumulate: the hand-written one, eval_check: the generated straight-line kernel}"""
    global _gen_hooks
    if _gen_hooks is None:
        src = C.cast(C.c_void_p(hooks_ptr()), C.POINTER(_lib.RkCircuitHooks)).contents
        h = _lib.RkCircuitHooks()
        h.user = src.user
        h.accumulate = src.accumulate
        h.eval_check = C.cast(load_gen().toy_gen_eval_check, _lib.EVAL_CHECK_FN)
        _gen_hooks = h
    return C.addressof(_gen_hooks)


def gen_poly_ext_fn():
    return C.cast(load_gen().toy_gen_poly_ext, _lib.POLY_EXT_FN)


def poly_ext_fn():
    """the verifier callback as a ctypes function pointer of type _lib.POLY_EXT_FN"""
    return C.cast(load().toy_circuit_poly_ext, _lib.POLY_EXT_FN)


def to_mont(x: np.ndarray) -> np.ndarray:
    return (np.asarray(x, dtype=np.uint64) % P * ((1 << 32) % P) % P).astype(np.uint32)


def toy_segment(po2: int, widths: Tuple[int, int, int] = (8, 4, 8), seed: int = 7, n_globals: int = 8,
                n_accum_mix: int = 8, break_row: int = -1) -> Segment:
    """A segment whose code/data columns satisfy the toy constraints; accum and check are left to
    the hooks.  `break_row` >= 0 corrupts d1 on that row (the proof must then fail)."""
    wa, wc, wd = widths
    assert wa >= 4 and wc >= 3 and wd >= 4 and n_accum_mix >= 4
    n = 1 << po2
    rng = np.random.Generator(np.random.PCG64(seed))
    code = rng.integers(0, P, size=(wc, n), dtype=np.uint64)
    data = rng.integers(0, P, size=(wd, n), dtype=np.uint64)
    code[0:3] = 0
    code[0, 0] = 1
    code[1, 1 % n] = 1 if n > 1 else 0
    code[2, n - 1] = 1
    fib = [int(data[0, 0]), int(data[0, 1])]
    for i in range(2, n):
        fib.append((fib[-1] + fib[-2]) % P)
    data[0] = np.array(fib[:n], dtype=np.uint64)
    data[1] = data[0] * np.roll(data[0], 1) % P
    data[3] = data[2][rng.permutation(n)]
    if break_row >= 0:
        data[1, break_row] = (data[1, break_row] + 1) % P
    taps = synthetic_tapset(wa, wc, wd)
    globals_ = rng.integers(0, P, size=(n_globals,), dtype=np.uint32)
    seg = Segment(po2=po2, taps=taps, groups=[None, to_mont(code), to_mont(data)], check=None, globals_=globals_,
                  n_accum_mix=n_accum_mix, circuit_info=b"TOY_CIRCUIT:v1__")
    seg.hooks = hooks_ptr
    return seg

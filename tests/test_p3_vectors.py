"""The route from "parity unpinned" to "pinned against Plonky3 itself" for BASELINE config 5.

`provers/hip/vectors-p3` is a small Rust program against the Plonky3 crates at the revision raiko pins through sp1
(reference Cargo.lock:4889-5127, rev 88ea2b86; the path `client.prove` of provers/sp1/driver/src/lib.rs:44-57 runs through
them).  It cannot be built in this image (no Rust toolchain, no network).  Run anywhere else, it writes three files into
tests/golden/plonky3-88ea2b8/: the width-16 Poseidon2 (with the constants of its instance), sponge and compression; the
coset LDE and the FRI fold of the two-adic PCS; one whole uni-stark proof of Plonky3's Fibonacci AIR in rk_p3_prove's word
order.  While that directory holds no .bin file the `reference` tests SKIP; once it does, the CPU oracle and -- under
-m gpu -- the HIP path through the C ABI are compared with every vector, word for word.

What runs today: the same comparison code on files of the same format made from the oracle itself, so the reader, the
checks and the failure on a single changed word are exercised."""
import glob
import os
import sys

import numpy as np
import pytest

import oracle_lib as o

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import p3_vector_format as F  # noqa: E402
from raiko_amd import hal as H, p3  # noqa: E402

VEC_DIR = os.path.join(HERE, "golden", "plonky3-88ea2b8")
HAVE = bool(glob.glob(os.path.join(VEC_DIR, "*.bin")))
need_vectors = pytest.mark.skipif(not HAVE, reason="no plonky3-88ea2b8 vector files (provers/hip/vectors-p3 has not been run: no Rust toolchain here)")
P = o.P


def instance_overrides(pos):
    return dict(p2_rc_ext=pos["rc_ext"], p2_rc_int=pos["rc_int"], p2_diag=pos["diag"])


# ---------------------------------------------------------------------------------------------- back ends
class OracleBackend:
    """the CPU restatement (oracle/) under the SP1 preset with the file's Poseidon2 instance"""

    def __init__(self, pos, **over):
        self.lib = o.oracle()
        o.oracle_set_params(1, **instance_overrides(pos), **over)

    def close(self):
        o.oracle_set_params()

    def permute(self, x):
        s = x.copy()
        self.lib.or_poseidon2_mix(o.ptr(s))
        return s

    def hash_row(self, row):
        out = np.zeros(8, dtype=np.uint32)
        self.lib.or_hash_elem_slice(o.ptr(np.ascontiguousarray(row)), row.size, 1, o.ptr(out))
        return out

    def compress(self, a, b):
        out = np.zeros(8, dtype=np.uint32)
        self.lib.or_hash_pair(o.ptr(a), o.ptr(b), o.ptr(out))
        return out

    def coset_lde_rows(self, trace):
        h, w = trace.shape
        out = np.zeros((2 * h, w), dtype=np.uint32)
        self.lib.or_pcs_coset_lde_rows(o.ptr(out), o.ptr(np.ascontiguousarray(trace)), h, w)
        return out

    def fold(self, evals, beta):
        n = evals.shape[0]
        out = np.zeros((n // 2, 4), dtype=np.uint32)
        self.lib.or_fri_fold_evals(o.ptr(out), o.ptr(np.ascontiguousarray(evals)), n // 2, o.ptr(np.ascontiguousarray(beta)))
        return out

    def prove_fib(self, trace, public):
        return o.oracle_p3_prove([p3.Table(p3.fibonacci_air(), trace, public)])


class GpuBackend:
    """the product through the C ABI (rk_mmcs_* for the hashing, rk_pcs_*, rk_fri_fold_evals, rk_p3_prove)"""

    def __init__(self, pos, **over):
        self.hal = H.HipHal(0)
        self.blob = self.hal.set_params(1, **instance_overrides(pos), **over)

    def close(self):
        self.hal.close()

    def permute(self, x):
        # one compression of (x[0..8), x[8..16)) shows the first 8 output cells; the full state through the sponge: a row of
        # 16 words is two absorbed blocks -- not the bare permutation.  The bare permutation is what the Poseidon2 chip's rows
        # hold: its last external round's state is the permutation's output
        d_rows, width = p3.poseidon2_chip_trace(self.hal, x.reshape(1, 16))
        return d_rows.to_host().reshape(width)[width - 17: width - 1]

    def hash_row(self, row):
        m = self.hal.copy_from_elem(np.ascontiguousarray(np.stack([row, row], axis=1)))     # column-major, two equal rows
        out = self.hal.alloc_elem(16)
        self.hal.hash_rows(out, m, 2, row.size)
        d = out.to_host().reshape(2, 8)
        assert np.array_equal(d[0], d[1])
        return d[0]

    def compress(self, a, b):
        nodes = self.hal.copy_from_elem(np.concatenate([np.zeros(16, dtype=np.uint32), a, b]))  # heap layout: children 2, 3 -> node 1
        self.hal.hash_fold(nodes, 2, 1)
        return nodes.to_host().reshape(4, 8)[1]

    def coset_lde_rows(self, trace):
        h, w = trace.shape
        out = self.hal.alloc_elem(2 * h * w)
        self.hal.pcs_coset_lde_rows(out, self.hal.copy_from_elem(trace), h, w)
        return out.to_host().reshape(2 * h, w)

    def fold(self, evals, beta):
        n = evals.shape[0]
        out = self.hal.alloc_elem(2 * n)
        self.hal.fri_fold_evals(out, self.hal.copy_from_elem(evals), n // 2, beta)
        return out.to_host().reshape(n // 2, 4)

    def prove_fib(self, trace, public):
        return p3.prove(self.hal, [p3.Table(p3.fibonacci_air(), trace, public)])


# ---------------------------------------------------------------------------------------------- the comparisons
def check_poseidon2(be, pos):
    for x, y in pos["kat"]:
        assert np.array_equal(be.permute(x), y)
    assert np.array_equal(be.hash_row(pos["row"]), pos["row_digest"])
    assert np.array_equal(be.compress(pos["left"], pos["right"]), pos["compressed"])


def check_pcs(be, pcs):
    assert np.array_equal(be.coset_lde_rows(pcs["trace"]), pcs["lde"])
    assert np.array_equal(be.fold(pcs["evals"], pcs["beta"]), pcs["folded"])


def check_fib(be, fib):
    got = be.prove_fib(fib["trace"], fib["public"])
    assert got.size == fib["proof"].size and np.array_equal(got, fib["proof"])


def fib_overrides(fib):
    return dict(queries=fib["queries"], pow_bits=fib["pow_bits"], blowup_log2=fib["log_blowup"])


# ---------------------------------------------------------------------------------------------- files made from the oracle
def write_from_oracle(d):
    """the three files in the Rust program's format, every answer computed by the oracle under the SP1 preset"""
    import p2_chip_ref as R
    rng = np.random.default_rng(0x7033)
    o.oracle_set_params(1)
    rc_ext, rc_int, diag, _ = R.tables_of()
    pos = {"rc_ext": o.to_mont(rc_ext.reshape(-1)), "rc_int": o.to_mont(rc_int), "diag": o.to_mont(diag)}
    be = OracleBackend(pos)
    kat = [o.rand_elems(rng, 16) for _ in range(3)]
    row, left, right = o.rand_elems(rng, 37), o.rand_elems(rng, 8), o.rand_elems(rng, 8)
    F.write(os.path.join(d, "p3_poseidon2.bin"), 5, [[16, 13], pos["rc_ext"], pos["rc_int"], pos["diag"]] + [np.concatenate([x, be.permute(x)]) for x in kat]
            + [[row.size], row, be.hash_row(row), left, right, be.compress(left, right)])
    trace = o.rand_elems(rng, (64, 5))
    evals, beta = o.rand_elems(rng, (32, 4)), o.rand_elems(rng, 4)
    F.write(os.path.join(d, "p3_pcs.bin"), 6, [[6, 5], trace, be.coset_lde_rows(trace), [32], beta, evals, be.fold(evals, beta)])
    be.close()
    tr, pub = p3.fibonacci_trace(6, 0, 1)
    tr, pub = p3.to_mont(tr), p3.to_mont(np.array(pub, dtype=np.uint64))
    be = OracleBackend(pos, queries=8, pow_bits=6, blowup_log2=1)
    F.write(os.path.join(d, "p3_fib_proof.bin"), 7, [[6, 8, 6, 1], pub, tr, [0]])          # placeholder, then the real proof
    proof = be.prove_fib(tr, pub)
    F.write(os.path.join(d, "p3_fib_proof.bin"), 7, [[6, 8, 6, 1], pub, tr, [proof.size], proof])
    be.close()


def run_all(d, make_backend):
    pos = F.read_poseidon2(os.path.join(d, "p3_poseidon2.bin"))
    assert pos["width"] == 16 and pos["rp"] == 13
    be = make_backend(pos)
    try:
        check_poseidon2(be, pos)
        check_pcs(be, F.read_pcs(os.path.join(d, "p3_pcs.bin")))
    finally:
        be.close()
    fib = F.read_fib_proof(os.path.join(d, "p3_fib_proof.bin"))
    be = make_backend(pos, **fib_overrides(fib))
    try:
        check_fib(be, fib)
    finally:
        be.close()


def test_machinery_on_files_made_from_the_oracle(tmp_path):
    d = str(tmp_path)
    write_from_oracle(d)
    run_all(d, OracleBackend)
    # one changed word anywhere is noticed
    for name, at in (("p3_poseidon2.bin", 2 + 2 + 128 + 13 + 16 + 20), ("p3_pcs.bin", 2 + 2 + 320 + 7), ("p3_fib_proof.bin", 2 + 4 + 3 + 128 + 1 + 40)):
        path = os.path.join(d, name)
        w = np.fromfile(path, dtype="<u4")
        w[at] = (int(w[at]) + 1) % P
        w.tofile(path)
        with pytest.raises(AssertionError):
            run_all(d, OracleBackend)
        w[at] = (int(w[at]) - 1) % P
        w.tofile(path)
    run_all(d, OracleBackend)
    with pytest.raises(ValueError):
        F.read_pcs(os.path.join(d, "p3_poseidon2.bin"))           # the kind word is checked


@need_vectors
def test_reference_vectors_against_the_oracle():
    run_all(VEC_DIR, OracleBackend)


@pytest.mark.gpu
def test_gpu_on_files_made_from_the_oracle(tmp_path):
    d = str(tmp_path)
    write_from_oracle(d)
    run_all(d, GpuBackend)


@pytest.mark.gpu
@need_vectors
def test_reference_vectors_against_the_gpu_path():
    run_all(VEC_DIR, GpuBackend)

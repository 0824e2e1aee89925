"""The route from "parity unpinned" to "pinned against risc0 itself" (DESIGN.md section 1).

`provers/hip/vectors` is a small Rust program against the crates raiko pins (risc0-zkp / risc0-core 1.0.1, reference
Cargo.lock:7243, :7171; the path `session.prove()` of provers/risc0/driver/src/bonsai.rs:271 runs through them).  It
cannot be built in this image (no Rust toolchain, no network).  Run anywhere else, it writes four files into
tests/golden/risc0-1.0.1/: the Poseidon2 permutation / compression / row sponge, interpolate + zk-shift + expansion of
seeded columns, the Fiat-Shamir generator's outputs after fixed commits, and one whole seal of a synthetic 2^10-cycle
circuit.  While that directory holds no .bin file the two `reference` tests SKIP; once it does, the CPU oracle and --
under -m gpu -- the HIP path through the C ABI are compared with every vector, bit for bit.

What runs today: the same comparison code on files of the same format made from the oracle itself (with the Rust
program's input generator), so the reader, the checks and the failure on a single changed word are exercised."""
import glob
import os
import sys

import numpy as np
import pytest

import oracle_lib as o

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import risc0_vector_format as F  # noqa: E402
from raiko_amd.segment import Segment, TapSet  # noqa: E402

VEC_DIR = os.path.join(HERE, "golden", "risc0-1.0.1")
HAVE = bool(glob.glob(os.path.join(VEC_DIR, "*.bin")))
need_vectors = pytest.mark.skipif(not HAVE, reason="no risc0-1.0.1 vector files (provers/hip/vectors has not been run: no Rust toolchain here)")


# ---------------------------------------------------------------------------------------------- back ends
class OracleBackend:
    """the CPU restatement (oracle/)"""

    def __init__(self):
        self.lib = o.oracle()

    def mix(self, cells):
        s = cells.copy()
        self.lib.or_poseidon2_mix(o.ptr(s))
        return s

    def hash_pair(self, a, b):
        out = np.zeros(8, dtype=np.uint32)
        self.lib.or_hash_pair(o.ptr(a), o.ptr(b), o.ptr(out))
        return out

    def hash_rows(self, matrix):
        cols, rows = matrix.shape
        out = np.zeros((rows, 8), dtype=np.uint32)
        self.lib.or_hash_rows(o.ptr(out), o.ptr(np.ascontiguousarray(matrix)), rows, cols)
        return out

    def ntt_chain(self, evals):
        count, n = evals.shape
        io = np.ascontiguousarray(evals.copy())
        self.lib.or_batch_interpolate_ntt(o.ptr(io), n, count)
        coeffs = io.copy()
        self.lib.or_zk_shift(o.ptr(io), n, count)
        shifted = io.copy()
        wide = np.zeros((count, 4 * n), dtype=np.uint32)
        self.lib.or_batch_expand_into_evaluate_ntt(o.ptr(wide), o.ptr(io), n, count, 2)
        return coeffs, shifted, wide

    def prove(self, seg):
        return o.oracle_prove(seg)


class HipBackend:
    """the product, through the C ABI (GPU)"""

    def __init__(self, hal):
        self.hal = hal

    def mix(self, cells):
        return None          # the bare permutation is not an entry point of the ABI: covered by hash_pair / hash_rows

    def hash_pair(self, a, b):
        nodes = self.hal.copy_from_elem(np.concatenate([np.zeros(16, dtype=np.uint32), a, b]))   # heap: children of node 1 at 2, 3
        self.hal.hash_fold(nodes, 2, 1)
        return nodes.to_host()[8:16]

    def hash_rows(self, matrix):
        cols, rows = matrix.shape
        out = self.hal.alloc_elem(rows * 8)
        self.hal.hash_rows(out, self.hal.copy_from_elem(matrix), rows, cols)
        return out.to_host().reshape(rows, 8)

    def ntt_chain(self, evals):
        count, n = evals.shape
        io = self.hal.copy_from_elem(evals)
        self.hal.batch_interpolate_ntt(io, count)
        coeffs = io.to_host().reshape(count, n)
        self.hal.zk_shift(io, count)
        shifted = io.to_host().reshape(count, n)
        wide = self.hal.alloc_elem(count * 4 * n)
        self.hal.batch_expand_into_evaluate_ntt(wide, io, count, 2)
        return coeffs, shifted, wide.to_host().reshape(count, 4 * n)

    def prove(self, seg):
        return self.hal.prove_segment(seg)


# ---------------------------------------------------------------------------------------------- the comparisons
def check_poseidon2(v, be):
    for cin, cout in v["mix"]:
        got = be.mix(cin)
        if got is not None:
            assert np.array_equal(got, cout), "poseidon2_mix differs from risc0's"
    assert np.array_equal(be.hash_pair(v["a"], v["b"]), v["ab"]), "hash_pair differs from risc0's"
    assert np.array_equal(be.hash_rows(v["matrix"]), v["digests"]), "hash_rows differs from risc0's"


def check_ntt(v, be):
    coeffs, shifted, wide = be.ntt_chain(v["evals"])
    assert np.array_equal(coeffs, v["coeffs"]), "batch_interpolate_ntt differs from risc0's"
    assert np.array_equal(shifted, v["shifted"]), "zk_shift differs from risc0's"
    assert np.array_equal(wide, v["expanded"]), "batch_expand_into_evaluate_ntt differs from risc0's"


def rng_outputs(d1, d2):
    """the transcript generator of the oracle (risc0-zkp core/hash/poseidon2/rng.rs restated): the sequence of rng.bin"""
    lib = o.oracle()
    iop = o.OrIop()
    lib.or_iop_init(iop)
    lib.or_iop_commit(iop, o.ptr(np.ascontiguousarray(d1)))
    bits20 = [lib.or_iop_random_bits(iop, 20) for _ in range(4)]
    elems = [lib.or_iop_random_elem(iop) for _ in range(4)]
    lib.or_iop_commit(iop, o.ptr(np.ascontiguousarray(d2)))
    ext = [lib.or_iop_random_elem(iop) for _ in range(4)]
    bits10 = lib.or_iop_random_bits(iop, 10)
    lib.or_iop_free(iop)
    return dict(bits20=np.array(bits20, dtype=np.uint32), elems=np.array(elems, dtype=np.uint32), ext=np.array(ext, dtype=np.uint32), bits10=bits10)


def check_rng(v):
    got = rng_outputs(v["d1"], v["d2"])
    for k in ("bits20", "elems", "ext"):
        assert np.array_equal(got[k], v[k]), "Fiat-Shamir generator: %s differs from risc0's" % k
    assert got["bits10"] == v["bits10"]


def segment_of(v):
    taps = TapSet(group_size=tuple(v["group_size"]), reg_group=v["reg_group"], reg_offset=v["reg_offset"], reg_combo=v["reg_combo"],
                  combo_off=v["combo_off"], combo_backs=v["combo_backs"])
    return Segment(po2=v["po2"], taps=taps, groups=[np.ascontiguousarray(g) for g in v["groups"]], check=np.ascontiguousarray(v["check"]),
                   globals_=v["globals"], n_accum_mix=v["n_accum_mix"], proof_system_info=v["proof_system_info"], circuit_info=v["circuit_info"])


def check_seal(v, be):
    got = be.prove(segment_of(v))
    assert got.size == v["seal"].size, "seal length differs from risc0's (%d vs %d words)" % (got.size, v["seal"].size)
    diff = np.nonzero(got != v["seal"])[0]
    assert diff.size == 0, "seal differs from risc0's from word %d on" % int(diff[0])


def check_dir(d, be, with_rng):
    check_poseidon2(F.read_poseidon2(os.path.join(d, "poseidon2.bin")), be)
    check_ntt(F.read_ntt(os.path.join(d, "ntt.bin")), be)
    if with_rng:
        check_rng(F.read_rng(os.path.join(d, "rng.bin")))
    check_seal(F.read_seal(os.path.join(d, "seal.bin")), be)


# ---------------------------------------------------------------------------------------------- tests
@need_vectors
def test_oracle_equals_risc0_reference_vectors():
    check_dir(VEC_DIR, OracleBackend(), with_rng=True)


@need_vectors
@pytest.mark.gpu
def test_gpu_equals_risc0_reference_vectors():
    from raiko_amd.hal import HipHal
    hal = HipHal(0)
    try:
        check_dir(VEC_DIR, HipBackend(hal), with_rng=False)     # the generator is host code of both sides: see the oracle test
    finally:
        hal.close()


def make_files_from_the_oracle(d):
    """files of the vector format with the Rust program's inputs and the ORACLE's outputs"""
    be = OracleBackend()
    lib = o.oracle()
    mont = lambda n, salt: o.to_mont(np.array([F.elem_canon(i, salt) for i in range(n)], dtype=np.uint64))

    def hash_elems(x):
        out = np.zeros(8, dtype=np.uint32)
        lib.or_hash_elem_slice(o.ptr(np.ascontiguousarray(x)), x.size, 1, o.ptr(out))
        return out
    a, b = hash_elems(mont(5, 7)), hash_elems(mont(9, 8))
    matrix = mont(8 * 40, 9).reshape(40, 8)
    F.write_poseidon2(os.path.join(d, "poseidon2.bin"),
                      dict(mix=[(c, be.mix(c)) for c in (mont(24, 1000 * (k + 1)) for k in range(3))], a=a, b=b, ab=be.hash_pair(a, b),
                           rows=8, cols=40, matrix=matrix, digests=be.hash_rows(matrix)))
    evals = mont(3 << 10, 21).reshape(3, 1 << 10)
    coeffs, shifted, wide = be.ntt_chain(evals)
    F.write_ntt(os.path.join(d, "ntt.bin"), dict(k=10, count=3, evals=evals, coeffs=coeffs, shifted=shifted, expanded=wide))
    d1, d2 = hash_elems(mont(3, 31)), hash_elems(mont(4, 32))
    F.write_rng(os.path.join(d, "rng.bin"), dict(d1=d1, d2=d2, **rng_outputs(d1, d2)))
    n, gs = 1 << 10, [4, 4, 8]
    from raiko_amd.segment import make_tapset
    taps = make_tapset([[(0, 1)] * 4, [(0,)] * 4, [((0, 1) if c < 2 else (0,)) for c in range(8)]])
    v = dict(po2=10, n_accum_mix=5, group_size=gs, reg_group=taps.reg_group, reg_offset=taps.reg_offset, reg_combo=taps.reg_combo,
             combo_off=taps.combo_off, combo_backs=taps.combo_backs, proof_system_info=b"RISC0_STARK:v1__", circuit_info=b"RKVECTOR:v1_____",
             globals=mont(6, 41), groups=[mont(n * gs[g], 42 + g).reshape(gs[g], n) for g in range(3)], check=mont(16 * n, 45).reshape(4, 4 * n))
    v["seal"] = be.prove(segment_of(dict(v, seal=None)))
    F.write_seal(os.path.join(d, "seal.bin"), v)


def test_the_comparison_machinery_on_oracle_made_files(tmp_path):
    d = str(tmp_path)
    make_files_from_the_oracle(d)
    check_dir(d, OracleBackend(), with_rng=True)
    # one changed word in any file is reported
    for name, reader, checker in (("poseidon2.bin", F.read_poseidon2, lambda v: check_poseidon2(v, OracleBackend())),
                                  ("ntt.bin", F.read_ntt, lambda v: check_ntt(v, OracleBackend())),
                                  ("rng.bin", F.read_rng, check_rng),
                                  ("seal.bin", F.read_seal, lambda v: check_seal(v, OracleBackend()))):
        path = os.path.join(d, name)
        w = np.fromfile(path, dtype="<u4")
        w[-1] = (int(w[-1]) + 1) % F.P
        w.tofile(path)
        with pytest.raises(AssertionError):
            checker(reader(path))
    with pytest.raises(ValueError):
        F.read_ntt(os.path.join(d, "rng.bin"))               # kind mismatch


@pytest.mark.gpu
def test_the_gpu_side_of_the_comparison_on_oracle_made_files(tmp_path):
    from raiko_amd.hal import HipHal
    d = str(tmp_path)
    make_files_from_the_oracle(d)
    hal = HipHal(0)
    try:
        check_dir(d, HipBackend(hal), with_rng=False)
    finally:
        hal.close()

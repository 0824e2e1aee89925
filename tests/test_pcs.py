"""The data-parallel steps of Plonky3's two-adic FRI PCS (oracle/or_pcs.c) against big-integer algebra.

What the three functions return is fixed by algebra once the orderings are -- the LDE is the interpolating polynomial's
values on the shifted coset in bit-reversed row order, the opened values are its values at z, a reduced opening is
sum_k alpha^k (p_k(x) - p_k(z)) / (x - z) -- so they are checked against direct O(n^2) evaluation with Python integers,
under risc0's and SP1's field / blow-up parameters, and against each other: a reduced opening folded with
or_fri_fold_evals down to `blowup` values must be constant (the assertion at the end of Plonky3's commit phase)."""
import numpy as np
import pytest

import oracle_lib as o

P = o.P


@pytest.fixture()
def params():
    yield o.oracle_set_params
    o.oracle_set_params()


def ext_mul(x, y, w):
    r = [0] * 7
    for i in range(4):
        for j in range(4):
            r[i + j] += x[i] * y[j]
    for k in (6, 5, 4):
        r[k - 4] += w * r[k]
    return [v % P for v in r[:4]]


def ext_pow(x, e, w):
    r = [1, 0, 0, 0]
    while e:
        if e & 1:
            r = ext_mul(r, x, w)
        x = ext_mul(x, x, w)
        e >>= 1
    return r


def ext_inv(x, w):
    # x^(p^4 - 2)
    return ext_pow(x, P ** 4 - 2, w)


def bitrev(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2) if bits else 0


def canon(a):
    return o.from_mont(np.asarray(a, dtype=np.uint32)).astype(object)


def setup(params, preset, k, w, seed):
    p = params(preset)
    rng = np.random.default_rng(seed)
    h = 1 << k
    ev = o.rand_elems(rng, (h, w))
    root, shift, blow, W = int(p.root_2_27), int(p.coset_shift), int(p.blowup_log2), int(p.ext_w)
    g = pow(root, 1 << (27 - k), P)
    evc = canon(ev)
    # coefficients of the interpolating polynomials: inverse DFT with integers
    hinv = pow(h, P - 2, P)
    coeff = [[sum(int(evc[j][c]) * pow(g, (-i * j) % h, P) for j in range(h)) * hinv % P for i in range(h)] for c in range(w)]
    return ev, coeff, root, shift, blow, W, h


def lde_of(ev, h, w, blow):
    orc = o.oracle()
    out = np.zeros(((h << blow), w), dtype=np.uint32)
    orc.or_pcs_coset_lde_rows(o.ptr(out), o.ptr(np.ascontiguousarray(ev)), h, w)
    return out


@pytest.mark.parametrize("preset,k,w", [(0, 3, 3), (1, 3, 2), (1, 4, 5), (0, 1, 1)])
def test_coset_lde_rows_is_the_polynomial_on_the_coset(params, preset, k, w):
    ev, coeff, root, shift, blow, _, h = setup(params, preset, k, w, 11 + k)
    out = canon(lde_of(ev, h, w, blow))
    kb = k + blow
    gK = pow(root, 1 << (27 - kb), P)
    for r in range(h << blow):
        x = shift * pow(gK, bitrev(r, kb), P) % P
        for c in range(w):
            assert int(out[r][c]) == sum(coeff[c][i] * pow(x, i, P) for i in range(h)) % P


@pytest.mark.parametrize("preset,k,w", [(0, 3, 3), (1, 4, 4)])
def test_eval_at_and_reduced_opening(params, preset, k, w):
    ev, coeff, root, shift, blow, W, h = setup(params, preset, k, w, 23 + k)
    orc = o.oracle()
    lde = lde_of(ev, h, w, blow)
    H, kb = h << blow, k + blow
    rng = np.random.default_rng(5)
    zs = o.rand_elems(rng, (2, 4))
    alpha = o.rand_elems(rng, (4,))
    ys = np.zeros((2, w, 4), dtype=np.uint32)
    for j in range(2):
        orc.or_pcs_eval_at(o.ptr(ys[j]), o.ptr(lde), H, w, o.ptr(zs[j]))
    zc, ac, yc = canon(zs), [int(v) for v in canon(alpha)], canon(ys)
    for j in range(2):
        z = [int(v) for v in zc[j]]
        for c in range(w):
            want, zp = [0, 0, 0, 0], [1, 0, 0, 0]
            for i in range(h):
                want = [(a + coeff[c][i] * b) % P for a, b in zip(want, zp)]
                zp = ext_mul(zp, z, W)
            assert [int(v) for v in yc[j][c]] == want
    # reduced openings of both points on top of a non-zero start, with an offset into the powers of alpha
    ro0 = o.rand_elems(rng, (H, 4))
    ro = ro0.copy()
    offset = 7
    orc.or_pcs_reduce_openings(o.ptr(ro), o.ptr(lde), H, w, 2, o.ptr(zs), o.ptr(ys), o.ptr(alpha), offset)
    roc, ro0c, ldec = canon(ro), canon(ro0), canon(lde)
    gK = pow(root, 1 << (27 - kb), P)
    for r in range(H):
        x = shift * pow(gK, bitrev(r, kb), P) % P
        acc = [int(v) for v in ro0c[r]]
        for j in range(2):
            z = [int(v) for v in zc[j]]
            num, ap = [0, 0, 0, 0], ext_pow(ac, offset + j * w, W)
            for c in range(w):
                d = [(-int(v)) % P for v in yc[j][c]]
                d[0] = (d[0] + int(ldec[r][c])) % P
                num = [(a + b) % P for a, b in zip(num, ext_mul(ap, d, W))]
                ap = ext_mul(ap, ac, W)
            den = [(-v) % P for v in z]
            den[0] = (den[0] + x) % P
            acc = [(a + b) % P for a, b in zip(acc, ext_mul(num, ext_inv(den, W), W))]
        assert [int(v) for v in roc[r]] == acc


@pytest.mark.parametrize("preset", [0, 1])
def test_reduced_opening_folds_to_a_constant(params, preset):
    """commit -> open -> FRI commit phase, without the hashing: (p(x) - p(z)) / (x - z) has degree < h - 1, so
    log2(h) folds of its evaluations on the coset (or_fri_fold_evals, any betas) leave `blowup` equal values."""
    p = params(preset)
    orc = o.oracle()
    k, w = 6, 9
    h, blow = 1 << k, int(p.blowup_log2)
    H = h << blow
    rng = np.random.default_rng(77)
    lde = lde_of(o.rand_elems(rng, (h, w)), h, w, blow)
    z, alpha = o.rand_elems(rng, (1, 4)), o.rand_elems(rng, (4,))
    ys = np.zeros((1, w, 4), dtype=np.uint32)
    orc.or_pcs_eval_at(o.ptr(ys), o.ptr(lde), H, w, o.ptr(z))
    cur = np.zeros((H, 4), dtype=np.uint32)
    orc.or_pcs_reduce_openings(o.ptr(cur), o.ptr(lde), H, w, 1, o.ptr(z), o.ptr(ys), o.ptr(alpha), 0)
    while cur.shape[0] > (1 << blow):
        nxt = np.zeros((cur.shape[0] // 2, 4), dtype=np.uint32)
        orc.or_fri_fold_evals(o.ptr(nxt), o.ptr(cur), nxt.shape[0], o.ptr(o.rand_elems(rng, (4,))))
        cur = nxt
    assert cur.any()
    assert (cur == cur[0]).all()
    # a wrong opened value breaks it
    ys2 = ys.copy()
    ys2[0, 3, 1] = (int(ys2[0, 3, 1]) + 1) % P
    bad = np.zeros((H, 4), dtype=np.uint32)
    orc.or_pcs_reduce_openings(o.ptr(bad), o.ptr(lde), H, w, 1, o.ptr(z), o.ptr(ys2), o.ptr(alpha), 0)
    while bad.shape[0] > (1 << blow):
        nxt = np.zeros((bad.shape[0] // 2, 4), dtype=np.uint32)
        orc.or_fri_fold_evals(o.ptr(nxt), o.ptr(bad), nxt.shape[0], o.ptr(o.rand_elems(rng, (4,))))
        bad = nxt
    assert not (bad == bad[0]).all()


def test_oracle_matches_committed_digests():
    import json, os
    from pcs_cases import PCS_CASES, oracle_outputs
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "pcs_digests.json")))
    assert sorted(gold) == sorted(PCS_CASES)
    for key in PCS_CASES:
        assert oracle_outputs(key) == gold[key], key


class DuplexChallengerPy:
    """p3-challenger DuplexChallenger, literally (RECALLED): the specification or_duplex_grind is checked against"""

    def __init__(self, orc, width, state, inputs):
        self.orc, self.width, self.rate = orc, width, width - 8
        self.state = np.array(state, dtype=np.uint32).copy()
        self.inputs = [int(v) for v in inputs]
        self.outputs = []

    def clone(self):
        c = DuplexChallengerPy(self.orc, self.width, self.state, self.inputs)
        c.outputs = list(self.outputs)
        return c

    def duplexing(self):
        assert len(self.inputs) <= self.rate
        for i, v in enumerate(self.inputs):
            self.state[i] = v
        self.inputs = []
        buf = np.zeros(24, dtype=np.uint32)
        buf[: self.width] = self.state
        self.orc.or_poseidon2_mix(o.ptr(buf))
        self.state = buf[: self.width].copy()
        self.outputs = [int(v) for v in self.state[: self.rate]]

    def observe(self, v):
        self.outputs = []
        self.inputs.append(int(v))
        if len(self.inputs) == self.rate:
            self.duplexing()

    def sample(self):
        if self.inputs or not self.outputs:
            self.duplexing()
        return self.outputs.pop()

    def sample_bits(self, bits):
        return int(o.from_mont(np.array([self.sample()], dtype=np.uint32))[0]) & ((1 << bits) - 1)

    def check_witness(self, bits, w):
        self.observe(int(o.to_mont(np.array([w], dtype=np.uint64))[0]))
        return self.sample_bits(bits) == 0


@pytest.mark.parametrize("preset", [0, 1])
@pytest.mark.parametrize("n_input", [0, 3, "last"])
def test_duplex_grind_is_the_smallest_accepted_witness(params, preset, n_input):
    p = params(preset)
    orc = o.oracle()
    width = int(p.p2_width)
    n_in = width - 9 if n_input == "last" else n_input      # "last": the witness fills the rate
    rng = np.random.default_rng(40 + preset)
    state, inputs = o.rand_elems(rng, (width,)), o.rand_elems(rng, (max(n_in, 1),))[:n_in]
    bits = 7
    w = orc.or_duplex_grind(o.ptr(state), o.ptr(inputs if n_in else np.zeros(1, dtype=np.uint32)), n_in, bits)
    ch = DuplexChallengerPy(orc, width, state, inputs)
    assert ch.clone().check_witness(bits, w)
    assert not any(ch.clone().check_witness(bits, v) for v in range(w))

"""Plonky3's TwoAdicFriPcs + FRI, end to end, with the GPU doing the data-parallel steps and Python the protocol:
commit (rk_pcs_coset_lde_rows + rk_mmcs_commit, two batches, matrices of two heights), open (rk_pcs_eval_at,
rk_pcs_reduce_openings), FRI commit phase (rk_mmcs_commit of the pair matrices, rk_fri_fold_evals, the shorter reduced
opening joining when the sizes meet), proof of work (rk_duplex_grind), queries (rk_mmcs_open) -- then a verifier written
from the other side (p3-fri verifier.rs as RECALLED: reduced openings recomputed from the opened rows, the fold as the
line through (x, e) and (-x, e') evaluated at beta, Merkle paths checked with rk_mmcs_verify) replays the transcript and
accepts; with one opened value or one sibling changed it rejects.  The transcript is a literal Python DuplexChallenger.
This is the executable form of the binding sketched in INTEGRATION.md 5.1; SP1's parameter set."""
import numpy as np
import pytest

import oracle_lib as o
from raiko_amd import hal as H
from test_pcs import DuplexChallengerPy, bitrev, ext_inv, ext_mul, ext_pow

pytestmark = pytest.mark.gpu
P = o.P


def canon4(m):
    return [int(v) for v in o.from_mont(np.asarray(m, dtype=np.uint32))]


def mont4(c):
    return o.to_mont(np.array(c, dtype=np.uint64))


def sample_ext(ch):
    return np.array([ch.sample() for _ in range(4)], dtype=np.uint32)      # EF::from_base_fn(|_| sample())


def observe_all(ch, words):
    for v in np.asarray(words, dtype=np.uint32).reshape(-1):
        ch.observe(int(v))


def test_commit_open_fri_verify():
    h = H.HipHal(0)
    par = h.set_params(preset=1)
    o.oracle_set_params(1)
    try:
        run(h, par)
    finally:
        o.oracle_set_params()
        h.close()


def run(h, par):
    orc = o.oracle()
    blow, W, shift, root = int(par.blowup_log2), int(par.ext_w), int(par.coset_shift), int(par.root_2_27)
    pow_bits, n_queries = 10, 6
    gen = lambda k: pow(root, 1 << (27 - k), P)
    rng = np.random.default_rng(2024)
    # two batches; (log2 height of the trace, width, number of opening points)
    spec = [[(10, 5, 2), (7, 4, 1)], [(10, 3, 1)]]
    batches = []
    for b in spec:
        mats = []
        for k, w, npts in b:
            n, Hh = 1 << k, (1 << k) << blow
            lde = h.alloc_elem(Hh * w)
            h.pcs_coset_lde_rows(lde, h.copy_from_elem(o.rand_elems(rng, (n, w))), n, w)
            mats.append(dict(k=k, lh=k + blow, w=w, npts=npts, lde=lde, H=Hh))
        nodes, rt = h.mmcs_commit([(m["lde"], m["H"], m["w"], True) for m in mats])
        batches.append(dict(mats=mats, nodes=nodes, root=rt))
    log_max = max(m["lh"] for b in batches for m in b["mats"])

    # ---------------------------------------------------------------- prover
    ch = DuplexChallengerPy(orc, 16, np.zeros(16, dtype=np.uint32), [])
    for b in batches:
        observe_all(ch, b["root"])
    zeta = sample_ext(ch)
    for b in batches:
        for m in b["mats"]:
            g = int(o.to_mont(np.array([gen(m["k"])], dtype=np.uint64))[0])
            zg = np.array([orc.or_fp_mul(int(v), g) for v in zeta], dtype=np.uint32)
            m["points"] = np.stack([zeta, zg][: m["npts"]])
    alpha = sample_ext(ch)
    ro, num_reduced = {}, {}
    for b in batches:
        for m in b["mats"]:
            lh = m["lh"]
            if lh not in ro:
                ro[lh] = h.copy_from_elem(np.zeros((1 << lh, 4), dtype=np.uint32))
                num_reduced[lh] = 0
            m["opened"] = h.pcs_eval_at_many(m["lde"], m["H"], m["w"], m["points"])
            h.pcs_reduce_openings(ro[lh], m["lde"], m["H"], m["w"], m["points"], m["opened"], alpha, num_reduced[lh])
            num_reduced[lh] += m["w"] * m["npts"]
    # FRI commit phase
    folded, n = ro[log_max], 1 << log_max
    layers = []
    while n > (1 << blow):
        nodes, rt = h.mmcs_commit([(folded, n // 2, 8, True)])
        observe_all(ch, rt)
        beta = sample_ext(ch)
        nxt = h.alloc_elem(n // 2 * 4)
        h.fri_fold_evals(nxt, folded, n // 2, beta)
        layers.append(dict(buf=folded, nodes=nodes, root=rt, n=n))
        folded, n = nxt, n // 2
        lg = n.bit_length() - 1
        if lg in ro and lg != log_max:
            h.eltwise_add_elem(folded, folded, ro[lg], n * 4)
    fin = folded.to_host().reshape(n, 4)
    assert (fin == fin[0]).all()                    # `blowup` evaluations of a constant polynomial
    final_poly = fin[0].copy()
    observe_all(ch, final_poly)
    witness = h.duplex_grind(ch.state, np.array(ch.inputs, dtype=np.uint32), pow_bits)
    assert ch.check_witness(pow_bits, witness)
    queries = []
    for _ in range(n_queries):
        index = ch.sample_bits(log_max)
        inp = []
        for b in batches:
            lb = max(m["lh"] for m in b["mats"])
            inp.append(h.mmcs_open([(m["lde"], m["H"], m["w"], True) for m in b["mats"]], b["nodes"], index >> (log_max - lb)))
        steps = []
        for i, L in enumerate(layers):
            idx = index >> i
            rows, path = h.mmcs_open([(L["buf"], L["n"] // 2, 8, True)], L["nodes"], idx >> 1)
            steps.append((rows[4 * ((idx ^ 1) & 1): 4 * ((idx ^ 1) & 1) + 4].copy(), path))
        queries.append((inp, steps))
    proof = dict(roots=[b["root"] for b in batches], opened=[[m["opened"] for m in b["mats"]] for b in batches],
                 commits=[L["root"] for L in layers], final_poly=final_poly, witness=witness, queries=queries)

    # ---------------------------------------------------------------- verifier
    shape = [[(m["lh"], m["w"], m["npts"], m["k"]) for m in b["mats"]] for b in batches]

    def verify(pf):
        vc = DuplexChallengerPy(orc, 16, np.zeros(16, dtype=np.uint32), [])
        for r in pf["roots"]:
            observe_all(vc, r)
        z = canon4(sample_ext(vc))
        al = canon4(sample_ext(vc))
        betas = []
        for r in pf["commits"]:
            observe_all(vc, r)
            betas.append(canon4(sample_ext(vc)))
        observe_all(vc, pf["final_poly"])
        if not vc.check_witness(pow_bits, pf["witness"]):
            return "pow"
        for inp, steps in pf["queries"]:
            index = vc.sample_bits(log_max)
            rop, apow = {}, {}
            for bi, (rows_path, bshape) in enumerate(zip(inp, shape)):
                rows, path = rows_path
                lb = max(s[0] for s in bshape)
                if H.mmcs_verify([1 << s[0] for s in bshape], [s[1] for s in bshape], index >> (log_max - lb), rows, path,
                                 pf["roots"][bi], params=par) != 0:
                    return "input opening"
                at = 0
                for mi, (lh, w, npts, k) in enumerate(bshape):
                    row = [int(v) for v in o.from_mont(rows[at:at + w])]
                    at += w
                    x = shift * pow(gen(lh), bitrev(index >> (log_max - lh), lh), P) % P
                    rop.setdefault(lh, [0, 0, 0, 0])
                    apow.setdefault(lh, [1, 0, 0, 0])
                    for j in range(npts):
                        zj = z if j == 0 else [v * gen(k) % P for v in z]
                        den = list(zj)
                        den[0] = (den[0] - x) % P
                        inv = ext_inv(den, W)
                        for c in range(w):
                            num = canon4(pf["opened"][bi][mi][j][c])
                            num[0] = (num[0] - row[c]) % P                      # p(z) - p(x)
                            term = ext_mul(apow[lh], ext_mul(num, inv, W), W)
                            rop[lh] = [(a + t) % P for a, t in zip(rop[lh], term)]
                            apow[lh] = ext_mul(apow[lh], al, W)
            folded_eval = [0, 0, 0, 0]
            x = pow(gen(log_max), bitrev(index, log_max), P)
            idx = index
            for i, (sib, path) in enumerate(steps):
                lfh = log_max - 1 - i
                if lfh + 1 in rop:
                    folded_eval = [(a + t) % P for a, t in zip(folded_eval, rop[lfh + 1])]
                evals = [None, None]
                evals[idx & 1], evals[(idx ^ 1) & 1] = folded_eval, canon4(sib)
                pair = np.concatenate([mont4(evals[0]), mont4(evals[1])]).astype(np.uint32)
                if H.mmcs_verify([1 << lfh], [8], idx >> 1, pair, path, pf["commits"][i], params=par) != 0:
                    return "commit-phase opening"
                xs = [x, x]
                xs[(idx ^ 1) & 1] = (-x) % P
                slope = [(b - a) % P for a, b in zip(evals[0], evals[1])]
                slope = [v * pow((xs[1] - xs[0]) % P, P - 2, P) % P for v in slope]
                bm = list(betas[i])
                bm[0] = (bm[0] - xs[0]) % P
                folded_eval = [(a + t) % P for a, t in zip(evals[0], ext_mul(bm, slope, W))]
                idx >>= 1
                x = x * x % P
            if folded_eval != canon4(pf["final_poly"]):
                return "final polynomial"
        return "ok"

    assert verify(proof) == "ok"
    bad = dict(proof)
    bad["opened"] = [[a.copy() for a in b] for b in proof["opened"]]
    bad["opened"][0][1][0][2][1] = (int(bad["opened"][0][1][0][2][1]) + 1) % P          # batch 0, short matrix, one cell
    assert verify(bad) in ("commit-phase opening", "final polynomial")      # the recomputed evaluation no longer hashes to the layer
    bad = dict(proof)
    q0 = proof["queries"][0]
    sib = q0[1][3][0].copy()
    sib[0] = (int(sib[0]) + 1) % P
    bad["queries"] = [(q0[0], q0[1][:3] + [(sib, q0[1][3][1])] + q0[1][4:])] + proof["queries"][1:]
    assert verify(bad) == "commit-phase opening"
    bad = dict(proof)
    bad["witness"] = proof["witness"] + 1
    assert verify(bad) in ("pow", "input opening", "commit-phase opening", "final polynomial")

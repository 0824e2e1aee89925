"""Seeded cases of the PCS steps shared by the golden generator, the CPU test (oracle == golden) and the GPU test
(rk_pcs_* == golden): (preset, log2 height, width, points)."""
import hashlib

import numpy as np

import oracle_lib as o

PCS_CASES = {
    "risc0_k8_w5_p2": (0, 8, 5, 2),
    "risc0_k11_w70_p1": (0, 11, 70, 1),
    "sp1_k8_w5_p2": (1, 8, 5, 2),
    "sp1_k12_w33_p3": (1, 12, 33, 3),
}


def pcs_inputs(key):
    preset, k, w, npts = PCS_CASES[key]
    rng = np.random.default_rng(abs(hash_str(key)) % (1 << 31))
    ev = o.rand_elems(rng, (1 << k, w))
    zs = o.rand_elems(rng, (npts, 4))
    alpha = o.rand_elems(rng, (4,))
    # the reduced opening's starting value is sized by the preset's blow-up: the caller draws it from the same generator
    return preset, k, w, npts, ev, zs, alpha, rng


def hash_str(s):
    return int.from_bytes(hashlib.sha256(s.encode()).digest()[:4], "little")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint32).tobytes()).hexdigest()


def oracle_outputs(key):
    """-> dict of sha256 digests of the LDE, the opened values and the reduced opening, from the oracle"""
    preset, k, w, npts, ev, zs, alpha, rng = pcs_inputs(key)
    p = o.oracle_set_params(preset)
    try:
        orc = o.oracle()
        h, H = 1 << k, (1 << k) << int(p.blowup_log2)
        lde = np.zeros((H, w), dtype=np.uint32)
        orc.or_pcs_coset_lde_rows(o.ptr(lde), o.ptr(ev), h, w)
        ys = np.zeros((npts, w, 4), dtype=np.uint32)
        for j in range(npts):
            orc.or_pcs_eval_at(o.ptr(ys[j]), o.ptr(lde), H, w, o.ptr(zs[j]))
        ro = o.rand_elems(rng, (H, 4))
        orc.or_pcs_reduce_openings(o.ptr(ro), o.ptr(lde), H, w, npts, o.ptr(zs), o.ptr(ys), o.ptr(alpha), 5)
        return {"lde": sha(lde), "opened": sha(ys), "reduced": sha(ro)}
    finally:
        o.oracle_set_params()

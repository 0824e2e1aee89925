"""Segments, tap sets and the synthetic "S20" workload of SURVEY.md section 8(d).

A *segment* is what raiko's risc0 driver proves one STARK for: the executor cuts
the guest run into pieces of at most 2^po2 cycles (`segment_limit_po2`, reference
provers/risc0/driver/src/bonsai.rs:246-250) and `session.prove()` (bonsai.rs:271)
proves each.  Witness generation and the rv32im constraint evaluator live in
crates that are absent from the reference tree, so a segment here is the data
those stages would hand to the prover: the three trace matrices (code / data /
accum register groups), the check polynomial evaluations, the globals and the
circuit's tap set.
"""
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

P = 2013265921
INV_RATE = 4
QUERIES = 50
FRI_FOLD = 16
FRI_MIN_DEGREE = 256
CHECK_SIZE = 16
EXT = 4

GROUP_ACCUM, GROUP_CODE, GROUP_DATA = 0, 1, 2

PROOF_SYSTEM_INFO = b"RISC0_STARK:v1__"
CIRCUIT_INFO = b"RV32IM:v1_______"


@dataclass
class TapSet:
    """risc0-zkp taps.rs `TapSet`: registers sorted by (group, offset), each with a combo of `back`s."""

    group_size: Tuple[int, int, int]
    reg_group: np.ndarray
    reg_offset: np.ndarray
    reg_combo: np.ndarray
    combo_off: np.ndarray
    combo_backs: np.ndarray

    @property
    def n_regs(self) -> int:
        return int(self.reg_group.shape[0])

    @property
    def n_combos(self) -> int:
        return int(self.combo_off.shape[0] - 1)

    def combo(self, c: int) -> np.ndarray:
        return self.combo_backs[self.combo_off[c]:self.combo_off[c + 1]]

    @property
    def tot_taps(self) -> int:
        sizes = np.diff(self.combo_off)
        return int(sizes[self.reg_combo].sum())


def make_tapset(backs_per_group: Sequence[Sequence[Sequence[int]]]) -> TapSet:
    """backs_per_group[g][col] = sorted tuple of backs of register `col` of group g (0 accum, 1 code, 2 data)."""
    combos = sorted({tuple(b) for g in backs_per_group for b in g})
    cid = {c: i for i, c in enumerate(combos)}
    rg, ro, rc = [], [], []
    for g, cols in enumerate(backs_per_group):
        for o, b in enumerate(cols):
            rg.append(g)
            ro.append(o)
            rc.append(cid[tuple(b)])
    off = np.zeros(len(combos) + 1, dtype=np.uint32)
    off[1:] = np.cumsum([len(c) for c in combos])
    backs = np.array([b for c in combos for b in c], dtype=np.uint32)
    return TapSet(
        group_size=tuple(len(g) for g in backs_per_group),
        reg_group=np.array(rg, dtype=np.uint32),
        reg_offset=np.array(ro, dtype=np.uint32),
        reg_combo=np.array(rc, dtype=np.uint32),
        combo_off=off,
        combo_backs=backs,
    )


def synthetic_tapset(w_accum: int = 16, w_code: int = 16, w_data: int = 224) -> TapSet:
    """Stand-in for the rv32im TapSet (not available offline): code registers are read at the
    current row only; every 4th data register also one row back, every 16th two rows back;
    accum registers at the current and previous row (running products)."""
    accum = [(0, 1)] * w_accum
    code = [(0,)] * w_code
    data = [((0, 1, 2) if c % 16 == 0 else (0, 1) if c % 4 == 0 else (0,)) for c in range(w_data)]
    return make_tapset([accum, code, data])


@dataclass
class Segment:
    po2: int
    taps: TapSet
    groups: List[np.ndarray]          # [accum, code, data], each uint32 (cols, 2^po2) C-order = column-major matrix
    check: np.ndarray                 # uint32 (4, 4 * 2^po2)
    globals_: np.ndarray              # uint32 (n_globals,)
    n_accum_mix: int = 40
    proof_system_info: bytes = PROOF_SYSTEM_INFO
    circuit_info: bytes = CIRCUIT_INFO
    # address of an rk_circuit_hooks (or a callable returning it): accum (groups[0]) and check are then
    # produced inside the proof by CircuitHal::accumulate / eval_check and may be None here
    hooks: object = None
    # a circuit_program.Program: eval_check then comes from the step list (hooks keeps `accumulate`)
    program: object = None

    @property
    def rows(self) -> int:
        return 1 << self.po2

    @property
    def cycles(self) -> int:
        return 1 << self.po2


def synthetic_segment(po2: int, widths: Tuple[int, int, int] = (16, 16, 224), seed: int = 20240807,
                      n_globals: int = 32, blowup_log2: int = 2) -> Segment:
    """One synthetic segment of 2^po2 cycles: i.i.d. uniform field elements (already in Montgomery
    form: the uniform distribution is invariant under the encoding), numpy PCG64 seeds
    seed + group index; check evaluations from seed + 3; globals from seed + 4.  `blowup_log2`: the
    check evaluations live on the LDE domain of 2^(po2 + blowup_log2) points (rk_params.blowup_log2)."""
    taps = synthetic_tapset(*widths)
    n = 1 << po2
    groups = []
    for g, w in enumerate(widths):
        rng = np.random.Generator(np.random.PCG64(seed + g))
        groups.append(rng.integers(0, P, size=(w, n), dtype=np.uint32))
    rng = np.random.Generator(np.random.PCG64(seed + 3))
    check = rng.integers(0, P, size=(EXT, n << blowup_log2), dtype=np.uint32)
    rng = np.random.Generator(np.random.PCG64(seed + 4))
    globals_ = rng.integers(0, P, size=(n_globals,), dtype=np.uint32)
    return Segment(po2=po2, taps=taps, groups=groups, check=check, globals_=globals_)


def algorithmic_bytes(po2: int, widths: Sequence[int], blowup_log2: int = 2, fold_log2: int = 4,
                      min_degree: int = FRI_MIN_DEGREE) -> dict:
    """Compulsory HBM traffic of one segment proof with every stage reading its input once and
    writing its output once ('unfused'): per group of W columns
    (9 + D/N)*N*W*4 + 4*D*32 (= 13*N*W*4 + ... at blow-up 4); check group as a 4*D/N-column group on top
    of a 4 x D iNTT; DEEP and FRI.  Defaults: risc0's shape (blow-up 4, fold 16, final degree 256)."""
    n = 1 << po2
    blow = 1 << blowup_log2
    fold = 1 << fold_log2
    d = blow * n
    chk = 4 * blow
    out = {}
    trace = 0
    for w in widths:
        trace += (9 + blow) * n * w * 4 + 4 * d * 32
    out["trace_groups"] = trace
    # check: iNTT over 4 x D (read+write), zk-shift, expand (N->D for the columns), hash rows + folds
    out["check_group"] = 2 * 4 * d * 4 + 2 * n * chk * 4 + (1 + blow) * n * chk * 4 + (d * chk * 4 + d * 32) + 3 * d * 32
    # DEEP: read every coefficient column once more (+ check), write/read combos ~ 4 ext polys
    out["deep"] = n * (sum(widths) + chk) * 4 + 2 * 4 * n * 16
    # FRI: rounds of (expand 4 planes, hash rows of 4*fold cols, folds), sizes N, N/fold, ...
    fri = 0
    size = n
    while size > min_degree and size >= fold:
        dom = size * blow
        fri += (size + dom) * 4 * 4 + dom * 4 * 4 + (dom // fold) * 32 * 4 + size * 4 * 4
        size //= fold
    out["fri"] = fri
    out["total"] = sum(out.values())
    return out


# VALU instructions of one Poseidon2 permutation as shipped (tools/census_p2.py on the gfx950 ISA of
# poseidon2_core.hpp: 6737 dynamic v_* instructions with all 24 output cells live -- 7159 with the
# closed-form partial rounds of round 1; kernels that keep only the digest / capacity cells run
# slightly fewer)
P2_VALU_PER_PERMUTATION = 6737


def poseidon2_permutations(po2: int, widths: Sequence[int]) -> dict:
    """Permutations of one segment proof: row hashing (one per 16 columns per LDE row: three trace
    groups, the 16-column check group, 64-column FRI rows) and Merkle folds (one per parent)."""
    n = 1 << po2
    d = INV_RATE * n
    rows = d * (sum((w + 15) // 16 for w in widths) + 1)
    fold = 4 * (d - 1)
    size = n
    while size > FRI_MIN_DEGREE:
        leaves = size * INV_RATE // FRI_FOLD
        rows += leaves * (FRI_FOLD * 4 // 16)
        fold += leaves - 1
        size //= FRI_FOLD
    return {"hash_rows": rows, "hash_fold": fold}

"""Layout of the risc0 known-answer vector files (provers/hip/vectors/src/main.rs writes them from the real risc0-zkp /
risc0-core 1.0.1 crates; tests/test_risc0_vectors.py compares the oracle and the GPU path with them).  Little-endian
u32 words; word 0 = MAGIC, word 1 = kind; field elements are Montgomery words.  `write_*` exist so that the test can
exercise its own comparison code on files made from the oracle when no real vectors are present."""
import numpy as np

MAGIC = 0x31564B52
P = 2013265921


def elem_canon(i, salt):
    """the Rust program's input generator: ((2654435761 * (i + 1) + salt) mod 2^32) mod p, canonical"""
    return ((2654435761 * (i + 1) + salt) & 0xFFFFFFFF) % P


class _R:
    def __init__(self, path, kind):
        self.w = np.fromfile(path, dtype="<u4")
        if self.w.size < 2 or int(self.w[0]) != MAGIC or int(self.w[1]) != kind:
            raise ValueError("%s: not a kind-%d vector file" % (path, kind))
        self.pos = 2

    def take(self, n):
        if self.pos + n > self.w.size:
            raise ValueError("vector file too short")
        out = self.w[self.pos:self.pos + n].astype(np.uint32)
        self.pos += n
        return out

    def one(self):
        return int(self.take(1)[0])

    def done(self):
        if self.pos != self.w.size:
            raise ValueError("trailing words in a vector file")


def _save(path, kind, parts):
    words = [np.array([MAGIC, kind], dtype=np.uint32)] + [np.asarray(p, dtype=np.uint32).reshape(-1) for p in parts]
    np.concatenate(words).astype("<u4").tofile(path)


def read_poseidon2(path):
    r = _R(path, 1)
    mix = [(r.take(24), r.take(24)) for _ in range(3)]
    a, b, ab = r.take(8), r.take(8), r.take(8)
    rows, cols = r.one(), r.one()
    matrix, digests = r.take(rows * cols).reshape(cols, rows), r.take(rows * 8).reshape(rows, 8)
    r.done()
    return dict(mix=mix, a=a, b=b, ab=ab, rows=rows, cols=cols, matrix=matrix, digests=digests)


def write_poseidon2(path, v):
    _save(path, 1, [x for pair in v["mix"] for x in pair] + [v["a"], v["b"], v["ab"], [v["rows"], v["cols"]], v["matrix"], v["digests"]])


def read_ntt(path):
    r = _R(path, 2)
    k, count = r.one(), r.one()
    n = 1 << k
    out = dict(k=k, count=count, evals=r.take(count * n).reshape(count, n), coeffs=r.take(count * n).reshape(count, n),
               shifted=r.take(count * n).reshape(count, n), expanded=r.take(count * n * 4).reshape(count, 4 * n))
    r.done()
    return out


def write_ntt(path, v):
    _save(path, 2, [[v["k"], v["count"]], v["evals"], v["coeffs"], v["shifted"], v["expanded"]])


def read_rng(path):
    r = _R(path, 3)
    out = dict(d1=r.take(8), d2=r.take(8), bits20=r.take(4), elems=r.take(4), ext=r.take(4), bits10=r.one())
    r.done()
    return out


def write_rng(path, v):
    _save(path, 3, [v["d1"], v["d2"], v["bits20"], v["elems"], v["ext"], [v["bits10"]]])


def read_seal(path):
    r = _R(path, 4)
    po2, n_globals, n_mix = r.one(), r.one(), r.one()
    gs = [r.one(), r.one(), r.one()]
    n_regs, n_combos, n_backs = r.one(), r.one(), r.one()
    out = dict(po2=po2, n_accum_mix=n_mix, group_size=gs, reg_group=r.take(n_regs), reg_offset=r.take(n_regs), reg_combo=r.take(n_regs),
               combo_off=r.take(n_combos + 1), combo_backs=r.take(n_backs),
               proof_system_info=r.take(4).astype("<u4").tobytes(), circuit_info=r.take(4).astype("<u4").tobytes(),
               globals=r.take(n_globals))
    n = 1 << po2
    out["groups"] = [r.take(n * gs[g]).reshape(gs[g], n) for g in range(3)]
    out["check"] = r.take(16 * n).reshape(4, 4 * n)
    out["seal"] = r.take(r.one())
    r.done()
    return out


def write_seal(path, v):
    gs = v["group_size"]
    _save(path, 4, [[v["po2"], len(v["globals"]), v["n_accum_mix"], gs[0], gs[1], gs[2], len(v["reg_group"]), len(v["combo_off"]) - 1,
                     len(v["combo_backs"])], v["reg_group"], v["reg_offset"], v["reg_combo"], v["combo_off"], v["combo_backs"],
                    np.frombuffer(v["proof_system_info"], dtype="<u4"), np.frombuffer(v["circuit_info"], dtype="<u4"), v["globals"],
                    v["groups"][0], v["groups"][1], v["groups"][2], v["check"], [len(v["seal"])], v["seal"]])

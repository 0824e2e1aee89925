"""Reader / writer of the vector files `provers/hip/vectors-p3` emits from the real Plonky3 crates (rev 88ea2b86).
Little-endian u32 words; word 0 = 0x31564B52 ("RKV1"), word 1 = kind; field elements as Montgomery words (R = 2^32).
  kind 5  p3_poseidon2.bin   width rp | rc_ext[8 width] rc_int[rp] diag[width] | 3 x {in[width] out[width]} |
                             row_len row[] digest[8] | left[8] right[8] out[8]
  kind 6  p3_pcs.bin         log_h w trace[h w] | lde[2 h w] | n beta[4] evals[4 n] folded[2 n]
  kind 7  p3_fib_proof.bin   log_n queries pow_bits log_blowup | public[3] | trace[2 n] | proof_words proof[]"""
import numpy as np

MAGIC = 0x31564B52


class Reader:
    def __init__(self, path, kind):
        self.w = np.fromfile(path, dtype="<u4")
        if self.w.size < 2 or int(self.w[0]) != MAGIC or int(self.w[1]) != kind:
            raise ValueError("%s: not a kind-%d vector file" % (path, kind))
        self.at = 2

    def word(self):
        self.at += 1
        return int(self.w[self.at - 1])

    def words(self, n):
        if self.at + n > self.w.size:
            raise ValueError("vector file too short")
        self.at += n
        return self.w[self.at - n: self.at].astype(np.uint32)

    def done(self):
        if self.at != self.w.size:
            raise ValueError("trailing words in vector file")


def read_poseidon2(path):
    r = Reader(path, 5)
    width, rp = r.word(), r.word()
    out = {"width": width, "rp": rp, "rc_ext": r.words(8 * width), "rc_int": r.words(rp), "diag": r.words(width),
           "kat": [(r.words(width), r.words(width)) for _ in range(3)]}
    n = r.word()
    out["row"], out["row_digest"] = r.words(n), r.words(8)
    out["left"], out["right"], out["compressed"] = r.words(8), r.words(8), r.words(8)
    r.done()
    return out


def read_pcs(path):
    r = Reader(path, 6)
    log_h, w = r.word(), r.word()
    h = 1 << log_h
    out = {"log_h": log_h, "w": w, "trace": r.words(h * w).reshape(h, w), "lde": r.words(2 * h * w).reshape(2 * h, w)}
    n = r.word()
    out["beta"], out["evals"], out["folded"] = r.words(4), r.words(4 * n).reshape(n, 4), r.words(2 * n).reshape(n // 2, 4)
    r.done()
    return out


def read_fib_proof(path):
    r = Reader(path, 7)
    out = {k: r.word() for k in ("log_n", "queries", "pow_bits", "log_blowup")}
    n = 1 << out["log_n"]
    out["public"], out["trace"] = r.words(3), r.words(2 * n).reshape(n, 2)
    out["proof"] = r.words(r.word())
    r.done()
    return out


def write(path, kind, parts):
    words = [np.array([MAGIC, kind], dtype="<u4")] + [np.ascontiguousarray(p, dtype="<u4").reshape(-1) for p in parts]
    np.concatenate(words).tofile(path)

"""The receipt cache file raiko's risc0 driver writes and reads:
`bincode::serialize(&(uuid: String, receipt: risc0_zkvm::Receipt))` at
`/tmp/risc0-cache/{label}.zkp` (reference provers/risc0/driver/src/bonsai.rs:274-310), label =
`hex(image_id) + "-" + hex(keccak(bytes(to_vec(expected_output))))` (bonsai.rs:100-108).

RECALLED layout (risc0-zkvm 1.0.1 receipt.rs / receipt/{composite,segment}.rs and
risc0-binfmt; not in the reference tree, no fixture pins it -- treat every field order below as
unverified until a real .zkp is available):
  Receipt          { inner: InnerReceipt, journal: Journal { bytes: Vec<u8> }, metadata: { verifier_parameters: Digest } }
  InnerReceipt     enum: 0 Composite(CompositeReceipt), 1 Succinct, 2 Groth16, 3 Fake
  CompositeReceipt { segments: Vec<SegmentReceipt>, assumption_receipts: Vec<..> (empty here), verifier_parameters: Digest }
  SegmentReceipt   { seal: Vec<u32>, index: u32, hashfn: String, verifier_parameters: Digest, claim: ReceiptClaim }
  ReceiptClaim     { pre: MaybePruned<SystemState>, post: MaybePruned<SystemState>, exit_code: ExitCode,
                     input: MaybePruned<Option<Input>>, output: MaybePruned<Option<Output>> }
  MaybePruned<T>   enum: 0 Value(T), 1 Pruned(Digest);  SystemState { pc: u32, merkle_root: Digest }
  ExitCode         enum: 0 Halted(u32), 1 Paused(u32), 2 SystemSplit, 3 SessionLimit
  Output           { journal: MaybePruned<Vec<u8>>, assumptions: MaybePruned<Assumptions(Vec<..>)> }
  Digest           [u32; 8]
bincode 1.x defaults: little-endian, fixed-width integers, u64 lengths, u32 variant indices.

What this backend cannot fill in: pre/post `SystemState` digests and the verifier-parameter
digests come from the executor / circuit crates; they are written as zero digests (pruned where
the type allows), so a file produced here has the risc0 shape but is not accepted by
`Receipt::verify` -- the seals inside are checked with rk_verify_segment instead."""
import struct
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

from .keccak import keccak256
from . import risc0_serde as rs

ZERO_DIGEST = (0,) * 8


class _W:
    def __init__(self):
        self.b = bytearray()

    def u32(self, x):
        self.b += struct.pack("<I", x)

    def u64(self, x):
        self.b += struct.pack("<Q", x)

    def bytes_(self, raw: bytes):
        self.u64(len(raw))
        self.b += raw

    def digest(self, d):
        self.b += struct.pack("<8I", *d)


class _R:
    def __init__(self, b: bytes):
        self.b, self.o = b, 0

    def take(self, n):
        if self.o + n > len(self.b):
            raise ValueError("truncated receipt file")
        self.o += n
        return self.b[self.o - n:self.o]

    def u32(self):
        return struct.unpack("<I", self.take(4))[0]

    def u64(self):
        return struct.unpack("<Q", self.take(8))[0]

    def bytes_(self):
        return self.take(self.u64())

    def digest(self):
        return struct.unpack("<8I", self.take(32))


@dataclass
class SegmentReceipt:
    seal: np.ndarray                      # uint32 transcript words
    index: int
    po2: int = 0                          # not a field of risc0's struct: recovered from the seal on load
    hashfn: str = "poseidon2"
    exit_code: Tuple[str, Optional[int]] = ("SystemSplit", None)


@dataclass
class Receipt:
    """`risc0_zkvm::Receipt` with `InnerReceipt::Composite`"""
    segments: List[SegmentReceipt]
    journal: bytes
    verifier_parameters: Tuple[int, ...] = ZERO_DIGEST

    @property
    def seals(self) -> List[np.ndarray]:
        return [s.seal for s in self.segments]


_EXIT = ["Halted", "Paused", "SystemSplit", "SessionLimit"]


def _write_claim(w: _W, seg: SegmentReceipt, journal: Optional[bytes]):
    for _ in range(2):          # pre, post: MaybePruned::Pruned(zero digest) -- unknown to this backend
        w.u32(1)
        w.digest(ZERO_DIGEST)
    name, code = seg.exit_code
    w.u32(_EXIT.index(name))
    if name in ("Halted", "Paused"):
        w.u32(int(code or 0))
    w.u32(1)                    # input: Pruned(zero digest)
    w.digest(ZERO_DIGEST)
    if journal is None:
        w.u32(0)                # output: Value(None)
        w.b += b"\0"
    else:
        w.u32(0)                # output: Value(Some(Output { journal: Value(bytes), assumptions: Pruned(zero) }))
        w.b += b"\1"
        w.u32(0)
        w.bytes_(journal)
        w.u32(1)
        w.digest(ZERO_DIGEST)


def _read_claim(r: _R):
    for _ in range(2):
        tag = r.u32()
        if tag == 1:
            r.digest()
        elif tag == 0:
            r.u32()
            r.digest()
        else:
            raise ValueError("bad MaybePruned tag")
    e = r.u32()
    if e >= len(_EXIT):
        raise ValueError("bad exit code")
    code = r.u32() if _EXIT[e] in ("Halted", "Paused") else None
    tag = r.u32()               # input
    if tag == 1:
        r.digest()
    else:
        if r.take(1) != b"\0":
            raise ValueError("inputs are not supported")
    tag = r.u32()               # output
    if tag == 1:
        r.digest()
    elif r.take(1) == b"\1":
        if r.u32() == 0:
            r.bytes_()
        else:
            r.digest()
        if r.u32() == 1:
            r.digest()
        else:
            if r.u64() != 0:
                raise ValueError("assumptions are not supported")
    return (_EXIT[e], code)


def serialize(uuid: str, receipt: Receipt) -> bytes:
    """bincode of `(String, Receipt)`: what `save_receipt` writes (bonsai.rs:294-302)"""
    w = _W()
    w.bytes_(uuid.encode("utf-8"))
    w.u32(0)                                        # InnerReceipt::Composite
    w.u64(len(receipt.segments))
    for i, seg in enumerate(receipt.segments):
        seal = np.ascontiguousarray(seg.seal, dtype="<u4")
        w.u64(seal.size)
        w.b += seal.tobytes()
        w.u32(seg.index)
        w.bytes_(seg.hashfn.encode("utf-8"))
        w.digest(receipt.verifier_parameters)
        last = i + 1 == len(receipt.segments)
        _write_claim(w, seg, receipt.journal if last else None)
    w.u64(0)                                        # assumption_receipts
    w.digest(receipt.verifier_parameters)
    w.bytes_(receipt.journal)                       # Journal { bytes }
    w.digest(receipt.verifier_parameters)           # ReceiptMetadata
    return bytes(w.b)


def deserialize(raw: bytes) -> Tuple[str, Receipt]:
    """`load_receipt` (bonsai.rs:274-292); raises ValueError on anything that is not such a file"""
    r = _R(raw)
    try:
        uuid = r.bytes_().decode("utf-8")
    except UnicodeDecodeError:
        raise ValueError("bad uuid string")
    if r.u32() != 0:
        raise ValueError("not a composite receipt")
    n = r.u64()
    if n > (1 << 20):
        raise ValueError("implausible segment count")
    segs = []
    for _ in range(n):
        words = r.u64()
        seal = np.frombuffer(r.take(4 * words), dtype="<u4").astype(np.uint32)
        index = r.u32()
        hashfn = r.bytes_().decode("utf-8")
        r.digest()
        exit_code = _read_claim(r)
        segs.append(SegmentReceipt(seal=seal, index=index, hashfn=hashfn, exit_code=exit_code))
    if r.u64() != 0:
        raise ValueError("assumption receipts are not supported")
    vp = r.digest()
    journal = r.bytes_()
    r.digest()
    if r.o != len(raw):
        raise ValueError("trailing bytes")
    return uuid, Receipt(segments=segs, journal=journal, verifier_parameters=vp)


def receipt_label(image_id: bytes, expected_output_b256: bytes) -> str:
    """bonsai.rs:100-108 for `O = B256`: hex(image id) - hex(keccak(bytes of the word-serialised output))"""
    words = rs.to_vec(rs.B256, expected_output_b256)
    return image_id.hex() + "-" + keccak256(rs.words_to_bytes(words)).hex()

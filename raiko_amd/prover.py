"""`HipProver`: the host-side mirror of raiko's `Prover` plugin interface for the HIP backend.

Reference interface (Champii/raiko @ 2024-08-07):
  * `trait Prover { async fn run(input, output, config, store) -> ProverResult<Proof>;
                    async fn cancel(proof_key, read) -> ProverResult<()> }`   lib/src/prover.rs:52-62
  * `Proof { proof, quote, kzg_proof }`                                       lib/src/prover.rs:29-38
  * `ProverError::{GuestError, FileIo, Param, StoreError}`                    lib/src/prover.rs:7-23
  * `Risc0Param { bonsai, snark, profile, execution_po2 }` read from
    `config["risc0"]`                                                         provers/risc0/driver/src/lib.rs:27-34,63
  * `Risc0Prover::run` / `maybe_prove` / `prove_locally`                      provers/risc0/driver/src/lib.rs:56-112,
                                                                              bonsai.rs:89-181, bonsai.rs:230-272

The reference toolchain (Rust) is absent from this image, so the host side above the C ABI is
Python here; INTEGRATION.md shows the Rust `provers/hip` crate a maintainer would add.  What is
NOT here because its source is not in the reference tree: the RV32IM executor, witness
generation and the rv32im constraint evaluator -- a `Session` therefore arrives already executed
(segments + journal), the point where `session.prove()` (bonsai.rs:271) starts.

Differences from the reference driver, on purpose:
  * errors are returned as `ProverError`, never panics (the reference `unwrap()`s at
    lib.rs:63,71,84 and bonsai.rs:266-271, leaving the task row at WorkInProgress);
  * the receipt cache lives in a per-process directory instead of the shared, wiped
    `/tmp/risc0-cache` (bonsai.rs:261-265 is a latent race under `concurrency_limit` = 16).
"""
import os
import struct
import tempfile
from dataclasses import dataclass
from typing import Any, List, Optional, Sequence, Tuple

import numpy as np

from . import receipt as receipt_mod
from . import risc0_serde as rs
from .receipt import Receipt, SegmentReceipt
from .segment import Segment

RISC0_PROVER_CODE = 3  # provers/risc0/driver/src/lib.rs:53; `hip` answers to proof_type "risc0"


class ProverError(Exception):
    """lib/src/prover.rs:7-23"""

    kind = "GuestError"

    def __str__(self):
        return "ProverError::%s `%s`" % (self.kind, self.args[0] if self.args else "")


class GuestError(ProverError):
    kind = "GuestError"


class FileIo(ProverError):
    kind = "FileIo"


class Param(ProverError):
    kind = "Param"


class StoreError(ProverError):
    kind = "StoreError"

    def __str__(self):
        return "Store error `%s`" % (self.args[0] if self.args else "")


@dataclass
class Proof:
    """The response body of a proof request (lib/src/prover.rs:29-38)."""

    proof: Optional[str] = None
    quote: Optional[str] = None
    kzg_proof: Optional[str] = None

    def to_json(self) -> dict:
        return {"proof": self.proof, "quote": self.quote, "kzg_proof": self.kzg_proof}


@dataclass
class Risc0Param:
    bonsai: bool
    snark: bool
    profile: bool
    execution_po2: int

    @staticmethod
    def deserialize(value: Any) -> "Risc0Param":
        if not isinstance(value, dict):
            raise Param("risc0 options must be an object")
        try:
            out = Risc0Param(bonsai=value["bonsai"], snark=value["snark"], profile=value["profile"],
                             execution_po2=value["execution_po2"])
        except KeyError as e:
            raise Param("missing field `%s`" % e.args[0])
        for name in ("bonsai", "snark", "profile"):
            if not isinstance(getattr(out, name), bool):
                raise Param("invalid type for `%s`: expected a boolean" % name)
        if isinstance(out.execution_po2, bool) or not isinstance(out.execution_po2, int) or out.execution_po2 < 0:
            raise Param("invalid type for `execution_po2`: expected u32")
        return out


def encode_journal_b256(h: bytes) -> bytes:
    """The journal the guest commits (`env::commit(&hash)`, provers/risc0/guest/src/main.rs:28) as the
    reference decodes it (`receipt.journal.decode::<B256>()`, bonsai.rs:157): risc0's word serde writes
    a `[u8; 32]` as 32 little-endian u32 words, one byte each (raiko_amd/risc0_serde.py) -- 128 bytes."""
    if len(h) != 32:
        raise ValueError("B256 is 32 bytes")
    return rs.words_to_bytes(rs.to_vec(rs.B256, h))


def decode_journal_b256(journal: bytes) -> Optional[bytes]:
    """inverse of encode_journal_b256; None if the bytes are not a word-serialised B256"""
    if len(journal) != 128:
        return None
    try:
        return bytes(rs.from_slice(rs.B256, struct.unpack("<32I", journal)))
    except ValueError:
        return None


@dataclass
class Session:
    """What `ExecutorImpl::run()` returns (bonsai.rs:267-269): the executed segments and the journal."""

    segments: List[Segment]
    journal: bytes
    image_id: bytes = b"\0" * 32
    # cycles per program counter from an executor run with profiling on (raiko_amd.executor.execute(profile=True))
    profile: Optional[list] = None

    @property
    def total_cycles(self) -> int:
        return sum(s.cycles for s in self.segments)


_CACHE_DIR = None
# `run` may be entered from up to concurrency_limit threads (host/src/lib.rs:38-41): rk_prove_session
# serialises sessions per device inside the library, where the contexts live


def _cache_dir() -> str:
    global _CACHE_DIR
    if _CACHE_DIR is None:
        _CACHE_DIR = tempfile.mkdtemp(prefix="raiko-hip-cache-")
    return _CACHE_DIR


def zkp_cache_path(label: str) -> str:
    """bonsai.rs:304-310, with a private directory"""
    return os.path.join(_cache_dir(), label + ".zkp")


def save_receipt(label: str, receipt_data: Tuple[str, Receipt]) -> None:
    """bonsai.rs:294-302: bincode of `(uuid, receipt)` (raiko_amd/receipt.py)"""
    try:
        with open(zkp_cache_path(label), "wb") as f:
            f.write(receipt_mod.serialize(receipt_data[0], receipt_data[1]))
    except OSError as e:
        raise FileIo(str(e))


def load_receipt(label: str) -> Optional[Tuple[str, Receipt]]:
    """bonsai.rs:274-292: None when there is no such file; a file that does not parse is a FileIo error"""
    try:
        with open(zkp_cache_path(label), "rb") as f:
            raw = f.read()
    except OSError:
        return None
    try:
        return receipt_mod.deserialize(raw)
    except ValueError as e:
        raise FileIo("cached receipt %s: %s" % (label, e))


def _rank_and_gpu(device: Optional[int]):
    """(rank, world, gpu, collective device): with torch.distributed initialised and no explicit
    `hip.device`, rank r proves on GPU LOCAL_RANK (or r) modulo the GPUs of the node, and the seal
    gather uses device tensors when the backend is RCCL ("nccl")."""
    rank, world, coll = 0, 1, None
    try:
        import torch
        import torch.distributed as tdist
        if tdist.is_available() and tdist.is_initialized():
            rank, world = tdist.get_rank(), tdist.get_world_size()
            if device is None:
                device = local_gpu_for_rank(int(os.environ.get("LOCAL_RANK", rank)))
            if tdist.get_backend() == "nccl":
                coll = torch.device("cuda", device)
    except ImportError:
        pass
    return rank, world, 0 if device is None else device, coll


def local_gpu_for_rank(local_rank: int) -> int:
    import ctypes as C
    from . import _lib
    n = C.c_int(0)
    _lib.load().rk_device_count(C.byref(n))
    return local_rank % n.value if n.value > 0 else 0


def prove_locally(segment_limit_po2: int, session: Session, device: Optional[int] = None, inflight: int = 3,
                  devices: Optional[Sequence[int]] = None) -> Receipt:
    """bonsai.rs:230-272 from the point the executor has produced the session: prove every
    segment (this rank's shard when torch.distributed is initialised), `inflight` at a time per
    GPU, and assemble the receipt.  `devices`: several GPUs driven from this one process
    (rk_prove_session's work queue) instead of one rank per GPU."""
    from . import dist as rdist
    for s in session.segments:
        if s.po2 > segment_limit_po2:
            raise GuestError("segment of 2^%d cycles exceeds segment_limit_po2 = %d" % (s.po2, segment_limit_po2))
    rank, world, gpu, coll_device = _rank_and_gpu(device)
    mine = rdist.shard_indices(len(session.segments), rank, world)
    # rk_prove_session (raiko_amd/csrc/session.hip): `inflight` proofs in flight, uploads staged ahead
    # on their own stream, and -- the `receipt.verify()` of the reference's tests (lib.rs:136) -- every
    # seal verified on a host thread while the GPU goes on.  The library serialises sessions per device and
    # keeps its contexts for the life of the process.
    from . import _lib
    from . import hal
    try:
        local = hal.prove_session([session.segments[i] for i in mine], device=gpu, inflight=inflight, devices=devices)
    except _lib.RkError as e:  # surface as GuestError like `From<String>` (prover.rs:19-23)
        if e.status == _lib.RK_ERR_VERIFY:
            raise GuestError("segment %d: seal failed verification" % mine[e.segment])
        raise GuestError(str(e))
    seals = rdist.gather_seals(local, len(session.segments), device=coll_device) if world > 1 else local
    if seals is None:  # non-root rank of a sharded proof
        seals = []
    n = len(seals)
    segs = [SegmentReceipt(seal=np.asarray(s, dtype=np.uint32), index=i, po2=session.segments[i].po2,
                           exit_code=("Halted", 0) if i + 1 == n else ("SystemSplit", None))
            for i, s in enumerate(seals)]
    return Receipt(segments=segs, journal=session.journal)


class HipProver:
    """Unit struct with associated functions, like `Risc0Prover` (lib.rs:51)."""

    @staticmethod
    def run(input: Any, output: Any, config: Any, store: Any = None) -> Proof:
        """`input` must expose `.session` (a `Session`) and `.chain_spec.chain_id`; `output.hash` is the
        32-byte expected journal; `config` is the whole request JSON (core/src/lib.rs:107)."""
        if not isinstance(config, dict) or "risc0" not in config:
            raise Param("missing `risc0` options in the proof request")
        param = Risc0Param.deserialize(config["risc0"])
        if param.bonsai:
            raise GuestError("the hip backend proves locally; bonsai = true is not available")
        if param.snark:
            raise GuestError("No STARK->SNARK (Groth16) stage in the hip backend")
        session = getattr(input, "session", None)
        if not isinstance(session, Session):
            raise GuestError("input carries no executed session (the RV32IM executor is outside this backend)")
        if param.profile:
            # bonsai.rs:252-255: `profile` switches the EXECUTOR's profiler on and names its output file.  The executor
            # runs before this call; its profile, when the session carries one, is written the way the reference
            # writes profile_r0_local.pb into the working directory.  A session executed without profiling proves
            # as usual (script/prove-block.sh:64-73 always sends profile = true).
            HipProver.last_profile_path = None
            if session.profile is not None:
                hip_cfg = config.get("hip", {}) if isinstance(config.get("hip", {}), dict) else {}
                path = str(hip_cfg.get("profile_path", "profile_rk_local.json"))
                try:
                    import json
                    with open(path, "w") as f:
                        json.dump({"unit": "cycles", "total_cycles": session.total_cycles,
                                   "by_pc": [{"pc": "0x%08x" % pc, "cycles": c} for pc, c in session.profile]}, f)
                except OSError as e:
                    raise FileIo(str(e))
                HipProver.last_profile_path = path
        expected = bytes(output.hash)
        if len(expected) != 32:
            raise Param("output.hash must be a B256")
        # bonsai.rs:100-108: hex(image id) - hex(keccak(bytes of to_vec(expected_output)))
        label = receipt_mod.receipt_label(session.image_id, expected)
        cached = load_receipt(label)
        if cached is not None:
            receipt = cached[1]
        else:
            hip = config.get("hip", {}) if isinstance(config.get("hip", {}), dict) else {}
            try:
                device = int(hip["device"]) if "device" in hip else None
                inflight = int(hip.get("inflight", 3))
                devices = [int(d) for d in hip["devices"]] if "devices" in hip else None
            except (TypeError, ValueError):
                raise Param("`hip.device` / `hip.inflight` / `hip.devices` must be integers")
            if inflight < 1 or inflight > 16:
                raise Param("`hip.inflight` must be in 1..16")
            receipt = prove_locally(param.execution_po2, session, device=device, inflight=inflight, devices=devices)
            if receipt.seals:
                save_receipt(label, ("", receipt))  # local proofs carry an empty uuid (bonsai.rs:139)
        # bonsai.rs:157-162: the journal is decoded and compared with the expected output; a mismatch
        # is logged, not fatal.  (A session whose journal is not word-serialised is compared raw.)
        decoded = decode_journal_b256(receipt.journal)
        HipProver.last_journal_matches = (decoded if decoded is not None else receipt.journal) == expected
        return Proof(proof=receipt.journal.hex(), quote=None, kzg_proof=None)

    last_journal_matches = None
    last_profile_path = None

    @staticmethod
    def cancel(proof_key: Tuple[int, bytes, int], read: Any) -> None:
        """A local proof cannot be interrupted (same as the SGX backend's no-op,
        provers/sgx/prover/src/lib.rs:151-153); nothing is stored under the key."""
        return None

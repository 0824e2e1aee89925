"""Host logic of the `Prover` mirror (raiko_amd/prover.py): parameter parsing and error behaviour
follow the reference driver (provers/risc0/driver/src/lib.rs:27-34,56-112), no GPU involved."""
import types

import numpy as np
import pytest

from raiko_amd import prover as pv
from raiko_amd.segment import synthetic_segment


def req(**kw):
    base = {"bonsai": False, "snark": False, "profile": True, "execution_po2": 18}
    base.update(kw)
    return {"proof_type": "risc0", "risc0": base}


def test_risc0_param_matches_prove_block_request():
    # the exact object script/prove-block.sh:64-73 sends
    p = pv.Risc0Param.deserialize({"bonsai": False, "snark": False, "profile": True, "execution_po2": 18})
    assert (p.bonsai, p.snark, p.profile, p.execution_po2) == (False, False, True, 18)
    with pytest.raises(pv.Param):
        pv.Risc0Param.deserialize({"bonsai": False, "snark": False, "profile": True})
    with pytest.raises(pv.Param):
        pv.Risc0Param.deserialize({"bonsai": "no", "snark": False, "profile": True, "execution_po2": 18})
    with pytest.raises(pv.Param):
        pv.Risc0Param.deserialize({"bonsai": False, "snark": False, "profile": True, "execution_po2": -1})


def test_errors_are_returned_not_panics():
    out = types.SimpleNamespace(hash=b"\x11" * 32)
    inp = types.SimpleNamespace(session=None, chain_spec=types.SimpleNamespace(chain_id=167009))
    with pytest.raises(pv.Param):
        pv.HipProver.run(inp, out, {"proof_type": "risc0"})
    with pytest.raises(pv.GuestError):
        pv.HipProver.run(inp, out, req(bonsai=True))
    with pytest.raises(pv.GuestError):
        pv.HipProver.run(inp, out, req(snark=True))
    with pytest.raises(pv.GuestError) as e:
        pv.HipProver.run(inp, out, req())
    assert str(e.value).startswith("ProverError::GuestError `")
    assert pv.HipProver.cancel((167009, b"\0" * 32, pv.RISC0_PROVER_CODE), None) is None


def test_segment_limit_is_enforced_before_touching_the_gpu():
    seg = synthetic_segment(6, (2, 2, 3))
    with pytest.raises(pv.GuestError):
        pv.prove_locally(5, pv.Session(segments=[seg], journal=b""))


def test_receipt_roundtrip_and_cache(tmp_path, monkeypatch):
    """the .zkp cache file is bincode of `(String, Receipt)` (bonsai.rs:294-302) under the label of
    bonsai.rs:100-108"""
    from raiko_amd import receipt as rc
    monkeypatch.setattr(pv, "_CACHE_DIR", str(tmp_path))
    segs = [rc.SegmentReceipt(seal=np.arange(7, dtype=np.uint32), index=0),
            rc.SegmentReceipt(seal=np.array([0xFFFFFFFF, 1], dtype=np.uint32), index=1, exit_code=("Halted", 0))]
    r = pv.Receipt(segments=segs, journal=pv.encode_journal_b256(b"\xab" * 32))
    pv.save_receipt("label", ("some-uuid", r))
    uuid, back = pv.load_receipt("label")
    assert uuid == "some-uuid" and back.journal == r.journal
    assert all(np.array_equal(a, b) for a, b in zip(back.seals, r.seals))
    assert [s.exit_code for s in back.segments] == [("SystemSplit", None), ("Halted", 0)]
    assert [s.hashfn for s in back.segments] == ["poseidon2", "poseidon2"]
    assert pv.load_receipt("missing") is None
    # bincode framing: u64 length + utf-8 of the uuid, then variant 0 (Composite), then a u64 segment count
    raw = open(pv.zkp_cache_path("label"), "rb").read()
    assert raw[:8] == (9).to_bytes(8, "little") and raw[8:17] == b"some-uuid"
    assert raw[17:21] == b"\0\0\0\0" and raw[21:29] == (2).to_bytes(8, "little")
    assert rc.serialize(uuid, back) == raw
    # a truncated or foreign file is an error, not a silent miss
    open(pv.zkp_cache_path("bad"), "wb").write(raw[:-3])
    with pytest.raises(pv.FileIo):
        pv.load_receipt("bad")
    # a cached receipt answers the request without proving (bonsai.rs:111-114)
    sess = pv.Session(segments=[], journal=b"", image_id=b"\x07" * 32)
    out = types.SimpleNamespace(hash=b"\xab" * 32)
    label = rc.receipt_label(sess.image_id, out.hash)
    assert label.startswith("07" * 32 + "-") and len(label) == 64 + 1 + 64
    pv.save_receipt(label, ("", r))
    proof = pv.HipProver.run(types.SimpleNamespace(session=sess), out, req())
    assert proof.to_json() == {"proof": r.journal.hex(), "quote": None, "kzg_proof": None}
    assert pv.HipProver.last_journal_matches is True


def test_profile_option_writes_the_executor_profile(tmp_path, monkeypatch):
    """`profile: true` (script/prove-block.sh:64-73 always sends it; bonsai.rs:252-255 hands it to the executor's
    profiler): a session executed with profiling leaves its cycle profile in the named file, one without proves as usual"""
    import json
    from raiko_amd import receipt as rc
    monkeypatch.setattr(pv, "_CACHE_DIR", str(tmp_path))
    r = pv.Receipt(segments=[rc.SegmentReceipt(seal=np.arange(3, dtype=np.uint32), index=0, exit_code=("Halted", 0))],
                   journal=pv.encode_journal_b256(b"\xcd" * 32))
    out = types.SimpleNamespace(hash=b"\xcd" * 32)
    sess = pv.Session(segments=[], journal=b"", image_id=b"\x09" * 32, profile=[(0x200800, 70), (0x200804, 30)])
    pv.save_receipt(rc.receipt_label(sess.image_id, out.hash), ("", r))       # answered from the cache: no GPU needed here
    cfg = req()
    cfg["hip"] = {"profile_path": str(tmp_path / "profile.json")}
    pv.HipProver.run(types.SimpleNamespace(session=sess), out, cfg)
    got = json.load(open(tmp_path / "profile.json"))
    assert got["by_pc"][0] == {"pc": "0x00200800", "cycles": 70} and pv.HipProver.last_profile_path == str(tmp_path / "profile.json")
    sess.profile = None
    pv.HipProver.run(types.SimpleNamespace(session=sess), out, cfg)            # nothing to write, not an error
    assert pv.HipProver.last_profile_path is None
    cfg["risc0"]["profile"] = False
    pv.HipProver.run(types.SimpleNamespace(session=sess), out, cfg)


def test_keccak256_and_label():
    import hashlib
    from raiko_amd import keccak as kk
    from raiko_amd import receipt as rc
    # published Keccak-256 answers (the empty-input one is the reference's KECCAK_EMPTY, lib/src/primitives/keccak.rs:22-23)
    assert kk.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert kk.keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    # the permutation and absorption against hashlib's SHA3-256 (same sponge, domain byte 0x06) on many lengths
    def sha3_with_our_permutation(data):
        rate = 136
        msg = bytearray(data) + b"\x06"
        while len(msg) % rate:
            msg.append(0)
        msg[-1] |= 0x80
        a = [[0] * 5 for _ in range(5)]
        for off in range(0, len(msg), rate):
            for i in range(rate // 8):
                a[i % 5][i // 5] ^= int.from_bytes(msg[off + 8 * i:off + 8 * i + 8], "little")
            a = kk._f1600(a)
        return b"".join(a[i % 5][i // 5].to_bytes(8, "little") for i in range(4))
    rng = np.random.default_rng(4)
    for n in (0, 1, 55, 135, 136, 137, 271, 272, 1000):
        data = bytes(rng.integers(0, 256, size=n, dtype=np.uint8))
        assert sha3_with_our_permutation(data) == hashlib.sha3_256(data).digest(), n
    # the label hashes the 128 bytes of the word-serialised B256, not the 32 raw bytes
    h = bytes(range(32))
    want = kk.keccak256(b"".join(bytes([b, 0, 0, 0]) for b in h)).hex()
    assert rc.receipt_label(b"\0" * 32, h).split("-")[1] == want


def test_risc0_word_serde_rules():
    from raiko_amd import risc0_serde as rs
    assert rs.to_vec(rs.U8, 7) == [7] and rs.to_vec(rs.U32, 0xDEADBEEF) == [0xDEADBEEF]
    assert rs.to_vec(rs.BOOL, True) == [1]
    assert rs.to_vec(rs.U64, 0x1122334455667788) == [0x55667788, 0x11223344]
    assert rs.to_vec(rs.B256, bytes(range(32))) == list(range(32))
    assert rs.to_vec(rs.BYTES, b"\x01\x02\x03\x04\x05") == [5, 0x04030201, 0x00000005]
    assert rs.to_vec(rs.STR, "ab") == [2, 0x6261]
    assert rs.to_vec(rs.Seq(rs.U16), [1, 2, 3]) == [3, 1, 2, 3]
    assert rs.to_vec(rs.Option(rs.U8), None) == [0] and rs.to_vec(rs.Option(rs.U8), 9) == [1, 9]
    risc0_param = rs.Struct(("bonsai", rs.BOOL), ("snark", rs.BOOL), ("profile", rs.BOOL), ("execution_po2", rs.U32))
    v = {"bonsai": False, "snark": False, "profile": True, "execution_po2": 18}
    assert rs.to_vec(risc0_param, v) == [0, 0, 1, 18]
    exit_code = rs.Enum(("Halted", rs.U32), ("Paused", rs.U32), ("SystemSplit", None))
    assert rs.to_vec(exit_code, ("Paused", 4)) == [1, 4] and rs.to_vec(exit_code, ("SystemSplit", None)) == [2]
    nested = rs.Struct(("id", rs.U64), ("blobs", rs.Seq(rs.BYTES)), ("hash", rs.Option(rs.B256)), ("t", rs.Tup(rs.U8, rs.STR)))
    val = {"id": 2**40 + 5, "blobs": [b"", b"xyz", bytes(8)], "hash": bytes(range(100, 132)), "t": (3, "hé")}
    words = rs.to_vec(nested, val)
    back = rs.from_slice(nested, words)
    assert back["id"] == val["id"] and back["blobs"] == val["blobs"] and bytes(back["hash"]) == val["hash"] and back["t"] == val["t"]
    for bad in ([], words[:-1], words + [0]):
        with pytest.raises(ValueError):
            rs.from_slice(nested, bad)
    with pytest.raises(ValueError):
        rs.to_vec(rs.U8, 256)
    with pytest.raises(ValueError):
        rs.from_slice(rs.BYTES, [3, 0xFF000000])       # padding must be zero


def test_permutation_count_of_s20():
    from raiko_amd.segment import poseidon2_permutations
    c = poseidon2_permutations(20, (16, 16, 224))
    d = 1 << 22
    fri_leaves = [(1 << 20) * 4 // 16, (1 << 16) * 4 // 16, (1 << 12) * 4 // 16]
    assert c["hash_rows"] == d * 17 + 4 * sum(fri_leaves)
    assert c["hash_fold"] == 4 * (d - 1) + sum(x - 1 for x in fri_leaves)
    # ragged widths round up to whole sponge blocks
    assert poseidon2_permutations(10, (3, 5, 33))["hash_rows"] == (1 << 12) * (1 + 1 + 3 + 1) + 4 * (1 << 12) // 16


def test_journal_word_serde_roundtrip():
    h = bytes(range(200, 232))
    j = pv.encode_journal_b256(h)
    assert len(j) == 128 and j[:8] == bytes([200, 0, 0, 0, 201, 0, 0, 0])
    assert pv.decode_journal_b256(j) == h
    assert pv.decode_journal_b256(j[:-1]) is None
    assert pv.decode_journal_b256(b"\x00\x01\x00\x00" * 32) is None  # a word above 0xff is not a byte
    with pytest.raises(ValueError):
        pv.encode_journal_b256(b"short")

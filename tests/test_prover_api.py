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
    monkeypatch.setattr(pv, "_CACHE_DIR", str(tmp_path))
    r = pv.Receipt(seals=[np.arange(7, dtype=np.uint32), np.array([0xFFFFFFFF, 1], dtype=np.uint32)],
                   journal=b"\xab" * 32, po2=[18, 17])
    pv.save_receipt("label", r)
    back = pv.load_receipt("label")
    assert back.journal == r.journal and back.po2 == r.po2
    assert all(np.array_equal(a, b) for a, b in zip(back.seals, r.seals))
    assert pv.load_receipt("missing") is None
    # a cached receipt answers the request without proving (bonsai.rs:111-114)
    sess = pv.Session(segments=[], journal=b"\xab" * 32)
    out = types.SimpleNamespace(hash=b"\xab" * 32)
    import hashlib
    pv.save_receipt(sess.image_id.hex() + "-" + hashlib.sha3_256(out.hash).hexdigest(), r)
    proof = pv.HipProver.run(types.SimpleNamespace(session=sess), out, req())
    assert proof.to_json() == {"proof": ("ab" * 32), "quote": None, "kzg_proof": None}
    assert pv.HipProver.last_journal_matches is True


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

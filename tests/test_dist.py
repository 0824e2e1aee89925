"""Segment sharding and the seal gather, world_size 2 over gloo on the CPU."""
import os
import socket

import numpy as np
import pytest

from raiko_amd.dist import gather_seals, shard_indices


def test_shard_indices_cover_everything_once():
    for n in (0, 1, 5, 8, 13):
        for world in (1, 2, 3, 8):
            got = sorted(i for r in range(world) for i in shard_indices(n, r, world))
            assert got == list(range(n))


def test_gather_single_process():
    seals = [np.arange(5, dtype=np.uint32), np.arange(3, dtype=np.uint32) + 0xFFFFFFF0]
    out = gather_seals(seals, 2)
    assert all(np.array_equal(a, b) for a, b in zip(out, seals))


def _worker(rank, world, port, n_segments, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        def seal_of(i):
            rng = np.random.default_rng(i)
            return rng.integers(0, 2**32, size=100 + 37 * i, dtype=np.uint32)  # ragged lengths, high bits set
        local = [seal_of(i) for i in shard_indices(n_segments, rank, world)]
        out = gather_seals(local, n_segments)
        if rank == 0:
            ok = len(out) == n_segments and all(np.array_equal(out[i], seal_of(i)) for i in range(n_segments))
            q.put(("ok" if ok else "mismatch"))
        else:
            q.put("ok" if out is None else "nonroot-got-data")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_segments", [2, 5])
def test_gather_gloo_world2(n_segments):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_segments, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert res == ["ok", "ok"]


def _prover_worker(rank, world, port, n_segments, q):
    """HipProver.run itself on two ranks: shard by rank, rank -> GPU mapping, seal gather, receipt on rank 0.
    No GPU here: hal.prove_session is replaced by a stand-in that records its arguments."""
    import types
    import torch.distributed as dist
    from raiko_amd import hal, prover as pv
    from raiko_amd.segment import synthetic_segment
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = str(rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        segs = [synthetic_segment(4, (2, 2, 3), seed=100 + i, n_globals=2) for i in range(n_segments)]
        seen = {}

        def fake_prove_session(segments, device=0, inflight=3, devices=None, **kw):
            seen["device"], seen["n"] = device, len(segments)
            return [np.full(10 + int(s.globals_[0] % 7), int(s.globals_[0]), dtype=np.uint32) for s in segments]

        hal.prove_session = fake_prove_session
        pv.local_gpu_for_rank = lambda local_rank: local_rank % 8       # an 8-GPU node
        journal = pv.encode_journal_b256(bytes([5]) * 32)
        sess = pv.Session(segments=segs, journal=journal, image_id=bytes([1]) * 32)
        inp = types.SimpleNamespace(session=sess, chain_spec=types.SimpleNamespace(chain_id=167009))
        out = types.SimpleNamespace(hash=bytes([5]) * 32)
        cfg = {"proof_type": "risc0", "risc0": {"bonsai": False, "snark": False, "profile": False, "execution_po2": 18}}
        proof = pv.HipProver.run(inp, out, cfg)
        ok = proof.proof == journal.hex() and seen["device"] == rank and seen["n"] == len(shard_indices(n_segments, rank, world))
        if rank == 0:
            from raiko_amd import receipt as rc
            _, cached = pv.load_receipt(rc.receipt_label(sess.image_id, out.hash))
            ok = ok and len(cached.seals) == n_segments and all(
                int(cached.seals[i][0]) == int(segs[i].globals_[0]) for i in range(n_segments))
            ok = ok and cached.segments[-1].exit_code == ("Halted", 0)
        q.put("ok" if ok else "mismatch")
    finally:
        dist.destroy_process_group()


def test_prover_run_sharded_over_gloo_world2():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_prover_worker, args=(r, 2, port, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert res == ["ok", "ok"]


def test_gather_unpack_restores_segment_order():
    """rk_gather_unpack: the host half of the C-level seal gather (rk_gather_seals = two ncclAllGather calls + this):
    simulated ranks' padded payloads come back in segment order, for totals that do and do not divide by the world size"""
    import ctypes as C
    from raiko_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(12)
    for world, n_total in ((1, 3), (2, 5), (4, 4), (8, 13), (3, 1)):
        seals = [rng.integers(0, 2**32, size=int(rng.integers(1, 50)), dtype=np.uint32) for _ in range(n_total)]
        per_rank = (n_total + world - 1) // world
        max_len = max(s.size for s in seals)
        lens = np.zeros((world, per_rank), dtype=np.uint32)
        pay = np.zeros((world, per_rank, max_len), dtype=np.uint32)
        for i, s in enumerate(seals):
            lens[i % world, i // world] = s.size
            pay[i % world, i // world, : s.size] = s
        outs = [np.zeros(max_len, dtype=np.uint32) for _ in range(n_total)]
        ptrs = (_lib.u32p * n_total)(*[o.ctypes.data_as(_lib.u32p) for o in outs])
        caps = (C.c_size_t * n_total)(*[max_len] * n_total)
        words = (C.c_size_t * n_total)()
        assert lib.rk_gather_unpack(lens.ctypes.data_as(_lib.u32p), pay.ctypes.data_as(_lib.u32p), world, per_rank, max_len, n_total,
                                    ptrs, caps, words) == 0
        for i, s in enumerate(seals):
            assert words[i] == s.size and np.array_equal(outs[i][: s.size], s)
        caps[0] = 0                                                    # one buffer too small: reported, the others delivered
        assert lib.rk_gather_unpack(lens.ctypes.data_as(_lib.u32p), pay.ctypes.data_as(_lib.u32p), world, per_rank, max_len, n_total,
                                    ptrs, caps, words) == -5
    assert lib.rk_gather_unpack(None, None, 1, 1, 0, 1, None, None, None) == -1

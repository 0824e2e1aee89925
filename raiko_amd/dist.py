"""Segment-parallel proving across the GPUs of one node.

The reference proves a session's segments serially (`session.prove()`,
provers/risc0/driver/src/bonsai.rs:271); segments are independent STARKs (the executor
already writes them as separate files, bonsai.rs:261-266), so they shard with no
data-path collective: segment i goes to rank i mod world.  The only exchange is the
final gather of the (variable-length, ~0.3 MB) seals to rank 0, which assembles the
composite receipt -- one padded all_gather over RCCL (backend "nccl" on ROCm) or gloo.
"""
from typing import List, Optional, Sequence

import numpy as np


def shard_indices(n_segments: int, rank: int, world: int) -> List[int]:
    """Static round-robin: the segments rank `rank` proves, in order."""
    return list(range(rank, n_segments, world))


def gather_seals(local: Sequence[np.ndarray], n_segments: int, group=None, device=None) -> Optional[List[np.ndarray]]:
    """Gather the seals of all ranks to rank 0 in segment order.

    `local` holds this rank's seals for shard_indices(n_segments, rank, world).  Returns the full
    ordered list on rank 0 and None elsewhere.  Works without torch.distributed initialised
    (single process)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        assert len(local) == n_segments
        return [np.asarray(s, dtype=np.uint32) for s in local]
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    mine = shard_indices(n_segments, rank, world)
    assert len(local) == len(mine), (len(local), len(mine))
    per_rank = (n_segments + world - 1) // world
    dev = device if device is not None else torch.device("cpu")
    # fixed-size length table, then one padded payload all_gather (latency-bound: < 3 MB total)
    lens = torch.zeros(per_rank, dtype=torch.int64, device=dev)
    for j, s in enumerate(local):
        lens[j] = int(s.size)
    all_lens = [torch.zeros_like(lens) for _ in range(world)]
    dist.all_gather(all_lens, lens, group=group)
    max_len = max(int(t.max().item()) for t in all_lens)
    payload = torch.zeros((per_rank, max(max_len, 1)), dtype=torch.int32, device=dev)
    for j, s in enumerate(local):
        payload[j, : s.size] = torch.from_numpy(np.asarray(s, dtype=np.uint32).view(np.int32)).to(dev)
    all_payload = [torch.zeros_like(payload) for _ in range(world)]
    dist.all_gather(all_payload, payload, group=group)
    if rank != 0:
        return None
    out: List[Optional[np.ndarray]] = [None] * n_segments
    for r in range(world):
        pl = all_payload[r].cpu().numpy().view(np.uint32)
        ln = all_lens[r].cpu().numpy()
        for j, seg_idx in enumerate(shard_indices(n_segments, r, world)):
            out[seg_idx] = pl[j, : int(ln[j])].copy()
    return out  # type: ignore[return-value]

"""Several segment proofs in flight on one GPU.

The reference proves the segments of a session one after the other (`session.prove()`,
provers/risc0/driver/src/bonsai.rs:271).  One proof is a chain of ~120 dependent kernel launches
with a host round trip at every Merkle root (the Fiat-Shamir transcript lives on the host), and
the last levels of every Merkle tree are single-workgroup kernels, so a lone proof leaves an
MI355X partly idle.  Segments are independent, so this module keeps `inflight` prover contexts on
the GPU -- each with its own HIP stream, scratch pool and host thread -- and hands them segments
from a shared queue: the latency-bound parts of one proof overlap the throughput-bound parts of
the others (35 ms -> 27 ms per 2^20-cycle segment at three in flight).
A context holds ~7 GiB of HBM at that size; 288 GB leaves room for far more than pays off.
"""
import threading
from typing import Callable, List, Optional, Sequence

import numpy as np

from .hal import HipHal
from .segment import Segment

DEFAULT_INFLIGHT = 3


class SegmentPipeline:
    """`inflight` HipHal contexts on one device, one worker thread per context while proving."""

    def __init__(self, device: int = 0, inflight: int = DEFAULT_INFLIGHT, streams: Optional[Sequence[int]] = None):
        """`streams`: optional hipStream_t handles owned by the caller (one per context), e.g. torch
        streams when the inputs are torch tensors produced on them; otherwise each context creates
        its own stream."""
        if inflight < 1:
            raise ValueError("inflight must be >= 1")
        if streams is not None and len(streams) != inflight:
            raise ValueError("need one stream per context")
        self.device = device
        self.hals: List[HipHal] = [HipHal(device, stream=streams[i] if streams is not None else None)
                                   for i in range(inflight)]

    @property
    def inflight(self) -> int:
        return len(self.hals)

    def close(self):
        for h in self.hals:
            h.close()
        self.hals = []

    def prove(self, segments: Sequence[Segment], device_inputs: Optional[Sequence] = None,
              on_done: Optional[Callable[[int, HipHal], None]] = None) -> List[np.ndarray]:
        """Seals of `segments`, in order.  `device_inputs[i]` (optional) = (groups[3], check) of
        HBM-resident inputs for segment i.  Segments are taken from a shared counter, so a short
        last segment does not leave a context idle behind a static assignment.  `on_done(i, hal)`
        runs on the worker right after segment i (e.g. to read `hal.last_timing()`)."""
        n = len(segments)
        out: List[Optional[np.ndarray]] = [None] * n
        nxt = [0]
        lock = threading.Lock()
        errors: List[BaseException] = []

        def worker(h: HipHal):
            while True:
                with lock:
                    i = nxt[0]
                    nxt[0] += 1
                if i >= n or errors:
                    return
                try:
                    out[i] = h.prove_segment(segments[i], device_inputs=device_inputs[i] if device_inputs else None)
                    if on_done is not None:
                        on_done(i, h)
                except BaseException as e:  # surfaced on the calling thread
                    errors.append(e)
                    return

        workers = self.hals[: max(1, min(len(self.hals), n))]
        if len(workers) == 1:
            worker(workers[0])
        else:
            ts = [threading.Thread(target=worker, args=(h,), name="raiko-hip-prove-%d" % k)
                  for k, h in enumerate(workers)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
        if errors:
            raise errors[0]
        return out  # type: ignore[return-value]

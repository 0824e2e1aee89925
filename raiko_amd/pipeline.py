"""Several segment proofs in flight on one GPU.

The reference proves the segments of a session one after the other (`session.prove()`,
provers/risc0/driver/src/bonsai.rs:271).  One proof is a chain of ~120 dependent kernel launches
with a host round trip at every Merkle root (the Fiat-Shamir transcript lives on the host), and
the last levels of every Merkle tree are single-workgroup kernels, so a lone proof leaves an
MI355X partly idle.  Segments are independent, so this module keeps `inflight` prover contexts on
the GPU -- each with its own HIP stream, scratch pool and host thread -- and hands them segments
from a shared queue: the latency-bound parts of one proof overlap the throughput-bound parts of
the others (32 ms -> 25 ms per 2^20-cycle segment at three in flight).
A context holds ~7 GiB of HBM at that size; 288 GB leaves room for far more than pays off.

Host-resident traces (what an executor hands over: `Segment.groups` as numpy arrays) are uploaded
by one more context on its own stream, `upload_ahead` segments ahead of the provers and into a ring
of reusable device buffers, so the PCIe transfer of segment i+1 (1.13 GB at S20, ~20 ms) runs
under the proof of segment i instead of in front of it in the same stream.
"""
import os
import queue
import threading
import time
from typing import Callable, List, Optional, Sequence

import numpy as np

from .hal import HipHal
from .segment import Segment

DEFAULT_INFLIGHT = 3


class SegmentPipeline:
    """`inflight` HipHal contexts on one device, one worker thread per context while proving."""

    def __init__(self, device: int = 0, inflight: int = DEFAULT_INFLIGHT, streams: Optional[Sequence[int]] = None,
                 upload_ahead: int = 2):
        """`streams`: optional hipStream_t handles owned by the caller (one per context), e.g. torch
        streams when the inputs are torch tensors produced on them; otherwise each context creates
        its own stream.  `upload_ahead`: host-resident segments staged in HBM ahead of the provers
        (0 = every prover uploads its own inputs in its own stream)."""
        if inflight < 1:
            raise ValueError("inflight must be >= 1")
        if streams is not None and len(streams) != inflight:
            raise ValueError("need one stream per context")
        self.device = device
        self.hals: List[HipHal] = [HipHal(device, stream=streams[i] if streams is not None else None)
                                   for i in range(inflight)]
        self.upload_ahead = max(0, int(upload_ahead))
        self._uploader: Optional[HipHal] = None  # created on first use: a context of its own = a stream of its own
        self._ring: List[dict] = []               # reusable staging slots {"key": shape key, "groups": [...], "check": buf}

    @property
    def inflight(self) -> int:
        return len(self.hals)

    def close(self):
        for slot in self._ring:
            _free_slot(slot)
        self._ring = []
        if self._uploader is not None:
            self._uploader.close()
            self._uploader = None
        for h in self.hals:
            h.close()
        self.hals = []

    def prove(self, segments: Sequence[Segment], device_inputs: Optional[Sequence] = None,
              on_done: Optional[Callable[[int, HipHal, np.ndarray], None]] = None) -> List[np.ndarray]:
        """Seals of `segments`, in order.  `device_inputs[i]` (optional) = (groups[3], check) of
        HBM-resident inputs for segment i.  Segments are taken from a shared counter, so a short
        last segment does not leave a context idle behind a static assignment.  `on_done(i, hal, seal)`
        runs on the worker right after segment i (e.g. to verify the seal or read `hal.last_timing()`)."""
        n = len(segments)
        out: List[Optional[np.ndarray]] = [None] * n
        nxt = [0]
        lock = threading.Lock()
        errors: List[BaseException] = []
        stage = None
        if device_inputs is None and self.upload_ahead > 0 and n > 1:
            stage = _Stager(self, segments, errors)
            stage.start()

        def worker(h: HipHal):
            while True:
                with lock:
                    i = nxt[0]
                    nxt[0] += 1
                if i >= n or errors:
                    return
                try:
                    if stage is not None:
                        slot = stage.take(i)
                        if slot is None:  # the stager failed; its error is in `errors`
                            return
                        try:
                            out[i] = h.prove_segment(segments[i], device_inputs=(slot["groups"], slot["check"]))
                        finally:
                            stage.give_back(slot)
                    else:
                        out[i] = h.prove_segment(segments[i], device_inputs=device_inputs[i] if device_inputs else None)
                    if on_done is not None:
                        on_done(i, h, out[i])
                except BaseException as e:  # surfaced on the calling thread
                    errors.append(e)
                    if stage is not None:
                        stage.abort()
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
        if stage is not None:
            stage.abort()
            stage.join()
        if errors:
            raise errors[0]
        return out  # type: ignore[return-value]


def _free_slot(slot: dict):
    for b in slot.get("groups", []):
        b.free()
    if slot.get("check") is not None:
        slot["check"].free()


class _Stager(threading.Thread):
    """Uploads host-resident segments, in order, into the pipeline's ring of device buffers on the
    uploader context's stream.  A slot goes back to the ring when its proof is done; the ring has
    `upload_ahead + inflight` slots, so at most `upload_ahead` finished uploads wait for a prover.
    All allocation and freeing of staging buffers happens on this thread (a context's allocator
    is not shared between threads)."""

    def __init__(self, pipe: "SegmentPipeline", segments: Sequence[Segment], errors: list):
        super().__init__(name="raiko-hip-upload")
        self.pipe, self.segments, self.errors = pipe, segments, errors
        if pipe._uploader is None:
            pipe._uploader = HipHal(pipe.device)
        self.hal = pipe._uploader
        self.n_slots = pipe.upload_ahead + len(pipe.hals)
        self.free: "queue.Queue" = queue.Queue()
        for slot in pipe._ring[: self.n_slots]:
            self.free.put(slot)
        self.cond = threading.Condition()
        self.ready = {}
        self.stop = False

    @staticmethod
    def _key(seg: Segment):
        return (seg.po2,) + tuple(int(g.shape[0]) for g in seg.groups)

    def _slot_for(self, seg: Segment) -> Optional[dict]:
        slot = None
        while slot is None:
            if self.stop:
                return None
            try:
                slot = self.free.get_nowait()
            except queue.Empty:
                if len(self.pipe._ring) < self.n_slots:  # grow the ring before waiting for a proof to end
                    slot = {"key": None, "groups": [], "check": None}
                    self.pipe._ring.append(slot)
                else:
                    try:
                        slot = self.free.get(timeout=0.05)
                    except queue.Empty:
                        pass
        if slot["key"] != self._key(seg):
            _free_slot(slot)
            slot["groups"] = [self.hal.alloc_elem(int(g.shape[0]) * seg.rows) for g in seg.groups]
            slot["check"] = self.hal.alloc_elem(4 * 4 * seg.rows)
            slot["key"] = self._key(seg)
        return slot

    def run(self):
        try:
            dbg = os.environ.get("RAIKO_PIPE_DEBUG")  # per-segment staging times on stdout
            for i, seg in enumerate(self.segments):
                t_a = time.perf_counter()
                slot = self._slot_for(seg)
                if slot is None:
                    return
                t_b = time.perf_counter()
                for g in range(3):
                    a = np.ascontiguousarray(seg.groups[g], dtype=np.uint32)
                    if a.shape != (seg.taps.group_size[g], seg.rows):
                        raise ValueError("segment %d: group %d has shape %s" % (i, g, a.shape))
                    slot["groups"][g].copy_from(a)
                chk = np.ascontiguousarray(seg.check, dtype=np.uint32)
                if chk.shape != (4, 4 * seg.rows):
                    raise ValueError("segment %d: check has shape %s" % (i, chk.shape))
                slot["check"].copy_from(chk)
                self.hal.sync()  # the host arrays may go away and the buffers change hands after this
                if dbg:
                    print("stage %d: slot wait %.1f ms, upload %.1f ms" % (i, 1e3 * (t_b - t_a), 1e3 * (time.perf_counter() - t_b)), flush=True)
                with self.cond:
                    self.ready[i] = slot
                    self.cond.notify_all()
        except BaseException as e:
            self.errors.append(e)
        finally:
            with self.cond:
                self.stop = True
                self.cond.notify_all()

    def take(self, i: int) -> Optional[dict]:
        with self.cond:
            while i not in self.ready and not self.stop:
                self.cond.wait(0.05)
            return self.ready.pop(i, None)

    def give_back(self, slot: dict):
        self.free.put(slot)

    def abort(self):
        with self.cond:
            self.stop = True
            self.cond.notify_all()

#!/usr/bin/env python3
"""H2D bandwidth of the box: pageable vs page-locked host memory, one and three streams.
Context for the PCIe-inclusive numbers in DESIGN.md section 5."""
import time
import torch

def bw(pinned, streams, mb=1024, reps=3):
    n = mb * (1 << 20) // 4
    hs = [torch.empty(n, dtype=torch.int32).pin_memory() if pinned else torch.empty(n, dtype=torch.int32) for _ in range(streams)]
    for h in hs:
        h.random_(0, 1000)
    ds = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(streams)]
    ss = [torch.cuda.Stream() for _ in range(streams)]
    best = 0.0
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for h, d, s in zip(hs, ds, ss):
            with torch.cuda.stream(s):
                d.copy_(h, non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = max(best, streams * mb / 1024 / dt)
    return best

for pinned in (False, True):
    for streams in (1, 3):
        print("pinned=%s streams=%d: %.1f GiB/s" % (pinned, streams, bw(pinned, streams)))


def under_load():
    """the same copies while three streams keep the CUs busy with integer kernels"""
    import threading
    stop = [False]
    x = [torch.randint(0, 1 << 30, (1 << 26,), dtype=torch.int32, device="cuda") for _ in range(3)]

    def burn(k):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            while not stop[0]:
                y = x[k]
                for _ in range(20):
                    y = (y * 1664525 + 1013904223) % 2013265921
                s.synchronize()

    ts = [threading.Thread(target=burn, args=(k,)) for k in range(3)]
    for t in ts:
        t.start()
    time.sleep(0.5)
    n = 1024 * (1 << 20) // 4
    for pinned in (False, True):
        h = torch.empty(n, dtype=torch.int32)
        h.random_(0, 1000)
        if pinned:
            h = h.pin_memory()
        d = torch.empty(n, dtype=torch.int32, device="cuda")
        for prio, name in ((0, "normal"), (-1, "high")):
            s = torch.cuda.Stream(priority=prio)
            best = 0.0
            for _ in range(3):
                s.synchronize()
                t0 = time.perf_counter()
                with torch.cuda.stream(s):
                    d.copy_(h, non_blocking=True)
                s.synchronize()
                best = max(best, 1.0 / (time.perf_counter() - t0))
            print("under load: pinned=%s priority=%s: %.1f GiB/s" % (pinned, name, best))
    stop[0] = True
    for t in ts:
        t.join()


under_load()

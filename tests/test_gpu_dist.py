"""Two ranks of bench.py on ONE GPU (rehearsal mode: gloo collectives, ranks share the card): the
segment-sharded path -- per-rank pipelines, the seal gather to rank 0, max-over-ranks timing --
end to end through torchrun, as the driver launches it for N > 1 (there over RCCL, one rank per GPU)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_ranks_share_one_gpu():
    env = dict(os.environ, RAIKO_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--po2", "12",
           "--inflight", "2", "--no-cpu"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    assert d["value"] > 0 and d["seal_verified"] is True
    assert d["config"]["parallelism"] == "segment-parallel x2"


_RCCL_ONE_RANK = r"""
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
from raiko_amd.dist import gather_seals
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", %(port)r)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
rng = np.random.default_rng(5)
local = [rng.integers(0, 2**32, size=n, dtype=np.uint32) for n in (7, 260000, 1, 33)]
out = gather_seals(local, len(local), device=dev)
assert out is not None and len(out) == len(local)
for a, b in zip(out, local):
    assert a.dtype == np.uint32 and np.array_equal(a, b)
dist.barrier()
dist.destroy_process_group()
print("rccl-gather-ok")
"""


def test_gather_seals_over_rccl_single_rank():
    """the device-tensor form of the seal gather (backend "nccl" = RCCL) with a one-rank communicator:
    the only multi-GPU collective of the path, exercised as far as a one-GPU box allows"""
    code = _RCCL_ONE_RANK % {"root": ROOT, "port": str(_free_port())}
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl-gather-ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


_RK_COMM_ONE_RANK = r"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, %(root)r)
from raiko_amd import _lib
lib = _lib.load()
uid = C.create_string_buffer(128)
assert lib.rk_comm_unique_id(uid) == 0
comm = C.c_void_p()
assert lib.rk_comm_create(uid, 0, 1, 0, C.byref(comm)) == 0
rng = np.random.default_rng(5)
local = [rng.integers(0, 2**32, size=n, dtype=np.uint32) for n in (7, 260000, 1, 33)]
n = len(local)
lp = (_lib.u32p * n)(*[a.ctypes.data_as(_lib.u32p) for a in local])
lw = (C.c_size_t * n)(*[a.size for a in local])
outs = [np.zeros(a.size, dtype=np.uint32) for a in local]
op = (_lib.u32p * n)(*[a.ctypes.data_as(_lib.u32p) for a in outs])
oc = (C.c_size_t * n)(*[a.size for a in outs])
ow = (C.c_size_t * n)()
st = lib.rk_gather_seals(comm, lp, lw, n, n, op, oc, ow)
assert st == 0, (st, lib.rk_comm_last_error(comm))
for a, b, w in zip(outs, local, ow):
    assert w == b.size and np.array_equal(a, b)
assert lib.rk_gather_seals(comm, lp, lw, n - 1, n, op, oc, ow) == -1      # not this rank's share of n segments
assert lib.rk_comm_destroy(comm) == 0
print("rk-comm-gather-ok")
"""


def test_rk_gather_seals_single_rank():
    """the C-level seal gather (rk_comm_* / rk_gather_seals: RCCL looked up at run time, two ncclAllGather calls) with a
    one-rank communicator -- what a one-GPU box can run; the unpacking for N ranks is tests/test_dist.py"""
    code = _RK_COMM_ONE_RANK % {"root": ROOT}
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rk-comm-gather-ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])

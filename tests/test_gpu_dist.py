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

"""The N > 1 bench path (sharding by chunk blocks, 584-byte exchange, merge + final exponentiation on rank 0)
end to end on ONE GPU: two ranks share device 0 and exchange over gloo (bench.py's documented test hook);
real multi-GPU runs use RCCL with one GPU per rank."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu():
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo", BENCH_ALL_ON_DEVICE0="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4096",
           "--no-cpu", "--no-aux"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8192 and d["value"] > 0 and d["scaling"] == "weak"

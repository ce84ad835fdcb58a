"""bench.py's N > 1 control flow (interleaved rows per rank, double-buffered asynchronous gather of the wave-1 hit
records, max-over-ranks timing, one JSON line from rank 0) run as two ranks on ONE GPU through the test hook
NRT_BENCH_TEST_SHARED_GPU=1 (gloo with CPU staging instead of RCCL, which cannot place two ranks on one device).
The driver's multi-GPU numbers never use the hook."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_share_one_gpu():
    env = dict(os.environ, NRT_BENCH_TEST_SHARED_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29613", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--builds", "1", "--no-cpu-baseline"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["scaling"] == "weak"
    assert out["config"]["rays_per_step"] > 2 * 4_000_000 and out["value"] > 100.0
    assert "RCCL gather" in out["config"]["parallelism"]
    assert out["roofline"]["frac"] > 0 and "cpu_baseline" not in out

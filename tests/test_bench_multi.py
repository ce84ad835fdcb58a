"""bench.py's N > 1 control flow (self-spawn of the ranks, interleaved rows per rank, double-buffered asynchronous gather
of BOTH waves' hit records, max-over-ranks timing, one JSON line from rank 0) run as two ranks on ONE GPU through the
test hook NRT_BENCH_TEST_SHARED_GPU=1 (gloo with CPU staging instead of RCCL, which cannot place two ranks on one device).
The driver's multi-GPU numbers never use the hook."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                          text=True, env=env, timeout=timeout, cwd=ROOT)


@pytest.mark.gpu
def test_two_ranks_share_one_gpu_through_the_self_spawn_path():
    # `python bench.py --gpus 2` with no launcher around it: bench.py starts the two ranks itself
    r = _run(["--gpus", "2", "--steps", "4", "--warmup", "1", "--builds", "1", "--no-cpu-baseline"], {"NRT_BENCH_TEST_SHARED_GPU": "1"})
    assert r.returncode == 0, r.stdout[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["scaling"] == "weak"
    assert out["config"]["rays_per_step"] > 2 * 4_000_000 and out["value"] > 100.0
    assert "RCCL gather of both waves" in out["config"]["parallelism"]
    mg = out["multi_gpu"]
    assert mg["rccl_ranks"] == 2 and len(mg["primary_kernel_ms"]) == 2 and len(mg["bounce_kernel_ms"]) == 2
    assert mg["gathered_bytes_per_step"] == 2 * 2 * 1920 * 1080 * 16  # both waves, both ranks
    assert mg["kernel_ms_max"] >= mg["kernel_ms_min"] > 0
    assert out["roofline"]["algorithmic"]["bytes_per_launch"] > 0 and out["roofline"]["build"]["frac"] <= 1.0
    assert "cpu_baseline" not in out
    # the default config with N > 1 also carries BASELINE.json's strong-scaling case: the fixed 4096x4096 frame in N tiles
    sc = out["strong_c4"]
    assert sc["scaling"] == "strong" and sc["rays_per_step"] > 2 * 4096 * 2048 * 0.9 and sc["value"] > 100.0
    assert "4096x2048" in sc["workload"] and sc["bvh"]["nodes"] > 4_000_000 and len(sc["multi_gpu"]["wall_ms_per_step"]) == 2
    # counters of rank 0's share, fractions against rank 0's own launch times
    rf = out["roofline"]
    assert rf["traffic_source"].startswith("in-run"), rf.get("pmc_error")
    assert 0.0 < rf["frac"] <= 1.0 and 0.0 < rf["per_rank_hbm_frac"]["min"] <= rf["per_rank_hbm_frac"]["max"] <= 1.0


@pytest.mark.gpu
def test_strong_scaling_config_splits_a_fixed_frame():
    r = _run(["--gpus", "2", "--config", "C4", "--steps", "2", "--warmup", "1", "--builds", "1", "--no-cpu-baseline"],
             {"NRT_BENCH_TEST_SHARED_GPU": "1"}, timeout=1500)
    assert r.returncode == 0, r.stdout[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["scaling"] == "strong" and out["n_gpus"] == 2 and out["config"]["name"] == "C4"
    assert "4096x2048" in out["config"]["workload"]  # each of the two ranks traces half of the 4096 rows
    assert out["bvh"]["nodes"] > 4_000_000
    # the counter sub-run traces THIS rank's share (4096x2048 rays), so the fractions stay fractions
    rf = out["roofline"]
    assert rf["traffic_source"].startswith("in-run"), rf.get("pmc_error")
    assert 0.0 < rf["frac"] <= 1.0 and 0.0 < rf["valu"]["frac"] <= 1.0 and 0.0 < rf["l1"]["frac"] <= 1.0
    assert abs(rf["per_wave"]["primary"]["rays"] - 4096 * 2048) == 0


@pytest.mark.gpu
def test_more_gpus_than_the_box_has_is_refused():
    import torch

    if torch.cuda.device_count() != 1:
        pytest.skip("needs a one-GPU box")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {}, timeout=300)
    assert r.returncode != 0
    assert "--gpus 2 requested but this box exposes 1 GPU" in r.stdout
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]

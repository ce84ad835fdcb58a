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


def _line_and_extras(r, extras_path):
    """the ONE small JSON line of rank 0 (benchlib/line.py) and the full result object it names"""
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    assert len(lines[0]) < 8192, len(lines[0])
    out = json.loads(lines[0])
    assert out["extras_file"] == extras_path
    return out, json.load(open(extras_path))


@pytest.mark.gpu
def test_two_ranks_share_one_gpu_through_the_self_spawn_path(tmp_path):
    # `python bench.py --gpus 2` with no launcher around it: bench.py starts the two ranks itself
    xp = str(tmp_path / "extras.json")
    r = _run(["--gpus", "2", "--steps", "4", "--warmup", "1", "--builds", "1", "--no-cpu-baseline", "--extras-file", xp], {"NRT_BENCH_TEST_SHARED_GPU": "1"})
    assert r.returncode == 0, r.stdout[-4000:]
    out, full = _line_and_extras(r, xp)
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["scaling"] == "weak"
    assert out["config"]["rays_per_step"] > 2 * 4_000_000 and out["value"] > 100.0
    assert "RCCL gather of both waves" in out["config"]["parallelism"]
    mg = out["multi_gpu"]
    assert mg["rccl_ranks"] == 2 and len(mg["primary_kernel_ms"]) == 2 and len(mg["bounce_kernel_ms"]) == 2
    assert mg["gathered_bytes_per_step"] == 2 * 2 * 1920 * 1080 * 16  # both waves, both ranks
    assert mg["kernel_ms_max"] >= mg["kernel_ms_min"] > 0
    assert out["roofline"]["bytes_per_launch"]["algorithmic"] > 0 and out["roofline"]["build"]["frac"] <= 1.0
    assert out["cpu_baseline"] is None
    # the default config with N > 1 also carries BASELINE.json's strong-scaling case: the fixed 4096x4096 frame in N tiles
    sc = out["strong_c4"]
    assert sc["scaling"] == "strong" and sc["rays_per_step"] > 2 * 4096 * 2048 * 0.9 and sc["value"] > 100.0
    fsc = full["strong_c4"]
    assert "4096x2048" in fsc["workload"] and fsc["bvh"]["nodes"] > 4_000_000 and len(fsc["multi_gpu"]["wall_ms_per_step"]) == 2
    # counters of rank 0's share, fractions against rank 0's own launch times
    rf = out["roofline"]
    assert rf["source"].startswith("in-run"), full["roofline"]["detail"].get("errors")
    assert rf["bound"] in ("hbm", "l1", "valu") and rf["frac"] == rf["fracs"][rf["bound"]] == max(rf["fracs"].values())
    assert 0.0 < rf["frac"] <= 1.0 and 0.0 < rf["per_rank_hbm_frac"]["min"] <= rf["per_rank_hbm_frac"]["max"] <= 1.0
    c = rf["bytes_per_launch"]
    assert c["algorithmic"] > c["requested"] > c["fetched"] > 0 and c["compulsory"] > 0


@pytest.mark.gpu
def test_strong_scaling_config_splits_a_fixed_frame(tmp_path):
    xp = str(tmp_path / "extras.json")
    r = _run(["--gpus", "2", "--config", "C4", "--steps", "2", "--warmup", "1", "--builds", "1", "--no-cpu-baseline", "--extras-file", xp],
             {"NRT_BENCH_TEST_SHARED_GPU": "1"}, timeout=1500)
    assert r.returncode == 0, r.stdout[-4000:]
    out, full = _line_and_extras(r, xp)
    assert out["scaling"] == "strong" and out["n_gpus"] == 2 and out["config"]["name"] == "C4"
    assert "4096x2048" in out["config"]["workload"]  # each of the two ranks traces half of the 4096 rows
    assert out["bvh"]["nodes"] > 4_000_000
    # the counter sub-run traces THIS rank's share (4096x2048 rays), so the fractions stay fractions
    rf = out["roofline"]
    assert rf["source"].startswith("in-run"), full["roofline"]["detail"].get("errors")
    assert 0.0 < rf["frac"] <= 1.0 and 0.0 < rf["fracs"]["valu"] <= 1.0 and 0.0 < rf["fracs"]["l1"] <= 1.0 and 0.0 < rf["hbm"]["frac"] <= 1.0
    assert full["roofline"]["detail"]["per_wave"]["primary"]["rays"] == 4096 * 2048


@pytest.mark.gpu
def test_more_gpus_than_the_box_has_is_refused():
    import torch

    if torch.cuda.device_count() != 1:
        pytest.skip("needs a one-GPU box")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {}, timeout=300)
    assert r.returncode != 0
    assert "--gpus 2 requested but this box exposes 1 GPU" in r.stdout
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]

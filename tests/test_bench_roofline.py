"""bench.py's roofline arithmetic on the CPU: counter means in, fractions of the stated peaks out (no GPU, no profiler)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def load_bench():
    """the counter arithmetic lives in benchlib/ (bench.py re-exports what tools/ use)"""
    import types

    import benchlib
    from benchlib import counters

    m = types.SimpleNamespace()
    for mod in (benchlib, counters):
        for k, v in vars(mod).items():
            if not k.startswith("_"):
                setattr(m, k, v)
    return m


def test_roofline_from_counters_uses_the_documented_formulas():
    b = load_bench()
    # the counter means of profiles/r02m (primary / bounce launches of C3)
    pmc = {
        "primary": {"FETCH_SIZE": 89752.8, "WRITE_SIZE": 44100.0, "TCC_HIT_sum": 2.0e6, "TCC_MISS_sum": 1.0e6,
                    "TCP_TOTAL_CACHE_ACCESSES_sum": 86277220.0, "TCP_TCC_READ_REQ_sum": 4158560.0,
                    "SQ_THREAD_CYCLES_VALU": 4.88e9, "SQ_INSTS_VALU": 1.1655e8, "SQ_WAIT_ANY": 4.0e8, "SQ_WAVE_CYCLES": 1.0e9,
                    "SQ_LDS_BANK_CONFLICT": 0.0, "GRBM_GUI_ACTIVE": 8 * 2.25e3 * 250.0},
        "bounce": {"FETCH_SIZE": 170000.0, "WRITE_SIZE": 43000.0, "TCC_HIT_sum": 4.0e6, "TCC_MISS_sum": 2.0e6,
                   "TCP_TOTAL_CACHE_ACCESSES_sum": 150973895.0, "TCP_TCC_READ_REQ_sum": 9737816.0,
                   "SQ_THREAD_CYCLES_VALU": 7.0e9, "SQ_INSTS_VALU": 2.137e8, "SQ_WAIT_ANY": 7.0e8, "SQ_WAVE_CYCLES": 1.8e9,
                   "SQ_LDS_BANK_CONFLICT": 0.0, "GRBM_GUI_ACTIVE": 8 * 2.25e3 * 465.0},
        "profiled_us": {"primary": 250.0, "bounce": 465.0},
    }
    k_ms = {"primary": 0.2506, "bounce": 0.4653}
    r = b.roofline_from_counters(pmc, k_ms, 256)
    hbm, valu, l1 = r["hbm"], r["valu"], r["l1"]
    # HBM: FETCH_SIZE (KiB) x 1024 x 2 + WRITE_SIZE (KiB) x 1024, both launches over both launch times
    want = (89752.8 * 2048 + 44100.0 * 1024 + 170000.0 * 2048 + 43000.0 * 1024) / ((0.2506 + 0.4653) * 1e-3) / 1e9
    assert abs(hbm["achieved_GBs"] - want) < 0.1 and abs(hbm["frac"] - want / b.HBM_PEAK_GBS) < 1e-3
    assert abs(hbm["l2_hit_rate"] - 2.0 / 3.0) < 1e-3
    # vector lanes: SQ_THREAD_CYCLES_VALU over 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz
    peak = 256 * 4 * b.VALU_LANES_PER_SIMD * b.CLOCK_GHZ * 1e9
    assert abs(valu["frac"] - (4.88e9 + 7.0e9) / ((0.2506 + 0.4653) * 1e-3) / peak) < 1e-3
    assert abs(valu["lane_util"] - (4.88e9 + 7.0e9) / (64 * (1.1655e8 + 2.137e8))) < 1e-3
    # vector L1: tag look-ups over the launch times against the micro-benchmark's measured rate
    assert abs(l1["frac"] - (86277220.0 + 150973895.0) / ((0.2506 + 0.4653) * 1e-3) / 1e9 / b.L1_PEAK_GACC_S) < 1e-3
    assert abs(l1["per_wave"]["bounce"]["requests_to_l2_per_lookup"] - 9737816.0 / 150973895.0) < 1e-3
    for part in (hbm, valu, l1):
        assert 0.0 < part["frac"] < 1.0


def test_missing_counters_leave_their_part_out():
    b = load_bench()
    pmc = {"primary": {"FETCH_SIZE": 1.0, "WRITE_SIZE": 1.0}, "bounce": {"FETCH_SIZE": 1.0, "WRITE_SIZE": 1.0}}
    r = b.roofline_from_counters(pmc, {"primary": 0.1, "bounce": 0.1}, 256)
    assert r["hbm"] is not None and r["valu"] is None and r["l1"] is None


def test_dry_run_preflight_reports_what_a_multi_gpu_run_would_trip_over():
    """`bench.py --gpus 8 --dry-run` launches nothing: on a box without 8 GPUs it names the device count as the failing check
    and still validates the library, the rendezvous, the tile split and the buffer sizes."""
    import json
    import subprocess
    import sys

    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--config", "C4", "--dry-run"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, cwd=ROOT)
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["dry_run"] is True and out["n_gpus"] == 8 and out["rays_per_rank_per_wave"] == 4096 * 512
    checks = {c["check"]: c for c in out["checks"]}
    for name in ("devices", "rccl_backend", "hsa_ipc_mode", "master_addr", "master_port", "library", "tiles", "memory", "gather"):
        assert name in checks, name
    assert checks["library"]["ok"] and checks["tiles"]["ok"] and checks["memory"]["ok"]
    import torch

    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    assert checks["devices"]["ok"] == (have >= 8)
    assert (r.returncode == 0) == out["ok"]

"""The driver's JSON line (benchlib/line.py): canned results in, a line of a few KB out that a JSON reader takes back — round 5's
20 KB line was not parsed by the driver (BENCH_r05.json: parsed null).  Also the `roofline` object: `bound` is computed from the
fractions, and the byte chain algorithmic -> requested -> l1 -> fetched -> compulsory is ordered and reproducible."""
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from benchlib import HBM_PEAK_GBS, L1_PEAK_GACC_S  # noqa: E402
from benchlib import line as bl  # noqa: E402
from benchlib import roofline as br  # noqa: E402
from benchlib.counters import roofline_from_counters  # noqa: E402

# the counter means of profiles/r05final_pmc (C3, primary / bounce launches)
PMC = {
    "primary": {"FETCH_SIZE": 70.0e3, "WRITE_SIZE": 34.4e3, "TCC_HIT_sum": 7.2e6, "TCC_MISS_sum": 2.8e6,
                "TCP_TOTAL_CACHE_ACCESSES_sum": 98.0e6, "TCP_TCC_READ_REQ_sum": 5.0e6,
                "SQ_THREAD_CYCLES_VALU": 4.6e9, "SQ_INSTS_VALU": 1.15e8, "SQ_WAIT_ANY": 4.0e8, "SQ_WAVE_CYCLES": 0.9e9,
                "SQ_LDS_BANK_CONFLICT": 0.0},
    "bounce": {"FETCH_SIZE": 140.0e3, "WRITE_SIZE": 34.1e3, "TCC_HIT_sum": 1.1e7, "TCC_MISS_sum": 5.0e6,
               "TCP_TOTAL_CACHE_ACCESSES_sum": 158.0e6, "TCP_TCC_READ_REQ_sum": 1.0e7,
               "SQ_THREAD_CYCLES_VALU": 6.9e9, "SQ_INSTS_VALU": 1.85e8, "SQ_WAIT_ANY": 8.0e8, "SQ_WAVE_CYCLES": 1.6e9,
               "SQ_LDS_BANK_CONFLICT": 0.0},
}
K_MS = {"primary": 0.2506, "bounce": 0.4263}


def fake_workload():
    import numpy as np

    wl = types.SimpleNamespace()
    wl.rb, wl.n1, wl.n2 = 4, 2073600, 2055142
    wl.RAY, wl.HIT = np.dtype([("x", "u1", 36)]), np.dtype([("x", "u1", 16)])
    wl.stats = {"num_branch_nodes": 284374}
    wl.faces = np.zeros((1000000, 3), dtype=np.uint32)
    return wl


def canned_full():
    wl = fake_workload()
    counters = roofline_from_counters(PMC, K_MS, 256, launch_ms=0.3246)
    bytes1, bytes2 = 3974134444, 4819029096
    comp, _ = br.compulsory_bytes(wl, bytes1, bytes2)
    counts = {"record_bytes": 128, "primary": {"steps": 14_000_000, "prims": 9_640_000}, "bounce": {"steps": 24_000_000, "prims": 14_600_000}}
    roof = br.headline("nrt::k_traverse_wide<float, 12, false, 0, true, false, 4, 0>", 0.3246, K_MS, (bytes1 + bytes2) // 2,
                       12991.6, comp, counters, br.requested_bytes(wl, counts), {"bytes": 74749960, "ms": 1.2736, "GBs": 58.7, "frac": 0.00734})
    roof["detail"] = {"counters": counters, "walk_counts": counts, "prose": "x" * 3000}
    # the other sections: what round 5's run produced (the 20 KB line kept under profiles/)
    old = None
    for ln in open(os.path.join(ROOT, "profiles", "r05final_bench.log")):
        if ln.startswith('{"metric'):
            old = json.loads(ln)
    assert old is not None
    full = {k: v for k, v in old.items() if k != "roofline"}
    full["roofline"] = roof
    full["build_host_ms"] = {"first": 31.2, "steady": 2.41, "upload": 0.9, "device": 1.29, "readback_lazy": 3.8, "unit": "ms"}
    return full


def test_the_line_is_small_and_round_trips():
    full = canned_full()
    assert len(json.dumps(full)) > 15000  # the payload that broke the reader is all in there
    line = bl.compact_line(full, "gpurun_out/bench_extras.json")
    text = json.dumps(line)
    assert len(text) < 8192 and len(text) <= bl.LINE_BUDGET, len(text)
    assert "\n" not in text
    back = json.loads(text)
    assert back == line
    for k in bl.CONTRACT_KEYS + ("config", "roofline", "cpu_baseline", "build_ms", "extras_file"):
        assert k in back, k
    assert back["config"]["name"] == "C3" and "workload" in back["config"] and "model" not in back["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "hbm", "bytes_per_launch", "ratios", "kernel", "launch_ms"):
        assert k in back["roofline"], k
    assert "detail" not in back["roofline"]
    for k in ("value", "unit", "cores", "kind", "sample", "parity_same_tree_bit_identical"):
        assert k in back["cpu_baseline"], k
    assert len(back["cpu_baseline"]["sample"]) <= 160
    assert back["cpu_baseline"]["parity_own_trees_within_1e-5"] is True
    assert set(back["configs"]) == {"C2", "C4tile", "C5"} and all("value" in v for v in back["configs"].values())
    assert back["next_rows"]["scene_10k"]["parity"] is True


def test_an_oversized_payload_sheds_optional_groups_not_required_ones():
    full = canned_full()
    full["next_rows"] = {"row%d" % i: {"value": float(i), "parity": {"bit_identical": True}} for i in range(400)}
    line = bl.compact_line(full, "x.json")
    assert len(json.dumps(line)) <= bl.LINE_BUDGET
    assert "next_rows" not in line and "roofline" in line and "cpu_baseline" in line and "value" in line


def test_bound_is_computed_and_the_byte_chain_is_ordered():
    full = canned_full()
    r = full["roofline"]
    fr = r["fracs"]
    assert r["bound"] == max(fr, key=fr.get) == "l1"
    assert r["frac"] == fr["l1"] and r["unit"] == "GB/s" and abs(r["peak"] - L1_PEAK_GACC_S * 16) < 1e-6
    # the contract's HBM figure is there whichever unit binds
    assert r["hbm"]["peak"] == HBM_PEAK_GBS and r["hbm"]["frac"] == fr["hbm"] and r["traffic"] > 0
    want_traffic = ((70.0e3 + 140.0e3) * 2048 + (34.4e3 + 34.1e3) * 1024) / 2
    assert abs(r["traffic"] - want_traffic) < 2
    c = r["bytes_per_launch"]
    assert c["algorithmic"] > c["requested"] > c["fetched"] > c["compulsory"] > 0
    assert c["l1"] == (98.0e6 + 158.0e6) / 2 * 16
    want_req = (128 * (14_000_000 + 24_000_000) + 40 * (9_640_000 + 14_600_000) + 53 * (2073600 + 2055142)) / 2
    assert c["requested"] == int(want_req)
    assert abs(r["ratios"]["requested_over_fetched"] - want_req / want_traffic) < 1e-3
    assert abs(r["ratios"]["fetched_over_compulsory"] - want_traffic / c["compulsory"]) < 1e-3
    # a binding HBM: only the hbm part measured
    only_hbm = roofline_from_counters({w: {"FETCH_SIZE": 1.0e6, "WRITE_SIZE": 1.0e5} for w in ("primary", "bounce")}, K_MS, 256, launch_ms=0.3)
    r2 = br.headline("k", 0.3, K_MS, 10**9, 1.0, 1.0e8, only_hbm, None, {})
    assert r2["bound"] == "hbm" and r2["unit"] == "GB/s" and r2["peak"] == HBM_PEAK_GBS and r2["frac"] == r2["hbm"]["frac"]
    # no counters: nothing is claimed
    r3 = br.headline("k", 0.3, K_MS, 10**9, 1.0, 1.0e8, None, None, {})
    assert r3["bound"] is None and r3["frac"] is None and r3["traffic"] is None and "UNMEASURED" in r3["source"]


def test_extras_file_is_written_and_named_relative_to_the_repo(tmp_path):
    p = bl.write_extras({"a": 1}, str(tmp_path / "x" / "extras.json"))
    assert p and json.load(open(tmp_path / "x" / "extras.json")) == {"a": 1}

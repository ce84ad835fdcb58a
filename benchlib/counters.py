"""Hardware counters collected in the same invocation (outside the timed region): rocprofv3 --pmc passes over a sub-run of
bench.py (--pmc-child), and the roofline fractions computed from them."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

from . import CLOCK_GHZ, HBM_PEAK_GBS, L1_PEAK_GACC_S, N_XCD, ROOT, VALU_CYCLES_PER_WAVE_INST, VALU_LANES_PER_SIMD
from .workload import Workload

# ---------------------------------------------------------------------------
# hardware counters, collected in the same invocation (outside the timed region)
# ---------------------------------------------------------------------------
# Counter passes: one rocprofv3 --pmc invocation each (kernel trace only, as MI355X_MICROARCH.md prescribes).  The TCC block has
# four counter slots (FETCH_SIZE takes 3, WRITE_SIZE 2), the TCP and SQ blocks have their own: three passes carry everything.
# Each entry: (tag, counters, fallback passes tried when the combined pass fails or returns no rows).
PMC_PASSES = [
    ("fetch_tcp", "FETCH_SIZE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum",
     [("fetch", "FETCH_SIZE"), ("tcp", "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum")]),
    ("write_tcc", "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum", [("write", "WRITE_SIZE"), ("tcc", "TCC_HIT_sum TCC_MISS_sum")]),
    ("sq", "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE", []),
]


def pmc_child(args):
    """The sub-run the counter passes profile: for every config named, the set-up (one primary launch) and then
    (warmup + steps) x (primary, bounce) launches of THIS rank's share of the workload — nothing else."""
    import torch

    done = []
    for name in args.pmc_configs.split(","):
        wl = Workload(name, rank=args.pmc_rank, world=args.pmc_world, builds=1, mesh_path=args.mesh if name == "C2" else None)
        for _ in range(args.warmup + args.steps):
            wl.accel.TraverseBatchDevice(wl.d_rays1, wl.d_hits1, wl.d_mask1)
            wl.accel.TraverseBatchDevice(wl.d_rays2, wl.d_hits2, wl.d_mask2)
        torch.cuda.synchronize()
        done.append({"name": name, "kernel": wl.accel.LastKernelName(), "n1": wl.n1, "n2": wl.n2})
        del wl
        torch.cuda.empty_cache()
    print(json.dumps({"pmc_child": True, "configs": done}), flush=True)


def walk_counts_child(args):
    """Sub-run under NRT_USE_PROF_LIB=1 (the profiling build of the same sources): the two waves once through the COUNTING
    instantiation of the timed kernel (tunable debug = 32 | 64: a ray's record holds the records it stepped through and the leaf
    primitives it tested in place of u, v).  Prints {"walk_counts": {...}} — benchlib/roofline.py turns it into `requested` bytes."""
    import torch

    wl = Workload(args.config, rank=args.pmc_rank, world=args.pmc_world, builds=1, mesh_path=args.mesh if args.config == "C2" else None)
    a = wl.accel
    a.TraverseBatchDevice(wl.d_rays1, wl.d_hits1, wl.d_mask1)
    timed_kernel = a.LastKernelName()
    a.SetTunable("debug", 32 | 64)
    targs = [x.strip() for x in timed_kernel.split("<", 1)[-1].rstrip(">").split(",")]  # <T, STACK, STATS, KIND, PLAIN, CLOCK, WIDTH, ORDER>
    width = int(targs[6]) if len(targs) >= 7 and targs[6].isdigit() else 2
    out = {"timed_kernel": timed_kernel, "record_bytes": (128 if width == 4 else 64) if wl.rb == 4 else 112}
    ft = torch.float32 if wl.rb == 4 else torch.float64
    for wave, d_r, d_h, n in (("primary", wl.d_rays1, wl.d_hits1, wl.n1), ("bounce", wl.d_rays2, wl.d_hits2, wl.n2)):
        if not n:
            out[wave] = {"steps": 0, "prims": 0}
            continue
        a.TraverseBatchDevice(d_r, d_h)
        torch.cuda.synchronize()
        rec = d_h[: n * wl.HIT.itemsize].view(ft).reshape(n, 4).to(torch.float64)
        out[wave] = {"steps": int(rec[:, 0].sum().item()), "prims": int(rec[:, 1].sum().item())}
        out["counting_kernel"] = a.LastKernelName()
    print(json.dumps({"walk_counts": out}), flush=True)


def walk_counts(config, mesh_path=None, rank=0, world=1):
    """Run walk_counts_child in a process of its own (the profiling library must not be in the bench process).  -> (dict or None, error)"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--walk-counts-child", "--config", config, "--pmc-rank", str(rank), "--pmc-world", str(world)]
    if mesh_path:
        cmd += ["--mesh", mesh_path]
    env = dict(os.environ, NRT_USE_PROF_LIB="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=300, cwd=ROOT)
    except Exception as e:  # pragma: no cover
        return None, repr(e)
    for line in r.stdout.splitlines():
        if line.startswith("{") and "walk_counts" in line:
            return json.loads(line)["walk_counts"], None
    return None, "rc %d: %s" % (r.returncode, r.stdout[-300:])


def _kernel_key(name):
    return name.replace("void ", "").split("(")[0].replace(" ", "")


def _pmc_pass(exe, tag, counters, child_args, out_root, env):
    """One rocprofv3 invocation.  Returns (counter rows, kernel-trace rows, the child's config list, error or None)."""
    out_dir = os.path.join(out_root, tag)
    cmd = [exe, "--kernel-trace", "--pmc"] + counters.split() + ["--output-format", "csv", "-d", out_dir, "-o", "p", "--",
                                                                   sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child"] + child_args
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=600, cwd="/tmp")
    except Exception as e:  # pragma: no cover
        return [], [], None, "%s: %r" % (tag, e)
    if r.returncode != 0:
        return [], [], None, "%s: rc %d: %s" % (tag, r.returncode, r.stdout[-300:])
    child = None
    for line in r.stdout.splitlines():
        if line.startswith("{") and "pmc_child" in line:
            child = json.loads(line)["configs"]
    rows, trace = [], []
    for path in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(path)))
    for path in glob.glob(os.path.join(out_dir, "**", "*kernel_trace.csv"), recursive=True):
        trace += list(csv.DictReader(open(path)))
    if not rows or child is None:
        return [], [], child, "%s: no counter rows" % tag
    return rows, trace, child, None


def pmc_collect(configs, mesh_path=None, rank=0, world=1, keep_dir=None, warmup=1, steps=3):
    """Run the counter passes over `bench.py --pmc-child` (ONE sub-run per pass traces every config named, this rank's
    share of it) and return {config: {"primary": {counter: per-launch mean}, "bounce": {...}, "profiled_us": {...}}}
    plus an error string (or None).  Launches are attributed by kernel name and dispatch order: per config one set-up
    launch (primary), then (primary, bounce) pairs."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    tmp = keep_dir or tempfile.mkdtemp(prefix="nrt_pmc_", dir="/tmp")
    os.makedirs(tmp, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    child_args = ["--pmc-configs", ",".join(configs), "--pmc-rank", str(rank), "--pmc-world", str(world), "--steps", str(steps), "--warmup", str(warmup)]
    if mesh_path:
        child_args += ["--mesh", mesh_path]
    per_launch = 1 + 2 * (warmup + steps)
    out = {c: {"primary": {}, "bounce": {}, "profiled_us": {"primary": None, "bounce": None}} for c in configs}
    errors = []

    def absorb(rows, trace, child, with_durations):
        by_kernel = {}
        for c in child:  # configs in launch order, grouped by the kernel variant they ran
            by_kernel.setdefault(_kernel_key(c["kernel"]), []).append(c["name"])
        for key, names in by_kernel.items():
            mine = [x for x in rows if _kernel_key(x.get("Kernel_Name", "")) == key]
            ids = sorted({int(x["Dispatch_Id"]) for x in mine})
            if len(ids) != per_launch * len(names):
                errors.append("%s: %d dispatches of %s, expected %d" % (",".join(names), len(ids), key, per_launch * len(names)))
                continue
            where = {d: (names[k // per_launch], k % per_launch) for k, d in enumerate(ids)}
            acc = {}
            for x in mine:
                name, k = where[int(x["Dispatch_Id"])]
                if k == 0:
                    continue  # the set-up launch
                wave = "primary" if k % 2 == 1 else "bounce"
                acc[(name, wave, x["Counter_Name"], k)] = acc.get((name, wave, x["Counter_Name"], k), 0.0) + float(x["Counter_Value"])
            lists = {}
            for (name, wave, cname, _k), v in acc.items():
                lists.setdefault((name, wave, cname), []).append(v)
            for (name, wave, cname), v in lists.items():
                out[name][wave][cname] = float(np.mean(v))
            if with_durations:
                tr = [x for x in trace if _kernel_key(x.get("Kernel_Name", "")) == key]
                tr.sort(key=lambda x: int(x["Start_Timestamp"]))
                if len(tr) == per_launch * len(names):
                    for k, x in enumerate(tr):
                        name, kk = names[k // per_launch], k % per_launch
                        if kk:
                            out[name].setdefault("_durs", {}).setdefault("primary" if kk % 2 == 1 else "bounce", []).append(
                                (int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) * 1e-3)

    for tag, counters, fallback in PMC_PASSES:
        rows, trace, child, err = _pmc_pass(exe, tag, counters, child_args, tmp, env)
        if err and fallback:  # the combined pass was refused: the blocks one by one
            errors.append(err + " (retried as %s)" % "+".join(t for t, _ in fallback))
            for ftag, fcounters in fallback:
                rows, trace, child, ferr = _pmc_pass(exe, ftag, fcounters, child_args, tmp, env)
                if ferr:
                    errors.append(ferr)
                else:
                    absorb(rows, trace, child, False)
            continue
        if err:
            errors.append(err)
            continue
        absorb(rows, trace, child, tag == "sq")
    for c in configs:
        d = out[c].pop("_durs", {})
        out[c]["profiled_us"] = {w: (float(np.mean(d[w])) if d.get(w) else None) for w in ("primary", "bounce")}
    if not keep_dir:
        shutil.rmtree(tmp, ignore_errors=True)
    return out, ("; ".join(errors) if errors else None)


def roofline_from_counters(pmc, k_ms, n_cus, launch_ms=None):
    """HBM, VALU and L1 rooflines of the primary / bounce launches from the in-run counter means (per launch).  The
    fractions divide by the launch times `k_ms` (per wave) — or, for the top-level HBM figure of the headline, by
    `launch_ms`, the average launch of the timed region itself."""
    simds = n_cus * 4
    lane_peak = simds * VALU_LANES_PER_SIMD * CLOCK_GHZ * 1e9  # lane-operations per second
    res = {"hbm": None, "valu": None, "l1": None}
    waves = ("primary", "bounce")
    if all("FETCH_SIZE" in pmc[w] and "WRITE_SIZE" in pmc[w] for w in waves):
        # rocprofv3 reports both in KiB; gfx950: FETCH_SIZE counts 128-B read requests as 64 B -> x2 (MI355X_MICROARCH.md §HBM)
        b = {w: pmc[w]["FETCH_SIZE"] * 1024.0 * 2.0 + pmc[w]["WRITE_SIZE"] * 1024.0 for w in waves}
        tot_ms = sum(k_ms[w] for w in waves) if launch_ms is None else 2.0 * launch_ms
        gbs = sum(b.values()) / (tot_ms * 1e-3) / 1e9
        res["hbm"] = {"bytes_per_launch": int(sum(b.values()) / 2), "achieved_GBs": round(gbs, 1), "peak_GBs": HBM_PEAK_GBS,
                      "frac": round(gbs / HBM_PEAK_GBS, 4),
                      "time_base": "per-wave kernel times" if launch_ms is None else "average launch of the timed region",
                      "per_wave_bytes": {w: int(b[w]) for w in waves},
                      "per_wave_frac": {w: round(b[w] / (k_ms[w] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) for w in waves},
                      "formula": "FETCH_SIZE KiB x 1024 x 2 (gfx950 wide-read correction) + WRITE_SIZE KiB x 1024"}
        if all("TCC_HIT_sum" in pmc[w] for w in waves):
            h = sum(pmc[w]["TCC_HIT_sum"] for w in waves)
            m = sum(pmc[w]["TCC_MISS_sum"] for w in waves)
            res["hbm"]["l2_hit_rate"] = round(h / max(1.0, h + m), 4)
    if all("TCP_TOTAL_CACHE_ACCESSES_sum" in pmc[w] for w in waves):
        # one look-up per active lane for scattered accesses, one per quad of lanes reading one 64-byte line (calibrated on the
        # micro-benchmark): the address / tag path of the vector L1, which the node and triangle fetches of this kernel load
        per = {}
        for w in waves:
            acc = pmc[w]["TCP_TOTAL_CACHE_ACCESSES_sum"]
            per[w] = {"lookups": int(acc), "frac": round(acc / (k_ms[w] * 1e-3) / 1e9 / L1_PEAK_GACC_S, 4)}
            if "TCP_TCC_READ_REQ_sum" in pmc[w]:
                per[w]["requests_to_l2_per_lookup"] = round(pmc[w]["TCP_TCC_READ_REQ_sum"] / max(1.0, acc), 4)
        tot = sum(per[w]["lookups"] for w in waves)
        tot_s = sum(k_ms[w] for w in waves) * 1e-3
        res["l1"] = {"lookups_per_launch": int(tot / 2), "achieved_Glookups_s": round(tot / tot_s / 1e9, 1), "peak_Glookups_s": L1_PEAK_GACC_S,
                     "peak_definition": "measured: tools/ubench/node_fetch.hip, every lane fetching scattered 16-byte pieces (L2-resident table)",
                     "frac": round(tot / tot_s / 1e9 / L1_PEAK_GACC_S, 4), "per_wave": per}
    if all("SQ_THREAD_CYCLES_VALU" in pmc[w] and "SQ_INSTS_VALU" in pmc[w] for w in waves):
        per = {}
        for w in waves:
            lane_ops, insts = pmc[w]["SQ_THREAD_CYCLES_VALU"], pmc[w]["SQ_INSTS_VALU"]
            secs = k_ms[w] * 1e-3
            per[w] = {"lane_ops": int(lane_ops), "wave_insts": int(insts), "frac": round(lane_ops / secs / lane_peak, 4),
                      "lane_util": round(lane_ops / (64.0 * insts), 4),
                      # a wave64 instruction holds its SIMD-32 for 2 cycles (packed / 3-input forms longer: a lower bound)
                      "issue_busy": round(insts * VALU_CYCLES_PER_WAVE_INST / (simds * secs * CLOCK_GHZ * 1e9), 4)}
            if "SQ_WAIT_ANY" in pmc[w] and pmc[w].get("SQ_WAVE_CYCLES"):
                per[w]["wait_frac_of_wave_cycles"] = round(pmc[w]["SQ_WAIT_ANY"] / pmc[w]["SQ_WAVE_CYCLES"], 4)
            if "SQ_LDS_BANK_CONFLICT" in pmc[w]:
                per[w]["lds_bank_conflict_cycles"] = int(pmc[w]["SQ_LDS_BANK_CONFLICT"])
            if pmc[w].get("GRBM_GUI_ACTIVE") and pmc.get("profiled_us", {}).get(w):
                per[w]["effective_clock_GHz_under_profiler"] = round(pmc[w]["GRBM_GUI_ACTIVE"] / N_XCD / (pmc["profiled_us"][w] * 1e3), 3)
        tot_ops = sum(per[w]["lane_ops"] for w in waves)
        tot_s = sum(k_ms[w] for w in waves) * 1e-3
        res["valu"] = {"lane_ops_per_launch": int(tot_ops / 2), "achieved_Tlaneops": round(tot_ops / tot_s / 1e12, 3),
                       "peak_Tlaneops": round(lane_peak / 1e12, 2),
                       "peak_definition": "%d CUs x 4 SIMDs x %d lanes x %.1f GHz (max clock)" % (n_cus, VALU_LANES_PER_SIMD, CLOCK_GHZ),
                       "frac": round(tot_ops / tot_s / lane_peak, 4),
                       "lane_util": round(tot_ops / (64.0 * sum(per[w]["wave_insts"] for w in waves)), 4),
                       "issue_busy": round(sum(per[w]["wave_insts"] for w in waves) * VALU_CYCLES_PER_WAVE_INST / (simds * tot_s * CLOCK_GHZ * 1e9), 4),
                       "per_wave": per}
    return res


def compact_roofline(r, per_wave_counts):
    """The per-config form of the counters: per wave {ms, hbm / valu / l1 fractions, lane utilisation, waiting share, L2 hit
    rate is per config} — every number recomputable from the raw rows kept under --pmc-dir."""
    out = {"waves": {}}
    for w in ("primary", "bounce"):
        e = dict(per_wave_counts.get(w, {}))
        if r.get("hbm"):
            e["hbm_bytes"] = r["hbm"]["per_wave_bytes"][w]
            e["hbm_frac"] = r["hbm"]["per_wave_frac"][w]
        if r.get("valu"):
            pw = r["valu"]["per_wave"][w]
            e.update({"valu_frac": pw["frac"], "lane_util": pw["lane_util"], "issue_busy": pw["issue_busy"]})
            if "wait_frac_of_wave_cycles" in pw:
                e["wait"] = pw["wait_frac_of_wave_cycles"]
        if r.get("l1"):
            e["l1_frac"] = r["l1"]["per_wave"][w]["frac"]
            e["l1_requests_to_l2_per_lookup"] = r["l1"]["per_wave"][w].get("requests_to_l2_per_lookup")
        out["waves"][w] = e
    if r.get("hbm"):
        out["hbm"] = {k: r["hbm"][k] for k in ("bytes_per_launch", "achieved_GBs", "frac") if k in r["hbm"]}
        if "l2_hit_rate" in r["hbm"]:
            out["l2_hit_rate"] = r["hbm"]["l2_hit_rate"]
    if r.get("valu"):
        out["valu"] = {k: r["valu"][k] for k in ("achieved_Tlaneops", "frac", "lane_util", "issue_busy")}
    if r.get("l1"):
        out["l1"] = {k: r["l1"][k] for k in ("achieved_Glookups_s", "frac")}
    fr = {k: out[k]["frac"] for k in ("hbm", "valu", "l1") if k in out}
    if fr:
        out["most_loaded"] = max(fr, key=fr.get)
    return out

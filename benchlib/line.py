"""The ONE JSON line the driver parses.  Everything the run measured goes to the side file (`extras_file`); the line carries the
contract's keys, `roofline`, `cpu_baseline` and a few figures of merit, and is kept far below the size at which a reader truncates it
(round 5's 20 KB line was not parsed: BENCH_r05.json).  tests/test_bench_line.py feeds canned results through this module."""
import json
import os

from . import ROOT

LINE_BUDGET = 6000  # bytes; the hard ceiling asserted by the tests is 8192

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")


def _parity_ok(p):
    """a parity() dict or a {"primary": ..., "bounce": ...} pair of them -> within tolerance?"""
    if p is None:
        return None
    if "within_tolerance_1e-5" in p:
        return bool(p["within_tolerance_1e-5"])
    subs = [v for v in p.values() if isinstance(v, dict) and "within_tolerance_1e-5" in v]
    return all(bool(v["within_tolerance_1e-5"]) for v in subs) if subs else None


def compact_cpu_baseline(cb):
    if not cb:
        return None
    out = {k: cb[k] for k in ("value", "unit", "cores", "kind", "build_ms", "value_march_x86_64_v3", "value_on_gpu_built_tree",
                              "parity_same_tree_bit_identical") if k in cb}
    out["sample"] = cb.get("sample_short") or str(cb.get("sample", ""))[:160]
    for k in ("parity_own_trees", "parity_same_tree"):
        if k in cb:
            out[k + "_within_1e-5"] = _parity_ok(cb[k])
    if "parity_own_trees" in cb:
        out["parity_rays"] = int(sum(v.get("rays", 0) for v in cb["parity_own_trees"].values() if isinstance(v, dict)))
    return out


def compact_roofline(r):
    if not r:
        return None
    return {k: v for k, v in r.items() if k != "detail"}


def compact_config(e):
    if "error" in e:
        return {"error": str(e["error"])[:80]}
    out = {k: e[k] for k in ("value", "ms_per_step", "build_ms", "dtype") if k in e}
    p = e.get("parity") or {}
    if "same_tree_bit_identical" in p:
        out["bit_identical"] = p["same_tree_bit_identical"]
    rf = e.get("roofline") or {}
    if "most_loaded" in rf:
        out["bound"] = rf["most_loaded"]
        out["frac"] = rf.get(rf["most_loaded"], {}).get("frac")
    return out


def compact_rows(rows):
    out = {}
    for name, e in rows.items():
        if name.startswith("_") or not isinstance(e, dict):
            continue
        if "error" in e:
            out[name] = {"error": str(e["error"])[:60]}
            continue
        c = {"value": e.get("value")}
        par = e.get("parity")
        if isinstance(par, dict):  # the row's own same-run parity sample: every boolean in it must hold
            flags = [v for v in par.values() if isinstance(v, bool)]
            if flags:
                c["parity"] = all(flags)
        out[name] = c
    return out


def compact_line(full, extras_file=None):
    """The driver's line from the full result object."""
    line = {k: full[k] for k in CONTRACT_KEYS if k in full}
    line["config"] = full["config"]
    for k in ("build_ms", "build_host_ms", "bvh"):
        if k in full:
            line[k] = full[k]
    line["roofline"] = compact_roofline(full.get("roofline"))
    line["cpu_baseline"] = compact_cpu_baseline(full.get("cpu_baseline"))
    if "multi_gpu" in full:
        line["multi_gpu"] = full["multi_gpu"]
    if "strong_c4" in full:
        s = full["strong_c4"]
        line["strong_c4"] = {k: s[k] for k in ("value", "unit", "ms_per_step", "rays_per_step", "build_ms", "scaling") if k in s}
    also = {}
    for k in ("pipelined", "multi_batch", "opt_in_distance_order", "primary_plus_shadow"):
        if isinstance(full.get(k), dict) and "value" in full[k]:
            also[k] = full[k]["value"]
    if also:
        also["unit"] = "Mrays/s (same work, other launch shapes; never `value`)"
        line["also"] = also
    if full.get("configs"):
        line["configs"] = {n: compact_config(e) for n, e in full["configs"].items()}
    if isinstance(full.get("next_rows"), dict):
        line["next_rows"] = compact_rows(full["next_rows"])
    if extras_file:
        line["extras_file"] = extras_file
    # the guard: optional groups go first, then the optional parts of the two required objects
    for victim in ("next_rows", "also", "configs", "strong_c4"):
        if len(json.dumps(line)) <= LINE_BUDGET:
            break
        line.pop(victim, None)
    if len(json.dumps(line)) > LINE_BUDGET and isinstance(line.get("multi_gpu"), dict):
        line["multi_gpu"] = {k: v for k, v in line["multi_gpu"].items() if not isinstance(v, (list, dict))}
    if len(json.dumps(line)) > LINE_BUDGET:
        line["config"] = {k: (v if not isinstance(v, str) else v[:120]) for k, v in line["config"].items()}
    return line


def write_extras(full, path=None):
    """The full result object beside the line.  Default: gpurun_out/bench_extras.json under the repo (merged back by gpurun), /tmp if
    that is not writable.  Returns the path written (relative to the repo where possible) or None."""
    cands = [path] if path else [os.path.join(ROOT, "gpurun_out", "bench_extras.json"), "/tmp/bench_extras.json"]
    for p in cands:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(p)), exist_ok=True)
            with open(p, "w") as f:
                json.dump(full, f, indent=1)
                f.write("\n")
            ap = os.path.abspath(p)
            return os.path.relpath(ap, ROOT) if ap.startswith(ROOT + os.sep) else ap
        except OSError:
            continue
    return None

"""Sweep traversal tunables (nrtSetTunable) on bench configs: kernel ms of the primary and the bounce wave (the kernel's own
stamps) and the time of a step measured like the headline (launches back to back, one event pair).  The combinations are
measured round-robin, ROUNDS times, and the medians reported — differences of 1-2 % need that.

    python tools/tune_probe.py C3,C2 "dict()" "dict(static_bands=1)" "dict(refill_min=32, leaf_min=16)"
"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402

import bench  # noqa: E402

ROUNDS = int(os.environ.get("ROUNDS", "5"))
names = sys.argv[1].split(",")
combos = [eval(x) for x in sys.argv[2:]] or [dict()]
for name in names:
    wl = bench.Workload(name, builds=1)
    a = wl.accel
    defaults = {}
    for combo in combos:
        for k in combo:
            if k not in defaults:
                defaults[k] = a.GetTunable(k)
    res = [dict(t1=[], t2=[], step=[], hsh=None) for _ in combos]
    for _ in range(60):  # (round 6: the first combination of a run read ~2 % low — the device's clocks ramp over some tens of milliseconds of work)
        a.TraverseBatchDevice(wl.d_rays1, wl.d_hits1, wl.d_mask1)
        a.TraverseBatchDevice(wl.d_rays2, wl.d_hits2, wl.d_mask2)
    torch.cuda.synchronize()
    for rnd in range(ROUNDS):
        for ci, combo in enumerate(combos):
            for k, v in defaults.items():
                a.SetTunable(k, combo.get(k, v))
            for _ in range(3):
                a.TraverseBatchDevice(wl.d_rays1, wl.d_hits1, wl.d_mask1)
                res[ci]["t1"].append(a.LastTraverseMs())
                a.TraverseBatchDevice(wl.d_rays2, wl.d_hits2, wl.d_mask2)
                res[ci]["t2"].append(a.LastTraverseMs())
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            torch.cuda.synchronize()
            ev[0].record()
            for _ in range(10):
                a.TraverseBatchDevice(wl.d_rays1, wl.d_hits1, wl.d_mask1)
                a.TraverseBatchDevice(wl.d_rays2, wl.d_hits2, wl.d_mask2)
            ev[1].record()
            torch.cuda.synchronize()
            res[ci]["step"].append(ev[0].elapsed_time(ev[1]) / 10)
            if rnd == 0:
                res[ci]["hsh"] = hashlib.md5(wl.d_hits1.cpu().numpy().tobytes() + wl.d_hits2.cpu().numpy()[: wl.n2 * wl.HIT.itemsize].tobytes()).hexdigest()
    for ci, combo in enumerate(combos):
        r = res[ci]
        step = float(np.median(r["step"]))
        print("%-6s %-60s primary %.4f  bounce %.4f  step %.4f ms (min %.4f)  %.0f Mrays/s  same=%s" % (
            name, combo, float(np.median(r["t1"])), float(np.median(r["t2"])), step, min(r["step"]), (wl.n1 + wl.n2) / step / 1e3, r["hsh"] == res[0]["hsh"]), flush=True)
    del wl
    torch.cuda.empty_cache()

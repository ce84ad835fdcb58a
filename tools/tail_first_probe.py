"""How much of a launch is the tail of its longest rays?  Per-ray costs come from the profiling kernel (tunable debug = 96:
u <- node steps, v <- triangle tests); the rays whose cost is in the top p % are moved to the FRONT of the batch (both groups
keep their original order, so coherence is kept) and the production kernel is timed.  An oracle experiment: a real batch does
not know its costs — the answer bounds what any predictor-driven ordering could win.

    python tools/tail_first_probe.py [C3 C2 C4tile]
"""
import os

os.environ["NRT_USE_PROF_LIB"] = "1"  # the profiling build of the library (include/nanort_hip_prof.h)
import sys

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402

import bench  # noqa: E402


def timeit(a, rays, reps=9):
    d = torch.from_numpy(np.ascontiguousarray(rays).view(np.uint8)).cuda()
    o = torch.empty(len(rays) * 16, dtype=torch.uint8, device="cuda")
    ts = []
    for _ in range(reps):
        a.TraverseBatchDevice(d, o)
        ts.append(a.LastTraverseMs())
    return float(np.median(ts)), o


for name in ([x for x in sys.argv[1:]] or ["C3"]):
    wl = bench.Workload(name, builds=1)
    a = wl.accel
    for wave, rays in (("bounce", wl.rays2), ("primary", wl.rays1)):
        a.SetTunable("debug", 96)
        _, o = timeit(a, rays, reps=1)
        h = o.cpu().numpy().view(wl.HIT)
        cost = h["u"] + h["v"]
        a.SetTunable("debug", 0)
        base, _ = timeit(a, rays)
        print("%s %-8s as given %.4f ms | cost per ray: mean %.1f, p50 %.0f, p90 %.0f, p99 %.0f, max %.0f" % (
            name, wave, base, cost.mean(), np.percentile(cost, 50), np.percentile(cost, 90), np.percentile(cost, 99), cost.max()), flush=True)
        for pct in (1, 5, 10, 25, 50):
            thr = np.percentile(cost, 100 - pct)
            long_ = cost >= thr
            first = np.concatenate([np.nonzero(long_)[0], np.nonzero(~long_)[0]])
            last = np.concatenate([np.nonzero(~long_)[0], np.nonzero(long_)[0]])
            ms_f, _ = timeit(a, rays[first])
            ms_l, _ = timeit(a, rays[last])
            print("    longest %2d %% first: %.4f ms (x%.3f)   last: %.4f ms (x%.3f)" % (pct, ms_f, base / ms_f, ms_l, base / ms_l), flush=True)
    del wl
    torch.cuda.empty_cache()

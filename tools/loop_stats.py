"""Loop occupancy and time split of the traversal kernel (profiling instantiation, tunable debug = 32) on a bench config:
per phase the wave iterations per 64 rays and the average number of active lanes, and which share of a wave's clock
ticks goes to refilling, the inner-node phase and the leaf phase.

    python tools/loop_stats.py [C3] [refill_min=32 static_bands=1 ...]      (name=value: nrtSetTunable)
"""
import os

os.environ["NRT_USE_PROF_LIB"] = "1"  # the profiling build of the library (include/nanort_hip_prof.h)
import ctypes
import sys

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402

import bench  # noqa: E402
from nanort_amd import capi  # noqa: E402

cfg = [a for a in sys.argv[1:] if "=" not in a] or ["C3"]
tun = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
L = capi.lib()
L.nrtDebugCounters.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
for name in cfg:
    wl = bench.Workload(name, builds=1)
    a = wl.accel
    for k, v in tun.items():
        a.SetTunable(k, int(v))
    for wave, d, o, n in (("primary", wl.d_rays1, wl.d_hits1, wl.n1), ("bounce", wl.d_rays2, wl.d_hits2, wl.n2)):
        a.SetTunable("debug", 0)
        ts = []
        for _ in range(5):
            a.TraverseBatchDevice(d, o)
            ts.append(a.LastTraverseMs())
        a.SetTunable("debug", 32)
        a.TraverseBatchDevice(d, o)
        ms_stats = a.LastTraverseMs()
        c = np.zeros(16, dtype=np.uint64)
        L.nrtDebugCounters(a._h, c.ctypes.data_as(ctypes.c_void_p), 16)
        it1, act1, idle2, it2, act2, refills, refilled, ent2, t_ref, t_p1, t_p2, act2b = [int(x) for x in c[:12]]
        it1_s, act1_s, it2_s, act2_s = [int(x) for x in c[12:16]]
        g = n / 64.0
        tt = max(1, t_ref + t_p1 + t_p2)
        print("%s %-8s %s  production %.4f ms (profiling variant %.4f ms, %s)" % (name, wave, tun, float(np.median(ts)), ms_stats, a.LastKernelName()))
        print("    inner-node phase: %.1f wave-iterations per 64 rays, %.1f lanes active | leaf phase: %.1f trips per 64 rays, %.1f lanes with a "
              "first record, %.1f with a second; entered %.2f times per 64 rays with %.1f lanes idle | refills %.2f per 64 rays, %.1f lanes each"
              % (it1 / g, act1 / max(1, it1), it2 / g, act2 / max(1, it2), act2b / max(1, it2), ent2 / g, idle2 / max(1, ent2), refills / g, refilled / max(1, refills)))
        # steady state (rays still being handed out) against the drain (every wave finishing what it holds)
        d_it1, d_act1, d_it2, d_act2 = it1 - it1_s, act1 - act1_s, it2 - it2_s, (act2 + act2b) - act2_s
        print("    steady: %.1f %% of the inner iterations at %.1f lanes, %.1f %% of the leaf trips at %.1f records (of 128) | drain: inner %.1f lanes, leaf %.1f records"
              % (100.0 * it1_s / max(1, it1), act1_s / max(1, it1_s), 100.0 * it2_s / max(1, it2), act2_s / max(1, it2_s), d_act1 / max(1, d_it1), d_act2 / max(1, d_it2)))
        print("    wave time: refill %.1f %%, inner-node phase %.1f %%, leaf phase %.1f %%  (ticks per 64 rays: %.0f / %.0f / %.0f; per refill %.0f, per inner iteration %.0f, per leaf trip %.0f)"
              % (100.0 * t_ref / tt, 100.0 * t_p1 / tt, 100.0 * t_p2 / tt, t_ref / g, t_p1 / g, t_p2 / g, t_ref / max(1, refills), t_p1 / max(1, it1), t_p2 / max(1, it2)), flush=True)
    del wl
    torch.cuda.empty_cache()

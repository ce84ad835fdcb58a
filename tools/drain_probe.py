"""When do the waves of a traversal launch run dry and finish?  (tunable debug bit 8192: per-wave realtime stamps, 100 MHz.)
Prints, for the primary and bounce waves of a bench config: launch span, when the first / median / last wave ran out of rays,
mean and longest drain (out of rays -> done), and the share of wave-time that is idle before the launch ends.

    python tools/drain_probe.py [C3] [static_bands=1 ...]
"""
import os

os.environ["NRT_USE_PROF_LIB"] = "1"  # the profiling build of the library (include/nanort_hip_prof.h)
import ctypes
import sys

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402

import bench  # noqa: E402

cfg = [a for a in sys.argv[1:] if "=" not in a] or ["C3"]
tun = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
for name in cfg:
    wl = bench.Workload(name, builds=1)
    a = wl.accel
    for k, v in tun.items():
        a.SetTunable(k, int(v))
    a.SetTunable("debug", a.GetTunable("debug") | 8192)
    a._L.nrtDebugWaveClocks.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
    a._L.nrtDebugWaveClocks.restype = ctypes.c_long
    for wave, d, o in (("primary", wl.d_rays1, wl.d_hits1), ("bounce", wl.d_rays2, wl.d_hits2)):
        for rep in range(3):
            a.TraverseBatchDevice(d, o)
            torch.cuda.synchronize()
            buf = np.zeros((16384, 3), dtype=np.uint64)
            n = a._L.nrtDebugWaveClocks(a._h, buf.ctypes.data_as(ctypes.c_void_p), 16384)
            c = buf[:n].astype(np.float64) / 100.0  # microseconds
            t0 = c[:, 0].min()
            dry, end = c[:, 1] - t0, c[:, 2] - t0
            q = lambda x, p: float(np.percentile(x, p))  # noqa: E731
            print("%s %s %s (%s): span %.1f us (stamps %.1f) | start spread %.1f | dry: first %.1f p10 %.1f median %.1f p90 %.1f last %.1f | done: p10 %.1f median %.1f p90 %.1f last %.1f | drain mean %.1f longest %.1f | idle wave-time before the end %.1f%%" % (
                name, wave, tun, a.LastKernelName()[-22:], end.max(), a.LastTraverseMs() * 1e3, (c[:, 0] - t0).max(), dry.min(), q(dry, 10), q(dry, 50), q(dry, 90), dry.max(),
                q(end, 10), q(end, 50), q(end, 90), end.max(), (end - dry).mean(), (end - dry).max(), 100.0 * (end.max() - end).sum() / (end.max() * n)), flush=True)
    del wl
    torch.cuda.empty_cache()

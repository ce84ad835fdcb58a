"""One bench config traced with given tunables, for tools/variant_pmc.sh (run under rocprofv3):
    python tools/pmc_child.py C3 "dict(order4=1)"      -> 1 set-up launch, then 4 x (primary, bounce)"""
import os
import sys

if "debug" in " ".join(sys.argv[2:]):
    os.environ["NRT_USE_PROF_LIB"] = "1"  # loop counters: the profiling build of the library (include/nanort_hip_prof.h)
import sys

sys.path.insert(0, ".")
import torch  # noqa: E402

import bench  # noqa: E402

name, combo = sys.argv[1], eval(sys.argv[2])
wl = bench.Workload(name, builds=1)
for k, v in combo.items():
    wl.accel.SetTunable(k, v)
for _ in range(4):
    wl.accel.TraverseBatchDevice(wl.d_rays1, wl.d_hits1, wl.d_mask1)
    wl.accel.TraverseBatchDevice(wl.d_rays2, wl.d_hits2, wl.d_mask2)
torch.cuda.synchronize()
print("kernel", wl.accel.LastKernelName(), "n1", wl.n1, "n2", wl.n2, flush=True)
if combo.get("debug", 0) & 32:
    import ctypes

    import numpy as np

    c = np.zeros(16, dtype=np.uint64)
    wl.accel._L.nrtDebugCounters.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    wl.accel._L.nrtDebugCounters(wl.accel._h, c.ctypes.data_as(ctypes.c_void_p), 16)
    print("debug counters of the last (bounce) launch per ray:", (c[:8] / max(1, wl.n2)).round(3).tolist(), flush=True)

"""A short run of the PROFILING library (roctx ranges compiled in: common.h NRT_RANGE) for
    rocprofv3 --marker-trace --kernel-trace --stats -d gpurun_out/r06_roctx -o t --output-format csv -- python tools/roctx_trace.py [C3]
two builds and three steps (primary + bounce) of a bench config: every build phase and every traversal launch is a named range."""
import os
import sys

os.environ["NRT_USE_PROF_LIB"] = "1"
sys.path.insert(0, ".")
import torch  # noqa: E402

import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
wl = bench.Workload(name, builds=2)
for _ in range(3):
    wl.accel.TraverseBatchDevice(wl.d_rays1, wl.d_hits1, wl.d_mask1)
    wl.accel.TraverseBatchDevice(wl.d_rays2, wl.d_hits2, wl.d_mask2)
torch.cuda.synchronize()
print("roctx_trace done:", name, wl.accel.LastKernelName())

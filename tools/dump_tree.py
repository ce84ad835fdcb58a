"""Save the GPU-built tree of a bench config (node array + index permutation) for CPU-side model experiments:
    python tools/dump_tree.py C3 gpurun_out/c3_tree.npz"""
import sys

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402

name, out = sys.argv[1], sys.argv[2]
wl = bench.Workload(name, builds=1)
nodes, idx = wl.accel.GetTree()
np.savez_compressed(out, nodes=nodes, idx=idx)
print(name, nodes.shape, idx.shape, wl.accel.GetStatistics())

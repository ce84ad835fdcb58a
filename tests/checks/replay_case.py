"""Replay a saved fuzz case on the GPU in both walks and save the records:  python tests/checks/replay_case.py case.npz out.npz"""
import sys

import numpy as np

sys.path.insert(0, '.')
from nanort_amd import BVHAccel, TriangleMesh  # noqa: E402

d = np.load(sys.argv[1])
v, f, rays, opts, nodes, idx = d["v"], d["f"], d["rays"], d["opts"], d["nodes"], d["idx"]
out = {}
for mode in (0, 1):
    a = BVHAccel(v.dtype.type)
    a.SetTunable("order4", mode)
    a.SetMesh(TriangleMesh(v, f))
    a.SetTree(nodes, idx)
    h, m = a.TraverseBatch(rays, opts)
    out["h%d" % mode], out["m%d" % mode] = h, m
    print(mode, a.LastKernelName(), int(m.sum()))
np.savez(sys.argv[2], **out)

"""Beyond BASELINE.json's sizes: build a Plane(nx, ny) of tens of millions of triangles on the GPU, trace 1920x1080 primaries + one
bounce with the default (two-level) walk and with the literal 40-byte-node loop (tunable wide = 0: an independent kernel over the
reference-format array) and compare every record — the arrays' 32-bit offsets, the packed leaf references and the builder's
workspace at several times the largest size the test suite runs.

    python tools/big_mesh_probe.py 10000 3200      # 64 M triangles
"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from nanort_amd import BVHAccel, TriangleMesh, scenes  # noqa: E402

nx, ny = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time()
v, f = scenes.plane(nx, ny)
print("Plane(%d, %d): %d triangles, %d vertices, generated in %.1f s" % (nx, ny, f.shape[0], v.shape[0], time.time() - t0), flush=True)
a = BVHAccel(np.float32)
m = TriangleMesh(v, f)
ts = []
for _ in range(3):
    assert a.Build(m.num_faces, m)
    ts.append(a.LastBuildMs())
st = a.GetStatistics()
print("build ms %s; nodes %d (branches %d, leaves %d), depth %d" % (["%.2f" % t for t in ts], st["num_leaf_nodes"] + st["num_branch_nodes"],
                                                                      st["num_branch_nodes"], st["num_leaf_nodes"], st["max_tree_depth"]), flush=True)
rays = scenes.camera_rays(1920, 1080)
h, mk = a.TraverseBatch(rays)
k1 = a.LastKernelName()
t1 = a.LastTraverseMs()
bounce = scenes.secondary_rays("bounce", v, f, rays, h, mk)
hb, mb = a.TraverseBatch(bounce)
a.SetTunable("wide", 0)
h0, m0 = a.TraverseBatch(rays)
k0 = a.LastKernelName()
hb0, mb0 = a.TraverseBatch(bounce)
same = h.tobytes() == h0.tobytes() and (mk == m0).all() and hb.tobytes() == hb0.tobytes() and (mb == mb0).all()
print("primaries: %d hits of %d; %s vs %s: records identical (primary + bounce): %s" % (int(mk.sum()), rays.shape[0], k1, k0, same), flush=True)
a.SetTunable("wide", 1)
import torch  # noqa: E402

d = torch.from_numpy(rays.view(np.uint8)).cuda()
o = torch.empty(rays.shape[0] * 16, dtype=torch.uint8, device="cuda")
tt = []
for _ in range(5):
    a.TraverseBatchDevice(d, o)
    tt.append(a.LastTraverseMs())
print("primary wave %.3f ms = %.0f Mrays/s" % (float(np.median(tt)), rays.shape[0] / float(np.median(tt)) / 1e3), flush=True)
sys.exit(0 if same else 1)

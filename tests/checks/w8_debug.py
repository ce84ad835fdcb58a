"""Staged bring-up of the 8-wide walk (each stage under its own `timeout` on the GPU box):
    python tests/checks/w8_debug.py layout|trace [mesh]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from nanort_amd import BVHAccel, TriangleMesh, scenes  # noqa: E402
from oracle.bindings import Oracle  # noqa: E402

stage = sys.argv[1]
mesh = sys.argv[2] if len(sys.argv) > 2 else "plane"
v, f = {"plane": lambda: scenes.plane(120, 77), "c1": scenes.load_c1_mesh, "tiny": lambda: scenes.plane(3, 2),
        "c3": lambda: scenes.plane(1000, 500)}[mesh]()
orc = Oracle()
a = BVHAccel(np.float32)
a.SetTunable("wide8", 1)
t0 = time.time()
assert a.Build(f.shape[0], TriangleMesh(v, f))
print("build ok %.3fs, device %.3f ms" % (time.time() - t0, a.LastBuildMs()), flush=True)
nodes, idx = a.GetTree()
gn, gr = a.GetWide8()
print("layout: %d wide nodes, %d leaf records" % (gn.shape[0], gr.shape[0]), flush=True)
model = orc.wide8_build(nodes, idx, v, f)
if stage == "layout":
    from test_gpu_wide8 import compare_layouts

    compare_layouts(gn, gr, *model.arrays())
    print("layout equals the model's", flush=True)
    for _ in range(3):
        assert a.Build(f.shape[0], TriangleMesh(v, f))
        print("rebuild device %.3f ms" % a.LastBuildMs(), flush=True)
else:
    rays = scenes.camera_rays(160, 120)
    h, m = a.TraverseBatch(rays)
    print("traced with", a.LastKernelName(), int(m.sum()), "hits", flush=True)
    mh, mm, c = model.traverse(rays, order_mode=1, cull_mode=1)
    print("model:", int(mm.sum()), "hits; identical:", h.tobytes() == mh.tobytes() and (m == mm).all(), flush=True)
    oh, om = orc.traverse(nodes, idx, v, f, rays)
    print("oracle: mask equal", (m == om).all(), "t equal", h["t"].tobytes() == oh["t"].tobytes(), "prim diffs", int((h["prim_id"] != oh["prim_id"]).sum()), flush=True)

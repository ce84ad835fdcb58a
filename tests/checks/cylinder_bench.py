"""SURVEY §8f row 4 measurement (cylinders): the cylinder example's workload (examples/cylinder_primitive/main.cc) at
scale — n random cylinders, the example's camera at 1920x1080 — GPU build + traversal through the C ABI, the unmodified
example on the host cores beside it (oracle/_ref/libcylinder_ref.so, when built), parity in the same run."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
from nanort_amd import BVHAccel, CylinderGeometry, scenes
from nanort_amd.wire import CYL_HIT_F32

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000  # the example's cylinders span the whole box: keep n modest
W, H = 1920, 1080
v, r = scenes.random_cylinders(n)
rays = scenes.particle_camera_rays(W, H)
a = BVHAccel(np.float32)
bms = []
for _ in range(4):
    assert a.Build(n, CylinderGeometry(v, r)); bms.append(a.LastBuildMs())
st = a.GetStatistics()
d = torch.from_numpy(rays.view(np.uint8)).cuda(); o = torch.empty(len(rays) * 28, dtype=torch.uint8, device='cuda'); m = torch.empty(len(rays), dtype=torch.uint8, device='cuda')
ts = []
for _ in range(8):
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); a.TraverseBatchDevice(d, o, m); t1.record(); torch.cuda.synchronize(); ts.append(t0.elapsed_time(t1))
ms = float(np.median(ts[2:]))
out = {"workload": "cylinder example: %d cylinders, %dx%d camera" % (n, W, H), "build_ms": round(float(np.median(bms[1:])), 3),
       "nodes": int(st["num_leaf_nodes"] + st["num_branch_nodes"]), "depth": int(st["max_tree_depth"]),
       "traverse_ms_incl_normal_pass": round(ms, 4), "Mrays_per_s": round(len(rays) / ms / 1e3, 1), "hits": int(m.sum().item())}
try:
    from oracle.bindings import CylinderReference, have_cylinder_reference
    if have_cylinder_reference():
        R = CylinderReference()
        t0 = time.perf_counter(); nodes, idx, rst = R.build(v, r); tb = time.perf_counter() - t0
        sub = np.ascontiguousarray(rays.reshape(H, W)[::16].reshape(-1))  # bounded CPU sample: every 16th row
        t0 = time.perf_counter(); rh, rm = R.traverse(sub); tt = time.perf_counter() - t0
        out["cpu_reference"] = {"build_ms": round(tb * 1e3, 1), "Mrays_per_s": round(len(sub) / tt / 1e6, 3), "threads": os.cpu_count(),
                                "nodes": int(nodes.shape[0]), "depth": rst["max_tree_depth"], "sample": "every 16th row (%d rays)" % len(sub)}
        b = BVHAccel(np.float32); b.SetMesh(CylinderGeometry(v, r)); b.SetTree(nodes, idx)
        bh, bm = b.TraverseBatch(sub)
        out["parity_on_reference_tree"] = {"mask_equal": bool(np.array_equal(bm, rm)), "records_bit_equal": bool(bh.tobytes() == rh.tobytes())}
except Exception as e:  # pragma: no cover
    out["cpu_reference_error"] = repr(e)
print(json.dumps(out))

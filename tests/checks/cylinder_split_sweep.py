"""Sweep of the cylinder segmentation knobs (tunables cyl_split / cyl_seg_radii) on the cylinder example's workload:
build ms, tree size, traversal rate, and the records against (a) the restated example over the same tree (every field)
and (b) the unmodified example's own records on its own tree, where oracle/_ref is built (flags and t).
    python tests/checks/cylinder_split_sweep.py [n]"""
import json, sys
import numpy as np
sys.path.insert(0, '.')
import torch
from nanort_amd import BVHAccel, CylinderGeometry, scenes
from oracle import bindings as ob

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
W, H = 1920, 1080
v, r = scenes.random_cylinders(n)
rays = scenes.particle_camera_rays(W, H)
d = torch.from_numpy(rays.view(np.uint8)).cuda(); o = torch.empty(len(rays) * 28, dtype=torch.uint8, device='cuda'); m = torch.empty(len(rays), dtype=torch.uint8, device='cuda')
ref = None
try:
    if ob.have_cylinder_reference():
        R = ob.CylinderReference()
        R.build(v, r)
        sub = np.ascontiguousarray(rays.reshape(H, W)[::24].reshape(-1))
        ref = (sub,) + tuple(R.traverse(sub))
except Exception as e:
    print("no reference:", repr(e))
for split, seg in ((1, 8), (4, 8), (8, 8), (16, 8), (32, 8), (64, 8), (32, 4), (32, 16), (64, 4), (64, 2), (64, 16)):
    a = BVHAccel(np.float32)
    a.SetTunable("cyl_split", split); a.SetTunable("cyl_seg_radii", seg)
    bms = []
    for _ in range(3):
        assert a.Build(n, CylinderGeometry(v, r)); bms.append(a.LastBuildMs())
    nodes, idx = a.GetTree()
    ts = []
    for _ in range(6):
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record(); a.TraverseBatchDevice(d, o, m); t1.record(); torch.cuda.synchronize(); ts.append(t0.elapsed_time(t1))
    ms = float(np.median(ts[2:]))
    out = {"cyl_split": split, "cyl_seg_radii": seg, "segments": int(idx.shape[0]), "nodes": int(nodes.shape[0]), "build_ms": round(float(np.median(bms[1:])), 3),
           "ms": round(ms, 3), "Mrays_s": round(len(rays) / ms / 1e3, 1)}
    sub = rays[::37]
    h, mk = a.TraverseBatch(sub)
    oh, om = ob.CylinderOracle().traverse(nodes, idx, v, r, sub)
    out["same_tree_identical"] = bool(np.array_equal(mk, om) and all(np.ascontiguousarray(h[f]).tobytes() == np.ascontiguousarray(oh[f]).tobytes() for f in ("t", "u", "v", "prim_id", "normal")))
    if ref is not None:
        gh, gm = a.TraverseBatch(ref[0])
        out["vs_reference_own_tree"] = {"rays": int(ref[0].shape[0]), "flag_mismatches": int((gm != ref[2]).sum()), "t_mismatches": int((gh["t"] != ref[1]["t"]).sum()),
                                        "prim_mismatches": int((gh["prim_id"] != ref[1]["prim_id"]).sum())}
    print(json.dumps(out), flush=True)

"""SURVEY §8f row 4 measurement: the particle example's workload (examples/particle_primitive/main.cc) at scale —
n random spheres, the example's camera at 1920x1080 — GPU build + traversal through the C ABI, the unmodified
example on the host cores beside it (oracle/_ref/libsphere_ref.so, when built), parity in the same run."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
from nanort_amd import BVHAccel, SphereGeometry, scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
W, H = 1920, 1080
c, r = scenes.random_spheres(n)
rays = scenes.particle_camera_rays(W, H)
a = BVHAccel(np.float32)
bms = []
for _ in range(4):
    assert a.Build(n, SphereGeometry(c, r)); bms.append(a.LastBuildMs())
st = a.GetStatistics()
d = torch.from_numpy(rays.view(np.uint8)).cuda(); o = torch.empty(len(rays) * 16, dtype=torch.uint8, device='cuda'); m = torch.empty(len(rays), dtype=torch.uint8, device='cuda')
ts = []
for _ in range(8):
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); a.TraverseBatchDevice(d, o, m); t1.record(); torch.cuda.synchronize(); ts.append(t0.elapsed_time(t1))
ms = float(np.median(ts[2:]))
out = {"workload": "particle example: %d spheres, %dx%d camera" % (n, W, H), "build_ms": round(float(np.median(bms[1:])), 3),
       "nodes": int(st["num_leaf_nodes"] + st["num_branch_nodes"]), "depth": int(st["max_tree_depth"]),
       "traverse_ms_incl_uv_pass": round(ms, 4), "Mrays_per_s": round(len(rays) / ms / 1e3, 1), "hits": int(m.sum().item())}
try:
    from oracle.bindings import SphereReference, have_sphere_reference
    if have_sphere_reference():
        R = SphereReference()
        t0 = time.perf_counter(); nodes, idx, rst = R.build(c, r); tb = time.perf_counter() - t0
        t0 = time.perf_counter(); rh, rm = R.traverse(rays); tt = time.perf_counter() - t0
        from nanort_amd.wire import HIT_F32
        gh = o.cpu().numpy().view(HIT_F32); gm = m.cpu().numpy()
        out["cpu_reference"] = {"build_ms": round(tb * 1e3, 1), "Mrays_per_s": round(len(rays) / tt / 1e6, 3), "threads": os.cpu_count(),
                                "nodes": int(nodes.shape[0]), "depth": rst["max_tree_depth"]}
        # same node array (the reference's tree loaded with nrtSetTree): must agree exactly
        b = BVHAccel(np.float32); b.SetMesh(SphereGeometry(c, r)); b.SetTree(nodes, idx)
        bh, bm = b.TraverseBatch(rays)
        out["parity_on_reference_tree"] = {"mask_equal": bool(np.array_equal(bm, rm)), "t_bit_equal": bool(bh["t"].tobytes() == rh["t"].tobytes()),
                                           "prim_id_equal": bool(np.array_equal(bh["prim_id"], rh["prim_id"])),
                                           "max_abs_du_dv": float(max(np.abs(bh["u"] - rh["u"]).max(), np.abs(bh["v"] - rh["v"]).max()))}
        # different trees: this intersector's t comes from b*b - 4ac with |org - c| >> r (catastrophic cancellation), so a
        # computed t may fall outside its sphere's box interval and the nearest hit found depends on the tree
        out["gpu_tree_vs_reference_tree"] = {"mask_equal_fraction": float((gm == rm).mean()), "t_equal_fraction": float((gh["t"] == rh["t"]).mean())}
except Exception as e:  # pragma: no cover
    out["cpu_reference_error"] = repr(e)
print(json.dumps(out))

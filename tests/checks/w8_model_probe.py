"""CPU-side counts of the 8-wide compressed walk's MODEL (oracle/wide8_model.inc) against the two-level walk's model on a
saved GPU-built tree (tools/dump_tree.py): steps per ray, leaves, triangle tests — what the kernel variants can win before
any of them is built.   python tests/checks/w8_model_probe.py gpurun_out/c3_tree.npz C3 [every]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from nanort_amd import scenes  # noqa: E402
from oracle.bindings import Oracle  # noqa: E402

tree, cfg = sys.argv[1], sys.argv[2]
every = int(sys.argv[3]) if len(sys.argv) > 3 else 16
d = np.load(tree)
nodes, idx = d["nodes"], d["idx"]
if cfg == "C3":
    v, f = scenes.plane(1000, 500)
else:
    v, f = scenes.sphere()
orc = Oracle()
W, H = 1920, 1080
rays1_all = scenes.camera_rays(W, H)
sel = np.arange(0, rays1_all.shape[0], every)
# wave 2 from the oracle's wave-1 hits on the sample
h1, m1 = orc.traverse(nodes, idx, v, f, rays1_all[sel])
rays1 = rays1_all[sel]
rays2 = scenes.secondary_rays("bounce", v, f, rays1, h1, m1, pixel_base=0)
print("tree %d nodes; %d primary, %d bounce rays (every %dth)" % (nodes.shape[0], rays1.shape[0], rays2.shape[0], every))
for mode in (0,):
    t0 = time.time()
    w8 = orc.wide8_build(nodes, idx, v, f, collapse_mode=mode)
    wn, wr = w8.arrays()
    nch = np.array([bin(int(a) | int(b)).count("1") for a, b in zip(wn["imask"][:20000], wn["lmask"][:20000])])
    print("collapse mode %d: %d wide nodes (%.1f MB), %d leaf records (%.1f MB), mean children %.2f, build %.1fs" % (
        mode, w8.num_nodes, w8.num_nodes * 80 / 1e6, w8.num_recs, w8.num_recs * 40 / 1e6, nch.mean(), time.time() - t0))
    for name, rays in (("primary", rays1), ("bounce", rays2)):
        oh, om, oc = orc.traverse(nodes, idx, v, f, rays, count=True)
        _, _, c4, _, _ = orc.traverse_wide4_model(nodes, idx, v, f, rays)
        n = rays.shape[0]
        if mode == 0:
            print("  %-7s binary loop: %.1f nodes %.2f leaves %.2f tris | two-level: %.2f steps" % (
                name, oc[0] / n, oc[1] / n, oc[2] / n, c4[0] / n))
        for om_, cm_ in ((0, 0), (0, 1), (3, 1), (1, 1), (2, 2)):
            hh, mm, c = w8.traverse(rays, order_mode=om_, cull_mode=cm_)
            bad_t = int(((hh["t"] != oh["t"]) & ~(np.isnan(hh["t"]) & np.isnan(oh["t"]))).sum())
            ties = int((hh["prim_id"] != oh["prim_id"]).sum())
            print("    order %d cull %d: steps %.2f (empty %.2f) leaves %.2f (rejected %.2f) tris %.2f stack %d dropped %.2f | mask==%s t!=:%d prim!=:%d" % (
                om_, cm_, c[0] / n, c[1] / n, c[2] / n, c[3] / n, c[4] / n, c[5], c[6] / n, bool((mm == om).all()), bad_t, ties))

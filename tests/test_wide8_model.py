"""The CPU MODEL of the 8-wide compressed walk (oracle/wide8_model.inc) against the restated reference loop, on the CPU: the
layout is conservative (every quantised box contains its child's) and the walk returns the reference's hit flags and t bit for
bit, u / v / prim_id up to exact-t ties (each differing ray re-verified) — for every ordering the model implements, on hostile
rays.  The GPU suite then checks the device's layout and records against this model (tests/test_gpu_wide8.py)."""
import glob
import os

import numpy as np
import pytest

from helpers import assert_hits_match
from nanort_amd import scenes
from nanort_amd.wire import TRACE_OPTIONS
from test_gpu_wide4 import hostile_rays

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def mesh_of(name):
    if name == "c1":
        return scenes.load_c1_mesh()
    if name == "plane":
        return scenes.plane(60, 37)
    return scenes.sphere(48, 24)


def check_layout(nodes, mn, mr):
    """Quantised child boxes contain the exact ones; masks, bases and leaf blocks are consistent."""
    scale = np.ldexp(1.0, mn["e"].astype(np.int64) - 127)  # [n, 3]
    used_child, used_rec = np.zeros(mn.shape[0], bool), np.zeros(mr.shape[0], bool)
    used_child[0] = True
    for i in range(mn.shape[0]):
        w = mn[i]
        assert (int(w["imask"]) & int(w["lmask"])) == 0
        ri, rl = 0, 0
        for s in range(8):
            inner, leaf = (w["imask"] >> s) & 1, (w["lmask"] >> s) & 1
            if not (inner or leaf):
                continue
            lo = w["p"].astype(np.float64) + w["qlo"][:, s].astype(np.float64) * scale[i]
            hi = w["p"].astype(np.float64) + w["qhi"][:, s].astype(np.float64) * scale[i]
            if inner:
                c = int(w["child_base"]) + ri
                ri += 1
                assert not used_child[c]
                used_child[c] = True
                b = nodes[int(mn[c]["root"])]
                bmin, bmax = b["bmin"], b["bmax"]
            else:
                r = int(w["leaf_base"]) + rl * int(w["stride"])
                rl += 1
                rec = mr["w"][r]
                cnt = int(rec[6])
                assert 1 <= cnt < int(w["stride"]) + 0 and not used_rec[r:r + 1 + cnt].any()
                used_rec[r:r + 1 + cnt] = True
                bmin, bmax = rec[:3].view(np.float32), rec[3:6].view(np.float32)
            assert (lo <= bmin.astype(np.float64)).all() and (hi >= bmax.astype(np.float64)).all(), (i, s)
    assert used_child.all()


@pytest.mark.parametrize("mesh", ["c1", "plane", "sphere"])
def test_model_walk_equals_the_reference_up_to_ties(mesh, oracle):
    v, f = mesh_of(mesh)
    nodes, idx, _ = oracle.build(v, f)
    model = oracle.wide8_build(nodes, idx, v, f)
    mn, mr = model.arrays()
    check_layout(nodes, mn, mr)
    rays = np.concatenate([hostile_rays(v, 20000, seed=51), scenes.camera_rays(96, 64)])
    oh, om = oracle.traverse(nodes, idx, v, f, rays)
    for order, cull in ((1, 1), (0, 0), (0, 1), (3, 1), (2, 2)):
        h, m, c = model.traverse(rays, order_mode=order, cull_mode=cull)
        assert_hits_match(oh, om, h, m, oracle, nodes, idx, v, f, rays)
        assert c[7] == rays.shape[0]
    opts = np.zeros(1, dtype=TRACE_OPTIONS)
    opts["prim_ids_range"] = (50, 700)
    opts["skip_prim_id"] = 123
    opts["cull_back_face"] = 1
    oh, om = oracle.traverse(nodes, idx, v, f, rays, opts)
    h, m, _ = model.traverse(rays, opts, order_mode=1, cull_mode=1)
    assert_hits_match(oh, om, h, m, oracle, nodes, idx, v, f, rays, base_opts=opts[0])


@pytest.mark.parametrize("case", sorted(glob.glob(os.path.join(GOLDEN, "fuzz_case_*f32*.npz"))), ids=os.path.basename)
def test_model_on_saved_fuzz_cases(case, oracle):
    """Adversarial integer-grid meshes: the reference's own answer is tree-dependent there beyond exact ties, so
    the check is the cross-tree one: flags and t agree on all but a handful of rays."""
    d = np.load(case)
    v, f, rays, nodes, idx = d["v"], d["f"], d["rays"], d["nodes"], d["idx"]
    leaves = nodes[nodes["flag"] != 0]
    if nodes[0]["flag"] != 0 or leaves["data"][:, 0].max() > 30 or leaves["data"][:, 0].min() < 1:
        pytest.skip("the 8-wide layout is built for trees with a branch root and 1..30 primitives per leaf")
    model = oracle.wide8_build(nodes, idx, v, f)
    oh, om = oracle.traverse(nodes, idx, v, f, rays)
    h, m, _ = model.traverse(rays, order_mode=1, cull_mode=1)
    same = (m == om) & ((h["t"] == oh["t"]) | (np.isnan(h["t"]) & np.isnan(oh["t"])))
    assert same.mean() > 0.998, same.mean()

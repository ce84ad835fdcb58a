"""The 8-wide compressed walk (tunable `wide8`: wide8.hip builds the layout, k_traverse_w8 walks it).

Two things are checked on the GPU:
 * the LAYOUT the device builds equals the CPU model's (oracle/wide8_model.inc), record by record — the two differ by a
   renumbering only (positions are handed out by atomics), so records are matched through the binary branch they were made
   from;
 * the RECORDS: bit-identical to the model's walk over the same layout (same order, same arithmetic), and — the contract's
   bar — against the restated reference loop on the same node array: hit flags and t bit-equal, u / v /
   prim_id equal except at exact-t ties, every differing ray re-verified (helpers.assert_hits_match)."""
import numpy as np
import pytest

from helpers import assert_hits_identical, assert_hits_match
from nanort_amd import BVHAccel, TriangleMesh, scenes
from nanort_amd.wire import TRACE_OPTIONS
from oracle.bindings import W8_NODE
from test_gpu_wide4 import hostile_rays

pytestmark = pytest.mark.gpu


def mesh_of(name):
    if name == "c1":
        return scenes.load_c1_mesh()
    if name == "plane":
        return scenes.plane(120, 77)
    if name == "sphere":
        return scenes.sphere(64, 40)
    return scenes.plane(3, 2)


def compare_layouts(gn_raw, gr, mn, mr):
    """GPU arrays vs model arrays, matched through `root`."""
    gn = gn_raw.reshape(-1).view(W8_NODE)
    assert gn.shape[0] == mn.shape[0], (gn.shape, mn.shape)
    go, mo = np.argsort(gn["root"]), np.argsort(mn["root"])
    g, m = gn[go], mn[mo]
    assert np.array_equal(g["root"], m["root"]) and np.unique(g["root"]).shape[0] == g.shape[0]
    for k in ("p", "e", "imask", "lmask", "stride", "qlo", "qhi"):
        assert g[k].tobytes() == m[k].tobytes(), "field %s differs" % k
    # inner children: the same binary branches, in the same (slot) order
    for arr_g, arr_m in ((gn, mn),):
        pass
    nin = np.array([bin(int(x)).count("1") for x in g["imask"]])
    for r in range(8):
        sel = nin > r
        assert np.array_equal(gn["root"][g["child_base"][sel] + r], mn["root"][m["child_base"][sel] + r])
    # leaf blocks: box record + `count` triangle records of every leaf child
    nlf = np.array([bin(int(x)).count("1") for x in g["lmask"]])
    for r in range(8):
        sel = nlf > r
        gb = g["leaf_base"][sel].astype(np.int64) + r * g["stride"][sel].astype(np.int64)
        mb = m["leaf_base"][sel].astype(np.int64) + r * m["stride"][sel].astype(np.int64)
        assert np.array_equal(gr[gb][:, :7], mr["w"][mb][:, :7])
        cnt = gr[gb][:, 6].astype(np.int64)
        for j in range(int(cnt.max()) if cnt.size else 0):
            has = cnt > j
            assert np.array_equal(gr[gb[has] + 1 + j], mr["w"][mb[has] + 1 + j])


def w8_accel(v, f, tree=None):
    a = BVHAccel(np.float32)
    a.SetTunable("wide8", 1)
    if tree is None:
        assert a.Build(f.shape[0], TriangleMesh(v, f))
    else:
        a.SetMesh(TriangleMesh(v, f))
        a.SetTree(*tree)
    return a


@pytest.mark.parametrize("mesh", ["c1", "plane", "sphere", "tiny"])
def test_layout_and_records(mesh, oracle):
    v, f = mesh_of(mesh)
    a = w8_accel(v, f)
    nodes, idx = a.GetTree()
    model = oracle.wide8_build(nodes, idx, v, f)
    mn, mr = model.arrays()
    gn, gr = a.GetWide8()
    compare_layouts(gn, gr, mn, mr)
    rays = np.concatenate([hostile_rays(v, 40000, seed=31), scenes.camera_rays(160, 120)])
    h, m = a.TraverseBatch(rays)
    assert a.LastKernelName().startswith("nrt::k_traverse_w8<true"), a.LastKernelName()
    mh, mm, _ = model.traverse(rays, order_mode=1, cull_mode=1)
    assert_hits_identical(mh, mm, h, m)  # the model's walk: same order, same arithmetic
    oh, om = oracle.traverse(nodes, idx, v, f, rays)
    assert_hits_match(oh, om, h, m, oracle, nodes, idx, v, f, rays)
    # the default walk on the same context data: bit-identical to the restatement
    a.SetTunable("wide8", 0)
    h0, m0 = a.TraverseBatch(rays)
    assert_hits_identical(oh, om, h0, m0)


def test_rejecting_trace_options_and_occlusion(oracle):
    v, f = scenes.sphere(48, 32)
    a = w8_accel(v, f)
    nodes, idx = a.GetTree()
    rays = hostile_rays(v, 30000, seed=37)
    for lo, hi, skip, cull in ((100, 2000, 0xFFFFFFFF, 0), (0, 0x7FFFFFFF, 777, 1)):
        opts = np.zeros(1, dtype=TRACE_OPTIONS)
        opts["prim_ids_range"] = (lo, hi)
        opts["skip_prim_id"] = skip
        opts["cull_back_face"] = cull
        h, m = a.TraverseBatch(rays, opts)
        assert a.LastKernelName() == "nrt::k_traverse_w8<false, false>"
        oh, om = oracle.traverse(nodes, idx, v, f, rays, opts)
        assert_hits_match(oh, om, h, m, oracle, nodes, idx, v, f, rays, base_opts=opts[0])
    occ = a.OccludedBatch(rays)
    assert a.LastKernelName().startswith("nrt::k_traverse_w8")
    _, om = oracle.traverse(nodes, idx, v, f, rays)
    assert np.array_equal(occ, om)


def test_reference_built_deep_tree_spills(oracle):
    """The reference builder's tree over a plane is ~100 levels deep: the group stack outgrows its LDS part."""
    v, f = scenes.plane(150, 100)
    nodes, idx, _ = oracle.build(v, f)
    a = w8_accel(v, f, tree=(nodes, idx))
    model = oracle.wide8_build(nodes, idx, v, f)
    compare_layouts(*a.GetWide8(), *model.arrays())
    rays = hostile_rays(v, 50000, seed=41)
    h, m = a.TraverseBatch(rays)
    assert a.LastKernelName().startswith("nrt::k_traverse_w8")
    mh, mm, c = model.traverse(rays, order_mode=1, cull_mode=1)
    assert_hits_identical(mh, mm, h, m)
    oh, om = oracle.traverse(nodes, idx, v, f, rays)
    assert_hits_match(oh, om, h, m, oracle, nodes, idx, v, f, rays)


def test_trees_the_layout_is_not_built_for_keep_their_walk(oracle):
    v, f = scenes.plane(40, 30)
    nodes, idx, _ = oracle.build(v, f)
    nodes = nodes.copy()
    branch = np.nonzero(nodes["flag"] == 0)[0]
    k = int(nodes["data"][branch[1]][0])
    nodes["bmax"][k] += 5.0  # a child box sticking out of its parent's: the exact leaf test no longer implies the ancestors'
    a = w8_accel(v, f, tree=(nodes, idx))
    rays = hostile_rays(v, 20000, seed=43)
    h, m = a.TraverseBatch(rays)
    assert a.LastKernelName().startswith("nrt::k_traverse_wide<"), a.LastKernelName()
    oh, om = oracle.traverse(nodes, idx, v, f, rays)
    assert_hits_identical(oh, om, h, m)
    # fp64 trees and single-leaf trees too
    v1, f1 = scenes.plane(1, 1)
    a1 = w8_accel(v1, f1)
    h1, m1 = a1.TraverseBatch(scenes.camera_rays(32, 32))
    assert a1.LastKernelName().startswith("nrt::k_traverse_wide<")

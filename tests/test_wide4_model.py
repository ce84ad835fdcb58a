"""The traversal kernel's two-levels-per-step walk, checked on the CPU against the restated reference loop
(oracle/wide4_model_body.inc; long runs: tests/checks/fuzz_wide4_model.py): same records, same sequence of visited
leaves — and a tree that breaks the precondition (a child box that does not contain its children's) to show why the
library walks such trees one level per step."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "checks"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def test_two_level_walk_visits_the_reference_loops_leaves_in_order():
    from fuzz_wide4_model import one_round
    from oracle.bindings import Oracle

    rng = np.random.default_rng(11)
    orc = Oracle()
    stats = {"rays": 0, "leaves": 0, "steps": 0, "nodes_ref": 0, "max_stack": 0}
    t_end = time.time() + 6.0
    rounds = 0
    while time.time() < t_end or rounds < 12:
        ok, _ = one_round(rng, orc, stats)
        assert ok, "records or leaf sequence differ from the reference loop (round %d, %s)" % (rounds, stats)
        rounds += 1
    assert stats["leaves"] > 10000
    assert stats["steps"] < stats["nodes_ref"]  # fewer dependent fetches than nodes popped by the binary loop


def test_c1_camera_wave_same_trail():
    from nanort_amd import scenes
    from oracle.bindings import Oracle

    orc = Oracle()
    v, f = scenes.load_c1_mesh()
    rays = scenes.camera_rays(160, 120)
    nodes, idx, _ = orc.build(v, f)
    oh, om, oc = orc.traverse(nodes, idx, v, f, rays, count=True)
    wh, wm, wc, t_ref, t_w4 = orc.traverse_wide4_model(nodes, idx, v, f, rays, trail_cap=int(oc[1]) + 1)
    assert np.array_equal(om, wm) and oh.tobytes() == wh.tobytes()
    assert np.array_equal(t_ref, t_w4) and int(wc[2]) == int(oc[2])
    # deepest stack: three pending entries per two levels at most (what api.hip sizes the overflow area for)
    depth = max_depth(nodes)
    assert int(wc[3]) <= 3 * (depth // 2 + 1)


def max_depth(nodes):
    depth, stack = 0, [(0, 0)]
    while stack:
        i, d = stack.pop()
        depth = max(depth, d)
        if nodes["flag"][i] == 0:
            stack.append((int(nodes["data"][i][0]), d + 1))
            stack.append((int(nodes["data"][i][1]), d + 1))
    return depth


def test_a_child_box_that_does_not_contain_its_children_breaks_the_equivalence():
    """Shrink the box of a branch that has branch children: the reference loop now culls rays at that box which the
    two-level step (which never tests it) lets through to the grandchildren — the leaf sequences differ.  This is the
    precondition nrtSetTree checks (`tree_nested`); trees that fail it are walked one level per step."""
    from nanort_amd import scenes
    from oracle.bindings import Oracle

    orc = Oracle()
    v, f = scenes.plane(40, 30)
    nodes, idx, _ = orc.build(v, f)
    nodes = nodes.copy()
    # a depth-1 branch whose children are branches
    root = nodes[0]
    k = int(root["data"][0])
    assert nodes["flag"][k] == 0
    c = (nodes["bmin"][k] + nodes["bmax"][k]) / 2
    nodes["bmin"][k] = c - 1e-3
    nodes["bmax"][k] = c + 1e-3
    rays = scenes.camera_rays(96, 64)
    oh, om, oc = orc.traverse(nodes, idx, v, f, rays, count=True)
    wh, wm, wc, t_ref, t_w4 = orc.traverse_wide4_model(nodes, idx, v, f, rays, trail_cap=4 * int(oc[1]) + 100000)
    assert len(t_w4) > len(t_ref)  # the model visits leaves the reference loop never reaches

"""Structural validator for a reference-format BVH (SURVEY.md §8a N1-N3, §7 step 4 gate (i)).

Pure numpy; used by the GPU build tests on trees copied back through the C ABI
and, on CPU, on the oracle's own trees (so the checker itself is tested).
"""
import numpy as np


def prim_bounds(verts, faces):
    p = verts[faces]  # (n, 3, 3)
    return p.min(axis=1), p.max(axis=1)


def validate_bvh(nodes, indices, verts, faces, min_leaf=4, max_depth=256, stats=None, exact_bounds=True, low_side_first=False):
    """Raises AssertionError on the first violated invariant; returns a dict of tree metrics."""
    n = faces.shape[0]
    nn = nodes.shape[0]
    assert nn >= 1
    assert indices.shape[0] == n
    assert np.array_equal(np.sort(indices), np.arange(n, dtype=indices.dtype)), "indices is not a permutation"
    flag = nodes["flag"]
    assert np.isin(flag, (0, 1)).all(), "flag must be 0 or 1"
    leaf = flag == 1
    branch = ~leaf
    nl, nb = int(leaf.sum()), int(branch.sum())
    assert nl == nb + 1, "binary tree: leaves == branches + 1"
    assert np.isin(nodes["axis"][branch], (0, 1, 2)).all(), "branch axis out of range"
    # DFS pre-order: left child == parent + 1, right child > left
    bi = np.nonzero(branch)[0]
    left = nodes["data"][bi, 0].astype(np.int64)
    right = nodes["data"][bi, 1].astype(np.int64)
    assert (left == bi + 1).all(), "left child must follow its parent (pre-order)"
    assert ((right > left) & (right < nn)).all(), "right child index out of order/range"
    # every node except the root is the child of exactly one branch
    ref = np.zeros(nn, dtype=np.int64)
    np.add.at(ref, left, 1)
    np.add.at(ref, right, 1)
    assert ref[0] == 0 and (ref[1:] == 1).all(), "node reachable zero or several times"
    # depth by propagation in index order (parents precede children in pre-order)
    depth = np.zeros(nn, dtype=np.int64)
    order_ok = True
    # vectorised level propagation
    frontier = np.array([0], dtype=np.int64)
    d = 0
    while frontier.size:
        depth[frontier] = d
        fb = frontier[branch[frontier]]
        frontier = np.concatenate([nodes["data"][fb, 0], nodes["data"][fb, 1]]).astype(np.int64)
        d += 1
        assert d <= nn + 1
    max_d = int(depth.max())
    assert max_d <= max_depth, "tree deeper than max_tree_depth"
    # leaves tile [0, n) and respect the leaf rule
    li = np.nonzero(leaf)[0]
    cnt = nodes["data"][li, 0].astype(np.int64)
    first = nodes["data"][li, 1].astype(np.int64)
    o = np.argsort(first, kind="stable")
    assert first[o][0] == 0 and (first[o][1:] == (first[o] + cnt[o])[:-1]).all() and (first[o] + cnt[o])[-1] == n, \
        "leaf ranges do not tile the index array"
    assert (cnt >= 1).all(), "empty leaf"
    big = cnt > max(min_leaf, 1)
    assert (depth[li][big] >= max_depth).all(), "leaf larger than min_leaf_primitives below max depth"
    # in pre-order, leaves appear in increasing `first`
    assert (np.diff(first) > 0).all(), "leaf order must follow the index array in pre-order"
    # bounds: leaf box == exact union of its primitives' boxes; branch box == union of children
    pmin, pmax = prim_bounds(verts, faces)
    starts = first[o]
    lmin = np.minimum.reduceat(pmin[indices], starts, axis=0)
    lmax = np.maximum.reduceat(pmax[indices], starts, axis=0)
    if exact_bounds:
        assert np.array_equal(lmin, nodes["bmin"][li][o]) and np.array_equal(lmax, nodes["bmax"][li][o]), \
            "leaf bounds are not the exact union of its primitives"
    else:
        assert (nodes["bmin"][li][o] <= lmin).all() and (nodes["bmax"][li][o] >= lmax).all()
    # bottom-up union by decreasing depth
    umin = nodes["bmin"].copy()
    umax = nodes["bmax"].copy()
    cmin = np.minimum(nodes["bmin"][left], nodes["bmin"][right])
    cmax = np.maximum(nodes["bmax"][left], nodes["bmax"][right])
    if exact_bounds:
        assert np.array_equal(cmin, nodes["bmin"][bi]) and np.array_equal(cmax, nodes["bmax"][bi]), \
            "branch bounds are not the union of its children"
    else:
        assert (nodes["bmin"][bi] <= cmin).all() and (nodes["bmax"][bi] >= cmax).all()
    del umin, umax
    # N1: data[0] is the LOW side of the split along `axis`, data[1] the high side (nanort.h:1841-1868: the partition
    # puts the primitives below the cut first; the traversal's near/far choice, :2538, relies on it).  Checked on the
    # centroids: no primitive of the left child lies beyond every primitive of the right child's low end (equality when the
    # centroids coincide and the split fell back to the middle of the range).  `low_side_first`: asserted for the GPU
    # builder's trees; the reference's own trees break it wherever its X-only binning (nanort.h:1357) left it with the
    # middle-of-the-range fallback (:1851-1856), whose halves are in index order, not in spatial order.
    sub_first = np.zeros(nn, dtype=np.int64)
    sub_cnt = np.zeros(nn, dtype=np.int64)
    sub_first[li] = first
    sub_cnt[li] = cnt
    for i in bi[::-1]:  # children follow their parent in pre-order: fill bottom-up
        l, r = int(nodes["data"][i, 0]), int(nodes["data"][i, 1])
        sub_first[i] = sub_first[l]
        sub_cnt[i] = sub_cnt[l] + sub_cnt[r]
        assert sub_first[r] == sub_first[l] + sub_cnt[l], "right child's primitives must follow the left child's"
    if low_side_first:  # the converse of the leaf rule (N3): nothing that should have been a leaf was split
        assert (sub_cnt[bi] > max(min_leaf, 1)).all(), "a range of at most min_leaf_primitives was split"
    cen = verts[faces].astype(np.float64).sum(axis=1)[indices]  # 3 x centroid, in index-array order
    scale = np.maximum(1.0, np.abs(cen[np.isfinite(cen)]).max()) if np.isfinite(cen).any() else 1.0
    bdepth = depth[bi]
    for d_ in (np.unique(bdepth) if low_side_first else ()):
        sel = bi[bdepth == d_]
        sel = sel[np.argsort(sub_first[sel], kind="stable")]
        l = nodes["data"][sel, 0].astype(np.int64)
        r = nodes["data"][sel, 1].astype(np.int64)
        ax = nodes["axis"][sel].astype(np.int64)
        starts = np.stack([sub_first[l], sub_first[r], sub_first[r] + sub_cnt[r]], axis=1).reshape(-1)
        pad = starts[-1] >= n
        st = starts[:-1] if pad else starts
        for a_ in range(3):
            m = ax == a_
            if not m.any():
                continue
            hi = np.maximum.reduceat(cen[:, a_], st)
            lo = np.minimum.reduceat(cen[:, a_], st)
            left_max = hi[0::3][: m.shape[0]][m]
            right_min = lo[1::3][: m.shape[0]][m]
            ok = (left_max <= right_min + 1e-5 * scale) | ~np.isfinite(left_max) | ~np.isfinite(right_min)
            assert ok.all(), "data[0] must hold the low side of the split along `axis`"
    # stack need of the reference traversal (512 entries, nanort.h:2497)
    assert max_d + 2 <= 512
    if stats is not None:
        assert int(stats["num_leaf_nodes"]) == nl and int(stats["num_branch_nodes"]) == nb, (stats, nl, nb)
        assert int(stats["max_tree_depth"]) == max_d, (stats, max_d)
    return {"num_nodes": nn, "num_leaves": nl, "num_branches": nb, "max_depth": max_d,
            "sah_cost": sah_cost(nodes), "mean_leaf_size": float(cnt.mean())}


def sah_cost(nodes, c_node=1.0, c_tri=1.0):
    """Expected traversal cost sum_nodes SA(node)/SA(root) * (c_node | c_tri * count)."""
    ext = (nodes["bmax"] - nodes["bmin"]).astype(np.float64)
    sa = ext[:, 0] * ext[:, 1] + ext[:, 1] * ext[:, 2] + ext[:, 2] * ext[:, 0]
    root = sa[0] if sa[0] > 0 else 1.0
    leaf = nodes["flag"] == 1
    cost = c_node * sa[~leaf].sum() + c_tri * (sa[leaf] * nodes["data"][leaf, 0]).sum()
    return float(cost / root)

"""The structural validator accepts the reference's own trees and rejects broken ones."""
import numpy as np
import pytest

from bvh_check import sah_cost, validate_bvh
from nanort_amd import scenes


def test_validator_accepts_reference_trees(oracle, c1_mesh):
    v, f = c1_mesh
    nodes, idx, st = oracle.build(v, f)
    m = validate_bvh(nodes, idx, v, f, stats=st)
    assert m["num_nodes"] == 713 and m["max_depth"] == 20
    pv, pf = scenes.plane(40, 20)
    nodes, idx, st = oracle.build(pv, pf, min_leaf=2, bin_size=8)
    validate_bvh(nodes, idx, pv, pf, min_leaf=2, stats=st)
    assert sah_cost(nodes) > 0


def test_validator_rejects_corruption(oracle, c1_mesh):
    v, f = c1_mesh
    nodes, idx, st = oracle.build(v, f)
    bad = nodes.copy()
    bad["bmax"][5, 0] -= 1.0
    with pytest.raises(AssertionError):
        validate_bvh(bad, idx, v, f)
    bad_idx = idx.copy()
    bad_idx[3] = bad_idx[4]
    with pytest.raises(AssertionError):
        validate_bvh(nodes, bad_idx, v, f)
    bad = nodes.copy()
    b = np.nonzero(bad["flag"] == 0)[0][3]
    bad["data"][b, 1] = bad["data"][b, 0]
    with pytest.raises(AssertionError):
        validate_bvh(bad, idx, v, f)

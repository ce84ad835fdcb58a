"""CPU proof-by-test for the next instruction-count lever of the GPU walk: inner boxes tested in the fused
form fma(plane, inv, -org * inv) with a slack that makes the test conservative, leaves re-tested with the reference's own
IntersectRayAABB before their triangles.  The model (oracle/fused_slab_model.inc) walks the reference-format tree in the
reference's order; on nested trees every hit record must equal the restatement's bit for bit, and the number of leaves and
triangles tested must be the restatement's — only inner nodes may be entered in addition."""
import numpy as np
import pytest

from nanort_amd import scenes


def rays_around(rng, n, lo, hi, bounded=False):
    from oracle import bindings as ob

    rays = np.zeros(n, dtype=ob.ray_dtype(np.float32))
    size = hi - lo
    org = rng.uniform(lo - 0.6 * size, hi + 0.6 * size, size=(n, 3))
    tgt = rng.uniform(lo, hi, size=(n, 3))
    d = tgt - org
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays["org"] = org.astype(np.float32)
    rays["dir"] = (d * rng.choice([1.0, 1.0, 0.01, 100.0], size=(n, 1))).astype(np.float32)
    rays["max_t"] = 3.0e38
    if bounded:
        rays["min_t"] = rng.choice([0.0, 0.3, 2.0], size=n).astype(np.float32)
        rays["max_t"] = rng.choice([1.5, 8.0, 3.0e38], size=n).astype(np.float32)
    # on-plane origins and axis-parallel directions: the fused form is not used for these rays (the model checks and falls back)
    rays["dir"][: n // 20, 0] = 0.0
    rays["dir"][n // 20 : n // 10, 1] = -0.0
    rays["org"][: n // 10] = np.round(rays["org"][: n // 10])
    return rays


@pytest.mark.parametrize("mesh", ["c1", "sphere", "plane", "soup", "far_from_origin", "tiny"])
@pytest.mark.parametrize("slack", [2, 8])
def test_fused_inner_tests_and_exact_leaves_give_the_reference_records(oracle, c1_mesh, mesh, slack):
    rng = np.random.default_rng(41)
    if mesh == "c1":
        v, f = c1_mesh
    elif mesh == "sphere":
        v, f = scenes.sphere(40, 20)
    elif mesh == "plane":
        v, f = scenes.plane(60, 40)
    elif mesh == "soup":
        c = rng.uniform(-1, 1, (3000, 1, 3))
        v = (c + rng.normal(0, 0.05, (3000, 3, 3))).reshape(-1, 3).astype(np.float32)
        f = np.arange(9000, dtype=np.uint32).reshape(3000, 3)
    elif mesh == "far_from_origin":  # |org * inv| is large against the distances: where the two roundings differ most
        v, f = scenes.sphere(30, 15)
        v = (v + np.array([4000.0, -2500.0, 7000.0], dtype=np.float32)).astype(np.float32)
    else:
        v, f = scenes.sphere(20, 10)
        v = (v * 1e-4).astype(np.float32)
    v = np.ascontiguousarray(v, dtype=np.float32)
    nodes, idx, _ = oracle.build(v, f)
    lo, hi = v.min(axis=0).astype(np.float64), v.max(axis=0).astype(np.float64)
    rays = np.concatenate([rays_around(rng, 6000, lo, hi), rays_around(rng, 3000, lo, hi, bounded=True)])
    oh, om, oc = oracle.traverse(nodes, idx, v, f, rays, count=True)
    h, m, c = oracle.traverse_fused_slab_model(nodes, idx, v, f, rays, slack_ulps=slack)
    assert np.array_equal(m, om)
    assert h.tobytes() == oh.tobytes()
    assert int(c[1]) == int(oc[1]) and int(c[2]) == int(oc[2])  # the same leaves, the same triangles
    extra = int(c[0]) - int(oc[0])
    assert extra >= 0 and extra == 2 * int(c[3])  # each inner node entered in addition pops its two children
    assert int(c[3]) <= 0.01 * int(oc[0])  # the price of the slack: well under 1 % more nodes

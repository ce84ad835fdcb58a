"""nrtTraverseBatchMulti: one host batch split row-interleaved over several contexts, each holding a replica of the tree — the
product's multi-GPU entry point, exercised on the one-GPU box with several contexts on device 0 (the split,
the strided copies into the caller's arrays, one host thread per context and the error paths are the same code)."""
import ctypes

import numpy as np
import pytest

from helpers import assert_hits_identical
from nanort_amd import BVHAccel, TriangleMesh, capi, scenes
from nanort_amd.wire import hit_dtype, widen_rays

pytestmark = pytest.mark.gpu


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def multi(ctxs, rays, row_len, real, with_mask=True):
    L = capi.lib()
    arr = (ctypes.c_void_p * len(ctxs))(*[a._h for a in ctxs])
    hits = np.zeros(rays.shape[0], dtype=hit_dtype(real))
    mask = np.full(rays.shape[0], 0xCD, dtype=np.uint8) if with_mask else None
    f = getattr(L, "nrtTraverseBatchMulti_" + ("f32" if real == np.float32 else "f64"))
    st = f(arr, len(ctxs), _p(rays), rays.shape[0], row_len, None, _p(hits), _p(mask))
    return st, hits, mask


@pytest.mark.parametrize("real", [np.float32, np.float64])
def test_batch_split_over_replicas_equals_one_context(real):
    v, f = scenes.sphere(96, 48)
    v = v.astype(real)
    W, H = 333, 71  # a ragged frame: the last row of the batch is shorter than row_len in the second case below
    rays = scenes.camera_rays(W, H)
    if real != np.float32:
        rays = widen_rays(rays)
    accs = []
    for _ in range(3):
        a = BVHAccel(real)
        assert a.Build(f.shape[0], TriangleMesh(v, f))
        accs.append(a)
    want_h, want_m = accs[0].TraverseBatch(rays)
    assert capi.lib().nrtDeviceCount() >= 1
    for row_len in (W, 1000, 0, 7 * W * H):  # image rows; rows that do not divide the batch; the default; one row for everything
        for n_ctx in (1, 2, 3):
            st, h, m = multi(accs[:n_ctx], rays, row_len, real)
            assert st == capi.NRT_OK, accs[0]._L.nrtLastError(accs[0]._h)
            assert_hits_identical(want_h, want_m, h, m)
    st, h, m = multi(accs, rays, W, real, with_mask=False)
    assert st == capi.NRT_OK and all(h[k].tobytes() == want_h[k].tobytes() for k in ("t", "u", "v", "prim_id"))
    # a deterministic build: the replicas hold the same bits
    t0, t1 = accs[0].GetTree(), accs[2].GetTree()
    assert t0[0].tobytes() == t1[0].tobytes() and t0[1].tobytes() == t1[1].tobytes()


def test_mismatched_contexts_are_refused():
    v, f = scenes.sphere(32, 16)
    a, b = BVHAccel(np.float32), BVHAccel(np.float32)
    assert a.Build(f.shape[0], TriangleMesh(v, f))
    v2, f2 = scenes.plane(20, 10)
    assert b.Build(f2.shape[0], TriangleMesh(v2, f2))
    rays = scenes.camera_rays(64, 64)
    st, _, _ = multi([a, b], rays, 64, np.float32)
    assert st == capi.NRT_ERR_INVALID and b"another tree" in a._L.nrtLastError(a._h)
    st, _, _ = multi([a, a], rays, 64, np.float32)
    assert st == capi.NRT_ERR_INVALID and b"twice" in a._L.nrtLastError(a._h)
    c = BVHAccel(np.float64)
    assert c.Build(f.shape[0], TriangleMesh(v.astype(np.float64), f))
    st, _, _ = multi([a, c], rays, 64, np.float32)
    assert st == capi.NRT_ERR_INVALID

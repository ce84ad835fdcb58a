"""CPU experiment behind DESIGN.md §10 (traversal, next lever): a CONSERVATIVE box test in fused form.

The reference's IntersectRayAABB (nanort.h:2285-2325) computes (plane - org) * inv per plane — a subtraction and a multiplication,
and a third multiplication (x 1.00000024) on the far side: 60 operations for the four boxes of a Wide4Node.  With b = -org * inv
per axis (once per ray) the same quantity is fma(plane, inv, b): 24 operations.  It is not the same ROUNDING, so it cannot decide
what the reference decides — but with a slack of k ulps of |org * inv| folded into two constants per axis (b_near = b - d,
b_far = b + d) and a slightly larger far-side factor it is CONSERVATIVE: it accepts every box the reference's form accepts.
Inner nodes may be tested conservatively if leaves are re-tested exactly (boxes are nested, the exact arithmetic is monotone: a
leaf whose exact test passes has only ancestors whose exact tests pass) — the scheme the 8-wide layout already uses.

    python tools/fma_slab_check.py        (numpy float32; the fma is emulated in float64, then rounded)
prints, per coordinate scale and slack k: boxes the exact form accepts, how many of them the fused form misses (must be 0), and how
many it accepts in addition (the price: extra steps)."""
import numpy as np

rng = np.random.default_rng(1)
f32 = np.float32


def exact(lo, hi, org, inv, tmin0, tmax0):
    t0 = (lo - org) * inv
    t1 = ((hi - org) * inv) * f32(1.00000024)
    return np.maximum(np.max(t0, axis=-1), tmin0) <= np.minimum(np.min(t1, axis=-1), tmax0)


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def fused(lo, hi, org, inv, tmin0, tmax0, k):
    b = -(org * inv)
    d = np.abs(b) * f32(k * 2.0 ** -23)
    t0 = fma(lo, inv, b - d)
    t1 = fma(hi, inv, b + d) * f32(1.0000005)
    return np.maximum(np.max(t0, axis=-1), tmin0) <= np.minimum(np.min(t1, axis=-1), tmax0)


N = 4_000_000
for scale, name in ((1.0, "unit scene"), (1000.0, "coordinates ~1000"), (1e-3, "coordinates ~1e-3")):
    org = (rng.uniform(-10, 10, (N, 3)) * scale).astype(f32)
    d = rng.normal(size=(N, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[: N // 10, 0] *= 1e-6  # nearly axis-parallel
    inv = (1.0 / d.astype(f32)).astype(f32)
    neg = d < 0
    c = (org + (rng.uniform(0, 1, (N, 1)) ** 2 * 30 * scale) * d + rng.normal(size=(N, 3)) * scale * rng.choice([1e-3, 0.1, 1.0], size=(N, 1))).astype(f32)
    h = (np.abs(rng.normal(size=(N, 3))) * scale * rng.choice([1e-4, 1e-2, 0.5], size=(N, 1))).astype(f32)
    lo = np.where(neg, c + h, c - h)
    hi = np.where(neg, c - h, c + h)
    e = exact(lo, hi, org, inv, f32(0), f32(3e38))
    for k in (2, 4, 8):
        m = fused(lo, hi, org, inv, f32(0), f32(3e38), k)
        print("%-18s slack %d ulp: exact accepts %d, missed by the fused form %d, extra accepts %d" % (name, k, int(e.sum()), int((e & ~m).sum()), int((m & ~e).sum())))

"""Deterministic workload of the sphere-primitive tests: the particle example's scene and
camera plus rays the camera never produces (origins inside the cloud, unnormalised and axis-parallel directions,
short max_t, rays starting inside a sphere)."""
import numpy as np

from nanort_amd import scenes
from nanort_amd.wire import RAY_F32

N_SPHERES = 5000
N_CYLINDERS = 4000
CAM_W, CAM_H = 160, 161


def scene():
    return scenes.random_spheres(N_SPHERES)


def rays():
    cam = scenes.particle_camera_rays(CAM_W, CAM_H)
    rng = np.random.default_rng(1234)
    n = 6000
    extra = np.zeros(n, dtype=RAY_F32)
    extra["org"] = rng.uniform(-1.2, 1.2, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[:1000] *= rng.uniform(0.01, 50.0, size=(1000, 1)).astype(np.float32)  # unnormalised
    d[1000:1300, 0] = 0.0  # axis-parallel components (vsafe_inverse path)
    d[1300:1600, 1] = 0.0
    d[1600:1700, :2] = 0.0
    extra["dir"] = d
    extra["min_t"] = 0.0
    extra["max_t"] = 1.0e30
    extra["max_t"][2000:2500] = rng.uniform(0.05, 1.0, size=500).astype(np.float32)
    extra["min_t"][2500:3000] = rng.uniform(0.0, 0.5, size=500).astype(np.float32)
    c, r = scene()
    k = rng.integers(0, N_SPHERES, size=500)
    extra["org"][3000:3500] = c[k] + (rng.uniform(-0.5, 0.5, size=(500, 3)) * r[k, None]).astype(np.float32)  # inside a sphere
    return np.concatenate([cam, extra])


def hostile_rays():
    """A thinned copy of rays() with zero, NaN and infinite components and negative max_t mixed in."""
    r = rays()[::5].copy()
    r["dir"][:50] = 0
    r["dir"][50:80, 0] = np.nan
    r["org"][80:100, 1] = np.inf
    r["max_t"][100:120] = -1
    return r


def degenerate_spheres(n=400):
    rng = np.random.default_rng(5)
    c = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    r = rng.uniform(0.01, 0.2, n).astype(np.float32)
    r[:20] = 0          # points
    r[20:30] = -0.1     # negative radii: inverted boxes
    c[30:40] = c[40:50]  # coincident centres
    return c, r


def degenerate_cylinders(n=400):
    rng = np.random.default_rng(6)
    v = rng.uniform(-1, 1, (n, 2, 3)).astype(np.float32)
    r = rng.uniform(0.01, 0.1, (n, 2)).astype(np.float32)
    v[:20, 1] = v[:20, 0]  # zero length
    r[20:30] = 0           # zero radius
    r[30:40, 0] = 0.2      # unequal radii (the intersector uses the larger)
    return v, r

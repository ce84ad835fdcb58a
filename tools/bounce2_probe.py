"""The SECOND bounce wave of a config (rays generated from the bounce-1 hits): how much of its lower rate is the small batch,
how much is lost coherence, and what would a coherence pre-pass buy at most (physical re-ordering on the host: no sort time)?
    python tools/bounce2_probe.py C3"""
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import torch  # noqa: E402,F401

import bench  # noqa: E402
from nanort_amd import scenes  # noqa: E402
from reorder_probe import cell_key, combine, octa_bin, octant, timeit  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
wl = bench.Workload(name, builds=1)
a = wl.accel
_, _, gh2, gm2 = wl.results()
a.TraverseBatchDevice(wl.d_rays2, wl.d_hits2, wl.d_mask2)
_, _, gh2, gm2 = wl.results()
rays3 = scenes.secondary_rays("bounce", wl.verts32, wl.faces, wl.rays2, gh2, gm2, pixel_base=7 * wl.n1)
n3 = rays3.shape[0]
nodes = a.GetNodes()
lo, hi = nodes[0]["bmin"].astype(np.float64), nodes[0]["bmax"].astype(np.float64)
ms2, _ = timeit(a, wl.rays2)
print("%s bounce-1 (%d rays) %.4f ms = %.0f Mrays/s" % (name, wl.n2, ms2, wl.n2 / ms2 / 1e3))
sub = wl.rays2[:: max(1, wl.n2 // n3)][:n3]
ms2s, _ = timeit(a, sub)
print("   bounce-1, every %d-th ray (%d rays): %.4f ms = %.0f Mrays/s   <- the small batch alone" % (max(1, wl.n2 // n3), sub.shape[0], ms2s, sub.shape[0] / ms2s / 1e3))
ms2h, _ = timeit(a, wl.rays2[:n3])
print("   bounce-1, the first %d rays: %.4f ms = %.0f Mrays/s" % (n3, ms2h, n3 / ms2h / 1e3))
base, out0 = timeit(a, rays3)
base_hits = out0.cpu().numpy().reshape(n3, -1)
print("bounce-2 (%d rays) as given %.4f ms = %.0f Mrays/s" % (n3, base, n3 / base / 1e3), flush=True)
org, dr = rays3["org"].astype(np.float64), rays3["dir"]
oc = octant(dr)
keys = {"random permutation": np.random.default_rng(1).permutation(n3).astype(np.uint64), "octant (stable)": oc[0]}
for b in (4, 5, 6, 8):
    keys["cell%d|octant" % b] = combine(cell_key(org, lo, hi, b), oc)
keys["cell10 only"] = cell_key(org, lo, hi, 10)[0]
keys["cell5|octa4x4"] = combine(cell_key(org, lo, hi, 5), octa_bin(dr, 4))
keys["octant|cell6"] = combine(oc, cell_key(org, lo, hi, 6))
for label, key in keys.items():
    perm = np.argsort(key, kind="stable")
    ms, out = timeit(a, rays3[perm])
    same = np.array_equal(out.cpu().numpy().reshape(n3, -1), base_hits[perm])
    print("   %-24s %.4f ms  x%.3f  records %s" % (label, ms, base / ms, "identical" if same else "DIFFER"), flush=True)

"""What would a coherence pre-pass of a ray batch buy?  The rays of a wave are PHYSICALLY re-ordered on the host by a
family of sort keys (origin cell in the root box, direction octant / octahedral bin, pixel tile, local blocks) and the
traversal kernel is timed on each order — an upper bound for an in-library sort (no gather, no sort time).  Also: the
WideNode / Wide4Node arrays in a pseudo-random order (tunable wide_scramble) as a bound on what any re-ordering of the
private node layout could change, and the work counters of every config.

    python tools/reorder_probe.py [C3 C4tile C2 C5] [--quick]
"""
import os
import sys

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402

import bench  # noqa: E402


def part1by2(x):
    x = x.astype(np.uint64) & 0x3FF
    x = (x | (x << 16)) & 0x30000FF
    x = (x | (x << 8)) & 0x300F00F
    x = (x | (x << 4)) & 0x30C30C3
    x = (x | (x << 2)) & 0x9249249
    return x


def part1by1(x):
    x = x.astype(np.uint64) & 0xFFFF
    x = (x | (x << 8)) & 0x00FF00FF
    x = (x | (x << 4)) & 0x0F0F0F0F
    x = (x | (x << 2)) & 0x33333333
    x = (x | (x << 1)) & 0x55555555
    return x


def cell_key(org, lo, hi, bits):
    ext = np.maximum(hi - lo, 1e-30)
    q = np.clip(((org - lo) / ext * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    return part1by2(q[:, 0]) | (part1by2(q[:, 1]) << np.uint64(1)) | (part1by2(q[:, 2]) << np.uint64(2)), 3 * bits


def octant(d):
    return ((d[:, 0] < 0).astype(np.uint64) | ((d[:, 1] < 0).astype(np.uint64) << np.uint64(1)) | ((d[:, 2] < 0).astype(np.uint64) << np.uint64(2))), 3


def octa_bin(d, n):
    """Octahedral map of the direction to an n x n grid, Morton-ordered."""
    d = d.astype(np.float64)
    s = np.abs(d).sum(axis=1)
    s[s == 0] = 1
    p = d / s[:, None]
    u, v = p[:, 0].copy(), p[:, 1].copy()
    neg = p[:, 2] < 0
    uu = (1 - np.abs(v)) * np.where(u >= 0, 1, -1)
    vv = (1 - np.abs(u)) * np.where(v >= 0, 1, -1)
    u[neg], v[neg] = uu[neg], vv[neg]
    iu = np.clip(((u * 0.5 + 0.5) * n).astype(np.int64), 0, n - 1)
    iv = np.clip(((v * 0.5 + 0.5) * n).astype(np.int64), 0, n - 1)
    b = int(np.log2(n))
    return part1by1(iu) | (part1by1(iv) << np.uint64(1)), 2 * b


def combine(*parts):
    """parts = (key, bits) most significant first."""
    k = np.zeros(parts[0][0].shape, np.uint64)
    for key, bits in parts:
        k = (k << np.uint64(bits)) | key
    return k


def timeit(accel, rays, reps=7):
    d = torch.from_numpy(np.ascontiguousarray(rays).view(np.uint8)).cuda()
    o = torch.empty(len(rays) * (16 if rays.dtype.itemsize == 36 else 32), dtype=torch.uint8, device="cuda")
    ts = []
    for _ in range(reps):
        accel.TraverseBatchDevice(d, o)
        ts.append(accel.LastTraverseMs())
    return float(np.median(ts)), o


def main():
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or ["C3"]
    quick = "--quick" in sys.argv
    for name in names:
        wl = bench.Workload(name, builds=2)
        a = wl.accel
        nodes = a.GetNodes()
        lo, hi = nodes[0]["bmin"].astype(np.float64), nodes[0]["bmax"].astype(np.float64)
        c1, c2 = wl.counters()
        print("== %s: %s" % (name, wl.describe()))
        print("   tree: %d nodes depth %d | primary %.2f nodes %.2f tris per ray | bounce %.2f nodes %.2f tris per ray | hit rate primary %.3f"
              % (wl.num_nodes, int(wl.stats["max_tree_depth"]), c1["nodes_visited"] / wl.n1, c1["tris_tested"] / wl.n1,
                 c2["nodes_visited"] / max(1, wl.n2), c2["tris_tested"] / max(1, wl.n2), float(wl.mask1.mean())), flush=True)
        for wave, rays in (("bounce", wl.rays2), ("primary", wl.rays1)):
            n = len(rays)
            org, dr = rays["org"].astype(np.float64), rays["dir"]
            base_ms, base_out = timeit(a, rays)
            base_hits = base_out.cpu().numpy()
            print("%-8s %-34s %.4f ms  (%.0f Mrays/s)" % (wave, "as given", base_ms, n / base_ms / 1e3), flush=True)
            keys = {}
            oc = octant(dr)
            keys["random permutation"] = np.random.default_rng(1).permutation(n).astype(np.uint64)
            keys["octant (stable)"] = oc[0]
            for b in ((4, 5, 6, 8) if not quick else (5,)):
                keys["cell%d|octant" % b] = combine(cell_key(org, lo, hi, b), oc)
            keys["cell10 only"] = cell_key(org, lo, hi, 10)[0]
            if not quick:
                for b, nb in ((4, 4), (5, 4), (5, 8), (6, 4), (3, 8), (4, 16)):
                    keys["cell%d|octa%dx%d" % (b, nb, nb)] = combine(cell_key(org, lo, hi, b), octa_bin(dr, nb))
                keys["octa8x8|cell5"] = combine(octa_bin(dr, 8), cell_key(org, lo, hi, 5))
                keys["octant|cell6"] = combine(oc, cell_key(org, lo, hi, 6))
                keys["octa4x4|cell10"] = combine(octa_bin(dr, 4), cell_key(org, lo, hi, 10))
                keys["cell5|octant|cell10"] = combine(cell_key(org, lo, hi, 5), oc, cell_key(org, lo, hi, 10))
                keys["cell4|octa4x4|cell10"] = combine(cell_key(org, lo, hi, 4), octa_bin(dr, 4), cell_key(org, lo, hi, 10))
                idx = np.arange(n, dtype=np.uint64)
                for blk in (256, 1024, 4096, 16384):
                    keys["blocks of %d: octant" % blk] = ((idx // np.uint64(blk)) << np.uint64(3)) | oc[0]
                    keys["blocks of %d: octa4x4" % blk] = ((idx // np.uint64(blk)) << np.uint64(4)) | octa_bin(dr, 4)[0]
            if wave == "primary":  # pixel tiles (rays are row-major W wide)
                W = wl.width
                x, y = np.arange(n, dtype=np.uint64) % np.uint64(W), np.arange(n, dtype=np.uint64) // np.uint64(W)
                keys["pixel Morton (8x8 tiles and up)"] = part1by1(x) | (part1by1(y) << np.uint64(1))
                keys["8x8 tiles row-major"] = ((y // np.uint64(8)) * np.uint64((W + 7) // 8) + x // np.uint64(8)) * np.uint64(64) + (y % np.uint64(8)) * np.uint64(8) + x % np.uint64(8)
            for label, key in keys.items():
                perm = np.argsort(key, kind="stable")
                ms, out = timeit(a, rays[perm])
                got = out.cpu().numpy().reshape(n, -1)
                same = np.array_equal(got, base_hits.reshape(n, -1)[perm])
                print("%-8s %-34s %.4f ms  x%.3f  records %s" % (wave, label, ms, base_ms / ms, "identical" if same else "DIFFER"), flush=True)
        # layout probe: the private node records in a pseudo-random order
        if not quick:
            from nanort_amd import BVHAccel, TriangleMesh

            acc2 = BVHAccel(wl.real)
            acc2.SetTunable("wide_scramble", 1)
            mesh = TriangleMesh(wl.verts, wl.faces)
            assert acc2.Build(mesh.num_faces, mesh)
            for wave, rays in (("bounce", wl.rays2), ("primary", wl.rays1)):
                m0, o0 = timeit(a, rays)
                m1, o1 = timeit(acc2, rays)
                print("%-8s node records scrambled: %.4f ms vs pre-order %.4f ms (x%.3f)  records %s" % (
                    wave, m1, m0, m1 / m0, "identical" if torch.equal(o0, o1) else "DIFFER"), flush=True)
            acc2.close()
        del wl
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

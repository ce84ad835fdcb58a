"""Randomised LIFECYCLE soak of one context (round 6): the order of calls, not the geometry, is what is random here.
A context is rebuilt over meshes of very different sizes while launches of its previous tree are still in flight on several streams
(more streams than launch slots), its scheduling tunables and walk variants flip between launches, closest-hit, occlusion and
multi-batch launches of ragged sizes are mixed with host-buffer calls — and every result must equal, byte for byte, what a FRESH
context with default tunables returns for the same mesh and rays on the default stream (builds are deterministic: same tree).
What this exercises: the launch slots' cursor sets and completion records across variants, the rebuild's wait for launches in
flight, grow-only buffers shrinking and growing, the slot take-over by a fifth stream, and that no tunable changes a record.
Usage: python tests/checks/fuzz_lifecycle.py [seconds] [seed]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import torch  # noqa: E402

from nanort_amd import BVHAccel, TriangleMesh, scenes  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
MESHES = [(2, 1), (3, 3), (20, 10), (100, 60), (300, 200), (700, 400)]  # Plane(nx, ny): 4 ... 560 000 triangles
SIZES = [1, 63, 64, 65, 1000, 4097, 60000, 250000]
TUN = {"static_pct": (0, 16, 16, 40, 75, 100), "static_bands": (1, 2, 8), "chunk": (32, 64, 128), "parts": (1, 3, 8), "refill_min": (1, 24, 44, 64),
       "trav_min4": (1, 12, 24, 48), "leaf_min": (1, 32), "leaf_compact": (0, 1), "wide4_big": (1, 2), "launch_timing": (0, 1), "dyn_head": (0, 1), "chunk_tail_pct": (0, 25)}
streams = [torch.cuda.Stream() for _ in range(6)]
all_rays = scenes.camera_rays(640, 400)


def expected(v, f, rays, kind):
    """a fresh context, default tunables, default stream"""
    a = BVHAccel(np.float32)
    assert a.Build(f.shape[0], TriangleMesh(v, f))
    out = a.OccludedBatch(rays) if kind == "occ" else a.TraverseBatch(rays)
    a.close()
    return out


ctx = BVHAccel(np.float32)
t_end = time.time() + budget
rounds = launches = rebuilds = 0
cur = None
pending = []  # (kind, device outputs, expected)
while time.time() < t_end:
    rounds += 1
    # ---- (re)build, possibly while the previous tree's launches are still in flight -----------------------------------
    if cur is None or rng.random() < 0.45:
        nx, ny = MESHES[int(rng.integers(0, len(MESHES)))]
        v, f = scenes.plane(nx, ny)
        if rng.random() < 0.3:  # the same topology, moved: a per-frame rebuild
            v = (v + rng.normal(scale=0.01, size=v.shape)).astype(np.float32)
        if rng.random() < 0.5:
            torch.cuda.synchronize()  # (else: the library itself must wait for what is in flight)
        assert ctx.Build(f.shape[0], TriangleMesh(v, f))
        rebuilds += 1
        # launches of the OLD tree were issued before the rebuild: the library waited for them, so they can be checked now
        torch.cuda.synchronize()
        for kind, outs, want in pending:
            if kind == "occ":
                assert np.array_equal(outs[0].cpu().numpy(), want), "occlusion flags differ (round %d)" % rounds
            else:
                assert outs[0].cpu().numpy().tobytes() == want[0].tobytes(), "records differ (round %d, %s)" % (rounds, kind)
                assert np.array_equal(outs[1].cpu().numpy(), want[1]), "flags differ (round %d, %s)" % (rounds, kind)
        pending = []
        cur = (v, f)
        exp_cache = {}
    v, f = cur
    for k, choices in TUN.items():
        if rng.random() < 0.3:
            ctx.SetTunable(k, int(choices[int(rng.integers(0, len(choices)))]))
    # ---- a burst of launches on several streams, no host synchronisation in between -----------------------------------
    for _ in range(int(rng.integers(1, 9))):
        n = SIZES[int(rng.integers(0, len(SIZES)))]
        off = int(rng.integers(0, all_rays.shape[0] - n))
        rays = all_rays[off:off + n]
        kind = str(rng.choice(["hit", "hit", "occ", "multi", "host"]))
        key = (off, n, "occ" if kind == "occ" else "hit")
        if key not in exp_cache:
            exp_cache[key] = expected(v, f, rays, key[2])
        want = exp_cache[key]
        s = streams[int(rng.integers(0, len(streams)))]
        launches += 1
        if kind == "host":
            h, m = ctx.TraverseBatch(rays)
            assert h.tobytes() == want[0].tobytes() and np.array_equal(m, want[1]), "host call differs (round %d)" % rounds
            continue
        d_r = torch.from_numpy(rays.view(np.uint8)).cuda()
        if kind == "occ":
            d_m = torch.zeros(n, dtype=torch.uint8, device="cuda")
            torch.cuda.current_stream().synchronize()
            with torch.cuda.stream(s):
                ctx.OccludedBatchDevice(d_r, d_m)
            pending.append(("occ", (d_m, d_r), want))
        elif kind == "multi" and n >= 2:
            cut = int(rng.integers(1, n))
            parts = [(d_r[: cut * 36], cut), (d_r[cut * 36:], n - cut)]
            outs = [(torch.zeros(c * 16, dtype=torch.uint8, device="cuda"), torch.zeros(c, dtype=torch.uint8, device="cuda")) for _, c in parts]
            torch.cuda.current_stream().synchronize()
            with torch.cuda.stream(s):
                ctx.TraverseBatchesDevice([(p[0], o[0], o[1]) for p, o in zip(parts, outs)])
            pending.append(("multi-a", (outs[0][0], outs[0][1], d_r), (want[0][:cut], want[1][:cut])))
            pending.append(("multi-b", (outs[1][0], outs[1][1], d_r), (want[0][cut:], want[1][cut:])))
        else:
            d_h = torch.zeros(n * 16, dtype=torch.uint8, device="cuda")
            d_m = torch.zeros(n, dtype=torch.uint8, device="cuda")
            torch.cuda.current_stream().synchronize()
            with torch.cuda.stream(s):
                ctx.TraverseBatchDevice(d_r, d_h, d_m)
            pending.append(("hit", (d_h, d_m, d_r), want))
torch.cuda.synchronize()
for kind, outs, want in pending:
    if kind == "occ":
        assert np.array_equal(outs[0].cpu().numpy(), want)
    else:
        assert outs[0].cpu().numpy().tobytes() == want[0].tobytes() and np.array_equal(outs[1].cpu().numpy(), want[1])
print("fuzz_lifecycle ok: %d rounds, %d rebuilds, %d launches, seed %d" % (rounds, rebuilds, launches, seed))

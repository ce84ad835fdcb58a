"""One persistent launch over several batches (nrtTraverseBatchesDevice) against separate launches and against two frames in
flight on two streams:   python tools/multi_probe.py C3 [order4]"""
import sys

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402

import bench  # noqa: E402

name = sys.argv[1]
wl = bench.Workload(name, builds=1)
a = wl.accel
if len(sys.argv) > 2:
    for kv in sys.argv[2:]:
        k, v = kv.split("=")
        a.SetTunable(k, int(v))
b1 = (wl.d_rays1, wl.d_hits1, wl.d_mask1, wl.n1)
b2 = (wl.d_rays2, wl.d_hits2, wl.d_mask2, wl.n2)
# a second frame's buffers (same rays, own outputs)
h1b, m1b = torch.empty_like(wl.d_hits1), torch.empty_like(wl.d_mask1)
h2b, m2b = torch.empty_like(wl.d_hits2), torch.empty_like(wl.d_mask2)
b1b, b2b = (wl.d_rays1, h1b, m1b, wl.n1), (wl.d_rays2, h2b, m2b, wl.n2)
K = 10


def timed(fn, reps=5):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(K):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / K)
    return float(np.median(out))


def separate():
    a.TraverseBatchDevice(*b1[:3])
    a.TraverseBatchDevice(*b2[:3])


def one_launch():
    a.TraverseBatchesDevice([b1, b2])


def two_frames_one_launch():
    a.TraverseBatchesDevice([b1, b2, b1b, b2b])


separate()
torch.cuda.synchronize()
ref1, ref2 = wl.d_hits1.clone(), wl.d_hits2.clone()
one_launch()
torch.cuda.synchronize()
same = bool(torch.equal(ref1, wl.d_hits1) and torch.equal(ref2[: wl.n2 * 16], wl.d_hits2[: wl.n2 * 16]))
n = wl.n1 + wl.n2
t_sep, t_one, t_two = timed(separate), timed(one_launch), timed(two_frames_one_launch)
print("%s %s: separate launches %.4f ms = %.0f Mrays/s | one launch over both waves %.4f ms = %.0f Mrays/s | two frames (4 batches) in one launch %.4f ms per frame = %.0f Mrays/s | records identical: %s" % (
    name, sys.argv[2:], t_sep, n / t_sep / 1e3, t_one, n / t_one / 1e3, t_two / 2, n / (t_two / 2) / 1e3, same))

import sys, time, numpy as np
sys.path.insert(0, '.')
from nanort_amd import BVHAccel, TriangleMesh, scenes
v, f = scenes.plane(1000, 500); mesh = TriangleMesh(v, f)
rays = scenes.camera_rays(1920, 1080)
a = BVHAccel(np.float32); a.Build(mesh.num_faces, mesh)
for _ in range(2): a.TraverseBatch(rays)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); h, m = a.TraverseBatch(rays); ts.append(time.perf_counter() - t0)
print('host entry point (pageable H2D 74.6 MB + kernel + D2H 35.3 MB): %.2f ms => %.1f Mrays/s; kernel alone %.3f ms' % (np.median(ts) * 1e3, len(rays) / np.median(ts) / 1e6, a.LastTraverseMs()))
t0 = time.perf_counter(); a.SetMesh(mesh); t1 = time.perf_counter(); ok = a.Build(mesh.num_faces, mesh); t2 = time.perf_counter(); n, i = a.GetTree(); t3 = time.perf_counter()
print('SetMesh %.2f ms, Build (host wall) %.2f ms (device %.2f), GetTree %.2f ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, a.LastBuildMs(), (t3 - t2) * 1e3))
# the same call with PAGE-LOCKED caller buffers (nrtHostAlloc / hipHostMalloc): upload, trace and download are pipelined
import ctypes, os, torch
from nanort_amd.wire import HIT_F32
L = a._L
pr = torch.empty(rays.nbytes, dtype=torch.uint8, pin_memory=True); pr.numpy()[:] = rays.view(np.uint8)
ph = torch.empty(len(rays) * HIT_F32.itemsize, dtype=torch.uint8, pin_memory=True); pm = torch.empty(len(rays), dtype=torch.uint8, pin_memory=True)
def call():
    st = L.nrtTraverseBatch_f32(a._h, pr.data_ptr(), len(rays), None, ph.data_ptr(), pm.data_ptr()); assert st == 0
for _ in range(2): call()
ts = []
for _ in range(7):
    t0 = time.perf_counter(); call(); ts.append(time.perf_counter() - t0)
same = ph.numpy().tobytes() == h.tobytes() and pm.numpy().tobytes() == m.tobytes()
print('host entry point, page-locked buffers, pipelined in 512K-ray pieces: %.2f ms => %.1f Mrays/s (records identical: %s)' % (np.median(ts) * 1e3, len(rays) / np.median(ts) / 1e6, same))
os.environ['NRT_HOST_PIPELINE'] = '0'
b = BVHAccel(np.float32); b.Build(mesh.num_faces, mesh)
def call2():
    st = L.nrtTraverseBatch_f32(b._h, pr.data_ptr(), len(rays), None, ph.data_ptr(), pm.data_ptr()); assert st == 0
for _ in range(2): call2()
ts = []
for _ in range(7):
    t0 = time.perf_counter(); call2(); ts.append(time.perf_counter() - t0)
print('host entry point, page-locked buffers, NOT pipelined: %.2f ms => %.1f Mrays/s' % (np.median(ts) * 1e3, len(rays) / np.median(ts) / 1e6))

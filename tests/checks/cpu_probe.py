import sys, os, time, numpy as np
sys.path.insert(0, '.')
from oracle import bindings as ob
from nanort_amd import scenes
print('cpu.max', open('/sys/fs/cgroup/cpu.max').read().strip() if os.path.exists('/sys/fs/cgroup/cpu.max') else 'n/a', 'nproc', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
v, f = scenes.plane(1000, 500)
R = ob.Reference(v, f)
t0 = time.time(); ok, st = R.build(parallel=True); print('ref parallel build s', time.time() - t0, st)
t0 = time.time(); ok, st = R.build(parallel=False); print('ref serial build s', time.time() - t0)
rays = scenes.camera_rays(1920, 1080)
sub = rays.reshape(-1, 1920)[::8].reshape(-1)
for th in (4, 8, 16, 32, 64, 128, 256):
    _, _, s = R.traverse(sub, threads=th, chunk=1920)
    print('threads', th, 'Mrays/s %.3f' % (len(sub) / s / 1e6), flush=True)

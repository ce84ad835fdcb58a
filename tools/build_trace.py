"""Timeline of ONE 1M-triangle build from a rocprofv3 kernel trace: every kernel of the last build in launch order
with its duration and the idle gap before it, then totals per kernel and the sum of the gaps.
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/bt -- python tools/build_trace.py run
  python tools/build_trace.py summarize gpurun_out/bt"""
import csv, glob, os, sys
import numpy as np
sys.path.insert(0, '.')
if sys.argv[1] == 'run':
    from nanort_amd import BVHAccel, TriangleMesh, scenes
    gx, gy = (int(x) for x in os.environ.get('NRT_TRACE_GRID', '1000x500').split('x'))
    v, f = scenes.plane(gx, gy)
    a = BVHAccel(np.float32); m = TriangleMesh(v, f)
    for _ in range(4):
        a.Build(m.num_faces, m)
    print("build ms", a.LastBuildMs())
else:
    path = glob.glob(os.path.join(sys.argv[2], '**', '*kernel_trace.csv'), recursive=True)[0]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    names = [r['Kernel_Name'].split('(')[0].replace('void nrt::', '').replace('nrt::', '') for r in rows]
    starts = [i for i, n in enumerate(names) if n.startswith('k_init_scene')]
    lo = starts[-1]
    hi = len(rows)
    t0 = int(rows[lo]['Start_Timestamp'])
    prev_end = t0
    tot, gaps, cnt = {}, 0, {}
    for i in range(lo, hi):
        s, e = int(rows[i]['Start_Timestamp']), int(rows[i]['End_Timestamp'])
        gap = max(0, s - prev_end)
        gaps += gap
        n = names[i].split('<')[0]
        tot[n] = tot.get(n, 0) + (e - s); cnt[n] = cnt.get(n, 0) + 1
        print("%8.1f us  +%5.1f gap  %7.1f us  %-18s grid %s" % ((s - t0) / 1e3, gap / 1e3, (e - s) / 1e3, n, rows[i].get('Grid_Size_X', rows[i].get('Grid_Size', '?'))))
        prev_end = max(prev_end, e)
    print("span %.1f us, kernels %.1f us, gaps %.1f us" % ((prev_end - t0) / 1e3, sum(tot.values()) / 1e3, gaps / 1e3))
    for n, t in sorted(tot.items(), key=lambda kv: -kv[1]):
        print("  %-18s x%-3d %8.1f us" % (n, cnt[n], t / 1e3))

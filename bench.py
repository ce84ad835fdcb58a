#!/usr/bin/env python3
"""bench.py — the headline measurement of the hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Metric (BASELINE.json): Mrays/s (primary + 1-bounce) at 1920x1080 on the
1M-triangle mesh; BVH build ms.  Workload = config C3 of SURVEY.md §8(d):
Plane(1000,500) (exactly 1 000 000 triangles), fp32, objrender camera.

One "step" = one pass of the hot path over one batch: wave 1 (W*H primary
rays) + wave 2 (one cosine-weighted bounce ray per wave-1 hit), both already
resident in HBM, traced by the batched traversal kernel through the C ABI
(nrtTraverseBatchDevice_f32) on torch's current stream.  The BVH is built on
the GPU (nrtBuild_f32) before the timed region; its device time is reported
as `build_ms` (median of several builds).

N > 1 (weak scaling): the image grows to 1920 x (1080*N) and rank r traces the
interleaved rows y = r (mod N), i.e. 1920x1080 rays per GPU, over its own
replica of the BVH (deterministic build, no broadcast).  The wave-1 hit
records are gathered to rank 0 with one RCCL gather per step (grouped send/recv),
issued asynchronously and double-buffered so it overlaps wave 2 and the next step.

Extra objects on the JSON line:
  roofline      dominant kernel k_traverse<float>: ALGORITHMIC bytes per launch
                (52 + 40*nodes_visited + 52*tris_tested per ray, SURVEY.md §8d,
                counted on the tree actually traversed by the kernel's own
                counting pass) / mean launch duration measured live with HIP
                events on the launch stream; peak = 8 TB/s HBM3E.
  cpu_baseline  the UNMODIFIED reference (oracle/_ref, OpenMP, all host cores)
                on a bounded sample of the same ray buffers; falls back to the
                single-thread C port (oracle/liboracle.so) when _ref is absent.
                Carries the parity check of the same run: the GPU's
                records of the timed waves against the reference on its own tree
                (1e-5 tolerance, prim ids equal except at exact-t ties) and
                against the reference over the GPU-built tree (bit-identical).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
WIDTH, HEIGHT = 1920, 1080


def algorithmic_bytes(counters, real_bytes=4):
    """per ray sizeof(Ray)+sizeof(Hit) + 40 B per node visit + 52 B per triangle test (fp32)."""
    if real_bytes == 4:
        return 52 * counters["num_rays"] + 40 * counters["nodes_visited"] + 52 * counters["tris_tested"]
    return 104 * counters["num_rays"] + 64 * counters["nodes_visited"] + 88 * counters["tris_tested"]


def parity(ref_hits, ref_mask, gpu_hits, gpu_mask):
    """SURVEY 8(d) parity check of one ray set: hit flags equal; |dt|, |du|, |dv| <= 1e-5 * max(1, |ref|); prim ids equal,
    a different prim id being tolerated only at a true tie (both primitives at the same t: the reference keeps whichever
    it tested last, so across different trees either may be named; u, v then belong to the named primitive)."""
    both = (ref_mask == 1) & (gpu_mask == 1)

    def rel(k, sel):
        r = ref_hits[k][sel].astype(np.float64)
        g = gpu_hits[k][sel].astype(np.float64)
        return float(np.max(np.abs(g - r) / np.maximum(1.0, np.abs(r)))) if r.size else 0.0

    same_prim = both & (ref_hits["prim_id"] == gpu_hits["prim_id"])
    other_prim = both & ~same_prim
    return {
        "rays": int(ref_mask.shape[0]),
        "hit_flag_mismatches": int((ref_mask != gpu_mask).sum()),
        "max_rel_err_t": rel("t", both),
        "max_rel_err_u_v_same_prim": max(rel("u", same_prim), rel("v", same_prim)),
        "prim_id_mismatches": int(other_prim.sum()),
        "prim_id_mismatches_at_exact_t_ties": int((other_prim & (ref_hits["t"] == gpu_hits["t"])).sum()),
        "within_tolerance_1e-5": bool(rel("t", both) <= 1e-5 and max(rel("u", same_prim), rel("v", same_prim)) <= 1e-5
                                      and int((ref_mask != gpu_mask).sum()) == 0
                                      and int(other_prim.sum()) == int((other_prim & (ref_hits["t"] == gpu_hits["t"])).sum())),
    }


def cpu_baseline(verts, faces, rays1, rays2, gpu_nodes, gpu_indices, gpu_results=None, budget_s=12.0):
    """Reference (or port) timed on the host cores over a bounded sample of the same buffers; with `gpu_results` =
    (hits1, mask1, hits2, mask2) of the GPU also the parity check of the same run."""
    from oracle import bindings as ob

    total = rays1.shape[0] + rays2.shape[0]
    if ob.reference_available():
        R = ob.Reference(verts, faces)
        # usable host parallelism: the box may expose more logical CPUs than its cgroup quota allows;
        # oversubscribing a quota makes OpenMP collapse, so probe a few thread counts and keep the best
        quota = None
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            quota = None if q == "max" else max(1, int(int(q) / int(per)))
        except Exception:
            quota = None
        hw = R.max_threads()
        cands = sorted({t for t in ((quota or hw), 2 * (quota or hw), hw) if 1 <= t <= hw})
        ok, st = R.build(parallel=True, threads=cands[0])
        build_ms = st["build_secs"] * 1e3
        probe = rays1.reshape(-1, WIDTH)[::40].reshape(-1)
        best_t, rate = cands[0], 0.0
        for t in cands:
            _, _, secs = R.traverse(probe, threads=t, chunk=WIDTH)
            if probe.shape[0] / secs > rate:
                best_t, rate = t, probe.shape[0] / secs
        frac = min(1.0, budget_s * rate / total)
        rows1 = max(8, int(rays1.shape[0] // WIDTH * frac))
        step = max(1, (rays1.shape[0] // WIDTH) // rows1)
        s1 = rays1.reshape(-1, WIDTH)[::step].reshape(-1)
        s2 = rays2[:: max(1, step)]
        best = 1e30
        for _ in range(2):
            rh1, rm1, t1 = R.traverse(s1, threads=best_t, chunk=WIDTH)
            rh2, rm2, t2 = R.traverse(s2, threads=best_t, chunk=WIDTH)
            best = min(best, t1 + t2)
        value = (s1.shape[0] + s2.shape[0]) / best / 1e6
        out = {
            "value": round(value, 4), "unit": "Mrays/s", "cores": int(best_t), "kind": "reference",
            "sample": "unmodified nanort.h (g++ -O3 -fopenmp, own parallel Build: %d nodes, depth %d), "
                      "every %d-th row of wave 1 (%d rays) + every %d-th wave-2 ray (%d rays), omp dynamic row loop, "
                      "best of 2; %d OpenMP threads = best of %s (host: %d logical CPUs, cgroup quota %s)" % (
                          st["num_leaf_nodes"] + st["num_branch_nodes"], st["max_tree_depth"], step,
                          s1.shape[0], step, s2.shape[0], best_t, cands, hw, quota),
            "build_ms": round(build_ms, 1),
        }
        if ob.reference_v3_available():  # the same code with -march=x86-64-v3: the stronger timing baseline of SURVEY 8(d)
            try:
                R3 = ob.ReferenceV3(verts, faces)
                R3.build(parallel=True, threads=cands[0])
                _, _, t1 = R3.traverse(s1, threads=best_t, chunk=WIDTH)
                _, _, t2 = R3.traverse(s2, threads=best_t, chunk=WIDTH)
                out["value_march_x86_64_v3"] = round((s1.shape[0] + s2.shape[0]) / (t1 + t2) / 1e6, 4)
            except Exception as e:  # pragma: no cover
                out["value_march_x86_64_v3"] = None
                out["v3_error"] = repr(e)
        # same traversal code over the GPU-built node array: separates "better tree" from "faster traversal"
        if gpu_results is not None:  # reference on ITS tree vs GPU on the GPU-built tree: equal up to exact-t ties in prim_id / u / v
            gh1, gm1, gh2, gm2 = gpu_results
            W = WIDTH
            out["parity_own_trees"] = {
                "primary": parity(rh1, rm1, gh1.reshape(-1, W)[::step].reshape(-1), gm1.reshape(-1, W)[::step].reshape(-1)),
                "bounce": parity(rh2, rm2, gh2[:: max(1, step)], gm2[:: max(1, step)])}
        if R.load_tree(gpu_nodes, gpu_indices):
            th1, tm1, t1 = R.traverse(rays1, threads=best_t, chunk=WIDTH)
            th2, tm2, t2 = R.traverse(rays2, threads=best_t, chunk=WIDTH)
            out["value_on_gpu_built_tree"] = round(total / (t1 + t2) / 1e6, 4)
            if gpu_results is not None:  # same node array: every field must be bit-identical
                gh1, gm1, gh2, gm2 = gpu_results
                same = all(np.array_equal(a, b) for a, b in ((tm1, gm1), (tm2, gm2)))
                for k in ("t", "u", "v", "prim_id"):
                    same = same and th1[k].tobytes() == gh1[k].tobytes() and th2[k].tobytes() == gh2[k].tobytes()
                out["parity_same_tree_bit_identical"] = bool(same)
        return out
    O = ob.Oracle()
    t0 = time.time()
    nodes, idx, _ = O.build(verts, faces)
    build_ms = (time.time() - t0) * 1e3
    s1 = rays1.reshape(-1, WIDTH)[::54].reshape(-1)
    t0 = time.time()
    O.traverse(nodes, idx, verts, faces, s1)
    dt = time.time() - t0
    return {"value": round(s1.shape[0] / dt / 1e6, 4), "unit": "Mrays/s", "cores": 1, "kind": "port",
            "sample": "liboracle.so single thread, every 54th row of wave 1 (%d rays)" % s1.shape[0],
            "build_ms": round(build_ms, 1)}


def pipelined(accel, torch, wave1, wave2, steps, rays_per_step, frames_in_flight=2):
    """K steps with `frames_in_flight` independent frames in flight, one stream per frame (one context: every
    launch owns a launch slot).  Same work as the timed region; reported beside it, never as `value`."""
    streams = [torch.cuda.Stream() for _ in range(frames_in_flight)]
    bufs = [(wave1, wave2)] + [tuple((w[0], torch.empty_like(w[1]), torch.empty_like(w[2])) for w in (wave1, wave2))
                               for _ in range(frames_in_flight - 1)]
    def run(k):
        for i in range(k):
            w1, w2 = bufs[i % frames_in_flight]
            with torch.cuda.stream(streams[i % frames_in_flight]):
                accel.TraverseBatchDevice(*w1)
                accel.TraverseBatchDevice(*w2)
    run(2 * frames_in_flight)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"frames_in_flight": frames_in_flight, "value": round(rays_per_step * steps / dt / 1e6, 1), "unit": "Mrays/s",
            "ms_per_step": round(dt / steps * 1e3, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--builds", type=int, default=5)
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed extras (2 frames in flight, primary+shadow): use for rocprofv3 runs, so that the "
                         "kernel statistics hold the timed region's launches only")
    args = ap.parse_args()

    import torch

    from nanort_amd import BVHAccel, TriangleMesh, scenes
    from nanort_amd.wire import HIT_F32, RAY_F32

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    # Test hook (tests/test_bench_multi.py): NRT_BENCH_TEST_SHARED_GPU=1 lets N ranks share GPU 0 so that the N > 1
    # control flow can be run on a one-GPU box; RCCL cannot put two ranks on one device, so the collectives then go
    # through gloo with CPU staging.  Never set by the driver: its numbers come from one rank per GPU over RCCL.
    shared = world > 1 and os.environ.get("NRT_BENCH_TEST_SHARED_GPU") == "1"
    if shared:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    comm_dev = "cpu" if shared else "cuda"
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    # ---- mesh + BVH (replicated per rank, deterministic) ----------------------
    verts, faces = scenes.plane(1000, 500)
    mesh = TriangleMesh(verts, faces)
    accel = BVHAccel(np.float32, device=local_rank)
    build_ms = []
    for _ in range(max(1, args.builds)):
        assert accel.Build(mesh.num_faces, mesh)
        build_ms.append(accel.LastBuildMs())
    stats = accel.GetStatistics()

    # ---- rays: wave 1 (this rank's interleaved rows), wave 2 from its hits --------
    H_glob = HEIGHT * world
    rays1 = scenes.camera_rays_rows(WIDTH, H_glob, rank, world, HEIGHT)
    n1 = rays1.shape[0]
    d_rays1 = torch.from_numpy(rays1.view(np.uint8)).cuda()
    d_hits1 = torch.empty(n1 * HIT_F32.itemsize, dtype=torch.uint8, device="cuda")
    d_mask1 = torch.empty(n1, dtype=torch.uint8, device="cuda")
    accel.TraverseBatchDevice(d_rays1, d_hits1, d_mask1)
    torch.cuda.synchronize()
    hits1 = d_hits1.cpu().numpy().view(HIT_F32)
    mask1 = d_mask1.cpu().numpy()
    # pixel index of ray i in the global image: row (rank + world * (i // W)), column i % W
    rays2 = scenes.secondary_rays("bounce", verts, faces, rays1, hits1, mask1, pixel_base=rank * n1)
    n2 = rays2.shape[0]
    d_rays2 = torch.from_numpy(rays2.view(np.uint8)).cuda()
    d_hits2 = torch.empty(max(1, n2) * HIT_F32.itemsize, dtype=torch.uint8, device="cuda")
    d_mask2 = torch.empty(max(1, n2), dtype=torch.uint8, device="cuda")
    # rank 0 assembles the frame: it receives every rank's records over its 7 direct xGMI links at once (33 MB each),
    # the other ranks only send.  The records are double-buffered so that the gather of step k (RCCL, its own stream)
    # overlaps wave 2 of step k and all of step k+1; it is waited for before its buffers are reused.
    hit_bufs = [d_hits1, torch.empty_like(d_hits1)] if world > 1 else [d_hits1]
    from nanort_amd import dist as nd

    proto = torch.empty(0, dtype=torch.uint8, device=comm_dev)
    gathered = [None, None]
    if world > 1 and rank == 0:
        gathered = [torch.empty(world * n1 * HIT_F32.itemsize, dtype=torch.uint8, device=comm_dev) for _ in range(2)]
    del proto
    pending = [None, None]

    # ---- work counters -> algorithmic bytes per launch ---------------------------
    c1 = accel.TraverseCountDevice(d_rays1)
    c2 = accel.TraverseCountDevice(d_rays2)
    bytes1, bytes2 = algorithmic_bytes(c1), algorithmic_bytes(c2)

    step_no = [0]

    def step(ev=None):
        b = step_no[0] % len(hit_bufs)
        step_no[0] += 1
        if world > 1 and pending[b] is not None:
            pending[b].wait()  # the gather that last used this buffer pair (two steps ago)
            pending[b] = None
        if ev is not None:
            ev[0].record()
        accel.TraverseBatchDevice(d_rays1, hit_bufs[b], d_mask1)
        if ev is not None:
            ev[1].record()
        if world > 1:
            src = hit_bufs[b].cpu() if shared else hit_bufs[b]  # (test hook: staged through the host for gloo)
            _, pending[b] = nd.gather_hit_records(src, world, rank, dist, out=gathered[b], async_op=True)
        if ev is not None:
            ev[2].record()
        accel.TraverseBatchDevice(d_rays2, d_hits2, d_mask2)
        if ev is not None:
            ev[3].record()

    def drain():
        for b in range(2):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    for _ in range(args.warmup):
        step()
    drain()
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])
    drain()  # every gather issued inside the timed region completes inside it
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0

    k_ms1 = float(np.mean([e[0].elapsed_time(e[1]) for e in events]))
    k_ms2 = float(np.mean([e[2].elapsed_time(e[3]) for e in events]))
    rays_per_step = n1 + n2
    if world > 1:
        t = torch.tensor([dt, float(rays_per_step), float(bytes1 + bytes2), k_ms1 + k_ms2], dtype=torch.float64,
                         device=comm_dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt = float(tmax[0])
        total_rays = float(tsum[1])
    else:
        total_rays = float(rays_per_step)

    if rank == 0:
        value = total_rays * args.steps / dt / 1e6
        achieved = (bytes1 + bytes2) / ((k_ms1 + k_ms2) * 1e-3) / 1e9  # GB/s, this rank's two launches
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic_c3.json")
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/s (primary + 1-bounce) at 1920x1080, 1M-tri mesh; BVH build ms",
            "value": round(value, 3),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "C3: Plane(1000,500) = 1,000,000 triangles fp32; %dx%d objrender-camera primaries "
                            "+ 1 cosine bounce per hit (%d + %d rays per GPU per step)" % (WIDTH, HEIGHT, n1, n2),
                "parallelism": "replicated BVH, interleaved image rows per GPU%s" % (
                    ", RCCL gather of the wave-1 hit records to rank 0 (send/recv over xGMI), double-buffered and overlapped with the following waves" if world > 1 else ""),
                "rays_per_step": int(total_rays),
            },
            "build_ms": round(float(np.median(build_ms)), 4),
            "bvh": {"nodes": int(stats["num_leaf_nodes"] + stats["num_branch_nodes"]),
                    "max_depth": int(stats["max_tree_depth"])},
            "roofline": {
                "bound": "hbm",
                "kernel": "nrt::k_traverse_wide<float,10>",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "algorithmic_bytes_per_launch": int((bytes1 + bytes2) // 2),
                "launch_ms": round((k_ms1 + k_ms2) / 2, 4),
                "per_wave": {
                    "primary": {"ms": round(k_ms1, 4), "rays": n1, "nodes_per_ray": round(c1["nodes_visited"] / n1, 2),
                                "tris_per_ray": round(c1["tris_tested"] / n1, 2), "bytes": int(bytes1)},
                    "bounce": {"ms": round(k_ms2, 4), "rays": n2,
                               "nodes_per_ray": round(c2["nodes_visited"] / max(1, n2), 2),
                               "tris_per_ray": round(c2["tris_tested"] / max(1, n2), 2), "bytes": int(bytes2)},
                },
            },
        }
        if world == 1 and not args.no_extras:
            # extras, outside the timed region: (a) the same K steps with two frames in flight (steps alternate
            # between two streams; a launch's drain tail is filled by the next frame's rays), (b) SURVEY 8(d)'s
            # primary + shadow pair
            out["pipelined"] = pipelined(accel, torch, (d_rays1, d_hits1, d_mask1), (d_rays2, d_hits2, d_mask2), args.steps, n1 + n2)
            rays_s = scenes.secondary_rays("shadow", verts, faces, rays1, hits1, mask1)
            d_rs = torch.from_numpy(rays_s.view(np.uint8)).cuda()
            d_hs = torch.empty(max(1, rays_s.shape[0]) * HIT_F32.itemsize, dtype=torch.uint8, device="cuda")
            ts = []
            for _ in range(5):
                accel.TraverseBatchDevice(d_rs, d_hs)
                ts.append(accel.LastTraverseMs())
            ms_s = float(np.median(ts))
            out["primary_plus_shadow"] = {"value": round((n1 + rays_s.shape[0]) / (k_ms1 + ms_s) / 1e3, 1), "unit": "Mrays/s",
                                          "shadow_ms": round(ms_s, 4), "shadow_rays": int(rays_s.shape[0])}
        if world == 1 and not args.no_cpu_baseline:
            nodes, indices = accel.GetTree()
            torch.cuda.synchronize()
            gpu_results = (d_hits1.cpu().numpy().view(HIT_F32), d_mask1.cpu().numpy(), d_hits2.cpu().numpy().view(HIT_F32)[:n2],
                           d_mask2.cpu().numpy()[:n2])
            out["cpu_baseline"] = cpu_baseline(verts, faces, rays1, rays2, nodes, indices, gpu_results)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

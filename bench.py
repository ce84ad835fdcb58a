#!/usr/bin/env python3
"""bench.py — the headline measurement of the hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2|C3|C4|C4tile|C5] [--mesh file.ply|.obj]

Metric (BASELINE.json): Mrays/s (primary + 1-bounce) at 1920x1080 on the 1M-triangle mesh; BVH build ms.
Default workload = config C3 of SURVEY.md §8(d): Plane(1000,500) (exactly 1 000 000 triangles), fp32, objrender
camera.  The other single-GPU configs of BASELINE.json are measured too: untimed, in the `configs` object of the
default line (C2 stand-in, the C4 4096x512 tile, C5 fp64), or as the headline with `--config`.

One "step" = one pass of the hot path over one batch: wave 1 (W*H primary rays) + wave 2 (one cosine-weighted
bounce ray per wave-1 hit), both already resident in HBM, traced by the batched traversal kernel through the C ABI
(nrtTraverseBatchDevice_*) on torch's current stream.  The BVH is built on the GPU (nrtBuild_*) before the timed
region; its device time is reported as `build_ms` (median of several builds).

N > 1: `python bench.py --gpus N` starts N ranks itself (torch.distributed.run, one per GPU, RCCL) when it was not
launched under a launcher already, and fails loudly when the box has fewer than N GPUs.  Each rank holds a replica
of the BVH (deterministic GPU build, no broadcast) and traces the interleaved image rows y = rank (mod N); the hit
records of BOTH waves are gathered to rank 0 (RCCL gather = grouped send/recv over xGMI), asynchronously and
double-buffered, so the exchange overlaps the following waves.  C2/C3/C5 scale weakly (the image grows to
W x (H*N): fixed work per GPU); `--config C4` is BASELINE.json's strong-scaling case: Plane(2500,2000) = 10M
triangles, a fixed 4096x4096 frame cut into N row-interleaved tiles.

Extra objects on the JSON line:
  roofline      the dominant kernel (named by the library: nrtLastKernelName).  `hbm`: HBM-side bytes per launch
                MEASURED IN THIS RUN (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over a 3-step sub-run of this
                same script, outside the timed region; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
                gfx950) over the launch time -> fraction of the 8 TB/s HBM3E peak.  `valu`: active VALU
                lane-operations per launch (SQ_THREAD_CYCLES_VALU) over the launch time -> fraction of the vector
                lane peak (SIMD-32: 2 cycles per wave64 instruction), with lane utilisation and issue-slot occupancy.  `algorithmic`: SURVEY §8(d)'s figure
                (52 + 40*nodes + 52*tris bytes per ray, counted on the tree actually traversed) — served almost
                entirely from L1/L2/Infinity Cache, stated as such.  `build`: compulsory bytes of the build
                (52N + 40*nodes + 4N) over its device time.  The top-level bound/achieved/peak/frac/traffic
                fields are the HBM figures (every frac <= 1).
  cpu_baseline  the UNMODIFIED reference (oracle/_ref, OpenMP, host cores) on a bounded sample of the same ray
                buffers (best of 3), with the parity check of the same run; falls back to the single-thread C port.
  end_to_end    the host entry point nrtTraverseBatch_f32 (H2D rays + kernel + D2H hits) on the primary wave: pageable
                buffers, page-locked buffers in one piece, page-locked buffers pipelined — Mrays/s and GB/s each way.
  configs       (default line only) {Mrays/s, build_ms, parity, roofline} for C2, the C4 tile and C5, each with the
                reference's answer on a bounded sample and its own hardware counters in the same run.
  next_rows     (default line only) one figure + one same-run parity sample for each SURVEY §8(f) row (bench_rows.py).
  strong_c4     (N > 1, default config) BASELINE.json's strong-scaling case beside the weak-scaled headline: the fixed
                4096x4096 frame over the 10M-triangle plane, cut into N row-interleaved tiles.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# Vector lane peak: 256 CUs x 4 SIMD-32 x 2.4 GHz max clock.  A wave64 fp32 VALU instruction occupies its SIMD for 2 cycles
# (MI355X_MICROARCH.md "Wave scheduling"; measured here with tools/ubench/valu_rate.hip -> profiles/r02a_valu_rate.txt:
# 2.8 cycles per v_fma_f32 / v_mul_f32 at 8 waves per SIMD, and twice that for the packed v_pk_* forms and v_max3/v_min3,
# i.e. packing saves issue slots, not lane-cycles).
VALU_LANES_PER_SIMD = 32
VALU_CYCLES_PER_WAVE_INST = 64 // VALU_LANES_PER_SIMD
N_XCD = 8  # GRBM_GUI_ACTIVE arrives summed over the XCDs
CLOCK_GHZ = 2.4
METRIC = "Mrays/s (primary + 1-bounce) at 1920x1080, 1M-tri mesh; BVH build ms"

CONFIGS = {
    # name: mesh generator, precision, image, scaling when N > 1
    "C2": {"mesh": "sphere", "real": "f32", "w": 1920, "h": 1080, "scaling": "weak",
           "text": "C2 stand-in: closed lumpy sphere 264x132 = 69,168 triangles fp32 (Stanford bun_zipper.ply when --mesh is given)"},
    "C3": {"mesh": ("plane", 1000, 500), "real": "f32", "w": 1920, "h": 1080, "scaling": "weak",
           "text": "C3: Plane(1000,500) = 1,000,000 triangles fp32"},
    "C4": {"mesh": ("plane", 2500, 2000), "real": "f32", "w": 4096, "h": 4096, "scaling": "strong",
           "text": "C4: Plane(2500,2000) = 10,000,000 triangles fp32, fixed 4096x4096 frame"},
    "C4tile": {"mesh": ("plane", 2500, 2000), "real": "f32", "w": 4096, "h": 4096, "scaling": "weak", "tile_of": 8,
               "text": "C4 tile: Plane(2500,2000) = 10,000,000 triangles fp32, one GPU's 4096x512 share (rows y = 0 mod 8) of the 4096x4096 frame"},
    "C5": {"mesh": ("plane", 1000, 500), "real": "f64", "w": 1920, "h": 1080, "scaling": "weak",
           "text": "C5: Plane(1000,500) = 1,000,000 triangles, fp64 build + traversal"},
}


def algorithmic_bytes(counters, real_bytes=4):
    """per ray sizeof(Ray)+sizeof(Hit) + 40 B per node visit + 52 B per triangle test (fp32)."""
    if real_bytes == 4:
        return 52 * counters["num_rays"] + 40 * counters["nodes_visited"] + 52 * counters["tris_tested"]
    return 104 * counters["num_rays"] + 64 * counters["nodes_visited"] + 88 * counters["tris_tested"]


def build_bytes(num_tris, num_nodes, real_bytes=4):
    """read the mesh once + write the tree once = N*(12 + 9*sizeof(T)) + nodes*sizeof(BVHNode) + 4N."""
    return num_tris * (12 + 9 * real_bytes) + num_nodes * (40 if real_bytes == 4 else 64) + 4 * num_tris


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


def bit_identical(h_a, m_a, h_b, m_b):
    same = np.array_equal(m_a, m_b)
    for k in ("t", "u", "v", "prim_id"):
        same = same and h_a[k].tobytes() == h_b[k].tobytes()
    return bool(same)


def host_threads():
    """Usable host parallelism: the box may expose more logical CPUs than its cgroup quota allows."""
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else max(1, int(int(q) / int(per)))
    except Exception:
        quota = None
    return quota


def reference_order_results(wl):
    """The two waves once more through the reference-order walk (tunable order4 = 0: every field bit-identical to the reference
    on the same node array), into scratch buffers, outside every timed region.  Returns (hits1, mask1, hits2, mask2)."""
    torch = wl.torch
    a = wl.accel
    was = a.GetTunable("order4")
    a.SetTunable("order4", 0)
    try:
        h1, m1 = torch.empty_like(wl.d_hits1), torch.empty_like(wl.d_mask1)
        h2, m2 = torch.empty_like(wl.d_hits2), torch.empty_like(wl.d_mask2)
        a.TraverseBatchDevice(wl.d_rays1, h1, m1)
        a.TraverseBatchDevice(wl.d_rays2, h2, m2)
        torch.cuda.synchronize()
        return (h1.cpu().numpy().view(wl.HIT), m1.cpu().numpy(), h2.cpu().numpy().view(wl.HIT)[:wl.n2], m2.cpu().numpy()[:wl.n2])
    finally:
        a.SetTunable("order4", was)


def cpu_baseline(verts, faces, rays1, rays2, gpu_nodes, gpu_indices, width, gpu_results=None, budget_s=12.0, gpu_results_ref_order=None,
                 timed_walk="default (the reference's slot order, tunable order4 = 0)"):
    """Reference (or port) timed on the host cores over a bounded sample of the same buffers; with `gpu_results` =
    (hits1, mask1, hits2, mask2) of the GPU's timed walk also the parity check of the same run;
    `gpu_results_ref_order`: the same waves through the reference-order walk (reference_order_results)."""
    from oracle import bindings as ob

    total = rays1.shape[0] + rays2.shape[0]
    if ob.reference_available():
        R = ob.Reference(verts, faces)
        # oversubscribing a cgroup quota makes OpenMP collapse, so probe a few thread counts and keep the best
        quota = host_threads()
        hw = R.max_threads()
        cands = sorted({t for t in ((quota or hw), 2 * (quota or hw), hw) if 1 <= t <= hw})
        ok, st = R.build(parallel=True, threads=cands[0])
        build_ms = st["build_secs"] * 1e3
        probe = rays1.reshape(-1, width)[::40].reshape(-1)
        best_t, rate = cands[0], 0.0
        for t in cands:
            _, _, secs = R.traverse(probe, threads=t, chunk=width)
            if probe.shape[0] / secs > rate:
                best_t, rate = t, probe.shape[0] / secs
        frac = min(1.0, budget_s / 3.0 * rate / total)  # three passes over the sample share the budget
        rows1 = max(8, int(rays1.shape[0] // width * frac))
        step = max(1, (rays1.shape[0] // width) // rows1)
        s1 = rays1.reshape(-1, width)[::step].reshape(-1)
        s2 = rays2[:: max(1, step)]
        best = 1e30
        for _ in range(3):
            rh1, rm1, t1 = R.traverse(s1, threads=best_t, chunk=width)
            rh2, rm2, t2 = R.traverse(s2, threads=best_t, chunk=width)
            best = min(best, t1 + t2)
        value = (s1.shape[0] + s2.shape[0]) / best / 1e6
        out = {
            "value": round(value, 4), "unit": "Mrays/s", "cores": int(best_t), "kind": "reference",
            "sample": "unmodified nanort.h (g++ -O3 -fopenmp, own parallel Build: %d nodes, depth %d), "
                      "every %d-th row of wave 1 (%d rays) + every %d-th wave-2 ray (%d rays), omp dynamic row loop, "
                      "best of 3; %d OpenMP threads = best of %s (host: %d logical CPUs, cgroup quota %s)" % (
                          st["num_leaf_nodes"] + st["num_branch_nodes"], st["max_tree_depth"], step,
                          s1.shape[0], step, s2.shape[0], best_t, cands, hw, quota),
            "build_ms": round(build_ms, 1),
        }
        if ob.reference_v3_available() and verts.dtype == np.float32:  # the same code with -march=x86-64-v3: SURVEY 8(d)'s stronger timing baseline
            try:
                R3 = ob.ReferenceV3(verts, faces)
                R3.build(parallel=True, threads=cands[0])
                _, _, t1 = R3.traverse(s1, threads=best_t, chunk=width)
                _, _, t2 = R3.traverse(s2, threads=best_t, chunk=width)
                out["value_march_x86_64_v3"] = round((s1.shape[0] + s2.shape[0]) / (t1 + t2) / 1e6, 4)
            except Exception as e:  # pragma: no cover
                out["value_march_x86_64_v3"] = None
                out["v3_error"] = repr(e)
        if gpu_results is not None:  # reference on ITS tree vs GPU on the GPU-built tree: equal up to exact-t ties in prim_id / u / v
            gh1, gm1, gh2, gm2 = gpu_results
            out["parity_own_trees"] = {
                "primary": parity(rh1, rm1, gh1.reshape(-1, width)[::step].reshape(-1), gm1.reshape(-1, width)[::step].reshape(-1)),
                "bounce": parity(rh2, rm2, gh2[:: max(1, step)], gm2[:: max(1, step)])}
        # same traversal code over the GPU-built node array: separates "better tree" from "faster traversal"
        if R.load_tree(gpu_nodes, gpu_indices):
            th1, tm1, t1 = R.traverse(rays1, threads=best_t, chunk=width)
            th2, tm2, t2 = R.traverse(rays2, threads=best_t, chunk=width)
            out["value_on_gpu_built_tree"] = round(total / (t1 + t2) / 1e6, 4)
            if gpu_results is not None:
                # same node array, the TIMED walk (the default walk: the reference's leaf sequence, so every count below is 0)
                gh1, gm1, gh2, gm2 = gpu_results
                out["parity_same_tree"] = {"walk": timed_walk,
                                           "primary": parity(th1, tm1, gh1, gm1), "bounce": parity(th2, tm2, gh2, gm2),
                                           "t_and_hit_flags_bit_identical": bool(np.array_equal(tm1, gm1) and np.array_equal(tm2, gm2) and
                                                                                 th1["t"].tobytes() == gh1["t"].tobytes() and th2["t"].tobytes() == gh2["t"].tobytes())}
            if gpu_results_ref_order is not None or gpu_results is not None:  # same node array, reference-order walk: every field bit-identical
                gh1, gm1, gh2, gm2 = gpu_results_ref_order if gpu_results_ref_order is not None else gpu_results
                out["parity_same_tree_bit_identical"] = bit_identical(th1, tm1, gh1, gm1) and bit_identical(th2, tm2, gh2, gm2)
                out["parity_same_tree_bit_identical_walk"] = "reference order (tunable order4 = 0), untimed launch" if gpu_results_ref_order is not None else "timed walk"
        return out
    O = ob.Oracle()
    t0 = time.time()
    nodes, idx, _ = O.build(verts, faces)
    build_ms = (time.time() - t0) * 1e3
    s1 = rays1.reshape(-1, width)[::54].reshape(-1)
    t0 = time.time()
    O.traverse(nodes, idx, verts, faces, s1)
    dt = time.time() - t0
    return {"value": round(s1.shape[0] / dt / 1e6, 4), "unit": "Mrays/s", "cores": 1, "kind": "port",
            "sample": "liboracle.so single thread, every 54th row of wave 1 (%d rays)" % s1.shape[0],
            "build_ms": round(build_ms, 1)}


# ---------------------------------------------------------------------------
# workload set-up (shared by the headline run, the `configs` extras and the PMC child)
# ---------------------------------------------------------------------------
def make_mesh(cfg, mesh_path=None):
    from nanort_amd import scenes

    if cfg["mesh"] == "sphere":
        if mesh_path:
            from nanort_amd import meshio

            v, f = meshio.load_mesh(mesh_path)
            # the C2 camera looks at (0, 5, 0) from z = 20: bring a user mesh into that frame (uniform scale to a 15-unit box)
            lo, hi = v.min(axis=0), v.max(axis=0)
            s = np.float32(15.0 / float((hi - lo).max()))
            v = ((v - (lo + hi) * np.float32(0.5)) * s + np.array([0, 5, 0], np.float32)).astype(np.float32)
            return np.ascontiguousarray(v), np.ascontiguousarray(f), "user mesh %s (%d triangles)" % (os.path.basename(mesh_path), f.shape[0])
        v, f = scenes.sphere()
        return v, f, None
    _, nx, ny = cfg["mesh"]
    v, f = scenes.plane(nx, ny)
    return v, f, None


class Workload:
    """One config on one rank: mesh, GPU-built BVH, wave 1 and wave 2 resident in HBM."""

    def __init__(self, name, rank=0, world=1, device=0, builds=5, mesh_path=None):
        import torch

        from nanort_amd import BVHAccel, TriangleMesh, scenes
        from nanort_amd.wire import hit_dtype, ray_dtype, widen_rays

        cfg = CONFIGS[name]
        self.name, self.cfg, self.rank, self.world, self.torch = name, cfg, rank, world, torch
        self.real = np.float32 if cfg["real"] == "f32" else np.float64
        self.rb = 4 if cfg["real"] == "f32" else 8
        self.RAY, self.HIT = ray_dtype(self.real), hit_dtype(self.real)
        v32, self.faces, self.mesh_note = make_mesh(cfg, mesh_path)
        self.verts32 = v32
        self.verts = v32 if self.real == np.float32 else v32.astype(np.float64)
        mesh = TriangleMesh(self.verts, self.faces)
        self.accel = BVHAccel(self.real, device=device)
        self.build_ms = []
        for _ in range(max(1, builds)):
            assert self.accel.Build(mesh.num_faces, mesh)
            self.build_ms.append(self.accel.LastBuildMs())
        self.stats = self.accel.GetStatistics()
        self.num_nodes = int(self.stats["num_leaf_nodes"] + self.stats["num_branch_nodes"])
        # image rows of this rank: interleaved; weak scaling grows the image, strong scaling cuts a fixed one
        W, H = cfg["w"], cfg["h"]
        self.width = W
        if "tile_of" in cfg:  # one GPU's share of the C4 frame
            t = cfg["tile_of"]
            self.h_glob, y0, step, rows = H, rank, t * world, H // (t * world)
        elif cfg["scaling"] == "strong":
            if H % world:
                raise SystemExit("--config %s: %d rows do not split into %d equal tiles" % (name, H, world))
            self.h_glob, y0, step, rows = H, rank, world, H // world
        else:
            self.h_glob, y0, step, rows = H * world, rank, world, H
        self.rows = rows
        rays1_f32 = scenes.camera_rays_rows(W, self.h_glob, y0, step, rows)
        self.rays1 = rays1_f32 if self.real == np.float32 else widen_rays(rays1_f32)
        self.n1 = self.rays1.shape[0]
        cuda = torch.device("cuda", device)
        self.d_rays1 = torch.from_numpy(self.rays1.view(np.uint8)).to(cuda)
        self.d_hits1 = torch.empty(self.n1 * self.HIT.itemsize, dtype=torch.uint8, device=cuda)
        self.d_mask1 = torch.empty(self.n1, dtype=torch.uint8, device=cuda)
        self.accel.TraverseBatchDevice(self.d_rays1, self.d_hits1, self.d_mask1)
        torch.cuda.synchronize()
        self.hits1 = self.d_hits1.cpu().numpy().view(self.HIT)
        self.mask1 = self.d_mask1.cpu().numpy()
        # wave 2 is generated on the host from the wave-1 hits, in fp32 as SURVEY 8(d) defines it (widened for C5);
        # pixel index of ray i in the global image: row (y0 + step * (i // W)), column i % W
        from nanort_amd.wire import HIT_F32

        h32 = self.hits1
        if self.real != np.float32:
            h32 = np.zeros(self.n1, dtype=HIT_F32)
            for k in ("t", "u", "v"):
                h32[k] = self.hits1[k].astype(np.float32)
            h32["prim_id"] = self.hits1["prim_id"]
        self.kind2 = "bounce"
        rays2_f32 = scenes.secondary_rays("bounce", v32, self.faces, rays1_f32, h32, self.mask1, pixel_base=rank * self.n1)
        self.rays1_f32, self.hits1_f32 = rays1_f32, h32
        self.rays2 = rays2_f32 if self.real == np.float32 else widen_rays(rays2_f32)
        self.n2 = self.rays2.shape[0]
        self.d_rays2 = torch.from_numpy(self.rays2.view(np.uint8)).to(cuda)
        # wave-2 records are padded to n1 so that every rank's gather slice has the same size
        self.d_hits2 = torch.empty(max(1, self.n1) * self.HIT.itemsize, dtype=torch.uint8, device=cuda)
        self.d_mask2 = torch.empty(max(1, self.n1), dtype=torch.uint8, device=cuda)

    def counters(self):
        c1 = self.accel.TraverseCountDevice(self.d_rays1)
        c2 = self.accel.TraverseCountDevice(self.d_rays2) if self.n2 else {"num_rays": 0, "nodes_visited": 0, "tris_tested": 0}
        return c1, c2

    def results(self):
        self.torch.cuda.synchronize()
        return (self.d_hits1.cpu().numpy().view(self.HIT), self.d_mask1.cpu().numpy(),
                self.d_hits2.cpu().numpy().view(self.HIT)[:self.n2], self.d_mask2.cpu().numpy()[:self.n2])

    def describe(self):
        return "%s; %dx%d objrender-camera primaries + 1 cosine bounce per hit (%d + %d rays per GPU per step)" % (
            self.mesh_note or self.cfg["text"], self.width, self.rows, self.n1, self.n2)


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


def opt_in_distance_order(wl, steps):
    """The same K steps through the OPT-IN walk (tunable order4 = 1: a record's four slots entered by entry distance), into
    scratch buffers, with its records compared with the timed default walk's on the same tree over BOTH whole waves.  Never
    `value`: its parity class is the contract's, not bit identity."""
    torch, a = wl.torch, wl.accel
    h1, m1 = torch.empty_like(wl.d_hits1), torch.empty_like(wl.d_mask1)
    h2, m2 = torch.empty_like(wl.d_hits2), torch.empty_like(wl.d_mask2)
    ref = wl.results()
    a.SetTunable("order4", 1)
    try:
        for _ in range(2):
            a.TraverseBatchDevice(wl.d_rays1, h1, m1)
            a.TraverseBatchDevice(wl.d_rays2, h2, m2)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(steps):
            a.TraverseBatchDevice(wl.d_rays1, h1, m1)
            a.TraverseBatchDevice(wl.d_rays2, h2, m2)
        ev[1].record()
        torch.cuda.synchronize()
        ms = float(ev[0].elapsed_time(ev[1])) / steps
        kernel = a.LastKernelName()
    finally:
        a.SetTunable("order4", 0)
    got = (h1.cpu().numpy().view(wl.HIT), m1.cpu().numpy(), h2.cpu().numpy().view(wl.HIT)[:wl.n2], m2.cpu().numpy()[:wl.n2])
    return {"tunable": "order4 = 1", "kernel": kernel, "value": round((wl.n1 + wl.n2) / ms / 1e3, 1), "unit": "Mrays/s", "ms_per_step": round(ms, 4),
            "vs_default_walk_same_tree": {"primary": parity(ref[0], ref[1], got[0], got[1]), "bounce": parity(ref[2], ref[3], got[2], got[3])},
            "note": "opt-in: another leaf sequence than the reference's — among primitives at exactly the same t another one may be named "
                    "(prim_id_mismatches, all of them at exact-t ties when within_tolerance is true)"}


def multi_batch(accel, torch, wave1, wave2, steps, rays_per_step):
    """The same K steps through nrtTraverseBatchesDevice — ONE stream, ONE persistent launch per step over both waves of the
    frame (one launch tail instead of two), and over the waves of two frames (four batches per launch) — with the records
    compared with the separate launches'.  Same work as the timed region; reported beside it, never as `value` (a step of
    the headline is two single-batch launches)."""
    (r1, h1, m1), (r2, h2, m2) = wave1, wave2
    accel.TraverseBatchDevice(r1, h1, m1)
    accel.TraverseBatchDevice(r2, h2, m2)
    torch.cuda.synchronize()
    ref1, ref2 = h1.clone(), h2.clone()
    h1.zero_()
    h2.zero_()
    accel.TraverseBatchesDevice([(r1, h1, m1), (r2, h2, m2)])
    torch.cuda.synchronize()
    same = bool(torch.equal(ref1, h1) and torch.equal(ref2, h2))
    h1b, m1b, h2b, m2b = torch.empty_like(h1), torch.empty_like(m1), torch.empty_like(h2), torch.empty_like(m2)

    def timed(batches, frames):
        for _ in range(2):
            accel.TraverseBatchesDevice(batches)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(max(1, steps // frames)):
            accel.TraverseBatchesDevice(batches)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (max(1, steps // frames) * frames)

    one = timed([(r1, h1, m1), (r2, h2, m2)], 1)
    two = timed([(r1, h1, m1), (r2, h2, m2), (r1, h1b, m1b), (r2, h2b, m2b)], 2)
    return {"entry_point": "nrtTraverseBatchesDevice_f32 (one stream)", "value": round(rays_per_step / one / 1e6, 1), "unit": "Mrays/s",
            "ms_per_step": round(one * 1e3, 4), "records_identical_to_separate_launches": same,
            "two_frames_per_launch": {"value": round(rays_per_step / two / 1e6, 1), "ms_per_step": round(two * 1e3, 4)}}


# ---------------------------------------------------------------------------
# hardware counters, collected in the same invocation (outside the timed region)
# ---------------------------------------------------------------------------
# Counter passes: one rocprofv3 --pmc invocation each (kernel trace only, as MI355X_MICROARCH.md prescribes).  The TCC block has
# four counter slots (FETCH_SIZE takes 3, WRITE_SIZE 2), the TCP and SQ blocks have their own: three passes carry everything.
# Each entry: (tag, counters, fallback passes tried when the combined pass fails or returns no rows).
PMC_PASSES = [
    ("fetch_tcp", "FETCH_SIZE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum",
     [("fetch", "FETCH_SIZE"), ("tcp", "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum")]),
    ("write_tcc", "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum", [("write", "WRITE_SIZE"), ("tcc", "TCC_HIT_sum TCC_MISS_sum")]),
    ("sq", "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE", []),
]
# What the vector L1 (TCP) sustains in tag look-ups per second when every lane of every wave fetches scattered 16-byte
# pieces, measured with tools/ubench/node_fetch.hip under the same counter (3145 M look-ups in 3.59 ms, table resident in
# L2; 896 G/s when resident in L1): profiles/r02g_node_fetch_ubench.txt, r02l_tcp_counter_calibration.txt.
L1_PEAK_GACC_S = 876.0


def pmc_child(args):
    """The sub-run the counter passes profile: for every config named, the set-up (one primary launch) and then
    (warmup + steps) x (primary, bounce) launches of THIS rank's share of the workload — nothing else."""
    import torch

    done = []
    for name in args.pmc_configs.split(","):
        wl = Workload(name, rank=args.pmc_rank, world=args.pmc_world, builds=1, mesh_path=args.mesh if name == "C2" else None)
        for _ in range(args.warmup + args.steps):
            wl.accel.TraverseBatchDevice(wl.d_rays1, wl.d_hits1, wl.d_mask1)
            wl.accel.TraverseBatchDevice(wl.d_rays2, wl.d_hits2, wl.d_mask2)
        torch.cuda.synchronize()
        done.append({"name": name, "kernel": wl.accel.LastKernelName(), "n1": wl.n1, "n2": wl.n2})
        del wl
        torch.cuda.empty_cache()
    print(json.dumps({"pmc_child": True, "configs": done}), flush=True)


def _kernel_key(name):
    return name.replace("void ", "").split("(")[0].replace(" ", "")


def _pmc_pass(exe, tag, counters, child_args, out_root, env):
    """One rocprofv3 invocation.  Returns (counter rows, kernel-trace rows, the child's config list, error or None)."""
    out_dir = os.path.join(out_root, tag)
    cmd = [exe, "--kernel-trace", "--pmc"] + counters.split() + ["--output-format", "csv", "-d", out_dir, "-o", "p", "--",
                                                                   sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child"] + child_args
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=600, cwd="/tmp")
    except Exception as e:  # pragma: no cover
        return [], [], None, "%s: %r" % (tag, e)
    if r.returncode != 0:
        return [], [], None, "%s: rc %d: %s" % (tag, r.returncode, r.stdout[-300:])
    child = None
    for line in r.stdout.splitlines():
        if line.startswith("{") and "pmc_child" in line:
            child = json.loads(line)["configs"]
    rows, trace = [], []
    for path in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(path)))
    for path in glob.glob(os.path.join(out_dir, "**", "*kernel_trace.csv"), recursive=True):
        trace += list(csv.DictReader(open(path)))
    if not rows or child is None:
        return [], [], child, "%s: no counter rows" % tag
    return rows, trace, child, None


def pmc_collect(configs, mesh_path=None, rank=0, world=1, keep_dir=None, warmup=1, steps=3):
    """Run the counter passes over `bench.py --pmc-child` (ONE sub-run per pass traces every config named, this rank's
    share of it) and return {config: {"primary": {counter: per-launch mean}, "bounce": {...}, "profiled_us": {...}}}
    plus an error string (or None).  Launches are attributed by kernel name and dispatch order: per config one set-up
    launch (primary), then (primary, bounce) pairs."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    tmp = keep_dir or tempfile.mkdtemp(prefix="nrt_pmc_", dir="/tmp")
    os.makedirs(tmp, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    child_args = ["--pmc-configs", ",".join(configs), "--pmc-rank", str(rank), "--pmc-world", str(world), "--steps", str(steps), "--warmup", str(warmup)]
    if mesh_path:
        child_args += ["--mesh", mesh_path]
    per_launch = 1 + 2 * (warmup + steps)
    out = {c: {"primary": {}, "bounce": {}, "profiled_us": {"primary": None, "bounce": None}} for c in configs}
    errors = []

    def absorb(rows, trace, child, with_durations):
        by_kernel = {}
        for c in child:  # configs in launch order, grouped by the kernel variant they ran
            by_kernel.setdefault(_kernel_key(c["kernel"]), []).append(c["name"])
        for key, names in by_kernel.items():
            mine = [x for x in rows if _kernel_key(x.get("Kernel_Name", "")) == key]
            ids = sorted({int(x["Dispatch_Id"]) for x in mine})
            if len(ids) != per_launch * len(names):
                errors.append("%s: %d dispatches of %s, expected %d" % (",".join(names), len(ids), key, per_launch * len(names)))
                continue
            where = {d: (names[k // per_launch], k % per_launch) for k, d in enumerate(ids)}
            acc = {}
            for x in mine:
                name, k = where[int(x["Dispatch_Id"])]
                if k == 0:
                    continue  # the set-up launch
                wave = "primary" if k % 2 == 1 else "bounce"
                acc[(name, wave, x["Counter_Name"], k)] = acc.get((name, wave, x["Counter_Name"], k), 0.0) + float(x["Counter_Value"])
            lists = {}
            for (name, wave, cname, _k), v in acc.items():
                lists.setdefault((name, wave, cname), []).append(v)
            for (name, wave, cname), v in lists.items():
                out[name][wave][cname] = float(np.mean(v))
            if with_durations:
                tr = [x for x in trace if _kernel_key(x.get("Kernel_Name", "")) == key]
                tr.sort(key=lambda x: int(x["Start_Timestamp"]))
                if len(tr) == per_launch * len(names):
                    for k, x in enumerate(tr):
                        name, kk = names[k // per_launch], k % per_launch
                        if kk:
                            out[name].setdefault("_durs", {}).setdefault("primary" if kk % 2 == 1 else "bounce", []).append(
                                (int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) * 1e-3)

    for tag, counters, fallback in PMC_PASSES:
        rows, trace, child, err = _pmc_pass(exe, tag, counters, child_args, tmp, env)
        if err and fallback:  # the combined pass was refused: the blocks one by one
            errors.append(err + " (retried as %s)" % "+".join(t for t, _ in fallback))
            for ftag, fcounters in fallback:
                rows, trace, child, ferr = _pmc_pass(exe, ftag, fcounters, child_args, tmp, env)
                if ferr:
                    errors.append(ferr)
                else:
                    absorb(rows, trace, child, False)
            continue
        if err:
            errors.append(err)
            continue
        absorb(rows, trace, child, tag == "sq")
    for c in configs:
        d = out[c].pop("_durs", {})
        out[c]["profiled_us"] = {w: (float(np.mean(d[w])) if d.get(w) else None) for w in ("primary", "bounce")}
    if not keep_dir:
        shutil.rmtree(tmp, ignore_errors=True)
    return out, ("; ".join(errors) if errors else None)


def roofline_from_counters(pmc, k_ms, n_cus, launch_ms=None):
    """HBM, VALU and L1 rooflines of the primary / bounce launches from the in-run counter means (per launch).  The
    fractions divide by the launch times `k_ms` (per wave) — or, for the top-level HBM figure of the headline, by
    `launch_ms`, the average launch of the timed region itself."""
    simds = n_cus * 4
    lane_peak = simds * VALU_LANES_PER_SIMD * CLOCK_GHZ * 1e9  # lane-operations per second
    res = {"hbm": None, "valu": None, "l1": None}
    waves = ("primary", "bounce")
    if all("FETCH_SIZE" in pmc[w] and "WRITE_SIZE" in pmc[w] for w in waves):
        # rocprofv3 reports both in KiB; gfx950: FETCH_SIZE counts 128-B read requests as 64 B -> x2 (MI355X_MICROARCH.md §HBM)
        b = {w: pmc[w]["FETCH_SIZE"] * 1024.0 * 2.0 + pmc[w]["WRITE_SIZE"] * 1024.0 for w in waves}
        tot_ms = sum(k_ms[w] for w in waves) if launch_ms is None else 2.0 * launch_ms
        gbs = sum(b.values()) / (tot_ms * 1e-3) / 1e9
        res["hbm"] = {"bytes_per_launch": int(sum(b.values()) / 2), "achieved_GBs": round(gbs, 1), "peak_GBs": HBM_PEAK_GBS,
                      "frac": round(gbs / HBM_PEAK_GBS, 4),
                      "time_base": "per-wave kernel times" if launch_ms is None else "average launch of the timed region",
                      "per_wave_bytes": {w: int(b[w]) for w in waves},
                      "per_wave_frac": {w: round(b[w] / (k_ms[w] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) for w in waves},
                      "formula": "FETCH_SIZE KiB x 1024 x 2 (gfx950 wide-read correction) + WRITE_SIZE KiB x 1024"}
        if all("TCC_HIT_sum" in pmc[w] for w in waves):
            h = sum(pmc[w]["TCC_HIT_sum"] for w in waves)
            m = sum(pmc[w]["TCC_MISS_sum"] for w in waves)
            res["hbm"]["l2_hit_rate"] = round(h / max(1.0, h + m), 4)
    if all("TCP_TOTAL_CACHE_ACCESSES_sum" in pmc[w] for w in waves):
        # one look-up per active lane for scattered accesses, one per quad of lanes reading one 64-byte line (calibrated on the
        # micro-benchmark): the address / tag path of the vector L1, which the node and triangle fetches of this kernel load
        per = {}
        for w in waves:
            acc = pmc[w]["TCP_TOTAL_CACHE_ACCESSES_sum"]
            per[w] = {"lookups": int(acc), "frac": round(acc / (k_ms[w] * 1e-3) / 1e9 / L1_PEAK_GACC_S, 4)}
            if "TCP_TCC_READ_REQ_sum" in pmc[w]:
                per[w]["requests_to_l2_per_lookup"] = round(pmc[w]["TCP_TCC_READ_REQ_sum"] / max(1.0, acc), 4)
        tot = sum(per[w]["lookups"] for w in waves)
        tot_s = sum(k_ms[w] for w in waves) * 1e-3
        res["l1"] = {"lookups_per_launch": int(tot / 2), "achieved_Glookups_s": round(tot / tot_s / 1e9, 1), "peak_Glookups_s": L1_PEAK_GACC_S,
                     "peak_definition": "measured: tools/ubench/node_fetch.hip, every lane fetching scattered 16-byte pieces (L2-resident table)",
                     "frac": round(tot / tot_s / 1e9 / L1_PEAK_GACC_S, 4), "per_wave": per}
    if all("SQ_THREAD_CYCLES_VALU" in pmc[w] and "SQ_INSTS_VALU" in pmc[w] for w in waves):
        per = {}
        for w in waves:
            lane_ops, insts = pmc[w]["SQ_THREAD_CYCLES_VALU"], pmc[w]["SQ_INSTS_VALU"]
            secs = k_ms[w] * 1e-3
            per[w] = {"lane_ops": int(lane_ops), "wave_insts": int(insts), "frac": round(lane_ops / secs / lane_peak, 4),
                      "lane_util": round(lane_ops / (64.0 * insts), 4),
                      # a wave64 instruction holds its SIMD-32 for 2 cycles (packed / 3-input forms longer: a lower bound)
                      "issue_busy": round(insts * VALU_CYCLES_PER_WAVE_INST / (simds * secs * CLOCK_GHZ * 1e9), 4)}
            if "SQ_WAIT_ANY" in pmc[w] and pmc[w].get("SQ_WAVE_CYCLES"):
                per[w]["wait_frac_of_wave_cycles"] = round(pmc[w]["SQ_WAIT_ANY"] / pmc[w]["SQ_WAVE_CYCLES"], 4)
            if "SQ_LDS_BANK_CONFLICT" in pmc[w]:
                per[w]["lds_bank_conflict_cycles"] = int(pmc[w]["SQ_LDS_BANK_CONFLICT"])
            if pmc[w].get("GRBM_GUI_ACTIVE") and pmc.get("profiled_us", {}).get(w):
                per[w]["effective_clock_GHz_under_profiler"] = round(pmc[w]["GRBM_GUI_ACTIVE"] / N_XCD / (pmc["profiled_us"][w] * 1e3), 3)
        tot_ops = sum(per[w]["lane_ops"] for w in waves)
        tot_s = sum(k_ms[w] for w in waves) * 1e-3
        res["valu"] = {"lane_ops_per_launch": int(tot_ops / 2), "achieved_Tlaneops": round(tot_ops / tot_s / 1e12, 3),
                       "peak_Tlaneops": round(lane_peak / 1e12, 2),
                       "peak_definition": "%d CUs x 4 SIMDs x %d lanes x %.1f GHz (max clock)" % (n_cus, VALU_LANES_PER_SIMD, CLOCK_GHZ),
                       "frac": round(tot_ops / tot_s / lane_peak, 4),
                       "lane_util": round(tot_ops / (64.0 * sum(per[w]["wave_insts"] for w in waves)), 4),
                       "issue_busy": round(sum(per[w]["wave_insts"] for w in waves) * VALU_CYCLES_PER_WAVE_INST / (simds * tot_s * CLOCK_GHZ * 1e9), 4),
                       "per_wave": per}
    return res


def compact_roofline(r, per_wave_counts):
    """The per-config form of the counters: per wave {ms, hbm / valu / l1 fractions, lane utilisation, waiting share, L2 hit
    rate is per config} — every number recomputable from the raw rows kept under --pmc-dir."""
    out = {"waves": {}}
    for w in ("primary", "bounce"):
        e = dict(per_wave_counts.get(w, {}))
        if r.get("hbm"):
            e["hbm_bytes"] = r["hbm"]["per_wave_bytes"][w]
            e["hbm_frac"] = r["hbm"]["per_wave_frac"][w]
        if r.get("valu"):
            pw = r["valu"]["per_wave"][w]
            e.update({"valu_frac": pw["frac"], "lane_util": pw["lane_util"], "issue_busy": pw["issue_busy"]})
            if "wait_frac_of_wave_cycles" in pw:
                e["wait"] = pw["wait_frac_of_wave_cycles"]
        if r.get("l1"):
            e["l1_frac"] = r["l1"]["per_wave"][w]["frac"]
            e["l1_requests_to_l2_per_lookup"] = r["l1"]["per_wave"][w].get("requests_to_l2_per_lookup")
        out["waves"][w] = e
    if r.get("hbm"):
        out["hbm"] = {k: r["hbm"][k] for k in ("bytes_per_launch", "achieved_GBs", "frac") if k in r["hbm"]}
        if "l2_hit_rate" in r["hbm"]:
            out["l2_hit_rate"] = r["hbm"]["l2_hit_rate"]
    if r.get("valu"):
        out["valu"] = {k: r["valu"][k] for k in ("achieved_Tlaneops", "frac", "lane_util", "issue_busy")}
    if r.get("l1"):
        out["l1"] = {k: r["l1"][k] for k in ("achieved_Glookups_s", "frac")}
    fr = {k: out[k]["frac"] for k in ("hbm", "valu", "l1") if k in out}
    if fr:
        out["most_loaded"] = max(fr, key=fr.get)
    return out


def end_to_end(wl, reps=5):
    """the host entry point end to end — H2D rays + kernel + D2H hits and flags — on the primary wave
    (nrtTraverseBatch_*): pageable caller buffers, page-locked buffers in one piece, page-locked buffers pipelined in
    512K-ray pieces (the library's default for page-locked buffers).  Never `value`."""
    import torch

    a = wl.accel
    rays = wl.rays1
    n = rays.shape[0]
    up, down = rays.nbytes, n * wl.HIT.itemsize + n
    fn = getattr(a._L, "nrtTraverseBatch_" + ("f32" if wl.rb == 4 else "f64"))

    def timed(call):
        for _ in range(2):
            call()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            call()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts))

    def entry(secs):
        return {"ms": round(secs * 1e3, 3), "Mrays_s": round(n / secs / 1e6, 1), "h2d_GBs": round(up / secs / 1e9, 2), "d2h_GBs": round(down / secs / 1e9, 2)}

    res = {}
    t = timed(lambda: a.TraverseBatch(rays))
    res["pageable"] = entry(t)
    pr = torch.empty(rays.nbytes, dtype=torch.uint8, pin_memory=True)
    pr.numpy()[:] = rays.view(np.uint8)
    ph = torch.empty(n * wl.HIT.itemsize, dtype=torch.uint8, pin_memory=True)
    pm = torch.empty(n, dtype=torch.uint8, pin_memory=True)

    def call():
        st = fn(a._h, pr.data_ptr(), n, None, ph.data_ptr(), pm.data_ptr())
        assert st == 0

    a.SetTunable("host_pipeline", 0)
    res["page_locked"] = entry(timed(call))
    a.SetTunable("host_pipeline", 1)
    res["page_locked_pipelined"] = entry(timed(call))
    res["records_identical_to_the_device_path"] = bool(ph.numpy().tobytes() == wl.hits1.tobytes() and pm.numpy().tobytes() == wl.mask1.tobytes())
    return {"workload": "%d primary rays: %.1f MB of rays up, %.1f MB of records and flags down per call" % (n, up / 1e6, down / 1e6),
            "unit": "Mrays/s end to end (never `value`)", **res}


# ---------------------------------------------------------------------------
# untimed extras: the other single-GPU configs of BASELINE.json
# ---------------------------------------------------------------------------
def per_wave_counts(wl, c1, c2, ms1, ms2):
    """Work per ray of the two waves (the counting pass of the literal kernel: identical to the CPU oracle's counts)."""
    return {"primary": {"ms": round(ms1, 4), "rays": wl.n1, "nodes_per_ray": round(c1["nodes_visited"] / max(1, wl.n1), 2),
                        "tris_per_ray": round(c1["tris_tested"] / max(1, wl.n1), 2), "algorithmic_bytes": int(algorithmic_bytes(c1, wl.rb))},
            "bounce": {"ms": round(ms2, 4), "rays": wl.n2, "nodes_per_ray": round(c2["nodes_visited"] / max(1, wl.n2), 2),
                       "tris_per_ray": round(c2["tris_tested"] / max(1, wl.n2), 2), "algorithmic_bytes": int(algorithmic_bytes(c2, wl.rb))}}


def measure_config(name, mesh_path=None, reps=5, parity_rays=200_000):
    """{Mrays/s, build_ms, parity} for one config on GPU 0, with the reference's answer on a bounded sample.  The counters of
    the config are attached by the caller (one profiled sub-run serves all configs)."""
    import torch

    wl = Workload(name, builds=3, mesh_path=mesh_path)
    a = wl.accel
    t1, t2 = [], []
    for _ in range(reps):
        a.TraverseBatchDevice(wl.d_rays1, wl.d_hits1, wl.d_mask1)
        t1.append(a.LastTraverseMs())
        a.TraverseBatchDevice(wl.d_rays2, wl.d_hits2, wl.d_mask2)
        t2.append(a.LastTraverseMs())
    ms1, ms2 = float(np.median(t1)), float(np.median(t2))
    # the figure of merit as the headline measures it: launches back to back, one event pair around all of them
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(reps):
        a.TraverseBatchDevice(wl.d_rays1, wl.d_hits1, wl.d_mask1)
        a.TraverseBatchDevice(wl.d_rays2, wl.d_hits2, wl.d_mask2)
    ev[1].record()
    torch.cuda.synchronize()
    step_ms = float(ev[0].elapsed_time(ev[1])) / reps
    timed_kernel = a.LastKernelName()  # (before the counting pass: that one launches the literal kernel)
    c1, c2 = wl.counters()
    out = {"workload": wl.describe(), "dtype": wl.cfg["real"], "value": round((wl.n1 + wl.n2) / step_ms / 1e3, 1), "unit": "Mrays/s",
           "ms_per_step": round(step_ms, 4),
           "primary_ms": round(ms1, 4), "bounce_ms": round(ms2, 4), "primary_Mrays_s": round(wl.n1 / ms1 / 1e3, 1),
           "build_ms": round(float(np.median(wl.build_ms)), 4), "kernel": timed_kernel,
           "bvh": {"nodes": wl.num_nodes, "max_depth": int(wl.stats["max_tree_depth"])},
           "roofline_build": {"bytes": int(build_bytes(wl.faces.shape[0], wl.num_nodes, wl.rb)),
                              "frac": round(build_bytes(wl.faces.shape[0], wl.num_nodes, wl.rb) / (float(np.median(wl.build_ms)) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
           "_k_ms": {"primary": ms1, "bounce": ms2},
           "_counts": per_wave_counts(wl, c1, c2, ms1, ms2)}
    try:
        from oracle import bindings as ob

        gh1, gm1, gh2, gm2 = wl.results()
        ro = reference_order_results(wl) if (wl.real == np.float32 and a.GetTunable("order4")) else (gh1, gm1, gh2, gm2)
        nodes, indices = a.GetTree()
        if ob.reference_available():
            R = ob.Reference(wl.verts, wl.faces)
            threads = host_threads() or 0
            # (a) the reference's Traverse over the GPU-built node array: every field bit-identical
            s1 = max(1, wl.n1 // parity_rays)
            s2 = max(1, wl.n2 // parity_rays)
            p = {"kind": "reference"}
            if R.load_tree(nodes, indices):
                th1, tm1, _ = R.traverse(wl.rays1[::s1], threads=threads, chunk=4096)
                th2, tm2, _ = R.traverse(wl.rays2[::s2], threads=threads, chunk=4096)
                # the reference-order walk (order4 = 0; fp64 trees always): every field; the timed default walk: t and flags, ties in prim_id
                p["same_tree_bit_identical"] = bit_identical(th1, tm1, ro[0][::s1], ro[1][::s1]) and bit_identical(th2, tm2, ro[2][::s2], ro[3][::s2])
                p["same_tree_timed_walk"] = {"primary": parity(th1, tm1, gh1[::s1], gm1[::s1]), "bounce": parity(th2, tm2, gh2[::s2], gm2[::s2])}
                p["same_tree_rays"] = int(th1.shape[0] + th2.shape[0])
            # (b) the reference on its own tree (its own Build): equal up to exact-t ties
            ok, st = R.build(parallel=True, threads=threads)
            s1b = max(1, wl.n1 // (parity_rays // 10))
            rh1, rm1, _ = R.traverse(wl.rays1[::s1b], threads=threads, chunk=4096)
            p["own_trees"] = parity(rh1, rm1, gh1[::s1b], gm1[::s1b])
            p["reference_tree"] = {"nodes": int(st["num_leaf_nodes"] + st["num_branch_nodes"]), "max_depth": int(st["max_tree_depth"]),
                                   "build_ms": round(st["build_secs"] * 1e3, 1)}
        else:
            O = ob.Oracle()
            s1 = max(1, wl.n1 // 20000)
            oh, om = O.traverse(nodes, indices, wl.verts, wl.faces, wl.rays1[::s1])
            p = {"kind": "port", "same_tree_bit_identical": bit_identical(oh, om, ro[0][::s1], ro[1][::s1]), "same_tree_rays": int(oh.shape[0])}
        out["parity"] = p
    except Exception as e:  # pragma: no cover
        out["parity"] = {"error": repr(e)}
    del wl
    torch.cuda.empty_cache()
    return out


# ---------------------------------------------------------------------------
def dry_run(args):
    """`bench.py --gpus N --dry-run`: everything a multi-GPU run can trip over BEFORE anything is launched — device count, the
    environment the ranks need, the RCCL backend, the library and its symbols, the rendezvous port, tile divisibility and the
    per-rank / root buffer sizes against the device's memory.  Prints one JSON object; exit code 0 when every check passes."""
    import socket

    checks = []

    def check(name, ok, detail):
        checks.append({"check": name, "ok": bool(ok), "detail": detail})

    n = args.gpus
    cfg = CONFIGS[args.config]
    try:
        import torch
        import torch.distributed as td

        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        check("devices", have >= n, "%d GPU(s) visible, %d requested" % (have, n))
        check("rccl_backend", td.is_available() and td.is_nccl_available(), "torch.distributed nccl (== RCCL on ROCm) available: %s" % (td.is_available() and td.is_nccl_available()))
        mem = [torch.cuda.get_device_properties(i).total_memory for i in range(min(have, n))]
    except Exception as e:  # pragma: no cover
        check("torch", False, repr(e))
        have, mem = 0, []
    ipc = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
    check("hsa_ipc_mode", n == 1 or ipc in (None, "0"), "HSA_ENABLE_IPC_MODE_LEGACY=%r (bench.py exports 0 for the ranks it spawns; anything else breaks RCCL's "
          "dmabuf IPC on this driver)" % ipc)
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    try:
        socket.getaddrinfo(addr, None)
        check("master_addr", True, "%s resolves" % addr)
    except Exception as e:
        check("master_addr", False, "%s does not resolve: %r (use 127.0.0.1)" % (addr, e))
    port = int(os.environ.get("MASTER_PORT", 29500 + (os.getpid() % 2000)))
    try:
        sk = socket.socket()
        sk.bind(("127.0.0.1", port))
        sk.close()
        check("master_port", True, "port %d is free" % port)
    except Exception as e:
        check("master_port", "MASTER_PORT" in os.environ and "WORLD_SIZE" in os.environ, "port %d: %r" % (port, e))
    try:
        from nanort_amd import capi

        L = capi.lib()
        missing = [f for f in ("nrtCreate", "nrtBuild_f32", "nrtTraverseBatchDevice_f32", "nrtTraverseBatchesDevice_f32") if not hasattr(L, f)]
        check("library", not missing, "%s loads%s" % (capi.LIB_PATH, (", missing " + ",".join(missing)) if missing else ""))
    except Exception as e:
        check("library", False, repr(e))
    W, H = cfg["w"], cfg["h"]
    strong = cfg["scaling"] == "strong" and "tile_of" not in cfg
    if strong:
        check("tiles", H % n == 0, "%d rows over %d ranks: %s" % (H, n, "equal row-interleaved tiles of %d rows" % (H // max(1, n)) if H % n == 0 else "do not split equally"))
    rows = H // (cfg.get("tile_of", 1) * n) if "tile_of" in cfg else (H // n if strong else H)
    rb = 4 if cfg["real"] == "f32" else 8
    ray_b, hit_b = (36, 16) if rb == 4 else (72, 32)
    n1 = W * rows
    if cfg["mesh"] == "sphere":
        tris = 69696
    else:
        tris = 2 * cfg["mesh"][1] * cfg["mesh"][2]
    # per rank: two waves of rays, two double-buffered record + flag sets, the tree and its private layouts, the build workspace
    tree_b = tris * (12 + 9 * rb // 3 + 4) + 2 * tris * (40 if rb == 4 else 64) + tris * (40 if rb == 4 else 80) + 2 * tris * (64 + 128 if rb == 4 else 112) + tris * 170
    per_rank = 2 * n1 * ray_b + 4 * n1 * (hit_b + 1) + tree_b
    root_extra = 4 * n * n1 * hit_b  # the root's two double-buffered gather targets per wave
    need = per_rank + root_extra
    cap = min(mem) if mem else 288 * 10**9
    check("memory", need < 0.8 * cap, "rank 0 needs about %.2f GB (%.2f GB per rank + %.2f GB of gather buffers at the root) of %.0f GB%s" % (
        need / 1e9, per_rank / 1e9, root_extra / 1e9, cap / 1e9, "" if mem else " (nominal: no device visible)"))
    check("gather", True, "%d x %d B = %.1f MB of hit records per wave reach the root over its direct xGMI links" % (n * n1, hit_b, n * n1 * hit_b / 1e6))
    ok = all(c["ok"] for c in checks)
    print(json.dumps({"dry_run": True, "ok": ok, "n_gpus": n, "config": args.config, "rays_per_rank_per_wave": n1, "checks": checks}), flush=True)
    return 0 if ok else 2


def self_spawn(args, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU, RCCL) and relay their output."""
    import torch

    shared = os.environ.get("NRT_BENCH_TEST_SHARED_GPU") == "1"
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus and not shared:
        sys.stderr.write("bench.py: --gpus %d requested but this box exposes %d GPU(s); refusing to report a %d-GPU line "
                         "from fewer devices\n" % (args.gpus, have, args.gpus))
        return 2
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env, cwd=ROOT)


class Timed:
    """The timed region of one workload on this rank's GPU: W warm-up steps, a short pass with an event pair around every
    launch (per-wave kernel times, outside the timed region), then exactly K steps bracketed by barrier + synchronize —
    max over ranks.  One step = wave 1 + wave 2 (+ for N > 1 the asynchronous, double-buffered gather of both waves' hit
    records to rank 0, every gather completing inside the region)."""

    def __init__(self, wl, steps, warmup, world, rank, dist, shared, check_gather=False):
        import torch

        from nanort_amd import dist as nd

        accel, n1, HIT = wl.accel, wl.n1, wl.HIT
        comm_dev = "cpu" if shared else "cuda"
        # the exchange runs whenever a process group exists: N > 1, or N = 1 under --force-dist (the RCCL path on a one-GPU box)
        use_dist = dist is not None
        nbuf = 2 if use_dist else 1
        hit_bufs1 = [wl.d_hits1] + [torch.empty_like(wl.d_hits1) for _ in range(nbuf - 1)]
        hit_bufs2 = [wl.d_hits2] + [torch.empty_like(wl.d_hits2) for _ in range(nbuf - 1)]
        gathered1 = gathered2 = [None, None]
        if use_dist and rank == 0:
            gathered1 = [torch.empty(world * n1 * HIT.itemsize, dtype=torch.uint8, device=comm_dev) for _ in range(2)]
            gathered2 = [torch.empty(world * n1 * HIT.itemsize, dtype=torch.uint8, device=comm_dev) for _ in range(2)]
        pending = [[None, None], [None, None]]  # [wave][buffer]
        step_no = [0]

        def step(ev=None):
            b = step_no[0] % nbuf
            step_no[0] += 1
            if use_dist:
                for w in (0, 1):
                    if pending[w][b] is not None:
                        pending[w][b].wait()  # the gathers that last used this buffer pair (two steps ago)
                        pending[w][b] = None
            if ev is not None:
                ev[0].record()
            accel.TraverseBatchDevice(wl.d_rays1, hit_bufs1[b], wl.d_mask1)
            if ev is not None:
                ev[1].record()
            if use_dist:
                src = hit_bufs1[b].cpu() if shared else hit_bufs1[b]  # (test hook: staged through the host for gloo)
                _, pending[0][b] = nd.gather_hit_records(src, world, rank, dist, out=gathered1[b], async_op=True)
            if ev is not None:
                ev[2].record()
            accel.TraverseBatchDevice(wl.d_rays2, hit_bufs2[b], wl.d_mask2)
            if ev is not None:
                ev[3].record()
            if use_dist:
                src = hit_bufs2[b].cpu() if shared else hit_bufs2[b]
                _, pending[1][b] = nd.gather_hit_records(src, world, rank, dist, out=gathered2[b], async_op=True)

        def drain():
            for w in (0, 1):
                for b in range(2):
                    if pending[w][b] is not None:
                        pending[w][b].wait()
                        pending[w][b] = None

        for _ in range(warmup):
            step()
        drain()
        # Per-wave kernel times: a short pass with an event pair around every launch, OUTSIDE the timed region (an event record
        # between two kernels of a stream keeps the second from starting for several microseconds).  The timed region itself
        # carries one event pair around all of its 2 x steps launches; the library records no event of its own (completion records).
        split_steps = 3
        events = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(split_steps)]
        for k in range(split_steps):
            step(events[k])
        drain()
        torch.cuda.synchronize()
        self.k_ms1 = float(np.mean([e[0].elapsed_time(e[1]) for e in events]))
        self.k_ms2 = float(np.mean([e[2].elapsed_time(e[3]) for e in events]))
        region = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        region[0].record()
        for k in range(steps):
            step()
        region[1].record()
        drain()  # every gather issued inside the timed region completes inside it
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        self.dt = time.perf_counter() - t0
        self.kernel_name = accel.LastKernelName()
        self.region_ms = float(region[0].elapsed_time(region[1]))  # HIP events on the launch stream over the timed region
        self.steps = steps
        self.rays_per_step = wl.n1 + wl.n2
        self.per_rank = None
        self.total_rays = float(self.rays_per_step)
        self.gather_check = None
        if use_dist:
            # a blocking gather of one wave's records, timed on its own (outside the timed region)
            g0 = time.perf_counter()
            src = hit_bufs1[0].cpu() if shared else hit_bufs1[0]
            _, wk = nd.gather_hit_records(src, world, rank, dist, out=gathered1[0], async_op=True)
            wk.wait()
            torch.cuda.synchronize()
            gather_ms = (time.perf_counter() - g0) * 1e3
            t = torch.tensor([self.dt, float(self.rays_per_step), self.k_ms1, self.k_ms2, gather_ms, self.region_ms], dtype=torch.float64, device=comm_dev)
            allt = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(allt, t)
            allt = torch.stack(allt).cpu().numpy()
            self.dt = float(allt[:, 0].max())
            self.total_rays = float(allt[:, 1].sum())
            self.per_rank = {"wall_ms_per_step": [round(float(x) / steps * 1e3, 4) for x in allt[:, 0]],
                             "primary_kernel_ms": [round(float(x), 4) for x in allt[:, 2]],
                             "bounce_kernel_ms": [round(float(x), 4) for x in allt[:, 3]],
                             "kernel_ms_max": round(float((allt[:, 2] + allt[:, 3]).max()), 4),
                             "kernel_ms_min": round(float((allt[:, 2] + allt[:, 3]).min()), 4),
                             "launch_ms": [round(float(x) / (2 * steps), 4) for x in allt[:, 5]],
                             "gather_ms_one_wave_blocking": [round(float(x), 4) for x in allt[:, 4]]}
            self.gathered_bytes_per_step = int(2 * world * n1 * HIT.itemsize)
            if check_gather:
                # The frame the root assembled from the LAST step's gathers against each rank's own records of that step (the
                # ranks' records travel once more, through an independent all_gather): de-interleaving included.
                last = (step_no[0] - 1) % nbuf  # the gather above reused buffer 0 for wave 1: compare wave 2 of the last step
                mine = (hit_bufs2[last].cpu() if shared else hit_bufs2[last]).contiguous()
                every = [torch.empty_like(mine) for _ in range(world)]
                dist.all_gather(every, mine)
                if rank == 0:
                    rows = n1 // wl.width
                    got = nd.assemble_image(gathered2[last].cpu().numpy(), wl.width, rows * world, world, HIT)
                    want = np.empty((rows * world, wl.width), dtype=HIT)
                    for r in range(world):
                        want[r::world] = every[r].cpu().numpy().view(HIT).reshape(rows, wl.width)
                    local_ok = bool(gathered2[last][: n1 * HIT.itemsize].cpu().numpy().tobytes() == hit_bufs2[last].cpu().numpy().tobytes())
                    self.gather_check = {"wave": "bounce, last timed step", "records": int(got.shape[0]),
                                         "assembled_frame_identical_to_the_ranks_records": bool(got.tobytes() == want.reshape(-1).tobytes()),
                                         "root_slice_identical_to_its_own_buffer": local_ok,
                                         "device_tensors": not shared, "backend": dist.get_backend()}
        self.value = self.total_rays * steps / self.dt / 1e6
        self.ms_per_step = self.dt / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS),
                    help="default: C3 (the config BASELINE.json's metric is quoted on; with --gpus N > 1 also its strong-scaling case C4 as `strong_c4`)")
    ap.add_argument("--mesh", default=None, help="C2: a .ply / .obj triangle mesh (e.g. Stanford bun_zipper.ply) instead of the procedural stand-in")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--builds", type=int, default=5)
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed extras (2 frames in flight, primary+shadow, end_to_end, next_rows): use for rocprofv3 runs, so "
                         "that the kernel statistics hold the timed region's launches only")
    ap.add_argument("--no-pmc", action="store_true", help="skip the in-run rocprofv3 counter passes (the line then says roofline UNMEASURED)")
    ap.add_argument("--no-configs", action="store_true", help="skip the untimed measurements of the other single-GPU configs")
    ap.add_argument("--no-next-rows", action="store_true", help="skip the untimed figures of the SURVEY 8(f) rows")
    ap.add_argument("--no-strong", action="store_true", help="N > 1, default config: skip the strong-scaling C4 sub-object")
    ap.add_argument("--dry-run", action="store_true", help="validate device count / environment / RCCL / buffer sizes for --gpus N without launching anything")
    ap.add_argument("--force-dist", action="store_true",
                    help="N = 1: run the N > 1 code path anyway — torch.distributed over RCCL (backend nccl, world size 1), the asynchronous "
                         "double-buffered gather of both waves' records on device tensors — so that the exchange executes on a one-GPU box")
    ap.add_argument("--check-gather", action="store_true", help="with a process group: compare the frame the root assembled from the gathers with the ranks' own records")
    ap.add_argument("--pmc-dir", default=None, help="keep the raw rocprofv3 counter CSVs here")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--pmc-configs", default="C3", help=argparse.SUPPRESS)
    ap.add_argument("--pmc-rank", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--pmc-world", type=int, default=1, help=argparse.SUPPRESS)
    args = ap.parse_args()
    default_config = args.config is None
    if default_config:
        args.config = "C3"
    if args.mesh and args.config != "C2":
        raise SystemExit("--mesh applies to --config C2")

    if args.pmc_child:
        return pmc_child(args)
    if args.dry_run:
        sys.exit(dry_run(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_spawn(args, sys.argv[1:]))

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    # Test hook (tests/test_bench_multi.py): NRT_BENCH_TEST_SHARED_GPU=1 lets N ranks share GPU 0 so that the N > 1
    # control flow can be run on a one-GPU box; RCCL cannot put two ranks on one device, so the collectives then go
    # through gloo with CPU staging.  Never set by the driver: its numbers come from one rank per GPU over RCCL.
    shared = world > 1 and os.environ.get("NRT_BENCH_TEST_SHARED_GPU") == "1"
    if shared:
        local_rank = 0
    if world > 1 and not shared and torch.cuda.device_count() < world:
        raise SystemExit("bench.py: %d ranks but only %d GPU(s) visible" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:  # --force-dist without a launcher: a one-rank group of our own
            os.environ.setdefault("MASTER_PORT", str(29500 + (os.getpid() % 2000)))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if shared:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    comm_dev = "cpu" if shared else "cuda"

    from nanort_amd import scenes

    wl = Workload(args.config, rank, world, local_rank, args.builds, args.mesh)
    accel, n1, n2 = wl.accel, wl.n1, wl.n2
    HIT = wl.HIT

    # ---- work counters -> algorithmic bytes per launch ---------------------------
    c1, c2 = wl.counters()
    bytes1, bytes2 = algorithmic_bytes(c1, wl.rb), algorithmic_bytes(c2, wl.rb)

    T = Timed(wl, args.steps, args.warmup, world, rank, dist, shared, check_gather=args.check_gather)
    k_ms1, k_ms2, kernel_name, region_ms = T.k_ms1, T.k_ms2, T.kernel_name, T.region_ms
    launch_ms = region_ms / (2 * args.steps)
    if world > 1:
        t = torch.tensor([float(bytes1 + bytes2)], dtype=torch.float64, device=comm_dev)
        allt = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        agg_bytes = float(torch.stack(allt).sum())
        agg_ms = T.per_rank["kernel_ms_max"]
    else:
        agg_bytes = float(bytes1 + bytes2)
        agg_ms = k_ms1 + k_ms2

    # ---- hardware counters of this run: rank 0's GPU, rank 0's share of the workload (every rank runs the same kernel on
    # the same number of interleaved rows); all ranks wait at the barrier below meanwhile ------------------------------------
    pmc_all, pmc_err = None, None
    other_configs = ["C2", "C4tile", "C5"] if (world == 1 and not args.no_configs and default_config) else []
    configs_out = {}
    if rank == 0 and other_configs:
        for name in other_configs:  # (before the counter passes: their own kernel times go into the fractions)
            try:
                configs_out[name] = measure_config(name)
            except Exception as e:  # pragma: no cover
                configs_out[name] = {"error": repr(e)}
    if rank == 0 and not args.no_pmc:
        keep = args.pmc_dir and os.path.abspath(args.pmc_dir)
        pmc_all, pmc_err = pmc_collect([args.config] + [c for c in other_configs if "error" not in configs_out.get(c, {})],
                                       args.mesh, rank=0, world=world, keep_dir=keep)
    n_cus = torch.cuda.get_device_properties(local_rank).multi_processor_count

    strong = None
    if world > 1 and default_config and not args.no_strong:
        # BASELINE.json's strong-scaling case beside the weak-scaled headline: every rank builds the 10M-triangle plane and
        # traces its 4096/N interleaved rows of the fixed 4096x4096 frame
        del wl.d_rays1, wl.d_rays2
        torch.cuda.empty_cache()
        wl4 = Workload("C4", rank, world, local_rank, 2, None)
        T4 = Timed(wl4, max(2, args.steps // 4), 1, world, rank, dist, shared)
        strong = {"config": "C4", "workload": wl4.describe(), "scaling": "strong", "value": round(T4.value, 3), "unit": "Mrays/s",
                  "steps": T4.steps, "ms_per_step": round(T4.ms_per_step, 4), "rays_per_step": int(T4.total_rays),
                  "build_ms": round(float(np.median(wl4.build_ms)), 4), "bvh": {"nodes": wl4.num_nodes, "max_depth": int(wl4.stats["max_tree_depth"])},
                  "multi_gpu": T4.per_rank}
        del wl4

    if rank == 0:
        k_ms = {"primary": k_ms1, "bounce": k_ms2}
        alg_gbs = agg_bytes / (agg_ms * 1e-3) / 1e9  # all ranks' algorithmic bytes over the slowest rank's two launches
        build_ms = float(np.median(wl.build_ms))
        bbytes = build_bytes(wl.faces.shape[0], wl.num_nodes, wl.rb)
        roof = {
            "kernel": kernel_name,
            "limiting": "the length of each wave's own instruction stream between two node fetches, and the fetch latency behind it: five waves "
                        "per SIMD each issue their ~200 instructions per step in order, so the loop's speed follows the step's instruction count "
                        "whatever unit executes it (round 3: 24 selects fewer per step = +2.7 %, three guarded pushes turned into stores = +2 %, the loop's own bookkeeping once per two steps = +1.8 %; "
                        "six more unpacking instructions for one fetch fewer = -1.7 %; 19 vector instructions moved to 59 scalar ones = -1.2 %: "
                        "profiles/r03K-r03R); waves wait on L1/L2 ~40 % of their cycles, the vector unit is ~38 % busy, a lane's eight 16-byte "
                        "pieces of a node record cost the L1 ~0.7 clk each (tools/ubench/node_fetch.hip); the last quarter of a launch is the "
                        "chain of its longest rays (tools/drain_probe.py, tools/tail_first_probe.py: an oracle ordering with the longest 1 % of "
                        "the rays first takes 11 % off the bounce wave).  HBM is far from saturated: see hbm / valu / l1 (DESIGN.md 3.1, 5); no "
                        "MFMA in this path",
            "launch_ms": round(launch_ms, 4),
            "launch_ms_note": "HIP events on the launch stream around the whole timed region / (2 x steps): the average launch of the two "
                              "waves, idle time between launches included; per_wave.ms: event pairs around single launches in a 3-step "
                              "pass outside the timed region",
            "algorithmic": {"bytes_per_launch": int((bytes1 + bytes2) // 2), "GBs": round(alg_gbs, 1),
                            "x_hbm_peak": round(alg_gbs / HBM_PEAK_GBS, 4), "served_from_cache": True,
                            "note": "SURVEY 8(d) bytes the reference's loop would touch (52 + 40*nodes + 52*tris per ray, counted by the "
                                    "kernel's own counting pass on the tree traversed); they are served by L1/L2/Infinity Cache, so this "
                                    "is not a bandwidth claim and may exceed the HBM peak"},
            "build": {"bytes": int(bbytes), "ms": round(build_ms, 4), "GBs": round(bbytes / (build_ms * 1e-3) / 1e9, 1),
                      "frac": round(bbytes / (build_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                      "note": "compulsory traffic 52N + 40*nodes + 4N (SURVEY 8d) over the build's device time"},
            "per_wave": per_wave_counts(wl, c1, c2, k_ms1, k_ms2),
        }
        source = None
        if pmc_all is not None and pmc_all.get(args.config, {}).get("primary"):
            r = roofline_from_counters(pmc_all[args.config], k_ms, n_cus, launch_ms=launch_ms)
            if r["hbm"]:
                roof["hbm"], source = r["hbm"], "in-run rocprofv3 --pmc passes (3 steps of the same workload, this rank's share)"
            if r["valu"]:
                roof["valu"] = r["valu"]
            if r["l1"]:
                roof["l1"] = r["l1"]
        if pmc_err:
            roof["pmc_error"] = pmc_err
        hb = roof.get("hbm")
        # Compulsory bytes of an average launch: every ray record read and every hit record + flag written once, plus the part of
        # the private tree a launch can touch read ONCE (an upper bound: the whole Wide4Node / WideNode array and every leaf record,
        # capped by what the walk fetched at all) — the denominator-free counterpart of `traffic`: traffic / compulsory is the
        # re-read factor, compulsory_frac the HBM fraction the launch would reach if every byte moved once.
        branches = int(wl.stats["num_branch_nodes"])
        rec_b = 128 if wl.rb == 4 else 112
        tri_b = 40 if wl.rb == 4 else 80
        tree_once = branches * rec_b + wl.faces.shape[0] * tri_b
        io_b = {w: n * (wl.RAY.itemsize + wl.HIT.itemsize + 1) for w, n in (("primary", n1), ("bounce", n2))}
        alg_tree = {"primary": bytes1 - 52 * n1 if wl.rb == 4 else bytes1 - 104 * n1, "bounce": bytes2 - 52 * n2 if wl.rb == 4 else bytes2 - 104 * n2}
        comp = sum(io_b[w] + min(tree_once, max(0, alg_tree[w])) for w in ("primary", "bounce")) / 2.0
        comp_gbs = comp / (launch_ms * 1e-3) / 1e9
        roof.update({"bound": "hbm", "achieved": hb["achieved_GBs"] if hb else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": hb["frac"] if hb else None, "traffic": hb["bytes_per_launch"] if hb else None,
                     "traffic_source": source or "UNMEASURED (no counter pass in this run)",
                     "achieved_definition": "HBM-side bytes per launch from this run's FETCH_SIZE / WRITE_SIZE counters over the average launch of the "
                                            "timed region (a fraction of a real peak); SURVEY 8(d)'s algorithmic bytes are `algorithmic_*` below — they "
                                            "count re-reads that L2 and the Infinity Cache serve and exceed the HBM peak",
                     "algorithmic_bytes": int((bytes1 + bytes2) // 2), "algorithmic_GBs": round(alg_gbs, 1),
                     "algorithmic_x_hbm_peak": round(alg_gbs / HBM_PEAK_GBS, 4),
                     "compulsory_bytes": int(comp), "compulsory_GBs": round(comp_gbs, 1), "compulsory_frac": round(comp_gbs / HBM_PEAK_GBS, 4),
                     "compulsory_definition": "per launch: rays in + hit records and flags out, once, + the private tree (%d branch records x %d B + %d leaf "
                                              "records x %d B = %.1f MB) read once — an upper bound on what a launch must move" % (
                                                  branches, rec_b, wl.faces.shape[0], tri_b, tree_once / 1e6),
                     "traffic_over_compulsory": round(hb["bytes_per_launch"] / comp, 3) if hb else None})
        if hb and world > 1:
            # per-rank fractions: this rank's measured bytes per launch (the ranks' shares are equally many interleaved rows of the
            # same frame) over every rank's own average launch of the timed region
            fr = [hb["bytes_per_launch"] / (x * 1e-3) / 1e9 / HBM_PEAK_GBS for x in T.per_rank["launch_ms"]]
            roof["per_rank_hbm_frac"] = {"max": round(max(fr), 4), "min": round(min(fr), 4),
                                         "note": "rank 0's counters (bytes per launch of its share) over each rank's own launch time"}
        cfg = wl.cfg
        par = "replicated BVH, interleaved image rows per GPU"
        if world > 1:
            par += (", RCCL gather of both waves' hit records to rank 0 (send/recv over xGMI), double-buffered and overlapped with "
                    "the following waves; %s" % ("strong scaling: fixed %dx%d frame cut into %d row-interleaved tiles" % (cfg["w"], cfg["h"], world)
                                                 if cfg["scaling"] == "strong" else "weak scaling: %dx%d rays per GPU" % (cfg["w"], wl.rows)))
        out = {
            "metric": METRIC,
            "value": round(T.value, 3),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(T.ms_per_step, 4),
            "higher_is_better": True,
            "scaling": cfg["scaling"] if "tile_of" not in cfg else "weak",
            "vs_baseline": None,
            "dtype": cfg["real"],
            "data": "synthetic" if not wl.mesh_note else "user mesh, synthetic rays",
            "config": {"name": args.config, "workload": wl.describe(), "parallelism": par, "rays_per_step": int(T.total_rays)},
            "build_ms": round(build_ms, 4),
            "bvh": {"nodes": wl.num_nodes, "max_depth": int(wl.stats["max_tree_depth"])},
            "roofline": roof,
        }
        if dist is not None:
            out["multi_gpu"] = dict(T.per_rank, rccl_ranks=world, backend="gloo (test hook)" if shared else "nccl (RCCL)",
                                    gathered_bytes_per_step=T.gathered_bytes_per_step)
            if T.gather_check is not None:
                out["multi_gpu"]["gather_check"] = T.gather_check
            if strong is not None:
                out["strong_c4"] = strong
        if world == 1 and not args.no_extras:
            # extras, outside the timed region: (a) the same K steps with two frames in flight (steps alternate
            # between two streams; a launch's drain tail is filled by the next frame's rays), (b) SURVEY 8(d)'s
            # primary + shadow pair, (c) the host entry point end to end
            out["pipelined"] = pipelined(accel, torch, (wl.d_rays1, wl.d_hits1, wl.d_mask1), (wl.d_rays2, wl.d_hits2, wl.d_mask2), args.steps, n1 + n2)
            if n2:
                out["multi_batch"] = multi_batch(accel, torch, (wl.d_rays1, wl.d_hits1, wl.d_mask1), (wl.d_rays2, wl.d_hits2[: n2 * HIT.itemsize], wl.d_mask2[:n2]),
                                                 args.steps, n1 + n2)
            if wl.real == np.float32 and not accel.GetTunable("order4"):
                out["opt_in_distance_order"] = opt_in_distance_order(wl, args.steps)
            rays_s = scenes.secondary_rays("shadow", wl.verts32, wl.faces, wl.rays1_f32, wl.hits1_f32, wl.mask1)
            if wl.real != np.float32:
                from nanort_amd.wire import widen_rays

                rays_s = widen_rays(rays_s)
            d_rs = torch.from_numpy(rays_s.view(np.uint8)).cuda()
            d_hs = torch.empty(max(1, rays_s.shape[0]) * HIT.itemsize, dtype=torch.uint8, device="cuda")
            ts = []
            for _ in range(5):
                accel.TraverseBatchDevice(d_rs, d_hs)
                ts.append(accel.LastTraverseMs())
            ms_s = float(np.median(ts))
            out["primary_plus_shadow"] = {"value": round((n1 + rays_s.shape[0]) / (k_ms1 + ms_s) / 1e3, 1), "unit": "Mrays/s",
                                          "shadow_ms": round(ms_s, 4), "shadow_rays": int(rays_s.shape[0])}
            if wl.real == np.float32 and rays_s.shape[0]:
                # ... and as ONE launch over both waves, the shadow wave as an occlusion query (nrtTraverseBatchesDevice): one launch tail
                d_ms = torch.empty(rays_s.shape[0], dtype=torch.uint8, device="cuda")
                pair = [(wl.d_rays1, wl.d_hits1, wl.d_mask1), (d_rs, None, d_ms, None, "occlusion")]
                for _ in range(2):
                    accel.TraverseBatchesDevice(pair)
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                torch.cuda.synchronize()
                ev[0].record()
                for _ in range(5):
                    accel.TraverseBatchesDevice(pair)
                ev[1].record()
                torch.cuda.synchronize()
                ms_pair = float(ev[0].elapsed_time(ev[1])) / 5
                accel.TraverseBatchDevice(d_rs, d_hs, d_ms.new_empty(d_ms.shape))
                flags_sep = torch.empty_like(d_ms)
                accel.TraverseBatchDevice(d_rs, d_hs, flags_sep)
                torch.cuda.synchronize()
                out["primary_plus_shadow"]["one_launch"] = {"value": round((n1 + rays_s.shape[0]) / ms_pair / 1e3, 1), "ms": round(ms_pair, 4),
                                                            "occlusion_flags_equal_closest_hit_flags": bool(torch.equal(d_ms, flags_sep))}
                del d_ms, flags_sep
            del d_rs, d_hs
            # a SECOND bounce: rays generated from the bounce-1 hits by the same host generator, traced in the order the
            # renderer produces them (how far does coherence decay with depth? — profiles/r03a_reorder_probe_*: a random
            # order of the bounce-1 wave costs 25-46 %)
            try:
                if wl.real == np.float32 and n2:
                    _, _, gh2, gm2 = wl.results()
                    rays3 = scenes.secondary_rays("bounce", wl.verts32, wl.faces, wl.rays2, gh2, gm2, pixel_base=7 * n1)
                    if rays3.shape[0]:
                        d_r3 = torch.from_numpy(rays3.view(np.uint8)).cuda()
                        d_h3 = torch.empty(rays3.shape[0] * HIT.itemsize, dtype=torch.uint8, device="cuda")
                        ts = []
                        for _ in range(5):
                            accel.TraverseBatchDevice(d_r3, d_h3)
                            ts.append(accel.LastTraverseMs())
                        ms3 = float(np.median(ts))
                        c3 = accel.TraverseCountDevice(d_r3)
                        # the same number of bounce-1 rays (every k-th): what a batch this small reaches with bounce-1's coherence
                        sub = np.ascontiguousarray(wl.rays2[:: max(1, n2 // rays3.shape[0])][: rays3.shape[0]])
                        d_rsub = torch.from_numpy(sub.view(np.uint8)).cuda()
                        ts = []
                        for _ in range(5):
                            accel.TraverseBatchDevice(d_rsub, d_h3)
                            ts.append(accel.LastTraverseMs())
                        ms_sub = float(np.median(ts))
                        out["bounce2"] = {"rays": int(rays3.shape[0]), "ms": round(ms3, 4), "Mrays_s": round(rays3.shape[0] / ms3 / 1e3, 1),
                                          "bounce1_Mrays_s": round(n2 / k_ms2 / 1e3, 1),
                                          "bounce1_subsampled_to_the_same_batch_size_Mrays_s": round(sub.shape[0] / ms_sub / 1e3, 1),
                                          "note": "the second bounce is slower per ray because the batch is small (the launch's ramp and tail), not because "
                                                  "coherence is lost: profiles/r04i_bounce2_probe.txt (best re-ordering +2.4 %)",
                                          "nodes_per_ray": round(c3["nodes_visited"] / max(1, rays3.shape[0]), 2),
                                          "tris_per_ray": round(c3["tris_tested"] / max(1, rays3.shape[0]), 2)}
                        del d_r3, d_h3, d_rsub
            except Exception as e:  # pragma: no cover
                out["bounce2"] = {"error": repr(e)}
            try:
                out["end_to_end"] = end_to_end(wl)
            except Exception as e:  # pragma: no cover
                out["end_to_end"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            nodes, indices = accel.GetTree()
            # the timed region's own output buffers
            accel.TraverseBatchDevice(wl.d_rays1, wl.d_hits1, wl.d_mask1)
            accel.TraverseBatchDevice(wl.d_rays2, wl.d_hits2, wl.d_mask2)
            budget = 12.0 if args.config in ("C3", "C2") else 6.0
            timed_walk = wl.results()
            ref_order = reference_order_results(wl) if accel.GetTunable("order4") and "k_traverse_wide" in accel.LastKernelName() and wl.real == np.float32 else None
            out["cpu_baseline"] = cpu_baseline(wl.verts, wl.faces, wl.rays1, wl.rays2, nodes, indices, wl.width, timed_walk, budget_s=budget,
                                               gpu_results_ref_order=ref_order)
        if configs_out:
            out["configs"] = {}
            for name, e in configs_out.items():
                k_ms_c, counts = e.pop("_k_ms", None), e.pop("_counts", None)
                if pmc_all is not None and k_ms_c and pmc_all.get(name, {}).get("primary"):
                    e["roofline"] = compact_roofline(roofline_from_counters(pmc_all[name], k_ms_c, n_cus), counts)
                elif counts:
                    e["roofline"] = {"waves": counts, "note": "UNMEASURED (no counter pass in this run)"}
                out["configs"][name] = e
        if world == 1 and default_config and not args.no_next_rows and not args.no_extras:
            del wl
            torch.cuda.empty_cache()
            try:
                import bench_rows

                out["next_rows"] = bench_rows.next_rows(counters=not args.no_pmc, pmc_dir=args.pmc_dir and os.path.abspath(args.pmc_dir))
            except Exception as e:  # pragma: no cover
                out["next_rows"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Untimed extras of the default run (never `value`): frames in flight, several batches per launch, the opt-in walk, the host
entry point end to end, the other single-GPU configs.  They go to the side file, not to the driver's line."""
import time

import numpy as np

from . import HBM_PEAK_GBS
from .baseline import bit_identical, host_threads, parity, reference_order_results
from .workload import Workload, build_bytes, per_wave_counts

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


def measure_config(name, mesh_path=None, reps=20, parity_rays=200_000, ramp=60):
    """{Mrays/s, build_ms, parity} for one config on GPU 0, with the reference's answer on a bounded sample.  The counters of
    the config are attached by the caller (one profiled sub-run serves all configs)."""
    import torch

    wl = Workload(name, builds=3, mesh_path=mesh_path)
    a = wl.accel
    for _ in range(ramp):  # (untimed: the device's clocks ramp over some tens of milliseconds of work, as in the headline's region)
        a.TraverseBatchDevice(wl.d_rays1, wl.d_hits1, wl.d_mask1)
        a.TraverseBatchDevice(wl.d_rays2, wl.d_hits2, wl.d_mask2)
    t1, t2 = [], []
    for _ in range(5):
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


def primary_plus_shadow(wl, k_ms1):
    """SURVEY 8(d)'s primary + shadow pair: the shadow wave generated on the host from the wave-1 hits, traced as a second launch and as
    ONE launch over both waves (the shadow wave as an occlusion query)."""
    import torch

    from nanort_amd import scenes

    accel, n1, HIT = wl.accel, wl.n1, wl.HIT
    out = {}
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
    return out["primary_plus_shadow"]


def bounce2(wl, k_ms2):
    """A SECOND bounce: rays generated from the bounce-1 hits by the same host generator, traced in the order the renderer produces them."""
    import torch

    from nanort_amd import scenes

    accel, n1, n2, HIT = wl.accel, wl.n1, wl.n2, wl.HIT
    out = {}
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
    return out.get("bounce2")

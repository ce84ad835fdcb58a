"""The checker side of the bench: parity of GPU records against the reference, and the reference (oracle/_ref) or the C port timed
on the box's host cores on a bounded sample (`cpu_baseline`).  The only bench module that touches oracle/."""
import time

import numpy as np

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
            "sample_short": "unmodified nanort.h, g++ -O3 -fopenmp, own Build; every %d-th row of wave 1 + every %d-th wave-2 ray = %d rays, best of 3" % (
                step, step, s1.shape[0] + s2.shape[0]),
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

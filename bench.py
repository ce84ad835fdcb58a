#!/usr/bin/env python3
"""bench.py — the headline measurement of the hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2|C3|C4|C4tile|C5] [--mesh file.ply|.obj]

Metric (BASELINE.json): Mrays/s (primary + 1-bounce) at 1920x1080 on the 1M-triangle mesh; BVH build ms.
Default workload = config C3 of SURVEY.md §8(d): Plane(1000,500) (exactly 1 000 000 triangles), fp32, objrender camera.

One "step" = one pass of the hot path over one batch: wave 1 (W*H primary rays) + wave 2 (one cosine-weighted bounce ray per
wave-1 hit), both already resident in HBM, traced by the batched traversal kernel through the C ABI (nrtTraverseBatchDevice_*)
on torch's current stream.  The BVH is built on the GPU (nrtBuild_*) before the timed region; its device time is `build_ms`
(median of several builds), the application-visible `BVHAccel::Build()` of include/nanort.h is `build_host_ms`.

N > 1: `python bench.py --gpus N` starts N ranks itself (torch.distributed.run, one per GPU, RCCL) when it was not launched
under a launcher already, and fails loudly when the box has fewer than N GPUs.  Each rank holds a replica of the BVH
(deterministic GPU build, no broadcast) and traces the interleaved image rows y = rank (mod N); the hit records of BOTH waves
are gathered to rank 0 (RCCL gather = grouped send/recv over xGMI), asynchronously and double-buffered.  C2/C3/C5 scale weakly
(the image grows to W x (H*N)); `--config C4` is BASELINE.json's strong-scaling case.

Output: ONE JSON line on stdout, kept small (benchlib/line.py: the contract's keys + `roofline` + `cpu_baseline` + a few figures of
merit, a few KB) — and everything else the run measured (other configs, SURVEY 8(f) rows, per-wave parity blocks, counter
detail) in the side file the line names (`extras_file`, default gpurun_out/bench_extras.json).  Progress notes go to stderr.

  roofline      the dominant kernel (named by the library: nrtLastKernelName).  `bound` = the unit this run's counters show
                closest to its peak (hbm / l1 / valu), with achieved / peak / frac of THAT unit; `hbm` = the contract's HBM figure
                whichever unit binds (FETCH_SIZE x 2 + WRITE_SIZE per launch, rocprofv3 --pmc passes over a 3-step sub-run of this
                script, over the average launch of the timed region); `traffic` = those HBM bytes per launch; `bytes_per_launch` =
                algorithmic (52 + 40*nodes + 52*tris per ray) -> requested (what the timed kernel asks of the L1: record
                bytes x steps + leaf-record bytes x primitives + ray + record, from the counting instantiation) -> l1 (looked up,
                counters) -> fetched (= traffic) -> compulsory (every byte once).  DESIGN.md 3.1 says what limits the kernel.
  cpu_baseline  the UNMODIFIED reference (oracle/_ref, OpenMP, host cores) on a bounded sample of the same ray buffers (best of
                3), with the parity check of the same run; falls back to the single-thread C port.

The code lives in benchlib/: workload.py (set-up), timed.py (the timed region), counters.py (rocprofv3 passes), roofline.py,
baseline.py (the only module that touches oracle/), extras.py (untimed extras), multigpu.py (self-spawn, --dry-run), line.py.
"""
import argparse
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchlib import HBM_PEAK_GBS, L1_PEAK_GACC_S, METRIC  # noqa: E402,F401  (re-exported: tools/, bench_rows.py)
from benchlib.counters import PMC_PASSES, compact_roofline, pmc_child, pmc_collect, roofline_from_counters, walk_counts, walk_counts_child  # noqa: E402,F401
from benchlib.workload import CONFIGS, Workload, algorithmic_bytes, build_bytes, per_wave_counts  # noqa: E402,F401


def note(msg):
    sys.stderr.write("[bench] %s\n" % msg)
    sys.stderr.flush()


def build_host_ms(config):
    """The application-visible `BVHAccel<T>::Build()` of include/nanort.h (what examples/path_tracer/main.cc:742-766 times),
    measured by the C++ helper tools/bin/build_host (tools/build_host.cc; built by __graft_entry__.build())."""
    exe = os.path.join(ROOT, "tools", "bin", "build_host")
    if not os.path.exists(exe):
        return {"error": "tools/bin/build_host not built"}
    cfg = CONFIGS[config]
    if cfg["mesh"] == "sphere":
        return None
    _, nx, ny = cfg["mesh"]
    try:
        r = subprocess.run([exe, str(nx), str(ny), cfg["real"]], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, cwd=ROOT)
        for ln in r.stdout.splitlines():
            if ln.startswith("{"):
                return json.loads(ln)
        return {"error": "rc %d: %s" % (r.returncode, (r.stderr or r.stdout)[-200:])}
    except Exception as e:  # pragma: no cover
        return {"error": repr(e)}


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
    ap.add_argument("--gather", default="torch", choices=("torch", "cabi"),
                    help="with a process group: how the records reach rank 0 — torch.distributed.gather over RCCL (default), or the C ABI's own "
                         "nrtGroupCreateRanked / nrtGroupTraverseGather (RCCL send / recv from the library: what a C++ host runs)")
    ap.add_argument("--check-gather", action="store_true", help="with a process group: compare the frame the root assembled from the gathers with the ranks' own records")
    ap.add_argument("--pmc-dir", default=None, help="keep the raw rocprofv3 counter CSVs here")
    ap.add_argument("--extras-file", default=None, help="where the full result object goes (default gpurun_out/bench_extras.json)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--walk-counts-child", action="store_true", help=argparse.SUPPRESS)
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
    if args.walk_counts_child:
        return walk_counts_child(args)
    if args.dry_run:
        from benchlib.multigpu import dry_run

        sys.exit(dry_run(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        from benchlib.multigpu import self_spawn

        sys.exit(self_spawn(args, sys.argv[1:]))

    import torch

    from benchlib import extras as bx
    from benchlib import line as bl
    from benchlib import roofline as br
    from benchlib.timed import Timed

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

    wl = Workload(args.config, rank, world, local_rank, args.builds, args.mesh)
    accel, n1, n2 = wl.accel, wl.n1, wl.n2
    HIT = wl.HIT

    # ---- work counters -> algorithmic bytes per launch ---------------------------
    c1, c2 = wl.counters()
    bytes1, bytes2 = algorithmic_bytes(c1, wl.rb), algorithmic_bytes(c2, wl.rb)

    if args.gather == "cabi" and dist is not None and not shared:
        from benchlib.timed import TimedCAbi

        T = TimedCAbi(wl, args.steps, args.warmup, world, rank, dist, check_gather=args.check_gather)
    else:
        T = Timed(wl, args.steps, args.warmup, world, rank, dist, shared, check_gather=args.check_gather)
    k_ms1, k_ms2, kernel_name, region_ms = T.k_ms1, T.k_ms2, T.kernel_name, T.region_ms
    launch_ms = region_ms / (2 * args.steps)
    if rank == 0:
        note("%s: %.1f Mrays/s, %.4f ms per step (%d steps), launch %.4f ms" % (args.config, T.value, T.ms_per_step, args.steps, launch_ms))
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
    counts, counts_err = None, None
    other_configs = ["C2", "C4tile", "C5"] if (world == 1 and not args.no_configs and default_config) else []
    configs_out = {}
    if rank == 0 and other_configs:
        for name in other_configs:  # (before the counter passes: their own kernel times go into the fractions)
            try:
                configs_out[name] = bx.measure_config(name)
                note("%s: %s Mrays/s, build %s ms" % (name, configs_out[name].get("value"), configs_out[name].get("build_ms")))
            except Exception as e:  # pragma: no cover
                configs_out[name] = {"error": repr(e)}
    if rank == 0 and not args.no_pmc:
        keep = args.pmc_dir and os.path.abspath(args.pmc_dir)
        pmc_all, pmc_err = pmc_collect([args.config] + [c for c in other_configs if "error" not in configs_out.get(c, {})],
                                       args.mesh, rank=0, world=world, keep_dir=keep)
        counts, counts_err = walk_counts(args.config, args.mesh, rank=0, world=world)
        note("counter passes done%s%s" % ("; " + pmc_err if pmc_err else "", "; walk counts: " + counts_err if counts_err else ""))
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
        build_roof = {"bytes": int(bbytes), "ms": round(build_ms, 4), "GBs": round(bbytes / (build_ms * 1e-3) / 1e9, 1),
                      "frac": round(bbytes / (build_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
        counters = None
        if pmc_all is not None and pmc_all.get(args.config, {}).get("primary"):
            counters = roofline_from_counters(pmc_all[args.config], k_ms, n_cus, launch_ms=launch_ms)
        comp, _tree_once = br.compulsory_bytes(wl, bytes1, bytes2)
        roof = br.headline(kernel_name, launch_ms, k_ms, (bytes1 + bytes2) // 2, alg_gbs, comp, counters, br.requested_bytes(wl, counts), build_roof)
        roof["detail"] = {"per_wave": per_wave_counts(wl, c1, c2, k_ms1, k_ms2), "counters": counters, "walk_counts": counts,
                          "launch_ms_definition": "HIP events on the launch stream around the whole timed region / (2 x steps)"}
        if pmc_err or counts_err:
            roof["detail"]["errors"] = [e for e in (pmc_err, counts_err) if e]
        hb = counters.get("hbm") if counters else None
        if hb and world > 1:
            # per-rank fractions: this rank's measured bytes per launch (the ranks' shares are equally many interleaved rows of the
            # same frame) over every rank's own average launch of the timed region
            fr = [hb["bytes_per_launch"] / (x * 1e-3) / 1e9 / HBM_PEAK_GBS for x in T.per_rank["launch_ms"]]
            roof["per_rank_hbm_frac"] = {"max": round(max(fr), 4), "min": round(min(fr), 4)}
        cfg = wl.cfg
        par = "replicated BVH, interleaved image rows per GPU"
        if world > 1:
            par += (", RCCL gather of both waves' hit records to rank 0, double-buffered and overlapped; %s" % (
                "strong scaling: fixed %dx%d frame in %d row-interleaved tiles" % (cfg["w"], cfg["h"], world)
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
            "config": {"name": args.config, "workload": wl.describe(), "parallelism": par, "rays_per_step": int(T.total_rays),
                       "untimed_clock_ramp_steps": int(T.prewarm_steps)},
            "build_ms": round(build_ms, 4),
            "bvh": {"nodes": wl.num_nodes, "max_depth": int(wl.stats["max_tree_depth"])},
            "roofline": roof,
        }
        if dist is not None:
            out["multi_gpu"] = dict(T.per_rank, rccl_ranks=world, backend="gloo (test hook)" if shared else ("RCCL through the C ABI (nrtGroup*)" if args.gather == "cabi" else "nccl (RCCL)"),
                                    gathered_bytes_per_step=T.gathered_bytes_per_step)
            if T.gather_check is not None:
                out["multi_gpu"]["gather_check"] = T.gather_check
            if strong is not None:
                out["strong_c4"] = strong
        if world == 1 and not args.no_extras:
            # extras, outside the timed region: (a) the same K steps with two frames in flight (steps alternate
            # between two streams; a launch's drain tail is filled by the next frame's rays), (b) SURVEY 8(d)'s
            # primary + shadow pair, (c) the host entry point end to end, (d) the application-visible Build()
            w1, w2 = (wl.d_rays1, wl.d_hits1, wl.d_mask1), (wl.d_rays2, wl.d_hits2, wl.d_mask2)
            out["pipelined"] = bx.pipelined(accel, torch, w1, w2, args.steps, n1 + n2)
            if n2:
                out["multi_batch"] = bx.multi_batch(accel, torch, w1, (wl.d_rays2, wl.d_hits2[: n2 * HIT.itemsize], wl.d_mask2[:n2]), args.steps, n1 + n2)
            if wl.real == np.float32 and not accel.GetTunable("order4"):
                out["opt_in_distance_order"] = bx.opt_in_distance_order(wl, args.steps)
            out["primary_plus_shadow"] = bx.primary_plus_shadow(wl, k_ms1)
            b2 = bx.bounce2(wl, k_ms2)
            if b2:
                out["bounce2"] = b2
            try:
                out["end_to_end"] = bx.end_to_end(wl)
            except Exception as e:  # pragma: no cover
                out["end_to_end"] = {"error": repr(e)}
            bh = build_host_ms(args.config)
            if bh:
                out["build_host_ms"] = bh
            note("extras done")
        if world == 1 and not args.no_cpu_baseline:
            from benchlib.baseline import cpu_baseline, reference_order_results

            nodes, indices = accel.GetTree()
            # the timed region's own output buffers
            accel.TraverseBatchDevice(wl.d_rays1, wl.d_hits1, wl.d_mask1)
            accel.TraverseBatchDevice(wl.d_rays2, wl.d_hits2, wl.d_mask2)
            budget = 12.0 if args.config in ("C3", "C2") else 6.0
            timed_walk = wl.results()
            ref_order = reference_order_results(wl) if accel.GetTunable("order4") and "k_traverse_wide" in accel.LastKernelName() and wl.real == np.float32 else None
            out["cpu_baseline"] = cpu_baseline(wl.verts, wl.faces, wl.rays1, wl.rays2, nodes, indices, wl.width, timed_walk, budget_s=budget,
                                               gpu_results_ref_order=ref_order)
            note("cpu baseline done: %s Mrays/s on %s threads" % (out["cpu_baseline"].get("value"), out["cpu_baseline"].get("cores")))
        if configs_out:
            out["configs"] = {}
            for name, e in configs_out.items():
                k_ms_c, cnts = e.pop("_k_ms", None), e.pop("_counts", None)
                if pmc_all is not None and k_ms_c and pmc_all.get(name, {}).get("primary"):
                    e["roofline"] = compact_roofline(roofline_from_counters(pmc_all[name], k_ms_c, n_cus), cnts)
                elif cnts:
                    e["roofline"] = {"waves": cnts, "note": "UNMEASURED (no counter pass in this run)"}
                out["configs"][name] = e
        if world == 1 and default_config and not args.no_next_rows and not args.no_extras:
            del wl
            torch.cuda.empty_cache()
            try:
                import bench_rows

                out["next_rows"] = bench_rows.next_rows(counters=not args.no_pmc, pmc_dir=args.pmc_dir and os.path.abspath(args.pmc_dir))
            except Exception as e:  # pragma: no cover
                out["next_rows"] = {"error": repr(e)}
        extras_file = bl.write_extras(out, args.extras_file)
        sys.stderr.flush()
        print(json.dumps(bl.compact_line(out, extras_file)), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

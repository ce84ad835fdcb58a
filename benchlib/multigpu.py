"""`bench.py --gpus N` without a launcher (self-spawn of the ranks) and the `--dry-run` preflight."""
import json
import os
import subprocess
import sys

from . import ROOT
from .workload import CONFIGS

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
        tris = 2 * 264 * 131  # the stand-in lat-long sphere 264 x 132 (nanort_amd/scenes.py): 69 168 triangles
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
           "--master-port", str(port), os.path.join(ROOT, "bench.py")] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env, cwd=ROOT)

"""RCCL executed on the MI355X ("RCCL gather of hit records over xGMI").  A one-GPU box cannot show a scaling
curve, but it can show that the exchange RUNS: torch.distributed with backend "nccl" (== RCCL on ROCm), world size 1, the
hit records of both waves gathered from DEVICE tensors, asynchronously and double-buffered, and the root's assembled frame
equal to what a single context returns.

  * through bench.py (`--force-dist --check-gather`): the N > 1 code path of the timed region — process group, barrier,
    max-over-ranks timing, nanort_amd.dist.gather_hit_records, assemble_image — in a child process under a timeout;
  * in THIS process: the same calls on a small frame (so that the driver's record of the libraries this pytest process
    loaded shows librccl next to libnanort_hip.so).
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

_child_ok = {"ran": False}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_runs_its_multi_gpu_path_over_rccl_on_one_gpu():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--check-gather", "--steps", "4", "--warmup", "1",
                        "--builds", "1", "--no-cpu-baseline", "--no-extras", "--no-pmc", "--no-configs"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["steps"] == 4 and out["value"] > 500.0
    mg = out["multi_gpu"]
    assert mg["backend"] == "nccl (RCCL)" and mg["rccl_ranks"] == 1
    assert mg["gathered_bytes_per_step"] == 2 * 1920 * 1080 * 16  # both waves' records reach the root every step
    gc = mg["gather_check"]
    assert gc["backend"] == "nccl" and gc["device_tensors"] is True
    assert gc["assembled_frame_identical_to_the_ranks_records"] is True and gc["root_slice_identical_to_its_own_buffer"] is True
    assert gc["records"] == 1920 * 1080
    assert mg["gather_ms_one_wave_blocking"][0] > 0.0
    _child_ok["ran"] = True


def test_bench_runs_its_multi_gpu_path_through_the_c_abi_group():
    """`--gather cabi`: the same timed region with the exchange done by the library itself — nrtGroupCreateRanked (the RCCL id is all
    torch hands round), wave 1 gathered in frame order, the ragged wave 2 in tile slots; a world of one sends to itself so that
    ncclSend / ncclRecv execute.  The root's frame and slots equal the rank's own records."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--gather", "cabi", "--check-gather", "--steps", "4",
                        "--warmup", "1", "--builds", "1", "--no-cpu-baseline", "--no-extras", "--no-pmc", "--no-configs"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 8192, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["steps"] == 4 and out["value"] > 500.0
    mg = out["multi_gpu"]
    assert mg["backend"].startswith("RCCL through the C ABI") and mg["rccl_ranks"] == 1
    assert mg["gathered_bytes_per_step"] == 2 * 1920 * 1080 * 16
    assert mg["last_gather_bytes"]["rccl"] == 1920 * 1080 * 17 and mg["last_gather_bytes"]["peer"] == 0  # records + flags of the root's own tile, sent to itself
    gc = mg["gather_check"]
    assert gc["frame_identical_to_the_ranks_records"] is True and gc["tile_slots_identical_to_the_ranks_records"] is True
    assert gc["records"] == 1920 * 1080


def test_rccl_gather_of_device_hit_records_in_this_process(oracle, c1_mesh):
    if not _child_ok["ran"]:
        pytest.skip("the child-process run of the same exchange did not pass (or was deselected): not risking this process")
    import torch
    import torch.distributed as dist

    from nanort_amd import BVHAccel, TriangleMesh, scenes
    from nanort_amd import dist as nd
    from nanort_amd.wire import HIT_F32

    W, H = 256, 256
    v, f = c1_mesh
    a = BVHAccel(np.float32)
    assert a.Build(f.shape[0], TriangleMesh(v, f))
    rays = scenes.camera_rays(W, H)
    want, want_mask = a.TraverseBatch(rays)  # the single-context frame
    nodes, idx = a.GetTree()
    oh, om = oracle.traverse(nodes, idx, v, f, rays)
    assert want.tobytes() == oh.tobytes() and np.array_equal(want_mask, om)

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        world, rank = dist.get_world_size(), dist.get_rank()
        d_rays = torch.from_numpy(rays.view(np.uint8)).cuda()
        bufs = [torch.empty(rays.shape[0] * HIT_F32.itemsize, dtype=torch.uint8, device="cuda") for _ in range(2)]
        outs = [nd.gather_buffer(b, world, rank) for b in bufs]
        pending = [None, None]
        for step in range(4):  # double-buffered, asynchronous: the gather of step k overlaps the trace of step k + 1
            b = step % 2
            if pending[b] is not None:
                pending[b].wait()
            bufs[b].fill_(0xCD)
            a.TraverseBatchDevice(d_rays, bufs[b])
            _, pending[b] = nd.gather_hit_records(bufs[b], world, rank, dist, out=outs[b], async_op=True)
        for p in pending:
            p.wait()
        torch.cuda.synchronize()
        assert dist.get_backend() == "nccl"
        for o in outs:
            assert o.is_cuda
            img = nd.assemble_image(o.cpu().numpy(), W, H, world, HIT_F32)
            assert img.tobytes() == want.tobytes()
        dist.barrier()
    finally:
        dist.destroy_process_group()
    maps = open("/proc/self/maps").read()
    assert "rccl" in maps, "librccl is not mapped into this process"
    assert "libnanort_hip.so" in maps

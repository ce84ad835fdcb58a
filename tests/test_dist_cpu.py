"""N > 1 path on CPU: world_size-2 gloo run of the tile partition + hit-record gather.  The CPU oracle stands
in for the kernel (tests may use it as the checker); what is under test is the sharding, the collective and
the reassembly, which are identical on the GPU path (backend "nccl" == RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nanort_amd import dist as nd
from nanort_amd import scenes
from nanort_amd.wire import HIT_F32

W, H = 96, 64


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.bindings import Oracle

    v, f = scenes.load_c1_mesh()
    orc = Oracle()
    nodes, idx, _ = orc.build(v, f)  # replicated, deterministic build on every rank
    rays = scenes.camera_rays_rows(W, H, rank, world, nd.rows_per_rank(H, rank, world))
    hits, _ = orc.traverse(nodes, idx, v, f, rays)
    local = torch.from_numpy(hits.view(np.uint8).copy())
    gathered, work = nd.gather_hit_records(local, world, rank, dist, async_op=True)
    work.wait()
    assert (gathered is None) == (rank != 0)
    if rank == 0:
        img = nd.assemble_image(gathered.numpy(), W, H, world, HIT_F32)
        np.save(out_path, img)
    dist.barrier()
    dist.destroy_process_group()


def test_rank_rows_cover_the_image_once():
    for world in (1, 2, 3, 8):
        rows = np.concatenate([nd.rank_rows(37, r, world) for r in range(world)])
        assert sorted(rows.tolist()) == list(range(37))
        assert [nd.rows_per_rank(37, r, world) for r in range(world)] == [len(nd.rank_rows(37, r, world)) for r in range(world)]


def test_interleaved_rows_generator_matches_full_frame():
    full = scenes.camera_rays(W, H).reshape(H, W)
    for world in (2, 4):
        for r in range(world):
            part = scenes.camera_rays_rows(W, H, r, world, H // world).reshape(-1, W)
            assert part.tobytes() == np.ascontiguousarray(full[r::world]).tobytes()


def test_two_rank_gloo_gather_reassembles_the_single_rank_result(tmp_path, oracle, c1_mesh):
    out = str(tmp_path / "img.npy")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    img = np.load(out)
    v, f = c1_mesh
    nodes, idx, _ = oracle.build(v, f)
    ref, _ = oracle.traverse(nodes, idx, v, f, scenes.camera_rays(W, H))
    assert img.tobytes() == ref.tobytes()

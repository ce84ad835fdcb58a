"""The timed region."""
import time

import numpy as np

PREWARM_STEPS = 150  # untimed clock ramp before the warm-up steps (see Timed; a COUNT, the same on every rank: the steps of an N > 1 run carry collectives)


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

        # Clock ramp, untimed and before the W warm-up steps: the device reaches its sustained clocks only after some tens of
        # milliseconds of traversal work (measured: K = 20 / W = 3 reads 2.3 % lower than K = 200 / W = 50 or K = 1000 / W = 100 on
        # the same box, profiles/r06s_steps_sensitivity.txt), so PREWARM_STEPS of the same steps (~0.1 s on C3) run first — whatever
        # K and W the caller asked for, the timed region then measures the steady state a renderer runs in.
        self.prewarm_steps = 0 if shared else PREWARM_STEPS  # (the test hook stages every step's records through the host: nothing to ramp)
        for _ in range(self.prewarm_steps):
            step()
        drain()
        torch.cuda.synchronize()
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


class TimedCAbi(Timed):
    """`--gather cabi`: the same K steps with the exchange done by the C ABI instead of torch.distributed — one ranked nrt_group
    (nrtGroupCreateRanked: this process owns tile `rank`; the 128-byte RCCL id is the only thing torch hands round), wave 1 through
    nrtGroupTraverseGather_* (the root GPU receives the records in FRAME order) and the ragged wave 2 through
    nrtGroupTraverseGatherTiles_* (tile-major slots of n1 records, like the torch path's padded buffers).  Everything is asynchronous
    on the group's own stream; the timed region ends with nrtGroupSynchronize on every rank.  What a C++ host would run."""

    def __init__(self, wl, steps, warmup, world, rank, dist, check_gather=False):
        import torch

        from nanort_amd import dist as nd

        accel, n1, n2, HIT, W = wl.accel, wl.n1, wl.n2, wl.HIT, wl.width
        grp = nd.RankedGroup(accel, rank, world, dist if world > 1 else None)
        if world == 1:
            grp.set_tunable("self_send", 1)  # (a world of one: the root's own records take the send / receive path, so that RCCL executes)
        assert grp.tile_rays(world * n1, W) == n1
        frame1 = [torch.empty(world * n1 * HIT.itemsize, dtype=torch.uint8, device="cuda") for _ in range(2)] if rank == 0 else [None, None]
        tiles2 = [torch.empty(world * n1 * HIT.itemsize, dtype=torch.uint8, device="cuda") for _ in range(2)] if rank == 0 else [None, None]
        step_no = [0]

        def step():
            b = step_no[0] % 2
            step_no[0] += 1
            grp.traverse_gather(wl.d_rays1, n1, world * n1, W, root=0, frame_hits=frame1[b])
            grp.traverse_gather_tiles(wl.d_rays2, n2, n1, root=0, tiles_hits=tiles2[b])

        self.prewarm_steps = PREWARM_STEPS
        for _ in range(PREWARM_STEPS + warmup):  # (the clock ramp of Timed, then the W warm-up steps)
            step()
        grp.synchronize()
        # per-wave kernel times: the same two launches without the exchange, an event pair around each (outside the timed region)
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(3)]
        for e in ev:
            e[0].record()
            accel.TraverseBatchDevice(wl.d_rays1, wl.d_hits1, wl.d_mask1)
            e[1].record()
            e[2].record()
            accel.TraverseBatchDevice(wl.d_rays2, wl.d_hits2, wl.d_mask2)
            e[3].record()
        torch.cuda.synchronize()
        self.k_ms1 = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
        self.k_ms2 = float(np.mean([e[2].elapsed_time(e[3]) for e in ev]))
        self.kernel_name = accel.LastKernelName()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        grp.synchronize()  # every exchange issued inside the timed region completes inside it
        if dist is not None:
            dist.barrier()
        self.dt = time.perf_counter() - t0
        self.region_ms = self.dt * 1e3  # (the group launches on its own stream: the wall clock of the region stands in for the event pair)
        self.steps = steps
        self.rays_per_step = n1 + n2
        self.total_rays = float(self.rays_per_step)
        self.gather_check = None
        g0 = time.perf_counter()
        grp.traverse_gather(wl.d_rays1, n1, world * n1, W, root=0, frame_hits=frame1[0])
        grp.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3
        traffic = grp.last_traffic()
        allt = np.array([[self.dt, float(self.rays_per_step), self.k_ms1, self.k_ms2, gather_ms, self.region_ms]])
        if dist is not None and world > 1:
            t = torch.tensor(allt[0], dtype=torch.float64, device="cuda")
            every = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(every, t)
            allt = torch.stack(every).cpu().numpy()
        self.dt = float(allt[:, 0].max())
        self.total_rays = float(allt[:, 1].sum())
        self.per_rank = {"wall_ms_per_step": [round(float(x) / steps * 1e3, 4) for x in allt[:, 0]],
                         "primary_kernel_ms": [round(float(x), 4) for x in allt[:, 2]],
                         "bounce_kernel_ms": [round(float(x), 4) for x in allt[:, 3]],
                         "kernel_ms_max": round(float((allt[:, 2] + allt[:, 3]).max()), 4),
                         "kernel_ms_min": round(float((allt[:, 2] + allt[:, 3]).min()), 4),
                         "launch_ms": [round(float(x) / (2 * steps), 4) for x in allt[:, 5]],
                         "trace_and_gather_ms_one_wave_blocking": [round(float(x), 4) for x in allt[:, 4]],
                         "exchange": "C ABI: nrtGroupCreateRanked + nrtGroupTraverseGather / ...Tiles (RCCL send / recv bound at run time)",
                         "last_gather_bytes": traffic}
        self.gathered_bytes_per_step = int(2 * world * n1 * HIT.itemsize)
        if check_gather:
            # the root's frame (wave 1, FRAME order) and tile slots (wave 2) against every rank's own records of plain launches
            accel.TraverseBatchDevice(wl.d_rays1, wl.d_hits1, wl.d_mask1)
            accel.TraverseBatchDevice(wl.d_rays2, wl.d_hits2, wl.d_mask2)
            torch.cuda.synchronize()
            mine1, mine2 = wl.d_hits1.contiguous(), wl.d_hits2.contiguous()
            n2s = torch.tensor([n2], dtype=torch.int64, device="cuda")
            e1, e2, en = [mine1], [mine2], [n2s]
            if dist is not None and world > 1:
                e1 = [torch.empty_like(mine1) for _ in range(world)]
                e2 = [torch.empty_like(mine2) for _ in range(world)]
                en = [torch.empty_like(n2s) for _ in range(world)]
                dist.all_gather(e1, mine1)
                dist.all_gather(e2, mine2)
                dist.all_gather(en, n2s)
            if rank == 0:
                last = (step_no[0] - 1) % 2
                rows = n1 // W
                want = np.empty((rows * world, W), dtype=HIT)
                for r in range(world):
                    want[r::world] = e1[r].cpu().numpy().view(HIT).reshape(rows, W)
                got = frame1[last].cpu().numpy().view(HIT)
                fields = ("t", "u", "v", "prim_id")  # (field by field: the fp64 record ends in padding no launch defines)
                frame_ok = all(got[f].tobytes() == want.reshape(-1)[f].tobytes() for f in fields)
                t2 = tiles2[last].cpu().numpy().view(HIT).reshape(world, n1)
                tiles_ok = all(t2[r][: int(en[r].item())][f].tobytes() == e2[r].cpu().numpy().view(HIT)[: int(en[r].item())][f].tobytes()
                               for r in range(world) for f in fields)
                self.gather_check = {"waves": "primary (frame order) and bounce (tile slots), last timed step", "records": int(got.shape[0]),
                                     "frame_identical_to_the_ranks_records": bool(frame_ok), "tile_slots_identical_to_the_ranks_records": bool(tiles_ok),
                                     "device_tensors": True, "backend": "C ABI (librccl bound at run time)"}
        self.value = self.total_rays * steps / self.dt / 1e6
        self.ms_per_step = self.dt / steps * 1e3
        grp.close()

"""The timed region."""
import time

import numpy as np

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

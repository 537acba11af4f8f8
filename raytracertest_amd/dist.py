"""Row-band sharding of one frame across the GPUs of a node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo"
on CPU for the tests).  Pixels are independent and every pixel's RNG stream is keyed by
its GLOBAL index (subsequence = x + y*W, RayTracer/Random.cu:21-27), so the frame is
invariant to the partition: rank g of G owns the contiguous rows [g*H/G, (g+1)*H/G) and
traces them with no exchange at all.  The only collective is the gather of the finished
BGRA8 tiles to rank 0 (the hand-off the reference does through its callback,
RayTracerImpl.cu:287-305).  On the fully connected xGMI mesh every peer has its own link
to the root, so the G-1 tile transfers run in parallel; a ring all-reduce would be the
wrong primitive (nothing is reduced).

The gather of frame i runs on torch's stream while the tracer's own HIP stream already
traces frame i+1 (two tile buffers, ordered with events through
torch.cuda.ExternalStream): the host never blocks inside the loop.

PyTorch is plumbing here (process group, device tensors for the gather); the compute is
the C-ABI library.
"""
import numpy as np


def band_rows(height, world, rank):
    """Rows [begin, begin+count) of rank `rank` of `world`: contiguous, balanced to one row."""
    begin = (height * rank) // world
    end = (height * (rank + 1)) // world
    return begin, end - begin


def gather_tiles(tile, world, rank, group=None, all_rows=None, out=None):
    """Gather equal-or-ragged row tiles (torch tensors, (rows_i, W)) to rank 0.

    Returns the stacked (sum rows_i, W) tensor on rank 0, None elsewhere.  Ragged bands
    are padded to the tallest band for the collective and cropped afterwards.
    all_rows: the band heights of all ranks if already known (skips the size exchange);
    out: preallocated list of `world` receive tensors on rank 0 (skips the allocation)."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return tile
    if all_rows is None:
        rows = torch.tensor([tile.shape[0]], dtype=torch.int64, device=tile.device)
        sizes = [torch.zeros_like(rows) for _ in range(world)]
        dist.all_gather(sizes, rows, group=group)
        all_rows = [int(r.item()) for r in sizes]
    tallest = max(all_rows)
    if tile.shape[0] != tallest:
        pad = torch.zeros((tallest - tile.shape[0],) + tuple(tile.shape[1:]), dtype=tile.dtype, device=tile.device)
        tile = torch.cat([tile, pad], dim=0)
    tile = tile.contiguous()
    if rank == 0 and out is None:
        out = [torch.empty_like(tile) for _ in range(world)]
    dist.gather(tile, out if rank == 0 else None, dst=0, group=group)
    if rank != 0:
        return None
    if len(set(all_rows)) == 1:
        return torch.cat(out, dim=0) if out[0].shape[0] == all_rows[0] else torch.cat([t[:all_rows[0]] for t in out], dim=0)
    return torch.cat([t[:r] for t, r in zip(out, all_rows)], dim=0)


def update_due(i, update_interval, have_callback=True):
    """The reference's update cadence, RayTracerImpl.cu:256."""
    return have_callback and i > 0 and update_interval > 0 and i % update_interval == 0


def progressive_trace(launch, tile, world, rank, iterations, samples, update_interval,
                      on_update=None, on_finished=None, stop_requested=None, all_reduce_max=None,
                      have_update_callback=None, fuse=1):
    """RayTracerImpl::TraceFunct (RayTracerImpl.cu:236-315) for a frame sharded in row bands.

    Every rank runs the same iteration loop on its own band; at an update iteration and at
    the end the BGRA8 tiles are gathered to rank 0, which fires the callbacks with the whole
    frame -- the per-update hand-off of the reference (:259-272,:287-305) at multi-GPU scale.
    Cancel granularity is one launch, as in the reference (:248): rank 0's stop request is
    agreed on by all ranks with a 1-element MAX all-reduce before each launch, so that no
    rank leaves the loop (and its collectives) alone.  A stopped run fires no finished
    callback (:280-284).  Returns True when the run completed.

    fuse > 1: iterations nobody observes in between (up to the next update point, at most `fuse`)
    run as one launch, `launch(samples, clear_first, emit_image, n)` -- same bits, one stop
    agreement and one state round trip per group (RayTracer.Launch(iterations=n)).

      launch(samples, clear_first, emit_image[, n])  -> enqueue one launch on this rank's band
      tile()                                    -> this rank's finished (rows, W) BGRA8 tensor
      all_reduce_max(flag: int) -> int          -> max of `flag` over ranks (identity if world == 1)
      have_update_callback                      -> must agree on all ranks (default: on_update given)
    """
    if all_reduce_max is None:
        def all_reduce_max(v):
            return v
    updates_on = (on_update is not None) if have_update_callback is None else bool(have_update_callback)

    def stop_agreed():
        want = 1 if (rank == 0 and stop_requested is not None and stop_requested()) else 0
        return bool(all_reduce_max(want))

    i = 0
    while i < iterations:
        if stop_agreed():
            return False
        e = i                                              # last iteration of this launch
        while e < iterations - 1 and e - i + 1 < fuse and not update_due(e, update_interval, updates_on):
            e += 1
        upd = update_due(e, update_interval, updates_on)
        if fuse > 1:
            launch(samples, i == 0, upd or e + 1 == iterations, e - i + 1)
        else:
            launch(samples, i == 0, upd or e + 1 == iterations)
        if upd:
            frame = gather_tiles(tile(), world, rank)
            if rank == 0 and on_update is not None:
                on_update(frame)
        i = e + 1
    if stop_agreed():
        return False
    if iterations == 0:
        return True
    frame = gather_tiles(tile(), world, rank)
    if rank == 0 and on_finished is not None:
        on_finished(frame)
    return True


class RowBandJob:
    """One rank's share of a frame: a tracer on its band + the tile gather.

    weak=True  : the frame is W x (H*world) and every rank owns H rows (fixed per-GPU work).
    weak=False : the frame is W x H split into `world` bands (strong scaling)."""

    def __init__(self, cfg, tris, spheres, world=1, rank=0, local_rank=0, weak=True,
                 samples_in_flight=0, lds_chunk=0, math_mode=0):
        import raytracertest_amd as R
        self.cfg, self.world, self.rank = cfg, world, rank
        self.torch = None
        if world > 1:
            import torch
            import torch.distributed as dist
            self.torch, self.dist = torch, dist
            # Rehearsal knobs (a 1-GPU box cannot host two RCCL ranks): RT_DIST_BACKEND=gloo stages the
            # tiles through host memory for the collective, RT_DIST_SHARE_GPU=1 lets all ranks use
            # the devices round-robin.  The product path is nccl (RCCL), one GPU per rank.
            import os
            self.backend = os.environ.get("RT_DIST_BACKEND", "nccl")
            if os.environ.get("RT_DIST_SHARE_GPU", "") == "1":
                local_rank = local_rank % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local_rank)
            if not dist.is_initialized():
                if self.backend == "nccl":
                    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
                else:
                    dist.init_process_group(backend=self.backend)
        W, H = cfg["width"], cfg["height"]
        if weak:
            full_h, row0, rows = H * world, H * rank, H
            self.all_rows = [H] * world
        else:
            full_h = H
            row0, rows = band_rows(H, world, rank)
            self.all_rows = [band_rows(H, world, r)[1] for r in range(world)]
        self.full_height, self.row0, self.rows = full_h, row0, rows
        self.tracer = R.RayTracer((W, rows), (0.0, 0.0, 0.0), cfg["angles"], cfg["fov"], cfg["focal"],
                                  cfg["aperture"], seed=cfg["seed"], device=local_rank, math_mode=math_mode,
                                  full_height=full_h if world > 1 else 0, row_begin=row0,
                                  samples_in_flight=samples_in_flight, lds_chunk=lds_chunk)
        if tris.shape[0]:
            assert self.tracer.UploadScene(tris)
        if spheres.shape[0]:
            self.tracer.UploadSpheres(spheres)
        self.frame = None
        self.step_index = 0
        if world > 1:
            t = self.torch
            tallest = max(self.all_rows)
            # two tile buffers: frame i is gathered from one while frame i+1 is copied into the other
            self.tiles = [t.zeros((tallest, W), dtype=t.int32, device="cuda") for _ in range(2)]
            self.recv = [[t.empty((tallest, W), dtype=t.int32, device="cuda") for _ in range(world)] for _ in range(2)] \
                if rank == 0 else [None, None]
            dev = t.device("cuda", local_rank)
            # the tracer's own HIP streams: trace launches run as two half-frame kernels on two streams
            self.trace_streams = [t.cuda.ExternalStream(self.tracer.Stream(), device=dev),
                                  t.cuda.ExternalStream(self.tracer.StreamB(), device=dev)]
            self.copied = [[t.cuda.Event() for _ in range(2)] for _ in range(2)]   # tile i written, one event per stream
            self.gathered = [t.cuda.Event() for _ in range(2)]    # tile i consumed by the gather (torch's stream)

    # ---- throughput path (bench.py) -------------------------------------------------------
    def step(self):
        """One Trace pass on device-resident buffers (+ tile gather when sharded).  Nothing here
        blocks the host: the trace kernel writes its BGRA8 tile straight into the gather's send
        buffer, the gather is ordered behind the trace through an event, and buffer reuse behind
        the previous gather."""
        cfg = self.cfg
        if self.world == 1:
            self.tracer.TraceEnqueue(cfg["iterations"], cfg["samples"])
            return
        from .api import BUF_IMAGE
        b = self.step_index & 1
        self.step_index += 1
        for s in self.trace_streams:
            s.wait_event(self.gathered[b])                        # buffer b is free again (no-op the first time)
        self.tracer.SetImageMirror(self.tiles[b].data_ptr())      # the kernel writes the send buffer itself: no copy
        self.tracer.TraceEnqueue(cfg["iterations"], cfg["samples"])
        cur = self.torch.cuda.current_stream()
        for ev, s in zip(self.copied[b], self.trace_streams):     # the gather waits for both halves, neither stream waits
            ev.record(s)
            cur.wait_event(ev)
        if self.backend == "nccl":
            self.dist.gather(self.tiles[b], self.recv[b] if self.rank == 0 else None, dst=0)
        else:                                                    # rehearsal: collective on host copies
            host = self.tiles[b].cpu()
            parts = [self.torch.empty_like(host) for _ in range(self.world)] if self.rank == 0 else None
            self.dist.gather(host, parts, dst=0)
            if self.rank == 0:
                for dst, src in zip(self.recv[b], parts):
                    dst.copy_(src)
        self.gathered[b].record(cur)
        self.last_buffer = b

    def finish(self):
        self.tracer.Sync()
        if self.world > 1:
            self.torch.cuda.synchronize()

    def gathered_image(self):
        """(full rows, W) uint32 BGRA8 on rank 0 after finish() (None elsewhere)."""
        if self.world == 1:
            return self.tracer.Image()
        if self.rank != 0 or self.step_index == 0:
            return None
        t = self.torch
        parts = [buf[:r] for buf, r in zip(self.recv[self.last_buffer], self.all_rows)]
        return t.cat(parts, dim=0).cpu().numpy().view(np.uint32)

    # ---- progressive path (callbacks on rank 0) ---------------------------------------------
    def _tile(self):
        """Finished BGRA8 band as a torch tensor ready for the collective (host-syncs the tracer)."""
        from .api import BUF_IMAGE
        if self.world == 1:
            import torch
            return torch.from_numpy(self.tracer.Image().view(np.int32))
        tile = self.tiles[0][:self.rows]
        self.tracer.CopyToDevice(BUF_IMAGE, tile.data_ptr(), self.rows * self.cfg["width"] * 4)
        return tile if self.backend == "nccl" else tile.cpu()

    def trace_progressive(self, iterations, samples, update_interval, on_update=None, on_finished=None,
                          stop_requested=None):
        """Multi-GPU Trace with the reference's callback cadence; callbacks run on rank 0 with
        the gathered (full rows, W) frame.  Whether updates happen at all is rank 0's choice
        (it passes on_update), agreed on by all ranks."""
        updates_on = bool(self._all_reduce_max(1 if (self.rank == 0 and on_update is not None) else 0))

        fuse = self.tracer.FusedIterations(samples)            # same on every rank (same build, same options)

        def launch(spp, clear_first, emit, n=1):
            self.tracer.Launch(spp, clear_first, emit, iterations=n)

        ok = progressive_trace(launch, self._tile, self.world, self.rank, iterations, samples, update_interval,
                               on_update=on_update if self.rank == 0 else None, on_finished=on_finished,
                               stop_requested=stop_requested, all_reduce_max=self._all_reduce_max,
                               have_update_callback=updates_on, fuse=fuse)
        self.tracer.Sync()
        return ok

    def _all_reduce_max(self, v):
        if self.world == 1:
            return v
        t = self.torch.tensor([int(v)], dtype=self.torch.int32, device="cuda" if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return int(t.item())

    # ---- bench plumbing ----------------------------------------------------------------------
    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        if self.world == 1:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device="cuda" if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        self.tracer.close()
        if self.world > 1 and self.dist.is_initialized():
            self.dist.destroy_process_group()

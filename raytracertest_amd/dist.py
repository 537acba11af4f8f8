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

PyTorch is plumbing here (process group, device tensors for the gather); the compute is
the C-ABI library.
"""
import numpy as np


def band_rows(height, world, rank):
    """Rows [begin, begin+count) of rank `rank` of `world`: contiguous, balanced to one row."""
    begin = (height * rank) // world
    end = (height * (rank + 1)) // world
    return begin, end - begin


def gather_tiles(tile, world, rank, group=None):
    """Gather equal-or-ragged row tiles (torch tensors, (rows_i, W)) to rank 0.

    Returns the stacked (sum rows_i, W) tensor on rank 0, None elsewhere.  Ragged bands
    are padded to the tallest band for the collective and cropped afterwards."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return tile
    rows = torch.tensor([tile.shape[0]], dtype=torch.int64, device=tile.device)
    all_rows = [torch.zeros_like(rows) for _ in range(world)]
    dist.all_gather(all_rows, rows, group=group)
    all_rows = [int(r.item()) for r in all_rows]
    tallest = max(all_rows)
    if tile.shape[0] != tallest:
        pad = torch.zeros((tallest - tile.shape[0],) + tuple(tile.shape[1:]), dtype=tile.dtype, device=tile.device)
        tile = torch.cat([tile, pad], dim=0)
    tile = tile.contiguous()
    out = [torch.empty_like(tile) for _ in range(world)] if rank == 0 else None
    dist.gather(tile, out, dst=0, group=group)
    if rank != 0:
        return None
    return torch.cat([t[:r] for t, r in zip(out, all_rows)], dim=0)


def update_due(i, update_interval, have_callback=True):
    """The reference's update cadence, RayTracerImpl.cu:256."""
    return have_callback and i > 0 and update_interval > 0 and i % update_interval == 0


def progressive_trace(launch, tile, world, rank, iterations, samples, update_interval,
                      on_update=None, on_finished=None, stop_requested=None, all_reduce_max=None):
    """RayTracerImpl::TraceFunct (RayTracerImpl.cu:236-315) for a frame sharded in row bands.

    Every rank runs the same iteration loop on its own band; at an update iteration and at
    the end the BGRA8 tiles are gathered to rank 0, which fires the callbacks with the whole
    frame -- the per-update hand-off of the reference (:259-272,:287-305) at multi-GPU scale.
    Cancel granularity is one iteration, as in the reference (:248): rank 0's stop request is
    agreed on by all ranks with a 1-element MAX all-reduce before each launch, so that no
    rank leaves the loop (and its collectives) alone.  A stopped run fires no finished
    callback (:280-284).  Returns True when the run completed.

      launch(samples, clear_first, emit_image)  -> enqueue one launch on this rank's band
      tile()                                    -> this rank's finished (rows, W) BGRA8 tensor
      all_reduce_max(flag: int) -> int          -> max of `flag` over ranks (identity if world == 1)
    """
    if all_reduce_max is None:
        def all_reduce_max(v):
            return v
    for i in range(iterations):
        want_stop = 1 if (rank == 0 and stop_requested is not None and stop_requested()) else 0
        if all_reduce_max(want_stop):
            return False
        upd = update_due(i, update_interval, on_update is not None)
        launch(samples, i == 0, upd or i + 1 == iterations)
        if upd:
            frame = gather_tiles(tile(), world, rank)
            if rank == 0 and on_update is not None:
                on_update(frame)
    want_stop = 1 if (rank == 0 and stop_requested is not None and stop_requested()) else 0
    if all_reduce_max(want_stop):
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
        self.image_t = None
        if world > 1:
            import torch
            import torch.distributed as dist
            self.torch, self.dist = torch, dist
            torch.cuda.set_device(local_rank)
            if not dist.is_initialized():
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        W, H = cfg["width"], cfg["height"]
        if weak:
            full_h, row0, rows = H * world, H * rank, H
        else:
            full_h = H
            row0, rows = band_rows(H, world, rank)
        self.full_height, self.row0, self.rows = full_h, row0, rows
        self.tracer = R.RayTracer((W, rows), (0.0, 0.0, 0.0), cfg["angles"], cfg["fov"], cfg["focal"],
                                  cfg["aperture"], seed=cfg["seed"], device=local_rank, math_mode=math_mode,
                                  full_height=full_h if world > 1 else 0, row_begin=row0,
                                  samples_in_flight=samples_in_flight, lds_chunk=lds_chunk)
        if tris.shape[0]:
            assert self.tracer.UploadScene(tris)
        if spheres.shape[0]:
            self.tracer.UploadSpheres(spheres)
        if world > 1:
            self.image_t = self.torch.empty((rows, W), dtype=self.torch.int32, device="cuda")
        self.frame = None

    def step(self):
        """One Trace pass on device-resident buffers (+ tile gather when sharded)."""
        cfg = self.cfg
        self.tracer.TraceEnqueue(cfg["iterations"], cfg["samples"])
        if self.world > 1:
            from .api import BUF_IMAGE
            # hipMemcpyAsync on the tracer's stream + stream sync, then RCCL on torch's stream
            self.tracer.CopyToDevice(BUF_IMAGE, self.image_t.data_ptr(), self.image_t.numel() * 4)
            self.frame = gather_tiles(self.image_t, self.world, self.rank)

    def _tile(self):
        """Finished BGRA8 band as a torch tensor ready for the collective (host-syncs the tracer)."""
        from .api import BUF_IMAGE
        if self.world == 1:
            import torch
            return torch.from_numpy(self.tracer.Image().view(np.int32))
        self.tracer.CopyToDevice(BUF_IMAGE, self.image_t.data_ptr(), self.image_t.numel() * 4)
        return self.image_t

    def trace_progressive(self, iterations, samples, update_interval, on_update=None, on_finished=None,
                          stop_requested=None, have_update_callback=None):
        """Multi-GPU Trace with the reference's callback cadence; callbacks run on rank 0 with
        the gathered (full rows, W) frame.  `have_update_callback` must be the same on every
        rank (defaults to: rank 0 passes on_update)."""
        flag = on_update is not None if have_update_callback is None else have_update_callback
        if self.world > 1 and have_update_callback is None:
            flag = bool(self._all_reduce_max(1 if on_update is not None else 0))
        def launch(spp, clear_first, emit):
            self.tracer.Launch(spp, clear_first, emit)
        ok = progressive_trace(launch, self._tile, self.world, self.rank, iterations, samples, update_interval,
                               on_update=on_update if self.rank == 0 else (None if not flag else (lambda f: None)),
                               on_finished=on_finished, stop_requested=stop_requested,
                               all_reduce_max=self._all_reduce_max)
        self.tracer.Sync()
        return ok

    def _all_reduce_max(self, v):
        if self.world == 1:
            return v
        t = self.torch.tensor([int(v)], dtype=self.torch.int32, device="cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return int(t.item())

    def finish(self):
        self.tracer.Sync()
        if self.world > 1:
            self.torch.cuda.synchronize()

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        if self.world == 1:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device="cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gathered_image(self):
        """(full rows, W) uint32 BGRA8 on rank 0 after the last step (None elsewhere)."""
        if self.world == 1:
            return self.tracer.Image()
        if self.frame is None:
            return None
        return self.frame.cpu().numpy().view(np.uint32)

    def close(self):
        self.tracer.close()
        if self.world > 1 and self.dist.is_initialized():
            self.dist.destroy_process_group()

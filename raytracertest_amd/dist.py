"""Row-band sharding of one frame across the GPUs of a node, host side.

Pixels are independent and every pixel's RNG stream is keyed by its GLOBAL index (subsequence =
x + y*W, RayTracer/Random.cu:21-27), so the frame is invariant to the partition: band g of G owns
the contiguous rows [g*H/G, (g+1)*H/G) and is traced with no exchange at all.  The only data
movement is the hand-off the reference does through its callback (RayTracerImpl.cu:287-305): the
finished BGRA8 tiles travel to the root.  That gather is NATIVE -- librt_mi355x calls RCCL itself
(grouped ncclSend/ncclRecv over xGMI, csrc/rt_multi.hpp) on gather streams of its own, ordered
behind the trace streams by events, double-buffered so that frame i is gathered while frame i+1 is
traced.  Two ways to drive it:

  * one process, several devices (`devices=[...]`): rt_tracer_create_multi -- the library owns the bands,
    one host thread per device, the reference's API unchanged (what `python bench.py --gpus N` uses);
  * one process per GPU (torch.distributed.run): every rank creates the tracer of its band and joins
    the group (rt_tracer_join_group); the launcher's rendezvous only carries the 128-byte id and the
    control-plane scalars (barrier, max, stop agreement) -- a gloo process group, no GPU tensors.

PyTorch is plumbing here; the compute and the collective are the C-ABI library's.
"""
import numpy as np


def band_rows(height, world, rank):
    """Rows [begin, begin+count) of rank `rank` of `world`: contiguous, balanced to one row
    (the same partition rt_tracer_create_multi / rt_tracer_join_group use)."""
    begin = (height * rank) // world
    end = (height * (rank + 1)) // world
    return begin, end - begin


def gather_tiles(tile, world, rank, group=None, all_rows=None, out=None):
    """Gather equal-or-ragged row tiles (torch tensors, (rows_i, W)) to rank 0 through torch.distributed --
    the exchange of the CPU tests (gloo); the GPU path gathers natively (see the module docstring).

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
                      have_update_callback=None, fuse=1, gather=None):
    """RayTracerImpl::TraceFunct (RayTracerImpl.cu:236-315) for a frame sharded in row bands over RANKS.

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
      tile()                                    -> this rank's finished tile, handed to `gather`
      gather(tile) -> frame on rank 0 / None    -> default: gather_tiles over torch.distributed
      all_reduce_max(flag: int) -> int          -> max of `flag` over ranks (identity if world == 1)
      have_update_callback                      -> must agree on all ranks (default: on_update given)
    """
    if all_reduce_max is None:
        def all_reduce_max(v):
            return v
    if gather is None:
        def gather(t):
            return gather_tiles(t, world, rank)
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
            frame = gather(tile())
            if rank == 0 and on_update is not None:
                on_update(frame)
        i = e + 1
    if stop_agreed():
        return False
    if iterations == 0:
        return True
    frame = gather(tile())
    if rank == 0 and on_finished is not None:
        on_finished(frame)
    return True


class NativeExchange:
    """The product exchange of one-process-per-GPU jobs: the band tracer joins the library's group
    (rt_tracer_join_group, RCCL); from then on every emitting TraceEnqueue/Launch carries its own gather."""

    def attach(self, job, row_begin=None):
        uid = [None]
        if job.rank == 0 and job.world > 1:
            from .api import group_unique_id
            uid[0] = group_unique_id()
        if job.world > 1:
            job.dist.broadcast_object_list(uid, src=0)      # 128 bytes through the launcher's rendezvous
        # The join is collective (ncclCommInitRank): a rank that cannot reach the others would block in it for ever.
        # It runs on a helper thread so that this one can give up with a message instead (RT_MI355X_JOIN_TIMEOUT seconds).
        import os
        import threading
        done, failure = threading.Event(), []

        def join():
            try:
                job.tracer.JoinGroup(job.world, job.rank, uid[0], row_begin=row_begin)
            except Exception as e:                          # handed to the waiting thread
                failure.append(e)
            done.set()

        threading.Thread(target=join, daemon=True).start()
        limit = float(os.environ.get("RT_MI355X_JOIN_TIMEOUT", "300"))
        if not done.wait(limit):
            raise TimeoutError("rank %d: rt_tracer_join_group did not return within %.0f s (RCCL communicator of %d ranks)"
                               % (job.rank, limit, job.world))
        if failure:
            raise failure[0]

    def detach(self, job):
        job.tracer.LeaveGroup()

    def after_emit(self, job):
        pass                                                # the library enqueued the gather behind the launch

    def frame(self, job):
        """(H, W) uint32 on rank 0 (waits for its own streams and the gather), None elsewhere."""
        if job.rank != 0:
            job.tracer.Sync()
            return None
        return job.tracer.Frame()


class RowBandJob:
    """One frame in row bands + the tile gather, in one of three shapes:

      world == 1, devices None   one tracer, the whole frame (single GPU);
      world == 1, devices [...]  rt_tracer_create_multi: band k on devices[k], everything inside the library;
      world  > 1                 this process is rank `rank`: a band tracer that joined the group.

    weak=True  : the frame is W x (H*G) and every band has H rows (fixed per-GPU work), G = bands or ranks;
    weak=False : the frame is W x H split into G bands (strong scaling).
    exchange / tracer_factory are seams for the tests (a CPU stand-in for the tracer, a host-staged exchange)."""

    def __init__(self, cfg, tris, spheres, world=1, rank=0, local_rank=0, weak=True, devices=None,
                 samples_in_flight=0, lds_chunk=0, math_mode=0, exchange=None, tracer_factory=None):
        self.cfg, self.world, self.rank = cfg, world, rank
        self.dist = None
        W, H = cfg["width"], cfg["height"]
        parts = world if world > 1 else (len(devices) if devices else 1)
        self.parts = parts
        full_h = H * parts if weak else H
        self.full_height = full_h
        if world > 1:
            import torch.distributed as dist
            self.dist = dist
            if not dist.is_initialized():
                dist.init_process_group(backend="gloo")     # control plane only: id, barrier, max, stop agreement
            row0, rows = band_rows(full_h, world, rank)
        else:
            row0, rows = 0, full_h
        self.row0, self.rows = row0, rows
        if tracer_factory is None:
            import raytracertest_amd as R

            def tracer_factory(**kw):
                return R.RayTracer((W, rows), (0.0, 0.0, 0.0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"],
                                   seed=cfg["seed"], math_mode=math_mode, samples_in_flight=samples_in_flight,
                                   lds_chunk=lds_chunk, **kw)
        if world > 1:
            self.tracer = tracer_factory(device=local_rank, full_height=full_h, row_begin=row0)
        elif devices:
            self.tracer = tracer_factory(devices=list(devices))
        else:
            self.tracer = tracer_factory(device=local_rank)
        if tris.shape[0]:
            assert self.tracer.UploadScene(tris)
        if spheres.shape[0]:
            self.tracer.UploadSpheres(spheres)
        self.exchange = None
        if world > 1:
            self.exchange = exchange if exchange is not None else NativeExchange()
            self.exchange.attach(self)

    # ---- throughput path (bench.py) -------------------------------------------------------
    def step(self):
        """One Trace pass on device-resident buffers + the gather of the finished tiles when the frame is
        sharded.  Nothing here blocks the host: the trace kernel writes its tile straight into the gather's
        buffer, the gather runs on its own stream behind an event while the next step already traces."""
        self.tracer.TraceEnqueue(self.cfg["iterations"], self.cfg["samples"])
        if self.exchange is not None:
            self.exchange.after_emit(self)

    def steps(self, n):
        """n consecutive step()s.  With the library's own exchange (or none) the loop runs inside the library
        (rt_tracer_trace_enqueue_n): one call, no per-step crossing of ctypes."""
        if self.exchange is None or isinstance(self.exchange, NativeExchange):
            if n > 0:
                self.tracer.TraceEnqueueN(self.cfg["iterations"], self.cfg["samples"], n)
            return
        for _ in range(n):
            self.step()

    def finish(self):
        self.tracer.Sync()

    def gathered_image(self):
        """(full rows, W) uint32 BGRA8 of the last step on rank 0 (None elsewhere)."""
        if self.world == 1:
            return self.tracer.Image()
        return self.exchange.frame(self)

    # ---- progressive path (callbacks on rank 0) ---------------------------------------------
    def trace_progressive(self, iterations, samples, update_interval, on_update=None, on_finished=None,
                          stop_requested=None):
        """Multi-rank Trace with the reference's callback cadence; callbacks run on rank 0 with the gathered
        (full rows, W) uint32 frame.  Whether updates happen at all is rank 0's choice, agreed on by all ranks.
        (A single-process job calls tracer.Trace() instead: the library runs this loop itself.)"""
        updates_on = bool(self._all_reduce_max(1 if (self.rank == 0 and on_update is not None) else 0))
        fuse = self.tracer.FusedIterations(samples)            # same on every rank (same build, same options)

        def launch(spp, clear_first, emit, n=1):
            self.tracer.Launch(spp, clear_first, emit, iterations=n)
            if emit and self.exchange is not None:
                self.exchange.after_emit(self)

        ok = progressive_trace(launch, lambda: None, self.world, self.rank, iterations, samples, update_interval,
                               on_update=on_update if self.rank == 0 else None, on_finished=on_finished,
                               stop_requested=stop_requested, all_reduce_max=self._all_reduce_max,
                               have_update_callback=updates_on, fuse=fuse, gather=lambda _t: self.gathered_image())
        self.tracer.Sync()
        return ok

    def _all_reduce_max(self, v):
        if self.world == 1:
            return v
        import torch
        t = torch.tensor([int(v)], dtype=torch.int32)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return int(t.item())

    # ---- load balance (strong scaling of scenes whose cost is not uniform over the rows) ------------
    def rebalance(self):
        """Re-partition the rows so that every band costs the same, from the bands' kernel times since the last
        KernelTime reset (collective in a multi-rank job).  Buffers and RNG states of the bands are re-created, as
        by Resize; the image of a Trace does not depend on the partition.  Returns the rows per band."""
        if self.world == 1:
            if self.parts > 1:
                self.tracer.Rebalance()
            return [b["rows"] for b in self.tracer.Bands()]
        import torch
        from .api import balance_rows
        ms, n = self.tracer.LaunchTime(reset=True)          # split launches count to the end of their later half
        mine = torch.tensor([ms / max(n, 1), float(self.row0), float(self.rows), float(n)], dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(every, mine)
        idle = [k for k, t in enumerate(every) if t[3].item() == 0]
        if idle:                                            # (same verdict on every rank: nobody is left in a collective)
            raise RuntimeError("rebalance: ranks %s have no sampled launch since the last reset (a cost of 0 would shrink their "
                               "bands to one granule)" % idle)
        begins = [int(t[1].item()) for t in every] + [self.full_height]
        fresh = balance_rows(begins, [float(t[0].item()) for t in every], 8)
        if fresh != begins:
            self.exchange.detach(self)
            self.row0, self.rows = fresh[self.rank], fresh[self.rank + 1] - fresh[self.rank]
            self.tracer.SetBand(self.row0, self.rows)
            self.exchange.attach(self, row_begin=fresh)
        return [fresh[k + 1] - fresh[k] for k in range(self.world)]

    # ---- bench plumbing ----------------------------------------------------------------------
    def barrier(self):
        """All ranks have drained their device work and reached this point."""
        self.tracer.Sync()
        if self.world > 1:
            self.dist.barrier()

    def max_over_ranks(self, seconds):
        if self.world == 1:
            return seconds
        import torch
        t = torch.tensor([seconds], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self, destroy_group=True):
        self.tracer.close()
        if destroy_group and self.world > 1 and self.dist.is_initialized():
            self.dist.destroy_process_group()

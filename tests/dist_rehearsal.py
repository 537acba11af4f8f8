#!/usr/bin/env python3
"""Rehearsal of the one-process-per-GPU job on a box with ONE GPU: two ranks (torch.distributed.run) share
device 0.  RCCL refuses two ranks on one device, so the tiles travel through the test's HostStagedExchange
(tests/dist_helpers.py) instead of the library's group; everything else is the product path of
raytracertest_amd.dist.RowBandJob -- band geometry, gloo control plane, band tracers with global-row RNG keys,
progressive driver.  Checks on rank 0 that the gathered frame equals a single tracer on the whole frame and that
the progressive driver fires the reference's callback cadence.  Launched by tests/test_gpu_parity.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import raytracertest_amd as R
from raytracertest_amd import scenes
from raytracertest_amd.dist import RowBandJob
from dist_helpers import HostStagedExchange

world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
cfg = dict(width=64, height=24, iterations=2, samples=3, angles=(0.0, 0.0), fov=70.0, focal=3.0, aperture=0.05, seed=9)
tris = scenes.cornell32()
for weak in (True, False):
    job = RowBandJob(cfg, tris, np.zeros((0, 4), np.float32), world=world, rank=rank, local_rank=0, weak=weak,
                     exchange=HostStagedExchange())
    for _ in range(3):
        job.step()
    job.finish()
    frame = job.gathered_image()
    job.barrier()
    t = job.max_over_ranks(0.5 + rank)
    assert abs(t - (0.5 + world - 1)) < 1e-9
    if rank == 0:
        H = cfg["height"] * world if weak else cfg["height"]
        g = R.RayTracer((cfg["width"], H), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=cfg["seed"], device=0)
        g.UploadScene(tris)
        for _ in range(3):                 # same three Trace passes (RNG streams continue across passes)
            g.Trace(cfg["iterations"], cfg["samples"], 0); assert g.Wait()
        assert frame is not None and frame.shape == (H, cfg["width"]), (None if frame is None else frame.shape)
        assert np.array_equal(frame, g.Image()), "gathered frame differs (weak=%s)" % weak
        g.close()
    else:
        assert frame is None
    updates, finished = [], []
    ok = job.trace_progressive(5, 1, 2, on_update=(lambda f: updates.append(1)) if rank == 0 else None,
                               on_finished=lambda f: finished.append(np.array(f, copy=True)))
    assert ok
    if rank == 0:
        assert len(updates) == 2 and len(finished) == 1 and finished[0].shape[1] == cfg["width"]
    job.barrier()
    job.close(destroy_group=False)
import torch.distributed as dist
dist.destroy_process_group()
if rank == 0:
    print("dist_rehearsal ok: world=%d" % world)

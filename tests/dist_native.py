#!/usr/bin/env python3
"""The one-process-per-GPU job on a box with at least two GPUs: rank r drives device r, the tiles travel through the LIBRARY's
group (rt_tracer_join_group: ncclCommInitRank + grouped ncclSend / ncclRecv over xGMI) -- raytracertest_amd.dist.RowBandJob with
its default NativeExchange, nothing staged by the test.  Rank 0 compares the gathered frame with the oracle's whole frame and
prints what the communicator is made of.  Launched by tests/test_gpu_round3.py (skipped on boxes with one GPU)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import raytracertest_amd as R
from raytracertest_amd import scenes
from raytracertest_amd.dist import RowBandJob
from oracle import oracle_py as orc

world, rank, local = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", os.environ["RANK"]))
assert R.device_count() >= world, "dist_native.py needs one GPU per rank"
cfg = dict(width=96, height=40, iterations=2, samples=3, angles=(0.0, 0.0), fov=70.0, focal=3.0, aperture=0.05, seed=9)
tris = scenes.cornell32()
for weak in (True, False):
    job = RowBandJob(cfg, tris, np.zeros((0, 4), np.float32), world=world, rank=rank, local_rank=local, weak=weak)
    info = job.tracer.GroupInfo()
    assert info["transport"] == "rccl" and info["ranks"] == world, info
    assert info["communicators"][0]["ranks_in_communicator"] == world and info["communicators"][0]["rank"] == rank, info
    for _ in range(3):
        job.step()
    job.finish()
    frame = job.gathered_image()
    job.tracer.GatherOnly()
    job.finish()
    job.barrier()
    if rank == 0:
        H = cfg["height"] * world if weak else cfg["height"]
        o = orc.OracleTracer(cfg["width"], H, cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=cfg["seed"], nthreads=4)
        o.upload_scene(tris)
        for _ in range(3):                 # same three Trace passes (RNG streams continue across passes)
            o.trace(cfg["iterations"], cfg["samples"])
        assert frame is not None and frame.shape == (H, cfg["width"])
        assert np.array_equal(frame, o.image), "gathered frame differs from the oracle (weak=%s)" % weak
        ms, n = job.tracer.GatherTime()
        assert n >= 4 and ms > 0.0, (ms, n)
        print("dist_native: weak=%s frame %dx%d == oracle; %s; gather %.3f ms" % (weak, cfg["width"], H, info, ms / n))
    job.close(destroy_group=False)
import torch.distributed as dist
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    print("dist_native ok: world=%d" % world)

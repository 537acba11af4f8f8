"""N > 1 path on CPU: world_size-2 gloo process group, row-band partition + tile gather
(raytracertest_amd/dist.py).  The oracle stands in for the per-rank tracer so the whole
sharding logic -- global-row RNG keys, ragged bands, gather order -- is checked without a GPU."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, outdir):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle_py as orc
    from raytracertest_amd import scenes
    from raytracertest_amd.dist import band_rows, gather_tiles
    row0, rows = band_rows(H, world, rank)
    o = orc.OracleTracer(W, H, (0.0, 0.0), 70.0, 3.0, 0.05, seed=1, row0=row0, rows=rows, nthreads=2)
    o.upload_scene(scenes.cornell32())
    o.trace(2, 3)
    tile = torch.from_numpy(o.image.view(np.int32).copy())
    frame = gather_tiles(tile, world, rank)
    if rank == 0:
        np.save(os.path.join(outdir, "frame.npy"), frame.numpy().view(np.uint32))
    else:
        assert frame is None
    dist.barrier()
    dist.destroy_process_group()


def test_band_rows_partition():
    from raytracertest_amd.dist import band_rows
    for H in (1, 7, 37, 1080, 2160):
        for world in (1, 2, 3, 4, 8):
            if world > H:
                continue
            bands = [band_rows(H, world, r) for r in range(world)]
            assert bands[0][0] == 0 and sum(n for _, n in bands) == H
            for (b0, n0), (b1, _) in zip(bands, bands[1:]):
                assert b0 + n0 == b1
            assert max(n for _, n in bands) - min(n for _, n in bands) <= 1
    assert [band_rows(2160, 8, r) for r in range(8)] == [(270 * r, 270) for r in range(8)]   # C5: 8 x 270 rows


@pytest.mark.timeout(300)
def test_two_rank_gloo_gather_equals_single_frame(tmp_path, orc):
    import torch.multiprocessing as mp
    from raytracertest_amd import scenes
    W, H, world = 41, 23, 2            # ragged: 11 + 12 rows
    port = _free_port()
    mp.spawn(_worker, args=(world, port, W, H, str(tmp_path)), nprocs=world, join=True)
    frame = np.load(os.path.join(str(tmp_path), "frame.npy"))
    whole = orc.OracleTracer(W, H, (0.0, 0.0), 70.0, 3.0, 0.05, seed=1, nthreads=4)
    whole.upload_scene(scenes.cornell32())
    whole.trace(2, 3)
    assert frame.shape == (H, W) and np.array_equal(frame, whole.image)

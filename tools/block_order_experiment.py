#!/usr/bin/env python3
"""Experiment: a small-scene launch as TWO kernels by kind of block -- blocks with a tile that generates rays (VALU-bound) and
blocks of certain-winner tiles only (they move memory and step the generators) -- instead of two row halves, the second kernel
optionally capped in blocks per CU by an LDS pad.  Orders are computed on the host from the stored tile words and uploaded
(rt_dbg_set_block_order); steps timed with the lists kept; the first frame of every variant compared bit for bit.
  python3 tools/block_order_experiment.py [steps]"""
import ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracertest_amd as R
from raytracertest_amd import scenes, api

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
cfg = scenes.CONFIGS["C3"]
W, H = cfg["width"], cfg["height"]
lib = R.load_library()
lib.rt_dbg_set_block_order.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
lib.rt_dbg_set_block_order.restype = C.c_int


def tracer():
    g = R.RayTracer((W, H), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=1)
    g.UploadScene(scenes.cornell32())
    return g


g = tracer()
g.Trace(1, cfg["samples"], 0); g.Wait()
ref = (g.RenderBuffer().copy(), g.RngStates().copy(), g.Image().copy())
w0 = g.DebugTileListWords()[:, :, 0]
gx, gy = (W + 31) // 32, (H + 7) // 8
rays = ((w0 >> 31) == 0).reshape(gy, gx, 4).sum(axis=2).ravel()          # tiles that generate rays, per block
by, bx = np.mgrid[0:gy, 0:gx]
ids = (bx | (by << 16)).astype(np.uint32).ravel()
print("blocks", ids.size, "with ray-generating tiles %.3f" % (rays > 0).mean())


def run(policy):
    global g
    g.close()
    g = tracer()
    if policy is not None:
        kind, pad = policy
        if kind == "raster":
            o, first = ids, 0
        elif kind == "grouped":                      # one launch, ray blocks first
            o, first = np.concatenate([ids[rays > 0], ids[rays == 0]]), 0
        elif kind == "split":                        # two launches: ray blocks | certain-winner blocks (padded)
            o, first = np.concatenate([ids[rays > 0], ids[rays == 0]]), int((rays > 0).sum())
        elif kind == "split_rev":                    # two launches: certain-winner blocks | ray blocks (padded)
            o, first = np.concatenate([ids[rays == 0], ids[rays > 0]]), int((rays == 0).sum())
        o = np.ascontiguousarray(o, np.uint32)
        assert lib.rt_dbg_set_block_order(g._h, o.ctypes.data, o.size, first, pad) == 0, g.LastError()
    g.Trace(1, cfg["samples"], 0); g.Wait()
    same = np.array_equal(g.RenderBuffer().view(np.uint32), ref[0].view(np.uint32)) and np.array_equal(g.RngStates(), ref[1]) and np.array_equal(g.Image(), ref[2])
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15: api.dbg_valu_peak(0)
    for _ in range(30): g.TraceEnqueue(1, cfg["samples"])
    g.Sync()
    t0 = time.perf_counter()
    for _ in range(steps): g.TraceEnqueue(1, cfg["samples"])
    g.Sync()
    return (time.perf_counter() - t0) / steps * 1e6, same


res = {}
for rnd in range(3):
    for pol in (None, ("raster", 0), ("grouped", 0), ("split", 0), ("split", 12 << 10), ("split", 20 << 10), ("split", 35 << 10), ("split", 60 << 10),
                ("split_rev", 0), ("split_rev", 20 << 10)):
        us, same = run(pol)
        res.setdefault(str(pol), []).append(round(us, 2))
        assert same, pol
print(json.dumps(res, indent=1))

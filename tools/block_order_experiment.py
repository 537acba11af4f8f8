#!/usr/bin/env python3
"""Experiment: does the ORDER in which a small-scene launch visits its trace blocks matter?  Orders are computed on the host
from the stored tile words and uploaded (rt_dbg_set_block_order); steps timed with the lists kept; frames compared bit for bit.
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
lib.rt_dbg_set_block_order.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
lib.rt_dbg_set_block_order.restype = C.c_int

g = R.RayTracer((W, H), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=1)
g.UploadScene(scenes.cornell32())
g.Trace(1, cfg["samples"], 0); g.Wait()
ref = (g.RenderBuffer().copy(), g.RngStates().copy(), g.Image().copy())
w0 = g.DebugTileListWords()[:, :, 0]
gx, gy = (W + 31) // 32, (H + 7) // 8
exp_tile = (w0 >> 31) == 0                                  # tiles that generate rays
cost = exp_tile.reshape(gy, gx, 4).sum(axis=2)              # expensive tiles per block, 0..4
r0 = ((H * 50 // 100 + 7) // 8) * 8
sb = r0 // 8
print("blocks", gx * gy, "with expensive tiles %.3f" % (cost > 0).mean(), "expensive tiles %.3f" % exp_tile.mean())


def order(policy):
    nb = gx * (gy + 1)
    band = order_of(policy, [(0, gy)])
    hv = order_of(policy, [(0, sb), (sb, gy)])
    out = np.zeros(2 * nb, np.uint32)
    out[:band.size] = band
    out[nb:nb + hv.size] = hv
    return out


def order_of(policy, parts):
    out = []
    for y0, y1 in parts:
        c = cost[y0:y1]
        by, bx = np.mgrid[0:y1 - y0, 0:gx]
        ids = (bx | (by << 16)).astype(np.uint32).ravel()
        cc = c.ravel()
        n = ids.size
        if policy == "raster":
            o = ids
        elif policy == "expensive_first":
            o = np.concatenate([ids[cc > 0], ids[cc == 0]])
        elif policy == "by_cost":
            o = ids[np.argsort(-cc, kind="stable")]
        elif policy == "cheap_first":
            o = np.concatenate([ids[cc == 0], ids[cc > 0]])
        elif policy.startswith("spread"):
            # expensive blocks spread evenly over the first `frac` of the queue, cheap ones fill the gaps and the rest
            frac = float(policy.split(":")[1])
            e, ch = ids[cc > 0], ids[cc == 0]
            span = max(len(e), int(n * frac))
            pos = np.unique((np.arange(len(e)) * span / max(1, len(e))).astype(int))
            assert len(pos) == len(e)
            o = np.empty(n, np.uint32); mask = np.zeros(n, bool); mask[pos] = True
            o[mask] = e; o[~mask] = ch
        else:
            raise KeyError(policy)
        assert np.array_equal(np.sort(o), np.sort(ids))
        out.append(o)
    return np.concatenate(out).astype(np.uint32)


def run(policy):
    global g
    g.close()
    g = R.RayTracer((W, H), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=1)   # (fresh RNG states)
    g.UploadScene(scenes.cornell32())
    if policy is not None:
        o = order(policy)
        assert lib.rt_dbg_set_block_order(g._h, o.ctypes.data, o.size) == 0
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
    for pol in (None, "raster", "expensive_first", "by_cost", "spread:0.5", "spread:0.7", "spread:0.9", "cheap_first"):
        us, same = run(pol)
        res.setdefault(str(pol), []).append(round(us, 2))
        assert same, pol
print(json.dumps(res, indent=1))

#!/usr/bin/env python3
"""Large-scene robustness + timing: default (three-level classification) vs plain full scan at a small
frame, then the default path's time at 4K.  python tools/big_scene.py [n_triangles]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracertest_amd as R
from raytracertest_amd import scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
scn = scenes.random_triangles(n, 4711)
def run(W, H, spp, **kw):
    g = R.RayTracer((W, H), (0, 0, 0), (0.0, 0.0), 70.0, 3.0, 0.05, seed=3, **kw)
    assert g.UploadScene(scn)
    t0 = time.perf_counter(); g.TraceEnqueue(1, spp); g.Sync(); dt = time.perf_counter() - t0
    return g.RenderBuffer().view(np.uint32).copy(), dt, g
a, ta, _ = run(256, 144, 2)
b, tb, _ = run(256, 144, 2, no_binning=True)
print("n=%d  256x144x2: classified %.1f ms, full scan %.1f ms, identical: %s" % (n, ta * 1e3, tb * 1e3, np.array_equal(a, b)))
_, _, g = run(3840, 2160, 4)
g.KernelTime()
for _ in range(3): g.TraceEnqueue(1, 4)
g.Sync(); ms, k = g.KernelTime()
st = g.TraceStats(4)
print("n=%d  3840x2160x4: %.2f ms per launch (%.1f Mray/s); candidates/tile-round %.1f, rounds/tile %.2f" % (
    n, ms / k, 3840 * 2160 * 4 / (ms / k) / 1e3, st["bin_candidates"] / max(st["bin_rounds"], 1), st["bin_rounds"] / (480 * 270)))

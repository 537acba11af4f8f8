#!/usr/bin/env python3
"""How even is the work of the row bands of a frame?  Times every band of `--config` split in N bands (each as its own
band tracer on device 0, one after the other) -- the number a static partition's strong-scaling efficiency is bounded by:
mean / max of the band times.  Usage: band_balance.py [--config C5] [--bands 8] [--reps 5]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracertest_amd as R
from raytracertest_amd import scenes
from raytracertest_amd.dist import band_rows

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C5")
ap.add_argument("--bands", type=int, default=8)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
cfg = scenes.CONFIGS[a.config]
tris, sph = scenes.scene_for(a.config)
times = []
for k in range(a.bands):
    r0, n = band_rows(cfg["height"], a.bands, k)
    g = R.RayTracer((cfg["width"], n), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=cfg["seed"],
                    full_height=cfg["height"], row_begin=r0)
    if tris.shape[0]:
        g.UploadScene(tris)
    if sph.shape[0]:
        g.UploadSpheres(sph)
    g.SetListReuse(False)
    g.TraceEnqueue(cfg["iterations"], cfg["samples"]); g.Sync()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        g.TraceEnqueue(cfg["iterations"], cfg["samples"])
    g.Sync()
    times.append((time.perf_counter() - t0) / a.reps * 1e3)
    g.close()
t = np.array(times)
print("band ms:", " ".join("%.3f" % x for x in t))
print("sum %.3f ms, max %.3f ms, mean/max = %.3f (bound on the strong-scaling efficiency of the equal-rows partition)" % (t.sum(), t.max(), t.mean() / t.max()))

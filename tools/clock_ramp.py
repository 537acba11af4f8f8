#!/usr/bin/env python3
"""How long does the device take to reach its steady clocks under this path?  Prints the wall time per C3 step in blocks
of consecutive steps from a cold process (bench.py's --warmup/--steps defaults are sized from this), optionally after
`--preheat N` runs of the VALU calibration loop (rt_dbg_valu_peak, ~4 ms each) or `--idle S` seconds of sleep."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracertest_amd as R
from raytracertest_amd import api, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--blocks", type=int, default=40)
ap.add_argument("--block", type=int, default=50)
ap.add_argument("--preheat", type=int, default=0)
ap.add_argument("--idle", type=float, default=0.0)
a = ap.parse_args()

cfg = dict(scenes.CONFIGS["C3"])
tris, sph = scenes.scene_for("C3")
g = R.RayTracer((cfg["width"], cfg["height"]), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=cfg["seed"])
g.UploadScene(tris)
g.SetListReuse(False)
g.Sync()
if a.preheat:
    t0 = time.perf_counter()
    for _ in range(a.preheat):
        lane_fma, ghz = api.dbg_valu_peak(0)
    print("preheat %d x valu_peak: %.1f ms, last clock %.3f GHz" % (a.preheat, (time.perf_counter() - t0) * 1e3, ghz))
if a.idle:
    time.sleep(a.idle)
out = []
t_start = time.perf_counter()
for b in range(a.blocks):
    t0 = time.perf_counter()
    for _ in range(a.block):
        g.TraceEnqueue(1, cfg["samples"])
    g.Sync()
    out.append((time.perf_counter() - t0) / a.block * 1e6)
print("us/step per block of %d (total %.0f ms): %s" % (a.block, (time.perf_counter() - t_start) * 1e3, " ".join("%.1f" % x for x in out)))

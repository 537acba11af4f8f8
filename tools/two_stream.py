#!/usr/bin/env python3
"""Hypothesis: back-to-back Trace passes lose ~7 % to the drain at the end of each launch; two tracers that
each own half the rows of the frame (band mode, own streams) overlap one half's drain with the other half's work."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracertest_amd as R
from raytracertest_amd import scenes
cfg = scenes.CONFIGS["C3"]; tris = scenes.cornell32()
W, H = cfg["width"], cfg["height"]
def mk(row0, rows):
    g = R.RayTracer((W, rows), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=1,
                    full_height=H if rows != H else 0, row_begin=row0)
    g.UploadScene(tris); g.SetListReuse(False)
    return g
def run(parts, steps=300):
    gs = [mk(r0, n) for r0, n in parts]
    for _ in range(20):
        for g in gs: g.TraceEnqueue(1, 16)
    for g in gs: g.Sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        for g in gs: g.TraceEnqueue(1, 16)
    for g in gs: g.Sync()
    dt = (time.perf_counter() - t0) / steps
    for g in gs: g.close()
    return dt
for rep in range(2):
    a = run([(0, H)])
    b = run([(0, 544), (544, H - 544)])
    c = run([(0, 360), (360, 360), (720, 360)])
    d = run([(0, 272), (272, 272), (544, 272), (816, 264)])
    print("1 tracer %.1f us/step | 2 halves %.1f | 3 thirds %.1f | 4 quarters %.1f" % (a * 1e6, b * 1e6, c * 1e6, d * 1e6))

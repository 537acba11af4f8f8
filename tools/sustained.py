#!/usr/bin/env python3
"""Sustained headline steps (C3, lists rebuilt per step): us/step of consecutive chunks, and the shader clock the VALU
calibration loop holds right after each chunk.  Usage: sustained.py [chunks] [steps per chunk]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracertest_amd as R
from raytracertest_amd import api, scenes
chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 20
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
cfg = scenes.CONFIGS["C3"]
g = R.RayTracer((cfg["width"], cfg["height"]), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=1)
g.UploadScene(scenes.cornell32()); g.SetListReuse(False)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.15: api.dbg_valu_peak(0)
out = []
for c in range(chunks):
    t0 = time.perf_counter()
    for _ in range(steps): g.TraceEnqueue(1, 16)
    g.Sync()
    us = (time.perf_counter() - t0) / steps * 1e6
    ghz = api.dbg_valu_peak(0)[1] if c % 5 == 4 else 0.0
    out.append("%.1f%s" % (us, (" [%.2f GHz]" % ghz) if ghz else ""))
print("us/step per chunk of %d steps: %s" % (steps, ", ".join(out)))

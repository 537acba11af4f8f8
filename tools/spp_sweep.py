#!/usr/bin/env python3
"""Kernel time vs samples per launch (fixed 1920x1080 frame): slope = per-sample cost,
intercept = fixed per-launch cost (state traffic, tile family + classification, ramp/tail)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracertest_amd as R
from raytracertest_amd import scenes
cfg = scenes.CONFIGS["C3"]
for label, scn in (("empty", None), ("cornell32", scenes.cornell32())):
    g = R.RayTracer((cfg["width"], cfg["height"]), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=1)
    if scn is not None:
        g.UploadScene(scn)
    for spp in (4, 8, 16, 32, 64):
        for _ in range(3):
            g.TraceEnqueue(1, spp)
        g.Sync(); g.KernelTime()
        for _ in range(20):
            g.TraceEnqueue(1, spp)
        g.Sync(); ms, n = g.KernelTime()
        print("%-10s spp=%3d  %8.1f us  (%.2f us/spp)" % (label, spp, ms / n * 1e3, ms / n * 1e3 / spp))

#!/usr/bin/env python3
"""Per-sample forms on/off (RT_MI355X_NO_PRETEST) around the size limits: 3840x2160 x 16 spp kernel time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracertest_amd as R
from raytracertest_amd import scenes
for n in (4096, 10000, 20000, 30000, 50000):
    g = R.RayTracer((3840, 2160), (0, 0, 0), (0.0, 0.0), 70.0, 3.0, 0.05, seed=3)
    g.UploadScene(scenes.random_triangles(n, 4711))
    g.TraceEnqueue(1, 16); g.Sync(); g.KernelTime()
    for _ in range(4): g.TraceEnqueue(1, 16)
    g.Sync(); ms, k = g.KernelTime()
    st = g.TraceStats(4)
    print("n=%6d  %.2f ms  candidates/round %.1f rounds/tile %.2f  (NO_PRETEST=%s)" % (n, ms / k, st["bin_candidates"] / max(st["bin_rounds"], 1), st["bin_rounds"] / (480 * 270), os.environ.get("RT_MI355X_NO_PRETEST", "0")))
    g.close()

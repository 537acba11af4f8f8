#!/usr/bin/env python3
"""Kernel time per launch at 1920x1080x16 for mid-size random scenes (and C4 at 4K x 64) -- run once per
library build (RT_MI355X_LIB) to compare builds on the same box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracertest_amd as R
from raytracertest_amd import scenes
cfg = scenes.CONFIGS["C3"]
out = []
for name, n, seed in (("rand300", 300, 777), ("rand1000", 1000, 9), ("rand3000", 3000, 10)):
    g = R.RayTracer((cfg["width"], cfg["height"]), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=1)
    g.UploadScene(scenes.random_triangles(n, seed))
    for _ in range(3): g.TraceEnqueue(1, 16)
    g.Sync(); g.KernelTime()
    for _ in range(24): g.TraceEnqueue(1, 16)
    g.Sync(); ms, k = g.KernelTime()
    out.append("%s %.1f us" % (name, ms / k * 1e3)); g.close()
c4 = scenes.CONFIGS["C4"]; tris, _ = scenes.scene_for("C4")
g = R.RayTracer((c4["width"], c4["height"]), (0, 0, 0), c4["angles"], c4["fov"], c4["focal"], c4["aperture"], seed=1)
g.UploadScene(tris)
g.TraceEnqueue(1, 64); g.Sync(); g.KernelTime()
for _ in range(6): g.TraceEnqueue(1, 64)
g.Sync(); ms, k = g.KernelTime()
out.append("C4 %.2f ms" % (ms / k))
print(os.path.basename(os.environ.get("RT_MI355X_LIB", "default")), " | ".join(out))

#!/usr/bin/env python3
"""FILTER (conservative wave-uniform early-outs per candidate) on/off for several scenes at 1920x1080x16
(and C4): kernel time per launch, alternating."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracertest_amd as R
from raytracertest_amd import scenes
cfg = scenes.CONFIGS["C3"]
cases = [("cornell32", scenes.cornell32()), ("demo3", scenes.demo3()), ("rand16", scenes.random_triangles(16, 5)),
         ("rand64", scenes.random_triangles(64, 6)), ("rand200", scenes.random_triangles(200, 7)), ("rand256", scenes.random_triangles(256, 8)),
         ("rand300", scenes.random_triangles(300, 777)), ("rand1000", scenes.random_triangles(1000, 9))]
for name, tris in cases:
    res = {}
    for rnd in range(2):
        for key, kw in (("filter", dict()), ("plain", dict(no_filter=True))):
            g = R.RayTracer((cfg["width"], cfg["height"]), (0, 0, 0), (0.0, 3.0) if name == "demo3" else cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=1, **kw)
            g.UploadScene(tris)
            for _ in range(3): g.TraceEnqueue(1, 16)
            g.Sync(); g.KernelTime()
            for _ in range(24): g.TraceEnqueue(1, 16)
            g.Sync(); ms, n = g.KernelTime()
            res.setdefault(key, []).append(ms / n * 1e3)
            g.close()
    print("%-10s filter %s   plain %s" % (name, " ".join("%.1f" % x for x in res["filter"]), " ".join("%.1f" % x for x in res["plain"])))

#!/usr/bin/env python3
"""Where does the C3 launch spend its time?  Same frame with (a) no geometry (ray generation,
background shade, RNG/accumulator traffic only), (b) the scene, for K = 1, 2, 4."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracertest_amd as R
from raytracertest_amd import scenes

cfg = scenes.CONFIGS["C3"]
for label, scn in (("empty", None), ("cornell32", scenes.cornell32()), ("rand10k", scenes.random_triangles(10000, 12345))):
    for k in (1, 2, 4):
        g = R.RayTracer((cfg["width"], cfg["height"]), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"],
                        cfg["aperture"], seed=1, samples_in_flight=k)
        if scn is not None:
            g.UploadScene(scn)
        for _ in range(3):
            g.TraceEnqueue(1, 16)
        g.Sync(); g.KernelTime()
        for _ in range(20):
            g.TraceEnqueue(1, 16)
        g.Sync(); ms, n = g.KernelTime()
        print("%-10s K=%d  %8.1f us" % (label, k, ms / n * 1e3))
        g.close()

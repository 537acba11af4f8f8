#!/usr/bin/env python3
"""End-to-end time of the reference app's default interactive workload through the async API:
Trace(100 iterations x 1 spp, update every 10) at the app's 38x21 and at 1920x1080."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracertest_amd as R
from raytracertest_amd import scenes
for (W, H) in ((38, 21), (640, 360), (1920, 1080)):
    g = R.RayTracer((W, H), (0, 0, 0), (0.0, 3.0), 70.0, 10.0, 0.5, seed=1)
    g.UploadScene(scenes.demo3())
    n = {"u": 0}
    g.SetUpdateCallback(lambda img, size: n.__setitem__("u", n["u"] + 1))
    g.Trace(100, 1, 10); g.Wait()
    t0 = time.perf_counter()
    for _ in range(5):
        g.Trace(100, 1, 10); assert g.Wait()
    dt = (time.perf_counter() - t0) / 5
    print("%4dx%-4d  Trace(100,1,10): %7.2f ms end to end  (%.1f us per iteration, %d updates)" % (W, H, dt * 1e3, dt * 1e4, n["u"] // 6))

#!/usr/bin/env python3
"""End-to-end (PCIe-inclusive) time of a blocking Trace(1, 16, 0) + Wait() at C3, image handed to the
finished callback in pinned host memory."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracertest_amd as R
from raytracertest_amd import scenes
cfg = scenes.CONFIGS["C3"]
g = R.RayTracer((cfg["width"], cfg["height"]), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=1)
g.UploadScene(scenes.cornell32())
got = []
g.SetFinishedCallback(lambda img, size: got.append(int(img[540, 960])))
for _ in range(5):
    g.Trace(1, 16, 0); g.Wait()
t0 = time.perf_counter()
N = 50
for _ in range(N):
    g.Trace(1, 16, 0); assert g.Wait()
dt = (time.perf_counter() - t0) / N
rays = cfg["width"] * cfg["height"] * 16
print("Trace(1,16,0)+Wait at 1920x1080: %.1f us end to end = %.0f Mray/s incl. thread start, host image hand-off (%d callbacks)" % (dt * 1e6, rays / dt / 1e6, len(got)))

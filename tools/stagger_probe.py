#!/usr/bin/env python3
"""Do the two half-frame kernels of back-to-back launches run in phase (both drain at the same time, nothing covers the
tail) or in anti-phase?  Delays the second stream once by a spin kernel of a given length before a long run of C3 steps
and compares the steady step time with the undelayed run."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracertest_amd as R
from raytracertest_amd import scenes

cfg = dict(scenes.CONFIGS["C3"])
tris, sph = scenes.scene_for("C3")
g = R.RayTracer((cfg["width"], cfg["height"]), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=cfg["seed"])
g.UploadScene(tris)
g.SetListReuse(False)
sb = torch.cuda.ExternalStream(g.StreamB())
sa = torch.cuda.ExternalStream(g.Stream())


def run(n):
    t0 = time.perf_counter()
    for _ in range(n):
        g.TraceEnqueue(1, cfg["samples"])
    g.Sync()
    return (time.perf_counter() - t0) / n * 1e6


run(1500)                                             # clocks
print("in phase (as launched)      %.1f %.1f us" % (run(2000), run(2000)))
for which, name in ((sb, "B"), (sa, "A")):
    for us in (20, 40, 60, 80):
        g.Sync()
        with torch.cuda.stream(which):
            torch.cuda._sleep(int(us * 1e-6 * 2.0e9))
        print("stream %s delayed once by ~%d us: %.1f us, then (after a Sync) %.1f us" % (name, us, run(2000), run(2000)))

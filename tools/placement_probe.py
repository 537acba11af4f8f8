#!/usr/bin/env python3
"""Does the speed of a tracer depend on what else is allocated / was allocated before?  (bench.py found a second tracer
20 % slower while the first one was still alive.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracertest_amd as R
from raytracertest_amd import scenes
cfg = dict(scenes.CONFIGS["C3"]); tris, sph = scenes.scene_for("C3")

def make():
    g = R.RayTracer((cfg["width"], cfg["height"]), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=1, no_sure_hit=True)
    g.UploadScene(tris); g.SetListReuse(False)
    return g

def t(g, n=200):
    for _ in range(20): g.TraceEnqueue(1, 16)
    g.Sync(); t0 = time.perf_counter()
    for _ in range(n): g.TraceEnqueue(1, 16)
    g.Sync(); return (time.perf_counter() - t0) / n * 1e6

a = make(); print("A alone            %.1f us" % t(a))
b = make(); print("B (A alive, idle)  %.1f us" % t(b)); print("A (B alive, idle)  %.1f us" % t(a))
a.close(); print("B (A closed)       %.1f us" % t(b))
b.Resize((cfg["width"], cfg["height"])); b.SetSeed(1); print("B after Resize     %.1f us" % t(b))
c = make(); print("C (B alive)        %.1f us" % t(c)); b.close(); print("C (B closed)       %.1f us" % t(c))

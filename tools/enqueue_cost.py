#!/usr/bin/env python3
"""Host cost of enqueueing headline steps (TraceEnqueue(1, 16) at C3, lists rebuilt per step) next to the device time:
per-call wall time of the enqueue for the first calls and in steady state, and the drained total.  Usage: enqueue_cost.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import raytracertest_amd as R
from raytracertest_amd import api, scenes
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cfg = scenes.CONFIGS["C3"]
g = R.RayTracer((cfg["width"], cfg["height"]), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=1)
g.UploadScene(scenes.cornell32()); g.SetListReuse(False)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.15: api.dbg_valu_peak(0)
for _ in range(5): g.TraceEnqueue(1, 16)
g.Sync()
for rep in range(3):
    ts = []
    t0 = time.perf_counter()
    for _ in range(steps):
        a = time.perf_counter(); g.TraceEnqueue(1, 16); ts.append(time.perf_counter() - a)
    enq = time.perf_counter() - t0
    g.Sync()
    tot = time.perf_counter() - t0
    ts = np.array(ts) * 1e6
    print("rep %d: %d steps: enqueue %.1f us/step (first 20: %.1f, last 100 median %.1f, max %.0f), drained total %.1f us/step"
          % (rep, steps, enq / steps * 1e6, ts[:20].mean(), np.median(ts[-100:]), ts.max(), tot / steps * 1e6))
# the loop inside the library (rt_tracer_trace_enqueue_n, what bench.py's timed region calls): one ctypes crossing for all steps
for rep in range(3):
    g.Sync(); t0 = time.perf_counter()
    g.TraceEnqueueN(1, 16, steps)
    enq = time.perf_counter() - t0
    g.Sync()
    tot = time.perf_counter() - t0
    print("TraceEnqueueN rep %d: %d steps: enqueue %.1f us/step inside the library, drained total %.1f us/step" % (rep, steps, enq / steps * 1e6, tot / steps * 1e6))
for n in (20, 20, 20, 50, 100):
    g.Sync(); t0 = time.perf_counter()
    for _ in range(n): g.TraceEnqueue(1, 16)
    g.Sync()
    print("%d steps from an idle device: %.1f us/step" % (n, (time.perf_counter() - t0) / n * 1e6))

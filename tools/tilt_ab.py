#!/usr/bin/env python3
"""Split launches by row halves against by even / odd block rows (RT_MI355X_ROW_INTERLEAVE=0/1; unset: chosen by the list builder's per-half counts) for cameras that put the
expensive tiles into one half of the frame.  python3 tools/tilt_ab.py"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time
sys.path.insert(0, %r)
import raytracertest_amd as R
from raytracertest_amd import scenes, api
ang = (%f, %f)
g = R.RayTracer((1920, 1080), (0, 0, 0), ang, 70.0, 3.0, 0.05, seed=1)
g.UploadScene(scenes.cornell32()); g.SetListReuse(False)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.15: api.dbg_valu_peak(0)
for _ in range(30): g.TraceEnqueue(1, 16)
g.Sync()
t0 = time.perf_counter()
for _ in range(400): g.TraceEnqueue(1, 16)
g.Sync()
us = (time.perf_counter() - t0) / 400 * 1e6
w0 = g.DebugTileListWords()[:, :, 0]
sure = (w0 >> 31) != 0
h = sure.shape[0] // 2
print("%%.1f %%.3f %%.3f" %% (us, 1 - sure[:h].mean(), 1 - sure[h:].mean()))
'''
for ang in ((0.0, 0.0), (0.0, 0.3), (0.0, -0.3), (0.3, 0.0), (0.0, 0.6)):
    row = []
    for il in ("1", "0", ""):
        env = dict(os.environ)
        if il: env["RT_MI355X_ROW_INTERLEAVE"] = il
        out = subprocess.run([sys.executable, "-c", CHILD % (ROOT, ang[0], ang[1])], env=env, capture_output=True, text=True)
        row.append(out.stdout.strip().splitlines()[-1] if out.returncode == 0 else "FAILED " + out.stderr[-200:])
    print("angles %s: even/odd block rows %s us | row halves %s us | chosen by the builder's counts %s us   (ray-generating share of the upper / lower half: %s)" % (
        ang, row[0].split()[0], row[1].split()[0], row[2].split()[0], " / ".join(row[1].split()[1:])))

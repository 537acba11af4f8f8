#!/usr/bin/env python3
"""Tiles of a small-scene frame by the length of their candidate list, and how many of the one-candidate tiles are
certainly hit (instrumented launch).  Usage: sure_hit_stats.py [config] [aperture]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracertest_amd as R
from raytracertest_amd import scenes
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = dict(scenes.CONFIGS[name]); tris, sph = scenes.scene_for(name)
if len(sys.argv) > 2:
    cfg["aperture"] = float(sys.argv[2])
g = R.RayTracer((cfg["width"], cfg["height"]), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=1)
g.UploadScene(tris)
st = g.TraceStats(2)
t = st["tiles_by_list"]; n = sum(t.values())
print(name, "aperture", cfg["aperture"], "tiles", n, {k: "%.3f" % (v / n) for k, v in t.items()}, "candidates/tile %.3f" % (st["bin_candidates"] / max(st["bin_rounds"], 1)))

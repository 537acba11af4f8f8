#!/usr/bin/env python3
"""C3: the stored tile lists -- how many candidates do the tiles that still generate rays carry, and which triangles (walls
0-9, light 10-11, short box 12-21, tall box 22-31)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracertest_amd as R
from raytracertest_amd import scenes
cfg = scenes.CONFIGS["C3"]; tris, _ = scenes.scene_for("C3")
g = R.RayTracer((cfg["width"], cfg["height"]), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=1)
g.UploadScene(tris)
g.TraceEnqueue(1, 16); g.Sync()
w = g.DebugTileListWords()
hdr = w[..., 0]
cnt = (hdr & 0x3FF).astype(int); sure = (hdr >> 31) != 0
H, Wt = cnt.shape[0], cnt.shape[1]
valid = np.ones_like(sure)
print("tiles", cnt.size, "certain", int(sure.sum()), "traced", int((~sure).sum()))
tr = cnt[~sure]
print("traced tiles by candidate count:", {int(k): int((tr == k).sum()) for k in np.unique(tr)})
print("mean candidates of traced tiles with >= 1:", tr[tr > 0].mean(), " sum of candidates:", int(tr.sum()))
walls = boxes = 0
for c in range(1, w.shape[-1]):
    m = (~sure) & (cnt >= c)
    idx = w[..., c][m]
    walls += int((idx < 12).sum()); boxes += int((idx >= 12).sum())
print("candidate entries of traced tiles: walls + light", walls, " boxes", boxes)

#!/usr/bin/env python3
"""How large are the regions of tiles with the SAME certain winner?  (Would a coarser pre-pass that decides whole regions
at once, sparing their tiles the per-wave prologue, find enough of them?)  Usage: sure_regions.py [config]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracertest_amd as R
from raytracertest_amd import scenes
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = dict(scenes.CONFIGS[name]); tris, sph = scenes.scene_for(name)
g = R.RayTracer((cfg["width"], cfg["height"]), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=1)
g.UploadScene(tris)
g.Trace(2, 2, 0); assert g.Wait()                     # the second launch stores the lists
count, winner, sure = g.DebugTileLists()
tx = (cfg["width"] + 7) // 8
count, winner, sure = count[:, :tx], winner[:, :tx], sure[:, :tx]
print(name, "tiles", sure.size, "certain winner %.3f" % sure.mean())
key = np.where(sure, winner, -1)
for ry, rx in ((1, 4), (2, 4), (4, 4), (4, 8), (8, 8), (8, 16)):      # region = ry x rx tiles (8*ry rows x 8*rx columns of pixels)
    H, W = (key.shape[0] // ry) * ry, (key.shape[1] // rx) * rx
    k = key[:H, :W].reshape(H // ry, ry, W // rx, rx)
    uniform = (k.min(axis=(1, 3)) == k.max(axis=(1, 3))) & (k.min(axis=(1, 3)) >= 0)
    print("  regions of %2d x %2d tiles (%3d x %3d px): %.3f of the tiles lie in an all-certain region with one winner" % (ry, rx, 8 * rx, 8 * ry, uniform.mean() * (H * W) / key.size))

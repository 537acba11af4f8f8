#!/usr/bin/env python3
"""Which 8x8 tiles of a configuration generate rays (1) and which have a certain winner (0), in trace-grid order (block row,
block, tile of the block), one byte per tile: the input of tools/dispatch_floor.  python tools/dump_kinds.py [C3] [out.bin]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracertest_amd as R
from raytracertest_amd import scenes

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/%s_kinds.bin" % name.lower()
cfg = dict(scenes.CONFIGS[name]); tris, sph = scenes.scene_for(name)
g = R.RayTracer((cfg["width"], cfg["height"]), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=cfg["seed"])
g.UploadScene(tris)
g.TraceEnqueue(1, cfg["samples"]); g.Sync()
count, winner, certain = g.DebugTileLists()
kinds = (~certain).astype(np.uint8)              # (tiles_y, tiles_x) with tiles_x = 4 * blocks: already block-major within a row
os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
kinds.tofile(out)
print("%s: %d x %d tiles, ray-generating %.4f (empty list %.4f) -> %s" % (name, kinds.shape[1], kinds.shape[0], kinds.mean(),
      ((count == 0) & ~certain).mean(), out))

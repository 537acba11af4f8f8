#!/usr/bin/env python3
"""What does a frame cost when EVERY tile has a certain winner?  One triangle that covers the whole view (1920x1080, 16 spp, the
C3 camera): the trace kernel does the RNG draws, the additions and the frame's memory traffic, nothing else -- the floor of the
certain-winner path, next to the same frame of the C3 scene.  python3 tools/sure_floor.py [steps]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracertest_amd as R
from raytracertest_amd import scenes, api

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
cfg = scenes.CONFIGS["C3"]
big = np.array([[-100, -100, -4, 0], [100, -100, -4, 0], [0, 200, -4, 0]], np.float32)
out = {}
for name, scn in (("one_triangle_all_certain", big), ("C3", scenes.cornell32())):
    for reuse in (True, False):
        g = R.RayTracer((cfg["width"], cfg["height"]), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=1)
        g.UploadScene(scn)
        g.SetListReuse(reuse)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.15: api.dbg_valu_peak(0)
        for _ in range(30): g.TraceEnqueue(1, cfg["samples"])
        g.Sync(); g.KernelTime()
        t0 = time.perf_counter()
        for _ in range(steps): g.TraceEnqueue(1, cfg["samples"])
        g.Sync(); wall = (time.perf_counter() - t0) / steps * 1e6
        w0 = g.DebugTileListWords()[:, :, 0]
        out["%s, lists %s" % (name, "kept" if reuse else "rebuilt")] = {"us_per_step": round(wall, 2), "certain": float(((w0 >> 31) != 0).mean()),
                                                                        "GB_s_algorithmic": round(165889536 / wall / 1e3, 1)}
        g.close()
print(json.dumps(out, indent=1))

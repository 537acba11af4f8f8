#!/usr/bin/env python3
"""Is a small band's inefficiency the length of its waves?  The 270-row C5 band, lists kept, 256 samples per pixel as 1 x 256,
2 x 128, 4 x 64, 8 x 32 launches (NOT the same bits: the accumulation order differs -- a timing experiment only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracertest_amd as R
from raytracertest_amd import api, scenes
cfg = scenes.CONFIGS["C5"]; tris, _ = scenes.scene_for("C5")
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.15: api.dbg_valu_peak(0)
for rows in (270, 2160):
    row0 = 810 if rows == 270 else 0
    g = R.RayTracer((cfg["width"], rows), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=1,
                    full_height=cfg["height"], row_begin=row0)
    g.UploadScene(tris)
    g.SetListReuse(True)
    for spp in (256, 128, 64, 32, 256):
        n = 256 // spp
        g.TraceEnqueueN(1, spp, 3 * n); g.Sync()
        t0 = time.perf_counter()
        g.TraceEnqueueN(1, spp, 10 * n); g.Sync()
        dt = (time.perf_counter() - t0) / 10
        print("rows %4d: %d x %3d spp: %.3f ms per 256 spp = %.1f Gray/s" % (rows, n, spp, dt * 1e3, cfg["width"] * rows * 256 / dt / 1e9), flush=True)
    g.close()

#!/usr/bin/env python3
"""One 270-row band of C5 (what one GPU of the 8-band strong-scaling record traces per step): step time with the lists rebuilt by
every step and kept, and the same for a 540-row band (4 GPUs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracertest_amd as R
from raytracertest_amd import api, scenes
cfg = scenes.CONFIGS["C5"]; tris, _ = scenes.scene_for("C5")
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.15: api.dbg_valu_peak(0)
for rows in (270, 540, 1080):
    row0 = (cfg["height"] // rows // 2) * rows
    g = R.RayTracer((cfg["width"], rows), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=1,
                    full_height=cfg["height"], row_begin=row0)
    g.UploadScene(tris)
    for reuse in (False, True, False, True):
        g.SetListReuse(reuse)
        g.TraceEnqueueN(1, cfg["samples"], 3); g.Sync()
        t0 = time.perf_counter()
        g.TraceEnqueueN(1, cfg["samples"], 20); g.Sync()
        dt = (time.perf_counter() - t0) / 20
        print("C5 band of %4d rows, lists %s: %.3f ms per step = %.1f Gray/s" % (rows, "kept   " if reuse else "rebuilt", dt * 1e3, cfg["width"] * rows * cfg["samples"] / dt / 1e9), flush=True)
    g.close()

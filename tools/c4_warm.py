#!/usr/bin/env python3
"""C4 (or C5) step time with the dense scene's lists rebuilt by every step (bench.py's headline rule) and kept across
Traces (the library's default for an unchanged view): what the accumulating launches of a progressive Trace cost."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracertest_amd as R
from raytracertest_amd import api, scenes
args = [a for a in sys.argv[1:] if not a.startswith("--")]
name = args[0] if args else "C4"
kw = {"no_super_bins": True} if "--no-super" in sys.argv else {}
cfg = scenes.CONFIGS[name]; tris, _ = scenes.scene_for(name)
g = R.RayTracer((cfg["width"], cfg["height"]), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=1, **kw)
g.UploadScene(tris)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.15: api.dbg_valu_peak(0)
for reuse in (False, True, False, True):
    g.SetListReuse(reuse)
    g.TraceEnqueueN(1, cfg["samples"], 3); g.Sync()
    t0 = time.perf_counter()
    g.TraceEnqueueN(1, cfg["samples"], 20); g.Sync()
    dt = (time.perf_counter() - t0) / 20
    print("%s lists %s: %.3f ms per step = %.1f Gray/s" % (name, "kept   " if reuse else "rebuilt", dt * 1e3, cfg["width"] * cfg["height"] * cfg["samples"] / dt / 1e9))

import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracertest_amd as R
from raytracertest_amd import api, scenes
cfg = scenes.CONFIGS["C4"]; tris, _ = scenes.scene_for("C4")
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.15: api.dbg_valu_peak(0)
for k, bl in ((4, 0), (2, 48), (2, 64), (4, 48), (4, 64)):
    g = R.RayTracer((cfg["width"], cfg["height"]), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=1, samples_in_flight=k, bin_list=bl)
    g.UploadScene(tris)
    for reuse in (False, True):
        g.SetListReuse(reuse)
        g.TraceEnqueueN(1, cfg["samples"], 3); g.Sync()
        t0 = time.perf_counter(); g.TraceEnqueueN(1, cfg["samples"], 15); g.Sync()
        dt = (time.perf_counter() - t0) / 15
        print("K=%d bin_list=%d lists %s: %.3f ms  (%s)" % (k, bl, "kept" if reuse else "rebuilt", dt * 1e3, g.Info()))
    g.close()

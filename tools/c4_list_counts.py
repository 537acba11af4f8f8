import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracertest_amd as R
from raytracertest_amd import scenes
cfg = scenes.CONFIGS["C4"]; tris, _ = scenes.scene_for("C4")
g = R.RayTracer((cfg["width"], cfg["height"]), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=1)
g.UploadScene(tris); g.TraceEnqueue(1, 4); g.Sync()
for h in (0, 1):
    c, cap = g.DebugWaveListCounts(h)
    ok = c[c != 0xFFFFFFFF]
    print("half", h, "tiles", c.size, "cap", cap, "overflow", int((c == 0xFFFFFFFF).sum()), "mean %.2f" % ok.mean(), "max", ok.max(), "p99", np.percentile(ok, 99), "p99.9", np.percentile(ok, 99.9), ">64:", int((ok > 64).sum()), ">48:", int((ok > 48).sum()))

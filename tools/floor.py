import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracertest_amd as R
from raytracertest_amd import scenes
cfg = scenes.CONFIGS["C3"]
for label, scn in (("empty", None), ("cornell32", scenes.cornell32())):
    g = R.RayTracer((cfg["width"], cfg["height"]), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=1)
    if scn is not None: g.UploadScene(scn)
    for spp in (1, 2, 3, 4):
        for _ in range(3): g.TraceEnqueue(1, spp)
        g.Sync(); g.KernelTime()
        for _ in range(30): g.TraceEnqueue(1, spp)
        g.Sync(); ms, n = g.KernelTime()
        print("%-10s spp=%d %7.1f us" % (label, spp, ms / n * 1e3))

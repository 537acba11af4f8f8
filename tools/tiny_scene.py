import os, sys, time
sys.path.insert(0, "/root/repo")
import raytracertest_amd as R
from raytracertest_amd import scenes
for (W, H) in ((38, 21), (1920, 1080)):
    for kw in (dict(), dict(no_binning=True)):
        g = R.RayTracer((W, H), (0, 0, 0), (0.0, 3.0), 70.0, 10.0, 0.5, seed=1, **kw)
        g.UploadScene(scenes.demo3())
        g.SetUpdateCallback(lambda img, size: None)
        g.Trace(100, 1, 10); g.Wait()
        t0 = time.perf_counter()
        for _ in range(5):
            g.Trace(100, 1, 10); assert g.Wait()
        dt = (time.perf_counter() - t0) / 5
        for spp in (1, 16):
            for _ in range(3): g.TraceEnqueue(1, spp)
            g.Sync(); g.KernelTime()
            for _ in range(20): g.TraceEnqueue(1, spp)
            g.Sync(); ms, n = g.KernelTime()
            print("%dx%d %s: single launch %d spp %.1f us" % (W, H, kw, spp, ms / n * 1e3))
        print("%dx%d %s: Trace(100,1,10) %.2f ms" % (W, H, kw, dt * 1e3))

import os, sys, time
sys.path.insert(0, "/root/repo")
import raytracertest_amd as R
from raytracertest_amd import scenes
for (W, H) in ((38, 21), (640, 360)):
    g = R.RayTracer((W, H), (0, 0, 0), (0.0, 3.0), 70.0, 10.0, 0.5, seed=1)
    g.UploadScene(scenes.demo3())
    g.Trace(100, 1, 10); g.Wait(); g.KernelTime()
    t0 = time.perf_counter()
    for _ in range(5):
        g.Trace(100, 1, 10); assert g.Wait()
    dt = (time.perf_counter() - t0) / 5
    ms, n = g.KernelTime()
    print("%dx%d: wall %.1f us/iter, kernel %.2f us/launch (%d launches)" % (W, H, dt * 1e4, ms / n * 1e3, n))
    t0 = time.perf_counter()
    for _ in range(5):
        g.TraceEnqueue(100, 1); g.Sync()
    dt = (time.perf_counter() - t0) / 5
    ms, n = g.KernelTime()
    print("   enqueue-only path: wall %.1f us/iter, kernel %.2f us/launch" % (dt * 1e4, ms / n * 1e3))

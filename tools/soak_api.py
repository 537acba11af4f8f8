#!/usr/bin/env python3
"""API soak: random Trace / Stop / Resize / RotateCamera / SetCameraParameters / UploadScene calls against a
running render thread (fused launches, pipelined update hand-off), then a final Trace checked against the
oracle.  python tools/soak_api.py [rounds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracertest_amd as R
from raytracertest_amd import scenes
from oracle import oracle_py as orc
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
W, H = 96, 54
angles = [0.0, 0.0]; fov, focal, ap = 70.0, 3.0, 0.05
g = R.RayTracer((W, H), (0, 0, 0), tuple(angles), fov, focal, ap, seed=5)
scn = scenes.cornell32(); g.UploadScene(scn)
ups, fins = [0], [0]
def on_up(img, size):
    assert size == img.size * 4; ups[0] += 1
def on_fin(img, size):
    fins[0] += 1
g.SetUpdateCallback(on_up); g.SetFinishedCallback(on_fin)
t0 = time.time()
for r in range(rounds):
    op = rng.integers(0, 9)
    if op <= 2:
        g.Trace(int(rng.integers(1, 60)), int(rng.integers(1, 4)), int(rng.integers(0, 7)))
    elif op == 3:
        g.Stop()
    elif op == 4:
        W, H = int(rng.integers(8, 200)), int(rng.integers(8, 120)); g.Resize((W, H))
    elif op == 5:
        d = (float(rng.uniform(-0.2, 0.2)), float(rng.uniform(-0.2, 0.2))); g.RotateCamera(d); angles[0] += d[0]; angles[1] += d[1]
    elif op == 6:
        fov, focal, ap = float(rng.uniform(30, 100)), float(rng.uniform(1, 6)), float(rng.uniform(0, 0.3)); g.SetCameraParameters(fov, focal, ap)
    elif op == 7:
        scn = scenes.random_triangles(int(rng.integers(1, 400)), int(rng.integers(0, 1000))) if rng.integers(0, 2) else scenes.cornell32()
        g.UploadScene(scn)
    else:                                   # destroy the tracer while its render thread may be running; start over
        if rng.integers(0, 4) == 0:
            g.close()
            angles = [0.0, 0.0]
            g = R.RayTracer((W, H), (0, 0, 0), tuple(angles), fov, focal, ap, seed=int(rng.integers(0, 100)))
            g.UploadScene(scn); g.SetUpdateCallback(on_up); g.SetFinishedCallback(on_fin)
    if rng.integers(0, 3) == 0:
        time.sleep(float(rng.uniform(0, 0.003)))
g.Wait()
err = g.LastError()
# final, quiet trace against the oracle (fresh tracer state: RNG states were advanced by the soak, so compare a new pair)
g2 = R.RayTracer((W, H), (0, 0, 0), (0.0, 0.0), fov, focal, ap, seed=9)
g2.UploadScene(scn); g2.Trace(7, 2, 3); assert g2.Wait()
o = orc.OracleTracer(W, H, (0.0, 0.0), fov, focal, ap, seed=9, nthreads=8); o.upload_scene(scn); o.trace(7, 2)
ok = np.array_equal(g2.RenderBuffer().view(np.uint32), o.render.view(np.uint32)) and np.array_equal(g2.Image(), o.image)
print("soak: %d calls in %.1f s, %d updates, %d finished callbacks, last error %r, final parity %s" % (rounds, time.time() - t0, ups[0], fins[0], err, ok))
sys.exit(0 if ok else 1)

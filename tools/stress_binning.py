#!/usr/bin/env python3
"""Randomised campaign: the default kernel (per-tile classification + ballot early-outs) must
equal the plain full scan in reference order (no binning, no filter) bit for bit, for random
scenes, scales, cameras, image sizes and both arithmetic modes.  Usage: stress_binning.py [N] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracertest_amd as R
from raytracertest_amd import scenes

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
bad = 0
t0 = time.time()
for it in range(N):
    n = int(rng.choice([1, 3, 17, 64, 65, 200, 300, 700, 1500, 4000, 4096, 6000, 12000]))   # >= 4096: the dense-scene kernels with per-sample forms
    scale = float(10.0 ** rng.uniform(-2, 2))
    kind = rng.integers(0, 5)
    c = rng.uniform(-1, 1, (n, 1, 3)) * scale * rng.choice([0.3, 1.0, 4.0])
    if kind == 0:
        c[:, :, 2] -= 2.0 * scale                       # in front of the camera
    elif kind == 1:
        c[:, :, 2] += 2.0 * scale                       # behind it (negative t hits count, Q4)
    size = scale * float(10.0 ** rng.uniform(-2.5, 0.5))
    tri = c + rng.uniform(-1, 1, (n, 3, 3)) * size
    if kind == 3:
        tri[:, :, rng.integers(0, 3)] *= 1e-5           # nearly degenerate / edge-on sheets
    if kind == 4 and n > 8:
        tri[::5] = tri[::5][:, [0, 0, 1]]              # exactly degenerate triangles
    scn = scenes._tri_rows(tri)
    W, H = int(rng.integers(9, 97)), int(rng.integers(9, 65))
    cam = dict(angles=(float(rng.uniform(-3.2, 3.2)), float(rng.uniform(-3.2, 3.2))), fov=float(rng.uniform(5, 150)),
               focal=float(scale * 10.0 ** rng.uniform(-1, 1)), aperture=float(scale * rng.choice([0.0, 0.01, 0.1, 1.0, 5.0])))
    mode = int(rng.integers(0, 2)); spp = int(rng.integers(1, 7)); iters = int(rng.integers(1, 3)); seed = int(rng.integers(1, 1 << 30))
    near = bool(rng.integers(0, 4) == 0)
    out = []
    for plain in (False, True):
        g = R.RayTracer((W, H), (0, 0, 0), cam["angles"], cam["fov"], cam["focal"], cam["aperture"], seed=seed,
                        math_mode=mode, no_binning=plain, no_filter=plain, nearest_hit=near)
        g.UploadScene(scn)
        g.Trace(iters, spp, 0); assert g.Wait()
        out.append((g.RenderBuffer().view(np.uint32), g.RngStates(), g.Image()))
        g.close()
    same = all(np.array_equal(a, b) for a, b in zip(out[0], out[1]))
    if not same:
        bad += 1
        d = np.argwhere(out[0][0] != out[1][0])
        print("MISMATCH it=%d n=%d scale=%g kind=%d %dx%d cam=%s mode=%d spp=%d near=%s seed=%d: %d values, first %s"
              % (it, n, scale, kind, W, H, cam, mode, spp, near, seed, d.shape[0], d[:3].tolist()), flush=True)
    if it % 25 == 24:
        print("... %d/%d done, %d mismatches, %.0f s" % (it + 1, N, bad, time.time() - t0), flush=True)
print("stress_binning: %d configurations, %d mismatches" % (N, bad))
sys.exit(1 if bad else 0)

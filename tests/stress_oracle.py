#!/usr/bin/env python3
"""Randomised campaign HIP (default kernel) vs the CPU oracle, bit for bit: random scenes,
spheres, cameras, image sizes, band splits, seeds, both arithmetic modes, both hit rules.
Lives under tests/ because it uses the oracle.  Usage: python tests/stress_oracle.py [N] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import raytracertest_amd as R
from raytracertest_amd import scenes
from oracle import oracle_py as orc

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
bad = 0
t0 = time.time()
for it in range(N):
    n = int(rng.choice([1, 5, 33, 120, 400]))
    scale = float(10.0 ** rng.uniform(-1, 1))
    c = rng.uniform(-1.5, 1.5, (n, 1, 3)) * scale
    c[:, :, 2] -= rng.choice([0.0, 2.5]) * scale
    tri = c + rng.uniform(-1, 1, (n, 3, 3)) * scale * float(10.0 ** rng.uniform(-1.5, 0.3))
    scn = scenes._tri_rows(tri)
    sph = None
    if rng.integers(0, 3) == 0:
        sph = np.concatenate([rng.uniform(-1, 1, (2, 3)) * scale - [0, 0, 2 * scale], rng.uniform(0.1, 0.6, (2, 1)) * scale], 1).astype(np.float32)
    W, H = int(rng.integers(9, 70)), int(rng.integers(9, 50))
    row0 = int(rng.integers(0, H // 2)); rows = int(rng.integers(1, H - row0 + 1))
    angles = (float(rng.uniform(-3.2, 3.2)), float(rng.uniform(-3.2, 3.2)))
    fov, focal = float(rng.uniform(10, 140)), float(scale * 10.0 ** rng.uniform(-0.7, 0.7))
    ap = float(scale * rng.choice([0.0, 0.02, 0.3, 2.0]))
    mode, near = int(rng.integers(0, 2)), bool(rng.integers(0, 3) == 0)
    spp, iters, seed = int(rng.integers(1, 6)), int(rng.integers(1, 3)), int(rng.integers(1, 1 << 40))
    g = R.RayTracer((W, rows), (0, 0, 0), angles, fov, focal, ap, seed=seed, math_mode=mode, nearest_hit=near,
                    full_height=H, row_begin=row0)
    o = orc.OracleTracer(W, H, angles, fov, focal, ap, seed=seed, row0=row0, rows=rows, contract=1 - mode, nthreads=8,
                         hit_mode=1 if near else 0)
    g.UploadScene(scn); o.upload_scene(scn)
    if sph is not None:
        g.UploadSpheres(sph); o.upload_spheres(sph)
    g.RotateCamera((0.05, -0.1)); o.rotate_camera((0.05, -0.1))
    g.Trace(iters, spp, 0); assert g.Wait()
    o.trace(iters, spp)
    ok = (np.array_equal(g.RenderBuffer().view(np.uint32), o.render.view(np.uint32)) and np.array_equal(g.RngStates(), o.rng)
          and np.array_equal(g.SampleCounts(), o.counts) and np.array_equal(g.Image(), o.image))
    g.close()
    if not ok:
        bad += 1
        print("MISMATCH it=%d n=%d scale=%g %dx%d band %d+%d angles=%s fov=%g focal=%g ap=%g mode=%d near=%s spp=%d seed=%d"
              % (it, n, scale, W, H, row0, rows, angles, fov, focal, ap, mode, near, spp, seed), flush=True)
print("stress_oracle: %d configurations, %d mismatches, %.0f s" % (N, bad, time.time() - t0))
sys.exit(1 if bad else 0)

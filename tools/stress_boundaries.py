#!/usr/bin/env python3
"""Adversarial campaign for the conservative classification: scenes aimed at its decision boundaries
(tests/adversarial.py) through the default kernel against the plain reference-order full scan
(RT_FLAG_NO_BINNING | RT_FLAG_NO_FILTER), bit for bit.  Usage: stress_boundaries.py [N] [seed] [share of large scenes]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import raytracertest_amd as R
from raytracertest_amd import scenes
from adversarial import adversarial_config

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2027)
large_share = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
bad = 0
dropped = kept = 0
t0 = time.time()
for it in range(N):
    c = adversarial_config(rng, large=rng.uniform() < large_share)
    scn = scenes._tri_rows(c["tris"])
    out = []
    for plain in (False, True):
        g = R.RayTracer((c["W"], c["H"]), (0, 0, 0), c["cam"]["angles"], c["cam"]["fov"], c["cam"]["focal"], c["cam"]["aperture"],
                        seed=c["seed"], math_mode=c["mode"], no_binning=plain, no_filter=plain, nearest_hit=c["nearest"])
        g.UploadScene(scn)
        g.Trace(c["iters"], c["spp"], 0); assert g.Wait()
        out.append((g.RenderBuffer().view(np.uint32), g.RngStates(), g.Image()))
        if not plain and it % 10 == 0:                      # how selective the classification is on these scenes
            st = g.TraceStats(1)
            kept += st["bin_candidates"]; dropped += max(st["bin_rounds"], 1) * c["tris"].shape[0] - st["bin_candidates"]
        g.close()
    if not all(np.array_equal(a, b) for a, b in zip(out[0], out[1])):
        bad += 1
        d = np.argwhere(out[0][0] != out[1][0])
        print("MISMATCH it=%d n=%d scale=%g %dx%d cam=%s mode=%d spp=%d near=%s seed=%d: %d values, first %s"
              % (it, c["tris"].shape[0], c["scale"], c["W"], c["H"], c["cam"], c["mode"], c["spp"], c["nearest"], c["seed"],
                 d.shape[0], d[:3].tolist()), flush=True)
    if it % 200 == 199:
        print("... %d/%d done, %d mismatches, %.0f s" % (it + 1, N, bad, time.time() - t0), flush=True)
print("stress_boundaries: %d configurations, %d mismatches (sampled: %.1f %% of the tile x triangle pairs dropped by the classification)"
      % (N, bad, 100.0 * dropped / max(dropped + kept, 1)))
sys.exit(1 if bad else 0)

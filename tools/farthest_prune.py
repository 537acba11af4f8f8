#!/usr/bin/env python3
"""How much would 'a candidate that is certainly nearer than a certainly-hit triangle can never be the farthest hit' prune?
For sampled regions of the C4 frame at the three levels (rt_dbg_classify): kept triangles, certainly-hit ones, and how many kept
triangles have q_hi below the largest q_lo of a certainly-hit one (by the winner rule's 1e-4 relative margin).
  python3 tools/farthest_prune.py [config] [regions per level]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracertest_amd as R
from raytracertest_amd import scenes

config = sys.argv[1] if len(sys.argv) > 1 else "C4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
cfg = scenes.CONFIGS[config]
tris, _ = scenes.scene_for(config)
W, H = cfg["width"], cfg["height"]
g = R.RayTracer((W, H), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=1)
g.UploadScene(tris)
rng = np.random.default_rng(5)
for level, (rw, rh) in ((2, (128, 64)), (1, (32, 8)), (0, (8, 8))):
    xs = rng.integers(0, W // rw, n) * rw; ys = rng.integers(0, H // rh, n) * rh
    kept = []; sure = []; pruned = []
    for i in range(0, n, 20):
        hdr, rec = g.DebugClassify(np.stack([xs[i:i + 20], ys[i:i + 20]], 1), level, False, 1000)
        for r in rec:
            fl = r[:, 0].astype(np.int32)
            k, s = (fl & 1) != 0, (fl & 3) == 3
            qlo, qhi = r[:, 7], r[:, 8]
            Q = qlo[s].max() if s.any() else -np.inf
            dead = k & (qhi < Q - 1e-4 * (np.abs(qhi) + abs(Q))) if s.any() else np.zeros_like(k)
            kept.append(k.sum()); sure.append(s.sum()); pruned.append(dead.sum())
    kept, sure, pruned = map(np.array, (kept, sure, pruned))
    print("level %d (%dx%d): kept mean %.1f, certainly hit mean %.2f (regions with one: %.2f), prunable mean %.1f = %.1f %% of kept" % (
        level, rw, rh, kept.mean(), sure.mean(), (sure > 0).mean(), pruned.mean(), 100.0 * pruned.sum() / max(1, kept.sum())))

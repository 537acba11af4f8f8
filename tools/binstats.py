#!/usr/bin/env python3
"""Where the candidates of the classified path die (wave-level stage counters) at C3 and C4."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracertest_amd as R
from raytracertest_amd import scenes
for name in ("C3", "C4"):
    cfg = scenes.CONFIGS[name]
    tris, _ = scenes.scene_for(name)
    g = R.RayTracer((cfg["width"], cfg["height"]), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=1)
    g.UploadScene(tris)
    spp = 4
    st = g.TraceStats(spp)
    tiles = ((cfg["width"] + 7) // 8) * ((cfg["height"] + 7) // 8)
    print(name, "tiles", tiles, {k: v for k, v in st.items()})
    cand = st["bin_candidates"]
    tot = max(st["skip_a"] + st["skip_b"] + st["skip_c"] + st["reach_d"], 1)
    print("  candidates/tile %.2f; of the candidate tests (per sample batch): skipped after A %.3f, B %.3f, C %.3f, reached D %.3f" % (
        cand / tiles, *[st[k] / tot for k in ("skip_a", "skip_b", "skip_c", "reach_d")]))

#!/usr/bin/env python3
"""Exit-point and wave-skip statistics of one launch (instrumented kernel)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracertest_amd as R
from raytracertest_amd import scenes

for config in sys.argv[1:] or ["C3"]:
    cfg = dict(scenes.CONFIGS[config]); tris, sph = scenes.scene_for(config)
    out = {"config": config}
    for nofilter in (True, False):
        g = R.RayTracer((cfg["width"], cfg["height"]), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"],
                        cfg["aperture"], seed=cfg["seed"], no_filter=nofilter)
        if tris.shape[0]: g.UploadScene(tris)
        st = g.TraceStats(cfg["samples"] if config != "C4" else 4)
        if nofilter:
            out.update({k: st[k] for k in ("exit_det", "exit_u", "exit_v", "exit_hit")})
        else:
            out.update({k: st[k] for k in ("skip_a", "skip_b", "skip_c", "reach_d")})
        g.close()
    tests = out["exit_det"] + out["exit_u"] + out["exit_v"] + out["exit_hit"]
    wt = out["skip_a"] + out["skip_b"] + out["skip_c"] + out["reach_d"]
    out["lane_fracs"] = {k: round(out[k] / tests, 4) for k in ("exit_det", "exit_u", "exit_v", "exit_hit")}
    out["wave_fracs"] = {k: round(out[k] / wt, 4) for k in ("skip_a", "skip_b", "skip_c", "reach_d")}
    out["alg_flop_per_test"] = round((20 * out["exit_det"] + 30 * out["exit_u"] + 46 * out["exit_v"] + 52 * out["exit_hit"]) / tests, 2)
    print(json.dumps(out))

#!/usr/bin/env python3
"""Campaign form of tests/test_gpu_classification.py: N adversarial + N/2 cover configurations, every tile x every triangle
(small scenes) or sampled regions of all levels (large scenes, with the per-sample forms), against the reference's own
per-ray arithmetic (tests/classification_check.py).  Prints and writes the margin record: violations per allowance scale,
the smallest passing scale, and how far the reference's values reach into the allowances ("needed scale").
Usage: classification_margin.py [N] [seed] [out.json] [share of the adversarial configurations that are dense scenes, default 0.04]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import raytracertest_amd as R
from raytracertest_amd import scenes
from oracle import oracle_py as orc
import classification_check as cc
from adversarial import adversarial_config, cover_config

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 31
out_path = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", "classification_margin.json")
large_share = float(sys.argv[4]) if len(sys.argv) > 4 else 0.04
rng = np.random.default_rng(seed)
lens = cc.lens_samples(orc, seed=seed, pixel_index=3)
small = np.concatenate([lens[:1], lens[1:113:2], lens[-8:]])
tallies = {"small_scenes": cc.Tally(), "cover_scenes": cc.Tally(), "small_scenes_region_level": cc.Tally(), "large_scenes_wave_forms": cc.Tally(),
           "large_scenes_block": cc.Tally(), "large_scenes_macro": cc.Tally(), "large_scenes_super": cc.Tally()}
t0 = time.time()
for it in range(N + N // 2):
    kind = "adv" if it < N else "cover"
    large = kind == "adv" and rng.uniform() < large_share
    c = cover_config(rng) if kind == "cover" else adversarial_config(rng, large=large)
    n = c["tris"].shape[0]
    g = R.RayTracer((c["W"], c["H"]), (0, 0, 0), c["cam"]["angles"], c["cam"]["fov"], c["cam"]["focal"], c["cam"]["aperture"],
                    seed=c["seed"], math_mode=c["mode"])
    o = orc.OracleTracer(c["W"], c["H"], c["cam"]["angles"], c["cam"]["fov"], c["cam"]["focal"], c["cam"]["aperture"], seed=c["seed"],
                         contract=1 - c["mode"], nthreads=1, rows=min(8, c["H"]))
    scn = scenes._tri_rows(c["tris"])
    assert g.UploadScene(scn) and o.upload_scene(scn)
    tag = "%s%d(n=%d, scale=%.3g, seed=%d)" % (kind, it, n, c["scale"], seed)
    tiles = [(x, y) for y in range(0, c["H"], 8) for x in range(0, c["W"], 8)]
    if large:
        pick = [tiles[i] for i in rng.choice(len(tiles), min(6, len(tiles)), replace=False)]
        tallies["large_scenes_wave_forms"].merge(cc.run(g, o, pick, 0, small, forms=True, ladder=(1000, 100, 0), tag=tag))
        blocks = [(x, y) for y in range(0, c["H"], 8) for x in range(0, c["W"], 32)]
        pick = [blocks[i] for i in rng.choice(len(blocks), min(3, len(blocks)), replace=False)]
        tallies["large_scenes_block"].merge(cc.run(g, o, pick, 1, small, ladder=(1000, 100, 0), tag=tag))
        tallies["large_scenes_macro"].merge(cc.run(g, o, [(0, 0)], 2, small, ladder=(1000, 100, 0), max_pixels=256, tag=tag))
        tallies["large_scenes_super"].merge(cc.run(g, o, [(0, 0)], 4, small, ladder=(1000, 100, 0), max_pixels=256, tag=tag))   # (4 x 4 macro tiles, clipped by the frame)
    else:
        stored = None
        if n <= 256:
            g.Trace(1, 1, 0); assert g.Wait()
            stored = g.DebugTileListWords()
        tallies["cover_scenes" if kind == "cover" else "small_scenes"].merge(cc.run(g, o, tiles, 0, lens, stored=stored, tag=tag))
        if n <= 256:
            regs = [(x, y) for y in range(0, c["H"], 16) for x in range(0, c["W"], 32)]
            tallies["small_scenes_region_level"].merge(cc.run(g, o, regs, 3, lens, ladder=(1000, 100, 0), tag=tag))
    g.close()
    if it % 50 == 49:
        print("... %d/%d configurations, %.0f s; needed scale so far %.4f" % (it + 1, N + N // 2, time.time() - t0,
                                                                              max(t.needed for t in tallies.values())), flush=True)
doc = {"library": R.load_library().rt_version().decode(), "configurations": N + N // 2, "seed": seed,
       "lens_samples_per_pixel": int(lens.shape[0]), "seconds": round(time.time() - t0, 1)}
for k, t in tallies.items():
    doc[k] = t.summary()
allt = cc.Tally()
for t in tallies.values():
    allt.merge(t)
doc["all"] = allt.summary()
os.makedirs(os.path.dirname(out_path), exist_ok=True)
json.dump(doc, open(out_path, "w"), indent=1)
s = doc["all"]
print("classification_margin: %d configurations, %d regions, %d (tile, triangle) pairs, %.3g ray-triangle probes" % (
    doc["configurations"], s["regions"], s["tile_triangle_pairs"], s["rays_per_pair_total"]))
print("  violations at the product's allowances:", s["violations_by_scale"]["1.0"], "containment", s["containment_violations_at_scale_1"],
      "q", s["q_violations_at_scale_1"], "forms", s["forms"]["rejected_but_hit"])
print("  smallest passing scale of the ladder: %s; needed scale %.4f (by quantity %s)" % (s["smallest_passing_scale"], s["needed_scale"], s["needed_scale_by_quantity"]))
bad = any(s["violations_by_scale"]["1.0"].values()) or s["containment_violations_at_scale_1"] or s["q_violations_at_scale_1"] or s["forms"]["rejected_but_hit"]
sys.exit(1 if bad else 0)

"""The conservative classification checked verdict by verdict -- per (tile, triangle), not per image -- against the
reference's own HitTriangle arithmetic (RayTracer/Kernels.cuh:29-65) and farthest-hit scan (:73-92) as the oracle restates
them (orc_tile_probe).  What is checked and why: tests/classification_check.py; the numbers a run produces are written to
gpurun_out/classification_margin_test.json (the campaign form is tools/classification_margin.py -> profiles/, CLASSIFICATION.md).

The reference tests every triangle for every ray; the product drops ~97 % of those tests on interval proofs and skips the rays
of half of C3's tiles.  Image equality is blind to a wrongly dropped triangle that is not the farthest hit, these tests are not."""
import json
import os

import numpy as np
import pytest

import classification_check as cc
from adversarial import adversarial_config, cover_config

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = {}
N_ADV, N_COVER = 100, 60


@pytest.fixture(scope="module")
def rt():
    import raytracertest_amd as R
    from raytracertest_amd import api
    assert R.device_count() >= 1, "no HIP device: the GPU tests need the real extension"
    return api


@pytest.fixture(scope="module", autouse=True)
def write_report():
    yield
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "classification_margin_test.json"), "w") as f:
        json.dump(REPORT, f, indent=1)


def pair(rt, orc, W, H, scn, cam, mode, seed=1, **kw):
    g = rt.RayTracer((W, H), (0, 0, 0), cam["angles"], cam["fov"], cam["focal"], cam["aperture"], seed=seed, math_mode=mode, **kw)
    # (the probe needs the oracle's camera, scene and frame size only: an 8-row band keeps its buffers and RNG states small)
    o = orc.OracleTracer(W, H, cam["angles"], cam["fov"], cam["focal"], cam["aperture"], seed=seed, contract=1 - mode, nthreads=1,
                         rows=min(8, H))
    assert g.UploadScene(scn) and o.upload_scene(scn)
    return g, o


def all_tiles(W, H, step=1):
    return [(x, y) for y in range(0, H, 8 * step) for x in range(0, W, 8 * step)]


def clean(t, what):
    """the bar at the product's allowances: no verdict contradicted by a ray, every reference value inside its interval"""
    s = t.summary()
    assert not any(t.bad[1000].values()), "%s: verdicts contradicted by the reference's rays: %s %s" % (what, t.bad[1000], t.examples)
    assert t.contain_bad == 0 and t.q_bad == 0, "%s: reference values outside the exported intervals: %s" % (what, t.examples)
    assert t.form_wrong == 0, "%s: the per-sample forms skip rays the reference hits: %s" % (what, t.examples)
    return s


def test_adversarial_scenes_per_tile_and_triangle(rt, orc):
    """100 adversarial configurations + 60 aimed at the certain-winner verdict (tests/adversarial.py: triangles aimed at the decision boundaries of the tiles' ray
    families, coordinate scales 1e-3 ... 1e4, apertures 0 ... 100 x the scene, both arithmetic modes): every tile of the
    frame x every triangle.  Small scenes also cross-check rt_dbg_classify against the lists and certain-winner verdicts the
    PRODUCT launch stored (rt_dbg_read_tile_lists): the harness sees what the trace kernel decided."""
    from raytracertest_amd import scenes
    rng = np.random.default_rng(20261)
    lens = cc.lens_samples(orc, seed=5, pixel_index=11)
    total, regions, onepass = cc.Tally(), cc.Tally(), 0
    for it in range(N_ADV + N_COVER):
        c = adversarial_config(rng) if it < N_ADV else cover_config(rng)
        n = c["tris"].shape[0]
        g, o = pair(rt, orc, c["W"], c["H"], scenes._tri_rows(c["tris"]), c["cam"], c["mode"], seed=c["seed"])
        stored = None
        if n <= 256:                                        # the product's own verdicts: one launch that stores its tiles' lists
            g.Trace(1, 1, 0); assert g.Wait()
            stored = g.DebugTileListWords()
            onepass += 1
        t = cc.run(g, o, all_tiles(c["W"], c["H"]), 0, lens, stored=stored, tag="adv%d(n=%d, scale=%.3g)" % (it, n, c["scale"]))
        total.merge(t)
        if n <= 256:                                        # the region level of the two-level list builder
            regions.merge(cc.run(g, o, [(x, y) for y in range(0, c["H"], 16) for x in range(0, c["W"], 32)], 3, lens, ladder=(1000, 0),
                                 tag="adv%d region" % it))
        if n > 256:                                         # block level of the larger scenes
            total.merge(cc.run(g, o, [(x, y) for y in range(0, c["H"], 8) for x in range(0, c["W"], 32)], 1, lens, tag="adv%d" % it))
        g.close()
    s = clean(total, "adversarial scenes")
    s["configurations"], s["with_product_lists_cross_checked"] = N_ADV + N_COVER, onepass
    REPORT["adversarial"] = s
    REPORT["adversarial_regions"] = clean(regions, "adversarial scenes, region level")
    assert total.dropped > 0 and total.sure > 100 and total.sure_tiles > 50
    # teeth: with no rounding allowance at all the reference's values DO leave the intervals (the test can see rounding),
    # and the product charges a multiple of what they need
    assert total.needed > 0.0, "the probe never reaches the zero-allowance intervals: no teeth"
    assert total.needed < 0.5, "the reference's values use more than half of the rounding allowance: %s" % s


def test_c3_every_tile_of_the_benchmarked_frame(rt, orc):
    """BASELINE configs[2] at full size: all 32 400 wave tiles x 32 triangles.  The product's stored lists and certain-winner
    verdicts (67.4 % of the tiles) are cross-checked tile by tile; every tile is probed with 65 lens samples, every 8th
    with the full set."""
    from raytracertest_amd import scenes
    cfg = scenes.CONFIGS["C3"]
    cam = dict(angles=cfg["angles"], fov=cfg["fov"], focal=cfg["focal"], aperture=cfg["aperture"])
    g, o = pair(rt, orc, cfg["width"], cfg["height"], scenes.cornell32(), cam, 0, seed=cfg["seed"])
    g.Trace(1, 1, 0); assert g.Wait()
    stored = g.DebugTileListWords()
    lens = cc.lens_samples(orc, seed=1, pixel_index=0)
    small = np.concatenate([lens[:1], lens[1:113:2], lens[-8:]])
    tiles = all_tiles(cfg["width"], cfg["height"])
    t = cc.run(g, o, tiles, 0, small, stored=stored, ladder=(1000, 0), tag="C3")
    t2 = cc.run(g, o, tiles[::8], 0, lens, stored=stored, tag="C3")
    t3 = cc.run(g, o, [(x, y) for y in range(0, cfg["height"], 16) for x in range(0, cfg["width"], 32)], 3, small, ladder=(1000, 0),
                max_pixels=128, tag="C3 region")
    REPORT["C3_regions"] = clean(t3, "C3 regions")
    g.close()
    s = clean(t, "C3 all tiles")
    s2 = clean(t2, "C3 every 8th tile, full lens set")
    REPORT["C3_all_tiles"], REPORT["C3_every_8th_tile_full_lens_and_ladder"] = s, s2
    assert t.regions == 32400 and t.sure_tiles > 21000 and t.dropped > 0.9 * t.pairs


def test_c4_scene_all_three_levels_and_the_forms(rt, orc):
    """BASELINE configs[3] (10 000 triangles, 3840x2160): wave tiles with the per-sample forms (the instantiation the dense-scene
    kernel classifies with: every ray the forms skip must be a miss), wave tiles without, blocks, macro tiles and super tiles -- each level's
    drops against the rays of that level's own region."""
    from raytracertest_amd import scenes
    cfg = scenes.CONFIGS["C4"]
    cam = dict(angles=cfg["angles"], fov=cfg["fov"], focal=cfg["focal"], aperture=cfg["aperture"])
    tris, _ = scenes.scene_for("C4")
    g, o = pair(rt, orc, cfg["width"], cfg["height"], tris, cam, 0, seed=cfg["seed"])
    rng = np.random.default_rng(404)
    lens = cc.lens_samples(orc, seed=1, pixel_index=0)
    small = np.concatenate([lens[:1], lens[1:113:2], lens[-8:]])
    W, H = cfg["width"], cfg["height"]
    tiles = [(int(rng.integers(0, W // 8)) * 8, int(rng.integers(0, H // 8)) * 8) for _ in range(20)]
    out = {}
    t = cc.run(g, o, tiles[:12], 0, lens, forms=True, ladder=(1000, 0), tag="C4 forms")
    out["wave_tiles_with_forms"] = clean(t, "C4 wave tiles, forms")
    assert t.form_rejects > 0.5 * t.form_tests > 0, "the forms should skip most candidate tests of a tile"
    t = cc.run(g, o, tiles[12:], 0, lens, ladder=(1000, 100, 0), tag="C4")
    out["wave_tiles"] = clean(t, "C4 wave tiles")
    blocks = [(int(rng.integers(0, W // 32)) * 32, int(rng.integers(0, H // 8)) * 8) for _ in range(6)]
    t = cc.run(g, o, blocks, 1, small, ladder=(1000, 0), tag="C4")
    out["blocks"] = clean(t, "C4 blocks")
    macros = [(int(rng.integers(0, W // 128)) * 128, int(rng.integers(0, (H + 63) // 64)) * 64) for _ in range(3)]
    t = cc.run(g, o, macros, 2, small, ladder=(1000, 0), max_pixels=192, tag="C4")
    out["macro_tiles"] = clean(t, "C4 macro tiles")
    assert t.dropped > 0.8 * t.pairs
    supers = [(int(rng.integers(0, (W + 511) // 512)) * 512, int(rng.integers(0, (H + 255) // 256)) * 256) for _ in range(2)] + [(3584, 2048)]
    t = cc.run(g, o, supers, 4, small, ladder=(1000, 0), max_pixels=256, tag="C4")       # (the last one is clipped by the frame)
    out["super_tiles"] = clean(t, "C4 super tiles")
    assert t.dropped > 0.5 * t.pairs
    g.close()
    REPORT["C4"] = out


def test_harness_has_teeth_on_decisions(rt, orc):
    """The same check must FAIL when the verdicts are wrong: a scene whose triangles are shifted by a third of a tile after the
    device classified the original -- rays now hit dropped triangles and miss certainly-hit ones."""
    from raytracertest_amd import scenes
    cam = dict(angles=(0.0, 0.0), fov=70.0, focal=3.0, aperture=0.05)
    scn = scenes.cornell32()
    g, o = pair(rt, orc, 96, 64, scn, cam, 0)
    moved = scn.copy()
    moved[:, 0] += 0.08
    assert o.upload_scene(moved)
    lens = cc.lens_samples(orc)
    t = cc.run(g, o, all_tiles(96, 64), 0, lens, ladder=(1000,), tag="moved scene")
    g.close()
    assert t.bad[1000]["drop_hit"] > 0 and t.bad[1000]["sure_miss"] > 0 and t.contain_bad > 0

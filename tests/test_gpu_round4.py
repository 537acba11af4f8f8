"""Round-4 GPU parity tests: full-size evidence for the dense-scene classification.

The default C4 / C5 launch drops ~99.9 % of the reference's ray x triangle tests (RayTracer/Kernels.cuh:75-92) on interval
proofs at three levels (macro tile, block, wave tile) plus the per-sample forms.  Here the WHOLE 3840x2160 frame of the
default launch is compared bit for bit with the reference's algorithm on the same device -- RT_FLAG_NO_BINNING: every ray
tests every triangle, no classification at all (that path is oracle-checked on small frames and on the committed bands) --
and with the oracle's rows spread over the frame: first and last rows, macro-tile seams, the split row, random rows."""
import json
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


@pytest.fixture(scope="module")
def rt():
    import raytracertest_amd as R
    from raytracertest_amd import api
    assert R.device_count() >= 1, "no HIP device: the GPU tests need the real extension"
    return api


@pytest.fixture(scope="module")
def golden():
    from conftest import GOLDEN
    return json.load(open(os.path.join(GOLDEN, "frames.json"))), np.load(os.path.join(GOLDEN, "frames_crops.npz"))


def buffers(g):
    return g.RenderBuffer(), g.SampleCounts(), g.RngStates(), g.Image()


def check_bands(golden, prefix, bufs, row_offset=0, first_row=0, n_rows=None):
    """every committed `prefix`* band that lies inside rows [first_row, first_row + n_rows) of the frame against `bufs`
    (arrays whose row 0 is frame row `row_offset`); returns the bands checked"""
    frames, crops = golden
    render, counts, rng, image = bufs
    n_rows = render.shape[0] if n_rows is None else n_rows
    done = []
    for key in sorted(k for k in frames if k.startswith(prefix) and k.endswith("/fma")):
        m = frames[key]
        r0, n = m["spec"]["row0"], m["spec"]["rows"]
        if r0 < first_row or r0 + n > first_row + n_rows:
            continue
        band = lambda a: np.ascontiguousarray(a[r0 - row_offset:r0 - row_offset + n])
        c0 = m["crop_origin"][1]
        assert np.array_equal(band(render)[0:16, c0:c0 + 16].view(np.uint32)[:n], crops[key.replace("/", "__")][:n]), key
        assert crc(band(counts)) == m["counts_crc32"] and crc(band(rng)) == m["rng_crc32"], key
        assert crc(band(render)) == m["render_crc32"] and crc(band(image)) == m["image_crc32"], key
        done.append(key)
    return done


def test_c4_full_frame_default_launch_equals_the_full_scan_and_the_oracle_rows(rt, golden):
    """BASELINE configs[3] at full size: 3840x2160x64 spp, 10 000 triangles.  (a) the default launch (split, macro lists,
    block lists, wave lists, per-sample forms) == the reference's full scan (every ray x every triangle, 5.3e12 tests on the
    device), all four buffers of the whole frame bit for bit; (b) nine oracle-made 4-row bands of that frame -- rows 0-3,
    62-65 and 1022-1025 (macro-tile seams), 1078-1081 (the split row), 2156-2159 and four more spread over the frame."""
    from raytracertest_amd import scenes
    cfg = scenes.CONFIGS["C4"]
    tris, _ = scenes.scene_for("C4")

    def full(**kw):
        g = rt.RayTracer((cfg["width"], cfg["height"]), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=cfg["seed"], **kw)
        assert g.UploadScene(tris)
        g.SetListReuse(False)
        g.TraceEnqueue(1, cfg["samples"]); g.Sync()        # bench.py's step
        out = buffers(g)
        g.close()
        return out
    base = full()
    bands = check_bands(golden, "C4_", base)
    assert len(bands) >= 9, bands
    assert (base[1] == cfg["samples"]).all() and not base[0][..., 3].any()
    ref = full(no_binning=True)
    for name, a, b in zip(("render", "counts", "rng", "image"), base, ref):
        same = np.ascontiguousarray(a).view(np.uint32) == np.ascontiguousarray(b).view(np.uint32)
        assert same.all(), "%s: the default launch differs from the full scan in %d words, first at %s" % (
            name, int((~same).sum()), np.argwhere(~same)[0].tolist())


@pytest.mark.parametrize("band", [3, 0, 7])
def test_c5_band_default_launch_equals_the_full_scan_and_the_oracle_rows(rt, golden, band):
    """BASELINE configs[4]: one 270-row band of the 8-band partition of the 3840x2160 frame at 256 spp (band 3 ends at the
    rows of C5_band4; bands 0 and 7 hold the frame's first / last rows), default launch == full scan over the band, and
    the oracle's rows that lie inside it."""
    from raytracertest_amd import scenes
    cfg = scenes.CONFIGS["C5"]
    tris, _ = scenes.scene_for("C5")
    rows, row0 = 270, 270 * band

    def run(**kw):
        g = rt.RayTracer((cfg["width"], rows), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=cfg["seed"],
                         full_height=cfg["height"], row_begin=row0, **kw)
        assert g.UploadScene(tris)
        g.SetListReuse(False)
        g.TraceEnqueue(1, cfg["samples"]); g.Sync()
        out = buffers(g)
        g.close()
        return out
    base = run()
    assert (base[1] == cfg["samples"]).all()
    if band != 3:                                          # (C5_band4 = rows 1078..1081 straddles bands 3 and 4: checked in test_gpu_round2)
        assert check_bands(golden, "C5_", base, row_offset=row0, first_row=row0, n_rows=rows), "no committed C5 rows inside band %d" % band
    if band == 3:                                          # the full scan over one band: 2.7e12 tests
        ref = run(no_binning=True)
        for name, a, b in zip(("render", "counts", "rng", "image"), base, ref):
            assert np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32)), name


def test_c4_classification_on_a_stratified_set_of_wave_tiles(rt, orc):
    """>= 500 wave tiles of the C4 frame -- the four corners, tiles along all four edges, one random tile in every cell of a
    24 x 18 grid over the frame -- with the per-sample forms (the instantiation the dense-scene kernel classifies with): no
    dropped triangle is hit by a probed ray, no ray the forms skip is a hit, the reference's det / U / V lie inside the
    exported intervals (tests/classification_check.py)."""
    import classification_check as cc
    from raytracertest_amd import scenes
    from test_gpu_classification import clean, pair
    cfg = scenes.CONFIGS["C4"]
    cam = dict(angles=cfg["angles"], fov=cfg["fov"], focal=cfg["focal"], aperture=cfg["aperture"])
    tris, _ = scenes.scene_for("C4")
    g, o = pair(rt, orc, cfg["width"], cfg["height"], tris, cam, 0, seed=cfg["seed"])
    W, H = cfg["width"], cfg["height"]
    tx, ty = W // 8, H // 8
    rng = np.random.default_rng(2604)
    tiles = {(0, 0), (tx - 1, 0), (0, ty - 1), (tx - 1, ty - 1)}
    for k in range(30):
        x = int(rng.integers(0, tx))
        tiles.add((x, 0)); tiles.add((x, ty - 1))
    for k in range(20):
        y = int(rng.integers(0, ty))
        tiles.add((0, y)); tiles.add((tx - 1, y))
    for cy in range(18):
        for cx in range(24):
            tiles.add((cx * tx // 24 + int(rng.integers(0, tx // 24)), cy * ty // 18 + int(rng.integers(0, ty // 18))))
    tiles = sorted((x * 8, y * 8) for x, y in tiles)
    assert len(tiles) >= 500
    lens = cc.lens_samples(orc, seed=1, pixel_index=0)
    small = np.concatenate([lens[:1], lens[1:113:2], lens[-8:]])
    total = cc.Tally()
    for i in range(0, len(tiles), 64):                     # (a region's exported records are 1.3 MB: in batches)
        total.merge(cc.run(g, o, tiles[i:i + 64], 0, small, forms=True, ladder=(1000,), tag="C4 stratified"))
    g.close()
    s = clean(total, "C4 stratified wave tiles, forms")
    assert total.regions == len(tiles) and total.dropped > 0.99 * total.pairs and total.form_tests > 0
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "classification_c4_stratified.json"), "w") as f:
        json.dump(s, f, indent=1)


def test_trace_enqueue_n_equals_n_calls(rt, orc):
    """rt_tracer_trace_enqueue_n(t, iterations, samples, n) == n calls of rt_tracer_trace_enqueue (the loop bench.py times
    runs inside the library), at a split-launch size, lists rebuilt per step: every buffer against the oracle."""
    from raytracertest_amd import scenes
    W, H = 320, 200
    g = rt.RayTracer((W, H), (0, 0, 0), (0.0, 0.0), 70.0, 3.0, 0.05, seed=9)
    o = orc.OracleTracer(W, H, (0.0, 0.0), 70.0, 3.0, 0.05, seed=9, contract=1, nthreads=8)
    scn = scenes.cornell32()
    assert g.UploadScene(scn) and o.upload_scene(scn)
    g.SetListReuse(False)
    g.TraceEnqueueN(2, 3, 5); g.Sync()
    for _ in range(5):
        o.trace(2, 3)
    assert np.array_equal(g.RenderBuffer().view(np.uint32), o.render.view(np.uint32))
    assert np.array_equal(g.SampleCounts(), o.counts) and np.array_equal(g.RngStates(), o.rng) and np.array_equal(g.Image(), o.image)
    g.TraceEnqueueN(1, 4, 0); g.Sync()                     # zero steps: nothing happens
    assert np.array_equal(g.RngStates(), o.rng)
    g.close()


def test_render_thread_is_reused_and_keeps_the_contract(rt, orc):
    """The persistent render thread (one per tracer instead of one std::thread per Trace, RayTracerImpl.cu:69-87): many short
    Traces back to back, callbacks not on the caller's thread, a Trace that cancels a running one fires no finished callback for
    it, Wait() reports completion -- and the buffers equal the oracle's after the last Trace."""
    import threading
    from raytracertest_amd import scenes
    W, H = 96, 54
    g = rt.RayTracer((W, H), (0, 0, 0), (0.0, 0.0), 70.0, 3.0, 0.05, seed=4)
    o = orc.OracleTracer(W, H, (0.0, 0.0), 70.0, 3.0, 0.05, seed=4, contract=1, nthreads=4)
    scn = scenes.cornell32()
    assert g.UploadScene(scn) and o.upload_scene(scn)
    me = threading.get_ident()
    seen, threads = [], set()
    g.SetFinishedCallback(lambda img, size: (seen.append(size), threads.add(threading.get_ident())))
    for _ in range(40):
        g.Trace(1, 2, 0); assert g.Wait()
        o.trace(1, 2)
    assert len(seen) == 40 and me not in threads and len(threads) == 1      # one render thread served every Trace
    assert np.array_equal(g.RenderBuffer().view(np.uint32), o.render.view(np.uint32)) and np.array_equal(g.RngStates(), o.rng)
    n0 = len(seen)
    g.Trace(400000, 1, 0)                                  # a long Trace ...
    g.Trace(1, 1, 0)                                       # ... cancelled by the next one (:72-77): no finished callback for it
    assert g.Wait()
    assert len(seen) == n0 + 1
    g.Trace(300000, 1, 0); g.Stop()
    assert not g.Wait() and len(seen) == n0 + 1            # a stopped run fires no finished callback (:280-284)
    g.close()


def _dense_pair(rt, W, H, n_tris, seed, **kw):
    from raytracertest_amd import scenes
    scn = scenes.random_triangles(n_tris, seed)
    g = rt.RayTracer((W, H), (0, 0, 0), (0.1, -0.05), 70.0, 3.0, 0.05, seed=8, **kw)
    ref = rt.RayTracer((W, H), (0, 0, 0), (0.1, -0.05), 70.0, 3.0, 0.05, seed=8, no_binning=True)
    assert g.UploadScene(scn) and ref.UploadScene(scn)
    return g, ref


def _same(g, ref):
    for name, a, b in zip(("render", "counts", "rng", "image"), buffers(g), buffers(ref)):
        assert np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32)), name


def test_dense_lists_in_hbm_are_kept_across_launches_and_rebuilt_when_the_view_changes(rt):
    """Dense scenes keep their per-wave candidate lists + forms in HBM (wave_lists_kernel, keyed like the macro lists): a Trace
    of several launches classifies once, a camera change or a new scene rebuilds -- every state against the full scan."""
    from raytracertest_amd import scenes
    g, ref = _dense_pair(rt, 200, 136, 6000, 31)
    for t in (g, ref):
        t.Trace(20, 8, 0); assert t.Wait()                 # 160 spp: three launches of fused iterations, the lists built by the first
    _same(g, ref)
    for t in (g, ref):
        t.RotateCamera((0.07, 0.03))
        t.TraceEnqueue(2, 5); t.TraceEnqueue(1, 3); t.Sync()
    _same(g, ref)
    scn2 = scenes.random_triangles(5000, 99)
    for t in (g, ref):
        assert t.UploadScene(scn2)
        t.SetCameraParameters(60.0, 2.5, 0.08)
        t.Trace(3, 2, 2); assert t.Wait()                  # update cadence: un-fused launches
    _same(g, ref)
    for t in (g, ref):
        t.Resize((168, 130))
        t.Launch(4, clear_first=True); t.Launch(4, emit_image=True); t.Sync()
    _same(g, ref)
    g.close(); ref.close()


def test_dense_list_overflow_falls_back_to_the_macro_list(rt):
    """A tile whose candidates exceed the list capacity is marked by wave_lists_kernel and traced with the exact tests over its
    macro tile's list: 20 000 triangles on a 96x64 frame with the smallest capacity (32) overflow every tile; 6 000 on 200x136
    overflow some.  Both against the full scan, split and unsplit launches."""
    for W, H, n, seed in ((96, 64, 20000, 5), (200, 136, 6000, 31)):
        g, ref = _dense_pair(rt, W, H, n, seed, bin_list=32)
        for t in (g, ref):
            t.TraceEnqueue(2, 5); t.Sync()
        _same(g, ref)
        counts, cap = g.DebugWaveListCounts(0)
        over = counts == 0xFFFFFFFF
        assert cap == 32 and over.any() and (counts[~over] <= cap).all(), (cap, int(over.sum()), counts.size)
        if n == 20000:
            assert over.mean() > 0.9                       # the fallback is what this frame ran
        g.close(); ref.close()


def test_dense_launch_machinery_soak(rt):
    """tools/soak_dense.py: random Trace / TraceEnqueue / Launch with changing sample counts and update cadences, camera swings,
    lens changes, scene swaps between dense, mid-size and small scenes, Resize and list reuse on / off against the same operations
    on a full-scan tracer -- every phase bit for bit (the keys of the macro and per-wave lists, their reuse and rebuilds)."""
    import subprocess, sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_dense.py"), "50", "11"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "soak_dense ok: 50 phases" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_super_tiles_change_no_word_and_only_shorten_the_macro_level(rt):
    """Dense scenes bin the triangles per super tile (4 x 4 macro tiles, super_bin_kernel) before the macro tiles: the C4
    scene at 1920x1080x8 spp with the level and without it (RT_FLAG_NO_SUPER_BINS), lists rebuilt and kept, all four buffers;
    the per-tile candidate counts are the same."""
    from raytracertest_amd import scenes
    cfg = scenes.CONFIGS["C4"]
    tris, _ = scenes.scene_for("C4")

    def run(**kw):
        g = rt.RayTracer((1920, 1080), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=cfg["seed"], **kw)
        assert g.UploadScene(tris)
        g.SetListReuse(False)
        g.TraceEnqueue(1, 8); g.Sync()
        g.SetListReuse(True)
        g.TraceEnqueue(2, 4); g.Sync()                        # lists kept
        g.RotateCamera((0.05, -0.02))                         # a new view: every level rebuilds
        g.TraceEnqueue(1, 8); g.Sync()
        out = buffers(g)
        counts = [g.DebugWaveListCounts(h) for h in (0, 1)]
        g.close()
        return out, counts
    (a, ca), (b, cb) = run(), run(no_super_bins=True)
    for name, x, y in zip(("render", "counts", "rng", "image"), a, b):
        assert np.array_equal(np.ascontiguousarray(x).view(np.uint32), np.ascontiguousarray(y).view(np.uint32)), name
    for (x, capx), (y, capy) in zip(ca, cb):
        assert capx == capy and x.size > 0 and np.array_equal(x, y)


@pytest.mark.parametrize("n_tris,size,macro_cap", [(3000, (1024, 640), None), (60000, (768, 512), None), (5000, (1024, 576), "40"),
                                                   (2048, (1152, 256), None)])
def test_super_tiles_beside_every_kind_of_macro_consumer(rt, monkeypatch, n_tris, size, macro_cap):
    """The super level feeds the macro tiles of every large-scene launch, not only the dense scenes with lists in HBM: 3 000
    triangles (classification inside the trace kernel, no forms), 60 000 (59 chunks, multi-round wave lists), macro lists that
    overflow their capacity (the consumers scan the scene), a frame of one row of super tiles with exactly two chunks -- default
    == without the super level == the reference's full scan, render and RNG states."""
    from raytracertest_amd import scenes
    scn = scenes.random_triangles(n_tris, 4242 + n_tris)
    if macro_cap is not None:
        monkeypatch.setenv("RT_MI355X_MACRO_CAP", macro_cap)

    def run(**kw):
        g = rt.RayTracer(size, (0, 0, 0), (0.03, -0.02), 70.0, 3.0, 0.05, seed=5, **kw)
        assert g.UploadScene(scn)
        g.Trace(2, 3, 0); assert g.Wait()
        out = (g.RenderBuffer().view(np.uint32).copy(), g.RngStates().copy())
        g.close()
        return out
    ref = run(no_binning=True)
    for kw in (dict(), dict(no_super_bins=True)):
        got = run(**kw)
        assert np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1]), kw


def test_dense_scene_at_8k_equals_the_full_scan(rt):
    """Beyond the BASELINE sizes: 7680x4320 (33 M pixels, 518 400 tiles, 2.8 GB of list slots per half), the C4 scene at 4 spp --
    the default launch (super / macro / block / wave lists in HBM) == the reference's full scan, CRC32 of all four buffers."""
    import zlib
    from raytracertest_amd import scenes
    scn = scenes.random_triangles(10000, 12345)
    crcs = []
    for kw in (dict(), dict(no_binning=True)):
        g = rt.RayTracer((7680, 4320), (0, 0, 0), (0.02, -0.01), 70.0, 3.0, 0.05, seed=3, **kw)
        assert g.UploadScene(scn)
        g.TraceEnqueue(1, 4); g.Sync()
        crcs.append([zlib.crc32(np.ascontiguousarray(b).tobytes()) for b in (g.RenderBuffer(), g.SampleCounts(), g.RngStates(), g.Image())])
        g.close()
    assert crcs[0] == crcs[1]

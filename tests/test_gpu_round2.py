"""Round-2 GPU parity tests: zero-sample launches, split (two-stream) launches against the oracle, the
benchmarked shapes of C4 / C5 under an assertion, the frame sharded over devices through the native
entry point (rt_tracer_create_multi), and unit-level ray generation (rt_dbg_get_ray)."""
import os
import zlib

import numpy as np
import pytest

from test_gpu_parity import assert_frame_equal, scene, u32

pytestmark = pytest.mark.gpu


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


@pytest.fixture(scope="module")
def rt():
    import raytracertest_amd as R
    from raytracertest_amd import api
    assert R.device_count() >= 1, "no HIP device: the GPU tests need the real extension"
    return api


def pair(rt, orc, W, H, scn, *, seed=3, mode=0, focal=3.0, aperture=0.05, **kw):
    g = rt.RayTracer((W, H), (0, 0, 0), (0.0, 0.0), 70.0, focal, aperture, seed=seed, math_mode=mode, **kw)
    o = orc.OracleTracer(W, H, (0.0, 0.0), 70.0, focal, aperture, seed=seed, contract=1 - mode, nthreads=8)
    if scn is not None:
        assert g.UploadScene(scn) and o.upload_scene(scn)
    return g, o


# ------------------------------------------------------------------ zero samples
def test_zero_sample_trace_then_normal_trace(rt, orc):
    """Trace(1, 0, 0) is a real launch with sampleCount 0 (Kernels.cuh:133-146: counts += 0, render += 0, the
    converter divides by a zero count): buffers cleared, image defined, finished callback fired -- and the
    stored tile lists of the next Trace are the ones a launch really wrote."""
    g, o = pair(rt, orc, 70, 40, scene("cornell"))
    done = []
    g.SetFinishedCallback(lambda img, size: done.append(img.copy()))
    g.Trace(2, 3, 0); assert g.Wait()
    o.trace(2, 3)
    assert_frame_equal(g, o)
    g.Trace(1, 0, 0); assert g.Wait()
    o.trace(1, 0)
    assert_frame_equal(g, o)
    assert not g.RenderBuffer().any() and not g.SampleCounts().any()
    assert len(done) == 2 and np.array_equal(done[1], o.image)
    g.Trace(2, 2, 0); assert g.Wait()
    o.trace(2, 2)
    assert_frame_equal(g, o)
    g.Launch(0, clear_first=True, emit_image=True); g.Launch(2, emit_image=True); g.Sync()
    o.trace(1, 0); o.launch(2); orc.lib().orc_convert(__import__("ctypes").byref(o._frame))
    assert_frame_equal(g, o)
    g.TraceEnqueue(0, 5); g.Sync()
    o.trace(0, 5)
    assert_frame_equal(g, o)


# ------------------------------------------------------------------ split launches
@pytest.mark.parametrize("name,W,H", [("cornell", 96, 200), ("rand300", 70, 136)])
@pytest.mark.parametrize("mode", [0, 1])
def test_split_launches_follow_the_oracle(rt, orc, name, W, H, mode, monkeypatch):
    """Frames of 128 rows or more run as two half-frame kernels on two streams (TraceEnqueue / Launch); every
    combination with Trace() (never split), stored tile lists written by one form and read by the other, the image
    mirror, and list reuse on/off ends bit-identical to the oracle and to an RT_MI355X_NO_SPLIT=1 tracer."""
    g, o = pair(rt, orc, W, H, scene(name), mode=mode)
    g.TraceEnqueue(3, 2); g.Sync()                     # split: launch 0 classifies, launch 1 stores, launch 2 loads
    o.trace(3, 2)
    assert_frame_equal(g, o)
    g.Trace(2, 2, 0); assert g.Wait()                  # unsplit, loads the lists the split launches stored
    o.trace(2, 2)
    assert_frame_equal(g, o)
    g.SetListReuse(False)
    g.Trace(3, 1, 0); assert g.Wait()                  # unsplit stores ...
    o.trace(3, 1)
    assert_frame_equal(g, o)
    g.Launch(2, clear_first=True); g.Launch(2); g.Launch(2, emit_image=True); g.Sync()    # ... split loads
    o.trace(3, 2)
    assert_frame_equal(g, o)
    g.SetListReuse(True)
    # image mirror in device-visible memory: a second tracer's image buffer serves as the target
    g2 = rt.RayTracer((W, H), seed=1)
    g.SetImageMirror(g2.DevicePointer(rt.BUF_IMAGE))
    g.TraceEnqueue(2, 3); g.Sync()
    o.trace(2, 3)
    assert_frame_equal(g, o)
    assert np.array_equal(g2.Image(), o.image), "mirror written by both half-frame kernels"
    g.SetImageMirror(None)
    states = (g.RenderBuffer(), g.SampleCounts(), g.RngStates(), g.Image())
    monkeypatch.setenv("RT_MI355X_NO_SPLIT", "1")
    h, _ = pair(rt, orc, W, H, scene(name), mode=mode)
    monkeypatch.delenv("RT_MI355X_NO_SPLIT")
    for it, spp in ((3, 2), (2, 2), (3, 1), (3, 2), (2, 3)):
        h.TraceEnqueue(it, spp)
    h.Sync()
    for a, b in zip(states, (h.RenderBuffer(), h.SampleCounts(), h.RngStates(), h.Image())):
        assert np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def test_split_launch_on_a_4k_band_of_the_c4_scene(rt, orc):
    """The large-scene kernels (macro lists per half, block lists, per-sample forms) under a split launch: a 136-row
    band of the 3840x2160 frame of the 10k-triangle scene, two accumulating launches, against the oracle."""
    from raytracertest_amd import scenes
    tris = scenes.random_triangles(10000, 12345)
    W, H, row0, rows = 3840, 2160, 1012, 136
    g = rt.RayTracer((W, rows), (0, 0, 0), (0.0, 0.0), 70.0, 3.0, 0.05, seed=1, full_height=H, row_begin=row0)
    o = orc.OracleTracer(W, H, (0.0, 0.0), 70.0, 3.0, 0.05, seed=1, row0=row0, rows=rows, contract=1, nthreads=16)
    assert g.UploadScene(tris) and o.upload_scene(tris)
    g.TraceEnqueue(2, 2); g.Sync()
    o.trace(2, 2)
    assert_frame_equal(g, o)


# ------------------------------------------------------------------ C4 / C5 at the benchmarked shape
def _golden():
    import json
    from conftest import GOLDEN
    frames = json.load(open(os.path.join(GOLDEN, "frames.json")))
    crops = np.load(os.path.join(GOLDEN, "frames_crops.npz"))
    return frames, crops


def _check_band_of_full_frame(key, render, counts, rng, image):
    """rows 1078..1081 of a full 3840x2160 frame against the oracle-made band fixture"""
    frames, crops = _golden()
    m = frames[key]
    r0, n = m["spec"]["row0"], m["spec"]["rows"]
    band = lambda a: np.ascontiguousarray(a[r0:r0 + n])
    c0 = m["crop_origin"][1]
    assert np.array_equal(band(render)[0:16, c0:c0 + 16].view(np.uint32), crops[key.replace("/", "__")])
    assert crc(band(counts)) == m["counts_crc32"] and crc(band(rng)) == m["rng_crc32"]
    assert crc(band(render)) == m["render_crc32"] and crc(band(image)) == m["image_crc32"]


def test_c4_full_frame_default_launch_and_its_variants(rt, monkeypatch):
    """BASELINE configs[3] at the benchmarked shape: the full 3840x2160x64 frame through the default (split, macro
    lists, per-sample forms) launch reproduces the oracle's rows 1078-1081, every pixel got its 64 samples, and the
    whole frame is bit-identical with the split, the macro level and the forms switched off one at a time."""
    from raytracertest_amd import scenes
    tris = scenes.random_triangles(10000, 12345)

    def full(**kw):
        g = rt.RayTracer((3840, 2160), (0, 0, 0), (0.0, 0.0), 70.0, 3.0, 0.05, seed=1, **kw)
        assert g.UploadScene(tris)
        g.TraceEnqueue(1, 64); g.Sync()                # bench.py's step
        out = (g.RenderBuffer(), g.SampleCounts(), g.RngStates(), g.Image())
        g.close()
        return out
    base = full()
    _check_band_of_full_frame("C4_band4/fma", *base)
    assert (base[1] == 64).all() and not base[0][..., 3].any()
    ref = [crc(a) for a in base]
    monkeypatch.setenv("RT_MI355X_NO_SPLIT", "1")
    assert [crc(a) for a in full()] == ref
    monkeypatch.delenv("RT_MI355X_NO_SPLIT")
    assert [crc(a) for a in full(no_macro_bins=True)] == ref
    monkeypatch.setenv("RT_MI355X_NO_PRETEST", "1")
    assert [crc(a) for a in full()] == ref


def test_c5_band_and_eight_band_frame(rt):
    """BASELINE configs[4] (the C4 scene at 256 spp in 8 row bands): band mode on one GPU against the oracle's
    fixture, then the whole frame through the native multi-device entry point with 8 bands on this device."""
    from raytracertest_amd import scenes
    frames, crops = _golden()
    tris = scenes.random_triangles(10000, 12345)
    spec = frames["C5_band4/fma"]["spec"]
    g = rt.RayTracer((spec["W"], spec["rows"]), (0, 0, 0), (0.0, 0.0), 70.0, 3.0, 0.05, seed=1, full_height=spec["H"],
                     row_begin=spec["row0"])
    assert g.UploadScene(tris)
    g.Trace(1, 256, 0); assert g.Wait()
    m = frames["C5_band4/fma"]
    assert crc(g.RenderBuffer()) == m["render_crc32"] and crc(g.Image()) == m["image_crc32"]
    assert crc(g.RngStates()) == m["rng_crc32"] and crc(g.SampleCounts()) == m["counts_crc32"]
    g.close()
    # 8 bands of 270 rows: rows 1078..1081 lie inside band 3 (rows 810..1079) AND band 4 (1080..1349)
    mt = rt.RayTracer((3840, 2160), (0, 0, 0), (0.0, 0.0), 70.0, 3.0, 0.05, seed=1, devices=[0] * 8)
    assert mt.UploadScene(tris)
    mt.TraceEnqueue(1, 256); mt.Sync()
    render, counts, rng, image = mt.RenderBuffer(), mt.SampleCounts(), mt.RngStates(), mt.Image()
    _check_band_of_full_frame("C5_band4/fma", render, counts, rng, image)
    assert (counts == 256).all()
    assert [b["rows"] for b in mt.Bands()] == [270] * 8
    mt.close()


# ------------------------------------------------------------------ the frame sharded over devices, native entry point
@pytest.mark.parametrize("bands", [1, 2, 3, 8])
@pytest.mark.parametrize("name,W,H,it,spp", [("cornell", 67, 45, 3, 2), ("rand300", 96, 150, 2, 3)])
def test_multi_device_tracer_equals_single_tracer_and_oracle(rt, orc, bands, name, W, H, it, spp):
    """rt_tracer_create_multi with `bands` bands on this one device: buffers, RNG states and the GATHERED frame equal
    the oracle's whole frame (hence a single tracer's); repeated TraceEnqueue passes reuse both frame buffers."""
    n_dev = rt.device_count()
    g, o = pair(rt, orc, W, H, scene(name), devices=[k % n_dev for k in range(bands)])
    assert len(g.Bands()) == bands and sum(b["rows"] for b in g.Bands()) == H
    g.Trace(it, spp, 0); assert g.Wait()
    o.trace(it, spp)
    assert_frame_equal(g, o)
    for _ in range(3):
        g.TraceEnqueue(it, spp)
        o.trace(it, spp)
    g.Sync()
    assert_frame_equal(g, o)
    assert np.array_equal(g.Frame(), o.image)
    g.Launch(spp, clear_first=True); g.Launch(spp, emit_image=True); g.Sync()
    o.trace(2, spp)
    assert_frame_equal(g, o)
    g.close()


def test_multi_device_callbacks_resize_camera_and_stop(rt, orc):
    """The reference's caller through the multi-device handle: update cadence i > 0 && i % interval == 0 with ONE
    callback per update carrying the whole frame, finished callback, a stopped run fires none
    (RayTracerImpl.cu:256,280-305), Resize re-partitions the bands, camera changes reach every band."""
    W, H = 64, 37
    g, o = pair(rt, orc, W, H, scene("cornell"), devices=[0] * 5)
    updates, finished = [], []
    g.SetUpdateCallback(lambda img, size: updates.append((img.shape, size)))
    g.SetFinishedCallback(lambda img, size: finished.append(img.copy()))
    g.Trace(25, 1, 10); assert g.Wait()
    o.trace(25, 1)
    assert updates == [((H, W), W * H * 4)] * 2 and len(finished) == 1
    assert np.array_equal(finished[0], o.image)
    assert_frame_equal(g, o)
    g.RotateCamera((0.05, -0.1)); g.SetCameraParameters(55.0, 2.5, 0.1)
    o.rotate_camera((0.05, -0.1)); o.set_camera_parameters(55.0, 2.5, 0.1)
    g.Trace(3, 2, 0); assert g.Wait()
    o.trace(3, 2)
    assert_frame_equal(g, o)
    g.Resize((40, 23))
    o2 = orc.OracleTracer(40, 23, (0.0, 0.0), 70.0, 3.0, 0.05, seed=3, nthreads=4)
    o2.upload_scene(scene("cornell")); o2.rotate_camera((0.05, -0.1)); o2.set_camera_parameters(55.0, 2.5, 0.1)
    g.Trace(2, 2, 0); assert g.Wait()
    o2.trace(2, 2)
    assert_frame_equal(g, o2)
    assert np.array_equal(g.Frame(), o2.image)          # (Frame() after Resize uses the NEW height)
    assert [b["rows"] for b in g.Bands()] == [4, 5, 4, 5, 5]
    n_fin = len(finished)
    g.Trace(100000, 1, 0); g.Stop()
    assert not g.Wait() and len(finished) == n_fin       # stopped: no finished callback
    g.Trace(1, 0, 0); assert g.Wait()                     # zero samples through the bands
    o2.trace(1, 0)
    assert_frame_equal(g, o2)
    assert g.LastError() == ""
    g.close()


def test_multi_device_gather_through_rccl_on_one_device(rt, orc, monkeypatch):
    """RT_MI355X_GATHER_SELF=1 sends the tiles of root-local bands through the RCCL path too (a one-rank
    communicator, grouped ncclSend/ncclRecv to self): rehearses symbol loading, communicator creation, the
    call sequence and the send-buffer double buffering on a 1-GPU box."""
    monkeypatch.setenv("RT_MI355X_GATHER_SELF", "1")
    g, o = pair(rt, orc, 80, 50, scene("cornell"), devices=[0, 0, 0])
    monkeypatch.delenv("RT_MI355X_GATHER_SELF")
    for _ in range(4):
        g.TraceEnqueue(2, 2)
        o.trace(2, 2)
    g.Sync()
    assert_frame_equal(g, o)
    ms, n = g.GatherTime()
    assert n == 4 and ms > 0.0
    done = []
    g.SetFinishedCallback(lambda img, size: done.append(img.copy()))
    g.Trace(2, 1, 0); assert g.Wait()
    o.trace(2, 1)
    assert np.array_equal(done[0], o.image)
    g.close()


def test_group_member_api_single_rank(rt, orc):
    """rt_tracer_join_group with one rank: the member path (tile target, gather stream, RT_BUF_FRAME) without a peer."""
    g, o = pair(rt, orc, 48, 30, scene("cornell"))
    g.JoinGroup(1, 0)
    for _ in range(3):
        g.TraceEnqueue(2, 2)
        o.trace(2, 2)
    g.Sync()
    assert_frame_equal(g, o)
    assert np.array_equal(g.Frame(), o.image)
    g.LeaveGroup()
    g.TraceEnqueue(1, 1); g.Sync()
    o.trace(1, 1)
    assert_frame_equal(g, o)


# ------------------------------------------------------------------ R3/R4 at unit level
@pytest.mark.parametrize("mode", [0, 1])
def test_get_ray_on_device_equals_oracle_camera(rt, orc, mode):
    """ThinLensCamera::GetRay (ThinLensCamera.cuh:30-52,111-130) for single pixels with given RNG states:
    rt_dbg_get_ray against the oracle's orc_camera_get_ray, rays and advanced states bit for bit."""
    import ctypes as C
    W, H = 321, 123
    rng = np.random.default_rng(5)
    for angles, fov, focal, ap in (((0.0, 0.0), 70.0, 3.0, 0.05), ((0.3, -1.1), 35.0, 10.0, 4.0), ((-0.7, 2.9), 120.0, 0.5, 0.0)):
        g = rt.RayTracer((W, H), (0, 0, 0), angles, fov, focal, ap, seed=1, math_mode=mode)
        cam = orc.camera(angles, fov, focal, ap)
        pix = np.stack([rng.integers(0, W, 200), rng.integers(0, H, 200)], axis=1).astype(np.uint32)
        states = np.stack([orc.rng_init(7, int(p)) for p in rng.integers(0, 2**31, 200)])
        rays, st = g.DebugGetRay(pix, states)
        L = orc.lib()
        for i in range(200):
            s = states[i].copy()
            out = np.zeros(6, np.float32)
            L.orc_camera_get_ray(C.byref(cam), int(pix[i, 0]), int(pix[i, 1]), W, H, orc._up(s), 1 - mode, orc._fp(out))
            assert np.array_equal(out.view(np.uint32), rays[i].view(np.uint32)), (i, out, rays[i])
            assert np.array_equal(s, st[i])
        g.close()


# ------------------------------------------------------------------ the tile family's focal box
@pytest.mark.parametrize("mode", [0, 1])
def test_focal_box_of_every_tile_holds_the_focal_points_of_its_pixels(rt, mode):
    """Full 8x8 tiles take their focal box from the four corner pixels' focal points plus a curvature term
    (csrc/rt_trace.hpp focal_bounds, rt_tracer.hip tile_corner_bound): the box must hold the focal point of every pixel
    of the tile exactly as the rays use it (rt_dbg_focal_boxes returns both), it must not be much wider than their
    range, partial tiles must still take the range of their in-image lanes, and -- the test's teeth -- with the
    curvature term scaled to 0 some pixel's focal point must lie outside its box."""
    rng = np.random.default_rng(11 + mode)
    cases = [((24, 24), 0, 0, (0.0, 0.0), 90.0, 3.0), ((1920, 1080), 0, 0, (0.0, 0.0), 70.0, 3.0),
             ((333, 77), 0, 0, (0.4, -2.0), 149.0, 0.25), ((64, 100), 200, 56, (3.0, 1.0), 5.0, 1e4),
             ((40, 40), 0, 0, (-1.3, 0.2), 120.0, 1e-3)]
    for _ in range(25):
        W, H = int(rng.integers(8, 200)), int(rng.integers(8, 120))
        cases.append(((W, H), 0, 0, (float(rng.uniform(-3.2, 3.2)), float(rng.uniform(-3.2, 3.2))), float(rng.uniform(5, 150)),
                      float(10.0 ** rng.uniform(-3, 4))))
    broke_without_curvature = n_corner = 0
    for (W, H), full_h, row0, angles, fov, focal in cases:
        g = rt.RayTracer((W, H), (0, 0, 0), angles, fov, focal, focal * 0.01, seed=1, math_mode=mode, full_height=full_h, row_begin=row0)
        boxes, F = g.DebugFocalBoxes()
        boxes0, _ = g.DebugFocalBoxes(curv_scale=0.0)
        g.close()
        assert np.isfinite(F).all()
        for ty in range((H + 7) // 8):
            for tx in range((W + 7) // 8):
                f = F[ty * 8:ty * 8 + 8, tx * 8:tx * 8 + 8].reshape(-1, 3)
                b = boxes[ty, tx]
                a = 14.0 * np.tan(np.radians(fov) / 2.0) / (full_h or H)     # the tile's sides in image-plane units (both)
                full = f.shape[0] == 64 and a <= 0.099            # corner path: full tiles whose curvature term is < 10 % of a side
                if f.shape[0] < 64 or a < 0.099 or a > 0.101:
                    assert b[7] == 1.0 and b[6] == (1.0 if full else 0.0), (W, H, tx, ty, b)
                else:
                    full = b[6] == 1.0
                lo, hi = f.min(axis=0), f.max(axis=0)
                assert (b[0:3] <= lo).all() and (hi <= b[3:6]).all(), (W, H, fov, focal, tx, ty, b, lo, hi)
                if full:
                    # not much wider than the range: the curvature term is second order in the tile's angular size
                    assert ((lo - b[0:3]) <= focal * (1.1 * a * a + 1e-4)).all() and ((b[3:6] - hi) <= focal * (1.1 * a * a + 1e-4)).all()
                    b0 = boxes0[ty, tx]
                    n_corner += 1
                    broke_without_curvature += int(not ((b0[0:3] <= lo).all() and (hi <= b0[3:6]).all()))
                else:
                    assert np.array_equal(b[0:3], lo) and np.array_equal(b[3:6], hi)
    assert n_corner > 1000 and broke_without_curvature >= 10, (n_corner, broke_without_curvature)


# ------------------------------------------------------------------ classification at its decision boundaries
def test_adversarial_boundary_campaign_default_kernel_equals_plain_full_scan(rt):
    """tools/stress_boundaries.py: 300 scenes built by tests/adversarial.py -- triangles with an edge or a vertex on a
    ray of a tile's family (+-64 ulp), planes that contain a ray up to 1e-7..1e-3 rad, det within a factor of 30 of
    the culling epsilon, vertices at the focal points of tile corners, scales 1e-3..1e4, apertures 0..100x the scene
    -- default kernel vs the plain reference-order full scan, bit for bit.  (20 000 configurations were run once on
    hardware)"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_boundaries.py"), "300", "777", "0.04"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "300 configurations, 0 mismatches" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_rebalanced_partition_gives_the_same_frame(rt, orc):
    """rt_tracer_rebalance / rt_tracer_set_band + rt_tracer_join_group_bands: an uneven, cost-balanced partition (rows in
    multiples of 8) leaves the frame bit-identical -- the RNG streams are keyed by the global pixel index."""
    from raytracertest_amd import scenes
    rng = np.random.default_rng(3)
    tri = np.concatenate([rng.uniform(-2, 2, (600, 1, 2)), rng.uniform(-6, -2, (600, 1, 1))], axis=2) + rng.uniform(-0.3, 0.3, (600, 3, 3))
    tri[:, :, 1] = tri[:, :, 1] * 0.3 + 1.0               # everything in the upper part of the picture: uneven bands
    scn = scenes._tri_rows(tri.astype(np.float32))
    W, H = 96, 160
    g, o = pair(rt, orc, W, H, scn, devices=[0] * 4)
    g.TraceEnqueue(2, 2); g.TraceEnqueue(2, 2); g.Sync()
    before = [b["rows"] for b in g.Bands()]
    g.Rebalance()
    after = g.Bands()
    assert sum(b["rows"] for b in after) == H and all(b["rows"] % 8 == 0 for b in after[:-1]) and before == [40] * 4
    g.Trace(2, 3, 0); assert g.Wait()                   # (Rebalance re-created the RNG states, like Resize)
    o.trace(2, 3)
    assert_frame_equal(g, o)
    g.close()
    assert rt.balance_rows([0, 270, 540, 810, 1080], [1.0, 3.0, 3.0, 1.0]) == [0, 360, 544, 720, 1080]
    # the multi-process form of the same move on one rank: another band of the frame, explicit partition
    b = rt.RayTracer((W, 40), (0, 0, 0), (0.0, 0.0), 70.0, 3.0, 0.05, seed=3, full_height=H, row_begin=0)
    assert b.UploadScene(scn)
    b.SetBand(0, H)
    b.JoinGroup(1, 0, None, row_begin=[0, H])
    b.TraceEnqueue(2, 3); b.Sync()
    assert np.array_equal(b.Frame(), o.image)
    b.close()


# ------------------------------------------------------------------ tiles whose one candidate is certainly hit
@pytest.mark.parametrize("mode", [0, 1])
def test_sure_hit_tiles_skip_their_tests_and_change_nothing(rt, orc, mode):
    """Small-scene kernels: a tile whose candidate list is ONE triangle that every ray of its family certainly hits runs no
    intersection arithmetic (the winner is known; flat shading needs no t, u, v).  Same bits as with RT_FLAG_NO_SURE_HIT, as
    the oracle, through separate launches, fused iterations and stored tile lists; off by itself with spheres, smooth
    normals and the nearest-hit rule; and it really triggers (instrumented launch)."""
    from raytracertest_amd import scenes
    W, H = 160, 96
    g, o = pair(rt, orc, W, H, scene("cornell"), mode=mode)
    h, _ = pair(rt, orc, W, H, scene("cornell"), mode=mode, no_sure_hit=True)
    for tr in (g, h):
        tr.Trace(5, 3, 2); assert tr.Wait()               # fused groups, stored + loaded lists
    o.trace(5, 3)
    assert_frame_equal(g, o)
    assert_frame_equal(h, o)
    st, st_off = g.TraceStats(4), h.TraceStats(4)
    tiles = ((W + 7) // 8) * ((H + 7) // 8)
    assert st["pretest_skips"] >= 10 and st_off["pretest_skips"] == 0, (st, tiles)      # (1080p C3: 52 % of the tiles)
    count, winner, sure = g.DebugTileLists()             # the verdicts travel with the stored lists
    assert int(sure[:, :(W + 7) // 8].sum()) == st["tiles_by_list"]["sure"] and (count[sure] >= 1).all() and (winner[sure] < 32).all()
    g.close(); h.close()
    for kw in (dict(nearest_hit=True), dict(spheres=True)):
        sph = np.array([[0.0, 0.0, -2.0, 0.4]], np.float32) if kw.pop("spheres", False) else None
        g = rt.RayTracer((W, H), (0, 0, 0), (0.0, 0.0), 70.0, 3.0, 0.05, seed=3, math_mode=mode, **kw)
        o = orc.OracleTracer(W, H, (0.0, 0.0), 70.0, 3.0, 0.05, seed=3, contract=1 - mode, nthreads=8, hit_mode=int(kw.get("nearest_hit", False)))
        assert g.UploadScene(scene("cornell")) and o.upload_scene(scene("cornell"))
        if sph is not None:
            g.UploadSpheres(sph); o.upload_spheres(sph)
        g.Trace(2, 3, 0); assert g.Wait()
        o.trace(2, 3)
        assert_frame_equal(g, o)
        assert g.TraceStats(2)["pretest_skips"] == 0
        g.close()

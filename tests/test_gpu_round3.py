"""Round-3 GPU parity tests: the launch shape bench.py's headline times, bit-checked at full size against the committed
golden frame and -- the RNG stream continues from step to step -- against the oracle on sampled row bands."""
import numpy as np
import pytest

from test_golden import FRAMES, _check, _spec_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rt():
    import raytracertest_amd as R
    from raytracertest_amd import api
    assert R.device_count() >= 1, "no HIP device: the GPU tests need the real extension"
    return api


def test_c3_headline_launch_shape_at_full_size(rt, orc):
    """What `python bench.py` times: 1920x1080, list reuse restricted to one Trace, TraceEnqueue(1, 16) = ONE launch split
    into two half-frame kernels on two streams, accumulators taken as zero (TRACE_ZERO_ACC), BGRA8 emitted by the same
    launch.  Step 1 == the golden frame C3_full/fma (CRCs of render / counts / RNG states / image + centre crop);
    steps 2..4 continue the pixels' RNG streams (RayTracerImpl.cu:94-103: states are re-created by Resize only), checked
    against the oracle on the first rows, the rows across the split (row 544) and the last rows."""
    spec = FRAMES["C3_full/fma"]["spec"]
    assert (spec["W"], spec["H"], spec["it"], spec["spp"]) == (1920, 1080, 1, 16)
    tris, _ = _spec_scene(spec)
    g = rt.RayTracer((spec["W"], spec["H"]), (0, 0, 0), (0.0, 0.0), spec["fov"], spec["focal"], spec["aperture"], seed=1)
    assert g.UploadScene(tris)
    g.SetListReuse(False)
    g.TraceEnqueue(1, 16); g.Sync()
    render, counts, rng, image = g.RenderBuffer(), g.SampleCounts(), g.RngStates(), g.Image()
    _check("C3_full/fma", render, counts, rng, image)
    steps = 4
    for _ in range(steps - 1):
        g.TraceEnqueue(1, 16)
    g.Sync()
    render, counts, rng, image = g.RenderBuffer(), g.SampleCounts(), g.RngStates(), g.Image()
    assert (counts == 16).all() and not render[..., 3].any()
    for row0, rows in ((0, 8), (536, 16), (1072, 8)):
        o = orc.OracleTracer(spec["W"], spec["H"], (0.0, 0.0), spec["fov"], spec["focal"], spec["aperture"], seed=1, row0=row0,
                             rows=rows, contract=1, nthreads=8)
        o.upload_scene(tris)
        for _ in range(steps):
            o.trace(1, 16)
        sl = slice(row0, row0 + rows)
        assert np.array_equal(render[sl].view(np.uint32), o.render.view(np.uint32)), "rows %d..: render after %d steps" % (row0, steps)
        assert np.array_equal(rng[sl], o.rng) and np.array_equal(image[sl], o.image) and np.array_equal(counts[sl], o.counts)
    g.close()


def test_bench_parity_leg_reads_the_same_fixture():
    """bench.py's parity_check leg (one untimed headline step on a fresh tracer against tests/golden/frames.json)."""
    import bench
    import raytracertest_amd as R
    from raytracertest_amd import scenes
    cfg = dict(scenes.CONFIGS["C3"])
    tris, _ = scenes.scene_for("C3")
    g = R.RayTracer((cfg["width"], cfg["height"]), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=cfg["seed"])
    assert g.UploadScene(tris)
    g.SetListReuse(False)
    g.TraceEnqueue(cfg["iterations"], cfg["samples"]); g.Sync()
    rec = bench.parity_check("C3", g)
    assert rec["fixture"] == "C3_full/fma" and rec["ok"] is True, rec
    g.TraceEnqueue(cfg["iterations"], cfg["samples"]); g.Sync()       # a second step is another frame: the leg must notice
    assert bench.parity_check("C3", g)["ok"] is False
    g.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_certain_winner_tiles_in_a_200_triangle_scene(rt, orc, mode):
    """The certain-winner verdict spans classification steps (tile_lists_kernel): the Cornell box plus 168 small triangles in
    front of its walls is a small scene of 200 triangles (four steps of 64) -- its wall tiles still have a certain winner,
    the image equals the oracle's and the one with RT_FLAG_NO_SURE_HIT, through Trace (fused groups), split TraceEnqueue
    launches and kept / rebuilt lists."""
    from raytracertest_amd import scenes
    rng = np.random.default_rng(5)
    centre = np.stack([rng.uniform(-0.9, 0.9, 168), rng.uniform(-0.9, 0.9, 168), rng.uniform(-2.6, -1.2, 168)], axis=1)
    small = centre[:, None, :] + rng.uniform(-0.04, 0.04, (168, 3, 3))
    scn = np.concatenate([scenes._tri_rows(small[:100]), scenes.cornell32(), scenes._tri_rows(small[100:])])    # the walls sit mid-list
    assert scn.shape[0] == 600
    W, H = 160, 136
    kw = dict(seed=9, math_mode=mode)
    g = rt.RayTracer((W, H), (0, 0, 0), (0.0, 0.0), 70.0, 3.0, 0.02, **kw)
    h = rt.RayTracer((W, H), (0, 0, 0), (0.0, 0.0), 70.0, 3.0, 0.02, no_sure_hit=True, **kw)
    o = orc.OracleTracer(W, H, (0.0, 0.0), 70.0, 3.0, 0.02, seed=9, contract=1 - mode, nthreads=8)
    assert g.UploadScene(scn) and h.UploadScene(scn) and o.upload_scene(scn)
    for tr in (g, h):
        tr.Trace(4, 3, 2); assert tr.Wait()
    o.trace(4, 3)
    for tr in (g, h):
        assert np.array_equal(tr.RenderBuffer().view(np.uint32), o.render.view(np.uint32)) and np.array_equal(tr.RngStates(), o.rng)
        assert np.array_equal(tr.Image(), o.image)
    g.SetListReuse(False)
    g.TraceEnqueue(2, 5); g.Sync()                       # split launches (136 rows), lists rebuilt by the first of them
    o.trace(2, 5)
    assert np.array_equal(g.RenderBuffer().view(np.uint32), o.render.view(np.uint32)) and np.array_equal(g.Image(), o.image)
    st, st_off = g.TraceStats(4), h.TraceStats(4)
    assert st["tiles_by_list"]["sure"] > 20 and st_off["tiles_by_list"]["sure"] == 0, (st, st_off)
    count, winner, sure = g.DebugTileLists()
    assert int(sure[:, :W // 8].sum()) == st["tiles_by_list"]["sure"]
    assert ((winner[sure] >= 100) & (winner[sure] < 132)).all()       # the winners are walls / box faces of the Cornell part
    g.close(); h.close()


# ------------------------------------------------------------------ the frame sharded over devices: transports
def _multi_pair(rt, orc, W, H, devices, transport):
    from raytracertest_amd import scenes
    g = rt.RayTracer((W, H), (0, 0, 0), (0.0, 0.0), 70.0, 3.0, 0.05, seed=3, devices=devices, transport=transport)
    o = orc.OracleTracer(W, H, (0.0, 0.0), 70.0, 3.0, 0.05, seed=3, contract=1, nthreads=8)
    scn = scenes.cornell32()
    assert g.UploadScene(scn) and o.upload_scene(scn)
    return g, o


@pytest.mark.parametrize("transport", ["rccl", "peer"])
def test_transports_with_every_band_on_one_device(rt, orc, transport, monkeypatch):
    """Both transports of rt_tracer_create_multi with three bands on device 0 == oracle (buffers and gathered frame), the group
    says what it is made of, a gather without tracing runs; a denied peer mapping (RT_MI355X_PEER_DENY=1) falls back with the
    reason on record.  (One device: nothing travels -- the multi-device forms are the tests below, skipped here.)"""
    g, o = _multi_pair(rt, orc, 96, 50, [0, 0, 0], transport)
    info = g.GroupInfo()
    assert info["transport"] == ("peer" if transport == "peer" else "local") and info["bands"] == 3 and info["band_ranks"] == [0, 0, 0], info
    g.Trace(2, 3, 0); assert g.Wait()
    o.trace(2, 3)
    assert np.array_equal(g.RenderBuffer().view(np.uint32), o.render.view(np.uint32)) and np.array_equal(g.Frame(), o.image)
    for _ in range(3):
        g.TraceEnqueue(1, 4); o.trace(1, 4)
    g.GatherOnly(); g.Sync()
    assert np.array_equal(g.Frame(), o.image)
    g.close()
    if transport == "peer":
        monkeypatch.setenv("RT_MI355X_PEER_DENY", "1")
        g, o = _multi_pair(rt, orc, 64, 40, [0, 0], "peer")
        info = g.GroupInfo()
        assert info["transport"] == "local" and "falling back to the RCCL gather" in info["note"], info
        g.Trace(1, 2, 0); assert g.Wait()
        o.trace(1, 2)
        assert np.array_equal(g.Frame(), o.image)
        g.close()


@pytest.mark.parametrize("transport", ["rccl", "peer"])
def test_transports_between_two_devices(rt, orc, transport):
    """the gather between DEVICES.  Four bands on two GPUs in one process (rt_tracer_create_multi): RCCL
    (ncclCommInitAll, grouped ncclSend / ncclRecv) and peer stores into the root's frame, each == the oracle's whole frame."""
    if rt.device_count() < 2:
        pytest.skip("needs at least two GPUs (the pool's boxes have one): run on a multi-GPU node")
    g, o = _multi_pair(rt, orc, 160, 150, [0, 1, 0, 1], transport)
    info = g.GroupInfo()
    assert info["ranks"] == 2 and info["transport"] in (transport, "rccl"), info
    if info["transport"] == "rccl":
        assert [c["ranks_in_communicator"] for c in info["communicators"]] == [2, 2], info
    g.Trace(3, 2, 0); assert g.Wait()
    o.trace(3, 2)
    assert np.array_equal(g.RenderBuffer().view(np.uint32), o.render.view(np.uint32)) and np.array_equal(g.Frame(), o.image)
    for _ in range(4):                                  # both frame buffers, sends double-buffered
        g.TraceEnqueue(1, 3); o.trace(1, 3)
    g.Sync()
    assert np.array_equal(g.Frame(), o.image)
    g.GatherOnly(); g.Sync()
    assert np.array_equal(g.Frame(), o.image)
    g.close()


def test_one_process_per_gpu_through_the_native_exchange(rt):
    """rt_tracer_join_group between processes (tests/dist_native.py: RowBandJob + NativeExchange, frame == oracle)."""
    if rt.device_count() < 2:
        pytest.skip("needs at least two GPUs (the pool's boxes have one): run on a multi-GPU node")
    import os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(root, "tests", "dist_native.py")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "dist_native ok: world=2" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


def test_group_info_and_gather_only_through_rccl_on_one_rank(rt, orc, monkeypatch):
    """RT_MI355X_GATHER_SELF=1 routes root-local tiles through the RCCL path on a one-rank communicator: the group reports
    what RCCL itself says it spans (version, ncclCommCount, device), a gather without tracing re-delivers the last frame, and
    a split launch's cost (rt_tracer_launch_time: to the end of the later half) is at least its upper half's kernel time."""
    monkeypatch.setenv("RT_MI355X_GATHER_SELF", "1")
    g, o = _multi_pair(rt, orc, 96, 300, [0, 0], "rccl")
    info = g.GroupInfo()
    assert info["transport"] == "rccl" and info["rccl_version"] > 0, info
    assert info["communicators"] == [{"rank": 0, "ranks_in_communicator": 1, "device": 0}], info
    for _ in range(3):
        g.TraceEnqueue(1, 3); o.trace(1, 3)
    g.Sync()
    assert np.array_equal(g.Frame(), o.image)
    g.GatherTime()
    g.GatherOnly(); g.GatherOnly(); g.Sync()
    ms, n = g.GatherTime()
    assert n == 2 and ms > 0.0 and np.array_equal(g.Frame(), o.image)
    g.close()
    monkeypatch.delenv("RT_MI355X_GATHER_SELF")
    h = rt.RayTracer((96, 300), (0, 0, 0), (0.0, 0.0), 70.0, 3.0, 0.05, seed=3)
    from raytracertest_amd import scenes
    h.UploadScene(scenes.cornell32())
    h.KernelTime()
    for _ in range(20):
        h.TraceEnqueue(1, 8)                      # 300 rows: split launches (every 16th launch carries timing events: two samples)
    h.Sync()
    span_ms, n1 = h.LaunchTime(reset=False)
    k_ms, n2 = h.KernelTime()
    assert n1 == n2 >= 2 and span_ms >= k_ms > 0.0
    h.close()


def test_small_scene_path_with_an_explicit_long_list(rt, orc):
    """rt_options.bin_list = 320 makes a 300-triangle scene a small scene (one list per tile, built ahead of the launch): more
    triangles than the two-level builder's region list holds, so the one-level builder serves it -- five classification steps
    per tile, the certain-winner verdict carried across them.  == oracle, through Trace and split TraceEnqueue launches."""
    from raytracertest_amd import scenes
    scn = np.concatenate([scenes.random_triangles(268, 99), scenes.cornell32()])
    W, H = 128, 136
    g = rt.RayTracer((W, H), (0, 0, 0), (0.0, 0.0), 70.0, 3.0, 0.03, seed=2, bin_list=320)
    o = orc.OracleTracer(W, H, (0.0, 0.0), 70.0, 3.0, 0.03, seed=2, contract=1, nthreads=8)
    assert g.UploadScene(scn) and o.upload_scene(scn)
    g.Trace(3, 2, 0); assert g.Wait()
    o.trace(3, 2)
    assert np.array_equal(g.RenderBuffer().view(np.uint32), o.render.view(np.uint32)) and np.array_equal(g.Image(), o.image)
    g.TraceEnqueue(2, 3); g.Sync()
    o.trace(2, 3)
    assert np.array_equal(g.RenderBuffer().view(np.uint32), o.render.view(np.uint32)) and np.array_equal(g.RngStates(), o.rng)
    count, winner, sure = g.DebugTileLists()
    assert 0 < count.max() <= 300 and (winner[sure] < 300).all()
    g.close()


def test_dense_scene_macro_lists_are_kept_between_launches_and_rebuilt_on_change(rt, orc):
    """Dense scenes: the macro-tile lists are re-binned when camera, scene or frame changed (or at the first launch of a Trace
    with list reuse off) and kept otherwise.  Accumulating split launches that reuse them, a camera rotation in between, a
    second scene and both reuse modes all end bit-identical to the oracle."""
    from raytracertest_amd import scenes
    scn = scenes.random_triangles(4200, 5)
    W, H = 96, 136
    g = rt.RayTracer((W, H), (0, 0, 0), (0.0, 0.0), 70.0, 6.0, 0.05, seed=4)
    o = orc.OracleTracer(W, H, (0.0, 0.0), 70.0, 6.0, 0.05, seed=4, contract=1, nthreads=8)
    assert g.UploadScene(scn) and o.upload_scene(scn)

    def same():
        g.Sync()
        assert np.array_equal(g.RenderBuffer().view(np.uint32), o.render.view(np.uint32)) and np.array_equal(g.RngStates(), o.rng)

    g.Launch(2, clear_first=True); g.Launch(1); g.Launch(2, emit_image=True)      # bins once, reuses twice (split launches)
    o.trace(0, 0); o.launch(2); o.launch(1); o.launch(2)
    same()
    g.RotateCamera((0.03, -0.05)); o.rotate_camera((0.03, -0.05))
    g.Launch(1); g.Launch(1)                                                     # new camera: re-binned, then reused
    o.launch(1); o.launch(1)
    same()
    g.SetListReuse(False)
    g.TraceEnqueue(2, 1); o.trace(2, 1)                                          # first launch of a Trace re-bins, the second reuses
    same()
    scn2 = scenes.random_triangles(4300, 6)
    assert g.UploadScene(scn2) and o.upload_scene(scn2)
    g.Trace(2, 1, 0); assert g.Wait(); o.trace(2, 1)                             # unsplit launches of the render thread
    same()
    assert np.array_equal(g.Image(), o.image)
    g.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_certain_winner_table_follows_sample_count_scene_and_launch_kind(rt, orc, mode):
    """Tiles with a certain winner read what their samples add up to -- and, on cleared accumulators, their BGRA8 word -- from
    a per-triangle table built for the launch's sample count (rtk::sure_table_kernel).  Back-to-back launches with changing
    sample counts (the table is rewritten while earlier launches may still be in flight), clearing / accumulating / emitting
    launches, fused iterations, a second scene and the table switched off all end bit-identical to the oracle."""
    import ctypes as C
    from oracle import oracle_py
    from raytracertest_amd import scenes
    W, H = 192, 136                                                    # split launches (>= 128 rows)
    g = rt.RayTracer((W, H), (0, 0, 0), (0.0, 0.0), 70.0, 3.0, 0.05, seed=6, math_mode=mode)
    o = orc.OracleTracer(W, H, (0.0, 0.0), 70.0, 3.0, 0.05, seed=6, contract=1 - mode, nthreads=8)
    scn = scenes.cornell32()
    assert g.UploadScene(scn) and o.upload_scene(scn)

    def same(image=False):
        g.Sync()
        assert np.array_equal(g.SampleCounts(), o.counts)
        assert np.array_equal(g.RngStates(), o.rng), "RNG states"
        assert np.array_equal(g.RenderBuffer().view(np.uint32), o.render.view(np.uint32)), "render buffer"
        if image:
            oracle_py.lib().orc_convert(C.byref(o._frame))            # Kernels.cuh:149-169 on the oracle's accumulators
            assert np.array_equal(g.Image(), o.image), "BGRA8 image"

    g.Trace(1, 16, 0); assert g.Wait(); o.trace(1, 16)                 # cleared accumulators + emit: the table's BGRA8 word
    words = g.DebugTileListWords()[:, :, 0]
    assert ((words >> 31) != 0).mean() > 0.3, "the frame has certain-winner tiles"
    same(image=True)
    seq = [1, 4, 16, 3, 16, 0, 5, 5, 2]
    g.Launch(seq[0], clear_first=True); o.trace(0, 0); o.launch(seq[0])
    for s in seq[1:]:
        g.Launch(s); o.launch(s)                                       # accumulating, no host sync in between
    same()
    g.Launch(7, emit_image=True); o.launch(7)                          # emit on accumulated buffers: converted per pixel
    same(image=True)
    g.Launch(6, clear_first=True, emit_image=True); o.trace(0, 0); o.launch(6)
    same(image=True)
    g.Trace(5, 3, 0); assert g.Wait(); o.trace(5, 3)                   # fused iterations of a Trace nobody observes
    same(image=True)
    big = np.array([[-50, -50, -4, 0], [50, -50, -4, 0], [0, 90, -4, 0], [-1, -1, -2, 0], [1, -1, -2, 0], [0, 1, -2, 0]], np.float32)
    assert g.UploadScene(big) and o.upload_scene(big)                  # another scene, same sample count: the table follows
    g.Trace(2, 3, 0); assert g.Wait(); o.trace(2, 3)
    same(image=True)
    g.close()


def test_certain_winner_table_can_be_switched_off(rt, orc, monkeypatch):
    """RT_MI355X_NO_SURE_TABLE=1 (the additions done per pixel): the same bits."""
    from raytracertest_amd import scenes
    import subprocess, sys, os
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import raytracertest_amd as R
from raytracertest_amd import scenes
g = R.RayTracer((192, 136), (0, 0, 0), (0.0, 0.0), 70.0, 3.0, 0.05, seed=6)
g.UploadScene(scenes.cornell32()); g.Trace(2, 9, 0); g.Wait()
import zlib
print(zlib.crc32(g.RenderBuffer().tobytes()), zlib.crc32(g.RngStates().tobytes()), zlib.crc32(g.Image().tobytes()))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for off in ("0", "1"):
        env = dict(os.environ, RT_MI355X_NO_SURE_TABLE=off)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1], outs
    o = orc.OracleTracer(192, 136, (0.0, 0.0), 70.0, 3.0, 0.05, seed=6, contract=1, nthreads=8)
    o.upload_scene(scenes.cornell32()); o.trace(2, 9)
    import zlib
    assert outs[0] == "%d %d %d" % (zlib.crc32(o.render.tobytes()), zlib.crc32(o.rng.tobytes()), zlib.crc32(o.image.tobytes()))


def test_long_lists_on_a_large_frame_use_the_short_ring(rt, orc):
    """A list buffer above 128 MiB (960-record lists on a 2048x1200 frame) gets a ring of 4 slots with a free event every
    2nd build instead of 8 : 4.  Twelve Traces with the lists rebuilt by each (the ring wraps three times) end bit-identical
    to the oracle on sampled row bands."""
    from raytracertest_amd import scenes
    W, H = 2048, 1200
    scn = scenes.random_triangles(300, 11)
    g = rt.RayTracer((W, H), (0, 0, 0), (0.0, 0.0), 70.0, 3.0, 0.03, seed=9, bin_list=960)
    assert g.UploadScene(scn)
    g.SetListReuse(False)
    for _ in range(12):
        g.TraceEnqueue(1, 1)
    g.Sync()
    render, rng_states = g.RenderBuffer(), g.RngStates()
    for row0 in (0, 592, 1184):
        o = orc.OracleTracer(W, H, (0.0, 0.0), 70.0, 3.0, 0.03, seed=9, contract=1, nthreads=8, row0=row0, rows=16)
        assert o.upload_scene(scn)
        for _ in range(12):
            o.trace(1, 1)
        assert np.array_equal(render[row0:row0 + 16].view(np.uint32), o.render.view(np.uint32)), "rows %d.." % row0
        assert np.array_equal(rng_states[row0:row0 + 16], o.rng), "RNG states, rows %d.." % row0
    g.close()


def test_split_halves_switch_between_rows_and_block_row_interleave(rt, orc):
    """Split small-scene launches use the band's upper / lower rows or its even / odd block rows, chosen from the list builder's
    per-half counts of ray-generating tiles (every 32nd build) -- a switch moves pixels between the two streams and joins them
    first.  A camera that swings between a balanced and a lopsided view while launches stay in flight, and both pinned modes,
    end bit-identical to the oracle on sampled rows."""
    import os, subprocess, sys, zlib
    from raytracertest_amd import scenes
    W, H = 512, 320
    code = r"""
import sys, zlib, numpy as np
sys.path.insert(0, %r)
import raytracertest_amd as R
from raytracertest_amd import scenes
g = R.RayTracer((512, 320), (0, 0, 0), (0.0, 0.0), 70.0, 3.0, 0.05, seed=5)
g.UploadScene(scenes.cornell32()); g.SetListReuse(False)
for k in range(6):
    for _ in range(40): g.TraceEnqueue(1, 2)            # no host sync in between: the switch happens under launches in flight
    g.RotateCamera((0.45 if k %% 2 == 0 else -0.45, 0.0))
g.Sync()
print(zlib.crc32(g.RenderBuffer().tobytes()), zlib.crc32(g.RngStates().tobytes()), zlib.crc32(g.Image().tobytes()))
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for mode in ("", "0", "1"):
        env = dict(os.environ)
        env.pop("RT_MI355X_ROW_INTERLEAVE", None)
        if mode: env["RT_MI355X_ROW_INTERLEAVE"] = mode
        env["RT_MI355X_LOG"] = "1"
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[mode] = r.stdout.strip().splitlines()[-1]
        if not mode:                                            # the free-running tracer did switch, both ways
            assert "halves by even / odd block rows" in r.stderr and "halves by rows" in r.stderr, r.stderr[-1500:]
    assert outs[""] == outs["0"] == outs["1"], outs
    o = orc.OracleTracer(W, H, (0.0, 0.0), 70.0, 3.0, 0.05, seed=5, contract=1, nthreads=8)
    o.upload_scene(scenes.cornell32())
    for k in range(6):
        for _ in range(40): o.trace(1, 2)
        o.rotate_camera((0.45 if k % 2 == 0 else -0.45, 0.0))
    assert outs[""] == "%d %d %d" % (zlib.crc32(o.render.tobytes()), zlib.crc32(o.rng.tobytes()), zlib.crc32(o.image.tobytes()))


def test_soak_of_the_split_launch_machinery():
    """tools/soak_split.py, short: random enqueues, launches, Traces, camera swings, list reuse toggles, scene swaps and Resizes
    at split-launch sizes with launches in flight, every phase bit-identical to the oracle on sampled rows."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_split.py"), "40", "21"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "bit-identical" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]

"""the planned (v0, e0, e1) + packed-normal scene layout with its own
KATs (the reference's NormalPackingTest is inconsistent), and the nearest-hit extension."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pack_unpack_kats_python_and_c_abi():
    from raytracertest_amd import api, meshes
    L = api.load_library()
    f = np.float32
    # the reference's own test vector (NormalPackingTest.cpp:30-33): (0,0,1) must round-trip
    kats = {(0.0, 0.0, 1.0): (127, 127, 254), (1.0, 0.0, 0.0): (254, 127, 127), (0.0, -1.0, 0.0): (127, 0, 127),
            (-1.0, -1.0, -1.0): (0, 0, 0), (1.0, 1.0, 1.0): (254, 254, 254)}
    for n, bytes_ in kats.items():
        expect = f(bytes_[0] / 256.0 + bytes_[1] / 65536.0 + bytes_[2] / 16777216.0)
        p = meshes.pack_normal(n)
        arr = np.array(n, f)
        pc = f(L.rt_pack_normal(arr.ctypes.data_as(C.POINTER(C.c_float))))
        assert p == expect == pc, n
        out = np.zeros(3, f)
        L.rt_unpack_normal(pc, out.ctypes.data_as(C.POINTER(C.c_float)))
        assert np.array_equal(out, arr) and np.array_equal(meshes.unpack_normal(p), arr), n
    # every representable normal (components k/127 - 1) round-trips exactly; others within 1/127
    k = np.arange(0, 255, dtype=f)
    grid = np.stack(np.meshgrid(k[::7], k[::11], k[::13], indexing="ij"), -1).reshape(-1, 3) / f(127.0) - f(1.0)
    assert np.array_equal(meshes.unpack_normal(meshes.pack_normal(grid)), grid.astype(f))
    rnd = np.random.default_rng(3).uniform(-1, 1, (2000, 3)).astype(f)
    assert np.abs(meshes.unpack_normal(meshes.pack_normal(rnd)) - rnd).max() <= 0.5 / 127 + 1e-6
    # what the reference's unpack multipliers give for its own test vector (why its test cannot pass)
    p = meshes.pack_normal((0.0, 0.0, 1.0))
    wrong = [np.floor(((p * m) - np.floor(p * m)) * f(256)) / f(127) - f(1) for m in (f(1), f(65536), f(16777216))]
    assert [float(x) for x in wrong] == [0.0, 1.0, -1.0]


def test_cpp_normal_packing_header(tmp_path):
    src = tmp_path / "np.cpp"
    src.write_text('#include <cstdio>\n#include "Common/NormalPacking.h"\nint main(){ const math::vec3 n(0.0f,0.0f,1.0f);'
                   ' const float p = rt::pack(n); const math::vec3 u = rt::unpack(p);'
                   ' std::printf("%a %g %g %g\\n", p, u.x, u.y, u.z); return (u == n) ? 0 : 1; }\n')
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), str(src), "-o",
                    str(tmp_path / "np")], check=True)
    out = subprocess.run([str(tmp_path / "np")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout     # EXPECT_EQ( normal, unpacked ) of the reference's test holds


def test_edge_format_conversion():
    from raytracertest_amd import meshes, scenes
    v = scenes.cornell32()
    e = meshes.to_edge_format(v, normals=np.tile(np.array([0.0, 0.0, 1.0], np.float32), (v.shape[0], 1)))
    t, te = v.reshape(-1, 3, 4), e.reshape(-1, 3, 4)
    assert np.array_equal(te[:, 0, :3], t[:, 0, :3])
    assert np.array_equal(te[:, 1, :3], t[:, 1, :3] - t[:, 0, :3]) and np.array_equal(te[:, 2, :3], t[:, 2, :3] - t[:, 0, :3])
    assert np.array_equal(meshes.vertex_normals(e), np.tile(np.array([0.0, 0.0, 1.0], np.float32), (v.shape[0], 1)))


@pytest.mark.gpu
def test_edge_format_upload_equals_vertex_upload():
    import raytracertest_amd as R
    from raytracertest_amd import meshes, scenes
    scn = scenes.random_triangles(400, 21)
    res = []
    for edges in (False, True):
        g = R.RayTracer((64, 36), (0, 0, 0), (0, 0), 70.0, 3.0, 0.05, seed=2)
        assert (g.UploadSceneEdges(meshes.to_edge_format(scn, normals=np.zeros((scn.shape[0], 3)))) if edges
                else g.UploadScene(scn))
        g.Trace(2, 3, 0)
        assert g.Wait()
        res.append((g.RenderBuffer(), g.Image()))
    assert np.array_equal(res[0][0].view(np.uint32), res[1][0].view(np.uint32)) and np.array_equal(res[0][1], res[1][1])
    assert g.UploadSceneEdges(np.zeros((4, 4), np.float32)) is False


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("no_binning", [False, True])
def test_nearest_hit_mode_matches_oracle(orc, mode, no_binning):
    import raytracertest_amd as R
    from raytracertest_amd import scenes
    W, H = 72, 40
    sph = np.array([[0.3, -0.2, -2.2, 0.35]], np.float32)
    for scn in (scenes.cornell32(), scenes.random_triangles(500, 8)):
        g = R.RayTracer((W, H), (0, 0, 0), (0.1, 0.2), 70.0, 3.0, 0.05, seed=4, math_mode=mode, nearest_hit=True,
                        no_binning=no_binning)
        o = orc.OracleTracer(W, H, (0.1, 0.2), 70.0, 3.0, 0.05, seed=4, contract=1 - mode, nthreads=8, hit_mode=1)
        g.UploadScene(scn); o.upload_scene(scn)
        g.UploadSpheres(sph); o.upload_spheres(sph)
        g.Trace(2, 3, 0)
        assert g.Wait()
        o.trace(2, 3)
        assert np.array_equal(g.RenderBuffer().view(np.uint32), o.render.view(np.uint32))
        assert np.array_equal(g.Image(), o.image) and np.array_equal(g.RngStates(), o.rng)
    # and it really differs from the reference's farthest-hit rule on the Cornell box (boxes become visible)
    f = R.RayTracer((W, H), (0, 0, 0), (0.1, 0.2), 70.0, 3.0, 0.05, seed=4, math_mode=mode)
    f.UploadScene(scenes.cornell32())
    f.Trace(2, 3, 0); assert f.Wait()
    n = R.RayTracer((W, H), (0, 0, 0), (0.1, 0.2), 70.0, 3.0, 0.05, seed=4, math_mode=mode, nearest_hit=True)
    n.UploadScene(scenes.cornell32())
    n.Trace(2, 3, 0); assert n.Wait()
    assert (f.Image() != n.Image()).mean() > 0.05

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


# ---- smooth shading from the packed vertex normals (build-defined) --------------

def _random_edge_scene(n, seed):
    from raytracertest_amd import meshes, scenes
    scn = scenes.random_triangles(n, seed)
    nrm = np.random.default_rng(seed).normal(size=(scn.shape[0], 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    return meshes.to_edge_format(scn, normals=nrm)


def _fma32(a, b, c):
    """float32 fma with a single, exact rounding (round to nearest even), via rationals."""
    from fractions import Fraction
    x = Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c))
    g = np.float32(float(x))
    cands = [g, np.nextafter(g, np.float32(-np.inf)), np.nextafter(g, np.float32(np.inf))]
    cands.sort(key=lambda q: (abs(Fraction(float(q)) - x), int(q.view(np.uint32)) & 1))
    return cands[0]


def test_oracle_smooth_shading_kats(orc):
    """One triangle, a pinhole camera, known barycentrics: the colour is the bit-exact
    |normalize((w*n0 + u*n1) + v*n2)| of the unpacked normals, in the documented order."""
    from raytracertest_amd import meshes
    f = np.float32
    v = np.array([[-1, -1, -3, 0], [1, -1, -3, 0], [-1, 1, -3, 0]], f)          # det > 0 seen from the origin
    normals = np.array([[0.0, 0.0, 1.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0]], f)
    rows = meshes.to_edge_format(v, normals=normals)
    n0, n1, n2 = meshes.unpack_normal(rows[:, 3])
    assert np.array_equal(np.stack([n0, n1, n2]), normals)                       # representable exactly
    for (px, py) in ((-0.5, -0.5), (0.25, -0.75), (-0.9, 0.7), (-0.99, -0.99)):
        d = np.array([px, py, -3.0], f)
        ray = np.concatenate([np.zeros(3, f), d / np.sqrt((d * d).sum(dtype=f), dtype=f)]).astype(f)
        for contract in (0, 1):
            rgb = orc.radiance(ray, rows, layout=1, shade_mode=1, contract=contract)
            hit, t, u, vv = orc.hit_triangle(ray, v[0, :3], v[1, :3], v[2, :3], contract=contract)
            assert hit and abs(u - (px + 1) / 2) < 1e-5 and abs(vv - (py + 1) / 2) < 1e-5
            w = f(f(1.0) - u) - vv
            if contract:                      # fma(v, n2, fma(u, n1, w*n0)), one rounding per fma
                m = np.array([_fma32(vv, n2[c], _fma32(u, n1[c], f(w * n0[c]))) for c in range(3)], f)
            else:
                m = ((w * n0).astype(f) + (u * n1).astype(f)).astype(f) + (vv * n2).astype(f)
            flat = orc.radiance(ray, rows, layout=1, shade_mode=0, contract=contract)
            assert np.array_equal(flat, np.array([0.0, 0.0, 1.0], f))            # face normal = +z
            # normalize = v * (1/sqrt(dot)): check direction and unit length, then exact bits through the oracle's own helper
            assert np.allclose(rgb, np.abs(m) / np.linalg.norm(m), rtol=0, atol=2e-7)
            assert np.array_equal(rgb, np.abs(orc.normalize(m, contract=contract)))
    # all three vertex normals equal -> that normal's |.| everywhere on the triangle, whatever u, v
    rows_c = meshes.to_edge_format(v, normals=np.tile(np.array([0.0, 0.0, 1.0], f), (3, 1)))
    ray = np.array([0, 0, 0, 0, 0, -1], f)
    assert np.array_equal(orc.radiance(ray, rows_c, layout=1, shade_mode=1), np.array([0.0, 0.0, 1.0], f))
    # shade_mode 1 has no effect on the reference's vertex layout
    assert np.array_equal(orc.radiance(ray, v, layout=0, shade_mode=1), orc.radiance(ray, v, layout=0, shade_mode=0))


def test_oracle_edge_layout_equals_vertex_layout_when_flat(orc):
    from raytracertest_amd import meshes, scenes
    scn = scenes.random_triangles(300, 5)
    a = orc.OracleTracer(40, 24, (0, 0), 70.0, 3.0, 0.05, seed=3, nthreads=4)
    b = orc.OracleTracer(40, 24, (0, 0), 70.0, 3.0, 0.05, seed=3, nthreads=4)
    a.upload_scene(scn)
    b.upload_scene_edges(meshes.to_edge_format(scn))
    a.trace(1, 4); b.trace(1, 4)
    assert np.array_equal(a.render.view(np.uint32), b.render.view(np.uint32))


def test_oracle_pack_unpack_equals_python(orc):
    from raytracertest_amd import meshes
    rnd = np.random.default_rng(9).uniform(-1, 1, (500, 3)).astype(np.float32)
    for n in rnd:
        p = orc.pack_normal(n)
        assert p == meshes.pack_normal(n)
        assert np.array_equal(orc.unpack_normal(p), meshes.unpack_normal(p))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("nearest", [False, True])
def test_smooth_shading_matches_oracle(orc, mode, nearest):
    import raytracertest_amd as R
    from raytracertest_amd import meshes
    W, H = 80, 48
    sph = np.array([[0.9, 0.5, -3.0, 0.4]], np.float32)
    for rows, kw in ((meshes.uv_sphere(n_lat=8, n_lon=16), dict(no_binning=False)),
                     (_random_edge_scene(700, 13), dict(no_binning=False)),
                     (_random_edge_scene(90, 14), dict(no_binning=True)),
                     (_random_edge_scene(90, 14), dict(no_filter=True))):
        g = R.RayTracer((W, H), (0, 0, 0), (0.05, -0.1), 60.0, 4.0, 0.1, seed=6, math_mode=mode, nearest_hit=nearest,
                        smooth_normals=True, **kw)
        o = orc.OracleTracer(W, H, (0.05, -0.1), 60.0, 4.0, 0.1, seed=6, contract=1 - mode, nthreads=8,
                             hit_mode=int(nearest), smooth_normals=True)
        assert g.UploadSceneEdges(rows) and o.upload_scene_edges(rows)
        g.UploadSpheres(sph); o.upload_spheres(sph)
        g.Trace(2, 3, 0)
        assert g.Wait()
        o.trace(2, 3)
        assert np.array_equal(g.RenderBuffer().view(np.uint32), o.render.view(np.uint32))
        assert np.array_equal(g.Image(), o.image) and np.array_equal(g.RngStates(), o.rng)


@pytest.mark.gpu
def test_smooth_flag_needs_an_edge_scene_and_changes_the_picture():
    import raytracertest_amd as R
    from raytracertest_amd import meshes, scenes
    W, H = 96, 64
    rows = meshes.uv_sphere(n_lat=10, n_lon=20)
    imgs = {}
    for smooth in (False, True):
        g = R.RayTracer((W, H), (0, 0, 0), (0, 0), 40.0, 4.0, 0.0, seed=2, smooth_normals=smooth)
        assert g.UploadSceneEdges(rows)
        g.Trace(1, 4, 0); assert g.Wait()
        imgs[smooth] = g.Image().copy()
    assert (imgs[False] != imgs[True]).mean() > 0.1
    # facets: the flat picture has at most one colour per visible triangle (+ background gradient);
    # the smooth one varies continuously inside the silhouette
    centre = (slice(H // 2 - 8, H // 2 + 8), slice(W // 2 - 8, W // 2 + 8))
    assert len(np.unique(imgs[True][centre])) > 4 * len(np.unique(imgs[False][centre]))
    # vertex-layout scenes have no normals: the flag is inert there
    scn = scenes.cornell32()
    a = R.RayTracer((40, 24), (0, 0, 0), (0, 0), 70.0, 3.0, 0.05, seed=2, smooth_normals=True)
    b = R.RayTracer((40, 24), (0, 0, 0), (0, 0), 70.0, 3.0, 0.05, seed=2)
    a.UploadScene(scn); b.UploadScene(scn)
    a.Trace(1, 3, 0); b.Trace(1, 3, 0); assert a.Wait() and b.Wait()
    assert np.array_equal(a.RenderBuffer().view(np.uint32), b.RenderBuffer().view(np.uint32))
    # and re-uploading in the vertex layout after an edge upload drops the normals
    g = R.RayTracer((40, 24), (0, 0, 0), (0, 0), 70.0, 3.0, 0.05, seed=2, smooth_normals=True)
    g.UploadSceneEdges(rows); g.UploadScene(scn)
    g.Trace(1, 3, 0); assert g.Wait()
    assert np.array_equal(g.RenderBuffer().view(np.uint32), b.RenderBuffer().view(np.uint32))

"""headless CLI + the reference's BMP format (Common/Bitmap.h:45-123)."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "raytracertest_amd", "lib")


def gpp(src, exe):
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), src,
                    "-L" + LIBDIR, "-lrt_mi355x", "-Wl,-rpath," + LIBDIR, "-pthread", "-o", exe], check=True)
    return exe


def test_bmp_header_layout_and_cpp_python_agree(tmp_path):
    from raytracertest_amd.bitmap import bmp_header, write_bmp, read_bmp, HEADER_BYTES
    h = bmp_header(3, 2)
    assert HEADER_BYTES == 138 == len(h)
    assert h[:2] == b"BM" and struct.unpack_from("<I", h, 2)[0] == 138 + 24           # mFileSize
    assert struct.unpack_from("<I", h, 10)[0] == 138                                     # mDataOffset
    assert struct.unpack_from("<Iii", h, 14) == (124, 3, -2)                             # header size, width, -height (top-down)
    assert struct.unpack_from("<HHI", h, 26) == (1, 32, 3)                               # planes, bpp, BI_BITFIELDS
    assert struct.unpack_from("<IIII", h, 54) == (0x00FF0000, 0x0000FF00, 0x000000FF, 0xFF000000)
    assert h[70:74] == b"BGRs" and h[74:] == bytes(64)                                   # 0x73524742 little endian
    exe = subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                          os.path.join(ROOT, "tests", "cpp", "bitmap_test.cpp"), "-o", str(tmp_path / "bt")], check=True)
    out = str(tmp_path / "t.bmp")
    subprocess.run([str(tmp_path / "bt"), out], check=True)
    img = np.array([[0xFF000000, 0xFFFF0000, 0xFF00FF00], [0xFF0000FF, 0x80123456, 0x04010203]], np.uint32)
    write_bmp(str(tmp_path / "p.bmp"), img)
    assert open(out, "rb").read() == open(str(tmp_path / "p.bmp"), "rb").read()
    assert np.array_equal(read_bmp(out), img)
    assert np.array_equal(read_bmp(out + ".filled"), np.full((2, 2), 0xFFFFFFFF, np.uint32))


def test_cli_parsers_share_the_reference_defaults():
    from raytracertest_amd.cli import build_parser
    a = build_parser().parse_args([])
    assert (a.w, a.height, a.samples, a.iterations, a.update) == (38, 21, 1, 100, 10)    # App.cpp:13-16
    assert (a.fov, a.focal, a.aperture) == (70.0, 10.0, 4.0)                              # App.cpp:19-21
    a = build_parser().parse_args("-w 64 -h 32 -s 4 -i 2 -u 0 -cya 170 -f 60 -l 3 -a 1 -o x.bmp".split())
    assert (a.w, a.height, a.samples, a.iterations, a.update, a.cya, a.fov_i, a.focal_i, a.aperture_i) == (64, 32, 4, 2, 0, 170, 60, 3, 1)


def test_cpp_cli_builds_and_fails_loudly_without_gpu(tmp_path):
    import raytracertest_amd as R
    exe = gpp(os.path.join(ROOT, "tools", "rt_cli.cpp"), str(tmp_path / "rt_cli"))
    if R.device_count() > 0:
        pytest.skip("a GPU is present")
    out = subprocess.run([exe, "-w", "8", "-h", "8"], capture_output=True, text=True)
    assert out.returncode == 1 and "no CPU fallback" in out.stderr


@pytest.mark.gpu
def test_cli_cpp_and_python_write_the_same_bmp_as_the_api(tmp_path, orc):
    import raytracertest_amd as R
    from raytracertest_amd import scenes
    from raytracertest_amd.bitmap import read_bmp
    exe = gpp(os.path.join(ROOT, "tools", "rt_cli.cpp"), str(tmp_path / "rt_cli"))
    scene_file = str(tmp_path / "cornell.f4")
    scenes.cornell32().astype("<f4").tofile(scene_file)
    common = ["-w", "96", "-h", "54", "-s", "4", "-i", "3", "-u", "1", "-f", "70", "-l", "3", "--aperture", "0.05",
              "--seed", "7", "--scene", scene_file]
    subprocess.run([exe] + common + ["-o", str(tmp_path / "c.bmp"), "-q"], check=True, timeout=120)
    subprocess.run([sys.executable, "-m", "raytracertest_amd.cli"] + common + ["-o", str(tmp_path / "p.bmp"), "-q"],
                   check=True, timeout=300, cwd=ROOT)
    g = R.RayTracer((96, 54), (0, 0, 0), (0, 0), 70.0, 3.0, 0.05, seed=7)
    g.UploadScene(scenes.cornell32())
    g.Trace(3, 4, 1)
    assert g.Wait()
    assert open(str(tmp_path / "c.bmp"), "rb").read() == open(str(tmp_path / "p.bmp"), "rb").read()
    assert np.array_equal(read_bmp(str(tmp_path / "c.bmp")), g.Image())
    o = orc.OracleTracer(96, 54, (0.0, 0.0), 70.0, 3.0, 0.05, seed=7, nthreads=4)      # ... and as the oracle's image
    o.upload_scene(scenes.cornell32())
    o.trace(3, 4)
    assert np.array_equal(read_bmp(str(tmp_path / "c.bmp")), o.image)
    # the reference's integer angle flag: -cya 180 looks at the demo triangles behind the camera
    subprocess.run([exe, "-w", "38", "-h", "21", "-s", "1", "-i", "4", "-u", "0", "-cya", "172", "--aperture", "0.5",
                    "--seed", "3", "-o", str(tmp_path / "d.bmp"), "-q"], check=True, timeout=120)
    assert len(np.unique(read_bmp(str(tmp_path / "d.bmp")))) > 3


@pytest.mark.gpu
def test_cli_reference_default_invocation_equals_the_oracle(tmp_path, orc):
    """The reference app started with no flags (OpenGLView/App.cpp:11-23: 38x21, 100 iterations x 1 sample, update every
    10, fov 70, focal 10, aperture 4; demo scene MainFrame.cpp:230-232) -- only --seed added, the reference seeds from
    the clock (Random.cu:45): the BMP both CLIs write is the oracle's image, and so is every BGRA8 update on the way."""
    from raytracertest_amd import scenes
    from raytracertest_amd.bitmap import read_bmp
    exe = gpp(os.path.join(ROOT, "tools", "rt_cli.cpp"), str(tmp_path / "rt_cli"))
    subprocess.run([exe, "--seed", "11", "-o", str(tmp_path / "c.bmp"), "-q"], check=True, timeout=120)
    subprocess.run([sys.executable, "-m", "raytracertest_amd.cli", "--seed", "11", "-o", str(tmp_path / "p.bmp"), "-q"],
                   check=True, timeout=300, cwd=ROOT)
    o = orc.OracleTracer(38, 21, (0.0, 0.0), 70.0, 10.0, 4.0, seed=11, nthreads=2)
    o.upload_scene(scenes.demo3())
    o.trace(100, 1)
    assert np.array_equal(read_bmp(str(tmp_path / "c.bmp")), o.image)
    assert np.array_equal(read_bmp(str(tmp_path / "p.bmp")), o.image)
    assert len(np.unique(o.image)) > 3          # (farthest hit with negative t accepted: the triangles behind the camera show)


@pytest.mark.gpu
def test_cli_edge_scene_with_smooth_shading(tmp_path):
    """--edges --smooth --nearest through both CLIs = the API with the same options."""
    import raytracertest_amd as R
    from raytracertest_amd import meshes
    from raytracertest_amd.bitmap import read_bmp
    exe = gpp(os.path.join(ROOT, "tools", "rt_cli.cpp"), str(tmp_path / "rt_cli"))
    rows = meshes.uv_sphere(n_lat=12, n_lon=24)
    scene_file = str(tmp_path / "sphere_edges.f4")
    rows.astype("<f4").tofile(scene_file)
    common = ["-w", "80", "-h", "60", "-s", "3", "-i", "2", "-u", "0", "-f", "40", "-l", "4", "--aperture", "0.1",
              "--seed", "5", "--scene", scene_file, "--edges", "--smooth", "--nearest"]
    subprocess.run([exe] + common + ["-o", str(tmp_path / "c.bmp"), "-q"], check=True, timeout=120)
    subprocess.run([sys.executable, "-m", "raytracertest_amd.cli"] + common + ["-o", str(tmp_path / "p.bmp"), "-q"],
                   check=True, timeout=300, cwd=ROOT)
    g = R.RayTracer((80, 60), (0, 0, 0), (0, 0), 40.0, 4.0, 0.1, seed=5, nearest_hit=True, smooth_normals=True)
    assert g.UploadSceneEdges(rows)
    g.Trace(2, 3, 0)
    assert g.Wait()
    assert np.array_equal(read_bmp(str(tmp_path / "c.bmp")), g.Image())
    assert np.array_equal(read_bmp(str(tmp_path / "p.bmp")), g.Image())

"""The source-compatible C++ class (include/RayTracer/RayTracer.h) driven exactly like the
reference's caller (OpenGLView/MainFrame.cpp), compiled with plain g++ against the C ABI."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "raytracertest_amd", "lib")


def build_driver(tmp_path):
    exe = os.path.join(str(tmp_path), "mainframe_like")
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "mainframe_like.cpp"), "-L" + LIBDIR, "-lrt_mi355x",
                    "-Wl,-rpath," + LIBDIR, "-pthread", "-o", exe], check=True)
    return exe


def test_cpp_api_compiles_and_fails_loudly_without_gpu(tmp_path):
    import raytracertest_amd as R
    exe = build_driver(tmp_path)
    if R.device_count() > 0:
        pytest.skip("a GPU is present; see the gpu-marked test")
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 2 and "CREATE_FAILED" in out.stdout and "no CPU fallback" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("bands", [0, 8])
def test_cpp_driver_matches_python_host_mirror(tmp_path, bands):
    """bands = 8: the same caller through the device-list constructor (rt_tracer_create_multi): the frame in 8 row
    bands, same callback cadence (ONE callback per update with the whole frame), same image."""
    import raytracertest_amd as R
    from raytracertest_amd import scenes
    exe = build_driver(tmp_path)
    out = subprocess.run([exe, "5", str(bands)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.search(r"RESULT done=(\d) updates=(\d+) finished=(\d+) size=(\d+) hash=(\d+) count0=(\d+)", out.stdout)
    assert m, out.stdout
    done, updates, finished, size, h, count0 = (int(x) for x in m.groups())
    assert (done, updates, finished, size, count0) == (1, 9, 1, 38 * 21 * 4, 100)   # i = 10, 20, .. 90
    g = R.RayTracer((38, 21), (0, 0, 0), (0, 0), 70.0, 10.0, 4.0, seed=5)
    g.UploadScene(scenes.demo3())
    g.RotateCamera((0.0, 3.0))
    g.SetCameraParameters(70.0, 10.0, 0.5)
    g.Trace(100, 1, 10)
    assert g.Wait()
    ref = 1469598103934665603
    for px in g.Image().ravel():
        ref = ((ref ^ int(px)) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert ref == h
    assert len(np.unique(g.Image())) > 4      # the rotated camera really sees the demo triangles

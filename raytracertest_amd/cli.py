"""Headless front end: `python -m raytracertest_amd.cli`.  The reference's command line
(OpenGLView/App.cpp:62-184: -w -h -s -i -u -cx -cy -cz -cxa -cya -f -l -a, integer values,
defaults App.cpp:11-23) without the GUI, plus scene/seed/output options; the image is saved
in the reference's BMP format (Common/Bitmap.h).  Same options as tools/rt_cli.cpp."""
import argparse
import math
import sys
import time

import numpy as np


def build_parser():
    p = argparse.ArgumentParser(prog="raytracertest_amd.cli", add_help=False)
    p.add_argument("--help", action="help")
    p.add_argument("-w", type=int, default=3840 // 100)
    p.add_argument("-h", type=int, default=2160 // 100, dest="height")
    p.add_argument("-s", type=int, default=1, dest="samples")
    p.add_argument("-i", type=int, default=100, dest="iterations")
    p.add_argument("-u", type=int, default=10, dest="update")
    for k in ("cx", "cy", "cz", "cxa", "cya"):
        p.add_argument("-" + k, type=int, default=0)
    p.add_argument("-f", type=int, default=None, dest="fov_i")
    p.add_argument("-l", type=int, default=None, dest="focal_i")
    p.add_argument("-a", type=int, default=None, dest="aperture_i")
    p.add_argument("--fov", type=float, default=70.0)
    p.add_argument("--focal", type=float, default=10.0)
    p.add_argument("--aperture", type=float, default=4.0)
    p.add_argument("--cxa-rad", type=float, default=None)
    p.add_argument("--cya-rad", type=float, default=None)
    p.add_argument("--scene", default="demo3", help="demo3 | cornell32 | rand10k | sphere1 | uvsphere | <file.f4> (raw float32 x,y,z,w)")
    p.add_argument("--edges", action="store_true", help="the scene file holds (v0, e0, e1) rows with packed vertex normals in .w")
    p.add_argument("--smooth", action="store_true", help="shade with the interpolated vertex normals (edge-format scenes)")
    p.add_argument("--nearest", action="store_true", help="keep the nearest hit with t > 0 instead of the reference's farthest")
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("-o", default="image0.bmp", dest="out")
    p.add_argument("-q", action="store_true", dest="quiet")
    p.add_argument("--device", type=int, default=0)
    return p


def load_scene(name):
    from . import scenes
    z = np.zeros((0, 4), np.float32)
    if name == "demo3":
        return scenes.demo3(), z
    if name == "cornell32":
        return scenes.cornell32(), z
    if name == "rand10k":
        return scenes.random_triangles(10000, 12345), z
    if name == "sphere1":
        return scenes.sphere1()
    if name == "uvsphere":                   # edge format with vertex normals: use with --edges [--smooth]
        from . import meshes
        return meshes.uv_sphere(n_lat=12, n_lon=24), z
    return np.fromfile(name, dtype="<f4").reshape(-1, 4), z


def main(argv=None):
    a = build_parser().parse_args(argv)
    from . import RayTracer
    from .bitmap import write_bmp
    rad = np.float32(0.01745329251994329576923690768489)
    cxa = np.float32(a.cxa) * rad if a.cxa_rad is None else np.float32(a.cxa_rad)     # glm::radians, App.cpp:148
    cya = np.float32(a.cya) * rad if a.cya_rad is None else np.float32(a.cya_rad)
    fov = float(a.fov_i) if a.fov_i is not None else a.fov
    focal = float(a.focal_i) if a.focal_i is not None else a.focal
    aperture = float(a.aperture_i) if a.aperture_i is not None else a.aperture
    tris, spheres = load_scene(a.scene)
    g = RayTracer((a.w, a.height), (a.cx, a.cy, a.cz), (float(cxa), float(cya)), fov, focal, aperture,
                  seed=a.seed, device=a.device, nearest_hit=a.nearest, smooth_normals=a.smooth)
    if tris.shape[0] and not (g.UploadSceneEdges if a.edges else g.UploadScene)(tris):
        sys.exit("scene '%s' has %d float4 (need a positive multiple of 3)" % (a.scene, tris.shape[0]))
    if spheres.shape[0]:
        g.UploadSpheres(spheres)
    state = {"updates": 0, "image": None}
    g.SetUpdateCallback(lambda img, size: state.__setitem__("updates", state["updates"] + 1))
    g.SetFinishedCallback(lambda img, size: state.__setitem__("image", img.copy()))
    t0 = time.perf_counter()
    g.Trace(a.iterations, a.samples, a.update)          # MainFrame.cpp:254-256
    done = g.Wait()
    ms = (time.perf_counter() - t0) * 1e3
    if not done or state["image"] is None:
        sys.exit("trace did not finish: " + g.LastError())
    write_bmp(a.out, state["image"])
    if not a.quiet:
        rays = a.w * a.height * a.iterations * a.samples
        print("%dx%d, %d x %d spp, %d triangles, %d updates: %.2f ms end to end (%.1f Mray/s incl. host hand-off) -> %s"
              % (a.w, a.height, a.iterations, a.samples, tris.shape[0] // 3, state["updates"], ms,
                 rays / ms / 1e3 if ms > 0 else math.inf, a.out))
    return 0


if __name__ == "__main__":
    sys.exit(main())

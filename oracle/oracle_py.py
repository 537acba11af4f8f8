"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so).

TEST INFRASTRUCTURE, NOT PRODUCT CODE: import this only from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product package
(raytracertest_amd) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")

STRICT, FMA = 0, 1


def build(force=False):
    """Compile the C restatement (gcc).  Building the checker is not using it."""
    src = [os.path.join(_HERE, f) for f in ("oracle.c", "oracle_core.inc", "oracle.h", "Makefile")]
    if force or not os.path.exists(_SO) or any(
            os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True,
                       stdout=subprocess.DEVNULL)
    return _SO


class Camera(C.Structure):
    _fields_ = [("angles", C.c_float * 2), ("fov", C.c_float), ("focal", C.c_float),
                ("aperture", C.c_float), ("M", C.c_float * 16)]


class Scene(C.Structure):
    _fields_ = [("tris", C.POINTER(C.c_float)), ("n_tris", C.c_uint32),
                ("spheres", C.POINTER(C.c_float)), ("n_spheres", C.c_uint32), ("hit_mode", C.c_int32),
                ("layout", C.c_int32), ("shade_mode", C.c_int32)]


class Frame(C.Structure):
    _fields_ = [("W", C.c_uint32), ("H", C.c_uint32), ("row0", C.c_uint32), ("rows", C.c_uint32),
                ("render", C.POINTER(C.c_float)), ("counts", C.POINTER(C.c_uint32)),
                ("rng", C.POINTER(C.c_uint32)), ("image", C.POINTER(C.c_uint32))]


# orc_probe_tri (oracle.h)
PROBE_DTYPE = np.dtype([("rays", np.uint32), ("hits", np.uint32), ("wins", np.uint32), ("nan_hits", np.uint32),
                        ("nan_rays", np.uint32), ("form_rejects", np.uint32), ("form_wrong", np.uint32), ("pad", np.uint32),
                        ("det_min", np.float64), ("det_max", np.float64), ("U_min", np.float64), ("U_max", np.float64),
                        ("V_min", np.float64), ("V_max", np.float64), ("q_min", np.float64), ("q_max", np.float64),
                        ("S_min", np.float64), ("S_max", np.float64)])

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        f32p, u32p = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
        L.orc_rng_seed.argtypes = [C.c_uint64, u32p]
        L.orc_rng_init.argtypes = [C.c_uint64, C.c_uint64, u32p]
        L.orc_rng_next.argtypes = [u32p]
        L.orc_rng_next.restype = C.c_uint32
        L.orc_rng_uniform.argtypes = [u32p]
        L.orc_rng_uniform.restype = C.c_float
        L.orc_rng_jump_columns.argtypes = [u32p]
        L.orc_rng_step_linear_n.argtypes = [u32p, C.c_uint64]
        L.orc_rng_matpow_apply.argtypes = [C.c_uint64, u32p]
        L.orc_sincos.argtypes = [C.c_float, f32p, f32p]
        L.orc_tan_half.argtypes = [C.c_float]
        L.orc_tan_half.restype = C.c_float
        L.orc_ray_make.argtypes = [f32p, f32p, C.c_int, C.c_int, f32p]
        L.orc_ray_point.argtypes = [f32p, C.c_float, C.c_int, f32p]
        L.orc_hit_triangle.argtypes = [f32p, f32p, f32p, f32p, C.c_int, C.c_int, f32p, f32p, f32p]
        L.orc_hit_triangle.restype = C.c_int
        L.orc_triangle_normal.argtypes = [f32p, f32p, f32p, C.c_int, f32p]
        L.orc_hit_sphere.argtypes = [f32p, f32p, C.c_int, f32p]
        L.orc_hit_sphere.restype = C.c_int
        L.orc_camera_init.argtypes = [C.POINTER(Camera), f32p, C.c_float, C.c_float, C.c_float]
        L.orc_camera_rotate.argtypes = [C.POINTER(Camera), f32p]
        L.orc_camera_set.argtypes = [C.POINTER(Camera), C.c_float, C.c_float, C.c_float]
        L.orc_camera_pinhole.argtypes = [C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.c_uint32, C.c_int, f32p]
        L.orc_camera_get_ray.argtypes = [C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.c_uint32, u32p, C.c_int, f32p]
        L.orc_uniform_on_disk.argtypes = [u32p, f32p]
        L.orc_radiance.argtypes = [C.POINTER(Scene), f32p, C.c_int, f32p]
        L.orc_frame_rng_init.argtypes = [C.POINTER(Frame), C.c_uint64, C.c_int]
        L.orc_frame_clear.argtypes = [C.POINTER(Frame)]
        L.orc_trace_launch.argtypes = [C.POINTER(Scene), C.POINTER(Camera), C.POINTER(Frame),
                                       C.c_uint32, C.c_int, C.c_int]
        L.orc_convert.argtypes = [C.POINTER(Frame)]
        L.orc_pack_color.argtypes = [C.c_float, C.c_float, C.c_float]
        L.orc_pack_color.restype = C.c_uint32
        L.orc_probe_init.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_probe_init.restype = None
        L.orc_tile_probe.argtypes = [C.POINTER(Scene), C.POINTER(Camera), C.c_uint32, C.c_uint32, u32p, C.c_uint32, f32p,
                                     C.c_uint32, C.c_int, f32p, f32p, C.c_void_p, u32p]
        L.orc_tile_probe.restype = None
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _up(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


# ---------------------------------------------------------------- single items
def rng_init(seed, subsequence):
    s = np.zeros(6, np.uint32)
    lib().orc_rng_init(seed, subsequence, _up(s))
    return s


def rng_seed(seed):
    s = np.zeros(6, np.uint32)
    lib().orc_rng_seed.argtypes = [C.c_uint64, C.POINTER(C.c_uint32)]
    lib().orc_rng_seed(seed, _up(s))
    return s


def rng_skip_subsequences(state, n):
    lib().orc_rng_skip_subsequences.argtypes = [C.POINTER(C.c_uint32), C.c_uint64]
    lib().orc_rng_skip_subsequences(_up(state), n)
    return state


def rng_next(state):
    return int(lib().orc_rng_next(_up(state)))


def rng_uniform(state):
    return np.float32(lib().orc_rng_uniform(_up(state)))


def uniform_on_disk(state):
    """random::UnifromOnDisk (Random.cuh:13-19) from the state {d, v0..v4}, advanced in place (three draws)."""
    xy = np.zeros(2, np.float32)
    lib().orc_uniform_on_disk(_up(state), _fp(xy))
    return xy


def sincos(x):
    s, c = C.c_float(), C.c_float()
    lib().orc_sincos(np.float32(x), C.byref(s), C.byref(c))
    return np.float32(s.value), np.float32(c.value)


def ray_make(o, d, normalize=True, contract=FMA):
    out = np.zeros(6, np.float32)
    lib().orc_ray_make(_fp(_f32(o)), _fp(_f32(d)), int(normalize), contract, _fp(out))
    return out


def ray_point(ray, t, contract=FMA):
    p = np.zeros(3, np.float32)
    lib().orc_ray_point(_fp(_f32(ray)), np.float32(t), contract, _fp(p))
    return p


def hit_triangle(ray, v0, v1, v2, contract=FMA, eps_mode=0):
    t, u, v = C.c_float(), C.c_float(), C.c_float()
    hit = lib().orc_hit_triangle(_fp(_f32(ray)), _fp(_f32(v0)), _fp(_f32(v1)), _fp(_f32(v2)),
                                 contract, eps_mode, C.byref(t), C.byref(u), C.byref(v))
    return bool(hit), np.float32(t.value), np.float32(u.value), np.float32(v.value)


def triangle_normal(a, b, c, contract=FMA):
    n = np.zeros(3, np.float32)
    lib().orc_triangle_normal(_fp(_f32(a)), _fp(_f32(b)), _fp(_f32(c)), contract, _fp(n))
    return n


def hit_sphere(ray, sph, contract=FMA):
    t = C.c_float()
    hit = lib().orc_hit_sphere(_fp(_f32(ray)), _fp(_f32(sph)), contract, C.byref(t))
    return bool(hit), np.float32(t.value)


def radiance(ray, tri_rows=None, spheres=None, contract=FMA, hit_mode=0, layout=0, shade_mode=0):
    """rt::Radiance (Kernels.cuh:68-107) of one ray against a scene given as float4 rows."""
    tris = np.ascontiguousarray(_f32(tri_rows if tri_rows is not None else np.zeros((0, 4))).reshape(-1, 12))
    sph = np.ascontiguousarray(_f32(spheres if spheres is not None else np.zeros((0, 4))).reshape(-1, 4))
    sc = Scene(_fp(tris), tris.shape[0], _fp(sph), sph.shape[0], hit_mode, layout, shade_mode)
    out = np.zeros(3, np.float32)
    lib().orc_radiance(C.byref(sc), _fp(_f32(ray)), contract, _fp(out))
    return out


def normalize(v, contract=FMA):
    out = np.zeros(3, np.float32)
    lib().orc_normalize.argtypes = [C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float)]
    lib().orc_normalize(_fp(_f32(v)), contract, _fp(out))
    return out


def pack_normal(n):
    lib().orc_pack_normal.restype = C.c_float
    return np.float32(lib().orc_pack_normal(_fp(_f32(n))))


def unpack_normal(packed):
    out = np.zeros(3, np.float32)
    lib().orc_unpack_normal.argtypes = [C.c_float, C.POINTER(C.c_float)]
    lib().orc_unpack_normal(C.c_float(float(packed)), _fp(out))
    return out


def camera(angles=(0.0, 0.0), fov_deg=70.0, focal=10.0, aperture=4.0):
    cam = Camera()
    lib().orc_camera_init(C.byref(cam), _fp(_f32(angles)), fov_deg, focal, aperture)
    return cam


def pack_color(r, g, b):
    return int(lib().orc_pack_color(np.float32(r), np.float32(g), np.float32(b)))


# ---------------------------------------------------------------- frames
class OracleTracer:
    """Mirror of RayTracerImpl's device state for a row band, on the CPU.

    RayTracerImpl.cu:17-46 (ctor), :69-87/:236-315 (Trace/TraceFunct), :94-103 (Resize),
    :105-117 (camera), :119-177 (UploadScene)."""

    def __init__(self, width, height, angles=(0.0, 0.0), fov_deg=70.0, focal=10.0, aperture=4.0,
                 seed=1, row0=0, rows=None, contract=FMA, nthreads=1, hit_mode=0, smooth_normals=False):
        self.W, self.H = int(width), int(height)
        self.row0 = int(row0)
        self.rows = int(self.H - self.row0 if rows is None else rows)
        self.contract, self.nthreads, self.seed, self.hit_mode = contract, nthreads, seed, hit_mode
        self.layout, self.shade_mode = 0, int(bool(smooth_normals))
        self.cam = camera(angles, fov_deg, focal, aperture)
        self.tris = np.zeros((0, 12), np.float32)
        self.spheres = np.zeros((0, 4), np.float32)
        n = self.rows * self.W
        self.render = np.zeros((self.rows, self.W, 4), np.float32)
        self.counts = np.zeros((self.rows, self.W), np.uint32)
        self.rng = np.zeros((self.rows, self.W, 6), np.uint32)
        self.image = np.zeros((self.rows, self.W), np.uint32)
        assert n > 0
        self._frame = Frame(self.W, self.H, self.row0, self.rows, _fp(self.render),
                            _up(self.counts), _up(self.rng), _up(self.image))
        lib().orc_frame_rng_init(C.byref(self._frame), seed, nthreads)

    def upload_scene(self, float4s):
        a = _f32(float4s).reshape(-1, 4)
        if a.shape[0] < 3 or a.shape[0] % 3 != 0:      # RayTracerImpl.cu:121-125
            return False
        self.tris = np.ascontiguousarray(a.reshape(-1, 12))
        self.layout = 0
        return True

    def upload_scene_edges(self, float4s):
        """(v0, e0, e1) rows with packed vertex normals in .w (Documentation/gpu.meshes.txt:16-34)."""
        if not self.upload_scene(float4s):
            return False
        self.layout = 1
        return True

    def upload_spheres(self, float4s):
        self.spheres = np.ascontiguousarray(_f32(float4s).reshape(-1, 4))

    def set_camera_parameters(self, fov_deg, focal, aperture):
        lib().orc_camera_set(C.byref(self.cam), fov_deg, focal, aperture)

    def rotate_camera(self, dangles):
        lib().orc_camera_rotate(C.byref(self.cam), _fp(_f32(dangles)))

    def _scene(self):
        return Scene(_fp(self.tris), self.tris.shape[0], _fp(self.spheres), self.spheres.shape[0], self.hit_mode,
                     self.layout, self.shade_mode)

    def tile_probe(self, pixels, lens, forms=None, fc=None):
        """orc_tile_probe for the rays of `pixels` [(x, y of the full frame)] x `lens` [(dx, dy) in the unit disk] against every
        triangle: (per-triangle structured array PROBE_DTYPE, rays that hit nothing)."""
        pix = np.ascontiguousarray(pixels, np.uint32).reshape(-1, 2)
        ln = np.ascontiguousarray(lens, np.float32).reshape(-1, 2)
        out = np.zeros(self.tris.shape[0], PROBE_DTYPE)
        L = lib()
        L.orc_probe_init(out.ctypes.data, out.shape[0])
        nohit = C.c_uint32(0)
        sc = self._scene()
        fo = None if forms is None else np.ascontiguousarray(forms, np.float32).reshape(-1, 18)
        fcv = None if fc is None else np.ascontiguousarray(fc, np.float32).reshape(3)
        assert fo is None or (fo.shape[0] == out.shape[0] and fcv is not None)
        L.orc_tile_probe(C.byref(sc), C.byref(self.cam), self.W, self.H, _up(pix), pix.shape[0], _fp(ln), ln.shape[0],
                         self.contract, None if fo is None else _fp(fo), None if fcv is None else _fp(fcv),
                         out.ctypes.data, C.byref(nohit))
        return out, int(nohit.value)

    def launch(self, samples):
        sc = self._scene()
        lib().orc_trace_launch(C.byref(sc), C.byref(self.cam), C.byref(self._frame), samples,
                               self.contract, self.nthreads)

    def trace(self, iterations, samples_per_iteration):
        lib().orc_frame_clear(C.byref(self._frame))            # RayTracerImpl.cu:242-243
        for _ in range(iterations):                            # :246
            self.launch(samples_per_iteration)                 # :249
        lib().orc_convert(C.byref(self._frame))                # :295
        return self.image

"""Deterministic synthetic scenes for the BASELINE.json configs.

A scene is what rt::RayTracer::UploadScene takes (RayTracer/RayTracer.h:34): a float32
array of shape (3*N, 4) -- three consecutive float4 per triangle, absolute vertices,
.w ignored (RayTracerImpl.cu:139-176, Kernels.cuh:78-81).  Spheres (build-defined
extension, Documentation/ray.sphere.png) are (M, 4): centre xyz + radius.

Everything is generated in float64 with exactly representable steps and cast once to
float32, so every platform produces the same bytes.
"""
import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(seed, n):
    """First n outputs of splitmix64(seed) as uint64 (vectorised, wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + _GOLDEN * np.arange(1, n + 1, dtype=np.uint64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _u01(seed, n):
    """n uniforms in [0,1) with 24 random bits each (exact in float32)."""
    return (splitmix64(seed, n) >> np.uint64(40)).astype(np.float64) * (1.0 / 16777216.0)


def _tri_rows(tris):
    """(N,3,3) vertices -> (3N,4) float4 rows with .w = 0."""
    t = np.asarray(tris, dtype=np.float64).reshape(-1, 3)
    out = np.zeros((t.shape[0], 4), np.float32)
    out[:, :3] = t.astype(np.float32)
    return out


def _quad(p0, p1, p2, p3):
    """Two triangles, counter-clockwise seen from the side cross(p1-p0, p2-p0) points to."""
    return [[p0, p1, p2], [p0, p2, p3]]


def demo3():
    """The reference's built-in demo scene, verbatim data (OpenGLView/MainFrame.cpp:230-232)."""
    return np.array([[0, 0, 10, 1], [0, 1, 10, 0], [1, 0, 10, 0],
                     [1, 0, 10, 0], [0, 1, 10, 1], [1, 1, 10, 0],
                     [0, 1, 10, 0], [0.5, 1.5, 10, 0], [1, 1, 10, 1]], np.float32)


def kat_triangle():
    """The triangle of UnitTests/TriangleHitTest.cpp:178 (z = -10), as a 1-triangle scene."""
    return _tri_rows([[[0, 0, -10], [1, 0, -10], [0, 1, -10]]])


def sphere1():
    """C2: one sphere, centre (0,0,-5), radius 1; no triangles."""
    return np.zeros((0, 4), np.float32), np.array([[0, 0, -5, 1]], np.float32)


def _box(cx, cz, hx, hy_top, hz, cos_a, sin_a, y0=-1.0):
    """Five outward-facing faces (no bottom) of a box standing on y0, rotated about +y."""
    def P(x, y, z):
        return [cx + cos_a * x + sin_a * z, y, cz - sin_a * x + cos_a * z]
    a, b, c, d = (-hx, -hz), (hx, -hz), (hx, hz), (-hx, hz)      # footprint, z = +hz is the camera side
    y1 = hy_top
    faces = []
    faces += _quad(P(d[0], y0, d[1]), P(c[0], y0, c[1]), P(c[0], y1, c[1]), P(d[0], y1, d[1]))  # +z
    faces += _quad(P(b[0], y0, b[1]), P(a[0], y0, a[1]), P(a[0], y1, a[1]), P(b[0], y1, b[1]))  # -z
    faces += _quad(P(c[0], y0, c[1]), P(b[0], y0, b[1]), P(b[0], y1, b[1]), P(c[0], y1, c[1]))  # +x
    faces += _quad(P(a[0], y0, a[1]), P(d[0], y0, d[1]), P(d[0], y1, d[1]), P(a[0], y1, a[1]))  # -x
    faces += _quad(P(d[0], y1, d[1]), P(c[0], y1, c[1]), P(b[0], y1, b[1]), P(a[0], y1, a[1]))  # +y (top)
    return faces


def cornell32():
    """C3: Cornell-style box of exactly 32 triangles in front of the origin, looking -Z.

    Room x,y in [-1,1], z in [-1,-3], open towards the camera: 5 inward-facing wall quads
    (10), ceiling light quad (2), short box (10), tall box (10).  Camera-facing sides are
    counter-clockwise (det > 0 in HitTriangle, Kernels.cuh:40-45).  Under the reference's
    farthest-hit rule (Kernels.cuh:73,84) the walls hide the boxes; the work is the same."""
    t = []
    t += _quad([-1, -1, -3], [1, -1, -3], [1, 1, -3], [-1, 1, -3])        # back   (+z)
    t += _quad([-1, -1, -1], [-1, -1, -3], [-1, 1, -3], [-1, 1, -1])      # left   (+x)
    t += _quad([1, -1, -3], [1, -1, -1], [1, 1, -1], [1, 1, -3])          # right  (-x)
    t += _quad([-1, -1, -1], [1, -1, -1], [1, -1, -3], [-1, -1, -3])      # floor  (+y)
    t += _quad([-1, 1, -3], [1, 1, -3], [1, 1, -1], [-1, 1, -1])          # ceiling(-y)
    t += _quad([-0.3, 0.99, -2.3], [0.3, 0.99, -2.3], [0.3, 0.99, -1.7], [-0.3, 0.99, -1.7])  # light (-y)
    t += _box(0.35, -1.75, 0.3, -0.4, 0.3, 0.95533648912560598, -0.29552020666133955)   # short, -0.3 rad
    t += _box(-0.35, -2.35, 0.3, 0.2, 0.3, 0.92106099400288510, 0.38941834230865052)    # tall,  +0.4 rad
    out = _tri_rows(t)
    assert out.shape == (96, 4)
    return out


def random_triangles(n=10000, seed=12345):
    """C4/C5: n small random triangles (documented generator: splitmix64(seed)).

    Per triangle 12 uniforms in order: centre x,y in [-4,4], z in [-12,-4]; then the
    three vertex offsets, each component in [-0.15,0.15].  Winding is whatever falls out
    (about half are back-facing and culled)."""
    u = _u01(seed, 12 * n).reshape(n, 12)
    centre = np.stack([-4.0 + 8.0 * u[:, 0], -4.0 + 8.0 * u[:, 1], -12.0 + 8.0 * u[:, 2]], axis=1)
    off = (-0.15 + 0.30 * u[:, 3:]).reshape(n, 3, 3)
    return _tri_rows(centre[:, None, :] + off)


# name -> (W, H, iterations, samples, camera kwargs, scene builder)   SURVEY.md section 8d
CONFIGS = {
    "C2": dict(width=512, height=512, iterations=1, samples=1, angles=(0.0, 0.0), fov=70.0,
               focal=10.0, aperture=0.0, seed=1),
    "C3": dict(width=1920, height=1080, iterations=1, samples=16, angles=(0.0, 0.0), fov=70.0,
               focal=3.0, aperture=0.05, seed=1),
    "C4": dict(width=3840, height=2160, iterations=1, samples=64, angles=(0.0, 0.0), fov=70.0,
               focal=3.0, aperture=0.05, seed=1),
    "C5": dict(width=3840, height=2160, iterations=1, samples=256, angles=(0.0, 0.0), fov=70.0,
               focal=3.0, aperture=0.05, seed=1),
}


def scene_for(config):
    """(triangles (3N,4), spheres (M,4)) for a config name."""
    if config == "C2":
        return sphere1()
    if config == "C3":
        return cornell32(), np.zeros((0, 4), np.float32)
    if config in ("C4", "C5"):
        return random_triangles(10000, 12345), np.zeros((0, 4), np.float32)
    raise KeyError(config)

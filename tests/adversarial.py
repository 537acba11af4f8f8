"""Adversarial scenes for the conservative triangle classification (csrc/rt_trace.hpp: tile_misses_triangle,
the per-sample forms, the wave-uniform early-outs of test_triangle).

The classification may drop a triangle for a whole tile only when interval bounds prove that every ray of the
tile's family misses it; a wrong "miss" silently changes a picture.  Uniformly random scenes rarely sit on a
decision boundary, so this generator aims at them.  For a camera and a frame it picks rays of the family --
tile-corner (and edge-midpoint) pixels, lens samples on the rim of the aperture or at its centre -- and builds triangles

  edge      a point of the ray lies exactly on an edge or a vertex (u = 0, v = 0, u + v = 1), then every
            coordinate is moved by 0, 1, 2, 4, 16 or 64 ulp at most: hit/miss of that ray flips inside the family;
  plane     the triangle's plane contains the ray (det = 0) up to a tilt of 0, 1e-7 ... 1e-3 rad, both windings:
            the culling bound det_hi < eps * |w| decides;
  epsdet    facing the ray, sized so that det lands within a factor of 30 of the culling epsilon 1e-10;
  focal     a vertex at the focal point of a tile-corner pixel (+- ulps): in-focus geometry, where the
            family's bounds are tightest, grazing the corner of the tile's focal box;
  graze     touches the family from OUTSIDE: one vertex on the ray of a tile-corner pixel, the rest of the
            triangle pointing away from the tile -- every other ray of the tile misses, the corner ray sits on
            u = v = 0; small and far (|e| down to 1e-5 of the distance), where the reference's own evaluation
            of u, v is dominated by rounding (error ~ 10 ulp * distance / size);
  cover     contains the whole footprint of a tile's family (the four corner pixels' rays, lens rim included) with a
            margin of 0, 1e-6, 1e-4, 1e-2 or 1 of its size: the "certainly hit" verdict (tiles whose one candidate
            needs no intersection test) decides just inside / just outside its own boundary; half of them are
            STACKED on the previous one at a distance larger by 0, 1e-7 ... 1e-2 relative (which of two certain hits is
            the farther one, Kernels.cuh:84, is decided by bounds as well);
  filler    plain random triangles around all of that (candidate lists of realistic length),

at coordinate scales 1e-3 ... 1e4 and apertures from 0 to many times the scene.  The camera math below is a
float64 re-statement used only to AIM (ThinLensCamera.cuh:30-52,111-141); the check itself is the device's
default kernel against its plain reference-order full scan, bit for bit (tools/stress_boundaries.py,
tests/test_gpu_round2.py)."""
import numpy as np


def cam_rotation(ax, ay):
    """mat3 of mat4_cast(angleAxis(ay, Y) * angleAxis(ax, X)): rotate about X first, then about Y."""
    cx, sx, cy, sy = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], np.float64)
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], np.float64)
    return ry @ rx


def family_ray(cam, W, H, px, py, lens):
    """origin, direction (unit), focal point of pixel (px, py) for the lens sample `lens` in the unit disk."""
    rot = cam_rotation(*cam["angles"])
    hh = np.tan(np.radians(cam["fov"]) / 2.0)
    cx = (2.0 * (px + 0.5) / W - 1.0) * hh * (W / float(H))
    cy = (1.0 - 2.0 * (py + 0.5) / H) * hh
    d = rot @ np.array([cx, cy, -1.0])
    d /= np.linalg.norm(d)
    focal = cam["focal"] * d
    o = np.array([lens[0] * cam["aperture"], lens[1] * cam["aperture"], 0.0])
    w = focal - o
    n = np.linalg.norm(w)
    return o, (w / n if n > 0 else d), focal


def ulp_jitter(rng, a, k=64):
    """every float32 moved by a random number of ulps in [-k, k]"""
    a = np.ascontiguousarray(a, np.float32)
    bits = a.view(np.int32).astype(np.int64)
    step = rng.integers(-k, k + 1, a.shape)
    moved = np.where(bits >= 0, bits + step, bits - step)            # sign-magnitude: larger pattern = larger magnitude
    out = np.clip(moved, -0x7F7FFFFF, 0x7F7FFFFF).astype(np.int32).view(np.float32)
    return np.where(np.isfinite(out), out, a)


def _perp(rng, d):
    a = rng.normal(size=3)
    a -= d * (a @ d)
    n = np.linalg.norm(a)
    return a / n if n > 1e-12 else _perp(rng, d)


def _pick_ray(rng, cam, W, H):
    corner = rng.integers(0, 4)
    tx, ty = rng.integers(0, (W + 7) // 8), rng.integers(0, (H + 7) // 8)
    px = min(W - 1, tx * 8 + (0 if corner & 1 else 7))
    py = min(H - 1, ty * 8 + (0 if corner & 2 else 7))
    pick = rng.integers(0, 4)
    if pick == 0:
        px, py = int(rng.integers(0, W)), int(rng.integers(0, H))
    elif pick == 1:
        # the middle of a tile edge (or of the tile): where the focal points of the tile lie farthest off the bilinear
        # interpolant of its four corner pixels' -- full tiles bound their family from those corners plus a curvature term
        mid = [3, 4][int(rng.integers(0, 2))]
        edge = [0, 7, 3, 4][int(rng.integers(0, 4))]
        ox, oy = (mid, edge) if rng.integers(0, 2) else (edge, mid)
        px, py = min(W - 1, tx * 8 + ox), min(H - 1, ty * 8 + oy)
    if rng.integers(0, 3) == 0:
        lens = (0.0, 0.0)
    else:
        th = rng.uniform(0, 2 * np.pi)
        r = rng.choice([1.0, 1.0, 0.999999, 0.5])
        lens = (r * np.cos(th), r * np.sin(th))
    o, d, focal = family_ray(cam, W, H, px, py, lens)
    _, dc, _ = family_ray(cam, W, H, tx * 8 + 3.5, ty * 8 + 3.5, lens)      # the tile's central ray: "away from the tile"
    return o, d, focal, dc


KINDS_DEFAULT = (0.18, 0.18, 0.09, 0.1, 0.1, 0.23, 0.12)
KINDS_COVER = (0.08, 0.05, 0.02, 0.05, 0.1, 0.1, 0.6)     # mostly covers: scenes whose tiles have certain winners


def adversarial_triangles(rng, cam, W, H, n, scale, kinds_p=KINDS_DEFAULT):
    """(n, 3, 3) float32 triangles around the rays of the camera's tile families (see the module docstring)."""
    tris = np.zeros((n, 3, 3), np.float64)
    kinds = rng.choice(7, n, p=list(kinds_p))
    jitter = rng.choice([0, 1, 2, 4, 16, 64], n, p=[0.15, 0.2, 0.2, 0.2, 0.15, 0.1])
    last_cover = None
    for i in range(n):
        o, d, focal, dc = _pick_ray(rng, cam, W, H)
        t = float(rng.choice([-1.0, 1.0, 1.0, 1.0]) * scale * 10.0 ** rng.uniform(-1.0, 1.0))
        if rng.integers(0, 3) == 0:
            t = float(np.linalg.norm(focal - o))                      # in focus
        P = o + t * d
        size = abs(t) * 10.0 ** rng.uniform(-5.0, 0.3) + 1e-30
        k = kinds[i]
        if k == 0:                                                    # edge / vertex through the ray
            a, b = _perp(rng, d), _perp(rng, d)
            tilt = d * rng.uniform(-1, 1)
            v0 = P + (a + tilt) * size * rng.uniform(-1, 1)
            v1 = P + (b + tilt) * size * rng.uniform(0.1, 1)
            v2 = P - (b - tilt * 0.3) * size * rng.uniform(0.1, 1)     # P on the edge v1-v2 (before the tilt is added)
            where = rng.integers(0, 3)
            if where == 1:
                v1 = P.copy()                                         # P at a vertex
            elif where == 2:
                v0 = P + (v1 - P) * -rng.uniform(0.1, 2.0)            # P on the edge v0-v1
            tris[i] = (v0, v1, v2)
        elif k == 1:                                                  # plane contains the ray, tilted by eps
            a = _perp(rng, d)
            eps = float(rng.choice([0.0, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3])) * rng.choice([-1.0, 1.0])
            nrm = np.cross(d, a)
            u = d + nrm * eps
            q = o + d * t * rng.uniform(0.2, 1.5) + a * size * rng.uniform(-0.2, 0.2)
            tris[i] = (q, q + u * size * rng.uniform(0.2, 2.0), q + a * size * rng.uniform(0.2, 2.0) + u * size * rng.uniform(-1, 1))
        elif k == 2:                                                  # det near the culling epsilon 1e-10
            a = _perp(rng, d)
            b = np.cross(d, a)
            s = np.sqrt(10.0 ** rng.uniform(-11.5, -8.5))             # |e1 x e2| ~ s^2
            tris[i] = (P - (a + b) * s / 3, P + a * s, P + b * s)
        elif k == 3:                                                  # a vertex at the focal point of a tile-corner pixel
            a, b = _perp(rng, d), _perp(rng, d)
            tris[i] = (focal, focal + a * size, focal + b * size + d * size * rng.uniform(-1, 1))
        elif k == 5:                                                  # touches the tile's family from outside at a corner ray
            out = d - dc
            out -= d * (out @ d)
            nn = np.linalg.norm(out)
            out = out / nn if nn > 1e-14 else _perp(rng, d)
            side = np.cross(d, out)
            v1 = P + (out + side * rng.uniform(0.0, 1.5)) * size + d * size * rng.uniform(-1, 1)
            v2 = P + (out - side * rng.uniform(0.0, 1.5)) * size + d * size * rng.uniform(-1, 1)
            tris[i] = (P, v1, v2) if rng.integers(0, 2) else (v1, v2, P)
        elif k == 6:                                                  # covers the tile family's footprint at distance t, by a margin
            tx, ty = rng.integers(0, (W + 7) // 8), rng.integers(0, (H + 7) // 8)
            if last_cover is not None and rng.integers(0, 2):          # stacked behind / in front of the previous cover
                tx, ty, t0 = last_cover
                t = t0 * (1.0 + float(rng.choice([0.0, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2])) * rng.choice([-1.0, 1.0]))
            last_cover = (tx, ty, t)
            pts = []
            for cxp, cyp in ((0, 0), (7, 0), (0, 7), (7, 7)):
                for lens in ((1.0, 0.0), (-1.0, 0.0), (0.0, 1.0), (0.0, -1.0)):
                    oo, dd, _ = family_ray(cam, W, H, min(W - 1, tx * 8 + cxp), min(H - 1, ty * 8 + cyp), lens)
                    pts.append(oo + dd * t)
            pts = np.array(pts)
            cen = pts.mean(axis=0)
            _, dcen, _ = family_ray(cam, W, H, tx * 8 + 3.5, ty * 8 + 3.5, (0.0, 0.0))
            a = _perp(rng, dcen)
            b = np.cross(dcen, a)
            rad = max(np.sqrt(((pts - cen) @ a) ** 2 + ((pts - cen) @ b) ** 2).max(), 1e-30)
            m = 2.0 * rad * (1.0 + float(rng.choice([0.0, 1e-6, 1e-4, 1e-2, 1.0])))     # inscribed circle of radius rad (1 + margin)
            tilt = dcen * rad * rng.uniform(-1, 1)
            tris[i] = (cen + a * m + tilt, cen + (-0.5 * a + 0.8660254 * b) * m, cen + (-0.5 * a - 0.8660254 * b) * m - tilt)
        else:                                                         # filler
            c = P + rng.normal(size=3) * size
            tris[i] = c + rng.uniform(-1, 1, (3, 3)) * size
        if rng.integers(0, 2):
            tris[i] = tris[i][[0, 2, 1]]                              # both windings
    out = np.nan_to_num(tris, nan=0.0, posinf=3e38, neginf=-3e38).astype(np.float32)
    for k in (1, 2, 4, 16, 64):
        sel = jitter == k
        if sel.any():
            out[sel] = ulp_jitter(rng, out[sel], k)
    return out


def cover_config(rng):
    """A configuration aimed at the certain-winner verdict: a few to 64 triangles, most of them covering a tile's whole
    ray family by a margin of 0 ... 1 of its size, half of those stacked 0 ... 1e-2 apart in depth; backdrops with slightly
    turned and shifted rivals (the pairwise ordering of certain winners); moderate apertures so that covers exist at all."""
    scale = float(10.0 ** rng.uniform(-3, 4))
    W, H = int(rng.integers(9, 73)), int(rng.integers(9, 49))
    cam = dict(angles=(float(rng.uniform(-3.2, 3.2)), float(rng.uniform(-3.2, 3.2))), fov=float(rng.uniform(5, 120)),
               focal=float(scale * 10.0 ** rng.uniform(-1, 1)),
               aperture=float(scale * rng.choice([0.0, 1e-4, 1e-3, 0.01, 0.03])))
    n = int(rng.choice([3, 8, 20, 40, 62]))
    tris = adversarial_triangles(rng, cam, W, H, n, scale, KINDS_COVER)
    if rng.uniform() < 0.75:
        # a backdrop: one or two big camera-facing triangles beyond (nearly) everything else -- the reference keeps the
        # FARTHEST hit (Kernels.cuh:84), so this is the winner of every tile it covers, like the walls of the Cornell box
        o, dcen, _ = family_ray(cam, W, H, W / 2.0 - 0.5, H / 2.0 - 0.5, (0.0, 0.0))
        far = float(np.abs(tris).max()) * float(rng.choice([1.5, 3.0, 30.0])) + scale
        half = far * np.tan(np.radians(min(cam["fov"], 150.0)) / 2.0) * max(W / float(H), 1.0) + cam["aperture"]
        a = _perp(rng, dcen)
        b = np.cross(dcen, a)
        back = []
        for k in range(int(rng.integers(1, 3))):
            T = far * (1.0 + 0.37 * k)
            cen = o + dcen * T
            R = half * (1.0 + 0.37 * k) * float(rng.choice([0.6, 2.5, 4.0]))      # 0.6: covers only the middle of the frame
            tri = np.array([cen + a * 2.0 * R, cen + (-a + 1.7320508 * b) * R, cen + (-a - 1.7320508 * b) * R])
            e1, e2 = tri[1] - tri[0], tri[2] - tri[0]
            if e1 @ np.cross(dcen, e2) < 0.0:                                    # det > 0: the camera-facing winding
                tri = tri[[0, 2, 1]]
            back.append(tri)
            if rng.uniform() < 0.6:
                # a RIVAL of the backdrop: the same triangle turned by a small angle about an axis through a point of the view
                # and shifted in depth by nothing ... 1e-2 of its distance -- the two q intervals overlap, the planes may cross
                # inside a tile's footprint (no winner can be certain there) or not (the pairwise bound may order them);
                # scanned before or after the backdrop (ties go to the first scanned, Kernels.cuh:84)
                ang = float(rng.choice([0.0, 1e-6, 1e-4, 1e-3, 1e-2, 1e-1])) * float(rng.choice([-1.0, 1.0]))
                axis = a if rng.integers(0, 2) else b
                piv = cen + (a * rng.uniform(-1, 1) + b * rng.uniform(-1, 1)) * half * float(rng.choice([0.0, 0.3, 1.5]))
                c_, s_ = np.cos(ang), np.sin(ang)
                rel = tri - piv
                rot = rel * c_ + np.cross(axis, rel) * s_ + axis * (rel @ axis)[:, None] * (1.0 - c_)      # Rodrigues
                shift = dcen * T * float(rng.choice([0.0, 1e-7, 1e-5, 1e-3, 1e-2])) * float(rng.choice([-1.0, 1.0]))
                rival = piv + rot + shift
                if rng.integers(0, 2):
                    back.insert(len(back) - 1, rival)
                else:
                    back.append(rival)
        tris = np.concatenate([tris, np.nan_to_num(np.array(back), posinf=3e38, neginf=-3e38).astype(np.float32)])
    return dict(scale=scale, W=W, H=H, cam=cam, tris=tris, mode=int(rng.integers(0, 2)), spp=int(rng.integers(1, 9)),
                iters=int(rng.integers(1, 3)), seed=int(rng.integers(1, 1 << 30)), nearest=False)


def adversarial_config(rng, large=False):
    """one configuration: camera, frame, scene and launch parameters"""
    scale = float(10.0 ** rng.uniform(-3, 4))
    W, H = int(rng.integers(9, 73)), int(rng.integers(9, 49))
    cam = dict(angles=(float(rng.uniform(-3.2, 3.2)), float(rng.uniform(-3.2, 3.2))), fov=float(rng.uniform(5, 150)),
               focal=float(scale * 10.0 ** rng.uniform(-1, 1)),
               aperture=float(scale * rng.choice([0.0, 0.0, 1e-3, 0.03, 0.3, 3.0, 100.0])))
    n = int(rng.choice([4096, 5000, 9000])) if large else int(rng.choice([8, 32, 64, 120, 250, 257, 400, 900]))
    if large:       # dense scenes (block / macro lists, per-sample forms): 400 aimed triangles among plain random ones
        aimed = adversarial_triangles(rng, cam, W, H, 400, scale)
        o, d, focal = family_ray(cam, W, H, W // 2, H // 2, (0.0, 0.0))
        c = (o + d * cam["focal"])[None, None, :] + rng.normal(size=(n - 400, 1, 3)) * cam["focal"] * 0.7
        filler = (c + rng.uniform(-1, 1, (n - 400, 3, 3)) * cam["focal"] * 10.0 ** rng.uniform(-2.5, -0.5)).astype(np.float32)
        tris = np.concatenate([aimed, filler])[rng.permutation(n)]
    else:
        tris = adversarial_triangles(rng, cam, W, H, n, scale)
    return dict(scale=scale, W=W, H=H, cam=cam, tris=tris, mode=int(rng.integers(0, 2)), spp=int(rng.integers(1, 9)),
                iters=int(rng.integers(1, 3)), seed=int(rng.integers(1, 1 << 30)), nearest=bool(rng.integers(0, 4) == 0))

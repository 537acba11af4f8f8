#!/usr/bin/env python3
"""How many of C4's kept (tile, triangle) pairs would a joint lens-feasibility test drop if the tile's focal box were split
(2x2 quadrants, 4x4, per pixel)?  numpy re-evaluation of lens_can_pass_forms on the forms rt_dbg_classify exports, for tiles
spread over the frame (axis-aligned camera: a pixel's focal offset is affine in the pixel, the box is split per axis)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracertest_amd as R
from raytracertest_amd import scenes

cfg = scenes.CONFIGS["C4"]; tris, _ = scenes.scene_for("C4")
g = R.RayTracer((cfg["width"], cfg["height"]), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=1)
g.UploadScene(tris)
rng = np.random.default_rng(5)
tiles = np.array([(int(rng.integers(0, 480)) * 8, int(rng.integers(0, 270)) * 8) for _ in range(300)], np.uint32)


def feasible(C, nx, ny, Rr):
    """vectorised lens_can_pass_forms core: C, nx, ny (..., 3) -> bool (...)"""
    R2 = (Rr * 1.001) ** 2 * 1.001
    out = np.zeros(C.shape[:-1], bool)
    for a in range(3):
        i, j = a, (a + 1) % 3
        Ci, Cj = C[..., i], C[..., j]
        li = nx[..., i] ** 2 + ny[..., i] ** 2; lj = nx[..., j] ** 2 + ny[..., j] ** 2
        inside = (Ci >= 0) & (Cj >= 0)
        best = np.full(Ci.shape, np.inf)
        with np.errstate(all="ignore"):
            t = -Ci / li; fx, fy = t * nx[..., i], t * ny[..., i]
            ok = (Ci < 0) & (li > 0) & (Cj + nx[..., j] * fx + ny[..., j] * fy >= 0)
            best = np.where(ok, np.minimum(best, Ci * Ci / li), best)
            t = -Cj / lj; fx, fy = t * nx[..., j], t * ny[..., j]
            ok = (Cj < 0) & (lj > 0) & (Ci + nx[..., i] * fx + ny[..., i] * fy >= 0)
            best = np.where(ok, np.minimum(best, Cj * Cj / lj), best)
            det = nx[..., i] * ny[..., j] - ny[..., i] * nx[..., j]
            par = np.abs(det) <= 1e-6 * np.sqrt(li * lj)
            vx = (-Ci * ny[..., j] + Cj * ny[..., i]) / det; vy = (-nx[..., i] * Cj + nx[..., j] * Ci) / det
            best = np.where(par, np.inf, np.minimum(best, vx * vx + vy * vy))
        out |= (~inside) & (~par) & (best > R2) & np.isfinite(best)
    return ~out


tot = {"kept": 0, 1: 0, 2: 0, 4: 0, 8: 0}
for k in range(0, len(tiles), 50):
    hdr, rec = g.DebugClassify(tiles[k:k + 50], 0, True, 1000)
    for t in range(hdr.shape[0]):
        keep = rec[t][:, 0].astype(int) & 1
        f = rec[t][keep == 1][:, 12:30].astype(np.float64)
        if not len(f):
            continue
        lo, hi, A = hdr[t][0:3].astype(np.float64), hdr[t][3:6].astype(np.float64), float(hdr[t][9])
        frad = (hi - lo) / 2
        c0 = f[:, [0, 3, 6]]; nx = f[:, [1, 4, 7]]; ny = f[:, [2, 5, 8]]
        gr = f[:, 9:18].reshape(-1, 3, 3)                      # [pair, form, axis]
        tot["kept"] += len(f)
        for n in (1, 2, 4, 8):
            any_ok = np.zeros(len(f), bool)
            for ix in range(n):
                for iy in range(n):
                    cx = ((ix + 0.5) / n - 0.5) * 2 * frad[0]; cy = ((iy + 0.5) / n - 0.5) * 2 * frad[1]
                    sub = np.array([frad[0] / n, frad[1] / n, frad[2]])
                    C = c0 + gr[:, :, 0] * cx + gr[:, :, 1] * cy + (np.abs(gr) * sub).sum(-1) * 1.001
                    any_ok |= feasible(C, nx, ny, A)
            tot[n] += int(any_ok.sum())
print("kept by the product (full box):", tot["kept"], "-> numpy full box", tot[1], "| 2x2 sub-boxes", tot[2], "| 4x4", tot[4], "| per pixel (8x8)", tot[8])

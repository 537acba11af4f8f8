#!/usr/bin/env python3
"""Prototype (float64, host): for the C3 tiles whose winner is certainly hit but blocked by the q range of a nearer triangle,
would a PAIRWISE bound order them?  D = Nt_A det'_B - Nt_B det'_A > 0 for every ray of the family means t_A > t_B whenever
both are hit (both det' > 0).  D is affine in (do, dF) up to a bilinear term; bounded like the other polynomials.
  python3 tools/pair_order_probe.py [tiles]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import classification_check as cc
from oracle import oracle_py as orc
from raytracertest_amd import api as rt, scenes

n_tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
cfg = scenes.CONFIGS["C3"]; tris, _ = scenes.scene_for("C3")
W, H = cfg["width"], cfg["height"]
g = rt.RayTracer((W, H), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=1, math_mode=0)
o = orc.OracleTracer(W, H, cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=1, contract=1, nthreads=1, rows=8)
assert g.UploadScene(tris) and o.upload_scene(tris)
g.Trace(1, 16, 0); assert g.Wait()
w0 = g.DebugTileListWords()[:, :, 0]
count, certain = (w0 & 0x3FF).astype(int), (w0 >> 31) != 0
ty, tx = np.nonzero(~certain & (count > 0))
rng = np.random.default_rng(3)
sel = rng.choice(ty.size, min(n_tiles, ty.size), replace=False)
regions = np.stack([tx[sel] * 8, ty[sel] * 8], 1).astype(np.uint32)
hdr, rec = g.DebugClassify(regions, 0, False, 1000)
lens = cc.lens_samples(orc, n_rng=64)
T = tris.reshape(-1, 3, 4)[:, :, :3].astype(np.float64)
v0, e1, e2 = T[:, 0], T[:, 1] - T[:, 0], T[:, 2] - T[:, 0]
N = np.cross(e2, e1)                                            # det' = w . N
tot = blocked = proven = false_pos = 0
for i, (x0, y0) in enumerate(regions):
    r = rec[i]; h = hdr[i].astype(np.float64)
    fl = r[:, 0].astype(np.int32); keep, sure = (fl & 1) != 0, (fl & 2) != 0
    if not (keep & sure).any(): continue
    qlo, qhi = r[:, 7], r[:, 8]
    cand = keep & sure
    A = int(np.flatnonzero(cand)[np.argmax(qlo[cand])])
    others = keep.copy(); others[A] = False
    blk = np.flatnonzero(others & ~(qhi < qlo[A] - 1e-4 * (np.abs(qhi) + abs(qlo[A]))))
    if blk.size == 0: continue
    pix = [(int(x0) + x, int(y0) + y) for y in range(8) for x in range(8)]
    probe, nohit = o.tile_probe(pix, lens)
    rays = len(pix) * lens.shape[0]
    truth = probe["wins"][A] == rays                            # A is every ray's farthest hit
    blocked += truth
    flo, fhi = h[0:3], h[3:6]; fc = 0.5 * (flo + fhi); rr = 0.5 * (fhi - flo)
    oc = np.zeros(3); a = h[10:13]                              # lens box radii (camera at the origin in C3)
    ok_all = True
    for B in blk:
        nA, nB = -(oc - v0[A]) @ N[A], -(oc - v0[B]) @ N[B]      # Nt at the lens centre
        dA, dB = (fc - oc) @ N[A], (fc - oc) @ N[B]              # det' at the centres
        Dc = nA * dB - nB * dA
        gF = nA * N[B] - nB * N[A]
        gO = -gF - dB * N[A] + dA * N[B]                         # Nt = n - do.N: d/d(do) of Nt_A det'_B - Nt_B det'_A
        cr = np.abs(np.cross(N[B], N[A]))
        bil = sum(cr[k] * (a[(k + 1) % 3] * rr[(k + 2) % 3] + a[(k + 2) % 3] * rr[(k + 1) % 3]) for k in range(3))
        D_rad = (rr * np.abs(gF)).sum() + (a * np.abs(gO)).sum() + bil
        NtA_max = abs(nA) + (a * np.abs(N[A])).sum(); NtB_max = abs(nB) + (a * np.abs(N[B])).sum()
        margin = 1e-4 * (NtA_max * r[B, 2] + NtB_max * r[A, 2])
        if not (Dc - 1.0001 * D_rad - margin > 0): ok_all = False
    proven += ok_all and truth
    false_pos += ok_all and not truth
    tot += 1
print("traced tiles sampled %d; with a certainly-hit candidate blocked by a q range: %d; A wins every ray in truth: %d; of those ordered by the pairwise bound: %d; ORDERED BUT NOT TRUE: %d" % (regions.shape[0], tot, blocked, proven, false_pos))

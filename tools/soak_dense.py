#!/usr/bin/env python3
"""Soak of the dense-scene launch machinery (macro lists + per-wave lists in HBM, keyed and kept; split and unsplit launches;
fused iterations) against the same operations on a full-scan tracer (RT_FLAG_NO_BINNING: every ray x every triangle, no lists at
all; frames wide enough for the super tiles now and then): random TraceEnqueue / Launch / Trace with an update cadence, camera swings and lens changes, scene swaps between dense,
mid-size and small scenes, Resize, list reuse on and off, launches in flight in between.  Every phase: all four buffers bit for bit.
  python tools/soak_dense.py [phases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracertest_amd as R
from raytracertest_amd import scenes

phases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
W, H = 256, 160
mk = lambda **kw: R.RayTracer((W, H), (0, 0, 0), (0.05, -0.02), 70.0, 3.0, 0.05, seed=seed, **kw)
g, ref = mk(), mk(no_binning=True)
both = (g, ref)


def scene(kind):
    n = {"dense": int(rng.integers(4096, 9000)), "mid": int(rng.integers(300, 3000)), "small": int(rng.integers(3, 200))}[kind]
    return scenes.random_triangles(n, int(rng.integers(1, 10**6))), n


tris, n = scene("dense")
for t in both:
    assert t.UploadScene(tris)
ops = 0
for ph in range(phases):
    r = rng.random()
    if r < 0.12:
        kind = rng.choice(["dense", "dense", "dense", "mid", "small"])
        tris, n = scene(kind)
        for t in both: assert t.UploadScene(tris)
    elif r < 0.30:
        a = (float(rng.normal(0, 0.05)), float(rng.normal(0, 0.05)))
        for t in both: t.RotateCamera(a)
    elif r < 0.40:
        fov, foc, ap = float(rng.uniform(50, 80)), float(rng.uniform(2, 5)), float(rng.choice([0.0, 0.02, 0.05, 0.2]))
        for t in both: t.SetCameraParameters(fov, foc, ap)
    elif r < 0.48:
        if rng.random() < 0.3:                              # wide enough for the super tiles above the macro tiles (> 16 macro tiles)
            W, H = int(rng.integers(600, 1100)), int(rng.integers(260, 420))
        else:
            W, H = int(rng.integers(64, 420)), int(rng.integers(40, 300))
        for t in both: t.Resize((W, H))
    elif r < 0.56:
        reuse = bool(rng.integers(0, 2))
        for t in both: t.SetListReuse(reuse)
    k = int(rng.integers(1, 5))
    for _ in range(k):
        kind = rng.random()
        spp = int(rng.choice([1, 2, 3, 4, 5, 8, 16]))
        if kind < 0.4:
            it = int(rng.integers(1, 4))
            for t in both: t.TraceEnqueue(it, spp)
        elif kind < 0.7:
            it, upd = int(rng.integers(1, 30)), int(rng.choice([0, 0, 2, 5]))
            for t in both:
                t.SetUpdateCallback((lambda img, size: None) if upd else None)
                t.Trace(it, spp, upd); assert t.Wait()
        else:
            clear, emit = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
            for t in both: t.Launch(spp, clear_first=clear, emit_image=emit)
        ops += 1
    for t in both: t.Sync()
    for name, a, b in zip(("render", "counts", "rng", "image"), (g.RenderBuffer(), g.SampleCounts(), g.RngStates(), g.Image()),
                          (ref.RenderBuffer(), ref.SampleCounts(), ref.RngStates(), ref.Image())):
        if not np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32)):
            print("MISMATCH in phase %d (%s): %dx%d, %d triangles, last error %r" % (ph, name, W, H, n, g.LastError()))
            sys.exit(1)
print("soak_dense ok: %d phases, %d operations, seed %d, last error %r" % (phases, ops, seed, g.LastError()))

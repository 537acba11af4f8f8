#!/usr/bin/env python3
"""Soak of the small-scene launch machinery at split-launch sizes: random sequences of TraceEnqueue / Launch (clearing,
accumulating, emitting, fused) / Trace with changing sample counts, camera swings (the halves switch between rows and
even / odd block rows), list reuse on and off (the list ring wraps), scene swaps and Resize -- launches stay in flight
between operations -- each phase compared bit for bit with the oracle on sampled rows.  python3 tools/soak_split.py [phases] [seed]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import raytracertest_amd as R
from raytracertest_amd import scenes
from oracle import oracle_py as orc

phases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
sizes = [(256, 136), (384, 200), (512, 320)]
scns = [scenes.cornell32(), scenes.random_triangles(20, 3), scenes.random_triangles(200, 4),
        np.array([[-50, -50, -4, 0], [50, -50, -4, 0], [0, 90, -4, 0], [-1, -1, -2, 0], [1, -1, -2, 0], [0, 1, -2, 0]], np.float32)]
W, H = sizes[0]
g = R.RayTracer((W, H), (0, 0, 0), (0.0, 0.0), 70.0, 3.0, 0.05, seed=11)
rots = []                                       # every camera rotation so far: a fresh oracle replays them
scn = scns[0]; g.UploadScene(scn)


def oracles():
    out = []
    for row0 in sorted({0, (H // 2 // 8) * 8, max(0, H - 16)}):
        o = orc.OracleTracer(W, H, (0.0, 0.0), 70.0, 3.0, 0.05, seed=11, contract=1, nthreads=8, row0=row0, rows=min(16, H - row0))
        o.upload_scene(scn)
        for d in rots: o.rotate_camera(d)
        out.append((row0, o))
    return out


O = oracles()
ops = 0
t0 = time.time()
for ph in range(phases):
    n_ops = int(rng.integers(5, 40))
    for _ in range(n_ops):
        k = int(rng.integers(0, 10))
        spp = int(rng.choice([0, 1, 2, 3, 5, 16]))
        if k <= 3:
            it = int(rng.integers(1, 4))
            g.TraceEnqueue(it, spp)
            for _, o in O: o.trace(it, spp)
        elif k <= 6:
            clear, emit = bool(rng.integers(0, 4) == 0), bool(rng.integers(0, 3) == 0)
            g.Launch(spp, clear_first=clear, emit_image=emit)
            for _, o in O:
                if clear: o.trace(0, 0)
                o.launch(spp)
        elif k == 7:
            d = (float(rng.choice([-0.45, -0.1, 0.1, 0.45])), float(rng.choice([-0.2, 0.0, 0.2])))
            g.RotateCamera(d); rots.append(d)
            for _, o in O: o.rotate_camera(d)
        elif k == 8:
            g.SetListReuse(bool(rng.integers(0, 2)))
        else:
            it = int(rng.integers(1, 6))
            g.Trace(it, spp, int(rng.integers(0, 3))); assert g.Wait()
            for _, o in O: o.trace(it, spp)
        ops += 1
    g.Sync()
    render, states, counts = g.RenderBuffer(), g.RngStates(), g.SampleCounts()
    for row0, o in O:
        n = o.render.shape[0]
        assert np.array_equal(render[row0:row0 + n].view(np.uint32), o.render.view(np.uint32)), "phase %d: render rows %d.." % (ph, row0)
        assert np.array_equal(states[row0:row0 + n], o.rng), "phase %d: RNG states rows %d.." % (ph, row0)
        assert np.array_equal(counts[row0:row0 + n], o.counts), "phase %d: counts rows %d.." % (ph, row0)
    what = int(rng.integers(0, 4))
    if what == 0:
        W, H = sizes[int(rng.integers(0, len(sizes)))]
        g.Resize((W, H)); O = oracles()             # (Resize re-creates the RNG states and clears the buffers, RayTracerImpl.cu:94-103)
    elif what == 1:
        scn = scns[int(rng.integers(0, len(scns)))]
        assert g.UploadScene(scn)
        for _, o in O: o.upload_scene(scn)
print("soak_split: %d phases, %d operations, %.0f s: every phase bit-identical to the oracle on sampled rows" % (phases, ops, time.time() - t0))

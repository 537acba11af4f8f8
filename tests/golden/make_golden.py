#!/usr/bin/env python3
"""Generates the committed golden fixtures from the CPU oracle (run in the build container:
`python tests/golden/make_golden.py`).  The reference itself cannot be built or run here
and ships no recorded output, so these vectors pin the ORACLE's results --
the HIP path and the oracle are then both checked against the same bytes on any machine.

Outputs (small, data only):
  rng_kats.json      curand_init/curand/curand_uniform restatement: states + first outputs
  sincos_kats.json   build-owned sincos at fixed arguments (bit patterns)
  frames.json        CRC32 of render/counts/rng/image buffers for small and full-size configs
  frames_crops.npz   16x16 pixel crops (float bit patterns) of the same frames
"""
import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle_py as orc          # noqa: E402
from raytracertest_amd import meshes, scenes  # noqa: E402

THREADS = int(os.environ.get("GOLDEN_THREADS", "8"))


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def rng_kats():
    out = []
    for seed, sub in [(0, 0), (1, 0), (1, 1), (1, 2073599), (12345, 8294399), (2**32 + 7, 4294967295),
                      (1791062901, 797)]:
        s = orc.rng_init(seed, sub)
        st = s.copy()
        raw = [orc.rng_next(st) for _ in range(4)]
        st = s.copy()
        uni = [int(np.float32(orc.rng_uniform(st)).view(np.uint32)) for _ in range(4)]
        out.append({"seed": seed, "subsequence": sub, "state": [int(x) for x in s], "next": raw, "uniform_bits": uni})
    # Marsaglia's own constants (no seed scramble): the published xorwow vector
    s = np.array([6615241, 123456789, 362436069, 521288629, 88675123, 5783321], np.uint32)
    marsaglia = [orc.rng_next(s) for _ in range(5)]
    return {"_note": "state = {d, v0..v4}; uniform_bits are float32 bit patterns in (0,1]",
            "marsaglia_xorwow_first5": marsaglia, "cases": out}


def sincos_kats():
    xs = np.array([0.0, 1e-10, 0.5, 0.78539816, 1.0, 1.5707964, 2.0, 3.1415927, 4.0, 4.712389, 6.0, 6.2831309,
                   6.2831855, -0.3, -2.5, 17.0, 100.0], np.float32)
    rows = []
    for x in xs:
        s, c = orc.sincos(x)
        rows.append([int(x.view(np.uint32)), int(s.view(np.uint32)), int(c.view(np.uint32))])
    return {"_note": "[x_bits, sin_bits, cos_bits] of the build-owned sincos", "rows": rows}


FRAMES = {
    # name: (W, H, row0, rows, scene, spheres, iterations, samples, camera kwargs, seed)
    "demo3_38x21": dict(W=38, H=21, scene="demo3", it=3, spp=1, fov=70.0, focal=10.0, aperture=4.0, angles=(0.0, 3.0)),
    "cornell_96x54": dict(W=96, H=54, scene="cornell", it=2, spp=5, fov=70.0, focal=3.0, aperture=0.05),
    "rand300_80x45": dict(W=80, H=45, scene="rand300", it=1, spp=16, fov=70.0, focal=3.0, aperture=0.05),
    "C2_full": dict(W=512, H=512, scene="sphere1", it=1, spp=1, fov=70.0, focal=10.0, aperture=0.0),
    "C3_full": dict(W=1920, H=1080, scene="cornell", it=1, spp=16, fov=70.0, focal=3.0, aperture=0.05),
    # build-defined extensions: edge-format rows with packed vertex normals, smooth shading, nearest hit
    "uvsphere_smooth_96x64": dict(W=96, H=64, scene="uvsphere", it=2, spp=4, fov=40.0, focal=4.0, aperture=0.1,
                                  edges=True, smooth=True, nearest=True),
    "C4_band4": dict(W=3840, H=2160, row0=1078, rows=4, scene="rand10k", it=1, spp=64, fov=70.0, focal=3.0, aperture=0.05),
    # BASELINE configs[4]: the C4 scene at 256 spp (one launch), same four rows of the 3840x2160 frame
    "C5_band4": dict(W=3840, H=2160, row0=1078, rows=4, scene="rand10k", it=1, spp=256, fov=70.0, focal=3.0, aperture=0.05),
}
# More rows of the full-size dense-scene frames (the default launch's three-level classification was
# oracle-checked on rows 1078..1081 only): the frame's first and last rows, the seams of the 128x64 macro tiles, rows
# spread over the frame -- and for C5 rows inside different 270-row bands of the 8-band partition.
for _r in (0, 62, 377, 707, 1022, 1533, 1899, 2156):
    FRAMES["C4_rows%04d" % _r] = dict(W=3840, H=2160, row0=_r, rows=4, scene="rand10k", it=1, spp=64, fov=70.0, focal=3.0, aperture=0.05)
for _r in (133, 1700, 2156):
    FRAMES["C5_rows%04d" % _r] = dict(W=3840, H=2160, row0=_r, rows=4, scene="rand10k", it=1, spp=256, fov=70.0, focal=3.0, aperture=0.05)


def scene_arrays(name):
    z = np.zeros((0, 4), np.float32)
    return {"demo3": (scenes.demo3(), z), "cornell": (scenes.cornell32(), z),
            "rand300": (scenes.random_triangles(300, 777), z), "sphere1": scenes.sphere1(),
            "rand10k": (scenes.random_triangles(10000, 12345), z),
            "uvsphere": (meshes.uv_sphere(n_lat=10, n_lon=20), z)}[name]


def render(spec, contract):
    tris, sph = scene_arrays(spec["scene"])
    o = orc.OracleTracer(spec["W"], spec["H"], spec.get("angles", (0.0, 0.0)), spec["fov"], spec["focal"],
                         spec["aperture"], seed=1, row0=spec.get("row0", 0), rows=spec.get("rows"),
                         contract=contract, nthreads=THREADS, hit_mode=int(spec.get("nearest", False)),
                         smooth_normals=spec.get("smooth", False))
    if tris.shape[0]:
        (o.upload_scene_edges if spec.get("edges") else o.upload_scene)(tris)
    if sph.shape[0]:
        o.upload_spheres(sph)
    o.trace(spec["it"], spec["spp"])
    return o


def crop_of(o):
    r0 = max(0, o.rows // 2 - 8)
    c0 = max(0, o.W // 2 - 8)
    return o.render[r0:r0 + 16, c0:c0 + 16].view(np.uint32).copy(), [r0, c0]


def frames(only=None):
    meta, crops = {}, {}
    for name, spec in FRAMES.items():
        if only and name not in only:
            continue
        modes = (1, 0) if not name.startswith(("C4_", "C5_")) else (1,)
        for contract in modes:
            key = "%s/%s" % (name, "fma" if contract else "strict")
            o = render(spec, contract)
            crop, origin = crop_of(o)
            meta[key] = {"spec": {k: (list(v) if isinstance(v, tuple) else v) for k, v in spec.items()},
                         "render_crc32": crc(o.render), "counts_crc32": crc(o.counts), "rng_crc32": crc(o.rng),
                         "image_crc32": crc(o.image), "crop_origin": origin,
                         "render_sum": [float(o.render[..., c].astype(np.float64).sum()) for c in range(3)]}
            crops[key.replace("/", "__")] = crop
            print(key, meta[key]["render_crc32"], flush=True)
    return meta, crops


def main():
    if len(sys.argv) > 1:                      # `make_golden.py NAME...`: (re)generate only these frames, keep the rest
        meta = json.load(open(os.path.join(HERE, "frames.json")))
        crops = dict(np.load(os.path.join(HERE, "frames_crops.npz")))
        m, c = frames(only=sys.argv[1:])
        meta.update(m); crops.update(c)
        json.dump(meta, open(os.path.join(HERE, "frames.json"), "w"), indent=1)
        np.savez_compressed(os.path.join(HERE, "frames_crops.npz"), **crops)
        return
    json.dump(rng_kats(), open(os.path.join(HERE, "rng_kats.json"), "w"), indent=1)
    json.dump(sincos_kats(), open(os.path.join(HERE, "sincos_kats.json"), "w"), indent=1)
    meta, crops = frames()
    json.dump(meta, open(os.path.join(HERE, "frames.json"), "w"), indent=1)
    np.savez_compressed(os.path.join(HERE, "frames_crops.npz"), **crops)


if __name__ == "__main__":
    main()

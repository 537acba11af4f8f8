#!/usr/bin/env python3
"""Kernel A/B helper: time the trace kernel of several builds of librt_mi355x.so (and
launch options) on the same device, each in its own subprocess, several rounds
interleaved.  Usage: tools/ab_bench.py --config C3 --rounds 3 name=lib.so[,k=4][,chunk=1024] ..."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys
sys.path.insert(0, %(root)r)
import raytracertest_amd as R
from raytracertest_amd import scenes
cfg = dict(scenes.CONFIGS[%(config)r]); tris, sph = scenes.scene_for(%(config)r)
g = R.RayTracer((cfg["width"], cfg["height"]), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"],
                seed=cfg["seed"], samples_in_flight=%(k)d, lds_chunk=%(chunk)d, no_filter=%(nofilter)r, no_binning=%(nobin)r, bin_list=%(binlist)d)
if tris.shape[0]: g.UploadScene(tris)
if sph.shape[0]: g.UploadSpheres(sph)
import time
from raytracertest_amd import api
if %(preheat)r:
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15: api.dbg_valu_peak(0)      # steady clocks, as bench.py does
if %(cold)r: g.SetListReuse(False)             # bench.py's headline mode: every step classifies afresh
for _ in range(%(warmup)d): g.TraceEnqueue(1, cfg["samples"])
g.Sync(); g.KernelTime()
t0 = time.perf_counter()
for _ in range(%(steps)d): g.TraceEnqueue(1, cfg["samples"])
g.Sync(); wall = (time.perf_counter() - t0) / %(steps)d * 1e6
ms, n = g.KernelTime()
print(json.dumps({"us": ms / n * 1e3, "wall_us": wall, "info": g.Info()}))
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--preheat", action="store_true", help="150 ms of the VALU calibration loop before the warmup steps (steady clocks)")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    variants = []
    for v in a.variants:
        name, rest = v.split("=", 1)
        parts = rest.split(",")
        opts = {"lib": parts[0], "k": 0, "chunk": 0, "nofilter": False, "nobin": False, "binlist": 0, "cold": False, "env": {}}
        for p in parts[1:]:
            key, val = p.split("=")
            if key == "env":                              # env=NAME:VALUE for this variant's subprocess
                k2, v2 = val.split(":", 1)
                opts["env"][k2] = v2
                continue
            opts[key] = (val == "1") if key in ("nofilter", "nobin", "cold") else int(val)
        variants.append((name, opts))
    res = {n: [] for n, _ in variants}
    wall = {n: [] for n, _ in variants}
    for _ in range(a.rounds):
        for name, o in variants:
            env = dict(os.environ)
            env.update(o["env"])
            lib = o["lib"]
            env["RT_MI355X_LIB"] = lib if os.path.isabs(lib) else os.path.join(ROOT, "raytracertest_amd", "lib", lib)
            code = CHILD % dict(root=ROOT, config=a.config, k=o["k"], chunk=o["chunk"], nofilter=o["nofilter"], nobin=o["nobin"], binlist=o["binlist"],
                                cold=o["cold"], warmup=a.warmup, steps=a.steps, preheat=a.preheat)
            out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
            if out.returncode != 0:
                print(name, "FAILED", out.stderr[-400:])
                continue
            line = json.loads(out.stdout.strip().splitlines()[-1])
            res[name].append(line["us"]); wall[name].append(line["wall_us"])
    for name, _ in variants:
        v = sorted(res[name])
        if v:
            w = sorted(wall[name])
            print("%-22s kernel min %9.1f median %9.1f us | step wall min %9.1f median %9.1f us  (%s)"
                  % (name, v[0], v[len(v) // 2], w[0], w[len(w) // 2], ", ".join("%.1f" % x for x in wall[name])))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Per-wave timeline of one trace launch (experiment build only):
  make -C raytracertest_amd/csrc OUT=../lib/exp_timeline.so BUILD=_build/tl EXTRA=-DRT_TIMELINE
  RT_MI355X_LIB=raytracertest_amd/lib/exp_timeline.so python tools/timeline.py [spp] [scene]
Marks (shader clock): 0 start, 1 family+classification done, 2 first sample batch done (includes
the wait for the RNG state), 3 all samples done, 4 stores issued; word 6 = HW_ID, word 7 = XCC_ID<<32 | realtime."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracertest_amd as R
from raytracertest_amd import api, scenes

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
scene = sys.argv[2] if len(sys.argv) > 2 else "cornell32"
cfg = scenes.CONFIGS["C4" if scene == "rand10k" else "C3"]
g = R.RayTracer((cfg["width"], cfg["height"]), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=1)
if scene == "cornell32":
    g.UploadScene(scenes.cornell32())
elif scene == "rand10k":
    g.UploadScene(scenes.random_triangles(10000, 12345))
for _ in range(5):
    g.TraceEnqueue(1, spp)
g.Sync(); g.KernelTime()
for _ in range(20):
    g.TraceEnqueue(1, spp)
g.Sync(); ms, n = g.KernelTime()
print("scene %s, %d spp: kernel %.1f us (instrumented build, timeline off)" % (scene, spp, ms / n * 1e3))
L = api.load_library()
words = ((cfg["width"] + 31) // 32) * ((cfg["height"] + 7) // 8) * 4 * 16
buf = np.zeros(words, np.uint64)
L.rt_dbg_trace_timeline.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]
for rep in range(2):                      # second run = warm
    rc = L.rt_dbg_trace_timeline(g._h, spp, buf.ctypes.data, words)
    assert rc == 0, g.LastError()
t = buf.reshape(-1, 16)
t = t[t[:, 0] > 0]
hw = t[:, 6].astype(np.uint64)
xcc = (t[:, 7] >> np.uint64(32)).astype(np.int64) & 0xF
simd = ((hw >> np.uint64(4)) & np.uint64(3)).astype(np.int64)
cu = ((hw >> np.uint64(8)) & np.uint64(15)).astype(np.int64)
sh = ((hw >> np.uint64(12)) & np.uint64(1)).astype(np.int64)
se = ((hw >> np.uint64(13)) & np.uint64(7)).astype(np.int64)
unit = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
print("waves %d, distinct SIMDs %d, XCCs %s" % (t.shape[0], len(np.unique(unit)), np.unique(xcc).tolist()))
c = t[:, :5].astype(np.int64)              # 100 MHz realtime counter (global): 10 ns units
us = (c - c[:, 0].min()) / 100.0
life = us[:, 4] - us[:, 0]
print("kernel span %.1f us; wave lifetime mean %.2f us (p10 %.2f p50 %.2f p90 %.2f)" % (us[:, 4].max(), life.mean(), *np.percentile(life, [10, 50, 90])))
d = np.diff(us, axis=1)
for i, nm in enumerate(["family+classify", "first batch (+RNG wait)", "other batches", "store issue"]):
    print("  %-26s mean %6.2f us  p10 %6.2f  p50 %6.2f  p90 %6.2f" % (nm, d[:, i].mean(), *np.percentile(d[:, i], [10, 50, 90])))
pm = (t[:, [0, 8, 9, 10, 1]].astype(np.int64) - t[:, [0]].astype(np.int64)) / 100.0
for i, nm in enumerate(["state loads issued", "pinhole + focal point + wave bounds", "make_family", "classification"]):
    seg = pm[:, i + 1] - pm[:, i]
    print("    prologue: %-30s mean %6.2f us  p10 %6.2f  p90 %6.2f" % (nm, seg.mean(), *np.percentile(seg, [10, 90])))
KK = 2 if scene != 'rand10k' else 4
nb = (spp + KK - 1) // KK
if nb > 1:
    pb = d[:, 2] / (nb - 1)
    print("  per batch %.2f us; first-batch excess (exposed RNG wait) %.2f us" % (pb.mean(), (d[:, 1] - pb).mean()))
res = 0.25
nbins = int(us[:, 4].max() / res) + 2
alive = np.zeros(nbins); loop = np.zeros(nbins); pro = np.zeros(nbins)
for s_, a, b, e in zip((us[:, 0] / res).astype(int), (us[:, 1] / res).astype(int), (us[:, 3] / res).astype(int), (us[:, 4] / res).astype(int)):
    alive[s_:e + 1] += 1; loop[a:b + 1] += 1; pro[s_:a + 1] += 1
ns = len(np.unique(unit))
print("avg per SIMD over the span: resident %.2f waves, in sample loop %.2f, in prologue %.2f" % (alive.mean() / ns, loop.mean() / ns, pro.mean() / ns))
print("t(us)  resident  in-loop  prologue   (waves per SIMD)")
step = max(1, nbins // 48)
for i in range(0, nbins, step):
    print("%6.1f  %5.2f  %5.2f  %5.2f" % (i * res, alive[i] / ns, loop[i] / ns, pro[i] / ns))
cnt = np.bincount(unit); cnt = cnt[cnt > 0]
print("waves per SIMD: min %d mean %.1f max %d" % (cnt.min(), cnt.mean(), cnt.max()))
np.save(os.environ.get("TIMELINE_NPY", "gpurun_out/timeline_%s_%d.npy" % (scene, spp)), t)

#!/usr/bin/env python3
"""Per-wave timeline of one whole-band C3 launch by KIND of tile (experiment build, see tools/timeline.py):
certain-winner waves (memory + generator steps) against waves that generate rays.
  make -C raytracertest_amd/csrc OUT=../lib/exp_timeline.so BUILD=_build/tl EXTRA=-DRT_TIMELINE
  RT_MI355X_LIB=raytracertest_amd/lib/exp_timeline.so python3 tools/timeline_kinds.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracertest_amd as R
from raytracertest_amd import api, scenes

cfg = scenes.CONFIGS["C3"]
W, H = cfg["width"], cfg["height"]
g = R.RayTracer((W, H), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=1)
g.UploadScene(scenes.cornell32())
for _ in range(30): g.TraceEnqueue(1, 16)
g.Sync()
L = api.load_library()
gx, gy = (W + 31) // 32, (H + 7) // 8
words = gx * gy * 4 * 16
buf = np.zeros(words, np.uint64)
L.rt_dbg_trace_timeline.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]
for rep in range(3):
    assert L.rt_dbg_trace_timeline(g._h, 16, buf.ctypes.data, words) == 0, g.LastError()
w0 = g.DebugTileListWords()[:, :, 0].reshape(-1)            # (gy, gx*4) in slot order = timeline slot order
t = buf.reshape(-1, 16)
ok = t[:, 0] > 0
sure = ((w0 >> 31) != 0)[ok]
t = t[ok]
hw = t[:, 6]
xcc = (t[:, 7] >> np.uint64(32)).astype(np.int64) & 0xF
simd = ((hw >> np.uint64(4)) & np.uint64(3)).astype(np.int64)
cu = ((hw >> np.uint64(8)) & np.uint64(15)).astype(np.int64)
sh = ((hw >> np.uint64(12)) & np.uint64(1)).astype(np.int64)
se = ((hw >> np.uint64(13)) & np.uint64(7)).astype(np.int64)
unit = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
ns = len(np.unique(unit))
start = t[:, 0].astype(np.int64); end = t[:, 4].astype(np.int64)
t0 = start.min()
s_us = (start - t0) / 100.0; e_us = (end - t0) / 100.0
life = e_us - s_us
print("waves %d on %d SIMDs; kernel span %.1f us (ONE kernel over the whole band, timestamps on)" % (t.shape[0], ns, e_us.max()))
for name, m in (("certain-winner", sure), ("ray-generating", ~sure)):
    l = life[m]
    print("%-15s %6d waves  lifetime mean %6.2f us  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f;  wave-time %.0f us total" % (
        name, m.sum(), l.mean(), *np.percentile(l, [10, 50, 90]), l.max(), l.sum()))
res = 0.5
nb = int(e_us.max() / res) + 2
al = {k: np.zeros(nb) for k in ("sure", "rays")}
starts = {k: np.zeros(nb) for k in ("sure", "rays")}
for s_, e_, su in zip((s_us / res).astype(int), (e_us / res).astype(int), sure):
    k = "sure" if su else "rays"
    al[k][s_:e_ + 1] += 1
    starts[k][s_] += 1
print("t(us)   resident waves per SIMD: certain | rays     wave starts per us: certain | rays")
step = max(1, nb // 40)
for i in range(0, nb, step):
    j = slice(i, i + step)
    print("%6.1f   %5.2f | %5.2f      %7.0f | %7.0f" % (i * res, al["sure"][j].mean() / ns, al["rays"][j].mean() / ns,
                                                     starts["sure"][j].sum() / (step * res), starts["rays"][j].sum() / (step * res)))

#!/bin/bash
# Shader clock and power (rocm-smi, read-only) while the C3 loop runs: the mixed frame, the same frame with every tile traced,
# and an all-certain frame.  bash tools/clock_under_load.sh
sample() { for i in 1 2 3 4; do sleep 1; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr '\n' ' '; echo; done; }
echo "== idle"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr '\n' ' '; echo
echo "== C3 mixed (lists kept)"
python3 - <<'PY' &
import sys, time; sys.path.insert(0, ".")
import raytracertest_amd as R
from raytracertest_amd import scenes
cfg = scenes.CONFIGS["C3"]
g = R.RayTracer((1920, 1080), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=1); g.UploadScene(scenes.cornell32())
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 6.0:
    for _ in range(2000): g.TraceEnqueue(1, 16)
    g.Sync(); n += 2000
print("mixed: %.1f us/step" % ((time.perf_counter() - t0) / n * 1e6))
PY
sample; wait
echo "== C3 every tile traced (RT_FLAG_NO_SURE_HIT)"
python3 - <<'PY' &
import sys, time; sys.path.insert(0, ".")
import raytracertest_amd as R
from raytracertest_amd import scenes
cfg = scenes.CONFIGS["C3"]
g = R.RayTracer((1920, 1080), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=1, no_sure_hit=True); g.UploadScene(scenes.cornell32())
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 6.0:
    for _ in range(1000): g.TraceEnqueue(1, 16)
    g.Sync(); n += 1000
print("all traced: %.1f us/step" % ((time.perf_counter() - t0) / n * 1e6))
PY
sample; wait
echo "== one triangle, every tile certain"
python3 - <<'PY' &
import sys, time; sys.path.insert(0, ".")
import numpy as np
import raytracertest_amd as R
from raytracertest_amd import scenes
cfg = scenes.CONFIGS["C3"]
g = R.RayTracer((1920, 1080), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=1)
g.UploadScene(np.array([[-100, -100, -4, 0], [100, -100, -4, 0], [0, 200, -4, 0]], np.float32))
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 6.0:
    for _ in range(4000): g.TraceEnqueue(1, 16)
    g.Sync(); n += 4000
print("all certain: %.1f us/step" % ((time.perf_counter() - t0) / n * 1e6))
PY
sample; wait

for rep in 1 2; do
for v in 0 1; do
  RT_MI355X_NO_PRETEST=$v python bench.py --config C4 --steps 10 --warmup 2 --no-valu --cpu-rows 0 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NO_PRETEST=$v', d['value'], d['ms_per_step'])"
done; done
python - <<'PY'
import sys; sys.path.insert(0, '.')
import raytracertest_amd as R
from raytracertest_amd import scenes
cfg = scenes.CONFIGS["C4"]; tris, _ = scenes.scene_for("C4")
g = R.RayTracer((cfg["width"], cfg["height"]), (0, 0, 0), cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=1)
g.UploadScene(tris); st = g.TraceStats(4); print(st)
tot = st["pretest_skips"] + st["skip_a"] + st["skip_b"] + st["skip_c"] + st["reach_d"]
print("pretest skips %.3f of candidate tests; then A %.3f B %.3f C %.3f D %.3f" % tuple(st[k] / tot for k in ("pretest_skips", "skip_a", "skip_b", "skip_c", "reach_d")))
PY

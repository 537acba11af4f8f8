import sys; sys.path.insert(0,'.')
import raytracertest_amd as R
from raytracertest_amd import scenes
for name, scn in (("cornell", scenes.cornell32()), ("rand10k", scenes.random_triangles(10000,12345))):
    g = R.RayTracer((480, 270), (0, 0, 0), (0, 0), 70.0, 3.0, 0.05, seed=1)
    g.UploadScene(scn); st = g.TraceStats(4); waves = (480//8)*((270+7)//8)
    print(name, "waves", waves, st)

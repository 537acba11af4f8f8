import sys; sys.path.insert(0,'/root/repo')
import raytracertest_amd as R
from raytracertest_amd import scenes
for cfgname in ("C3",):
    cfg=scenes.CONFIGS[cfgname]; tris,sph=scenes.scene_for(cfgname)
    g=R.RayTracer((cfg["width"],cfg["height"]),(0,0,0),cfg["angles"],cfg["fov"],cfg["focal"],cfg["aperture"],seed=1)
    g.UploadScene(tris)
    st=g.TraceStats(16)
    waves=((cfg["width"]+7)//8)*((cfg["height"]+7)//8)
    print(cfgname, "tiles", waves, "sure-hit batches", st["pretest_skips"], "-> tiles", st["pretest_skips"]/8.0, "share %.3f"%(st["pretest_skips"]/8.0/waves), "cand/tile", st["bin_candidates"]/st["bin_rounds"])

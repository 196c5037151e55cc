import time, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import raytracer_amd as rt
sp, mt = rt.scene_rtiow(1)
w = rt.BVH(sp, mt)
cam = rt.config_rtiow(3840, 100, 50).camera()
buf = np.zeros((cam.image_height, 3840, 3))
cam.render_sum(w, seed=1, spp_end=1, rgb_sum=buf)
for spp in (1, 100):
    t0 = time.perf_counter(); _, st = cam.render_sum(w, seed=1, spp_end=spp, rgb_sum=buf); dt = time.perf_counter() - t0
    print("spp", spp, "wall ms %.2f kernel ms %.2f overhead ms %.2f" % (dt*1e3, st["kernel_ms"], dt*1e3 - st["kernel_ms"]))

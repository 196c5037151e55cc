"""Times the two ways from a finished render to PPM text at 7680x4320: host frame + host quantise, or device quantise + bytes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as rt
sp, mt = rt.scene_rtiow(1)
w = rt.BVH(sp, mt)
cam = rt.config_rtiow(7680, 1, 50).camera()
fd = os.open("/dev/null", os.O_WRONLY)
cam.render(w, seed=1, fd=fd)
t0 = time.perf_counter(); st = cam.render(w, seed=1, fd=fd); t1 = time.perf_counter()
print("rt_render_ppm: wall %.0f ms (kernel %.0f ms)" % ((t1 - t0) * 1e3, st["kernel_ms"]))
t0 = time.perf_counter(); buf, st = cam.render_sum(w, seed=1); t1 = time.perf_counter()
rt.write_ppm(fd, buf, 7680, cam.image_height, 1); t2 = time.perf_counter()
print("rt_render + rt_write_ppm: render+copy %.0f ms (kernel %.0f ms), quantise+format %.0f ms" % ((t1 - t0) * 1e3, st["kernel_ms"], (t2 - t1) * 1e3))

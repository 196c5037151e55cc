#!/usr/bin/env python3
"""A scene ten times the largest configured one: built by each device builder and rendered; frames and ray counts compared.
python tools/exp_max_scene.py [n_spheres]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    t0 = time.time()
    spheres, materials = rt.scene_field(4, n, 0.15, 0.15, 1.0)
    print("scene of %d spheres generated in %.1f s" % (n, time.time() - t0), flush=True)
    cam = rt.Camera(rt.config_field(n, 1920, 8, 50))
    frames = {}
    for name, builder in (("device sah", rt.BUILD_DEVICE_SAH), ("device clustering", rt.BUILD_DEVICE_LBVH)):
        t0 = time.time()
        w = rt.BVH(spheres, materials, builder=builder)
        info = w.info()
        print("%-18s tree %.1f ms, scene creation %.1f ms, %d nodes, depth %d, %d device nodes" % (name, info["build_ms"], 1e3 * (time.time() - t0), info["n_nodes"], info["depth"], info["n_device_nodes"]), flush=True)
        t0 = time.time()
        img, stats = cam.render_sum(w, seed=12345)
        rays = stats["rays"]
        dt = time.time() - t0
        print("%-18s %d rays in %.1f ms: %.0f Mrays/s (upload and download included)" % (name, rays, 1e3 * dt, rays / dt / 1e6), flush=True)
        frames[name] = (np.asarray(img).copy(), rays)
        w.close()
    a, b = frames["device sah"], frames["device clustering"]
    print("ray counts equal: %s; frames bit-identical: %s" % (a[1] == b[1], np.array_equal(a[0], b[0])), flush=True)


if __name__ == "__main__":
    main()

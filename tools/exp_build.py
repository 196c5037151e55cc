#!/usr/bin/env python3
"""Scene creation with each tree builder, timed:  python tools/exp_build.py [n ...]   (under rocprofv3 --kernel-trace --stats for the kernels' shares)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt

BUILDERS = {"host": rt.BUILD_HOST_SAH, "device clustering": rt.BUILD_DEVICE_LBVH, "device sah": rt.BUILD_DEVICE_SAH}


def main():
    sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [10000, 100000, 1000000]
    only = os.environ.get("BUILDERS", "host,device clustering,device sah").split(",")
    for n in sizes:
        spheres, materials = rt.scene_field(4, n, 0.15, 0.15, 1.0)
        for name in only:
            ms, wall = [], []
            for _ in range(3):
                t0 = time.time()
                w = rt.BVH(spheres, materials, builder=BUILDERS[name])
                wall.append(1e3 * (time.time() - t0))
                ms.append(w.info()["build_ms"])
                w.close()
            print("%8d spheres, %-18s tree %.1f ms (of three: %s), scene creation as a whole %.1f ms" % (n, name, min(ms), " ".join("%.1f" % v for v in ms), min(wall)), flush=True)


if __name__ == "__main__":
    main()

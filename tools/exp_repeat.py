#!/usr/bin/env python3
"""Times the same render several times back to back (clock ramp-up and launch-to-launch spread): python tools/exp_repeat.py c3 3840 64 [repeats]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch

import raytracer_amd as rt
from run_configs import CONFIGS

name, width, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
repeats = int(sys.argv[4]) if len(sys.argv) > 4 else 6
if name == "c1":
    spheres, materials = rt.scene_rtiow(1)
    cfg = rt.config_rtiow(width, spp, 50)
else:
    n, pl, pm, fz, _, _, depth, seed = CONFIGS[name]
    spheres, materials = rt.scene_field(seed, n, pl, pm, fz)
    cfg = rt.config_field(n, width, spp, depth)
world = rt.BVH(spheres, materials)
cam = cfg.camera()
frame = torch.zeros((cam.image_height, width, 3), dtype=torch.float64, device="cuda")
out = []
for r in range(repeats):
    frame.zero_()
    torch.cuda.synchronize()
    st = cam.render_device(world, frame.data_ptr(), seed=12345, spp_begin=0, spp_end=spp)
    out.append("%.0f (%.1f ms)" % (st["rays"] / st["kernel_ms"] / 1e3, st["kernel_ms"]))
print(name, width, spp, "Mrays/s per launch:", ", ".join(out))

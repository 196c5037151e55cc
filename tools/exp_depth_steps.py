#!/usr/bin/env python3
"""Node steps per ray by path depth (camera rays alone, camera rays + first bounce, whole paths) on the configurations of
tools/run_configs.py, from the instrumented kernel's counters:

  RT_PROFILE_BLOCKS=1 python tools/exp_depth_steps.py c2,c3,c4
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch

import raytracer_amd as rt
from run_configs import CONFIGS

for name in sys.argv[1].split(","):
    n, pl, pm, fz, width, spp, depth, seed = CONFIGS[name]
    spp = max(4, spp // 4)
    spheres, materials = rt.scene_rtiow(seed) if n is None else rt.scene_field(seed, n, pl, pm, fz)
    world = rt.BVH(spheres, materials)
    for d in (1, 2, depth):
        cfg = rt.config_rtiow(width, spp, d) if n is None else rt.config_field(n, width, spp, d)
        cam = cfg.camera()
        frame = torch.zeros((cam.image_height, width, 3), dtype=torch.float64, device="cuda")
        st = cam.render_device(world, frame.data_ptr(), seed=12345)
        print(name, "depth", d, "rays", st["rays"], "steps/ray %.3f tests/ray %.3f" % (st["node_steps"] / st["rays"], st["sphere_tests"] / st["rays"]), flush=True)
    world.close()

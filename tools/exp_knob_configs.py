#!/usr/bin/env python3
"""Runs stress configurations with debug knobs set before the scene is made: python tools/exp_knob_configs.py c3,c4 force_global_stack=1"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

import raytracer_amd as rt  # noqa: E402
from run_configs import CONFIGS  # noqa: E402

names = sys.argv[1].split(",")
knobs = dict(kv.split("=") for kv in sys.argv[2:])
for k, v in knobs.items():
    rt._lib.check(rt._lib.lib().rt_debug_set(k.encode(), C.c_double(float(v))))
for name in names:
    n, pl, pm, fz, width, spp, depth, seed = CONFIGS[name]
    spp = max(1, spp // 2)
    spheres, materials = rt.scene_field(seed, n, pl, pm, fz)
    cfg = rt.config_field(n, width, spp, depth)
    world = rt.BVH(spheres, materials)
    cam = cfg.camera()
    frame = torch.zeros((cam.image_height, width, 3), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    cam.render_device(world, frame.data_ptr(), seed=1, spp_begin=0, spp_end=1)
    st = cam.render_device(world, frame.data_ptr(), seed=12345)
    info = world.info()
    print(json.dumps({"config": name, "knobs": knobs, "nodes_in_lds": info["n_nodes_in_lds"], "lds_bytes": info["lds_bytes"], "depth": info["depth"],
                      "Mrays_per_s": st["rays"] / st["kernel_ms"] / 1e3, "ms": st["kernel_ms"],
                      "node_steps_per_ray": st["node_steps"] / max(1, st["rays"]), "sphere_tests_per_ray": st["sphere_tests"] / max(1, st["rays"])}), flush=True)
    world.close()

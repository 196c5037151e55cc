#!/usr/bin/env python3
"""C2-like runs with debug knobs set before the scene is made: python tools/exp_c2_wide.py [knob=value ...]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

import raytracer_amd as rt  # noqa: E402

knobs = dict(kv.split("=") for kv in sys.argv[1:])
for k, v in knobs.items():
    rt._lib.check(rt._lib.lib().rt_debug_set(k.encode(), C.c_double(float(v))))
world = rt.BVH(*rt.scene_rtiow(1))
for width, spp in ((3840, 200), (830, 750)):
    cam = rt.config_rtiow(width, spp, 50).camera()
    frame = torch.zeros((cam.image_height, width, 3), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    cam.render_device(world, frame.data_ptr(), seed=1, spp_begin=0, spp_end=1)
    best = None
    for _ in range(2):
        st = cam.render_device(world, frame.data_ptr(), seed=12345)
        if best is None or st["kernel_ms"] < best["kernel_ms"]:
            best = st
    info = world.info()
    print(json.dumps({"width": width, "spp": spp, "knobs": knobs, "nodes": info["n_device_nodes"], "nodes_in_lds": info["n_nodes_in_lds"], "lds_bytes": info["lds_bytes"],
                      "in_lds": info["spheres_in_lds"], "Mrays_per_s": best["rays"] / best["kernel_ms"] / 1e3, "ms": best["kernel_ms"],
                      "node_steps_per_ray": best["node_steps"] / max(1, best["rays"]), "sphere_tests_per_ray": best["sphere_tests"] / max(1, best["rays"])}), flush=True)

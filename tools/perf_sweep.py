#!/usr/bin/env python3
"""Times the render kernel under different values of a tuning knob: python tools/perf_sweep.py NAME v1 v2 ... [--spp N] [--width W]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse

import torch

import raytracer_amd as rt

ap = argparse.ArgumentParser()
ap.add_argument("name")
ap.add_argument("values", nargs="+", type=float)
ap.add_argument("--spp", type=int, default=100)
ap.add_argument("--width", type=int, default=3840)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--builder", type=int, default=0, help="0: host SAH, 1: device LBVH")
ap.add_argument("--config", default=None, help="c3, c4 or c5 of tools/run_configs.py instead of the reference scene (--width and --spp still apply)")
a = ap.parse_args()

L = rt._lib.lib()
if a.config:
    from run_configs import CONFIGS

    n, pl, pm, fz, _, _, depth, seed = CONFIGS[a.config]
    spheres, materials = rt.scene_field(seed, n, pl, pm, fz)
    cfg = rt.config_field(n, a.width, a.spp, depth)
else:
    spheres, materials = rt.scene_rtiow(1)
    cfg = rt.config_rtiow(a.width, a.spp, 50)
SCENE_KNOBS = ("first_leaf", "force_wide", "force_global_stack", "wide_lds_stack")  # read when a scene is created: the scene is rebuilt per value
world = rt.BVH(spheres, materials, builder=a.builder)
cam = cfg.camera()
frame = torch.zeros((cam.image_height, a.width, 3), dtype=torch.float64, device="cuda")
best = {}
for r in range(a.rounds):  # interleaved rounds in one process
    for v in a.values:
        rt._lib.check(L.rt_debug_set(a.name.encode(), C.c_double(v)))
        if a.name in SCENE_KNOBS:
            world.close()
            world = rt.BVH(spheres, materials, builder=a.builder)
        frame.zero_()
        torch.cuda.synchronize()
        st = cam.render_device(world, frame.data_ptr(), seed=12345, spp_begin=0, spp_end=a.spp)
        g = st["rays"] / st["kernel_ms"] / 1e6
        best.setdefault(v, []).append(g)
for v in a.values:
    xs = sorted(best[v])
    print("%s=%g: Grays/s median %.2f max %.2f  (steps/ray %.2f, tests/ray %.2f)" % (a.name, v, xs[len(xs) // 2], xs[-1], st["node_steps"] / st["rays"], st["sphere_tests"] / st["rays"]))

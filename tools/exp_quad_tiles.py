#!/usr/bin/env python3
"""Quarter-tile unit mapping (debug knob quad_tiles) against the 8x8 mapping: same frame, time per configuration."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import raytracer_amd as rt

L = rt._lib.lib()
spheres, materials = rt.scene_rtiow(1)
world = rt.BVH(spheres, materials)
for width, spp in ((3840, 256), (830, 750), (1920, 512)):
    cfg = rt.config_rtiow(width, spp, 50)
    cam = cfg.camera()
    frames = {}
    for rounds in range(3):
        for v in (0.0, 1.0):
            rt._lib.check(L.rt_debug_set(b"quad_tiles", C.c_double(v)))
            frame = torch.zeros((cam.image_height, width, 3), dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            st = cam.render_device(world, frame.data_ptr(), seed=12345, spp_begin=0, spp_end=spp)
            frames[v] = frame
            print("%dx%d %d spp quad_tiles=%d: %.2f ms, %.2f Grays/s" % (width, cam.image_height, spp, v, st["kernel_ms"], st["rays"] / st["kernel_ms"] / 1e6), flush=True)
    print("frames equal:", bool(torch.equal(frames[0.0], frames[1.0])), flush=True)
rt._lib.check(L.rt_debug_set(b"quad_tiles", C.c_double(1.0)))

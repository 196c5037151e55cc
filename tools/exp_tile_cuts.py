#!/usr/bin/env python3
"""Render rate of each configuration with the tiles' entry lists on and off (one GPU):  python tools/exp_tile_cuts.py [--only c1,c2] [--scale 0.25]"""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.run_configs import CONFIGS


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="c1,c2,c5,c3,c4")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--ks", default="0,2,3,4")
    args = ap.parse_args()
    import torch

    import raytracer_amd as rt

    L = rt._lib.lib()
    for name in args.only.split(","):
        n, pl, pm, fz, width, spp, depth, seed = CONFIGS[name]
        spp = max(1, int(spp * args.scale))
        if n is None:
            spheres, materials = rt.scene_rtiow(seed)
            cfg = rt.config_rtiow(width, spp, depth)
        else:
            spheres, materials = rt.scene_field(seed, n, pl, pm, fz)
            cfg = rt.config_field(n, width, spp, depth)
        world = rt.BVH(spheres, materials)
        cam = cfg.camera()
        frame = torch.zeros((cam.image_height, width, 3), dtype=torch.float64, device="cuda")
        cam.render_device(world, frame.data_ptr(), seed=1, spp_begin=0, spp_end=1)
        sums = {}
        for k in [int(x) for x in args.ks.split(",")]:
            rt._lib.check(L.rt_debug_set(b"tile_cuts", C.c_double(k)))
            frame.zero_()
            torch.cuda.synchronize()
            st = cam.render_device(world, frame.data_ptr(), seed=12345)
            torch.cuda.synchronize()
            sums[k] = float(frame.sum().item())
            print(json.dumps({"config": name, "tile_cuts": k, "Mrays_per_s": st["rays"] / st["kernel_ms"] / 1e3, "render_ms": st["kernel_ms"], "rays": st["rays"],
                              "node_steps": st["node_steps"], "same_sum_as_first": sums[k] == list(sums.values())[0]}), flush=True)
        rt._lib.check(L.rt_debug_set(b"tile_cuts", C.c_double(8)))
        world.close()
        del frame


if __name__ == "__main__":
    main()

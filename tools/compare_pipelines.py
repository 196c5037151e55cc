#!/usr/bin/env python3
"""Persistent kernel against the wavefront harness on the benchmark's frame (3840 x 2160, reference scene) at a
sample count both finish in seconds.  Prints one JSON line per pipeline; run under rocprofv3 for the counters.

  python tools/compare_pipelines.py [--spp 16] [--only persistent|wavefront]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the harness lives in the experiments library, not in the shipped one
os.environ.setdefault("RAYTRACER_AMD_LIB", os.path.join(ROOT, "raytracer_amd", "libraytracer_amd_experiments.so"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    import torch

    import raytracer_amd as rt

    world = rt.BVH(*rt.scene_rtiow(1))
    cam = rt.config_rtiow(args.width, args.spp, 50).camera()
    frame = torch.zeros((cam.image_height, args.width, 3), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for name, fn in (("persistent", cam.render_device), ("wavefront", cam.render_wavefront_device)):
        if args.only and args.only != name:
            continue
        fn(world, frame.data_ptr(), seed=1, spp_begin=0, spp_end=1)  # warm
        frame.zero_()
        torch.cuda.synchronize()
        st = fn(world, frame.data_ptr(), seed=12345)
        torch.cuda.synchronize()
        print(json.dumps({"pipeline": name, "width": args.width, "spp": args.spp, "rays": st["rays"], "ms": st["kernel_ms"], "launches": st["launches"],
                          "Mrays_per_s": st["rays"] / st["kernel_ms"] / 1e3, "mean_radiance": float(frame.mean().item()) / args.spp}), flush=True)
    world.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Renders the stress configurations on one GPU and prints one JSON line each:
scene build time, render time, rays, Mrays/s.  Sample counts are reduced with
--scale so that a run takes seconds; throughput does not depend on them.

  python tools/run_configs.py [--only c3,c5] [--scale 1.0]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CONFIGS = {
    # name: (spheres, p_lambertian, p_metal, max_fuzz, width, spp, depth, scene seed)
    "c1": (None, 0, 0, 0, 830, 750, 50, 1),
    "c2": (None, 0, 0, 0, 3840, 1000, 50, 1),
    "c3": (100000, 1 / 3, 1 / 3, 1.0, 3840, 64, 50, 3),
    "c4": (1000000, 0.15, 0.15, 1.0, 7680, 32, 50, 4),
    "c5": (10000, 0.0, 0.5, 0.3, 3840, 128, 200, 5),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="c1,c3,c4,c5")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--builder", choices=["host", "device", "device-sah"], default="host")  # device: the clustering builder
    ap.add_argument("--rooflines", action="store_true",
                    help="also run the CPU oracle on a 160-pixel-wide, 2-spp sample of each configuration for the reference's node visits and sphere "
                         "tests per ray, and price the measured ray rate against the vector-ALU and HBM ceilings")
    args = ap.parse_args()
    import torch

    import raytracer_amd as rt

    builder = {"host": rt.BUILD_HOST_SAH, "device": rt.BUILD_DEVICE_LBVH, "device-sah": rt.BUILD_DEVICE_SAH}[args.builder]
    if args.builder != "host":  # the first device build of a process pays one-time set-up (code object load, sort temporaries)
        ws, wm = rt.scene_field(1, 1000, 1 / 3, 1 / 3, 1.0)
        rt.BVH(ws, wm, builder=builder).close()
    for name in args.only.split(","):
        n, pl, pm, fz, width, spp, depth, seed = CONFIGS[name]
        spp = max(1, int(spp * args.scale))
        t0 = time.perf_counter()
        if n is None:
            spheres, materials = rt.scene_rtiow(seed)
            cfg = rt.config_rtiow(width, spp, depth)
        else:
            spheres, materials = rt.scene_field(seed, n, pl, pm, fz)
            cfg = rt.config_field(n, width, spp, depth)
        t1 = time.perf_counter()
        world = rt.BVH(spheres, materials, builder=builder)
        t2 = time.perf_counter()
        cam = cfg.camera()
        frame = torch.zeros((cam.image_height, width, 3), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        cam.render_device(world, frame.data_ptr(), seed=1, spp_begin=0, spp_end=1)  # warm: code object, clocks
        frame.zero_()
        torch.cuda.synchronize()
        st = cam.render_device(world, frame.data_ptr(), seed=12345)
        torch.cuda.synchronize()
        info = world.info()
        extra = {}
        if args.rooflines:
            from oracle import oracle as orc

            oscene = orc.Scene(spheres.view(orc.SPHERE_DTYPE), materials.view(orc.MATERIAL_DTYPE))
            small = cfg.image_width(160).samples_per_pixel(2)
            _, ost = oscene.render(small.record().view(orc.CONFIG_DTYPE), 12345, 0, 2, nthreads=min(16, os.cpu_count() or 1))
            o_rays, o_visits, o_tests = (float(x) for x in ost)
            v_bar, p_bar = o_visits / o_rays, o_tests / o_rays
            rate = st["rays"] / (st["kernel_ms"] * 1e-3)
            flops = 25.0 * v_bar + 28.0 * p_bar + 60.0  # per ray: slab tests, sphere tests, hit record + scatter + spawn
            nbytes = 32.0 * v_bar + 32.0 * p_bar + 32.0  # per ray: 32 B per node visit (a 64-byte node holds two), f64 sphere, material
            extra = {
                "oracle_node_visits_per_ray": v_bar, "oracle_sphere_tests_per_ray": p_bar, "flops_per_ray": flops, "bytes_per_ray": nbytes,
                "achieved_TFLOPs": rate * flops / 1e12, "frac_of_fp32_vector_peak_157.3": rate * flops / 1e12 / 157.3,
                "achieved_algorithmic_GBs": rate * nbytes / 1e9, "frac_of_hbm_8000": rate * nbytes / 1e9 / 8000.0,
                "binding": "vector ALU (tree in LDS)" if info["n_nodes_in_lds"] == info["n_device_nodes"] else "dependent node fetches out of L1/L2 (latency and load issue); the algorithmic bytes exceed what HBM could deliver only if they came from HBM, and they do not",
            }
        print(json.dumps({
            **extra,
            "config": name, "spheres": len(spheres), "width": width, "height": cam.image_height, "spp": spp, "depth": depth,
            "scene_gen_s": t1 - t0, "bvh_build_upload_s": t2 - t1, "builder": args.builder, "depth_levels": info["depth"], "build_ms": info["build_ms"], "device_nodes": info["n_device_nodes"],
            "nodes_in_lds": info["n_nodes_in_lds"], "device_bytes": info["device_bytes"], "render_ms": st["kernel_ms"], "launches": st["launches"],
            "rays": st["rays"], "mean_path_length": st["rays"] / float(st["samples"]), "Mrays_per_s": st["rays"] / st["kernel_ms"] / 1e3,
            "mean_radiance": float(frame.mean().item()) / spp,
        }), flush=True)
        world.close()
        del frame


if __name__ == "__main__":
    main()

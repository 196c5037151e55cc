"""Run by tests/test_gpu_wavefront.py in a child process that loads libraytracer_amd_experiments.so:
python tests/wavefront_check.py rtiow|field"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

import raytracer_amd as rt


def main(case):
    if case == "rtiow":
        world = rt.BVH(*rt.scene_rtiow(1))
        cfg = rt.config_rtiow(image_width=160, samples_per_pixel=12, max_depth=50)
    else:  # a scene that is walked through the four-wide tree
        world = rt.BVH(*rt.scene_field(3, 20000, 1.0 / 3.0, 1.0 / 3.0))
        cfg = rt.config_field(20000, 160, 6, 50)
    cam = cfg.camera()
    spp = cfg.get("samples_per_pixel")
    a = torch.zeros((cam.image_height, 160, 3), dtype=torch.float64, device="cuda")
    b = torch.zeros_like(a)
    torch.cuda.synchronize()
    st = cam.render_device(world, a.data_ptr(), seed=5)
    sw = cam.render_wavefront_device(world, b.data_ptr(), seed=5)
    torch.cuda.synchronize()
    assert sw["samples"] == st["samples"] == 160 * cam.image_height * spp
    assert abs(sw["rays"] - st["rays"]) <= 1e-3 * st["rays"]

    def agreement():
        diff = (a - b).abs().amax(dim=2)
        scale = torch.clamp(a.abs().amax(dim=2), min=1.0)
        return float((diff <= 1e-9 * scale).double().mean().item())

    first = agreement()
    assert first >= 0.99, first
    # a sample range on top of what the frame holds, with fewer paths in flight than the range needs at once
    rt._lib.check(rt._lib.lib().rt_debug_set(b"wavefront_paths", C.c_double(2 * 160 * cam.image_height)))
    cam.render_device(world, a.data_ptr(), seed=5, spp_begin=3, spp_end=8)
    cam.render_wavefront_device(world, b.data_ptr(), seed=5, spp_begin=3, spp_end=8)
    torch.cuda.synchronize()
    second = agreement()
    assert second >= 0.99, second
    world.close()
    print("wavefront harness %s: %.2f%% and %.2f%% of pixels equal to 1e-9, rays %d vs %d" % (case, 100 * first, 100 * second, sw["rays"], st["rays"]))


if __name__ == "__main__":
    main(sys.argv[1])

"""The three stress configurations at their stated sizes, through the C ABI.

  C3  100 000 spheres, mixed materials, 3840 x 2160, depth 50
  C4  1 000 000 spheres, 70 % glass, 7680 x 4320, depth 50
  C5  10 000 glass/metal spheres (fuzz < 0.3), 3840 x 2160, depth 200

For each: the scene at full sphere count against oracle results committed as fixtures (closest hits of 4096
rays to the bars of test_gpu_parity, a 160 x 90 render pixel by pixel), then one frame at the stated
resolution, kept on the device, checked through properties that need no oracle: it is the sum of its tiles and
sample ranges (same paths, same ray count), finite, non-negative, and its mean path length and radiance are
those of the oracle's small render.  C5 is also traced live against the oracle (32 768 rays), whose tree for
10 000 spheres builds in a second.
"""
import os

import numpy as np
import pytest

import raytracer_amd as rt
from test_gpu_parity import _pixel_agreement, _secondary_rays, compare_hits

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    return np.load(os.path.join(GOLDEN, "config_%s.npz" % name))


def _scene(G):
    spheres, materials = rt.scene_field(int(G["scene_seed"]), int(G["n_small"]), float(G["p_lambertian"]), float(G["p_metal"]),
                                        float(G["max_fuzz"]))
    return spheres, materials, rt.BVH(spheres, materials)


def _check_against_fixture(name, G, world, bar=0.995):
    n = int(G["n_small"])
    got = world.trace_batch(G["org"], G["dir"])
    want = dict(prim=G["prim"], t=G["t"], normal=G["normal"], front=G["front"])
    ties, rel_t, dn = compare_hits(got, want)
    assert ties <= 4
    small = rt.config_field(n, int(G["small_width"]), int(G["small_spp"]), int(G["depth"]))
    img, st = small.camera().render_sum(world, seed=int(G["render_seed"]))
    frac, worst = _pixel_agreement(img, G["rgb_sum"])
    o_rays = int(G["render_stats"][0])
    print("%s: %d near-tie index differences on %d rays (rel dt %.1e); small render %.2f%% of pixels equal to 1e-9, rays %d vs oracle %d"
          % (name, ties, len(G["prim"]), rel_t, 100 * frac, st["rays"], o_rays))
    assert frac >= bar
    assert abs(st["rays"] - o_rays) <= 0.005 * o_rays
    return o_rays / float(int(G["small_width"]) * img.shape[0] * int(G["small_spp"])), G["rgb_sum"].mean() / int(G["small_spp"])


def _full_frame_properties(name, G, world, spp, split):
    """One frame at the configuration's resolution on the device; returns (mean path length, mean radiance)."""
    import torch

    n, width = int(G["n_small"]), int(G["width"])
    cam = rt.config_field(n, width, spp, int(G["depth"])).camera()
    h = cam.image_height
    whole = torch.zeros((h, width, 3), dtype=torch.float64, device="cuda")
    parts = torch.zeros_like(whole)
    torch.cuda.synchronize()
    st = cam.render_device(world, whole.data_ptr(), seed=12345, spp_begin=0, spp_end=spp)
    assert st["samples"] == width * h * spp
    rays = 0
    xm, ym = width // 2 + 3, h // 3 + 5  # tile edges off the 8-pixel grid
    for s0, s1 in split:
        for tile in ((0, 0, xm, h), (xm, 0, width, ym), (xm, ym, width, h)):
            rays += cam.render_device(world, parts.data_ptr(), seed=12345, spp_begin=s0, spp_end=s1, tile=tile)["rays"]
    torch.cuda.synchronize()
    assert rays == st["rays"]
    scale = float(whole.abs().max().item())
    assert float((parts - whole).abs().max().item()) <= 1e-12 * scale
    if len(split) == 1:
        assert torch.equal(parts, whole)  # tiles alone reproduce the frame bit for bit
    assert bool(torch.isfinite(whole).all().item()) and float(whole.min().item()) >= 0.0
    path, mean = st["rays"] / float(st["samples"]), float(whole.mean().item()) / spp
    print("%s: %d x %d x %d spp on the device: %d rays in %.1f ms (%.0f Mrays/s), path length %.3f, mean radiance %.4f"
          % (name, width, h, spp, st["rays"], st["kernel_ms"], st["rays"] / st["kernel_ms"] / 1e3, path, mean))
    return path, mean


@pytest.mark.parametrize("name,spp,split", [("c3", 2, ((0, 1), (1, 2))), ("c4", 1, ((0, 1),)), ("c5", 3, ((0, 1), (1, 3)))])
def test_stress_configuration_at_stated_size(name, spp, split):
    G = _load(name)
    spheres, materials, world = _scene(G)
    info = world.info()
    assert info["n_spheres"] == int(G["n_small"]) + 1 and info["n_nodes_in_lds"] < info["n_device_nodes"]
    # (10^6 spheres spread over 1000 units: the last-bit differences of contracted multiply-adds are larger in absolute terms and more
    # paths part ways -- observed 98.6 % there, 99.7 - 99.9 % on the other two)
    o_path, o_mean = _check_against_fixture(name, G, world, bar=0.98 if name == "c4" else 0.995)
    path, mean = _full_frame_properties(name, G, world, spp, split)
    # the same scene through the same camera at 1/24 (1/48) of the resolution: path length and radiance agree statistically
    assert abs(path - o_path) <= 0.03 * o_path
    assert abs(mean - o_mean) <= 0.03 * o_mean
    world.close()


def test_deep_bounce_configuration_live_against_oracle(oracle):
    # C5 as stated: 10 000 glass/metal spheres, depth 200 -- 32 768 rays and a small depth-200 render against the oracle
    G = _load("c5")
    spheres, materials, world = _scene(G)
    oworld = oracle.Scene(*oracle.scene_field(5, 10000, 0.0, 0.5, 0.3))
    assert oworld.spheres.tobytes() == spheres.tobytes()
    cfg = rt.config_field(10000, 3840, 128, 200)
    rng = np.random.default_rng(55)
    n = 1 << 14
    ijs = np.stack([rng.integers(0, 3840, n), rng.integers(0, 2160, n), rng.integers(0, 128, n)], axis=1)
    org, d = oracle.primary_rays(cfg.record().view(oracle.CONFIG_DTYPE), 12345, ijs)
    org2, d2 = _secondary_rays(spheres[1:], rng, n)
    ties, rel_t, dn = compare_hits(world.trace_batch(org, d), oworld.trace_batch(org, d))
    ties2, rel_t2, dn2 = compare_hits(world.trace_batch(org2, d2), oworld.trace_batch(org2, d2))
    print("c5 live: near-tie index differences %d + %d on 2 x %d rays, rel dt %.1e %.1e" % (ties, ties2, n, rel_t, rel_t2))
    assert ties + ties2 <= 8
    small = cfg.image_width(128).samples_per_pixel(6)
    got, st = small.camera().render_sum(world, seed=9)
    want, ost = oworld.render(small.record().view(oracle.CONFIG_DTYPE), 9, 0, 6, nthreads=8)
    frac, worst = _pixel_agreement(got, want)
    print("c5 live render 128 x 72 x 6 at depth 200: %.2f%% of pixels equal to 1e-9, path length %.2f" % (100 * frac, st["rays"] / st["samples"]))
    assert frac >= 0.995 and abs(st["rays"] - int(ost[0])) <= 0.005 * int(ost[0])  # (observed: 99.76 %)
    world.close()

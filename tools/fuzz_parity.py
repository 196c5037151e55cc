#!/usr/bin/env python3
"""Random scenes, cameras, tiles and sample ranges: the GPU path against the CPU oracle.

Not part of the test suite (it runs for minutes); prints one line per case and
exits non-zero at the first disagreement beyond the bars of tests/test_gpu_parity.py.

  python tools/fuzz_parity.py [--cases 60] [--seed 1]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: F401  (first, so that the library shares its ROCm runtime)

import raytracer_amd as rt
from oracle import oracle as orc

REL = 1e-5


def random_scene(rng):
    kind = rng.integers(0, 5)
    n = int(rng.choice([1, 2, 3, 7, 40, 300, 2000, 6000]))
    if kind == 0:  # the library's field generator
        n = max(n, 4)
        return rt.scene_field(int(rng.integers(1, 1 << 30)), n, float(rng.uniform(0, 0.6)), float(rng.uniform(0, 0.4)), float(rng.uniform(0, 1))) + (False,)
    spheres = np.zeros(n, dtype=rt.SPHERE_DTYPE)
    materials = np.zeros(max(1, min(n, int(rng.integers(1, 9)))), dtype=rt.MATERIAL_DTYPE)
    if kind == 1:  # a cloud
        spheres["center"] = rng.normal(size=(n, 3)) * rng.uniform(0.5, 30)
        spheres["radius"] = rng.uniform(0.01, 1.5, n)
    elif kind == 2:  # clusters far apart, tiny and huge radii, overlaps
        centres = rng.normal(size=(4, 3)) * 1e3
        spheres["center"] = centres[rng.integers(0, 4, n)] + rng.normal(size=(n, 3)) * 2
        spheres["radius"] = np.exp(rng.uniform(np.log(1e-3), np.log(20), n))
    elif kind == 3:  # collinear centres, a few coincident, some negative radii (inside-out shells, src/sphere.rs:29-48 keeps them)
        t = rng.uniform(-20, 20, n)
        spheres["center"] = np.outer(t, rng.normal(size=3))
        spheres["center"][: n // 3] = spheres["center"][0]
        spheres["radius"] = rng.uniform(0.05, 2.0, n) * np.where(rng.random(n) < 0.2, -1.0, 1.0)
    else:  # nested shells around one point
        spheres["center"] = rng.normal(size=3) + rng.normal(size=(n, 3)) * 1e-3
        spheres["radius"] = np.sort(rng.uniform(0.1, 10, n))
    spheres["material"] = rng.integers(0, len(materials), n)
    materials["type"] = rng.integers(0, 3, len(materials))
    materials["albedo"] = rng.uniform(0, 1, (len(materials), 3))
    materials["param"] = np.where(materials["type"] == 2, rng.uniform(1.1, 2.4, len(materials)), rng.uniform(0, 1, len(materials)))
    # thousands of boxes inside each other: no tree helps, every ray tests most of the spheres (the oracle needs minutes for a frame)
    return spheres, materials, kind >= 3 and n >= 2000


def random_config(rng, spheres):
    c = spheres["center"].mean(axis=0)
    span = max(1.0, float(np.abs(spheres["center"] - c).max() + np.abs(spheres["radius"]).max()))
    frm = c + rng.normal(size=3) * span * rng.uniform(0.3, 3.0)
    w = int(rng.choice([1, 7, 8, 9, 33, 64, 100]))
    cfg = (rt.Config().aspect_ratio(float(rng.choice([1.0, 16 / 9, 0.5, 3.0]))).image_width(w).vfov(float(rng.uniform(5, 120)))
           .lookfrom(tuple(frm)).lookat(tuple(c + rng.normal(size=3) * 0.1 * span)).vup(tuple(rng.normal(size=3)))
           .defocus_angle(float(rng.choice([0.0, 0.5, 5.0]))).focus_dist(float(rng.uniform(0.5, 2) * span))
           .samples_per_pixel(int(rng.choice([1, 2, 5, 16, 70]))).max_depth(int(rng.choice([1, 2, 7, 50, 200]))))
    return cfg


def check_hits(got, want):
    gp, wp = got["prim"], want["prim"]
    differ = gp != wp
    if differ.any():
        both = (gp >= 0) & (wp >= 0)
        assert (both | ~differ).all(), "hit/miss disagreement"
        dt = np.abs(got["t"][differ] - want["t"][differ]) / np.abs(want["t"][differ])
        assert (dt <= REL).all(), "index mismatch that is not a near-tie: %g" % dt.max()
    hit = (wp >= 0) & ~differ
    rel = np.abs(got["t"][hit] - want["t"][hit]) / np.abs(want["t"][hit])
    assert rel.size == 0 or rel.max() <= REL, rel.max()
    assert np.array_equal(got["front"][hit], want["front"][hit])
    dn = np.abs(got["normal"][hit] - want["normal"][hit])
    assert dn.size == 0 or dn.max() <= REL, dn.max()
    return int(differ.sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--probe", type=int, default=-1, help="case number: print how far (in bounces) its render agrees with the oracle, skip the other cases' checks")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    t_start = time.time()
    for case in range(a.cases):
        spheres, materials, heavy = random_scene(rng)
        builder = (rt.BUILD_HOST_SAH, rt.BUILD_DEVICE_LBVH, rt.BUILD_DEVICE_SAH)[int(rng.integers(0, 3))]
        world = rt.BVH(spheres, materials, builder=builder)
        oworld = orc.Scene(spheres.view(orc.SPHERE_DTYPE), materials.view(orc.MATERIAL_DTYPE))
        cfg = random_config(rng, spheres)
        cam = cfg.camera()
        w, h = cfg.get("image_width"), cam.image_height
        spp, depth = int(cfg.get("samples_per_pixel")), int(cfg.get("max_depth"))
        # closest hits: rays from the camera position and from sphere surfaces
        m = 3000
        k = rng.integers(0, len(spheres), m)
        u = rng.normal(size=(m, 3))
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        org = spheres["center"][k] + np.abs(spheres["radius"][k, None]) * u * rng.choice([1.0, 1.0, 3.0], (m, 1))
        d = rng.normal(size=(m, 3))
        ties = check_hits(world.trace_batch(org, d), oworld.trace_batch(org, d))
        # a render of a random rectangle and sample range
        x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
        tile = (x0, y0, int(rng.integers(x0 + 1, w + 1)), int(rng.integers(y0 + 1, h + 1))) if rng.random() < 0.5 else None
        s0 = int(rng.integers(0, spp))
        s1 = int(rng.integers(s0 + 1, spp + 1))
        if heavy:  # a small rectangle, two samples at most
            s1 = min(s1, s0 + 2)
            x0, y0 = (tile[0], tile[1]) if tile else (0, 0)
            tile = (x0, y0, min(w, x0 + 12), min(h, y0 + 12))
        seed = int(rng.integers(0, 1 << 62))
        # Paths are chaotic: after some ten bounces off curved glass a last-bit difference (the device contracts
        # multiply-adds, the oracle does not) sends a path elsewhere.  So the pixel-for-pixel comparison is made with
        # at most 7 bounces, and at the full depth only the frame's mean and the ray count are compared.
        if a.probe >= 0:
            if case == a.probe:
                for dd in list(range(1, 31)) + [50, 100, 200]:
                    c2 = cfg.max_depth(dd)
                    g, gs = c2.camera().render_sum(world, seed=seed, spp_begin=s0, spp_end=s1, tile=tile)
                    o, os_ = oworld.render(c2.record().view(orc.CONFIG_DTYPE), seed, s0, s1, tile=tile, nthreads=8)
                    print("depth %3d: rays %d / %d, worst |difference| %.3e, sums %.6e / %.6e" % (dd, gs["rays"], int(os_[0]), np.abs(g - o).max(), g.sum(), o.sum()))
                return
            world.close()
            continue
        strict = cfg.max_depth(min(depth, 7))
        got, st = strict.camera().render_sum(world, seed=seed, spp_begin=s0, spp_end=s1, tile=tile)
        want, ost = oworld.render(strict.record().view(orc.CONFIG_DTYPE), seed, s0, s1, tile=tile, nthreads=8)
        diff = np.abs(got - want).max(axis=2)
        scale = np.maximum(1.0, np.abs(want).max(axis=2))
        frac = (diff <= 1e-9 * scale).mean()
        # ill-conditioned geometry (shells 1e-3 apart, radii from 1e-3 to 20 around centres at 1e3) turns the last-bit
        # differences into 1e-8: equal paths, slightly different numbers.  A wrong path shows up as a difference of order 1.
        n_off = int((diff > 1e-6 * scale).sum())
        rays_ok = abs(st["rays"] - int(ost[0])) <= n_off * 7 + 0.002 * int(ost[0])
        line = ("case %3d: %5d spheres %s tree, %3dx%-3d spp [%d,%d) depth %3d tile %s: near-ties %d, pixels equal %.4f, rays %d / %d"
                % (case, len(spheres), {rt.BUILD_HOST_SAH: "sah ", rt.BUILD_DEVICE_LBVH: "lbvh", rt.BUILD_DEVICE_SAH: "dsah"}[builder], w, h, s0, s1, depth, tile, ties, frac, st["rays"], int(ost[0])))
        line += " [worst relative difference %.1e, %d pixels beyond 1e-6]" % ((diff / scale).max(), n_off)
        assert np.isfinite(got).all()
        assert n_off <= 0.03 * diff.size + 1 and rays_ok, line
        if depth > 7:
            deep, dst = cam.render_sum(world, seed=seed, spp_begin=s0, spp_end=s1, tile=tile)
            dwant, dost = oworld.render(cfg.record().view(orc.CONFIG_DTYPE), seed, s0, s1, tile=tile, nthreads=8)
            line += "; full depth: mean %.5f / %.5f, rays %d / %d" % (deep.mean(), dwant.mean(), dst["rays"], int(dost[0]))
            assert np.isfinite(deep).all(), line
            n_samples = diff.size * (s1 - s0) if tile is None else (tile[2] - tile[0]) * (tile[3] - tile[1]) * (s1 - s0)
            if n_samples >= 2000:  # a handful of chaotic paths says nothing about a mean
                assert abs(deep.mean() - dwant.mean()) <= 0.05 * abs(dwant.mean()) + 1e-6, line
                assert abs(dst["rays"] - int(dost[0])) <= 0.1 * int(dost[0]) + depth, line
        print(line, flush=True)
        world.close()
    print("fuzz: %d cases agree (%.0f s)" % (a.cases, time.time() - t_start))


if __name__ == "__main__":
    main()

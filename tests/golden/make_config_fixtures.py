#!/usr/bin/env python3
"""Writes tests/golden/config_c{3,4,5}.npz: oracle results on the three stress configurations at their stated
sphere counts (1e5 mixed, 1e6 with 70 % glass, 1e4 glass/metal at depth 200), so that the GPU tests can check the
full-size scenes without building the oracle's tree on the GPU box (78 s for 1e6 spheres).

Per configuration: the oracle's render of the configuration's camera at 160 x 90, 4 samples per pixel; closest
hits of 2048 primary rays of the full-resolution camera and of 2048 rays leaving sphere surfaces.  Scenes come from
the oracle's own generator (oracle.scene_field).  Re-run from the repository root (about three minutes):

    python tests/golden/make_config_fixtures.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import raytracer_amd as rt  # noqa: E402  (host-only helpers: Config defaults and the field camera)
from oracle import oracle as orc  # noqa: E402

# name: (small spheres, p_lambertian, p_metal, max_fuzz, full width, depth, scene seed)
CONFIGS = {
    "c3": (100000, 1.0 / 3.0, 1.0 / 3.0, 1.0, 3840, 50, 3),
    "c4": (1000000, 0.15, 0.15, 1.0, 7680, 50, 4),
    "c5": (10000, 0.0, 0.5, 0.3, 3840, 200, 5),
}
SMALL_W, SMALL_SPP, RENDER_SEED = 160, 4, 12345


def main():
    for name, (n, pl, pm, fz, width, depth, seed) in CONFIGS.items():
        spheres, materials = orc.scene_field(seed, n, pl, pm, fz)
        scene = orc.Scene(spheres, materials)
        full = rt.config_field(n, width, 4, depth)
        small = full.image_width(SMALL_W).samples_per_pixel(SMALL_SPP)
        img, stats = scene.render(small.record().view(orc.CONFIG_DTYPE), RENDER_SEED, 0, SMALL_SPP, nthreads=os.cpu_count() or 1)
        rng = np.random.default_rng(1000 + seed)
        m = 2048
        h = full.camera().image_height
        ijs = np.stack([rng.integers(0, width, m), rng.integers(0, h, m), rng.integers(0, 4, m)], axis=1).astype(np.uint32)
        porg, pdir = orc.primary_rays(full.record().view(orc.CONFIG_DTYPE), RENDER_SEED, ijs)
        k = rng.integers(1, len(spheres), m)
        u = rng.normal(size=(m, 3))
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        sorg = spheres["c"][k] + spheres["r"][k, None] * u
        sdir = rng.normal(size=(m, 3))
        org, d = np.concatenate([porg, sorg]), np.concatenate([pdir, sdir])
        hit = scene.trace_batch(org, d)
        np.savez_compressed(os.path.join(HERE, "config_%s.npz" % name), n_small=n, p_lambertian=pl, p_metal=pm, max_fuzz=fz, width=width,
                            depth=depth, scene_seed=seed, small_width=SMALL_W, small_spp=SMALL_SPP, render_seed=RENDER_SEED,
                            rgb_sum=img, render_stats=stats, ijs=ijs, org=org, dir=d, prim=hit["prim"], t=hit["t"], normal=hit["normal"],
                            front=hit["front"], trace_stats=hit["stats"])
        rays, visits, tests = (int(x) for x in stats)
        print("%s: %d spheres, small render %d rays (path length %.3f, %.1f node visits and %.2f sphere tests per ray), %d of %d rays hit"
              % (name, len(spheres), rays, rays / (SMALL_W * 90.0 * SMALL_SPP), visits / rays, tests / rays, (hit["prim"] >= 0).sum(), 2 * m), flush=True)
        scene.close()


if __name__ == "__main__":
    main()

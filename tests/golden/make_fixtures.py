#!/usr/bin/env python3
"""Writes the golden input/output vectors under tests/golden/ from the CPU oracle.

The reference (a Rust binary) cannot be built or run in this environment, so
these vectors come from the oracle, whose slab test and AABB algebra are pinned
by the reference's own unit-test constants (reference_kats.json) and whose other
functions restate the reference line by line.  Re-run from the repository root:

    python tests/golden/make_fixtures.py

The scene arrays come from the library's seeded generator (host code only; no
GPU is needed to run this script).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import raytracer_amd as rt  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    spheres, materials = rt.scene_rtiow(1)
    scene = orc.Scene(spheres.view(orc.SPHERE_DTYPE), materials.view(orc.MATERIAL_DTYPE))
    cfg = rt.config_rtiow()
    ocfg = cfg.record().view(orc.CONFIG_DTYPE)
    rng = np.random.default_rng(2026)

    # camera: derived fields of the reference's own configuration and of Config::new()
    h, cam = orc.camera_derive(ocfg)
    h0, cam0 = orc.camera_derive(orc.make_config())
    np.savez(os.path.join(HERE, "camera.npz"), rtiow_height=h, default_height=h0,
             **{"rtiow_" + k: v for k, v in cam.items()}, **{"default_" + k: v for k, v in cam0.items()})

    # closest hits: primary rays of the camera and rays leaving sphere surfaces
    n = 2048
    ijs = np.stack([rng.integers(0, 830, n), rng.integers(0, h, n), rng.integers(0, 750, n)], axis=1).astype(np.uint32)
    porg, pdir = orc.primary_rays(ocfg, 12345, ijs)
    k = rng.integers(1, len(spheres), n)
    u = rng.normal(size=(n, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    sorg = spheres["center"][k] + spheres["radius"][k, None] * u
    sdir = rng.normal(size=(n, 3))
    org, d = np.concatenate([porg, sorg]), np.concatenate([pdir, sdir])
    hit = scene.trace_batch(org, d, want_p=True)
    np.savez_compressed(os.path.join(HERE, "hits_rtiow.npz"), ijs=ijs, org=org, dir=d, prim=hit["prim"], t=hit["t"], normal=hit["normal"],
                        front=hit["front"], p=hit["p"], stats=hit["stats"])

    # scatter: 1024 events per material kind with explicit uniforms
    m = 1024
    by_type = [np.nonzero(materials["type"] == t)[0] for t in range(3)]
    mat = np.concatenate([rng.choice(ix, m) for ix in by_type]).astype(np.uint32)
    normal = rng.normal(size=(3 * m, 3))
    normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    in_dir = rng.normal(size=(3 * m, 3))
    flip = (in_dir * normal).sum(axis=1) > 0
    normal[flip] *= -1
    in_dir[::5] -= 0.97 * (in_dir[::5] * normal[::5]).sum(axis=1, keepdims=True) * normal[::5]  # grazing: total internal reflection
    front = rng.random(3 * m) < 0.5
    p = rng.uniform(-10, 10, (3 * m, 3))
    uu = rng.random((3 * m, 2))
    scattered, att, out_dir = orc.scatter_batch(materials.view(orc.MATERIAL_DTYPE), mat, in_dir, p, normal, front, uu)
    np.savez_compressed(os.path.join(HERE, "scatter_rtiow.npz"), material=mat, in_dir=in_dir, p=p, normal=normal, front=front, u=uu,
                        scattered=scattered, attenuation=att, out_dir=out_dir)

    # one small image: 64 x 36, 64 samples per pixel, depth 50, seed 12345 (per-pixel radiance sums)
    small = rt.config_rtiow(image_width=64, samples_per_pixel=64, max_depth=50)
    img, stats = scene.render(small.record().view(orc.CONFIG_DTYPE), 12345, 0, 64, nthreads=os.cpu_count() or 1)
    np.savez_compressed(os.path.join(HERE, "image_64x36_64spp.npz"), rgb_sum=img, stats=stats, ppm_bytes=orc.quantise(img, 64))

    # the same image converged: 4096 samples per pixel with the reference's own samplers (the rejection loops of
    # src/vector.rs:16-47 on a sequential stream), to check the direct-map samplers of the GPU path against
    high = rt.config_rtiow(image_width=64, samples_per_pixel=4096, max_depth=50)
    conv, cstats = scene.render(high.record().view(orc.CONFIG_DTYPE), 99, 0, 4096, mode=orc.SAMPLER_REJECTION, nthreads=os.cpu_count() or 1)
    np.savez_compressed(os.path.join(HERE, "image_64x36_4096spp_reference_samplers.npz"), rgb_mean=conv / 4096.0, stats=cstats)

    # the scene itself, so the fixtures are self-contained
    np.savez_compressed(os.path.join(HERE, "scene_rtiow_seed1.npz"), spheres=spheres, materials=materials)
    print("wrote fixtures to", HERE)


if __name__ == "__main__":
    main()

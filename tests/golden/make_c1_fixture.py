#!/usr/bin/env python3
"""The reference's own configuration (src/main.rs:86-96: 830 x 466, 750 samples per pixel, depth 50) rendered whole by
the CPU oracle on the seeded scene, render seed 12345, and the byte-boundary vector of write_colour (src/camera.rs:85-95).

    python tests/golden/make_c1_fixture.py        (about a minute on eight cores)

image_c1_830x466_750spp.npz: ppm_bytes (466*830 x 3, the whole frame as the PPM holds it), rgb_sum_rows (the f64 sums of
every 8th row), rows, stats (rays, node visits, sphere tests).
quantise_boundaries.npz: sums, spp and the oracle's bytes for sums whose gamma-corrected value sits on, just below and
just above every byte boundary, and for NaN, infinities, negatives and tiny values.
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
    cfg = rt.config_rtiow(830, 750, 50)
    img, stats = scene.render(cfg.record().view(orc.CONFIG_DTYPE), 12345, 0, 750, nthreads=os.cpu_count() or 1)
    rows = np.arange(0, img.shape[0], 8)
    np.savez_compressed(os.path.join(HERE, "image_c1_830x466_750spp.npz"), ppm_bytes=orc.quantise(img, 750), rgb_sum_rows=img[rows], rows=rows,
                        stats=np.asarray(stats))

    spp = 7
    k = np.arange(0, 256, dtype=np.float64)
    edge = (k / 255.0) ** 2 * spp
    up, down = np.nextafter(edge, np.inf), np.nextafter(edge, -np.inf)
    special = np.array([np.nan, np.inf, -np.inf, -1.0, -0.0, 0.0, 1e-320, 1e-300, 1e300, spp * 1.0, spp * (1 - 2 ** -53)])
    sums = np.concatenate([down, edge, up, np.nextafter(up, np.inf), np.nextafter(down, -np.inf), special])
    sums = np.concatenate([sums, np.zeros((-len(sums)) % 3)])
    np.savez_compressed(os.path.join(HERE, "quantise_boundaries.npz"), sums=sums, spp=spp, bytes=orc.quantise(sums.reshape(-1, 3), spp))
    print("wrote", HERE, "rays", stats)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""What eight-wide nodes would save over four-wide ones, counted on the real trees with the workload's own rays (host code only).

The reference's tree (host builder) of a configuration is collapsed into nodes of up to `k` children the way the library
collapses it into four-wide nodes (open the child with the largest box until the node is full), and camera rays with up
to three diffuse bounces off whatever they hit are walked through it as the render kernel walks: the child entered first
next, the other children that were hit set aside in a fixed order, closest hits by the sphere test of src/sphere.rs:29-48.
Counted per ray: nodes fetched (the dependent steps of a walk), child boxes tested, sphere tests.

  python tools/exp_eight_wide.py [c5,c3,c4] [rays]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import raytracer_amd as rt  # noqa: E402  (host code only: generator and builder)
from run_configs import CONFIGS  # noqa: E402

LEAF = 1 << 31


def collapse(lo, hi, kind, soc, leaf_first, k, skip_root_leaf=None):
    """Wide nodes over the pre-order two-wide tree: (child boxes lo/hi [n, k, 3], refs [n, k] with LEAF | sphere or node index or -1)."""
    area = 2.0 * np.prod(np.stack([(hi - lo)[:, [0, 1, 2]], (hi - lo)[:, [1, 2, 0]]]), axis=0).sum(axis=1)
    roots, out_lo, out_hi, out_ref = [0], [], [], []
    head = 0
    while head < len(roots):
        b = roots[head]
        head += 1
        ch = [b + 1, b + int(soc[b])]
        while len(ch) < k:
            inner = [c for c in ch if kind[c] != 3]
            if not inner:
                break
            pick = max(inner, key=lambda c: area[c])
            at = ch.index(pick)
            ch[at] = pick + 1
            ch.append(pick + int(soc[pick]))
        nlo, nhi, ref = np.full((k, 3), np.inf), np.full((k, 3), -np.inf), np.full(k, -1, dtype=np.int64)
        for j, c in enumerate(ch):
            nlo[j], nhi[j] = lo[c], hi[c]
            if kind[c] == 3:
                ref[j] = LEAF | int(leaf_first[c])
            else:
                ref[j] = len(roots)
                roots.append(c)
        out_lo.append(nlo)
        out_hi.append(nhi)
        out_ref.append(ref)
    return np.array(out_lo), np.array(out_hi), np.array(out_ref)


def walk(W, centre, radius, o, d, counts):
    """Closest hit (t, sphere) of the ray through the wide tree W; counts += (nodes fetched, boxes tested, sphere tests)."""
    wlo, whi, wref = W
    inv = 1.0 / np.where(d == 0.0, 1e-300, d)
    t_best, prim = np.inf, -1
    stack = [0]
    while stack:
        r = stack.pop()
        if r & LEAF:
            s = r & (LEAF - 1)
            counts[2] += 1
            oc = o - centre[s]
            hb = oc @ d
            disc = hb * hb - (oc @ oc - radius[s] * radius[s])
            if disc >= 0.0:
                q = np.sqrt(disc)
                for t in (-hb - q, -hb + q):
                    if 1e-3 < t < t_best:
                        t_best, prim = t, s
                        break
            continue
        counts[0] += 1
        ref = wref[r]
        present = ref >= 0
        counts[1] += int(present.sum())
        t1, t2 = (wlo[r] - o) * inv, (whi[r] - o) * inv
        tn = np.maximum(np.fmin(t1, t2).max(axis=1), 1e-3)
        tf = np.minimum(np.fmax(t1, t2).min(axis=1), t_best)
        hit = present & (tn <= tf)
        if not hit.any():
            continue
        first = int(np.argmin(np.where(hit, tn, np.inf)))
        for j in range(len(ref) - 1, -1, -1):  # the others wait in a fixed order, as in the kernel
            if hit[j] and j != first:
                stack.append(int(ref[j]))
        stack.append(int(ref[first]))
    return t_best, prim


def main():
    names = (sys.argv[1] if len(sys.argv) > 1 else "c5,c3,c4").split(",")
    n_paths = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    for name in names:
        n, pl, pm, fz, width, spp, depth, seed = CONFIGS[name]
        spheres, _ = rt.scene_rtiow(1) if n is None else rt.scene_field(seed, n, pl, pm, fz)
        cam = (rt.config_rtiow(width, 1, 50) if n is None else rt.config_field(n, width, 1, 50)).camera()
        boxes, kind, soc, leaf, _ = rt.host_bvh(spheres)
        lo, hi = boxes[:, :3], boxes[:, 3:]
        # leaves of one sphere each in these scenes: soc of a leaf is its count, `leaf` lists the spheres in leaf order
        first = np.zeros(len(kind), dtype=np.int64)
        is_leaf = kind == 3
        assert (soc[is_leaf] == 1).all(), "leaves of several spheres: not modelled"
        first[is_leaf] = leaf[np.cumsum(soc[is_leaf]) - 1]
        centre, radius = spheres["center"], spheres["radius"]
        trees = {k: collapse(lo, hi, kind, soc, first, k) for k in (2, 4, 8)}
        rng = np.random.default_rng(5)
        counts = {k: np.zeros(3) for k in trees}
        rays = 0
        for _ in range(n_paths):
            i, j = rng.integers(0, width), rng.integers(0, cam.image_height)
            o = cam.center.copy()
            d = cam.pixel00_loc + i * cam.pixel_delta_u + j * cam.pixel_delta_v - o
            for bounce in range(4):
                d = d / np.linalg.norm(d)
                res = [walk(trees[k], centre, radius, o, d, counts[k]) for k in trees]
                assert all(r[1] == res[0][1] for r in res), "the closest hit must not depend on the tree"
                rays += 1
                t, s = res[0]
                if s < 0:
                    break
                o = o + t * d
                nrm = (o - centre[s]) / radius[s]
                z, a = 1.0 - 2.0 * rng.random(), 2.0 * np.pi * rng.random()
                q = np.sqrt(max(0.0, 1.0 - z * z))
                d = nrm + np.array([q * np.cos(a), q * np.sin(a), z])
        print("%s: %d spheres, %d rays (camera rays and up to three diffuse bounces)" % (name, len(spheres), rays))
        for k in trees:
            f, b, s = counts[k] / rays
            print("  %d-wide: %7d nodes; per ray %.2f nodes fetched, %.2f child boxes tested, %.2f sphere tests" % (k, len(trees[k][2]), f, b, s))
        f4, f8 = counts[4][0], counts[8][0]
        print("  eight-wide over four-wide: %.2f of the dependent fetches, %.2f of the box tests" % (f8 / f4, counts[8][1] / counts[4][1]), flush=True)


if __name__ == "__main__":
    main()

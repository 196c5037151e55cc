#!/usr/bin/env python3
"""What spatial splits (src/bvh.rs:112-120,158-180, src/sphere.rs:54-74: unfinished in the reference) could save on the
host trees of the stress configurations, bounded from above in numpy.

A ray visits a node with probability ~ surface area of its box (the surface-area heuristic the builder minimises), so the
expected node visits of a long random ray are V = sum over nodes of SA(node) / SA(root).  A spatial split differs from an
object split in that its two children do not overlap; what it can save at a split with children A and B is at most the
double visit of rays through the overlap, SA(A n B) / SA(root); overlaps further down are counted at their own splits.
(The duplicated references a spatial split creates cost sphere tests, which are not counted against it here.)
Prints V, the estimate and their ratio per configuration:  python tools/sbvh_gate.py [c5,c3,c4]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import raytracer_amd as rt  # noqa: E402  (host code only: generator and builder)
from run_configs import CONFIGS  # noqa: E402


def area(lo, hi):
    d = np.maximum(hi - lo, 0.0)
    return 2.0 * (d[..., 0] * d[..., 1] + d[..., 1] * d[..., 2] + d[..., 2] * d[..., 0])


def span(kind, soc, i):
    """Nodes of the subtree rooted at i in the pre-order array (skip = distance to the right child)."""
    j = i
    while kind[j] != 3:  # follow right children to the subtree's end
        j = j + int(soc[j])
    return j - i + 1


def main():
    names = (sys.argv[1] if len(sys.argv) > 1 else "c1,c5,c3,c4").split(",")
    for name in names:
        n, pl, pm, fz, *_ = CONFIGS[name]
        spheres, _ = rt.scene_rtiow(1) if n is None else rt.scene_field(CONFIGS[name][7], n, pl, pm, fz)
        boxes, kind, soc, leaf, depth = rt.host_bvh(spheres)
        lo, hi = boxes[:, :3], boxes[:, 3:]
        inner = np.nonzero(kind != 3)[0]
        left, right = inner + 1, inner + soc[inner].astype(np.int64)
        sa = area(lo, hi)
        whole = sa[0]
        # overlap of the two children of every split
        olo, ohi = np.maximum(lo[left], lo[right]), np.minimum(hi[left], hi[right])
        overlap = np.where((ohi >= olo).all(axis=1), area(olo, ohi), 0.0)
        # splits at or above the scene-sized leaf (a one-sphere leaf with more than half the tree's area: the ground) are left out, as
        # the device's copy of the tree leaves that sphere out: what remains is the tree over the small spheres
        ground = np.nonzero((kind == 3) & (sa > 0.5 * whole))[0]
        keep = np.ones(len(inner), dtype=bool)
        if len(ground):
            g = ground[0]
            above = [i for i in inner[:4096] if i < g < i + span(kind, soc, i)]
            keep[np.isin(inner, above)] = False
        ref = sa[inner][keep].max()  # the tree below the scene-sized leaf
        visits = (sa[left][keep].sum() + sa[right][keep].sum()) / ref
        saved = overlap[keep].sum() / ref
        big = overlap[keep] > 1e-3 * ref
        print("%s: %d spheres, %d splits; expected node visits of a random ray (sum of child areas over the tree's) %.2f; "
              "splits whose children overlap %.1f %% (%d of them by more than 0.1 %% of the tree's area); disjoint children would save "
              "at most %.4f visits = %.2f %%"
              % (name, len(spheres), keep.sum(), visits, 100.0 * (overlap[keep] > 0).mean(), int(big.sum()), saved, 100.0 * saved / visits))


if __name__ == "__main__" and "--rays" not in sys.argv:
    main()


# ---- the same question asked of the workload's own rays ----
# Camera rays of the configuration and two diffuse bounces each are walked through the host tree twice, counting inner-node
# visits: with the children's boxes as they are, and with the boxes of every split cut at the middle of their overlap along
# the split axis (what disjoint children would look like; spheres that straddle the cut would be referenced from both
# sides, whose extra sphere tests are not counted).  Run with --rays:  python tools/sbvh_gate.py --rays c3,c4
def walk_count(T, o, d, clipped):
    inv = 1.0 / np.where(d == 0, 1e-300, d)
    steps = 0
    stack = [0]
    lo, hi, kind, soc = T
    while stack:
        i = stack.pop()
        if kind[i] == 3:
            continue
        steps += 1
        a, b = i + 1, i + int(soc[i])
        la, ha, lb, hb = lo[a].copy(), hi[a].copy(), lo[b].copy(), hi[b].copy()
        if clipped:
            ax = int(kind[i])
            if ha[ax] > lb[ax] and la[ax] <= lb[ax]:  # a below b along the axis, overlapping
                m = 0.5 * (ha[ax] + lb[ax])
                ha[ax] = m
                lb[ax] = m
            elif hb[ax] > la[ax] and lb[ax] <= la[ax]:
                m = 0.5 * (hb[ax] + la[ax])
                hb[ax] = m
                la[ax] = m
        for c, (l, h) in ((a, (la, ha)), (b, (lb, hb))):
            t1, t2 = (l - o) * inv, (h - o) * inv
            if max(np.fmin(t1, t2).max(), 1e-3) <= np.fmax(t1, t2).min():
                stack.append(c)
    return steps


def rays_mode(names, n_pixels=250):
    import math

    rng = np.random.default_rng(3)
    for name in names:
        n, pl, pm, fz, width, spp, depth, seed = CONFIGS[name]
        spheres, _ = rt.scene_rtiow(1) if n is None else rt.scene_field(seed, n, pl, pm, fz)
        cfg = rt.config_rtiow(width, 1, 50) if n is None else rt.config_field(n, width, 1, 50)
        cam = cfg.camera()
        boxes, kind, soc, leaf, _ = rt.host_bvh(spheres)
        T = (boxes[:, :3], boxes[:, 3:], kind, soc)
        centre, radius = spheres["center"], spheres["radius"]
        plain = cut = rays = 0
        for _ in range(n_pixels):
            i, j = rng.integers(0, width), rng.integers(cam.image_height // 3, cam.image_height)
            o = cam.center.copy()
            d = cam.pixel00_loc + i * cam.pixel_delta_u + j * cam.pixel_delta_v - o
            for bounce in range(3):
                d = d / np.linalg.norm(d)
                plain += walk_count(T, o, d, False)
                cut += walk_count(T, o, d, True)
                rays += 1
                # the closest hit by brute force over the spheres near the ray is not needed for a count: bounce off the ground sphere
                oc = o - centre[0]
                hb = oc @ d
                disc = hb * hb - (oc @ oc - radius[0] ** 2)
                if disc < 0:
                    break
                t = -hb - math.sqrt(disc)
                if t < 1e-3:
                    break
                o = o + t * d
                nrm = (o - centre[0]) / radius[0]
                z, a = 1 - 2 * rng.random(), 2 * math.pi * rng.random()
                r = math.sqrt(max(0.0, 1 - z * z))
                d = nrm + np.array([r * math.cos(a), r * math.sin(a), z])
        print("%s: %d rays (camera rays and two diffuse bounces off the ground): %.2f inner-node visits per ray with the tree's boxes, %.2f with "
              "every split's children cut apart at the middle of their overlap: %.2f %% fewer" % (name, rays, plain / rays, cut / rays, 100.0 * (plain - cut) / plain))


if __name__ == "__main__" and "--rays" in sys.argv:
    rays_mode(sys.argv[sys.argv.index("--rays") + 1].split(","))

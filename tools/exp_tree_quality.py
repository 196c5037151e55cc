#!/usr/bin/env python3
"""Would a tree other than the reference's save node steps?  Counted on the workload's own rays (host code only).

The reference's tree (binned surface-area splits, 32 buckets, src/bvh.rs:92-266) against a tree whose every split is the
cheapest of all n - 1 boundaries along each axis (a full sweep over the sorted centroids), both walked two-wide as the
render kernel walks, camera rays and up to three diffuse bounces; closest hits are the same through either.

  python tools/exp_tree_quality.py [c2,c5] [paths]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import raytracer_amd as rt  # noqa: E402  (host code only: generator and builder)
from exp_eight_wide import LEAF, collapse, walk  # noqa: E402
from run_configs import CONFIGS  # noqa: E402


def half_area(lo, hi):
    d = hi - lo
    return d[..., 0] * d[..., 1] + d[..., 1] * d[..., 2] + d[..., 2] * d[..., 0]


def sweep_tree(blo, bhi):
    """Two-wide nodes (child boxes [n, 2, 3] twice, refs [n, 2]) of the full-sweep surface-area tree over the boxes."""
    cen = 0.5 * (blo + bhi)
    out_lo, out_hi, out_ref = [], [], []

    def build(idx):
        if len(idx) == 1:
            return LEAF | int(idx[0])
        best = (np.inf, None, None)
        for ax in range(3):
            order = idx[np.argsort(cen[idx, ax], kind="stable")]
            lo_l, hi_l = np.minimum.accumulate(blo[order]), np.maximum.accumulate(bhi[order])
            lo_r, hi_r = np.minimum.accumulate(blo[order][::-1])[::-1], np.maximum.accumulate(bhi[order][::-1])[::-1]
            n = len(order)
            k = np.arange(1, n)
            cost = half_area(lo_l[:-1], hi_l[:-1]) * k + half_area(lo_r[1:], hi_r[1:]) * (n - k)
            at = int(np.argmin(cost))
            if cost[at] < best[0]:
                best = (cost[at], order, at + 1)
        _, order, cut = best
        me = len(out_ref)
        out_lo.append(None), out_hi.append(None), out_ref.append(None)
        halves = (order[:cut], order[cut:])
        refs = [build(h) for h in halves]
        out_lo[me] = np.array([blo[h].min(axis=0) for h in halves])
        out_hi[me] = np.array([bhi[h].max(axis=0) for h in halves])
        out_ref[me] = np.array(refs, dtype=np.int64)
        return me

    sys.setrecursionlimit(100000)
    build(np.arange(len(blo)))
    return np.array(out_lo), np.array(out_hi), np.array(out_ref)


def main():
    names = (sys.argv[1] if len(sys.argv) > 1 else "c2,c5").split(",")
    n_paths = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    for name in names:
        n, pl, pm, fz, width, spp, depth, seed = CONFIGS[name]
        spheres, _ = rt.scene_rtiow(1) if n is None else rt.scene_field(seed, n, pl, pm, fz)
        cam = (rt.config_rtiow(width, 1, 50) if n is None else rt.config_field(n, width, 1, 50)).camera()
        boxes, kind, soc, leaf, _ = rt.host_bvh(spheres)
        first = np.zeros(len(kind), dtype=np.int64)
        is_leaf = kind == 3
        first[is_leaf] = leaf[np.cumsum(soc[is_leaf]) - 1]
        centre, radius = spheres["center"], spheres["radius"]
        r = np.abs(radius)[:, None]
        trees = {"reference (32 buckets)": collapse(boxes[:, :3], boxes[:, 3:], kind, soc, first, 2), "full sweep": sweep_tree(centre - r, centre + r)}
        rng = np.random.default_rng(5)
        counts = {k: np.zeros(3) for k in trees}
        by_kind = {k: np.zeros(2) for k in trees}  # node steps of camera rays, of scattered rays
        rays = np.zeros(2)
        for _ in range(n_paths):
            i, j = rng.integers(0, width), rng.integers(0, cam.image_height)
            o = cam.center.copy()
            d = cam.pixel00_loc + i * cam.pixel_delta_u + j * cam.pixel_delta_v - o
            for bounce in range(4):
                d = d / np.linalg.norm(d)
                res = []
                for k in trees:
                    before = counts[k][0]
                    res.append(walk(trees[k], centre, radius, o, d, counts[k]))
                    by_kind[k][min(bounce, 1)] += counts[k][0] - before
                assert res[0][1] == res[1][1]
                rays[min(bounce, 1)] += 1
                t, s = res[0]
                if s < 0:
                    break
                o = o + t * d
                nrm = (o - centre[s]) / radius[s]
                z, a = 1.0 - 2.0 * rng.random(), 2.0 * np.pi * rng.random()
                q = np.sqrt(max(0.0, 1.0 - z * z))
                d = nrm + np.array([q * np.cos(a), q * np.sin(a), z])
        print("%s: %d spheres, %d camera rays, %d scattered rays" % (name, len(spheres), rays[0], rays[1]))
        for k in trees:
            print("  %-22s node steps per camera ray %.2f, per scattered ray %.2f, sphere tests per ray %.2f"
                  % (k, by_kind[k][0] / rays[0], by_kind[k][1] / rays[1], counts[k][2] / rays.sum()), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""A CPU model of how full the waves of the render kernel run, and of what a second path per lane would buy.

Part 1 traces real paths of the reference scene (numpy, f64, the device's traversal order: both child boxes
tested at the parent, the nearer one first) and records, per ray, the sequence of blocks it asks for
(I = inner node step, L = leaf test).  Part 2 replays those sequences through the kernel's wave scheduler
(the policy of render_kernel: traversal rounds while inner_min / leaf_min lanes want them, else the block most
lanes wait for) with the measured cycles per block run, and through a variant in which every lane owns two
paths: one in the traversal engine, one parked (waiting to be shaded, or shaded and waiting for the engine).

Nothing here runs on the GPU or is part of the library; it needs the host-only entry points only.

  python tools/sim_lane_fill.py [--tiles 6] [--samples 256]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt  # noqa: E402
from raytracer_amd import api  # noqa: E402

T_MIN = 1e-3


# ------------------------------------------------------------------ part 1: paths and their block sequences ----
class Tree:
    def __init__(self, spheres):
        boxes, kind, soc, leaf, depth = api.host_bvh(spheres)
        n = len(kind)
        self.lo, self.hi = boxes[:, :3].copy(), boxes[:, 3:].copy()
        self.is_leaf = kind == 3  # 0..2: the split axis of an inner node
        self.left = np.arange(n) + 1
        self.right = np.where(self.is_leaf, 0, np.arange(n) + soc.astype(np.int64))  # a split's right child is `skip` entries on
        # leaves take their sphere indices from `leaf` in pre-order
        first = np.zeros(n, dtype=np.int64)
        first[self.is_leaf] = np.cumsum(np.where(self.is_leaf, soc.astype(np.int64), 0))[self.is_leaf] - soc[self.is_leaf].astype(np.int64)
        assert (soc[self.is_leaf] == 1).all(), "one sphere per leaf expected for this scene"
        self.sphere_of = np.where(self.is_leaf, leaf[np.minimum(first, len(leaf) - 1)].astype(np.int64), -1)
        self.depth = depth


def slab(lo, hi, o, inv, t0, t1):
    a, b = (lo - o) * inv, (hi - o) * inv
    tn = np.maximum(np.minimum(a, b).max(axis=1), t0)
    tf = np.minimum(np.maximum(a, b).min(axis=1), t1)
    return tn <= tf, tn


def trace(tree, centers, radii, o, d, max_events=160):
    """Closest hits of n rays; events[n, max_events] holds 1 for an inner step, 2 for a leaf test, 0 after the end."""
    n = len(o)
    with np.errstate(divide="ignore"):
        inv = 1.0 / d
    cur = np.zeros(n, dtype=np.int64)
    stack = np.zeros((n, tree.depth + 2), dtype=np.int64)
    sp = np.zeros(n, dtype=np.int64)
    t_best = np.full(n, np.inf)
    best = np.full(n, -1, dtype=np.int64)
    events = np.zeros((n, max_events), dtype=np.uint8)
    n_ev = np.zeros(n, dtype=np.int64)
    alive = np.ones(n, dtype=bool)
    if tree.is_leaf[0]:
        raise SystemExit("a one-sphere scene has nothing to schedule")

    def pop(idx):
        empty = sp[idx] == 0
        done = idx[empty]
        alive[done] = False
        go = idx[~empty]
        sp[go] -= 1
        cur[go] = stack[go, sp[go]]

    while alive.any():
        idx = np.nonzero(alive)[0]
        at_leaf = tree.is_leaf[cur[idx]]
        li = idx[at_leaf]
        if len(li):
            s = tree.sphere_of[cur[li]]
            oc = o[li] - centers[s]
            half_b = (oc * d[li]).sum(axis=1)
            c = (oc * oc).sum(axis=1) - radii[s] ** 2
            disc = half_b * half_b - c
            ok = disc >= 0
            sq = np.sqrt(np.where(ok, disc, 0.0))
            t = -half_b - sq
            t = np.where(T_MIN < t, t, -half_b + sq)
            hit = ok & (T_MIN < t) & (t < t_best[li])
            t_best[li[hit]] = t[hit]
            best[li[hit]] = s[hit]
            events[li, np.minimum(n_ev[li], max_events - 1)] = 2
            n_ev[li] += 1
            pop(li)
        ii = idx[~at_leaf]
        if len(ii):
            node = cur[ii]
            l, r = tree.left[node], tree.right[node]
            h0, tn0 = slab(tree.lo[l], tree.hi[l], o[ii], inv[ii], T_MIN, t_best[ii])
            h1, tn1 = slab(tree.lo[r], tree.hi[r], o[ii], inv[ii], T_MIN, t_best[ii])
            events[ii, np.minimum(n_ev[ii], max_events - 1)] = 1
            n_ev[ii] += 1
            both = h0 & h1
            swap = tn1 < tn0
            near = np.where(swap, r, l)
            far = np.where(swap, l, r)
            b = ii[both]
            stack[b, sp[b]] = far[both]
            sp[b] += 1
            one = h0 ^ h1
            cur[ii[both]] = near[both]
            cur[ii[one]] = np.where(h0, l, r)[one]
            pop(ii[~(h0 | h1)])
    return best, t_best, events


def unit_vectors(rng, n):
    z = 1.0 - 2.0 * rng.random(n)
    phi = 2.0 * np.pi * rng.random(n)
    r = np.sqrt(np.maximum(0.0, 1.0 - z * z))
    return np.stack([r * np.cos(phi), r * np.sin(phi), z], axis=1)


def scatter(rng, mat_type, albedo, param, d, normal, front):
    n = len(d)
    out = np.zeros((n, 3))
    ok = np.ones(n, dtype=bool)
    lam = mat_type == 0
    out[lam] = normal[lam] + unit_vectors(rng, lam.sum())
    met = mat_type == 1
    refl = d - 2.0 * (d * normal).sum(axis=1, keepdims=True) * normal
    out[met] = refl[met] + param[met, None] * unit_vectors(rng, met.sum())
    ok[met] = (out[met] * normal[met]).sum(axis=1) > 0
    die = mat_type == 2
    if die.any():
        ratio = np.where(front[die], 1.0 / param[die], param[die])
        cos_t = np.minimum(-(d[die] * normal[die]).sum(axis=1), 1.0)
        sin_t = np.sqrt(np.maximum(0.0, 1.0 - cos_t * cos_t))
        r0 = ((1 - ratio) / (1 + ratio)) ** 2
        refl_p = r0 + (1 - r0) * (1 - cos_t) ** 5
        reflect = (ratio * sin_t > 1.0) | (refl_p > rng.random(die.sum()))
        perp = ratio[:, None] * (d[die] + cos_t[:, None] * normal[die])
        par = -np.sqrt(np.abs(1.0 - (perp * perp).sum(axis=1)))[:, None] * normal[die]
        out[die] = np.where(reflect[:, None], refl[die], perp + par)
    out /= np.linalg.norm(out, axis=1, keepdims=True)
    return ok, out


def make_paths(width, tiles, samples, max_depth, seed):
    """Per tile: a list over the 64 pixels of a list over samples of the path's rays, each ray = (events, hit?)."""
    spheres, materials = rt.scene_rtiow(1)
    tree = Tree(spheres)
    centers, radii = spheres["center"].copy(), spheres["radius"].copy()
    mat_of = spheres["material"].astype(np.int64)
    cfg = rt.config_rtiow(width, samples, max_depth)
    c = cfg.camera()
    cam = {k: getattr(c, k) for k in ("image_height", "center", "pixel00_loc", "pixel_delta_u", "pixel_delta_v", "defocus_disk_u", "defocus_disk_v")}
    rng = np.random.default_rng(seed)
    h = int(cam["image_height"])
    out = []
    for _ in range(tiles):
        tx, ty = int(rng.integers(0, width // 8)), int(rng.integers(0, h // 8))
        ii, jj = np.meshgrid(np.arange(8) + 8 * tx, np.arange(8) + 8 * ty)
        pix_i = np.repeat(ii.ravel(), samples).astype(np.float64)
        pix_j = np.repeat(jj.ravel(), samples).astype(np.float64)
        n = len(pix_i)
        px, py = rng.random(n) - 0.5, rng.random(n) - 0.5
        target = cam["pixel00_loc"] + (pix_i + px)[:, None] * cam["pixel_delta_u"] + (pix_j + py)[:, None] * cam["pixel_delta_v"]
        rad, phi = np.sqrt(rng.random(n)), 2 * np.pi * rng.random(n)
        o = cam["center"] + (rad * np.cos(phi))[:, None] * cam["defocus_disk_u"] + (rad * np.sin(phi))[:, None] * cam["defocus_disk_v"]
        d = target - o
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        rays_of_path = [[] for _ in range(n)]
        live = np.arange(n)
        for bounce in range(max_depth):
            if not len(live):
                break
            best, t, ev = trace(tree, centers, radii, o, d)
            hit = best >= 0
            for k, p in enumerate(live):
                e = ev[k]
                rays_of_path[p].append((e[: np.count_nonzero(e)].copy(), bool(hit[k])))
            hp = o[hit] + t[hit, None] * d[hit]
            s = best[hit]
            nrm = (hp - centers[s]) / radii[s, None]
            front = (d[hit] * nrm).sum(axis=1) < 0
            nrm = np.where(front[:, None], nrm, -nrm)
            m = mat_of[s]
            ok, nd = scatter(rng, materials["type"][m], materials["albedo"][m], materials["param"][m], d[hit], nrm, front)
            live = live[hit][ok]
            o, d = hp[ok], nd[ok]
        out.append([[rays_of_path[p * samples + s] for s in range(samples)] for p in range(64)])
    return out


# ------------------------------------------------------------------------- part 2: the wave scheduler ----
COST = dict(camera=3605.0, inner=1042.0, inner_single=600.0, leaf=683.0, shade=3657.0)
# the two-path variant: shading and the camera no longer set up the traversal; the switch block does (ray_box,
# ~36 instructions, + 23 register swaps + 6 LDS operations at 13.5 cycles per vector instruction and wave)
COST2 = dict(camera=3050.0, inner=1042.0, inner_single=600.0, leaf=683.0, shade=3100.0, switch=900.0)


class Slot:
    """One path in flight: walks through the rays of its unit's paths."""
    __slots__ = ("unit", "path", "ray", "pos", "state")

    def __init__(self):
        self.unit = None
        self.state = "camera"  # camera | inner | leaf | shade | sky | ready | idle


class Supply:
    def __init__(self, tile, chunk):
        samples = len(tile[0])
        self.units = [(p, c) for c in range(samples // chunk) for p in range(64)]
        self.tile, self.chunk, self.next = tile, chunk, 0

    def take(self):
        if self.next >= len(self.units):
            return None
        p, c = self.units[self.next]
        self.next += 1
        return [self.tile[p][c * self.chunk + k] for k in range(self.chunk)]


def ray_state(slot):
    ev = slot.unit[slot.path][slot.ray][0]
    if slot.pos >= len(ev):
        return "shade" if slot.unit[slot.path][slot.ray][1] else "sky"
    return "inner" if ev[slot.pos] == 1 else "leaf"


def start_path(slot, supply):
    """The camera block for one slot: the next sample of its unit, or the next unit; False if there is no work left."""
    if slot.unit is not None and slot.path + 1 < len(slot.unit):
        slot.path += 1
    else:
        slot.unit = supply.take()
        if slot.unit is None:
            slot.state = "idle"
            return False
        slot.path = 0
    slot.ray, slot.pos = 0, 0
    return True


def shade(slot):
    """The shade block for one slot: the next ray of the path, or the end of the path (absorbed / depth)."""
    if slot.ray + 1 < len(slot.unit[slot.path]):
        slot.ray += 1
        slot.pos = 0
        return True
    return False


def step_inner(slot, n):
    ev = slot.unit[slot.path][slot.ray][0]
    for _ in range(n):
        if slot.pos < len(ev) and ev[slot.pos] == 1:
            slot.pos += 1
    slot.state = ray_state(slot)


def step_leaf(slot):
    slot.pos += 1
    slot.state = ray_state(slot)


class Tally:
    def __init__(self):
        self.cycles = 0.0
        self.runs = {}
        self.lanes = {}
        self.time = {}
        self.rays = 0

    def run(self, block, lanes, cost):
        self.cycles += cost
        self.runs[block] = self.runs.get(block, 0) + 1
        self.lanes[block] = self.lanes.get(block, 0) + lanes
        self.time[block] = self.time.get(block, 0.0) + cost

    def report(self, title):
        print("%s: %.0f cycles per ray-lane (%d rays)" % (title, 64.0 * self.cycles / max(1, self.rays), self.rays))
        for b in sorted(self.runs):
            print("   %-7s runs %7d  mean lanes %5.1f  share of time %.3f" % (b, self.runs[b], self.lanes[b] / self.runs[b], self.time[b] / self.cycles))


def simulate_one(tile, chunk, inner_min=10, leaf_min=10, both_min=8):
    """The shipped kernel's policy, one path per lane."""
    supply, tally = Supply(tile, chunk), Tally()
    lanes = [Slot() for _ in range(64)]
    block = "camera"

    def count(st):
        return sum(1 for s in lanes if s.state == st)

    while True:
        if block == "shade":
            act = [s for s in lanes if s.state == "shade"]
            tally.run("shade", len(act), COST["shade"])
            for s in act:
                tally.rays += 1
                if shade(s):
                    s.state = ray_state(s)
                else:
                    s.state = "camera"
        if block == "camera":
            act = [s for s in lanes if s.state in ("camera", "sky")]
            tally.run("camera", len(act), COST["camera"])
            for s in act:
                if s.state == "sky":
                    tally.rays += 1
                if start_path(s, supply):
                    s.state = ray_state(s)
        finished = False
        while True:
            while True:
                ran = False
                if count("inner") >= inner_min:
                    act = [s for s in lanes if s.state == "inner"]
                    tally.run("inner", len(act), COST["inner"])
                    for s in act:
                        step_inner(s, 2)
                    ran = True
                if count("leaf") >= leaf_min:
                    act = [s for s in lanes if s.state == "leaf"]
                    tally.run("leaf", len(act), COST["leaf"])
                    for s in act:
                        step_leaf(s)
                    ran = True
                if not ran:
                    break
            n_inner, n_leaf, n_shade = count("inner"), count("leaf"), count("shade")
            n_camera = count("camera") + count("sky")
            if n_inner + n_leaf + n_shade + n_camera == 0:
                finished = True
                break
            pick, most = "inner", n_inner
            if n_leaf > most:
                pick, most = "leaf", n_leaf
            if n_shade > most:
                pick, most = "shade", n_shade
            if n_camera > most:
                pick = "camera"
            if pick in ("shade", "camera"):
                block = pick
                break
            if pick == "inner" or n_inner >= both_min:
                act = [s for s in lanes if s.state == "inner"]
                if act:
                    tally.run("inner", len(act), COST["inner_single"])
                    for s in act:
                        step_inner(s, 1)
            if pick == "leaf" or n_leaf >= both_min:
                act = [s for s in lanes if s.state == "leaf"]
                if act:
                    tally.run("leaf", len(act), COST["leaf"])
                    for s in act:
                        step_leaf(s)
        if finished:
            return tally


def simulate_two(tile, chunk, inner_min=10, leaf_min=10, switch_min=12, shade_min=40, camera_min=40):
    """Two paths per lane: slot a is in the traversal engine, slot b is parked.  The shade and camera blocks work on
    the parked slots only; the switch block swaps a finished (or empty) a with a ready b and sets up its traversal."""
    supply, tally = Supply(tile, chunk), Tally()
    A = [Slot() for _ in range(64)]
    B = [Slot() for _ in range(64)]
    for s in A:
        s.state = "empty"

    def wants_switch(k):
        return A[k].state in ("shade", "sky", "empty", "camera", "idle") and B[k].state == "ready"

    def run_shade():
        act = [s for s in B if s.state == "shade"]
        tally.run("shade", len(act), COST2["shade"])
        for s in act:
            tally.rays += 1
            s.state = "ready" if shade(s) else "camera"

    def run_camera():
        act = [s for s in B if s.state in ("camera", "sky")]
        tally.run("camera", len(act), COST2["camera"])
        for s in act:
            if s.state == "sky":
                tally.rays += 1
            if start_path(s, supply):
                s.state = "ready"

    def run_switch():
        ks = [k for k in range(64) if wants_switch(k)]
        tally.run("switch", len(ks), COST2["switch"])
        for k in ks:
            A[k], B[k] = B[k], A[k]
            A[k].state = ray_state(A[k])
            if B[k].state == "empty":
                B[k].state = "camera"

    while True:
        n_inner = sum(1 for s in A if s.state == "inner")
        n_leaf = sum(1 for s in A if s.state == "leaf")
        n_switch = sum(1 for k in range(64) if wants_switch(k))
        n_shade = sum(1 for s in B if s.state == "shade")
        n_camera = sum(1 for s in B if s.state in ("camera", "sky"))
        # a finished a whose b is not ready cannot move on before b has been shaded: counts as waiting on that block
        if n_inner + n_leaf + n_switch + n_shade + n_camera == 0:
            # what is left: finished rays in a with an idle b -- move them over so that they are shaded
            left = [k for k in range(64) if A[k].state in ("shade", "sky") and B[k].state == "idle"]
            if not left:
                return tally
            for k in left:
                A[k], B[k] = B[k], A[k]
            continue
        ran = False
        if n_switch >= switch_min:
            run_switch()
            ran = True
        if n_inner >= inner_min:
            act = [s for s in A if s.state == "inner"]
            tally.run("inner", len(act), COST2["inner"])
            for s in act:
                step_inner(s, 2)
            ran = True
        n_leaf = sum(1 for s in A if s.state == "leaf")
        if n_leaf >= leaf_min:
            act = [s for s in A if s.state == "leaf"]
            tally.run("leaf", len(act), COST2["leaf"])
            for s in act:
                step_leaf(s)
            ran = True
        if n_shade >= shade_min:
            run_shade()
            ran = True
        if n_camera >= camera_min:
            run_camera()
            ran = True
        if ran:
            continue
        # nothing has its quorum: the block with the most takers, weighted so that the cheap ones go first
        options = dict(inner=n_inner, leaf=n_leaf, switch=n_switch, shade=n_shade, camera=n_camera)
        pick = max(options, key=lambda b: options[b])
        if pick == "inner":
            act = [s for s in A if s.state == "inner"]
            tally.run("inner", len(act), COST2["inner_single"])
            for s in act:
                step_inner(s, 1)
        elif pick == "leaf":
            act = [s for s in A if s.state == "leaf"]
            tally.run("leaf", len(act), COST2["leaf"])
            for s in act:
                step_leaf(s)
        elif pick == "switch":
            run_switch()
        elif pick == "shade":
            run_shade()
        else:
            run_camera()


def simulate_two_greedy(tile, chunk, weights=None, pair_min=24):
    """Two paths per lane, with the simplest policy: always the block with the most takers (times a weight)."""
    w = dict(inner=1.0, leaf=1.0, switch=1.0, shade=1.0, camera=1.0)
    w.update(weights or {})
    supply, tally = Supply(tile, chunk), Tally()
    A = [Slot() for _ in range(64)]
    B = [Slot() for _ in range(64)]
    for s in A:
        s.state = "empty"

    def wants_switch(k):
        return A[k].state in ("shade", "sky", "empty", "camera", "idle") and B[k].state == "ready"

    while True:
        n = dict(
            inner=sum(1 for s in A if s.state == "inner"),
            leaf=sum(1 for s in A if s.state == "leaf"),
            switch=sum(1 for k in range(64) if wants_switch(k)),
            shade=sum(1 for s in B if s.state == "shade"),
            camera=sum(1 for s in B if s.state in ("camera", "sky")),
        )
        if sum(n.values()) == 0:
            left = [k for k in range(64) if A[k].state in ("shade", "sky") and B[k].state == "idle"]
            if not left:
                return tally
            for k in left:
                A[k], B[k] = B[k], A[k]
            continue
        pick = max(n, key=lambda b: n[b] * w[b])
        if pick == "inner":
            act = [s for s in A if s.state == "inner"]
            pair = len(act) >= pair_min
            tally.run("inner", len(act), COST2["inner"] if pair else COST2["inner_single"])
            for s in act:
                step_inner(s, 2 if pair else 1)
        elif pick == "leaf":
            act = [s for s in A if s.state == "leaf"]
            tally.run("leaf", len(act), COST2["leaf"])
            for s in act:
                step_leaf(s)
        elif pick == "switch":
            ks = [k for k in range(64) if wants_switch(k)]
            tally.run("switch", len(ks), COST2["switch"])
            for k in ks:
                A[k], B[k] = B[k], A[k]
                A[k].state = ray_state(A[k])
                if B[k].state == "empty":
                    B[k].state = "camera"
        elif pick == "shade":
            act = [s for s in B if s.state == "shade"]
            tally.run("shade", len(act), COST2["shade"])
            for s in act:
                tally.rays += 1
                s.state = "ready" if shade(s) else "camera"
        else:
            act = [s for s in B if s.state in ("camera", "sky")]
            tally.run("camera", len(act), COST2["camera"])
            for s in act:
                if s.state == "sky":
                    tally.rays += 1
                if start_path(s, supply):
                    s.state = "ready"


def merge(tallies):
    out = Tally()
    for t in tallies:
        out.cycles += t.cycles
        out.rays += t.rays
        for b in t.runs:
            out.runs[b] = out.runs.get(b, 0) + t.runs[b]
            out.lanes[b] = out.lanes.get(b, 0) + t.lanes[b]
            out.time[b] = out.time.get(b, 0.0) + t.time[b]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=6)
    ap.add_argument("--samples", type=int, default=256)
    ap.add_argument("--chunk", type=int, default=64)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--sweep", action="store_true")
    args = ap.parse_args()
    tiles = make_paths(args.width, args.tiles, args.samples, args.depth, 7)
    n_rays = sum(len(path) for t in tiles for px in t for path in px)
    n_paths = args.tiles * 64 * args.samples
    ev = [len(r[0]) for t in tiles for px in t for path in px for r in path]
    inner = sum(int((r[0] == 1).sum()) for t in tiles for px in t for path in px for r in path)
    print("paths %d, rays %d (%.3f per path), inner steps per ray %.2f, leaf tests per ray %.2f" %
          (n_paths, n_rays, n_rays / n_paths, inner / n_rays, (sum(ev) - inner) / n_rays))
    base = merge([simulate_one(t, args.chunk) for t in tiles])
    base.report("one path per lane (the shipped policy)")
    two = merge([simulate_two(t, args.chunk) for t in tiles])
    two.report("two paths per lane")
    print("ratio of cycles per ray: %.3f" % ((two.cycles / two.rays) / (base.cycles / base.rays)))
    for wts in ({}, dict(inner=1.5), dict(inner=1.5, leaf=1.5), dict(shade=0.7, camera=0.7), dict(inner=1.3, leaf=1.6, switch=1.3),
                dict(inner=2.0, leaf=2.0, switch=1.5)):
        g = merge([simulate_two_greedy(t, args.chunk, wts) for t in tiles])
        g.report("two paths per lane, greedy %s" % wts)
        print("ratio of cycles per ray: %.3f" % ((g.cycles / g.rays) / (base.cycles / base.rays)))
    if args.sweep:
        for sm in (24, 32, 40, 48, 56):
            for sw in (4, 8, 12, 16, 24):
                t = merge([simulate_two(tl, args.chunk, switch_min=sw, shade_min=sm, camera_min=sm) for tl in tiles])
                print("shade/camera_min %2d switch_min %2d: ratio %.3f" % (sm, sw, (t.cycles / t.rays) / (base.cycles / base.rays)))


if __name__ == "__main__":
    main()

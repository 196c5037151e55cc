"""Pins the CPU oracle against every known-answer test the reference holds
(src/aabb.rs:89-194, src/sphere.rs:83-117; constants in golden/reference_kats.json)
and against the published Philox4x32 vectors (10 and 7 rounds)."""
import json
import math
import os

import numpy as np

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


def box(orc, spec):
    if spec == "new":
        return orc.aabb_new()
    if "points" in spec:
        return orc.aabb_bounding_box(spec["points"])
    return np.asarray(spec["min"], float), np.asarray(spec["max"], float)


def test_aabb_intersection_empty(oracle):
    for case in G["aabb_intersection_empty"]:
        mn, mx = oracle.aabb_intersection([box(oracle, case["a"]), box(oracle, case["b"])])
        assert oracle.aabb_is_empty(mn, mx)


def test_aabb_intersection(oracle):
    c = G["aabb_intersection"]
    mn, mx = oracle.aabb_intersection([box(oracle, c["a"]), box(oracle, c["b"])])
    emn, emx = box(oracle, c["expect"])
    assert (mn == emn).all() and (mx == emx).all()


def test_aabb_union(oracle):
    c = G["aabb_union"]
    mn, mx = oracle.aabb_union([box(oracle, c["a"]), box(oracle, c["b"])])
    emn, emx = box(oracle, c["expect"])
    assert (mn == emn).all() and (mx == emx).all()


def test_ray_aabb_intersection(oracle):
    c = G["ray_aabb_intersection"]
    mn, mx = box(oracle, c["box"])
    for case in c["cases"]:
        got = oracle.ray_aabb_intersection(mn, mx, c["origin"], c["direction"], *case["interval"])
        assert got == case["some"], case


def test_ray_aabb_intersection_angle(oracle):
    s = math.sqrt(1.0 / 3.0)
    mn, mx = oracle.aabb_bounding_box([[0, 0, 0], [s, s, s]])
    for case in G["ray_aabb_intersection_angle"]["cases"]:
        got = oracle.ray_aabb_intersection(mn, mx, [-s, -s, -s], [1.0, 1.0, 1.0], *case["interval"])
        assert got == case["some"], case


def test_ray_aabb_intersection_failure(oracle):
    c = G["ray_aabb_intersection_failure"]
    mn, mx = box(oracle, c["box"])
    assert oracle.ray_aabb_intersection(mn, mx, c["origin"], c["direction"], c["interval"][0], math.inf) == c["some"]


def test_clipped_aabb(oracle):
    c = G["clipped_aabb"]
    mn, mx = oracle.sphere_aabb(c["center"], c["radius"])
    assert (mn == np.asarray(c["aabb"]["min"], float)).all() and (mx == np.asarray(c["aabb"]["max"], float)).all()
    for case in c["cases"]:
        mn, mx = oracle.sphere_clipped_aabb(c["center"], c["radius"], case["axis"], case["t1"], case["t2"])
        if case.get("empty"):
            assert oracle.aabb_is_empty(mn, mx)
        else:
            assert (mn == np.asarray(case["expect"]["min"], float)).all(), (mn, mx)
            assert (mx == np.asarray(case["expect"]["max"], float)).all(), (mn, mx)


def test_philox_known_answers(oracle):
    # Random123 kat_vectors, philox4x32 with 10 rounds.
    kats = [
        ([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
        ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
        ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0],
         [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
    ]
    for ctr, key, want in kats:
        got = oracle.philox4x32_10(key, ctr)
        assert [int(x) for x in got] == want
        assert [int(x) for x in oracle.philox4x32(key, ctr, 10)] == want
    # The same file's vectors for 7 rounds, which the render loop draws from.
    kats7 = [
        ([0, 0, 0, 0], [0, 0], [0x5F6FB709, 0x0D893F64, 0x4F121F81, 0x4F730A48]),
        ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x5207DDC2, 0x45165E59, 0x4D8EE751, 0x8C52F662]),
        ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0],
         [0x4DFCCABA, 0x190A87F0, 0xC47362BA, 0xB6B5242A]),
    ]
    for ctr, key, want in kats7:
        assert [int(x) for x in oracle.philox4x32(key, ctr, 7)] == want

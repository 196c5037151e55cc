"""The device's binned surface-area build (RT_BUILD_DEVICE_SAH, csrc/rt_build_sah.hip) makes the tree of the host builder,
which makes the reference's (src/bvh.rs:92-266; tests/test_gpu_parity.py compares that one with the oracle's node by node):
a render through either takes the same node steps and sphere tests, ray for ray in sum, and gives the same frame."""
import os

import numpy as np
import pytest

import raytracer_amd as rt

pytestmark = pytest.mark.gpu


def _render(world, cam, seed):
    os.environ["RT_PROFILE_BLOCKS"] = "1"  # the instrumented kernel counts node steps and sphere tests
    try:
        return cam.render_sum(world, seed=seed)
    finally:
        del os.environ["RT_PROFILE_BLOCKS"]


def _same_tree(spheres, materials, cam, capfd):
    host = rt.BVH(spheres, materials, builder=rt.BUILD_HOST_SAH)
    dev = rt.BVH(spheres, materials, builder=rt.BUILD_DEVICE_SAH)
    try:
        hi, di = host.info(), dev.info()
        assert di["builder"] == rt.BUILD_DEVICE_SAH and hi["builder"] == rt.BUILD_HOST_SAH
        a, sa = _render(host, cam, 3)
        b, sb = _render(dev, cam, 3)
        capfd.readouterr()  # (the block profile the instrumented kernel prints)
        assert sa["rays"] == sb["rays"] and sa["rays"] > 0 and np.array_equal(a, b)
        assert hi["depth"] == di["depth"] and hi["n_device_nodes"] == di["n_device_nodes"] and hi["lds_bytes"] == di["lds_bytes"]
        assert sa["node_steps"] == sb["node_steps"] and sa["sphere_tests"] == sb["sphere_tests"], (sa, sb)
        return hi, di
    finally:
        host.close()
        dev.close()


# 3 .. 128 spheres: one wave builds the whole tree; 129 and up: the top of the tree level by level; 488: the reference scene;
# 513 .. 640: the largest scenes that are walked through the two-wide tree in LDS, where the order of a node's children shows
@pytest.mark.parametrize("n", [3, 4, 7, 100, 129, 130, 488, 513, 600, 2049, 10001, 100001])
def test_fields_get_the_host_tree(n, capfd):
    if n == 488:
        spheres, materials = rt.scene_rtiow(1)
        cam = rt.config_rtiow(image_width=160, samples_per_pixel=4, max_depth=20).camera()
    else:
        spheres, materials = rt.scene_field(11, n - 1, 0.3, 0.3, 0.5)
        cam = rt.config_field(n - 1, 160, 4, 20).camera()
    hi, di = _same_tree(spheres, materials, cam, capfd)
    print("%d spheres: host build %.1f ms, device build %.1f ms, depth %d" % (len(spheres), hi["build_ms"], di["build_ms"], di["depth"]))


@pytest.mark.parametrize("n,seed", [(400, 5), (600, 5), (640, 6), (5000, 7)])
def test_clouds_get_the_host_tree(n, seed, capfd):
    # spheres spread through a cube: every level of the tree has a real choice of axis and boundary
    rng = np.random.default_rng(seed)
    sph = np.zeros(n, dtype=rt.SPHERE_DTYPE)
    sph["center"] = rng.uniform(-10.0, 10.0, size=(n, 3))
    sph["radius"] = rng.uniform(0.05, 0.2, size=n)
    mats = np.zeros(1, dtype=rt.MATERIAL_DTYPE)
    mats["albedo"] = ((0.6, 0.5, 0.4),)
    cfg = (rt.Config().aspect_ratio(1.5).image_width(150).vfov(50.0).lookfrom((0.0, 0.0, 25.0)).lookat((0.0, 0.0, 0.0)).vup((0.0, 1.0, 0.0))
           .defocus_angle(0.5).focus_dist(25.0).samples_per_pixel(4).max_depth(30))
    _same_tree(sph, mats, cfg.camera(), capfd)


def test_a_million_spheres_get_the_host_tree_in_a_fraction_of_the_time(capfd):
    spheres, materials = rt.scene_field(4, 1000000, 0.15, 0.15, 1.0)
    hi, di = _same_tree(spheres, materials, rt.config_field(1000000, 256, 2, 20).camera(), capfd)
    print("10^6 spheres: host build %.0f ms, device build %.1f ms" % (hi["build_ms"], di["build_ms"]))
    assert di["build_ms"] < 0.2 * hi["build_ms"]

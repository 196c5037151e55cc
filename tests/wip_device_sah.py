"""The device's binned surface-area build makes the host builder's tree: same node steps and sphere tests per ray, same frame."""
import os
import time

import numpy as np
import pytest

import raytracer_amd as rt

pytestmark = pytest.mark.gpu


def _stats(world, cam, seed):
    os.environ["RT_PROFILE_BLOCKS"] = "1"
    try:
        rgb, st = cam.render_sum(world, seed=seed)
    finally:
        del os.environ["RT_PROFILE_BLOCKS"]
    return rgb, st


@pytest.mark.parametrize("n", [3, 4, 7, 100, 488, 513, 514, 600, 2049, 10001, 100001, 1000001])
def test_same_tree_as_the_host(n, capfd):
    if n == 488:
        spheres, materials = rt.scene_rtiow(1)
        cam = rt.config_rtiow(image_width=160, samples_per_pixel=4, max_depth=20).camera()
    else:
        spheres, materials = rt.scene_field(11, n - 1, 0.3, 0.3, 0.5)
        cam = rt.config_field(n - 1, 160, 4, 20).camera()
    t0 = time.time()
    host = rt.BVH(spheres, materials, builder=rt.BUILD_HOST_SAH)
    t1 = time.time()
    dev = rt.BVH(spheres, materials, builder=rt.BUILD_DEVICE_SAH)
    t2 = time.time()
    try:
        hi, di = host.info(), dev.info()
        assert di["builder"] == rt.BUILD_DEVICE_SAH
        a, sa = _stats(host, cam, 3)
        b, sb = _stats(dev, cam, 3)
        capfd.readouterr()
        print("n", len(spheres), "host build %.1f ms device build %.1f ms (wall %.1f / %.1f); depth %d / %d; nodes %d / %d; steps %d / %d tests %d / %d"
              % (hi["build_ms"], di["build_ms"], 1e3 * (t1 - t0), 1e3 * (t2 - t1), hi["depth"], di["depth"], hi["n_device_nodes"], di["n_device_nodes"],
                 sa["node_steps"], sb["node_steps"], sa["sphere_tests"], sb["sphere_tests"]))
        assert sa["rays"] == sb["rays"]
        assert np.array_equal(a, b)
        assert hi["depth"] == di["depth"] and hi["n_device_nodes"] == di["n_device_nodes"]
        assert sa["node_steps"] == sb["node_steps"] and sa["sphere_tests"] == sb["sphere_tests"]
    finally:
        host.close()
        dev.close()

"""Camera rays that start from their tile's entry list (launch_tile_cuts) find the same closest hits as rays that walk
the tree from the root: frames and ray counts are bit for bit the same with the lists on and off, through the C ABI."""
import ctypes as C

import numpy as np
import pytest

import raytracer_amd as rt

pytestmark = pytest.mark.gpu


def _set(name, value):
    rt._lib.check(rt._lib.lib().rt_debug_set(name, C.c_double(value)))


def _both_ways(world, cam, **kw):
    out = []
    try:
        for k in (0.0, 8.0, 3.0):
            _set(b"tile_cuts", k)
            out.append(cam.render_sum(world, **kw))
    finally:
        _set(b"tile_cuts", 8.0)
    return out


def _check(out):
    (a, sa), rest = out[0], out[1:]
    assert sa["rays"] > 0 and np.isfinite(a).all()
    for b, sb in rest:
        assert sa["rays"] == sb["rays"] and sa["samples"] == sb["samples"]
        assert np.array_equal(a, b)


CAMERAS = {
    # lookfrom, lookat, vfov, defocus angle, focus distance
    "reference": ((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), 20.0, 0.6, 10.0),
    "pinhole": ((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), 20.0, 0.0, 10.0),
    "wide lens": ((6.0, 1.0, 2.0), (0.0, 0.5, 0.0), 60.0, 8.0, 4.0),
    "among the spheres": ((0.3, 0.25, 0.4), (4.0, 1.0, 0.0), 90.0, 2.0, 3.0),
    "from below the horizon": ((9.0, 0.05, 9.0), (0.0, 0.3, 0.0), 35.0, 0.3, 12.0),
    "steeply down": ((0.5, 30.0, 0.5), (0.6, 0.0, 0.4), 40.0, 0.5, 30.0),
    "axis aligned": ((0.0, 0.2, 20.0), (0.0, 0.2, 0.0), 30.0, 0.0, 20.0),
    "looking away": ((13.0, 2.0, 3.0), (26.0, 5.0, 6.0), 40.0, 0.6, 10.0),
}


@pytest.mark.parametrize("name", list(CAMERAS))
def test_reference_scene_same_frame_with_and_without_entry_lists(name):
    spheres, materials = rt.scene_rtiow(1)
    lookfrom, lookat, vfov, defocus, focus = CAMERAS[name]
    cfg = (rt.Config().aspect_ratio(16.0 / 9.0).image_width(136).vfov(vfov).lookfrom(lookfrom).lookat(lookat).vup((0.0, 1.0, 0.0))
           .defocus_angle(defocus).focus_dist(focus).samples_per_pixel(6).max_depth(12))
    world = rt.BVH(spheres, materials)
    try:
        _check(_both_ways(world, cfg.camera(), seed=3))
    finally:
        world.close()


def test_rectangles_and_sample_ranges():
    spheres, materials = rt.scene_rtiow(1)
    cam = rt.config_rtiow(image_width=200, samples_per_pixel=40, max_depth=20).camera()
    world = rt.BVH(spheres, materials)
    try:
        for tile, s0, s1 in (((3, 5, 77, 41), 0, 40), ((64, 16, 200, 112), 7, 9), ((0, 100, 200, 112), 0, 33)):
            _check(_both_ways(world, cam, seed=9, tile=tile, spp_begin=s0, spp_end=s1))
    finally:
        world.close()


@pytest.mark.parametrize("n,builder", [(3000, 0), (3000, 1), (60000, 0), (60000, 1)])
def test_fields_through_the_four_wide_tree(n, builder):
    spheres, materials = rt.scene_field(7, n, 0.3, 0.3, 0.5)
    world = rt.BVH(spheres, materials, builder=builder)
    try:
        assert world.info()["n_nodes_in_lds"] < world.info()["n_device_nodes"] or n <= 3000
        _check(_both_ways(world, rt.config_field(n, 160, 6, 30).camera(), seed=4))
        # a camera low over the field, looking along it: beams that cross many boxes
        side = float(np.ceil(np.sqrt(n)))
        low = (rt.Config().aspect_ratio(2.0).image_width(128).vfov(50.0).lookfrom((0.5 * side, 1.0, 0.1)).lookat((-0.5 * side, 0.0, 0.0))
               .vup((0.0, 1.0, 0.0)).defocus_angle(1.0).focus_dist(0.3 * side).samples_per_pixel(4).max_depth(8))
        _check(_both_ways(world, low.camera(), seed=5))
    finally:
        world.close()


def test_forced_wide_and_global_stacks():
    spheres, materials = rt.scene_rtiow(1)
    cam = rt.config_rtiow(image_width=120, samples_per_pixel=5, max_depth=16).camera()
    for knob in (b"force_wide", b"force_global_stack"):
        _set(knob, 1.0)
        try:
            world = rt.BVH(spheres, materials)
        finally:
            _set(knob, 0.0)
        try:
            _check(_both_ways(world, cam, seed=11))
        finally:
            world.close()


def test_shards_with_entry_lists_match_the_whole_frame():
    import torch

    spheres, materials = rt.scene_rtiow(1)
    cam = rt.config_rtiow(image_width=152, samples_per_pixel=9, max_depth=20).camera()
    world = rt.BVH(spheres, materials)
    try:
        whole, _ = cam.render_sum(world, seed=2)
        frame = torch.zeros((cam.image_height, 152, 3), dtype=torch.float64, device="cuda")
        for shard in cam.partition(3, rt.PARTITION_TILES):
            cam.render_shard_device(world, frame.data_ptr(), shard, seed=2)
        torch.cuda.synchronize()
        assert np.array_equal(frame.cpu().numpy(), whole)
    finally:
        world.close()


def test_tiny_scenes():
    # one and two spheres: the tree is a leaf or a single node
    mats = np.zeros(2, dtype=rt.api.MATERIAL_DTYPE)
    mats["type"] = (0, 1)
    mats["albedo"] = ((0.5, 0.5, 0.5), (0.8, 0.8, 0.8))
    for n in (1, 2):
        sph = np.zeros(n, dtype=rt.api.SPHERE_DTYPE)
        sph["center"][:, 0] = np.arange(n) * 2.5
        sph["radius"] = 1.0
        sph["material"] = np.arange(n) % 2
        world = rt.BVH(sph, mats)
        cfg = (rt.Config().aspect_ratio(1.0).image_width(64).vfov(60.0).lookfrom((1.0, 0.5, 6.0)).lookat((1.0, 0.0, 0.0)).vup((0.0, 1.0, 0.0))
               .defocus_angle(1.0).focus_dist(6.0).samples_per_pixel(4).max_depth(6))
        try:
            _check(_both_ways(world, cfg.camera(), seed=1))
        finally:
            world.close()


def test_entry_lists_save_node_steps(capfd):
    # the lists must not only be harmless but in use: the instrumented kernel counts node steps
    import os

    spheres, materials = rt.scene_rtiow(1)
    cam = rt.config_rtiow(image_width=320, samples_per_pixel=8, max_depth=50).camera()
    world = rt.BVH(spheres, materials)
    os.environ["RT_PROFILE_BLOCKS"] = "1"
    try:
        steps = {}
        for k in (0.0, 8.0):
            _set(b"tile_cuts", k)
            _, st = cam.render_sum(world, seed=3)
            steps[k] = st["node_steps"] / st["rays"]
    finally:
        del os.environ["RT_PROFILE_BLOCKS"]
        _set(b"tile_cuts", 8.0)
        world.close()
    capfd.readouterr()  # (the instrumented kernel prints its block profile)
    assert steps[0.0] > 8.0 and steps[8.0] < 0.8 * steps[0.0], steps

"""One frame shared out over several devices behind the C ABI (rt_scene_create_multi, rt_render_multi,
rt_render_shard_device, rt_render_ppm).  A GPU box with one device still runs every code path: an ordinal may
repeat in the device list, which puts several copies of the scene on the one GPU.
"""
import os
import tempfile

import numpy as np
import pytest

import raytracer_amd as rt

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rtiow():
    return rt.scene_rtiow(1)


@pytest.fixture(scope="module")
def single(rtiow):
    w = rt.BVH(*rtiow)
    yield w
    w.close()


@pytest.fixture(scope="module")
def triple(rtiow):
    w = rt.BVH(*rtiow, devices=[0, 0, 0])
    yield w
    w.close()


def _set(name, value):
    import ctypes as C

    rt._lib.check(rt._lib.lib().rt_debug_set(name, C.c_double(value)))


def test_multi_scene_on_one_device_is_the_single_device_path(rtiow, single):
    one = rt.BVH(*rtiow, devices=[0])
    assert one.info()["n_devices"] == 1 and single.info()["n_devices"] == 1
    cam = rt.config_rtiow(image_width=100, samples_per_pixel=20, max_depth=50).camera()
    want, st = cam.render_sum(single, seed=5)
    for part in (rt.PARTITION_TILES, rt.PARTITION_SAMPLES):
        got, st2 = cam.render_multi(one, seed=5, partition=part)
        assert np.array_equal(got, want) and st2["rays"] == st["rays"] and st2["samples"] == st["samples"]
    # all visible devices: NULL device list
    everything = rt.BVH(*rtiow, devices="all")
    assert everything.info()["n_devices"] == rt.device_count()
    got, _ = cam.render_multi(everything, seed=5)
    assert np.array_equal(got, want)
    # a single-device scene is accepted by rt_render_multi too
    got, _ = cam.render_multi(single, seed=5)
    assert np.array_equal(got, want)
    one.close()
    everything.close()


def test_tile_partition_is_bit_identical(single, triple):
    assert triple.info()["n_devices"] == 3
    # three copies on one GPU: no pair of different GPUs, and every copy can read every other's memory
    ti = triple.info()
    assert ti["peer_pairs"] == 0 and ti["peer_pairs_enabled"] == 0
    assert all(ti["peer_matrix"] >> (a * 8 + b) & 1 for a in range(3) for b in range(3)) and ti["peer_matrix"] >> 27 == 0
    # 130 x 73: the last tile row and column stick out of the image; 200 samples are cut into several chunks
    cam = rt.config_rtiow(image_width=130, samples_per_pixel=200, max_depth=50).camera()
    want, st = cam.render_sum(single, seed=12345)
    got, st3 = cam.render_multi(triple, seed=12345, partition=rt.PARTITION_TILES)
    assert np.array_equal(got, want)
    assert st3["rays"] == st["rays"] and st3["samples"] == st["samples"] and st3["launches"] >= 3
    # a rectangle off the tile grid, added on top of what the buffer holds
    tile = (13, 5, 121, 70)
    want_t, _ = cam.render_sum(single, seed=9, tile=tile, spp_begin=3, spp_end=40, rgb_sum=np.full_like(want, 0.25))
    got_t, _ = cam.render_multi(triple, seed=9, tile=tile, spp_begin=3, spp_end=40, rgb_sum=np.full_like(want, 0.25))
    assert np.array_equal(got_t, want_t)
    # more devices than tile rows: some get nothing
    five = rt.BVH(single.spheres, single.materials, devices=[0] * 5)
    small = rt.config_rtiow(image_width=40, samples_per_pixel=4, max_depth=50).camera()  # 22 rows: 3 tile rows
    a, sa = small.render_sum(single, seed=1)
    b, sb = small.render_multi(five, seed=1)
    assert np.array_equal(a, b) and sa["rays"] == sb["rays"]
    five.close()


def test_sample_partition_sums_in_device_order(single, triple):
    cam = rt.config_rtiow(image_width=130, samples_per_pixel=50, max_depth=50).camera()
    want, st = cam.render_sum(single, seed=77)
    got, st3 = cam.render_multi(triple, seed=77, partition=rt.PARTITION_SAMPLES)
    assert st3["rays"] == st["rays"] and st3["samples"] == st["samples"]
    assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()
    # the same three sample ranges rendered one after the other on one device and added in order: the same bits
    shards = cam.partition(3, rt.PARTITION_SAMPLES)
    assert [(int(s["spp_begin"]), int(s["spp_end"])) for s in shards] == [(0, 17), (17, 34), (34, 50)]
    by_hand = np.zeros_like(want)
    for s in shards:
        part, _ = cam.render_sum(single, seed=77, spp_begin=int(s["spp_begin"]), spp_end=int(s["spp_end"]))
        by_hand += part
    assert np.array_equal(got, by_hand)
    # frames brought together through copies on the first device instead of peer reads: the same bits again
    _set(b"force_staged_sum", 1.0)
    try:
        staged, _ = cam.render_multi(triple, seed=77, partition=rt.PARTITION_SAMPLES)
    finally:
        _set(b"force_staged_sum", 0.0)
    assert np.array_equal(staged, got)
    # fewer samples than devices
    one_sample, s1 = cam.render_multi(triple, seed=77, spp_begin=4, spp_end=5, partition=rt.PARTITION_SAMPLES)
    ref, r1 = cam.render_sum(single, seed=77, spp_begin=4, spp_end=5)
    assert np.array_equal(one_sample, ref) and s1["rays"] == r1["rays"]


def test_shards_into_one_device_frame(single):
    import torch

    cam = rt.config_rtiow(image_width=200, samples_per_pixel=130, max_depth=50).camera()
    h = cam.image_height
    whole = torch.zeros((h, 200, 3), dtype=torch.float64, device="cuda")
    parts = torch.zeros_like(whole)
    torch.cuda.synchronize()
    tile = (3, 2, 197, h - 1)
    st = cam.render_device(single, whole.data_ptr(), seed=3, tile=tile)
    rays = 0
    for shard in cam.partition(4, rt.PARTITION_TILES, tile=tile):
        rays += cam.render_shard_device(single, parts.data_ptr(), shard, seed=3)["rays"]
    torch.cuda.synchronize()
    assert rays == st["rays"] and torch.equal(parts, whole)
    with pytest.raises(rt.RtError):
        bad = cam.partition(2)[0].copy()
        bad["row_first"] = 2
        cam.render_shard_device(single, parts.data_ptr(), bad)


def test_ppm_text_does_not_depend_on_the_number_of_devices(single, triple):
    cfg = rt.config_rtiow(image_width=150, samples_per_pixel=12, max_depth=50)
    texts = []
    for world in (single, triple):
        with tempfile.TemporaryFile() as f:
            st = cfg.camera().render(world, seed=4, fd=f.fileno())
            f.seek(0)
            texts.append(f.read())
            assert st["samples"] == 150 * 84 * 12
    assert texts[0] == texts[1] and texts[0].startswith(b"P3\n150 84\n255\n")
    got, _ = cfg.camera().render_sum(single, seed=4)
    with tempfile.TemporaryFile() as f:
        rt.write_ppm(f.fileno(), got, 150, 84, 12)
        f.seek(0)
        assert f.read() == texts[0]


def test_concurrent_multi_calls_serialise(triple, single):
    import threading

    cam = rt.config_rtiow(image_width=96, samples_per_pixel=8, max_depth=50).camera()
    want, _ = cam.render_sum(single, seed=2)
    out, errors = [None] * 4, []

    def work(k):
        try:
            out[k] = cam.render_multi(triple, seed=2, partition=k % 2)[0]
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for k in range(4):
        if k % 2 == 0:
            assert np.array_equal(out[k], want)
        else:
            assert np.abs(out[k] - want).max() <= 1e-12 * np.abs(want).max()


def test_device_list_is_checked(rtiow):
    n = rt.device_count()
    with pytest.raises(rt.RtError) as e:
        rt.BVH(*rtiow, devices=[n])
    assert e.value.code == 1 and "does not exist" in str(e.value)
    with pytest.raises(rt.RtError) as e:
        rt.BVH(*rtiow, devices=[-1])
    assert e.value.code == 1
    with pytest.raises(rt.RtError) as e:
        rt.BVH(*rtiow, devices=[0] * 17)
    assert e.value.code == 4
    with pytest.raises(rt.RtError):
        rt.BVH(*rtiow, devices=[])


@pytest.mark.parametrize("builder", [rt.BUILD_DEVICE_LBVH, rt.BUILD_DEVICE_SAH, rt.BUILD_AUTO])
def test_multi_scene_through_the_four_wide_walk_and_device_builder(builder):
    # a scene that does not fit in LDS, copies built on the device: tiles still bit-identical, samples to rounding
    spheres, materials = rt.scene_field(3, 20000, 1.0 / 3.0, 1.0 / 3.0)
    one = rt.BVH(spheres, materials, builder=builder)
    two = rt.BVH(spheres, materials, builder=builder, devices=[0, 0])
    made_by = rt.BUILD_DEVICE_SAH if builder == rt.BUILD_AUTO else builder
    assert two.info()["n_devices"] == 2 and two.info()["builder"] == made_by and one.info()["builder"] == made_by
    cam = rt.config_field(20000, 120, 20, 50).camera()
    want, st = cam.render_sum(one, seed=3)
    got, st2 = cam.render_multi(two, seed=3, partition=rt.PARTITION_TILES)
    assert np.array_equal(got, want) and st2["rays"] == st["rays"]
    got, st2 = cam.render_multi(two, seed=3, partition=rt.PARTITION_SAMPLES)
    assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max() and st2["rays"] == st["rays"]
    one.close()
    two.close()

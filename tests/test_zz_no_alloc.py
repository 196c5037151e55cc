"""rt_render_device_noalloc (static device workspace, no allocation) against rt_render_device, through the C ABI."""
import numpy as np
import pytest

import raytracer_amd as rt

pytestmark = pytest.mark.gpu


def test_no_alloc_entry_point_matches_the_allocating_one():
    import torch

    world = rt.BVH(*rt.scene_rtiow(1))
    # first call: 256 x 144, 9 samples cut into 9 chunks of one sample: 7.96 MB of per-chunk sums in the static array
    # second call, straight after it and smaller: a rectangle of a 96-pixel frame, 5 samples; the array still holds the
    # first call's counters and sums
    for width, spp, tile, seed in ((256, 9, None, 11), (96, 5, (5, 3, 90, 50), 12)):
        cam = rt.config_rtiow(image_width=width, samples_per_pixel=spp, max_depth=50).camera()
        want = torch.full((cam.image_height, width, 3), 0.5, dtype=torch.float64, device="cuda")
        got = want.clone()
        torch.cuda.synchronize()
        st = cam.render_device(world, want.data_ptr(), seed=seed, tile=tile)
        assert st["launches"] == 3  # the tiles' entry lists, the render, the sum over chunks: the per-chunk sums are in use
        st2 = cam.render_device_noalloc(world, got.data_ptr(), seed=seed, tile=tile)
        torch.cuda.synchronize()
        assert st2["rays"] == st["rays"] and st2["samples"] == st["samples"] and st2["launches"] == 3
        assert torch.equal(got, want)  # bit for bit: both cut the request into the same chunks
        assert float((want - 0.5).abs().max().item()) > 0.1
    # without stats the call only enqueues; the caller waits for the stream
    cam = rt.config_rtiow(image_width=96, samples_per_pixel=5, max_depth=50).camera()
    a = torch.zeros((cam.image_height, 96, 3), dtype=torch.float64, device="cuda")
    b = torch.zeros_like(a)
    torch.cuda.synchronize()
    cam.render_device(world, a.data_ptr(), seed=3)
    assert cam.render_device_noalloc(world, b.data_ptr(), seed=3, want_stats=False) is None
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    # a request whose per-chunk sums do not fit is refused with a clear error, and nothing is rendered
    big = rt.config_rtiow(image_width=830, samples_per_pixel=750, max_depth=50).camera()
    c = torch.zeros((big.image_height, 830, 3), dtype=torch.float64, device="cuda")
    with pytest.raises(rt.RtError) as e:
        big.render_device_noalloc(world, c.data_ptr(), seed=1)
    assert e.value.code == 4 and "static workspace holds" in str(e.value)
    torch.cuda.synchronize()
    assert not bool(c.any().item())
    # one sample per pixel is one chunk: written straight into the frame, any frame size fits
    one = rt.config_rtiow(image_width=1920, samples_per_pixel=1, max_depth=50).camera()
    d = torch.zeros((one.image_height, 1920, 3), dtype=torch.float64, device="cuda")
    e2 = torch.zeros_like(d)
    torch.cuda.synchronize()
    s1 = one.render_device(world, d.data_ptr(), seed=8)
    s2 = one.render_device_noalloc(world, e2.data_ptr(), seed=8)
    torch.cuda.synchronize()
    assert s1["launches"] == 2 and s2["launches"] == 2 and torch.equal(d, e2)
    world.close()


def test_no_alloc_entry_point_on_a_scene_walked_through_the_four_wide_tree():
    # the traversal stacks' overflow columns and the entry lists are carved out of the static array behind the per-chunk sums
    import torch

    n = 60000
    world = rt.BVH(*rt.scene_field(7, n, 0.3, 0.3, 0.5), builder=rt.BUILD_AUTO)  # (built on the device, as rt_scene_create would)
    info = world.info()
    assert info["n_nodes_in_lds"] < info["n_device_nodes"] and info["builder"] == rt.BUILD_DEVICE_SAH
    for width, spp, seed in ((160, 16, 5), (96, 6, 6)):
        cam = rt.config_field(n, width, spp, 30).camera()
        want = torch.zeros((cam.image_height, width, 3), dtype=torch.float64, device="cuda")
        got = torch.zeros_like(want)
        torch.cuda.synchronize()
        st = cam.render_device(world, want.data_ptr(), seed=seed)
        st2 = cam.render_device_noalloc(world, got.data_ptr(), seed=seed)
        torch.cuda.synchronize()
        assert st2["rays"] == st["rays"] and st2["launches"] == st["launches"] == 3
        assert torch.equal(got, want) and bool(want.any().item())
    # a frame whose per-chunk sums leave no room for the stacks is refused
    cam = rt.config_field(n, 640, 64, 30).camera()
    c = torch.zeros((cam.image_height, 640, 3), dtype=torch.float64, device="cuda")
    with pytest.raises(rt.RtError) as e:
        cam.render_device_noalloc(world, c.data_ptr(), seed=1)
    assert e.value.code == 4 and "static workspace holds" in str(e.value)
    world.close()

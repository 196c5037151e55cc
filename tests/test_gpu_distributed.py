"""Two processes on one GPU render their shares of a frame with the HIP path and bring them together with a
collective (`gloo`, on host copies: two ranks cannot share one device under RCCL).  What bench.py does with one
rank per GPU and RCCL, at world size 2 with the real renderer: shares through rt_partition and
rt_render_shard_device, the reduce, the frame on rank 0 against a single-process render."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world_size, port, mode, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import raytracer_amd as rt
    from raytracer_amd.parallel import render_sample_split

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    torch.cuda.set_device(0)
    rt.set_device(0)
    world = rt.BVH(*rt.scene_rtiow(1))
    cam = rt.config_rtiow(image_width=200, samples_per_pixel=90, max_depth=50).camera()
    frame = torch.zeros((cam.image_height, 200, 3), dtype=torch.float64, device="cuda")
    host = torch.zeros((cam.image_height, 200, 3), dtype=torch.float64)
    torch.cuda.synchronize()

    if mode == "samples":
        def render(s0, s1, tile):
            cam.render_device(world, frame.data_ptr(), seed=12345, spp_begin=s0, spp_end=s1, tile=tile)
            torch.cuda.synchronize()
            host.copy_(frame)

        render_sample_split(render, host, 0, 90, rank, world_size)
    else:
        shard = cam.partition(world_size, rt.PARTITION_TILES, spp_begin=0, spp_end=90)[rank]
        cam.render_shard_device(world, frame.data_ptr(), shard, seed=12345)
        torch.cuda.synchronize()
        host.copy_(frame)
        dist.reduce(host, dst=0)  # the shares are disjoint: adding zeros brings them together exactly
    if rank == 0:
        np.save(out_path, host.numpy())
    dist.barrier()
    dist.destroy_process_group()
    world.close()


@pytest.mark.parametrize("mode", ["samples", "tiles"])
def test_two_processes_with_the_hip_renderer(mode, tmp_path):
    import torch.multiprocessing as mp

    import raytracer_amd as rt

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, port, mode, out), nprocs=2, join=True)
    got = np.load(out)
    world = rt.BVH(*rt.scene_rtiow(1))
    want, _ = rt.config_rtiow(image_width=200, samples_per_pixel=90, max_depth=50).camera().render_sum(world, seed=12345)
    world.close()
    if mode == "tiles":
        assert np.array_equal(got, want)
    else:
        assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()

"""The multi-rank partitioning of a frame, exercised with two processes over
`gloo` on CPU tensors.  The CPU oracle stands in for the renderer here (test
infrastructure); bench.py drives the same functions with the HIP renderer and
RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partitions_cover_everything_once():
    from raytracer_amd.parallel import partition_samples, partition_tiles

    for n in (1, 2, 3, 8):
        for lo, hi in ((0, 1000), (5, 6), (0, 0), (7, 14)):
            parts = partition_samples(lo, hi, n)
            assert parts[0][0] == lo and parts[-1][1] == hi and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1
        cover = np.zeros((45, 80), dtype=int)
        for rects in partition_tiles(80, 45, n):
            for x0, y0, x1, y1 in rects:
                cover[y0:y1, x0:x1] += 1
        assert (cover == 1).all()


def _worker(rank, world_size, port, mode, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import raytracer_amd as rt
    from oracle import oracle as orc
    from raytracer_amd.parallel import render_sample_split, render_tile_split

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    z = np.load(os.path.join(ROOT, "tests", "golden", "scene_rtiow_seed1.npz"))
    scene = orc.Scene(z["spheres"].view(orc.SPHERE_DTYPE), z["materials"].view(orc.MATERIAL_DTYPE))
    cfg = rt.config_rtiow(image_width=64, samples_per_pixel=64, max_depth=50)
    ocfg = cfg.record().view(orc.CONFIG_DTYPE)
    h = orc.camera_derive(ocfg)[0]
    frame = torch.zeros((h, 64, 3), dtype=torch.float64)
    view = frame.numpy()

    def render(s0, s1, tile):
        scene.render(ocfg, 12345, s0, s1, tile=tile, rgb_sum=view, nthreads=2)

    if mode == "samples":
        render_sample_split(render, frame, 0, 64, rank, world_size)
    else:
        render_tile_split(render, 64, h, 0, 64, rank, world_size)
        dist.reduce(frame, dst=0)  # only to bring the disjoint tiles together for the comparison
    if rank == 0:
        np.save(out_path, view)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["samples", "tiles"])
def test_two_ranks_reproduce_the_single_rank_frame(mode, tmp_path):
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, port, mode, out), nprocs=2, join=True)
    got = np.load(out)
    want = np.load(os.path.join(ROOT, "tests", "golden", "image_64x36_64spp.npz"))["rgb_sum"]
    if mode == "tiles":
        assert np.array_equal(got, want)  # disjoint pixels, same per-pixel order of additions
    else:
        assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()  # same paths; two partial sums added instead of one running sum

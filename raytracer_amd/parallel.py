"""Splitting one frame over the GPUs of a node (one process per GPU).

Every (pixel, sample) of the render loop is independent (src/camera.rs:144-151)
and sample s of pixel (i, j) is the same path whichever rank renders it, so a
frame can be cut two ways:

  tiles    each rank renders disjoint bands of rows for all samples; nothing is
           exchanged, rank r simply owns its rows of the result;
  samples  each rank renders all pixels for a contiguous range of sample indices
           and the per-rank accumulation buffers are summed onto one rank with a
           single reduce (RCCL over xGMI when the tensors live on GPUs).

The functions here only do the bookkeeping and the collective; the renderer is
passed in, so the same code runs under `gloo` on CPU tensors in the tests.
"""


def partition_samples(spp_begin, spp_end, world_size):
    """Contiguous sample ranges, one per rank, sizes differing by at most one."""
    n = spp_end - spp_begin
    base, extra = divmod(n, world_size)
    out, s = [], spp_begin
    for r in range(world_size):
        e = s + base + (1 if r < extra else 0)
        out.append((s, e))
        s = e
    return out


def partition_tiles(width, height, world_size, band=8):
    """Row bands of `band` rows dealt round-robin, so cheap sky rows and costly ground rows mix on every rank.

    Returns, per rank, a list of rectangles (x0, y0, x1, y1)."""
    out = [[] for _ in range(world_size)]
    for k, y0 in enumerate(range(0, height, band)):
        out[k % world_size].append((0, y0, width, min(y0 + band, height)))
    return out


def render_sample_split(render, frame, spp_begin, spp_end, rank, world_size, dst=0, group=None):
    """`render(s0, s1, tile)` adds samples [s0, s1) of `tile` (None = whole frame) into `frame`.

    Each rank renders its sample range into its own zeroed frame; one reduce(SUM) leaves the full
    frame on rank `dst`.  Returns the range this rank rendered."""
    import torch.distributed as dist

    s0, s1 = partition_samples(spp_begin, spp_end, world_size)[rank]
    if s1 > s0:
        render(s0, s1, None)
    if world_size > 1:
        dist.reduce(frame, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return s0, s1


def render_tile_split(render, width, height, spp_begin, spp_end, rank, world_size, band=8):
    """Each rank renders its row bands for all samples; no communication.  Returns the rank's rectangles."""
    rects = partition_tiles(width, height, world_size, band)[rank]
    for rect in rects:
        render(spp_begin, spp_end, rect)
    return rects

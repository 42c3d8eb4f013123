"""One-node multi-GPU rendering: one process per GPU, full scene replica each.

The reference renders a frame as independent tiles on a thread pool
(src/main.rs:157-161) and has no cross-tile state, so a frame shards two ways:

  mode "sample": rank r renders a contiguous range of the frame's samples for
                 every pixel; the float4 accumulation buffers are summed
                 (channel w sums the sample counts, src/main.rs:39).
  mode "tile":   rank r renders the tiles i with i % world == r (the
                 reference's row-major tile numbering, src/main.rs:72-85) at all
                 samples; buffers are disjoint, the same sum gathers them.

Either way the exchange is one sum-reduce of the width*height*4 f32 buffer to
rank 0 (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).  Random
numbers are keyed by (seed, pixel, sample), so the union over ranks is the
single-GPU image up to the order of f32 additions (bit-identical for "tile").
"""
from __future__ import annotations


def shard_plan(rank, world, mode, samples, sample_offset=0):
    """Render arguments of rank `rank` of `world` for a frame of `samples` spp."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("rank %r of world %r" % (rank, world))
    if mode == "sample":
        base, rem = divmod(samples, world)
        mine = base + (1 if rank < rem else 0)
        start = sample_offset + rank * base + min(rank, rem)
        return dict(samples=mine, sample_offset=start, shard_index=0, shard_count=1)
    if mode == "tile":
        return dict(samples=samples, sample_offset=sample_offset, shard_index=rank, shard_count=world)
    raise ValueError("mode must be 'sample' or 'tile'")


def node_render(render_fn, accum, rank, world, mode, samples, sample_offset=0, dist=None, wait=True):
    """Renders this rank's shard into `accum` (a zero-initialised torch tensor
    [H, W, 4] f32 that render_fn adds into) and sum-reduces it to rank 0.

    render_fn(accum, **plan) -> total rays traced by this rank (int).
    Returns this rank's ray count; after the call rank 0's accum holds the frame.

    wait=False starts the reduce and returns (rays, work): the collective then
    runs on the backend's own stream while the caller renders the next frame
    into another buffer; `accum` must not be touched before work.wait()
    (work is None when world == 1).
    """
    plan = shard_plan(rank, world, mode, samples, sample_offset)
    rays = 0
    if plan["samples"] > 0:
        rays = int(render_fn(accum, **plan))
    work = None
    if world > 1:
        if dist is None:
            import torch.distributed as dist
        work = dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM, async_op=not wait)
    if wait:
        return rays
    return rays, work

"""One image over several GPUs: how the sample slices of one render_multi call (render.rs:128-160) are
shared out over ranks, and how the per-rank images are combined.

render_multi gives each of its K worker threads samples_per_pixel / K samples of every pixel
(render.rs:130) and adds the K tonemapped slice images under a mutex (render.rs:141-150).  Slices
never exchange data, so a job of `world` ranks is ONE render_multi call with world * K slices: rank r
renders slices [r * K, (r + 1) * K) and a single sum-reduce of the H x W x 3 images to rank 0 stands in
for the mutex.  Per-GPU work does not depend on `world` (weak scaling); the random streams are keyed by
the absolute sample number, so the union over ranks is exactly the single-GPU render of all slices.
"""
from . import device


def rank_params(width, height, spp_per_gpu, slices_per_gpu, max_depth, seed, rank, world, flags=0):
    """Parameters of rank `rank`'s share of a world-wide render with world * spp_per_gpu samples per pixel."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside a world of {world}")
    return device.make_params(width, height, spp_per_gpu * world, n_slices=slices_per_gpu * world, max_depth=max_depth,
                              seed=seed, slice_begin=rank * slices_per_gpu, slice_end=(rank + 1) * slices_per_gpu,
                              flags=flags)


def reduce_images(dist, tensors, dst=0):
    """Sum the per-rank images onto rank `dst` (the mutex sum of render.rs:141-150); no-op in a world of one."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for t in tensors:
        dist.reduce(t, dst=dst)

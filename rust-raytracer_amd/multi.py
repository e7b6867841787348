"""One image over several GPUs: how the work of one render_multi call (render.rs:128-160) is shared out over
ranks (one process per GPU), and how the per-rank images are combined.

render_multi gives each of its K worker threads samples_per_pixel / K samples of every pixel (render.rs:130)
and adds the K tonemapped slice images under a mutex (render.rs:141-150).  Neither slices nor pixels
exchange data while they render, so a job can be cut along both:

  sample slices   rank g of G renders slices [g*K/G, (g+1)*K/G) of the whole image; the ranks' images are
                  SUMMED onto one rank (a reduce: it stands in for the mutex sum);
  bands of rows   rank b of B renders image rows [b*H/B, (b+1)*H/B) with all K slices; the bands are disjoint
                  pixels, so they are only GATHERED.

A world of B*G ranks does both: rank r is band r // G, slice group r % G; the G ranks of a band reduce onto
the band's first rank, and the B first ranks gather their bands on rank 0.  The random streams are keyed by
(pixel position in the whole image, absolute sample number), so the union of the ranks' work is exactly the
single-GPU render of the whole job: bit for bit for bands, up to the association of the f32 sum for slices.

Which job the ranks share is the caller's choice: `job_params(..., scaling="strong")` is ONE render of `spp`
samples and K slices whatever the world size (the time per image falls with the world size);
`scaling="weak"` is a render of world * spp samples in world * K slices (the work per GPU stays the same).
"""
from . import device

PARTITIONS = ("slices", "rows", "both")


def grid(world, partition="slices"):
    """(B, G): bands of rows x slice groups, B * G == world."""
    if world < 1:
        raise ValueError("world must be at least 1")
    if partition == "slices":
        return 1, world
    if partition == "rows":
        return world, 1
    if partition == "both":  # the smallest divisor of world that is >= sqrt(world) bands; the rest slice groups
        b = next(d for d in range(1, world + 1) if world % d == 0 and d * d >= world)
        return b, world // b
    raise ValueError(f"partition must be one of {PARTITIONS}")


def job_params(width, height, spp, n_slices, max_depth, seed, world=1, scaling="strong", flags=0):
    """The whole job the ranks share: `spp` samples in `n_slices` slices (strong), or world times both (weak)."""
    if scaling not in ("strong", "weak"):
        raise ValueError("scaling must be 'strong' or 'weak'")
    m = world if scaling == "weak" else 1
    return device.make_params(width, height, spp * m, n_slices=n_slices * m, max_depth=max_depth, seed=seed, flags=flags)


def _cut(n, parts, i):
    return (i * n) // parts, ((i + 1) * n) // parts


def rank_share(job, rank, world, partition="slices"):
    """Parameters of rank `rank`'s share of `job` (a copy with its slice range and its band of rows set)."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside a world of {world}")
    bands, groups = grid(world, partition)
    if bands > job.height:
        raise ValueError(f"{bands} bands of rows for an image of {job.height} rows")
    b, g = divmod(rank, groups)
    s0, s1 = _cut(job.n_slices, groups, g)  # (more slice groups than slices: some ranks render nothing and add zeros)
    r0, r1 = _cut(job.height, bands, b)
    share = device.make_params(job.width, job.height, job.samples_per_pixel, n_slices=job.n_slices, max_depth=job.max_depth,
                               seed=job.seed, slice_begin=s0, slice_end=s1, sample_offset=job.sample_offset, flags=job.flags,
                               row_begin=r0, row_end=r1)
    return share


def rank_params(width, height, spp_per_gpu, slices_per_gpu, max_depth, seed, rank, world, flags=0):
    """Weak scaling over sample slices: rank `rank`'s share of a render with world * spp_per_gpu samples per pixel."""
    job = job_params(width, height, spp_per_gpu, slices_per_gpu, max_depth, seed, world, "weak", flags)
    return rank_share(job, rank, world, "slices")


class Combiner:
    """The collectives that turn the ranks' images into the job's image on rank 0.

    Creating one is collective (every rank of the world calls it with the same arguments): it sets up one process
    group per band (for the reduce over slice groups) and one of the bands' first ranks (for the gather)."""

    def __init__(self, dist, world, partition, height, stage_through_host=False):
        # stage_through_host: the images are copied to host memory for the collectives and the result back to the device of
        # rank 0's tensors (for a process group that cannot take device tensors, e.g. gloo in a rehearsal on shared GPUs)
        self.stage = bool(stage_through_host)
        self.dist = dist if (dist is not None and world > 1) else None
        self.world = world
        self.bands, self.groups = grid(world, partition)
        self.height = height
        self.rank = self.dist.get_rank() if self.dist else 0
        self.band, self.group = divmod(self.rank, self.groups)
        self.band_group = self.roots_group = None
        if self.dist:
            if self.dist.get_world_size() != world:
                raise ValueError("world does not match the process group")
            if self.groups > 1 and self.bands > 1:
                for b in range(self.bands):  # every rank creates every group, in the same order
                    pg = self.dist.new_group(list(range(b * self.groups, (b + 1) * self.groups)))
                    if b == self.band:
                        self.band_group = pg
                self.roots_group = self.dist.new_group([b * self.groups for b in range(self.bands)])
        self.max_rows = max(_cut(height, self.bands, b)[1] - _cut(height, self.bands, b)[0] for b in range(self.bands))

    def combine(self, tensors):
        """tensors: this rank's band images (rows x W x 3 each).  Returns the whole-image tensors on rank 0 and
        None elsewhere.  One reduce per tensor within the band (slice groups), one gather per tensor over the bands."""
        if self.dist is None:
            return list(tensors)
        import torch
        dist = self.dist
        tensors = list(tensors)
        home = tensors[0].device if tensors else None
        if self.stage:
            tensors = [t.cpu() for t in tensors]
            back = lambda ts: [t.to(home) for t in ts]  # noqa: E731
        else:
            back = lambda ts: ts  # noqa: E731
        if self.groups > 1:  # the mutex sum of render.rs:141-150 over the slice groups of this band
            root = self.band * self.groups
            for t in tensors:
                dist.reduce(t, dst=root, group=self.band_group)
        if self.bands == 1:
            return back(tensors) if self.rank == 0 else None
        if self.group != 0:
            return None
        out = []
        for t in tensors:  # disjoint rows: gathered, padded to the tallest band (NCCL gathers equal shapes)
            pad = t
            if t.shape[0] != self.max_rows:
                pad = torch.zeros((self.max_rows,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
                pad[:t.shape[0]] = t
            parts = [torch.empty_like(pad) for _ in range(self.bands)] if self.rank == 0 else None
            dist.gather(pad.contiguous(), parts, dst=0, group=self.roots_group)
            if self.rank == 0:
                rows = [_cut(self.height, self.bands, b) for b in range(self.bands)]
                out.append(torch.cat([parts[b][:r1 - r0] for b, (r0, r1) in enumerate(rows)], dim=0))
        return back(out) if self.rank == 0 else None


def reduce_images(dist, tensors, dst=0):
    """Sum the per-rank images onto rank `dst` (the mutex sum of render.rs:141-150); no-op in a world of one."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for t in tensors:
        dist.reduce(t, dst=dst)

"""Two ranks share one render over gloo on the CPU: the slice and row arithmetic and the collectives of
rust_raytracer_amd.multi, with the CPU oracle standing in for the GPU renderer of each rank."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import rust_raytracer_amd  # noqa: F401
from rust_raytracer_amd import host, multi
import oracle_lib as O

W, H, SPP_PER_RANK, K, DEPTH, SEED = 40, 31, 8, 2, 20, 11


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, scaling, partition):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _, flat, _ = host.build_config_scene("cornell", seed=SEED)
        job = multi.job_params(W, H, SPP_PER_RANK, K, DEPTH, SEED, world, scaling)
        p = multi.rank_share(job, rank, world, partition)
        r = O.render(flat, p, n_threads=2)
        np.save(os.path.join(out_dir, f"sums{rank}.npy"), r["slice_sums"])
        comb = multi.Combiner(dist, world, partition, H)
        out = comb.combine((torch.from_numpy(r["resolved"].copy()), torch.from_numpy(r["linear"].copy())))
        assert (out is not None) == (rank == 0)
        if rank == 0:
            np.save(os.path.join(out_dir, "resolved.npy"), out[0].numpy())
            np.save(os.path.join(out_dir, "linear.npy"), out[1].numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,scaling,partition", [(2, "weak", "slices"), (2, "strong", "slices"), (2, "strong", "rows"),
                                                     (2, "weak", "rows"), (4, "strong", "both"), (3, "strong", "rows")])
def test_ranks_render_the_shares_of_one_image(built, tmp_path, world, scaling, partition):
    mp.start_processes(_worker, args=(world, _free_port(), str(tmp_path), scaling, partition), nprocs=world, start_method="spawn")
    _, flat, _ = host.build_config_scene("cornell", seed=SEED)
    job = multi.job_params(W, H, SPP_PER_RANK, K, DEPTH, SEED, world, scaling)
    assert job.samples_per_pixel == SPP_PER_RANK * (world if scaling == "weak" else 1)
    whole = O.render(flat, job, n_threads=4)
    # every rank's share is bit for bit its part of the single-process render: streams are keyed by the pixel's place in the
    # whole image and the absolute sample number
    for rank in range(world):
        p = multi.rank_share(job, rank, world, partition)
        got = np.load(tmp_path / f"sums{rank}.npy")
        r1 = p.row_end or H
        assert got.shape == (p.slice_end - p.slice_begin, r1 - p.row_begin, W, 3)
        assert np.array_equal(got, whole["slice_sums"][p.slice_begin:p.slice_end, p.row_begin:r1], equal_nan=True)
    resolved, linear = np.load(tmp_path / "resolved.npy"), np.load(tmp_path / "linear.npy")
    assert resolved.shape == (H, W, 3)
    if multi.grid(world, partition)[1] == 1:
        # bands only: a gather of disjoint rows, the whole image bit for bit
        assert np.array_equal(resolved, whole["resolved"]) and np.array_equal(linear, whole["linear"])
        assert np.array_equal(O.cast_u8(resolved), whole["rgb8"])
    else:
        # the reduce adds the ranks' partial sums: same image up to the association of the f32 sum
        assert np.allclose(resolved, whole["resolved"], rtol=1e-6, atol=1e-4)
        assert np.allclose(linear, whole["linear"], rtol=1e-6, atol=1e-6)
        assert np.abs(O.cast_u8(resolved).astype(int) - whole["rgb8"].astype(int)).max() <= 1


def test_shares_partition_the_job():
    for world, partition in [(4, "slices"), (8, "rows"), (8, "both"), (6, "both"), (1, "both"), (7, "rows")]:
        job = multi.job_params(64, 50, 100, 8, 50, 1, world, "strong")
        assert (job.samples_per_pixel, job.n_slices) == (100, 8)
        cells = np.zeros((job.n_slices, job.height), dtype=int)
        bands, groups = multi.grid(world, partition)
        assert bands * groups == world
        for rank in range(world):
            p = multi.rank_share(job, rank, world, partition)
            assert p.row_end != 0
            cells[p.slice_begin:p.slice_end, p.row_begin:p.row_end] += 1
        assert (cells == 1).all(), (world, partition)  # every (slice, row) belongs to exactly one rank
    assert multi.grid(8, "both") == (4, 2) and multi.grid(4, "both") == (2, 2) and multi.grid(2, "both") == (2, 1)
    seen = []
    for rank in range(4):  # weak scaling over slices: the job grows with the world
        p = multi.rank_params(64, 48, 100, 8, 50, 1, rank, 4)
        assert p.samples_per_pixel == 400 and p.n_slices == 32 and p.rows == 48
        seen += list(range(p.slice_begin, p.slice_end))
    assert seen == list(range(32))
    with pytest.raises(ValueError):
        multi.rank_params(64, 48, 100, 8, 50, 1, 4, 4)
    with pytest.raises(ValueError):
        multi.rank_share(multi.job_params(8, 2, 4, 1, 5, 1), 0, 4, "rows")  # more bands than rows
    multi.reduce_images(None, ())  # a world of one reduces nothing
    one = multi.Combiner(None, 1, "both", 48)
    assert one.combine(("a", "b")) == ["a", "b"]

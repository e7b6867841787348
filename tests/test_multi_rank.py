"""Two ranks share one render over gloo on the CPU: the slice arithmetic and the reduce of
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

W, H, SPP_PER_RANK, K, DEPTH, SEED = 40, 30, 8, 2, 20, 11


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _, flat, _ = host.build_config_scene("cornell", seed=SEED)
        p = multi.rank_params(W, H, SPP_PER_RANK, K, DEPTH, SEED, rank, world)
        r = O.render(flat, p, n_threads=2)
        np.save(os.path.join(out_dir, f"sums{rank}.npy"), r["slice_sums"])
        resolved, linear = torch.from_numpy(r["resolved"].copy()), torch.from_numpy(r["linear"].copy())
        multi.reduce_images(dist, (resolved, linear))
        if rank == 0:
            np.save(os.path.join(out_dir, "resolved.npy"), resolved.numpy())
            np.save(os.path.join(out_dir, "linear.npy"), linear.numpy())
    finally:
        dist.destroy_process_group()


def test_two_ranks_render_the_slices_of_one_image(built, tmp_path):
    world = 2
    mp.start_processes(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, start_method="spawn")
    _, flat, _ = host.build_config_scene("cornell", seed=SEED)
    whole = O.render(flat, O.make_params(W, H, SPP_PER_RANK * world, n_slices=K * world, max_depth=DEPTH, seed=SEED), n_threads=4)
    # every rank's slices are bit for bit the slices of the single-process render: streams are keyed by absolute sample
    for rank in range(world):
        got = np.load(tmp_path / f"sums{rank}.npy")
        assert np.array_equal(got, whole["slice_sums"][rank * K:(rank + 1) * K], equal_nan=True)
    # the reduce adds the ranks' partial sums: same image up to the association of the f32 sum
    resolved, linear = np.load(tmp_path / "resolved.npy"), np.load(tmp_path / "linear.npy")
    assert np.allclose(resolved, whole["resolved"], rtol=1e-6, atol=1e-4)
    assert np.allclose(linear, whole["linear"], rtol=1e-6, atol=1e-6)
    d8 = np.abs(O.cast_u8(resolved).astype(int) - whole["rgb8"].astype(int))
    assert d8.max() <= 1


def test_rank_params_partition_the_slices():
    seen = []
    for rank in range(4):
        p = multi.rank_params(64, 48, 100, 8, 50, 1, rank, 4)
        assert p.samples_per_pixel == 400 and p.n_slices == 32
        seen += list(range(p.slice_begin, p.slice_end))
    assert seen == list(range(32))
    with pytest.raises(ValueError):
        multi.rank_params(64, 48, 100, 8, 50, 1, 4, 4)
    multi.reduce_images(None, ())  # a world of one reduces nothing

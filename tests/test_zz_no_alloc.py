"""rr_render_device_noalloc (the entry point that allocates no device memory: static 64 MiB workspace inside the library)
against rr_render_device, through the C ABI: same bits in every output, on calls in a row that share the static workspace."""
import numpy as np
import pytest
import torch

import rust_raytracer_amd  # noqa: F401
from rust_raytracer_amd import abi, device, host

pytestmark = pytest.mark.gpu


def _both(ctx, p):
    """(outputs of rr_render_device, outputs of rr_render_device_noalloc) for the same parameters, in caller-owned device buffers."""
    dev = torch.device("cuda", 0)
    k, h, w = p.slice_end - p.slice_begin, p.rows, p.width
    res = []
    for fn in (ctx.render_device, ctx.render_device_noalloc):
        sums = torch.full((k, h, w, 3), -7.0, dtype=torch.float32, device=dev)
        resolved = torch.full((h, w, 3), -7.0, dtype=torch.float32, device=dev)
        linear = torch.full((h, w, 3), -7.0, dtype=torch.float32, device=dev)
        counters = torch.zeros(abi.N_COUNTERS, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        fn(p, sums.data_ptr(), resolved.data_ptr(), linear.data_ptr(), counters.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        res.append([t.cpu().numpy() for t in (sums, resolved, linear, counters)])
    return res


def _assert_same(a, b):
    for x, y in zip(a, b):
        assert np.array_equal(x, y, equal_nan=True)


@pytest.mark.parametrize("name", ["cornell", "suzanne", "final"])
def test_noalloc_entry_equals_render_device_on_calls_in_a_row(gpu_ctx, name):
    b, flat, _ = host.build_config_scene(name)
    gpu_ctx.upload(flat)
    # 1: several slice groups and sample batches through the static workspace (1200 x 900 x 4 slices = 4.3 M slots > 4 Mi paths)
    big = device.make_params(1200, 900, 8, n_slices=4, max_depth=12, seed=5)
    a, b_ = _both(gpu_ctx, big)
    _assert_same(a, b_)
    assert a[3][0] == 1200 * 900 * 8
    # 2: a smaller call right after it, on the workspace the first one left behind
    small = device.make_params(64, 48, 36, n_slices=3, max_depth=30, seed=6, slice_begin=1, slice_end=3, row_begin=7, row_end=40,
                               flags=abi.FLAG_COUNTERS)
    a, b_ = _both(gpu_ctx, small)
    _assert_same(a, b_)
    # 3: and once more the first
    a2, b2 = _both(gpu_ctx, big)
    _assert_same(a2, b2)


def test_noalloc_entry_allocates_nothing():
    """On a context of its own whose only renders go through rr_render_device_noalloc: the context's workspace stays
    unallocated (rr_scene_info.workspace_bytes == 0: nothing was hipMalloc'ed behind the call), and the free device memory the
    runtime reports is the same before and after warm calls (the first call of a kernel may make the runtime allocate its
    private-segment scratch, so the comparison starts after one call of every kernel build involved)."""
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    for name in ("cornell", "final"):
        b, flat, _ = host.build_config_scene(name)
        ctx = device.Context(0)
        try:
            ctx.upload(flat)
            p = device.make_params(320, 200, 24, n_slices=4, max_depth=20, seed=9)
            sums = torch.empty((4, 200, 320, 3), dtype=torch.float32, device=dev)
            resolved = torch.empty((200, 320, 3), dtype=torch.float32, device=dev)
            linear = torch.empty((200, 320, 3), dtype=torch.float32, device=dev)
            counters = torch.zeros(abi.N_COUNTERS, dtype=torch.int64, device=dev)
            args = (p, sums.data_ptr(), resolved.data_ptr(), linear.data_ptr(), counters.data_ptr(), stream)
            ctx.render_device_noalloc(*args)  # warm: code objects loaded, runtime scratch in place
            torch.cuda.synchronize()
            first = resolved.cpu().numpy()
            free_before, _ = torch.cuda.mem_get_info()
            for _ in range(3):
                ctx.render_device_noalloc(*args)
            torch.cuda.synchronize()
            free_after, _ = torch.cuda.mem_get_info()
            assert free_after == free_before, (name, free_before, free_after)
            assert ctx.info()["workspace_bytes"] == 0, name
            assert np.array_equal(resolved.cpu().numpy(), first)
            # the allocating entry on the same context does allocate, and the getter sees it
            ctx.render_device(*args)
            torch.cuda.synchronize()
            assert ctx.info()["workspace_bytes"] > 0
            assert np.array_equal(resolved.cpu().numpy(), first)
        finally:
            ctx.close()


def test_noalloc_entry_rejects_what_it_cannot_hold(gpu_ctx):
    b, flat, _ = host.build_config_scene("cornell")
    gpu_ctx.upload(flat)
    dev = torch.device("cuda", 0)
    sums = torch.zeros(16, dtype=torch.float32, device=dev)
    too_big = device.make_params(4096, 1025, 1, n_slices=1)  # 4096 x 1025 pixels > RR_NOALLOC_MAX_PATHS
    assert too_big.width * too_big.height > abi.NOALLOC_MAX_PATHS
    with pytest.raises(device.DeviceError, match="static workspace"):
        gpu_ctx.render_device_noalloc(too_big, sums.data_ptr())
    band = device.make_params(4096, 1025, 1, n_slices=1, row_begin=0, row_end=1024)  # exactly the limit: accepted
    out = torch.zeros((1024, 4096, 3), dtype=torch.float32, device=dev)
    gpu_ctx.render_device_noalloc(band, out.data_ptr())
    torch.cuda.synchronize()
    assert float(out.abs().sum()) > 0
    for flags in (abi.FLAG_PERSISTENT, abi.FLAG_WAVEFRONT):
        with pytest.raises(device.DeviceError, match="path-granular"):
            gpu_ctx.render_device_noalloc(device.make_params(8, 8, 1, flags=flags), sums.data_ptr())

"""Dead paths.  Once the weight a path multiplies everything it meets by is exactly zero in every channel, the reference walks
the path to its end all the same (render.rs:29-34) and adds 0; the library's default ends it there (include/rr_hip.h,
RR_FLAG_TRACE_DEAD_PATHS).  The two agree on every sample except where the rest of the walk meets a NaN or an infinity
(0 * NaN = NaN, which the reference's per-slice scrub turns into one zeroed pixel of one slice, render.rs:102-119).  Held here:
the C++ oracle and every kernel organisation implement the same rule, the flag restores the reference's walk, and how rare the
differing samples are in the scenes that have a light the estimator can sample."""
import numpy as np
import pytest

import rust_raytracer_amd  # noqa: F401
from rust_raytracer_amd import abi, device, host
import oracle_lib as O

TRACE = abi.FLAG_TRACE_DEAD_PATHS


def _diff(full, cut):
    """Per-sample arrays of the two walks -> (entries that differ, of which the full walk is NaN, entries)."""
    differ = ~((full == cut) | (np.isnan(full) & np.isnan(cut)))
    return int(differ.sum()), int((differ & np.isnan(full)).sum()), full.size


@pytest.mark.parametrize("name", ["cornell", "final", "suzanne"])
def test_oracle_walks_agree_except_where_the_full_walk_is_nan(built, name):
    """One sample per slice, so that every entry of slice_sums is one sample: the cut walk returns the full walk's value for
    every sample whose full value is not NaN, it traces fewer rays, and the differing samples are a tiny fraction in the
    scenes with a light."""
    b, flat, _ = host.build_config_scene(name)
    kw = dict(n_slices=16, max_depth=50, seed=5)
    full = O.render(flat, O.make_params(48, 48, 16, flags=TRACE, **kw), n_threads=8, counters=True)
    cut = O.render(flat, O.make_params(48, 48, 16, **kw), n_threads=8, counters=True)
    n_diff, n_diff_nan, n = _diff(full["slice_sums"], cut["slice_sums"])
    assert n_diff == n_diff_nan, "a sample changed whose full walk is not NaN"
    assert n_diff <= 3e-4 * n, (n_diff, n)
    assert cut["counters"]["samples"] == full["counters"]["samples"]
    assert cut["counters"]["rays"] < full["counters"]["rays"]
    if name == "final":  # most rays of this scene scatter on in the fog after their weight has become zero
        assert cut["counters"]["rays"] < 0.6 * full["counters"]["rays"]


def test_oracle_without_a_light_the_nan_samples_move(built):
    """The random spheres have no light to sample: half of the Lambertian bounces are the 0/0 of render.rs:34, a dead path
    meets one with high probability, and the cut walk then keeps a 0 where the reference has a NaN.  Both images are mostly
    scrubbed pixels; this is the case the flag exists for."""
    b, flat, _ = host.build_config_scene("spheres")
    kw = dict(n_slices=16, max_depth=50, seed=5)
    full = O.render(flat, O.make_params(40, 24, 16, flags=TRACE, **kw), n_threads=8)
    cut = O.render(flat, O.make_params(40, 24, 16, **kw), n_threads=8)
    n_diff, n_diff_nan, n = _diff(full["slice_sums"], cut["slice_sums"])
    assert n_diff == n_diff_nan and n_diff > 0
    assert np.isnan(full["slice_sums"]).mean() > 0.03


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cornell", "final", "suzanne", "spheres"])
def test_device_walks_match_the_oracles_and_each_other(gpu_ctx, name):
    """Both walks on the device against the same walk of the oracle (sample by sample, the bars of the other parity tests),
    every kernel organisation the same bits under either, and full against cut on the device as on the oracle."""
    b, flat, _ = host.build_config_scene(name)
    gpu_ctx.upload(flat)
    kw = dict(n_slices=16, max_depth=50, seed=5)
    res = {}
    for label, tf in (("full", TRACE), ("cut", 0)):
        got = gpu_ctx.render(device.make_params(48, 48, 16, flags=tf, **kw))
        want = O.render(flat, O.make_params(48, 48, 16, flags=tf, **kw), n_threads=16, counters=True)
        a, o = got["slice_sums"], want["slice_sums"]
        assert (np.isnan(a) == np.isnan(o)).mean() > 0.995
        both = ~np.isnan(a) & ~np.isnan(o)
        assert np.isclose(a[both], o[both], rtol=1e-3, atol=1e-5).mean() > 0.985
        assert abs(got["counters"]["rays"] - want["counters"]["rays"]) <= 0.01 * want["counters"]["rays"]
        for flags in (abi.FLAG_PATHS, abi.FLAG_PERSISTENT, abi.FLAG_WAVEFRONT, abi.FLAG_WAVEFRONT | abi.FLAG_NO_FINISH_KERNEL, abi.FLAG_COUNTERS):
            other = gpu_ctx.render(device.make_params(48, 48, 16, flags=flags | tf, **kw))
            assert np.array_equal(a, other["slice_sums"], equal_nan=True), (label, flags)
        res[label] = got
    n_diff, n_diff_nan, n = _diff(res["full"]["slice_sums"], res["cut"]["slice_sums"])
    assert n_diff == n_diff_nan
    if name != "spheres":
        assert n_diff <= 3e-4 * n
    assert res["cut"]["counters"]["rays"] < res["full"]["counters"]["rays"]


@pytest.mark.gpu
def test_how_rare_the_differing_samples_are_at_scale(gpu_ctx):
    """final_scene, 800x800, 64 spp in 64 slices (41 M samples, every entry one sample): the samples the cut walk changes are
    NaN in the full walk, and there are fewer than 5 in a million of them; the 8-bit images of the two walks at K = 8 differ in
    a few pixels at most."""
    b, flat, _ = host.build_config_scene("final")
    gpu_ctx.upload(flat)
    kw = dict(n_slices=64, max_depth=50, seed=1)
    full = gpu_ctx.render(device.make_params(800, 800, 64, flags=TRACE, **kw))
    cut = gpu_ctx.render(device.make_params(800, 800, 64, **kw))
    n_diff, n_diff_nan, n = _diff(full["slice_sums"], cut["slice_sums"])
    print(f"final_scene: {n_diff} of {n // 3} samples differ between the walks ({n_diff_nan} NaN channels in the full walk); "
          f"rays {full['counters']['rays']} -> {cut['counters']['rays']}")
    assert n_diff == n_diff_nan
    assert n_diff <= 5e-6 * n
    assert cut["counters"]["rays"] < 0.5 * full["counters"]["rays"]
    k8 = dict(n_slices=8, max_depth=50, seed=1)
    a = gpu_ctx.render(device.make_params(800, 800, 64, flags=TRACE, **k8), slice_sums=False)["rgb8"]
    c = gpu_ctx.render(device.make_params(800, 800, 64, **k8), slice_sums=False)["rgb8"]
    assert (a != c).any(axis=2).sum() <= 40

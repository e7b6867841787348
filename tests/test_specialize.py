"""rr_scene_specialize: the path kernel compiled at run time for one scene's program (its ops as constants, the op loop
unrolled) against the interpreter the library ships: the same bits in every output, for every kind of program, and the
interpreter back in place whenever the compiled kernel does not apply (another scene, counting flags, RR_SPECIALIZE_OFF)."""
import ctypes as C

import numpy as np
import pytest
import torch  # noqa: F401  (before the library: a process that uses both loads torch's HIP runtime first, as bench.py does)

import rust_raytracer_amd  # noqa: F401
from rust_raytracer_amd import abi, build, device, host


def _lib():
    return abi.bind(C.CDLL(abi.HIP_LIB), abi.HIP_PROTOTYPES)


def test_compile_options_are_those_of_the_library_build():
    """What decides the code a source line becomes must be the same in both builds (the arithmetic contract of the path: one
    IEEE f32 operation per source operation, no packed pairs): every such flag of build.HIPCC_FLAGS is an option of the
    per-scene compile, which names the target its own way."""
    opts = _lib().rr_specialize_options().decode().split()
    assert "--gpu-architecture=gfx950" in opts
    for flag in build.HIPCC_FLAGS:
        if flag in ("--offload-arch=gfx950", "-fPIC", "-shared", "-Wall"):
            continue
        assert flag in opts, flag
    assert set(opts) - {"--gpu-architecture=gfx950"} <= set(build.HIPCC_FLAGS)


def test_device_sources_inside_the_library_compile_without_a_gpu():
    """The library carries the text of its device headers; the runtime compiler takes them as they are (no GPU needed for a
    compile: the target is named).  A two-op program, the lean kernel build."""
    lib = _lib()
    ms = C.c_float(0)
    rc = lib.rr_specialize_selftest(C.byref(ms))
    assert rc == 0, lib.rr_last_error().decode()[:3000]
    assert 10.0 < ms.value < 120000.0


def _images(ctx, p):
    r = ctx.render(p)
    return r["slice_sums"], r["resolved"], r["linear"], r["rgb8"], r["counters"]["samples"], r["counters"]["rays"]


def _assert_same(a, b):
    for x, y in zip(a, b):
        assert np.array_equal(x, y, equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize("name,auto", [("cornell", True), ("spheres", False), ("suzanne", False), ("final", False)])
def test_specialized_kernel_renders_the_interpreters_bits_on_the_benchmark_scenes(gpu_ctx, name, auto):
    b, flat, _ = host.build_config_scene(name)
    info = gpu_ctx.upload(flat)
    assert not info["specialized"]
    p = device.make_params(160, 120, 24, n_slices=4, max_depth=30, seed=9, flags=abi.FLAG_PATHS)
    band = device.make_params(160, 120, 24, n_slices=4, max_depth=30, seed=9, slice_begin=1, slice_end=3, row_begin=11, row_end=97,
                              sample_offset=5, flags=abi.FLAG_PATHS)
    want, want_band = _images(gpu_ctx, p), _images(gpu_ctx, band)
    info = gpu_ctx.specialize("auto")  # only programs without BVH subtrees
    assert info["specialized"] == auto
    info = gpu_ctx.specialize("always")
    assert info["specialized"] and info["specialize_ms"] > 0.0
    _assert_same(_images(gpu_ctx, p), want)
    _assert_same(_images(gpu_ctx, band), want_band)
    # counting flags are the interpreter's business: same images, and the counters of the reference's visit order are there
    counted = gpu_ctx.render(device.make_params(160, 120, 24, n_slices=4, max_depth=30, seed=9, flags=abi.FLAG_PATHS | abi.FLAG_COUNTERS))
    assert np.array_equal(counted["slice_sums"], want[0], equal_nan=True)
    assert counted["counters"]["rect_tests"] + counted["counters"]["sphere_tests"] + counted["counters"]["tri_tests"] > 0
    # a second call finds the kernel in place; OFF takes it out
    again = gpu_ctx.specialize("always")
    assert again["specialized"]
    assert not gpu_ctx.specialize("off")["specialized"]
    _assert_same(_images(gpu_ctx, p), want)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(4))
def test_specialized_kernel_on_randomly_assembled_programs(gpu_ctx, seed):
    """Programs with every kind of op (media with all kinds of boundaries, transforms, lists flattened into the program,
    general and simple subtrees, leaf refs): the compile covers the general builds as well."""
    from test_gpu_parity import _random_scene
    b, flat = _random_scene(70 + seed)
    gpu_ctx.upload(flat)
    p = device.make_params(64, 48, 8, n_slices=8, max_depth=20, seed=3, flags=abi.FLAG_PATHS)
    want = _images(gpu_ctx, p)
    assert gpu_ctx.specialize("always")["specialized"]
    _assert_same(_images(gpu_ctx, p), want)


@pytest.mark.gpu
@pytest.mark.parametrize("semantics", [abi.SEMANTICS_INTENDED, abi.SEMANTICS_BOOK])
def test_specialized_kernel_under_the_labelled_semantics(gpu_ctx, semantics):
    """The labelled what-if modes select other builds of the kernel (the semantics switch is a run-time value there): the
    per-scene compile instantiates the same one."""
    for name in ("cornell", "final"):
        b, flat, _ = host.build_config_scene(name)
        flat.semantics = semantics
        gpu_ctx.upload(flat)
        p = device.make_params(96, 72, 8, n_slices=2, max_depth=12, seed=4, flags=abi.FLAG_PATHS)
        want = _images(gpu_ctx, p)
        assert gpu_ctx.specialize("always")["specialized"]
        _assert_same(_images(gpu_ctx, p), want)
        flat.semantics = abi.SEMANTICS_REFERENCE


@pytest.mark.gpu
def test_a_new_scene_drops_the_compiled_kernel_and_the_table_remembers_programs(gpu_ctx):
    b1, cornell, _ = host.build_config_scene("cornell")
    b2, spheres, _ = host.build_config_scene("spheres")
    p = device.make_params(80, 60, 8, n_slices=2, max_depth=10, seed=2, flags=abi.FLAG_PATHS)
    gpu_ctx.upload(cornell)
    want_cornell = _images(gpu_ctx, p)
    first = gpu_ctx.specialize("always")
    assert first["specialized"]
    info = gpu_ctx.upload(spheres)  # the compiled kernel was for the other program
    assert not info["specialized"]
    want_spheres = _images(gpu_ctx, p)
    assert gpu_ctx.specialize("always")["specialized"]
    _assert_same(_images(gpu_ctx, p), want_spheres)
    gpu_ctx.upload(cornell)
    second = gpu_ctx.specialize("always")  # compiled before in this process: found, not compiled again
    assert second["specialized"] and second["specialize_ms"] < max(200.0, 0.5 * first["specialize_ms"]) or first["specialize_ms"] < 400.0
    _assert_same(_images(gpu_ctx, p), want_cornell)
    gpu_ctx.specialize("off")


@pytest.mark.gpu
def test_the_entry_point_that_allocates_nothing_uses_the_compiled_kernel_too(gpu_ctx):
    b, flat, _ = host.build_config_scene("cornell")
    gpu_ctx.upload(flat)
    p = device.make_params(200, 150, 16, n_slices=4, max_depth=16, seed=8)
    dev = torch.device("cuda", 0)

    def run(fn):
        sums = torch.full((4, 150, 200, 3), -7.0, dtype=torch.float32, device=dev)
        fn(p, sums.data_ptr(), None, None, None, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return sums.cpu().numpy()

    want = run(gpu_ctx.render_device)
    assert gpu_ctx.specialize("auto")["specialized"]
    assert np.array_equal(run(gpu_ctx.render_device), want)
    assert np.array_equal(run(gpu_ctx.render_device_noalloc), want)
    gpu_ctx.specialize("off")


@pytest.mark.gpu
def test_in_process_multi_device_route_specializes_every_device():
    b, flat, _ = host.build_config_scene("cornell")
    mc = device.MultiContext([0, 0])
    mc.upload(flat)
    job = device.make_params(96, 64, 16, n_slices=4, max_depth=12, seed=6)
    want = mc.render(job, abi.SPLIT_ROWS)
    mc.specialize("always")
    got = mc.render(job, abi.SPLIT_ROWS)
    for k in ("rgb8", "resolved", "linear"):
        assert np.array_equal(want[k], got[k], equal_nan=True)
    mc.close()


@pytest.mark.gpu
def test_long_programs_are_left_to_the_interpreter(gpu_ctx):
    """A scene of 300 top-level spheres is a program of 300 ops: AUTO declines (its limit is 64 unrolled ops), ALWAYS reports
    the limit (256) instead of starting a compile of that size; a 40-op scene is compiled and renders the same bits."""
    def scene(n):
        b = host.SceneBuilder(5)
        m = b.lambertian(b.solid_color((0.6, 0.5, 0.4)))
        rng = np.random.default_rng(n)
        for c in rng.uniform(-20, 20, (n, 3)):
            b.add(b.sphere(c, 1.5, m))
        b.set_lights(b.xzrect(-5, 5, -5, 5, 30.0, b.emissive((4, 4, 4))))
        b.set_camera((0, 0, -70), (0, 0, 0), (0, 1, 0), 40.0, 0.0, 10.0, 1.0)
        b.set_background((0.5, 0.6, 0.9))
        return b, b.flatten()

    b, flat = scene(300)
    gpu_ctx.upload(flat)
    assert not gpu_ctx.specialize("auto")["specialized"]
    with pytest.raises(Exception, match="not unrolled"):
        gpu_ctx.specialize("always")
    assert not gpu_ctx.info()["specialized"]
    b, flat = scene(40)
    gpu_ctx.upload(flat)
    p = device.make_params(64, 48, 8, n_slices=2, max_depth=8, seed=1)
    want = _images(gpu_ctx, p)
    assert gpu_ctx.specialize("auto")["specialized"]
    _assert_same(_images(gpu_ctx, p), want)
    gpu_ctx.specialize("off")

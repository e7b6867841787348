"""examples/render_scene.c: both C ABIs used from plain C, without ctypes in between.

CPU: the program builds with gcc -std=c99 -pedantic against the headers, reports usage errors, and fails loudly
when there is no GPU (the path has no CPU fallback).  GPU: its image is byte-identical to the one the ctypes
path produces for the same scene, parameters and seed — the struct layouts a C compiler sees are the ones the
Python mirror declares.
"""
import os
import subprocess

import numpy as np
import pytest

import rust_raytracer_amd  # noqa: F401
from rust_raytracer_amd import abi, device, host

EXE = os.path.join(abi.REPO_DIR, "examples", "render_scene")


def run(*args):
    return subprocess.run([EXE, *map(str, args)], capture_output=True, text=True, timeout=300)


def read_ppm(path):
    data = open(path, "rb").read()
    magic, dims, maxval, rest = data.split(b"\n", 3)
    assert magic == b"P6" and maxval == b"255"
    w, h = map(int, dims.split())
    return np.frombuffer(rest, dtype=np.uint8).reshape(h, w, 3)


def test_example_builds_and_reports_errors(built, tmp_path):
    assert os.access(EXE, os.X_OK)
    r = run()
    assert r.returncode == 1 and "usage" in r.stderr
    r = run("no_such_scene", tmp_path / "x.ppm")
    assert r.returncode == 2 and "rrh_build_preset" in r.stderr and "no_such_scene" in r.stderr
    if device.device_count() == 0:
        r = run("cornell_box", tmp_path / "x.ppm", 16, 16, 8)
        assert r.returncode == 2 and "no CPU path" in r.stderr
        assert not os.path.exists(tmp_path / "x.ppm")


@pytest.mark.gpu
@pytest.mark.parametrize("scene,W,H,spp,depth,slices,seed", [("cornell_box", 64, 48, 32, 12, 8, 5), ("random_spheres", 80, 45, 16, 10, 4, 9)])
def test_example_image_equals_ctypes_path(gpu_ctx, tmp_path, scene, W, H, spp, depth, slices, seed):
    out = tmp_path / "c.ppm"
    r = run(scene, out, W, H, spp, depth, slices, seed)
    assert r.returncode == 0, r.stderr
    got = read_ppm(out)
    b = host.SceneBuilder(seed)
    b.preset(scene, repair=True)
    gpu_ctx.upload(b.flatten())
    res = gpu_ctx.render(device.make_params(W, H, spp, n_slices=slices, max_depth=depth, seed=seed))
    assert got.shape == (H, W, 3)
    assert np.array_equal(got, res["rgb8"])
    assert got.max() > 0
    # (the example asks for RR_SPECIALIZE_AUTO: the Cornell box, a program without BVH subtrees, was rendered by the kernel
    # compiled for it; the ctypes render above by the interpreter)
    assert ("compiled for this scene's program" in r.stderr) == (scene == "cornell_box"), r.stderr

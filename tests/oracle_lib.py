"""ctypes face of oracle/librr_oracle.so — the CPU checker.  Test infrastructure only."""
import ctypes as C
import os

import numpy as np

import rust_raytracer_amd  # noqa: F401
from rust_raytracer_amd import abi

ORACLE_LIB = os.path.join(abi.REPO_DIR, "oracle", "librr_oracle.so")
_lib = None

f3 = C.POINTER(C.c_float)
_PROTOS = {
    "rro_philox4x32_10": (None, [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rro_stream_floats": (None, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, f3]),
    "rro_aabb_hit": (C.c_int, [f3, f3, f3, C.c_float, C.c_float]),
    "rro_perlin_noise": (C.c_float, [C.POINTER(abi.Perlin), f3]),
    "rro_perlin_turb": (C.c_float, [C.POINTER(abi.Perlin), f3, C.c_uint32]),
    "rro_texture_color": (None, [C.POINTER(abi.FlatScene), C.c_uint32, C.c_float, C.c_float, f3, f3]),
    "rro_light_pdf_value": (C.c_float, [C.POINTER(abi.FlatScene), f3, f3]),
    "rro_camera_ray": (None, [C.POINTER(abi.FlatScene), C.POINTER(abi.RenderParams), C.c_uint32, C.c_uint32, C.c_uint32, f3]),
    "rro_primary_hits": (C.c_int, [C.POINTER(abi.FlatScene), C.c_uint32, f3, C.c_float, C.c_float, C.c_uint64,
                                   C.POINTER(abi.Hit), C.POINTER(abi.Counters)]),
    "rro_render": (C.c_int, [C.POINTER(abi.FlatScene), C.POINTER(abi.RenderParams), f3, f3, f3, C.POINTER(abi.Counters),
                             C.c_int]),
    "rro_cast_u8": (None, [f3, C.POINTER(C.c_uint8), C.c_uint64]),
}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_LIB):
            raise RuntimeError(f"{ORACLE_LIB} missing: run `make -C oracle`")
        _lib = abi.bind(C.CDLL(ORACLE_LIB), _PROTOS)
    return _lib


HIT_DTYPE = np.dtype([("prim_id", "<i4"), ("material", "<u4"), ("t", "<f4"), ("point", "<f4", 3), ("normal", "<f4", 3),
                      ("u", "<f4"), ("v", "<f4"), ("front_face", "<u4")])
assert HIT_DTYPE.itemsize == C.sizeof(abi.Hit)


def fptr(a):
    return a.ctypes.data_as(f3)


def primary_hits(flat, rays, t_min=0.001, t_max=3.40282347e38, seed=0, counters=False):
    rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 7)
    out = np.zeros(len(rays), dtype=HIT_DTYPE)
    cnt = abi.Counters()
    rc = lib().rro_primary_hits(C.byref(flat), len(rays), fptr(rays), t_min, t_max, seed,
                                out.ctypes.data_as(C.POINTER(abi.Hit)), C.byref(cnt) if counters else None)
    assert rc == 0
    return (out, cnt.as_dict()) if counters else out


def make_params(width, height, spp, n_slices=1, max_depth=50, seed=1, slice_begin=0, slice_end=None, sample_offset=0,
                flags=0, row_begin=0, row_end=0):
    return abi.RenderParams(width, height, spp, n_slices, slice_begin, n_slices if slice_end is None else slice_end,
                            max_depth, seed, sample_offset, flags, row_begin, row_end)


def render(flat, params, n_threads=8, counters=False):
    """→ dict(slice_sums[K',H,W,3], resolved[H,W,3], linear[H,W,3], rgb8[H,W,3], counters)"""
    k = params.slice_end - params.slice_begin
    h, w = params.rows, params.width
    ss = np.zeros((k, h, w, 3), dtype=np.float32)
    res = np.zeros((h, w, 3), dtype=np.float32)
    lin = np.zeros((h, w, 3), dtype=np.float32)
    cnt = abi.Counters()
    rc = lib().rro_render(C.byref(flat), C.byref(params), fptr(ss), fptr(res), fptr(lin),
                          C.byref(cnt) if counters else None, n_threads)
    assert rc == 0
    rgb = np.zeros((h, w, 3), dtype=np.uint8)
    lib().rro_cast_u8(fptr(res), rgb.ctypes.data_as(C.POINTER(C.c_uint8)), res.size)
    return dict(slice_sums=ss, resolved=res, linear=lin, rgb8=rgb, counters=cnt.as_dict())


def cast_u8(resolved):
    """The saturating `as u8` of render.rs:156-159 on a summed image."""
    res = np.ascontiguousarray(resolved, dtype=np.float32)
    rgb = np.zeros(res.shape, dtype=np.uint8)
    lib().rro_cast_u8(fptr(res), rgb.ctypes.data_as(C.POINTER(C.c_uint8)), res.size)
    return rgb


def camera_rays(flat, params, pixels, sample=0):
    """Camera rays the renderer would shoot for (i, j_from_bottom) pixels."""
    out = np.zeros((len(pixels), 7), dtype=np.float32)
    for n, (i, j) in enumerate(pixels):
        lib().rro_camera_ray(C.byref(flat), C.byref(params), i, j, sample, fptr(out[n]))
    return out

"""The HIP path behind the C ABI (librr_hip.so).

There is no CPU fallback: a missing library or a missing GPU raises.  The functions
here mirror the reference's render seam (render_multi, src/render.rs:128) — upload
the flattened scene once, then render — plus the primary-hit debug entry.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_lib = None


class DeviceError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(abi.HIP_LIB):
            raise DeviceError(f"{abi.HIP_LIB} is missing: the HIP extension was not built (no CPU fallback exists)")
        _lib = abi.bind(C.CDLL(abi.HIP_LIB), abi.HIP_PROTOTYPES)
    return _lib


def _check(rc):
    if rc != 0:
        raise DeviceError(f"rr_hip error {rc}: {lib().rr_last_error().decode()}")


def device_count():
    n = C.c_int(0)
    _check(lib().rr_device_count(C.byref(n)))
    return n.value


HIT_DTYPE = np.dtype([("prim_id", "<i4"), ("material", "<u4"), ("t", "<f4"), ("point", "<f4", 3), ("normal", "<f4", 3),
                      ("u", "<f4"), ("v", "<f4"), ("front_face", "<u4")])


def make_params(width, height, spp, n_slices=1, max_depth=50, seed=1, slice_begin=0, slice_end=None, sample_offset=0,
                flags=0, row_begin=0, row_end=0):
    return abi.RenderParams(width, height, spp, n_slices, slice_begin, n_slices if slice_end is None else slice_end,
                            max_depth, seed, sample_offset, flags, row_begin, row_end)


class Context:
    """One GPU (one process per GPU): owns the uploaded scene."""

    def __init__(self, device=0):
        self._l = lib()
        h = C.c_void_p()
        _check(self._l.rr_context_create(device, C.byref(h)))
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._l.rr_context_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, flat):
        _check(self._l.rr_scene_upload(self._h, C.byref(flat)))
        return self.info()

    def info(self):
        i = abi.SceneInfo()
        _check(self._l.rr_scene_info_get(self._h, C.byref(i)))
        return dict(stack_words=i.stack_words, max_inst_depth=i.max_inst_depth, device_bytes=i.device_bytes,
                    n_prims=i.n_prims, wavefront=bool(i.wavefront), bvh_nodes_by_class=list(i.bvh_nodes_by_class),
                    subtree_ops=list(i.subtree_ops), wide_nodes=bool(i.wide_nodes),
                    workspace_bytes=i.workspace_bytes, specialized=bool(i.specialized), specialize_ms=float(i.specialize_ms))

    SPECIALIZE = {"auto": abi.SPECIALIZE_AUTO, "always": abi.SPECIALIZE_ALWAYS, "off": abi.SPECIALIZE_OFF}

    def specialize(self, mode="auto"):
        """rr_scene_specialize: compile the path kernel for the uploaded scene's program ("auto": only where it is expected to
        pay, "always", "off": back to the interpreter).  Returns info()."""
        _check(self._l.rr_scene_specialize(self._h, self.SPECIALIZE[mode]))
        return self.info()

    def primary_hits(self, rays, t_min=0.001, t_max=3.40282347e38, seed=0):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 7)
        out = np.zeros(len(rays), dtype=HIT_DTYPE)
        _check(self._l.rr_primary_hits(self._h, len(rays), rays.ctypes.data_as(C.POINTER(C.c_float)), t_min, t_max, seed,
                                       out.ctypes.data_as(C.POINTER(abi.Hit))))
        return out

    def eval_textures(self, textures, uvp, what=abi.EVAL_COLOR):
        """Texture::get_color / Perlin::turb / Perlin::noise on the device for n x (u, v, px, py, pz) -> n x 3."""
        tex = np.ascontiguousarray(textures, dtype=np.uint32).reshape(-1)
        uvp = np.ascontiguousarray(uvp, dtype=np.float32).reshape(-1, 5)
        assert len(tex) == len(uvp)
        out = np.zeros((len(tex), 3), dtype=np.float32)
        fp = C.POINTER(C.c_float)
        _check(self._l.rr_eval_textures(self._h, len(tex), tex.ctypes.data_as(C.POINTER(C.c_uint32)), uvp.ctypes.data_as(fp), what,
                                        out.ctypes.data_as(fp)))
        return out

    def render(self, params, slice_sums=True):
        """Synchronous render into host arrays → dict(rgb8, resolved, linear, slice_sums, counters)."""
        h, w = max(0, params.rows), params.width  # (a reversed row range is the library's to report)
        k = max(0, params.slice_end - params.slice_begin)
        rgb = np.zeros((h, w, 3), dtype=np.uint8)
        res = np.zeros((h, w, 3), dtype=np.float32)
        lin = np.zeros((h, w, 3), dtype=np.float32)
        ss = np.zeros((k, h, w, 3), dtype=np.float32) if slice_sums else None
        cnt = abi.Counters()
        fp = C.POINTER(C.c_float)
        _check(self._l.rr_render(self._h, C.byref(params), rgb.ctypes.data_as(C.POINTER(C.c_uint8)), res.ctypes.data_as(fp),
                                 lin.ctypes.data_as(fp), ss.ctypes.data_as(fp) if slice_sums else None, C.byref(cnt)))
        return dict(rgb8=rgb, resolved=res, linear=lin, slice_sums=ss, counters=cnt.as_dict())

    def render_device(self, params, d_slice_sums, d_resolved=None, d_linear=None, d_counters=None, stream=None):
        """Asynchronous render into caller-owned DEVICE buffers (raw addresses / torch data_ptr())."""
        _check(self._l.rr_render_device(self._h, C.byref(params), d_slice_sums, d_resolved, d_linear, d_counters, stream))

    def render_device_noalloc(self, params, d_slice_sums, d_resolved=None, d_linear=None, d_counters=None, stream=None):
        """render_device without any device allocation inside the library (static 64 MiB workspace, path-granular kernel)."""
        _check(self._l.rr_render_device_noalloc(self._h, C.byref(params), d_slice_sums, d_resolved, d_linear, d_counters, stream))

    def cast_u8_device(self, d_resolved, d_rgb8, n, stream=None):
        _check(self._l.rr_cast_u8_device(self._h, d_resolved, d_rgb8, n, stream))

    def last_kernel_ms(self):
        ms = C.c_float(0)
        _check(self._l.rr_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value


class MultiContext:
    """Several GPUs driven from this one process (rr_multi_*): the in-process counterpart of rust_raytracer_amd.multi."""

    REDUCE = {"auto": abi.REDUCE_AUTO, "peer": abi.REDUCE_PEER, "rccl": abi.REDUCE_RCCL}

    def __init__(self, devices, reduce="auto"):
        self._l = lib()
        arr = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        _check(self._l.rr_multi_create(len(devices), arr, C.byref(h)))
        self._h = h
        self.devices = list(devices)
        if reduce != "auto":
            self.set_reduce(reduce)

    def set_reduce(self, mode):
        """How a band's slice groups are summed: "auto" (RCCL when it applies, else peer copies), "peer", "rccl" (or fail)."""
        _check(self._l.rr_multi_set_reduce(self._h, self.REDUCE[mode]))

    def info(self):
        used = C.c_uint32(0)
        _check(self._l.rr_multi_reduce_used(self._h, C.byref(used)))
        return dict(reduce={abi.REDUCE_PEER: "peer", abi.REDUCE_RCCL: "rccl"}[used.value])

    def close(self):
        if getattr(self, "_h", None):
            self._l.rr_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, flat):
        _check(self._l.rr_multi_scene_upload(self._h, C.byref(flat)))

    def specialize(self, mode="auto"):
        _check(self._l.rr_multi_scene_specialize(self._h, Context.SPECIALIZE[mode]))

    def render(self, job, split=abi.SPLIT_SLICES):
        """The whole job `job` (all slices, all rows) -> dict(rgb8, resolved, linear, counters) in host arrays."""
        h, w = job.height, job.width
        rgb = np.zeros((h, w, 3), dtype=np.uint8)
        res = np.zeros((h, w, 3), dtype=np.float32)
        lin = np.zeros((h, w, 3), dtype=np.float32)
        cnt = abi.Counters()
        fp = C.POINTER(C.c_float)
        _check(self._l.rr_multi_render(self._h, C.byref(job), split, rgb.ctypes.data_as(C.POINTER(C.c_uint8)), res.ctypes.data_as(fp),
                                       lin.ctypes.data_as(fp), C.byref(cnt)))
        return dict(rgb8=rgb, resolved=res, linear=lin, counters=cnt.as_dict())


def render_hip(flat, background_unused=None, *, max_depth, samples_per_pixel, img_width, img_height, n_slices=8, seed=1,
               device=0):
    """Drop-in for render_multi (src/render.rs:128): flattened scene in, H x W x 3 u8 image out.

    The scene, lights, camera and background travel inside `flat`; n_slices plays the
    role of the reference's num_cpus::get() (render.rs:129).
    """
    ctx = Context(device)
    try:
        ctx.upload(flat)
        p = make_params(img_width, img_height, samples_per_pixel, n_slices, max_depth, seed)
        return ctx.render(p, slice_sums=False)["rgb8"]
    finally:
        ctx.close()

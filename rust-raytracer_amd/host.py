"""Host-side scene construction: Python face of librr_host.so.

Method names follow the reference's constructors (src/collider.rs:56-267,
src/material.rs:58-82, src/texture.rs:41-64, src/camera.rs:34) so that scene code
reads like src/main.rs.  Nothing here runs on the GPU and nothing intersects rays.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(abi.HOST_LIB):
            raise RuntimeError(f"{abi.HOST_LIB} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _lib = abi.bind(C.CDLL(abi.HOST_LIB), abi.HOST_PROTOTYPES)
    return _lib


class HostError(RuntimeError):
    pass


AXIS_X, AXIS_Y, AXIS_Z = 0, 1, 2


class SceneBuilder:
    """One scene under construction (SceneColliders + lights + camera + background)."""

    def __init__(self, seed=1, bvh="reference"):
        """bvh: "reference" = the reference's tree (collider.rs:173-212); "sah" = a surface-area-heuristic tree, a labelled
        option that is NOT the reference's (meant for the intended / book semantics, where a BVH only accelerates)."""
        self._l = lib()
        self._b = self._l.rrh_builder_new(seed)
        if not self._b:
            raise HostError("rrh_builder_new failed")
        self._flat = None
        if bvh != "reference":
            self._h(self._l.rrh_set_bvh_builder(self._b, {"sah": 1}[bvh]))

    def close(self):
        if self._b:
            self._l.rrh_builder_free(self._b)
            self._b = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _h(self, rc):
        if rc < 0:
            raise HostError(self._l.rrh_last_error().decode())
        return rc

    # utils.rs:41-55
    def random(self):
        return self._l.rrh_random(self._b)

    def randrange(self, lo, hi):
        return self._l.rrh_randrange(self._b, lo, hi)

    def randuint(self, lo, hi):
        return self._l.rrh_randuint(self._b, lo, hi)

    # textures
    def solid_color(self, rgb):
        return self._h(self._l.rrh_tex_solid_color(self._b, *map(float, rgb)))

    def checkered(self, odd, even):
        return self._h(self._l.rrh_tex_checkered(self._b, abi.vec3(odd), abi.vec3(even)))

    def noise(self, scale, turb):
        return self._h(self._l.rrh_tex_noise(self._b, scale, turb))

    def load_image_rgb8(self, array):
        a = np.ascontiguousarray(array, dtype=np.uint8)
        if a.ndim != 3 or a.shape[2] != 3:
            raise ValueError("image must be H x W x 3 uint8")
        return self._h(self._l.rrh_load_image_rgb8(self._b, a.shape[1], a.shape[0], a.ctypes.data_as(C.POINTER(C.c_uint8))))

    def load_image(self, path):
        """SceneColliders::load_image (collider.rs:747): PPM natively, anything PIL reads otherwise."""
        if path.lower().endswith(".ppm"):
            return self._h(self._l.rrh_load_image_ppm(self._b, path.encode()))
        from PIL import Image
        return self.load_image_rgb8(np.asarray(Image.open(path).convert("RGB")))

    # materials
    def lambertian(self, texture):
        return self._h(self._l.rrh_mat_lambertian(self._b, texture))

    def glossy(self, rgb, fuzz):
        return self._h(self._l.rrh_mat_glossy(self._b, *map(float, rgb), fuzz))

    def dielectric(self, ior):
        return self._h(self._l.rrh_mat_dielectric(self._b, ior))

    def isotropic(self, rgb):
        return self._h(self._l.rrh_mat_isotropic(self._b, *map(float, rgb)))

    def emissive(self, rgb):
        return self._h(self._l.rrh_mat_emissive(self._b, *map(float, rgb)))

    # geometry
    def sphere(self, center, radius, material):
        return self._h(self._l.rrh_sphere(self._b, abi.vec3(center), radius, material))

    def moving_sphere(self, c0, c1, time0, time1, radius, material):
        return self._h(self._l.rrh_moving_sphere(self._b, abi.vec3(c0), abi.vec3(c1), time0, time1, radius, material))

    def xyrect(self, x0, x1, y0, y1, k, material):
        return self._h(self._l.rrh_xyrect(self._b, x0, x1, y0, y1, k, material))

    def xzrect(self, x0, x1, z0, z1, k, material):
        return self._h(self._l.rrh_xzrect(self._b, x0, x1, z0, z1, k, material))

    def yzrect(self, y0, y1, z0, z1, k, material):
        return self._h(self._l.rrh_yzrect(self._b, y0, y1, z0, z1, k, material))

    def cuboid(self, p0, p1, material):
        return self._h(self._l.rrh_cuboid(self._b, abi.vec3(p0), abi.vec3(p1), material))

    def constant_medium(self, boundary, density, rgb):
        return self._h(self._l.rrh_constant_medium(self._b, boundary, density, *map(float, rgb)))

    def instance_translation(self, geometry, displacement):
        return self._h(self._l.rrh_instance_translation(self._b, geometry, abi.vec3(displacement)))

    def instance_rotation(self, geometry, axis, angle):
        return self._h(self._l.rrh_instance_rotation(self._b, geometry, axis, angle))

    def bvh_node(self, geometries, time_0=0.0, time_1=1.0):
        arr = (C.c_int * len(geometries))(*geometries)
        return self._h(self._l.rrh_bvh_node(self._b, arr, len(geometries), time_0, time_1))

    def bvh_node_explicit(self, left, right, axis, box=None):
        if box is None:
            z = abi.vec3((0, 0, 0))
            return self._h(self._l.rrh_bvh_node_explicit(self._b, left, right, axis, 0, z, z))
        return self._h(self._l.rrh_bvh_node_explicit(self._b, left, right, axis, 1, abi.vec3(box[0]), abi.vec3(box[1])))

    def collider_list(self, geometries):
        arr = (C.c_int * max(1, len(geometries)))(*geometries)
        return self._h(self._l.rrh_collider_list(self._b, arr, len(geometries)))

    def triangle(self, p0, p1, p2, material):
        return self._h(self._l.rrh_triangle(self._b, abi.vec3(p0), abi.vec3(p1), abi.vec3(p2), material))

    def load_obj(self, path, scale, material, hardened=False):
        """Geometry::load_obj (collider.rs:227-267).  hardened=True is not the reference's reader: "vn" lines are not
        vertices, indices may be negative, polygons become fans."""
        f = self._l.rrh_load_obj_hardened if hardened else self._l.rrh_load_obj
        return self._h(f(self._b, path.encode(), scale, material))

    def bounding_box(self, geometry, time_0=0.0, time_1=1.0):
        out = (C.c_float * 6)()
        rc = self._h(self._l.rrh_bounding_box(self._b, geometry, time_0, time_1, out))
        if rc == 0:
            return None
        return np.array(out[:3], dtype=np.float32), np.array(out[3:], dtype=np.float32)

    # scene assembly
    def add(self, geometry):
        self._h(self._l.rrh_scene_add(self._b, geometry))

    def set_lights(self, geometry):
        self._h(self._l.rrh_set_lights(self._b, geometry))

    def set_camera(self, look_from, look_at, vup, vfov, aperture, focus_dist, aspect_ratio, shutter_open=0.0,
                   shutter_close=1.0):
        self._h(self._l.rrh_set_camera(self._b, abi.vec3(look_from), abi.vec3(look_at), abi.vec3(vup), vfov, aperture,
                                       focus_dist, aspect_ratio, shutter_open, shutter_close))

    def set_background(self, rgb):
        self._h(self._l.rrh_set_background(self._b, *map(float, rgb)))

    def preset(self, name, repair=False, moving_sphere=False, obj_path=None, image=-1, n_triangles=0, n_spheres=0):
        """One of the scene functions of src/main.rs (or "synthetic_mesh"); returns its default image settings."""
        opt = abi.PresetOptions(int(repair), int(moving_sphere), obj_path.encode() if obj_path else None, image,
                                n_triangles, n_spheres)
        d = abi.PresetDefaults()
        self._h(self._l.rrh_build_preset(self._b, name.encode(), C.byref(opt), C.byref(d)))
        return dict(width=d.image_width, height=d.image_height, samples_per_pixel=d.samples_per_pixel,
                    max_depth=d.max_depth, aspect_ratio=d.aspect_ratio)

    def flatten(self):
        """Flatten into an abi.FlatScene view (valid while this builder lives and until the next flatten)."""
        fs = abi.FlatScene()
        self._h(self._l.rrh_flatten(self._b, C.byref(fs)))
        fs._owner = self
        self._flat = fs
        return fs


ASSETS = os.path.join(abi.REPO_DIR, "assets")


def build_config_scene(name, seed=1, **kw):
    """Scenes of the benchmark configurations by short name → (builder, FlatScene, defaults).
    bvh="sah": every BVH of the scene is a surface-area-heuristic tree instead of the reference's (a labelled option).

    cornell        cornell_box() verbatim (main.rs:325-368)
    spheres        random_spheres() (main.rs:50-115)
    suzanne        cornell_pedestal() + light repair, Suzanne mesh at scale 80 (main.rs:416-464)
    final          final_scene() + light repair + the motion-blurred sphere (main.rs:466-541)
    mesh           synthetic height field + spheres (n_triangles, n_spheres keyword arguments)
    """
    b = SceneBuilder(seed, bvh=kw.pop("bvh", "reference"))
    if name == "cornell":
        d = b.preset("cornell_box")
    elif name == "spheres":
        d = b.preset("random_spheres")
    elif name == "suzanne":
        d = b.preset("cornell_pedestal", repair=kw.get("repair", True), obj_path=os.path.join(ASSETS, "suzanne.obj"))
    elif name == "final":
        img = b.load_image(os.path.join(ASSETS, "earthmap.ppm"))
        d = b.preset("final_scene", repair=kw.get("repair", True), moving_sphere=kw.get("moving_sphere", True), image=img)
    elif name == "mesh":
        d = b.preset("synthetic_mesh", n_triangles=kw.get("n_triangles", 1_000_000), n_spheres=kw.get("n_spheres", 100_000))
    else:
        d = b.preset(name, **kw)
    return b, b.flatten(), d

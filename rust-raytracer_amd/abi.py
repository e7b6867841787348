"""ctypes mirror of include/rr_hip.h and include/rr_host.h.

Only layout and prototypes live here; behaviour is in the shared libraries.
"""
import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)

RR_OK = 0
RR_E_INVALID, RR_E_NO_DEVICE, RR_E_HIP, RR_E_NO_SCENE, RR_E_LIMIT = -1, -2, -3, -4, -5

(TAG_SPHERE, TAG_MOVING_SPHERE, TAG_RECT, TAG_TRIANGLE, TAG_CUBOID, TAG_MEDIUM, TAG_INSTANCE, TAG_BVH,
 TAG_LIST) = range(9)
REF_NONE = 0xFFFFFFFF
RECT_XY, RECT_XZ, RECT_YZ = 0, 1, 2
INST_TRANSLATE, INST_ROTATE_X, INST_ROTATE_Y, INST_ROTATE_Z = 0, 1, 2, 3
MAT_LAMBERTIAN, MAT_GLOSSY, MAT_DIELECTRIC, MAT_ISOTROPIC, MAT_EMISSIVE = 0, 1, 2, 3, 4
TEX_SOLID, TEX_CHECKERED, TEX_NOISE, TEX_IMAGE = 0, 1, 2, 3
FLAG_COUNTERS = 1
FLAG_PERSISTENT = 2
FLAG_WAVEFRONT = 4
FLAG_NO_FINISH_KERNEL = 8
FLAG_PATHS = 16
FLAG_COUNT_PRODUCTION = 32
FLAG_NARROW_NODES = 64
FLAG_TRACE_DEAD_PATHS = 128
FLAG_ROW_ORDER = 256
N_COUNTERS = 10  # u64 words of rr_counters
PERLIN_POINTS = 256
NOALLOC_MAX_PATHS = 4194304
EVAL_COLOR, EVAL_TURB, EVAL_NOISE = 0, 1, 2
SPLIT_SLICES, SPLIT_ROWS, SPLIT_BOTH = 0, 1, 2
REDUCE_AUTO, REDUCE_PEER, REDUCE_RCCL = 0, 1, 2
SPECIALIZE_AUTO, SPECIALIZE_ALWAYS, SPECIALIZE_OFF = 0, 1, 2
SEMANTICS_REFERENCE, SEMANTICS_INTENDED, SEMANTICS_BOOK_ESTIMATOR, SEMANTICS_BOOK = 0, 1, 2, 3


def ref(tag, index):
    return (tag << 28) | index


def ref_tag(r):
    return r >> 28


def ref_index(r):
    return r & 0x0FFFFFFF


f32, u32, u64, i32, u8 = C.c_float, C.c_uint32, C.c_uint64, C.c_int32, C.c_uint8


class Sphere(C.Structure):
    _fields_ = [("center", f32 * 3), ("radius", f32), ("material", u32), ("prim_id", u32)]


class MovingSphere(C.Structure):
    _fields_ = [("center0", f32 * 3), ("center1", f32 * 3), ("time0", f32), ("time1", f32), ("radius", f32),
                ("material", u32), ("prim_id", u32)]


class Rect(C.Structure):
    _fields_ = [("a0", f32), ("a1", f32), ("b0", f32), ("b1", f32), ("k", f32), ("axis", u32), ("material", u32),
                ("prim_id", u32)]


class Triangle(C.Structure):
    _fields_ = [("p0", f32 * 3), ("p1", f32 * 3), ("p2", f32 * 3), ("normal", f32 * 3), ("material", u32),
                ("prim_id", u32)]


class Cuboid(C.Structure):
    _fields_ = [("first_rect", u32)]


class Medium(C.Structure):
    _fields_ = [("boundary", u32), ("neg_inv_density", f32), ("material", u32), ("prim_id", u32)]


class Instance(C.Structure):
    _fields_ = [("kind", u32), ("child", u32), ("v", f32 * 3)]


class BvhNode(C.Structure):
    _fields_ = [("bmin", f32 * 3), ("bmax", f32 * 3), ("left", u32), ("right", u32), ("axis", u32), ("has_box", u32)]


class List(C.Structure):
    _fields_ = [("first_child", u32), ("n_children", u32)]


class Material(C.Structure):
    _fields_ = [("type", u32), ("texture", u32), ("color", f32 * 3), ("param", f32)]


class Texture(C.Structure):
    _fields_ = [("type", u32), ("color0", f32 * 3), ("color1", f32 * 3), ("index", u32), ("scale", f32), ("depth", u32)]


class Perlin(C.Structure):
    _fields_ = [("ranvec", (f32 * 3) * PERLIN_POINTS), ("perm_x", u8 * PERLIN_POINTS), ("perm_y", u8 * PERLIN_POINTS),
                ("perm_z", u8 * PERLIN_POINTS)]


class Image(C.Structure):
    _fields_ = [("width", u32), ("height", u32), ("offset", u64)]


class Light(C.Structure):
    _fields_ = [("is_xzrect", u32), ("x0", f32), ("x1", f32), ("z0", f32), ("z1", f32), ("k", f32)]


class Camera(C.Structure):
    _fields_ = [("origin", f32 * 3), ("lower_left", f32 * 3), ("horizontal", f32 * 3), ("vertical", f32 * 3),
                ("u", f32 * 3), ("v", f32 * 3), ("lens_radius", f32), ("shutter_open", f32), ("shutter_close", f32)]


def _arr(name, typ):
    return [("n_" + name, u32), (name, C.POINTER(typ))]


class FlatScene(C.Structure):
    _fields_ = (_arr("spheres", Sphere) + _arr("moving_spheres", MovingSphere) + _arr("rects", Rect) +
                _arr("triangles", Triangle) + _arr("cuboids", Cuboid) + _arr("media", Medium) +
                _arr("instances", Instance) + _arr("bvh_nodes", BvhNode) + _arr("lists", List) +
                _arr("child_refs", u32) + _arr("materials", Material) + _arr("textures", Texture) +
                _arr("perlin", Perlin) + _arr("images", Image) +
                [("n_image_bytes", u64), ("image_bytes", C.POINTER(u8)), ("root", u32), ("light", Light),
                 ("camera", Camera), ("background", f32 * 3), ("semantics", u32), ("n_lights", u32), ("lights", C.POINTER(Light))])


class RenderParams(C.Structure):
    _fields_ = [("width", u32), ("height", u32), ("samples_per_pixel", u32), ("n_slices", u32), ("slice_begin", u32),
                ("slice_end", u32), ("max_depth", u32), ("seed", u64), ("sample_offset", u32), ("flags", u32),
                ("row_begin", u32), ("row_end", u32)]

    @property
    def rows(self):
        """Number of image rows the call renders (row_end = 0 stands for the image height)."""
        return (self.row_end or self.height) - self.row_begin


class Counters(C.Structure):
    _fields_ = [("samples", u64), ("rays", u64), ("box_tests", u64), ("prim_tests", u64), ("inst_visits", u64),
                ("sphere_tests", u64), ("rect_tests", u64), ("tri_tests", u64),
                ("machine_lane_steps", u64), ("machine_wave_steps", u64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class Hit(C.Structure):
    _fields_ = [("prim_id", i32), ("material", u32), ("t", f32), ("point", f32 * 3), ("normal", f32 * 3), ("u", f32),
                ("v", f32), ("front_face", u32)]


class SceneInfo(C.Structure):
    _fields_ = [("stack_words", u32), ("max_inst_depth", u32), ("device_bytes", u64), ("n_prims", u32),
                ("wavefront", u32), ("bvh_nodes_by_class", u32 * 4), ("subtree_ops", u32 * 2),
                ("wide_nodes", u32), ("workspace_bytes", u64), ("specialized", u32), ("specialize_ms", f32)]


class PresetOptions(C.Structure):
    _fields_ = [("repair", C.c_int), ("moving_sphere", C.c_int), ("obj_path", C.c_char_p), ("image", C.c_int),
                ("n_triangles", u32), ("n_spheres", u32)]


class PresetDefaults(C.Structure):
    _fields_ = [("image_width", u32), ("image_height", u32), ("samples_per_pixel", u32), ("max_depth", u32),
                ("aspect_ratio", f32)]


P = C.POINTER
vp = C.c_void_p
f3 = P(f32)

# name -> (restype, argtypes); the single list the symbol-export test walks.
HIP_PROTOTYPES = {
    "rr_device_count": (C.c_int, [P(C.c_int)]),
    "rr_context_create": (C.c_int, [C.c_int, P(vp)]),
    "rr_context_destroy": (C.c_int, [vp]),
    "rr_scene_upload": (C.c_int, [vp, P(FlatScene)]),
    "rr_scene_info_get": (C.c_int, [vp, P(SceneInfo)]),
    "rr_scene_specialize": (C.c_int, [vp, u32]),
    "rr_specialize_options": (C.c_char_p, []),
    "rr_specialize_selftest": (C.c_int, [f3]),
    "rr_primary_hits": (C.c_int, [vp, u32, f3, f32, f32, u64, P(Hit)]),
    "rr_eval_textures": (C.c_int, [vp, u32, P(u32), f3, u32, f3]),
    "rr_render_device": (C.c_int, [vp, P(RenderParams), vp, vp, vp, vp, vp]),
    "rr_render_device_noalloc": (C.c_int, [vp, P(RenderParams), vp, vp, vp, vp, vp]),
    "rr_render": (C.c_int, [vp, P(RenderParams), P(u8), f3, f3, f3, P(Counters)]),
    "rr_cast_u8_device": (C.c_int, [vp, vp, vp, u64, vp]),
    "rr_last_kernel_ms": (C.c_int, [vp, f3]),
    "rr_multi_create": (C.c_int, [C.c_int, P(C.c_int), P(vp)]),
    "rr_multi_destroy": (C.c_int, [vp]),
    "rr_multi_scene_upload": (C.c_int, [vp, P(FlatScene)]),
    "rr_multi_render": (C.c_int, [vp, P(RenderParams), u32, P(u8), f3, f3, P(Counters)]),
    "rr_multi_scene_specialize": (C.c_int, [vp, u32]),
    "rr_multi_set_reduce": (C.c_int, [vp, u32]),
    "rr_multi_reduce_used": (C.c_int, [vp, P(u32)]),
    "rr_last_error": (C.c_char_p, []),
}

HOST_PROTOTYPES = {
    "rrh_builder_new": (vp, [u64]),
    "rrh_builder_free": (None, [vp]),
    "rrh_set_bvh_builder": (C.c_int, [vp, u32]),
    "rrh_last_error": (C.c_char_p, []),
    "rrh_random": (f32, [vp]),
    "rrh_randrange": (f32, [vp, f32, f32]),
    "rrh_randuint": (u32, [vp, u32, u32]),
    "rrh_tex_solid_color": (C.c_int, [vp, f32, f32, f32]),
    "rrh_tex_checkered": (C.c_int, [vp, f3, f3]),
    "rrh_tex_noise": (C.c_int, [vp, f32, u32]),
    "rrh_load_image_rgb8": (C.c_int, [vp, u32, u32, P(u8)]),
    "rrh_load_image_ppm": (C.c_int, [vp, C.c_char_p]),
    "rrh_mat_lambertian": (C.c_int, [vp, C.c_int]),
    "rrh_mat_glossy": (C.c_int, [vp, f32, f32, f32, f32]),
    "rrh_mat_dielectric": (C.c_int, [vp, f32]),
    "rrh_mat_isotropic": (C.c_int, [vp, f32, f32, f32]),
    "rrh_mat_emissive": (C.c_int, [vp, f32, f32, f32]),
    "rrh_sphere": (C.c_int, [vp, f3, f32, C.c_int]),
    "rrh_moving_sphere": (C.c_int, [vp, f3, f3, f32, f32, f32, C.c_int]),
    "rrh_xyrect": (C.c_int, [vp, f32, f32, f32, f32, f32, C.c_int]),
    "rrh_xzrect": (C.c_int, [vp, f32, f32, f32, f32, f32, C.c_int]),
    "rrh_yzrect": (C.c_int, [vp, f32, f32, f32, f32, f32, C.c_int]),
    "rrh_cuboid": (C.c_int, [vp, f3, f3, C.c_int]),
    "rrh_constant_medium": (C.c_int, [vp, C.c_int, f32, f32, f32, f32]),
    "rrh_instance_translation": (C.c_int, [vp, C.c_int, f3]),
    "rrh_instance_rotation": (C.c_int, [vp, C.c_int, C.c_int, f32]),
    "rrh_bvh_node": (C.c_int, [vp, P(C.c_int), u32, f32, f32]),
    "rrh_bvh_node_explicit": (C.c_int, [vp, C.c_int, C.c_int, u32, C.c_int, f3, f3]),
    "rrh_collider_list": (C.c_int, [vp, P(C.c_int), u32]),
    "rrh_triangle": (C.c_int, [vp, f3, f3, f3, C.c_int]),
    "rrh_load_obj": (C.c_int, [vp, C.c_char_p, f32, C.c_int]),
    "rrh_load_obj_hardened": (C.c_int, [vp, C.c_char_p, f32, C.c_int]),
    "rrh_bounding_box": (C.c_int, [vp, C.c_int, f32, f32, f3]),
    "rrh_scene_add": (C.c_int, [vp, C.c_int]),
    "rrh_set_lights": (C.c_int, [vp, C.c_int]),
    "rrh_set_camera": (C.c_int, [vp, f3, f3, f3, f32, f32, f32, f32, f32, f32]),
    "rrh_set_background": (C.c_int, [vp, f32, f32, f32]),
    "rrh_build_preset": (C.c_int, [vp, C.c_char_p, P(PresetOptions), P(PresetDefaults)]),
    "rrh_flatten": (C.c_int, [vp, P(FlatScene)]),
}

# RR_HIP_LIB points experiments (A/B builds of the kernels) at another build of the same ABI.
HIP_LIB = os.environ.get("RR_HIP_LIB") or os.path.join(PKG_DIR, "librr_hip.so")
HOST_LIB = os.path.join(PKG_DIR, "librr_host.so")


def bind(lib, prototypes):
    for name, (restype, argtypes) in prototypes.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


def vec3(v):
    return (f32 * 3)(float(v[0]), float(v[1]), float(v[2]))

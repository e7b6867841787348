"""The C-ABI libraries load without a GPU and export exactly what include/*.h declares;
the ctypes mirror has the C layout.  No compute calls here."""
import ctypes as C
import os
import re
import subprocess

import pytest

import rust_raytracer_amd  # noqa: F401
from rust_raytracer_amd import abi

INCLUDE = os.path.join(abi.REPO_DIR, "include")


def declared_functions(header, prefix):
    text = open(os.path.join(INCLUDE, header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(" + prefix + r"\w+)\s*\(", text))


def test_hip_library_exports_every_declared_symbol(built):
    declared = declared_functions("rr_hip.h", "rr_")
    assert declared == set(abi.HIP_PROTOTYPES), declared ^ set(abi.HIP_PROTOTYPES)
    lib = C.CDLL(abi.HIP_LIB)
    for name in declared:
        assert getattr(lib, name) is not None


def test_host_library_exports_every_declared_symbol(built):
    declared = declared_functions("rr_host.h", "rrh_")
    assert declared == set(abi.HOST_PROTOTYPES), declared ^ set(abi.HOST_PROTOTYPES)
    lib = C.CDLL(abi.HOST_LIB)
    for name in declared:
        assert getattr(lib, name) is not None


def test_ctypes_layout_matches_c(built, tmp_path):
    structs = {"rr_sphere": abi.Sphere, "rr_moving_sphere": abi.MovingSphere, "rr_rect": abi.Rect,
               "rr_triangle": abi.Triangle, "rr_cuboid": abi.Cuboid, "rr_medium": abi.Medium, "rr_instance": abi.Instance,
               "rr_bvh_node": abi.BvhNode, "rr_list": abi.List, "rr_material": abi.Material, "rr_texture": abi.Texture,
               "rr_perlin": abi.Perlin, "rr_image": abi.Image, "rr_light": abi.Light, "rr_camera": abi.Camera,
               "rr_flat_scene": abi.FlatScene, "rr_render_params": abi.RenderParams, "rr_counters": abi.Counters,
               "rr_hit": abi.Hit, "rr_scene_info": abi.SceneInfo, "rrh_preset_options": abi.PresetOptions,
               "rrh_preset_defaults": abi.PresetDefaults}
    offsets = [("rr_flat_scene", "root"), ("rr_flat_scene", "camera"), ("rr_flat_scene", "background"),
               ("rr_flat_scene", "image_bytes"), ("rr_render_params", "seed"), ("rr_render_params", "flags"),
               ("rr_hit", "front_face"), ("rr_counters", "tri_tests")]
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "rr_host.h"', "int main(void){"]
    for name in structs:
        src.append(f'printf("S {name} %zu\\n", sizeof({name}));')
    for s, f in offsets:
        src.append(f'printf("O {s} {f} %zu\\n", offsetof({s}, {f}));')
    src.append("return 0;}")
    cfile = tmp_path / "layout.c"
    cfile.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", INCLUDE, "-o", str(exe), str(cfile)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    for line in out:
        parts = line.split()
        if not parts:
            continue
        if parts[0] == "S":
            assert C.sizeof(structs[parts[1]]) == int(parts[2]), parts
        else:
            assert getattr(structs[parts[1]], parts[2]).offset == int(parts[3]), parts


def test_device_entry_points_fail_loudly_without_a_gpu(built):
    """On a host without a GPU the device count is 0 and context creation is an error, never a fallback."""
    from rust_raytracer_amd import device
    if device.device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(device.DeviceError, match="no HIP device"):
        device.Context(0)

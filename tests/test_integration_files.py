"""The Rust binding under integration/ cannot be compiled here (no Rust toolchain), but it can be kept honest:
its declarations are generated from the ctypes mirror, which other tests check against the C header."""
import os
import re
import subprocess
import sys

import rust_raytracer_amd  # noqa: F401
from rust_raytracer_amd import abi, build

INTEGRATION = os.path.join(abi.REPO_DIR, "integration")


def test_generated_rust_declarations_are_current():
    r = subprocess.run([sys.executable, os.path.join(INTEGRATION, "gen_rust_sys.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, "integration/rr-hip-sys/src/lib.rs is stale: run python integration/gen_rust_sys.py\n" + r.stderr


def test_rust_declarations_cover_the_header():
    text = open(os.path.join(INTEGRATION, "rr-hip-sys", "src", "lib.rs")).read()
    header = open(os.path.join(abi.REPO_DIR, "include", "rr_hip.h")).read()
    declared = set(re.findall(r"pub fn (rr_\w+)\(", text))
    in_header = set(re.findall(r"^(?:int|const char\*) (rr_\w+)\(", header, flags=re.M))
    assert declared == in_header == set(abi.HIP_PROTOTYPES)
    structs = set(re.findall(r"pub struct (rr_\w+)", text))
    assert structs == set(re.findall(r"^} (rr_\w+);", header, flags=re.M)) | {"rr_context", "rr_multi"}
    for name in re.findall(r"^#define (RR_[A-Z_0-9]+) ", header, flags=re.M):
        if name not in ("RR_HIP_H",):
            assert f"pub const {name}:" in text, name


def test_build_rs_uses_the_library_flags():
    text = open(os.path.join(INTEGRATION, "rr-hip-sys", "build.rs")).read()
    for flag in build.HIPCC_FLAGS:
        if flag != "-Wall":
            assert f'"{flag}"' in text, flag


def test_render_hip_uses_only_declared_names():
    """Every rr_* / RR_* identifier render_hip.rs uses exists in the generated declarations."""
    text = open(os.path.join(INTEGRATION, "render_hip.rs")).read()
    text = re.sub(r"//.*", "", text)
    decl = open(os.path.join(INTEGRATION, "rr-hip-sys", "src", "lib.rs")).read()
    for name in set(re.findall(r"\b(rr_[a-z_0-9]+|RR_[A-Z_0-9]+)\b", text)) - {"rr_hip_sys", "rr_hip"}:
        assert re.search(rf"\b(struct|fn|const) {name}\b", decl), name
    # every field set in a struct literal exists in that struct
    fields = {m.group(1): set(re.findall(r"pub (\w+):", m.group(2))) for m in re.finditer(r"pub struct (rr_\w+) \{(.*?)\n\}", decl, flags=re.S)}
    for m in re.finditer(r"\b(rr_\w+) \{([^{}]*)\}", text):
        if m.group(1) in fields:
            used, depth, item = set(), 0, ""
            for ch in m.group(2) + ",":  # split at top-level commas; `name: value` or the shorthand `name`
                if ch == "," and depth == 0:
                    if item.strip():
                        used.add(re.match(r"\s*(\w+)", item).group(1))
                    item = ""
                    continue
                depth += ch in "([" 
                depth -= ch in ")]"
                item += ch
            assert used == fields[m.group(1)], (m.group(1), used ^ fields[m.group(1)])  # a Rust literal names every field


# The enum fields of the three rect kinds, in the order the reference stores them: XYRect and XZRect as their
# constructors take them (collider.rs:68-79); yzrect(y0, y1, z0, z1, k) stores YZRect(z0, z1, y0, y1, k)
# (collider.rs:80-84) and the dispatch reads it back in that order (collider.rs:688).
RECT_ENUM_ORDER = {"XYRect": ["x0", "x1", "y0", "y1", "k"], "XZRect": ["x0", "x1", "z0", "z1", "k"], "YZRect": ["z0", "z1", "y0", "y1", "k"]}
# what rr_rect's (a0, a1, b0, b1, k) mean per axis (include/rr_hip.h: XY a = x, b = y; XZ a = x, b = z; YZ a = y, b = z)
RECT_ABI_ORDER = {"XYRect": ("RR_RECT_XY", ["x0", "x1", "y0", "y1", "k"]), "XZRect": ("RR_RECT_XZ", ["x0", "x1", "z0", "z1", "k"]),
                  "YZRect": ("RR_RECT_YZ", ["y0", "y1", "z0", "z1", "k"])}


def rect_arms(text):
    """{kind: (names bound to the enum positions, axis constant, names passed as (a0, a1, b0, b1, k))} of a flattener's match arms."""
    arms = {}
    for m in re.finditer(r"GeometryType::(XYRect|XZRect|YZRect)\(([^)]*)\)\s*=>\s*self\.rect\((RR_RECT_\w+),\s*\(([^)]*)\)", text):
        bound = [x.strip() for x in m.group(2).split(",")]
        passed = [x.strip().lstrip("*") for x in m.group(4).split(",")]
        arms[m.group(1)] = (bound, m.group(3), passed)
    return arms


def check_rect_arms(text):
    arms = rect_arms(text)
    assert set(arms) == set(RECT_ENUM_ORDER)
    for kind, (bound, axis, passed) in arms.items():
        # rename the arm's own variable names to what the enum position holds, then compare with what the ABI wants
        meaning = dict(zip(bound, RECT_ENUM_ORDER[kind]))
        assert len(bound) == 5 and len(set(bound)) == 5, (kind, bound)
        assert axis == RECT_ABI_ORDER[kind][0], (kind, axis)
        assert [meaning[x] for x in passed] == RECT_ABI_ORDER[kind][1], (kind, bound, passed)


def test_rect_arms_bind_the_enum_positions_of_the_reference():
    """rr_rect's a/b ranges per axis against the order in which the reference's enum holds them: the YZ arm must swap."""
    check_rect_arms(open(os.path.join(INTEGRATION, "render_hip.rs")).read())
    # the check itself sees the mistake of passing YZRect's fields straight through
    wrong = ("GeometryType::XYRect(a0, a1, b0, b1, k) => self.rect(RR_RECT_XY, (*a0, *a1, *b0, *b1, *k), material),\n"
             "GeometryType::XZRect(a0, a1, b0, b1, k) => self.rect(RR_RECT_XZ, (*a0, *a1, *b0, *b1, *k), material),\n"
             "GeometryType::YZRect(a0, a1, b0, b1, k) => self.rect(RR_RECT_YZ, (*a0, *a1, *b0, *b1, *k), material),\n")
    try:
        check_rect_arms(wrong)
    except AssertionError:
        pass
    else:
        raise AssertionError("the straight-through YZRect arm was not caught")
    ref = "/root/reference/src/collider.rs"
    if os.path.exists(ref):  # where the reference is at hand, RECT_ENUM_ORDER is read off its constructors
        src = open(ref).read()
        for kind, ctor in (("XYRect", "xyrect"), ("XZRect", "xzrect"), ("YZRect", "yzrect")):
            m = re.search(rf"pub fn {ctor}\([^)]*\) -> Self \{{\s*Self \{{\s*geometry_type: GeometryType::{kind}\(([^)]*)\)", src)
            assert [x.strip() for x in m.group(1).split(",")] == RECT_ENUM_ORDER[kind], kind


def test_host_flattener_puts_y_in_a_and_z_in_b_for_yz_rects():
    """The C++ host path on a non-square yz rect and a non-cubic cuboid: a = y range, b = z range, k = x."""
    from rust_raytracer_amd import host
    b = host.SceneBuilder(1)
    m = b.lambertian(b.solid_color((0.5, 0.5, 0.5)))
    b.add(b.yzrect(1.0, 2.0, 30.0, 50.0, 7.0, m))        # yzrect(y0, y1, z0, z1, k), collider.rs:80
    b.add(b.cuboid((0.0, 10.0, 200.0), (1.0, 40.0, 900.0), m))
    b.set_camera((0, 0, -10), (0, 0, 0), (0, 1, 0), 40.0, 0.0, 10.0, 1.0)
    flat = b.flatten()
    r = flat.rects[0]
    assert (r.axis, r.a0, r.a1, r.b0, r.b1, r.k) == (abi.RECT_YZ, 1.0, 2.0, 30.0, 50.0, 7.0)
    sides = [flat.rects[1 + i] for i in range(6)]        # constructor order, collider.rs:88-95
    assert [s.axis for s in sides] == [abi.RECT_XY, abi.RECT_XY, abi.RECT_XZ, abi.RECT_XZ, abi.RECT_YZ, abi.RECT_YZ]
    for s, k in zip(sides[4:], (1.0, 0.0)):
        assert (s.a0, s.a1, s.b0, s.b1, s.k) == (10.0, 40.0, 200.0, 900.0, k)

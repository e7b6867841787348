"""Parity of the HIP path (through the C ABI, on a real MI355X) with the CPU oracle and the golden fixtures.

Bars: primitive ids exact; t, point and normals within 1e-5 relative (bit-exact where no
transcendental is involved); per-slice pixel sums of whole renders within 1e-3 relative for at
least 98 % of the channels (paths share their random numbers with the oracle, so the remainder are
paths that a last-bit difference in sin/cos/log10 sent somewhere else) and image means within 1 %.
"""
import ctypes as C
import os

import numpy as np
import pytest

import rust_raytracer_amd  # noqa: F401
from rust_raytracer_amd import abi, device, host
import golden_scenes as G
import oracle_lib as O
from test_golden_cpu import compare_hits, intended

pytestmark = pytest.mark.gpu

CONFIG_SCENES = {}


def counts(r):
    """The counters of a render that do not depend on how the kernels organise their work (the steps of the per-lane
    traversal machine do: the wavefront pipeline walks the whole tree with it, the persistent kernels only the BVHs)."""
    return {k: v for k, v in r["counters"].items() if not k.startswith("machine_")}


def config_scene(name):
    if name not in CONFIG_SCENES:
        CONFIG_SCENES[name] = host.build_config_scene(name)
    return CONFIG_SCENES[name]


# ------------------------------------------------------------ primary hits
@pytest.mark.parametrize("name", sorted(G.SCENES))
def test_primary_hits_match_golden_fixtures(gpu_ctx, name):
    fx = np.load(f"{G.GOLDEN_DIR}/hits_{name}.npz")
    A, flat = G.build_host(name)
    gpu_ctx.upload(flat)
    got = gpu_ctx.primary_hits(fx["rays"], seed=int(fx["media_seed"]))
    compare_hits(got, fx["hits"], exact_t=(name != "media"))


@pytest.mark.parametrize("name,side", [("cornell", 96), ("spheres", 96), ("suzanne", 96), ("final", 96)])
def test_primary_hits_match_oracle_on_benchmark_scenes(gpu_ctx, name, side):
    """Camera rays of the benchmark scenes (thin lens, shutter time and all): ids exact, t and normals 1e-5."""
    b, flat, d = config_scene(name)
    gpu_ctx.upload(flat)
    W, H = d["width"], d["height"]
    p = O.make_params(W, H, 4, n_slices=1, seed=11)
    pixels = [(int(i * (W - 1) / (side - 1)), int(j * (H - 1) / (side - 1))) for j in range(side) for i in range(side)]
    rays = O.camera_rays(flat, p, pixels, sample=2)
    want = O.primary_hits(flat, rays, seed=5)
    got = gpu_ctx.primary_hits(rays, seed=5)
    has_media = flat.n_media > 0
    compare_hits(got, want, exact_t=not has_media)
    assert np.array_equal(got["material"], want["material"])
    assert (want["prim_id"] >= 0).mean() > 0.5


def test_primary_hits_honour_the_t_range(gpu_ctx):
    b, flat, _ = config_scene("cornell")
    gpu_ctx.upload(flat)
    rays = np.array([[278, 278, -800, 0, 0, 1, 0]], dtype=np.float32)
    for t_min, t_max in [(0.001, 3.4e38), (0.001, 900.0), (1200.0, 3.4e38), (1400.0, 3.4e38)]:
        got, want = gpu_ctx.primary_hits(rays, t_min, t_max), O.primary_hits(flat, rays, t_min, t_max)
        assert got["prim_id"][0] == want["prim_id"][0] and got["t"][0] == want["t"][0]
    assert gpu_ctx.primary_hits(rays, 1400.0, 3.4e38)["prim_id"][0] == -1
    assert len(gpu_ctx.primary_hits(np.zeros((0, 7), dtype=np.float32))) == 0


def _stress_rays(flat, n, seed):
    """Rays from inside and around the scene in every direction, with the special cases a slab test trips over:
    axis-parallel directions (zero and negative-zero components), origins on box planes, long and short directions."""
    rng = np.random.default_rng(seed)
    look = np.array(flat.camera.origin[:], dtype=np.float32)
    span = max(300.0, float(np.abs(look).max()))
    o = (rng.uniform(-1, 1, (n, 3)) * span * 0.6).astype(np.float32)
    o[: n // 4] = look  # a quarter leave the camera position
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d *= np.float32(10.0) ** rng.uniform(-2, 2.5, (n, 1)).astype(np.float32)  # directions are never normalised (render.rs:32)
    k = n // 8
    d[:k, rng.integers(0, 3)] = 0.0
    d[k:2 * k, 0] = -0.0
    d[2 * k:3 * k, 1:] = 0.0
    d[3 * k:4 * k] = np.round(d[3 * k:4 * k])  # many exact zeros and ties
    o[4 * k:5 * k] = np.round(o[4 * k:5 * k] / 5) * 5  # origins on lattice planes, where many boxes and rects lie
    t = rng.uniform(0, 1, (n, 1)).astype(np.float32)
    return np.concatenate([o, d, t], axis=1)


@pytest.mark.parametrize("name,kw", [("suzanne", {}), ("spheres", {}), ("final", {}),
                                     ("mesh", dict(n_triangles=20000, n_spheres=2000))])
def test_culling_and_visit_skipping_never_change_a_hit(gpu_ctx, name, kw):
    """The production traversal skips subtrees whose (margin-grown) box the ray misses and visits it need not repeat;
    the oracle makes every visit the reference makes.  200 000 adversarial rays per scene: identical records."""
    b, flat, _ = host.build_config_scene(name, seed=3, **kw)
    gpu_ctx.upload(flat)
    rays = _stress_rays(flat, 200_000, seed=9)
    got = gpu_ctx.primary_hits(rays, seed=4)
    want = O.primary_hits(flat, rays, seed=4)
    assert np.array_equal(got["prim_id"], want["prim_id"])
    compare_hits(got, want, exact_t=(name != "final"))


@pytest.mark.parametrize("name", ["spheres", "suzanne"])
def test_bvh_whose_root_box_does_not_hold_the_boxes_below(gpu_ctx, name):
    """The moves for simple subtrees form the margin of their culling test from the root's box, which must then hold every box
    below it.  A tree whose boxes do not nest (a caller may hand over any boxes) goes to the general machine, and either way
    the hits are the oracle's: here every box below the roots is grown past its root's, which lets more rays in and changes
    no hit the reference's own traversal of those boxes would not change."""
    b, flat, _ = config_scene(name)
    simple = gpu_ctx.upload(flat)["subtree_ops"]
    assert simple[0] >= 1 and simple[1] == 0
    nodes = flat.bvh_nodes
    children = set()
    for i in range(flat.n_bvh_nodes):
        for r in (nodes[i].left, nodes[i].right):
            if (r >> 28) == abi.TAG_BVH:
                children.add(r & ((1 << 28) - 1))
    saved = [(i, nodes[i].bmin[1], nodes[i].bmax[1]) for i in sorted(children)[:7]]
    try:
        for i, lo, hi in saved:
            nodes[i].bmin[1], nodes[i].bmax[1] = lo - 5000.0, hi + 5000.0
        info = gpu_ctx.upload(flat)
        assert info["subtree_ops"][0] < simple[0] and info["subtree_ops"][1] >= 1
        rays = _stress_rays(flat, 100_000, seed=21)
        got, want = gpu_ctx.primary_hits(rays, seed=4), O.primary_hits(flat, rays, seed=4)
        assert np.array_equal(got["prim_id"], want["prim_id"])
        compare_hits(got, want, exact_t=True)
        assert (want["prim_id"] >= 0).mean() > 0.05
        kw = dict(n_slices=2, max_depth=12, seed=5)
        ref = gpu_ctx.render(device.make_params(48, 40, 8, flags=abi.FLAG_COUNTERS, **kw))["slice_sums"]  # every visit, in the reference's order
        for flags in (0, abi.FLAG_PATHS, abi.FLAG_WAVEFRONT):
            assert np.array_equal(gpu_ctx.render(device.make_params(48, 40, 8, flags=flags, **kw))["slice_sums"], ref, equal_nan=True), flags
    finally:
        for i, lo, hi in saved:
            nodes[i].bmin[1], nodes[i].bmax[1] = lo, hi


def test_which_scenes_take_the_wavefront_pipeline(gpu_ctx):
    """rr_scene_info.wavefront: scenes that are all BVH with more than 2 MiB of BVH and primitive records go through the wavefront pipeline by default, every
    other scene through the path kernel; whichever runs, the bits are the other's."""
    for name, kw, want in [("spheres", {}, False), ("final", {}, False), ("suzanne", {}, False),
                           ("mesh", dict(n_triangles=60_000, n_spheres=6_000), True)]:
        b, flat, _ = host.build_config_scene(name, **kw)
        info = gpu_ctx.upload(flat)
        assert info["wavefront"] == want, (name, info)
    p = dict(n_slices=2, max_depth=8, seed=3)
    auto = gpu_ctx.render(device.make_params(96, 54, 4, **p))["slice_sums"]
    for flags in (abi.FLAG_PATHS, abi.FLAG_WAVEFRONT):
        assert np.array_equal(auto, gpu_ctx.render(device.make_params(96, 54, 4, flags=flags, **p))["slice_sums"], equal_nan=True), flags


def _random_scene(seed):
    """A scene thrown together from every kind of geometry, in nested BVHs, instances and lists, with the
    awkward cases in: tiny and huge and negative radii, degenerate triangles, flat boxes, a medium in a BVH,
    a BVH inside a rotated, translated BVH, a moving sphere in a BVH."""
    rng = np.random.default_rng(seed)
    b = host.SceneBuilder(seed)
    mats = [b.lambertian(b.solid_color((0.7, 0.6, 0.5))), b.glossy((0.8, 0.8, 0.9), 0.3), b.dielectric(1.5),
            b.emissive((4, 4, 4))]

    def mat():
        i = int(rng.integers(0, len(mats)))
        if seed % 2 == 1 and rng.random() < 0.6:
            i = 0  # odd seeds: mostly Lambertian surfaces, so that few samples are NaN (render.rs:34 with pdf 0)
        return mats[i]

    def prim(scale):
        c = rng.uniform(-scale, scale, 3)
        k = int(rng.integers(0, 7))
        if k == 0:
            return b.sphere(c, float(10.0 ** rng.uniform(-3, np.log10(scale))), mat())
        if k == 1:
            return b.sphere(c, -float(rng.uniform(0.1, 0.3) * scale), mat())  # hollow-glass style negative radius
        if k == 2:
            e = rng.uniform(0.01, 0.5, 2) * scale
            f = [b.xyrect, b.xzrect, b.yzrect][int(rng.integers(0, 3))]
            return f(c[0], c[0] + e[0], c[1], c[1] + e[1], float(c[2]), mat())
        if k == 3:
            e = rng.uniform(0.0, 0.4, 3) * scale  # an edge can be zero: a flat box
            return b.cuboid(c, c + e, mat())
        if k == 4:
            p = c + rng.normal(size=(3, 3)) * scale * float(10.0 ** rng.uniform(-3, -0.3))
            return b.triangle(p[0], p[1], p[2], mat())
        if k == 5:
            return b.triangle(c, c, c + 1.0, mat())  # two equal corners
        return b.moving_sphere(c, c + rng.normal(size=3), 0.0, 1.0, float(rng.uniform(0.05, 0.3) * scale), mat())

    def group(n, scale):
        items = [prim(scale) for _ in range(n)]
        if rng.random() < 0.3:
            items.append(b.constant_medium(b.sphere(rng.uniform(-scale, scale, 3), float(0.3 * scale), mats[0]), 0.5, (1, 1, 1)))
        return b.bvh_node(items)

    inner = group(int(rng.integers(3, 40)), 5.0)
    nested = b.instance_translation(b.instance_rotation(inner, int(rng.integers(0, 3)), float(rng.uniform(-170, 170))), rng.uniform(-20, 20, 3))
    b.add(b.bvh_node([nested] + [prim(30.0) for _ in range(int(rng.integers(2, 60)))]))
    b.add(group(int(rng.integers(1, 200)), 50.0))
    b.add(b.collider_list([prim(10.0), b.instance_translation(group(int(rng.integers(2, 30)), 3.0), rng.uniform(-40, 40, 3))]))
    b.add(b.sphere((0, -1000, 0), 990.0, mats[0]))  # a huge sphere next to the small ones
    # textured surfaces and media with other boundaries than a sphere, at the top level and below instances
    img = rng.integers(0, 256, (7, 5, 3), dtype=np.uint8)
    tex = [b.lambertian(b.checkered((0.1, 0.2, 0.3), (0.9, 0.8, 0.7))), b.lambertian(b.noise(2.5, 5)),
           b.lambertian(b.load_image_rgb8(img)), b.isotropic((0.6, 0.7, 0.8))]
    for m in tex:
        c = rng.uniform(-30, 30, 3)
        b.add(b.sphere(c, float(rng.uniform(2, 8)), m))
        b.add(b.instance_rotation(b.xzrect(c[0], c[0] + 9, c[2], c[2] + 7, float(c[1]), m), int(rng.integers(0, 3)), float(rng.uniform(-90, 90))))
    c = rng.uniform(-25, 25, 3)
    b.add(b.constant_medium(b.cuboid(c, c + rng.uniform(3, 15, 3), mats[0]), 0.15, (0.9, 0.9, 0.9)))
    two = b.bvh_node([b.sphere(c + 20, 6.0, mats[0]), b.sphere(c + 24, 5.0, mats[0])])
    b.add(b.instance_translation(b.constant_medium(two, 0.3, (0.2, 0.3, 0.4)), rng.uniform(-10, 10, 3)))
    b.add(b.constant_medium(b.instance_translation(b.instance_rotation(b.sphere((0, 0, 0), 9.0, mats[0]), 1, 30.0), rng.uniform(-30, 30, 3)),
                            0.1, (1, 1, 1)))
    # a chain of seven instances around a small BVH, and a list inside a list (a list has no box, so none inside a BVH)
    deep = group(int(rng.integers(2, 12)), 4.0)
    for i in range(7):
        deep = (b.instance_rotation(deep, int(rng.integers(0, 3)), float(rng.uniform(-60, 60))) if i % 2 == 0
                else b.instance_translation(deep, rng.uniform(-6, 6, 3)))
    b.add(deep)
    c = rng.uniform(-40, 40, 3)  # (no degenerate triangle here: outside a BVH it would turn every ray of the scene into NaN)
    b.add(b.collider_list([b.sphere(c, 3.0, mat()), b.collider_list([b.yzrect(c[1], c[1] + 5, c[2], c[2] + 6, float(c[0]) - 8, mat()),
                                                                       group(3, 2.0), b.sphere(c + 7, 2.0, mat())]),
                           b.cuboid(c - 9, c - 5, mat())]))
    # media bounded by a bare sphere (one fused op of the scene program): around much of the scene, small, inside instances
    b.add(b.constant_medium(b.sphere(rng.uniform(-5, 5, 3), 70.0, mats[0]), 0.004, (0.8, 0.9, 1.0)))
    b.add(b.constant_medium(b.sphere(rng.uniform(-30, 30, 3), float(rng.uniform(3, 9)), mats[0]), 0.2, (0.3, 0.2, 0.1)))
    b.add(b.instance_translation(b.instance_rotation(b.constant_medium(b.sphere((1, 2, 3), 6.0, mats[0]), 0.3, (0.5, 0.5, 0.5)), 2, 25.0),
                                 rng.uniform(-25, 25, 3)))
    b.set_lights(b.xzrect(-5, 5, -5, 5, 60.0, mats[3]))
    b.set_camera((0, 20, -120), (0, 0, 0), (0, 1, 0), 40.0, 0.0, 10.0, 1.0)
    b.set_background((0.5, 0.6, 0.9))
    return b, b.flatten()


@pytest.mark.parametrize("seed", range(16))
def test_production_traversal_on_random_scenes(gpu_ctx, seed):
    """Culling classes, visit skipping, one-word frames and the scene program against the recursive oracle on
    scenes nobody tuned them for: identical primitive, t, point and normal for 60 000 rays each."""
    b, flat = _random_scene(seed)
    info = gpu_ctx.upload(flat)
    rays = _stress_rays(flat, 60_000, seed=100 + seed)
    rays[:, :3] *= 0.3  # most origins inside the cloud of objects
    got = gpu_ctx.primary_hits(rays, seed=4)
    want = O.primary_hits(flat, rays, seed=4)
    compare_hits(got, want, exact_t=(flat.n_media == 0))
    assert (want["prim_id"] >= 0).mean() > 0.2
    assert sum(info["bvh_nodes_by_class"]) > 0


@pytest.mark.parametrize("seed", range(8))
def test_renders_of_random_scenes_agree_between_all_kernel_forms(gpu_ctx, seed):
    """Whole paths (bounce rays start on surfaces, inside media, at grazing angles) through randomly assembled
    scenes: production kernels, reference-order kernels and the persistent kernel produce the same bits."""
    b, flat = _random_scene(40 + seed)
    gpu_ctx.upload(flat)
    kw = dict(n_slices=8, max_depth=25, seed=3)  # one sample per slice: a NaN sample (render.rs:34 with pdf 0) spoils only itself
    base = gpu_ctx.render(device.make_params(48, 40, 8, **kw))
    for flags in (abi.FLAG_COUNTERS, abi.FLAG_PERSISTENT, abi.FLAG_PERSISTENT | abi.FLAG_COUNTERS, abi.FLAG_WAVEFRONT | abi.FLAG_NO_FINISH_KERNEL):
        other = gpu_ctx.render(device.make_params(48, 40, 8, flags=flags, **kw))
        assert np.array_equal(base["slice_sums"], other["slice_sums"], equal_nan=True), flags
    want = O.render(flat, O.make_params(48, 40, 8, **kw), n_threads=8)
    nan_got, nan_want = np.isnan(base["slice_sums"]), np.isnan(want["slice_sums"])
    assert (nan_got == nan_want).mean() > 0.995
    both = ~nan_got & ~nan_want
    assert both.mean() > 0.05, "too few samples without NaN to compare"
    assert np.isclose(base["slice_sums"][both], want["slice_sums"][both], rtol=1e-3, atol=1e-5).mean() > 0.995


def _random_bvh_scene(seed):
    """Nothing but what the wide node move is for: BVHs over spheres (tiny and huge), rects, flat and ordinary
    cuboids and triangles, BVHs inside BVHs, BVHs under instances, next to top-level primitives and a fog."""
    rng = np.random.default_rng(1000 + seed)
    b = host.SceneBuilder(seed)
    mats = [b.lambertian(b.solid_color((0.7, 0.6, 0.5))), b.lambertian(b.checkered((0.1, 0.2, 0.3), (0.9, 0.8, 0.7))),
            b.glossy((0.8, 0.8, 0.9), 0.3), b.dielectric(1.5), b.emissive((4, 4, 4))]

    def mat():
        return mats[0] if rng.random() < 0.6 else mats[int(rng.integers(0, len(mats)))]

    def prim(scale, c0):
        c = c0 + rng.uniform(-scale, scale, 3)
        k = int(rng.integers(0, 6))
        if k == 0:
            return b.sphere(c, float(10.0 ** rng.uniform(-2, np.log10(0.4 * scale))), mat())
        if k == 1:  # (no negative radius: the reference's box of such a sphere is inside out, and a node over it keeps the reference's class)
            return b.sphere(c, float(rng.uniform(0.05, 0.2) * scale), mat())
        if k == 2:
            e = rng.uniform(0.01, 0.5, 2) * scale
            f = [b.xyrect, b.xzrect, b.yzrect][int(rng.integers(0, 3))]
            return f(c[0], c[0] + e[0], c[1], c[1] + e[1], float(c[2]), mat())
        if k == 3:
            e = rng.uniform(0.0, 0.4, 3) * scale
            return b.cuboid(c, c + e, mat())
        p = c + rng.normal(size=(3, 3)) * scale * float(10.0 ** rng.uniform(-2, -0.3))
        return b.triangle(p[0], p[1], p[2], mat())  # (no slivers: a node over one loses its culling class, and the scene the wide move)

    def group(n, scale, c0):
        return b.bvh_node([prim(scale, c0) for _ in range(n)])

    zero = np.zeros(3)
    inner = [group(int(rng.integers(2, 30)), 6.0, rng.uniform(-25, 25, 3)) for _ in range(int(rng.integers(1, 5)))]
    b.add(b.bvh_node(inner + [prim(30.0, zero) for _ in range(int(rng.integers(20, 300)))]))  # BVHs inside a BVH
    b.add(group(1, 10.0, rng.uniform(-20, 20, 3)))  # a span-1 tree
    b.add(group(int(rng.integers(2, 6)), 10.0, rng.uniform(-20, 20, 3)))
    b.add(b.instance_translation(b.instance_rotation(group(int(rng.integers(5, 80)), 8.0, zero), int(rng.integers(0, 3)),
                                                     float(rng.uniform(-170, 170))), rng.uniform(-30, 30, 3)))
    b.add(b.sphere((0, -1000, 0), 985.0, mats[1]))
    b.add(b.xzrect(-8, 8, -8, 8, 55.0, mats[4]))
    b.add(b.constant_medium(b.sphere(rng.uniform(-5, 5, 3), 80.0, mats[0]), 0.003, (0.8, 0.9, 1.0)))
    b.set_lights(b.xzrect(-8, 8, -8, 8, 55.0, mats[4]))
    b.set_camera((0, 25, -110), (0, 0, 0), (0, 1, 0), 40.0, 0.0, 10.0, 1.0)
    b.set_background((0.4, 0.5, 0.8))
    return b, b.flatten()


@pytest.mark.parametrize("seed", range(24))
def test_wide_node_move_on_random_bvh_scenes(gpu_ctx, seed):
    """The node move that tests both children's boxes at their parent (the wavefront pipeline's, for scenes whose BVH nodes
    are all of a culling class) against the one-box move, the path kernel, the reference-order kernels, the item-granular
    kernel and the oracle, on randomly built BVH scenes, whole paths: the same bits from every device kernel."""
    for attempt in range(8):  # (a random triangle can come out a sliver: the node over it keeps the reference's class, and the scene the one-box move)
        b, flat = _random_bvh_scene(seed + 1000 * attempt)
        info = gpu_ctx.upload(flat)
        if info["wide_nodes"]:
            break
    assert info["wide_nodes"] and info["subtree_ops"][0] >= 3 and info["subtree_ops"][1] == 0, info
    kw = dict(n_slices=8, max_depth=25, seed=5)
    base = gpu_ctx.render(device.make_params(64, 48, 16, **kw))
    for flags in (abi.FLAG_NARROW_NODES, abi.FLAG_COUNTERS, abi.FLAG_PERSISTENT, abi.FLAG_WAVEFRONT, abi.FLAG_WAVEFRONT | abi.FLAG_NARROW_NODES,
                  abi.FLAG_WAVEFRONT | abi.FLAG_NO_FINISH_KERNEL, abi.FLAG_COUNT_PRODUCTION, abi.FLAG_WAVEFRONT | abi.FLAG_COUNT_PRODUCTION):
        other = gpu_ctx.render(device.make_params(64, 48, 16, flags=flags, **kw))
        assert np.array_equal(base["slice_sums"], other["slice_sums"], equal_nan=True), flags
        assert other["counters"]["rays"] == base["counters"]["rays"]
    want = O.render(flat, O.make_params(64, 48, 16, **kw), n_threads=8)
    nan_got, nan_want = np.isnan(base["slice_sums"]), np.isnan(want["slice_sums"])
    assert (nan_got == nan_want).mean() > 0.995
    both = ~nan_got & ~nan_want
    assert both.mean() > 0.3
    assert np.isclose(base["slice_sums"][both], want["slice_sums"][both], rtol=1e-3, atol=1e-5).mean() > 0.99
    rays = _stress_rays(flat, 20_000, seed=300 + seed)
    rays[:, :3] *= 0.3
    compare_hits(gpu_ctx.primary_hits(rays, seed=4), O.primary_hits(flat, rays, seed=4), exact_t=False)


@pytest.mark.parametrize("which", ["final", "cornell", "suzanne", "random 41", "random 44"])
def test_fused_ops_of_the_scene_program_equal_the_ops_they_stand_for(gpu_ctx, monkeypatch, which):
    """OP_MED_SPHERE (a medium bounded by a bare sphere: both probes from one discriminant) and OP_XFORM (a run of reset,
    translate, rotate) against the same scene uploaded with one op per step (RR_NO_FUSE): same hits, same samples, same
    counters, bit for bit."""
    if which.startswith("random"):
        b, flat = _random_scene(int(which.split()[1]))
    else:
        b, flat, _ = config_scene(which)
    rays = _stress_rays(flat, 40_000, seed=21)
    kw = dict(n_slices=4, max_depth=30, seed=8)
    out = []
    for no_fuse in (False, True):
        if no_fuse:
            monkeypatch.setenv("RR_NO_FUSE", "1")
        ctx = device.Context(0)  # (a context reads the switch when it is created)
        try:
            ctx.upload(flat)
            hits = ctx.primary_hits(rays, seed=4)
            r = ctx.render(device.make_params(72, 56, 8, flags=abi.FLAG_COUNTERS, **kw))
            r2 = ctx.render(device.make_params(72, 56, 8, **kw))
        finally:
            ctx.close()
        out.append((hits, r, r2))
    monkeypatch.delenv("RR_NO_FUSE")
    (h0, a0, b0), (h1, a1, b1) = out
    assert np.array_equal(h0.tobytes(), h1.tobytes()) or all(np.array_equal(h0[f], h1[f], equal_nan=True) for f in h0.dtype.names)
    assert np.array_equal(a0["slice_sums"], a1["slice_sums"], equal_nan=True) and counts(a0) == counts(a1)
    assert np.array_equal(b0["slice_sums"], b1["slice_sums"], equal_nan=True)
    assert np.array_equal(a0["slice_sums"], b0["slice_sums"], equal_nan=True)


# ------------------------------------------------- the labelled intended semantics
@pytest.mark.parametrize("name", G.INTENDED_SCENES)
def test_intended_semantics_match_golden_fixtures(gpu_ctx, name):
    fx = np.load(f"{G.GOLDEN_DIR}/hits_{name}_intended.npz")
    A, flat = G.build_host(name)
    gpu_ctx.upload(intended(flat))
    compare_hits(gpu_ctx.primary_hits(fx["rays"], seed=int(fx["media_seed"])), fx["hits"], exact_t=True)


@pytest.mark.parametrize("name,kw", [("spheres", {}), ("suzanne", {}), ("final", {}), ("mesh", dict(n_triangles=20000, n_spheres=2000))])
def test_intended_semantics_match_the_oracle_on_benchmark_scenes(gpu_ctx, name, kw):
    """RR_SEMANTICS_INTENDED (never the default): hits identical to the oracle's under the same switch for camera rays and the
    adversarial set, renders identical between all kernel organisations and in parity with the oracle's, and more geometry
    found than under the reference's semantics (whose miss branch loses the right subtree, collider.rs:555-561)."""
    b, flat, d = host.build_config_scene(name, seed=3, **kw)
    fi = intended(flat)
    gpu_ctx.upload(fi)
    rays = _stress_rays(flat, 60_000, seed=9)
    got, want = gpu_ctx.primary_hits(rays, seed=4), O.primary_hits(fi, rays, seed=4)
    compare_hits(got, want, exact_t=(flat.n_media == 0))
    ref = O.primary_hits(flat, rays, seed=4)
    assert (want["prim_id"] >= 0).sum() > (ref["prim_id"] >= 0).sum()
    kw2 = dict(n_slices=4, max_depth=20, seed=5)
    base = gpu_ctx.render(device.make_params(64, 48, 8, **kw2))
    for flags in (abi.FLAG_COUNTERS, abi.FLAG_COUNT_PRODUCTION, abi.FLAG_PERSISTENT, abi.FLAG_WAVEFRONT, abi.FLAG_PATHS):
        other = gpu_ctx.render(device.make_params(64, 48, 8, flags=flags, **kw2))
        assert np.array_equal(base["slice_sums"], other["slice_sums"], equal_nan=True), flags
    assert_render_parity(base, O.render(fi, O.make_params(64, 48, 8, **kw2), n_threads=8), min_close=0.97)
    gpu_ctx.upload(flat)
    assert not np.array_equal(gpu_ctx.render(device.make_params(64, 48, 8, **kw2))["slice_sums"], base["slice_sums"], equal_nan=True)


@pytest.mark.parametrize("seed", range(4))
def test_intended_semantics_on_random_scenes(gpu_ctx, seed):
    b, flat = _random_scene(20 + seed)
    fi = intended(flat)
    gpu_ctx.upload(fi)
    rays = _stress_rays(flat, 40_000, seed=300 + seed)
    rays[:, :3] *= 0.3
    compare_hits(gpu_ctx.primary_hits(rays, seed=4), O.primary_hits(fi, rays, seed=4), exact_t=(flat.n_media == 0))


# ------------------------------------------------------------ whole renders
def render_both(ctx, flat, p, n_threads=16):
    ctx.upload(flat)
    return ctx.render(p), O.render(flat, p, n_threads=n_threads, counters=True)


def assert_render_parity(got, want, min_close=0.98, mean_tol=0.01):
    a, c = got["slice_sums"], want["slice_sums"]
    nan_a, nan_c = np.isnan(a), np.isnan(c)
    assert (nan_a == nan_c).mean() > 0.995, "NaN pixels (zeroed per slice by render.rs:106-108) disagree"
    both = ~nan_a & ~nan_c
    close = np.isclose(a[both], c[both], rtol=1e-3, atol=1e-5)
    assert close.mean() >= min_close, f"only {close.mean():.4f} of the per-slice channel sums agree"
    ma, mc = float(a[both].mean()), float(c[both].mean())
    assert abs(ma - mc) <= mean_tol * max(abs(mc), 1e-6), (ma, mc)
    # 8-bit images: identical almost everywhere, never far apart
    d8 = np.abs(got["rgb8"].astype(np.int32) - want["rgb8"].astype(np.int32))
    assert (d8 <= 1).mean() > 0.97 and np.allclose(got["linear"][both.all(axis=0)], want["linear"][both.all(axis=0)], rtol=0.5, atol=1.0)


def test_cornell_render_matches_oracle(gpu_ctx):
    b, flat, _ = config_scene("cornell")
    p = device.make_params(96, 96, 16, n_slices=4, max_depth=200, seed=21)
    got, want = render_both(gpu_ctx, flat, p)
    assert_render_parity(got, want)
    assert got["counters"]["samples"] == want["counters"]["samples"] == 96 * 96 * 16
    assert abs(got["counters"]["rays"] - want["counters"]["rays"]) <= 2e-3 * want["counters"]["rays"]


def test_random_spheres_render_matches_oracle(gpu_ctx):
    """Sky-lit scene with glossy and dielectric spheres and an empty lights list: exercises the zero
    scattering pdf of non-Lambertian materials and the NaN samples of pdf 0 (render.rs:34)."""
    b, flat, _ = config_scene("spheres")
    p = device.make_params(80, 45, 16, n_slices=4, max_depth=50, seed=2)
    got, want = render_both(gpu_ctx, flat, p)
    assert np.isnan(want["slice_sums"]).any(), "this scene is expected to produce NaN samples"
    assert_render_parity(got, want)


def test_suzanne_render_matches_oracle(gpu_ctx):
    b, flat, _ = config_scene("suzanne")
    p = device.make_params(64, 64, 8, n_slices=2, max_depth=100, seed=4)
    got, want = render_both(gpu_ctx, flat, p)
    assert_render_parity(got, want)


def test_final_scene_render_matches_oracle(gpu_ctx):
    """Media, Perlin and image textures, moving sphere, instanced sphere BVH, cuboid BVH."""
    b, flat, _ = config_scene("final")
    p = device.make_params(64, 64, 8, n_slices=2, max_depth=50, seed=8)
    got, want = render_both(gpu_ctx, flat, p)
    assert_render_parity(got, want, min_close=0.97)


def test_paths_match_the_numpy_restatement(gpu_ctx):
    """The whole-path fixture written by the numpy restatement of ray_color (tests/golden/paths_box.npz: every material
    and texture kind, a medium, a moving sphere, a thin lens), sample by sample, through both kernel organisations."""
    from test_golden_cpu import _paths_fixture, compare_samples
    GP, gold = _paths_fixture()
    b, flat = GP.build_host()
    gpu_ctx.upload(flat)
    R = GP.RENDER
    for flags in (abi.FLAG_WAVEFRONT, abi.FLAG_PERSISTENT, abi.FLAG_PATHS):  # (the fixtures are the reference's walk, dead paths included)
        got = gpu_ctx.render(device.make_params(R["W"], R["H"], R["spp"], n_slices=R["spp"], max_depth=R["max_depth"], seed=R["seed"],
                                                flags=flags | abi.FLAG_TRACE_DEAD_PATHS))["slice_sums"]
        compare_samples(got, gold, 0.985)
    from test_golden_cpu import _nolight_fixture
    GP, gold = _nolight_fixture()  # an empty list as `lights`: the NaN samples of render.rs:34
    b, flat = GP.build_host(light=False)
    gpu_ctx.upload(flat)
    N = GP.NOLIGHT
    got = gpu_ctx.render(device.make_params(N["W"], N["H"], N["spp"], n_slices=N["spp"], max_depth=R["max_depth"], seed=R["seed"],
                                            flags=abi.FLAG_TRACE_DEAD_PATHS))["slice_sums"]
    compare_samples(got, gold, 0.985)


def test_device_perlin_and_textures_match_the_fixture_and_the_oracle(gpu_ctx):
    """Perlin::noise and Perlin::turb on the device against the numpy restatement's fixture (tests/golden/perlin.npz: table and
    128 points, bit for bit: the functions are plain f32 arithmetic and integer table look-ups), and Texture::get_color of all
    four texture kinds against the C++ oracle on 4 000 random arguments (checker and noise end in a sin: 2 ulp)."""
    fx = np.load(f"{G.GOLDEN_DIR}/perlin.npz")
    b = host.SceneBuilder(5)
    img = np.random.default_rng(7).integers(0, 256, (9, 13, 3), dtype=np.uint8)
    texs = [b.noise(4.0, 7), b.solid_color((0.1, 0.2, 0.3)), b.checkered((0.2, 0.3, 0.1), (0.9, 0.9, 0.9)), b.load_image_rgb8(img)]
    for t in texs:
        b.add(b.sphere((0, 0, 0), 1.0, b.lambertian(t)))
    b.set_camera((0, 0, -10), (0, 0, 0), (0, 1, 0), 40.0, 0.0, 10.0, 1.0)
    flat = b.flatten()
    kinds = [flat.textures[i].type for i in range(flat.n_textures)]
    assert sorted(kinds) == [abi.TEX_SOLID, abi.TEX_CHECKERED, abi.TEX_NOISE, abi.TEX_IMAGE]
    noise_tex = kinds.index(abi.TEX_NOISE)
    pn = flat.perlin[flat.textures[noise_tex].index]  # the fixture's table in place of the builder's
    for i in range(256):
        for a in range(3):
            pn.ranvec[i][a] = fx["ranvec"][i, a]
        pn.perm_x[i], pn.perm_y[i], pn.perm_z[i] = int(fx["perms"][0, i]), int(fx["perms"][1, i]), int(fx["perms"][2, i])
    gpu_ctx.upload(flat)
    n = len(fx["points"])
    uvp = np.concatenate([np.zeros((n, 2), dtype=np.float32), fx["points"]], axis=1)
    tex = np.full(n, noise_tex, dtype=np.uint32)
    assert np.array_equal(gpu_ctx.eval_textures(tex, uvp, abi.EVAL_NOISE)[:, 0], fx["noise"])
    assert np.array_equal(gpu_ctx.eval_textures(tex, uvp, abi.EVAL_TURB)[:, 0], fx["turb"])
    # far, huge and non-finite points go through the saturating float-to-integer casts of perlin.rs:69-71
    odd = np.array([[1e9, -1e9, 3e9], [1e20, -1e20, 5.5], [-0.0, 255.999, 256.0], [np.inf, 1.0, 2.0], [np.nan, 0.5, -7.25], [-3e38, 3e38, 0.1]], dtype=np.float32)
    uvo = np.concatenate([np.zeros((len(odd), 2), dtype=np.float32), odd], axis=1)
    got = gpu_ctx.eval_textures(np.full(len(odd), noise_tex, dtype=np.uint32), uvo, abi.EVAL_NOISE)[:, 0]
    want = np.array([O.lib().rro_perlin_noise(C.byref(pn), abi.vec3(q)) for q in odd], dtype=np.float32)
    assert np.array_equal(got, want, equal_nan=True)
    rng = np.random.default_rng(11)
    m = 4000
    uvp = np.concatenate([rng.uniform(-0.2, 1.2, (m, 2)), rng.normal(size=(m, 3)) * 10.0 ** rng.uniform(-1, 2.5, (m, 1))], axis=1).astype(np.float32)
    tex = rng.integers(0, flat.n_textures, m).astype(np.uint32)
    got = gpu_ctx.eval_textures(tex, uvp)
    want = np.zeros((m, 3), dtype=np.float32)
    out = (C.c_float * 3)()
    for i in range(m):
        O.lib().rro_texture_color(C.byref(flat), int(tex[i]), float(uvp[i, 0]), float(uvp[i, 1]), abi.vec3(uvp[i, 2:]), out)
        want[i] = out[:]
    exact = np.isin(tex, [kinds.index(abi.TEX_SOLID), kinds.index(abi.TEX_IMAGE)])
    assert np.array_equal(got[exact], want[exact])
    # a checker cell flips where the product of sines is within rounding of zero, and sin of a large argument differs in its
    # last bits between the two math libraries: all but a handful of samples agree
    close = np.isclose(got, want, rtol=0, atol=3e-6).all(axis=1)
    assert close.mean() > 0.995, close.mean()
    with pytest.raises(device.DeviceError, match="noise texture"):
        gpu_ctx.eval_textures([kinds.index(abi.TEX_SOLID)], [[0, 0, 1, 2, 3]], abi.EVAL_TURB)
    with pytest.raises(device.DeviceError, match="outside"):
        gpu_ctx.eval_textures([99], [[0, 0, 1, 2, 3]])


PRESETS = ["default_scene", "random_spheres", "random_moving_spheres", "two_spheres", "two_perlin_spheres", "earth", "rect_light",
           "cornell_box", "cornell_smoke", "cornell_pedestal", "final_scene"]


@pytest.mark.parametrize("name", PRESETS)
def test_every_scene_function_of_main_rs_as_written(gpu_ctx, name):
    """The scene functions of src/main.rs exactly as HEAD has them (no light repair: cornell_smoke, cornell_pedestal and
    final_scene pass a list as `lights`, whose pdf is 0 and whose sample is (1,0,0), collider.rs:714-725), one sample
    per slice so that every sample is compared on its own: NaN samples agree, the others within 1e-3."""
    import os
    b = host.SceneBuilder(13)
    kw = {}
    if name in ("earth", "final_scene"):
        kw["image"] = b.load_image(os.path.join(host.ASSETS, "earthmap.ppm"))
    if name == "cornell_pedestal":
        kw["obj_path"] = os.path.join(host.ASSETS, "suzanne.obj")
    b.preset(name, repair=False, **kw)
    flat = b.flatten()
    gpu_ctx.upload(flat)
    p = device.make_params(48, 32, 8, n_slices=8, max_depth=20, seed=6)
    got = gpu_ctx.render(p)
    want = O.render(flat, p, n_threads=8)
    nan_got, nan_want = np.isnan(got["slice_sums"]), np.isnan(want["slice_sums"])
    assert (nan_got == nan_want).mean() > 0.995
    both = ~nan_got & ~nan_want
    if both.any():
        assert np.isclose(got["slice_sums"][both], want["slice_sums"][both], rtol=1e-3, atol=1e-5).mean() > 0.99
    assert (np.abs(got["rgb8"].astype(np.int32) - want["rgb8"].astype(np.int32)) <= 1).mean() > 0.97
    other = gpu_ctx.render(device.make_params(48, 32, 8, n_slices=8, max_depth=20, seed=6,
                                              flags=abi.FLAG_PERSISTENT if flat.n_bvh_nodes else abi.FLAG_WAVEFRONT))
    assert np.array_equal(got["slice_sums"], other["slice_sums"], equal_nan=True)


def test_intersection_counters_match_oracle(gpu_ctx):
    """The instrumented kernel counts the same box and primitive tests as the recursive oracle."""
    b, flat, _ = config_scene("cornell")
    gpu_ctx.upload(flat)
    p = device.make_params(48, 48, 8, n_slices=2, max_depth=50, seed=3, flags=abi.FLAG_COUNTERS)
    got = gpu_ctx.render(p)["counters"]
    assert got["rect_tests"] == 18 * got["rays"] and got["inst_visits"] == 4 * got["rays"] and got["box_tests"] == 0
    assert got["prim_tests"] == got["rect_tests"] + got["sphere_tests"] + got["tri_tests"]
    b, flat, _ = config_scene("suzanne")
    gpu_ctx.upload(flat)
    p = device.make_params(32, 32, 4, n_slices=2, max_depth=20, seed=3, flags=abi.FLAG_COUNTERS)
    got = gpu_ctx.render(p)["counters"]
    want = O.render(flat, p, n_threads=8, counters=True)["counters"]
    for k in ("rays", "box_tests", "tri_tests", "rect_tests", "inst_visits"):
        assert abs(got[k] - want[k]) <= 5e-3 * want[k], (k, got[k], want[k])


@pytest.mark.parametrize("name", ["cornell", "spheres", "suzanne", "final"])
def test_production_counters_count_the_production_traversal(gpu_ctx, name):
    """RR_FLAG_COUNT_PRODUCTION switches the counters on without changing the traversal: same image as without flags,
    never more tests than the reference-order count, and far fewer box tests wherever BVH subtrees are culled.  The kernel
    organisations agree on rays, samples and instance visits exactly and on the tests within 3 %: the culling test of the
    kernels that enter a subtree from the scene program grows every box of the subtree by the margin of the subtree's root,
    the machine that walks the whole tree by each node's own (smaller) margin, so the former enter a few more nodes whose
    boxes the ray only just misses (and find nothing there: the images are the same bits)."""
    b, flat, _ = config_scene(name)
    gpu_ctx.upload(flat)
    kw = dict(n_slices=2, max_depth=30, seed=29)
    plain = gpu_ctx.render(device.make_params(60, 44, 6, **kw))
    prod = gpu_ctx.render(device.make_params(60, 44, 6, flags=abi.FLAG_COUNT_PRODUCTION, **kw))
    ref = gpu_ctx.render(device.make_params(60, 44, 6, flags=abi.FLAG_COUNTERS, **kw))
    assert np.array_equal(plain["slice_sums"], prod["slice_sums"], equal_nan=True)
    cp, cr = prod["counters"], ref["counters"]
    assert cp["rays"] == cr["rays"] == plain["counters"]["rays"] and cp["samples"] == 60 * 44 * 6
    assert plain["counters"]["box_tests"] == plain["counters"]["prim_tests"] == 0  # counters are off without a flag
    for k in ("box_tests", "sphere_tests", "rect_tests", "tri_tests", "prim_tests"):
        assert cp[k] <= cr[k], k
    assert cp["inst_visits"] == cr["inst_visits"] or flat.n_media > 0
    assert cp["prim_tests"] == cp["sphere_tests"] + cp["rect_tests"] + cp["tri_tests"]
    if flat.n_bvh_nodes:
        assert 0 < cp["box_tests"] < 0.7 * cr["box_tests"]
        assert 0 < cp["machine_lane_steps"] <= 64 * cp["machine_wave_steps"]
    else:
        assert counts(prod) == counts(ref) and cp["machine_lane_steps"] == 0
    # The kernel organisations against each other, all with the node move that tests one box per visit (RR_FLAG_NARROW_NODES:
    # the item-granular kernel has no other).
    narrow = gpu_ctx.render(device.make_params(60, 44, 6, flags=abi.FLAG_COUNT_PRODUCTION | abi.FLAG_NARROW_NODES, **kw))
    assert np.array_equal(plain["slice_sums"], narrow["slice_sums"], equal_nan=True)
    for flags in (abi.FLAG_PATHS, abi.FLAG_WAVEFRONT, abi.FLAG_PERSISTENT):
        other = gpu_ctx.render(device.make_params(60, 44, 6, flags=flags | abi.FLAG_COUNT_PRODUCTION | abi.FLAG_NARROW_NODES, **kw))
        assert np.array_equal(plain["slice_sums"], other["slice_sums"], equal_nan=True)
        co = counts(other)
        for k, v in counts(narrow).items():
            assert co[k] == v if k in ("rays", "samples", "inst_visits") else abs(co[k] - v) <= 0.03 * v, (flags, k, co[k], v)
    # The wide node move of the wavefront pipeline's traversal kernel (wherever every BVH node is of a culling class: all
    # three scenes with BVHs here) tests the boxes of both children at their parent: the same primitives are reached (the same
    # image, the same primitive tests) in fewer steps of the machine; it tests a few more boxes (both children, where the
    # one-box move would not have come to the second).
    info = gpu_ctx.info()
    assert info["wide_nodes"] == (flat.n_bvh_nodes != 0)
    wide = gpu_ctx.render(device.make_params(60, 44, 6, flags=abi.FLAG_WAVEFRONT | abi.FLAG_COUNT_PRODUCTION, **kw))
    one = gpu_ctx.render(device.make_params(60, 44, 6, flags=abi.FLAG_WAVEFRONT | abi.FLAG_COUNT_PRODUCTION | abi.FLAG_NARROW_NODES, **kw))
    assert np.array_equal(plain["slice_sums"], wide["slice_sums"], equal_nan=True)
    cw, cn = wide["counters"], one["counters"]
    if info["wide_nodes"]:
        # (the camera rays walk the whole tree with the wide move; the paths' later rays go through the scene program and
        # the one-box move with the subtree's margin in both runs, so the counts of the two runs differ by the camera rays')
        for k in ("rays", "samples", "inst_visits", "sphere_tests", "rect_tests", "tri_tests"):
            assert cw[k] == cn[k], (k, cw[k], cn[k])
        assert cn["box_tests"] <= cw["box_tests"] <= 1.35 * cn["box_tests"]
        assert cw["machine_lane_steps"] < cn["machine_lane_steps"]
    else:
        assert counts(wide) == counts(one)
    with pytest.raises(device.DeviceError, match="exclude"):
        gpu_ctx.render(device.make_params(8, 8, 2, flags=abi.FLAG_COUNTERS | abi.FLAG_COUNT_PRODUCTION))


@pytest.mark.parametrize("name", ["cornell", "spheres", "suzanne", "final"])
def test_fast_traversal_equals_reference_order_traversal(gpu_ctx, name):
    """The production kernel leaves out BVH visits that cannot change the result (the re-test of `left`
    after it missed, collider.rs:555-561, and the second visit of a span-1 node's only child) wherever
    no constant medium lies below; the counting kernel makes every visit the reference makes.  Both must
    produce the same bits."""
    b, flat, _ = config_scene(name)
    gpu_ctx.upload(flat)
    kw = dict(n_slices=2, max_depth=50, seed=17)
    fast = gpu_ctx.render(device.make_params(56, 40, 6, **kw))
    full = gpu_ctx.render(device.make_params(56, 40, 6, flags=abi.FLAG_COUNTERS, **kw))
    assert np.array_equal(fast["slice_sums"], full["slice_sums"], equal_nan=True)
    assert np.array_equal(fast["rgb8"], full["rgb8"])


@pytest.mark.parametrize("name", ["cornell", "spheres", "suzanne", "final"])
def test_wavefront_pipeline_equals_persistent_kernel(gpu_ctx, name):
    """The two kernel organisations (one persistent kernel; traversal + shading kernels over a queue of
    live paths) key every random stream and order every f32 sum alike: same bits, same counters."""
    b, flat, _ = config_scene(name)
    gpu_ctx.upload(flat)
    kw = dict(n_slices=3, max_depth=12, seed=23)
    for extra, wf_extra in ((0, 0), (abi.FLAG_COUNTERS, 0), (0, abi.FLAG_NO_FINISH_KERNEL)):
        one = gpu_ctx.render(device.make_params(45, 37, 9, flags=abi.FLAG_PERSISTENT | extra, **kw))
        two = gpu_ctx.render(device.make_params(45, 37, 9, flags=abi.FLAG_WAVEFRONT | extra | wf_extra, **kw))
        assert np.array_equal(one["slice_sums"], two["slice_sums"], equal_nan=True)
        assert np.array_equal(one["rgb8"], two["rgb8"])
        assert counts(one) == counts(two)
        for flags in (extra, abi.FLAG_PATHS | extra):  # whichever organisation the library picks; the path-granular kernel
            third = gpu_ctx.render(device.make_params(45, 37, 9, flags=flags, **kw))
            assert np.array_equal(one["slice_sums"], third["slice_sums"], equal_nan=True) and counts(one) == counts(third)
    # a slice range, a sample offset and zero depth go through the same bookkeeping
    for kw2 in (dict(slice_begin=1, slice_end=3), dict(sample_offset=40), dict(max_depth=0), dict(max_depth=1)):
        args = dict(kw)
        args.update(kw2)
        one = gpu_ctx.render(device.make_params(45, 37, 9, flags=abi.FLAG_PERSISTENT, **args))
        two = gpu_ctx.render(device.make_params(45, 37, 9, flags=abi.FLAG_WAVEFRONT, **args))
        assert np.array_equal(one["slice_sums"], two["slice_sums"], equal_nan=True), kw2
        assert counts(one) == counts(two)
        auto = gpu_ctx.render(device.make_params(45, 37, 9, **args))
        assert np.array_equal(one["slice_sums"], auto["slice_sums"], equal_nan=True) and counts(one) == counts(auto), kw2


def test_wavefront_batches_overlap_and_accumulate_in_order(gpu_ctx, monkeypatch):
    """Many small batches (the workspace is capped for the test), alternating between the two workspaces, the
    rest of the paths and the accumulation of one running beside the bounces of the next: same bits as one lane per
    pixel summing its samples in order."""
    b, flat, _ = config_scene("suzanne")
    gpu_ctx.upload(flat)
    kw = dict(n_slices=2, max_depth=40, seed=31)
    one = gpu_ctx.render(device.make_params(40, 32, 46, flags=abi.FLAG_PERSISTENT, **kw))
    monkeypatch.setenv("RR_WF_WORKSPACE_BYTES", str(80 * 2 * 40 * 32 * 5))  # five samples per slot and batch: 5 batches of 23
    capped = device.Context(0)  # (a context reads the switch when it is created)
    try:
        capped.upload(flat)
        for flags in (abi.FLAG_WAVEFRONT, abi.FLAG_WAVEFRONT | abi.FLAG_NO_FINISH_KERNEL, abi.FLAG_PATHS):
            two = capped.render(device.make_params(40, 32, 46, flags=flags, **kw))
            assert np.array_equal(one["slice_sums"], two["slice_sums"], equal_nan=True)
            assert counts(one) == counts(two)
    finally:
        capped.close()


def test_path_granular_kernel_equals_item_granular_kernel(gpu_ctx, monkeypatch):
    """Scenes without BVH subtrees render with the persistent kernel that hands out single paths and adds the samples
    up afterwards (the default) or with the one whose lanes take whole (slice, pixel) items (RR_FLAG_PERSISTENT): same
    bits, also over many small batches, slice ranges, sample offsets, no samples at all and depth 0."""
    b, flat, _ = config_scene("cornell")
    gpu_ctx.upload(flat)
    kw = dict(n_slices=3, max_depth=30, seed=19)
    for extra in (dict(), dict(flags=abi.FLAG_COUNTERS), dict(slice_begin=1, slice_end=3), dict(sample_offset=77), dict(max_depth=0)):
        args = dict(kw)
        args.update(extra)
        flags = args.pop("flags", 0)
        one = gpu_ctx.render(device.make_params(53, 41, 14, flags=abi.FLAG_PERSISTENT | flags, **args))
        two = gpu_ctx.render(device.make_params(53, 41, 14, flags=flags, **args))
        assert np.array_equal(one["slice_sums"], two["slice_sums"], equal_nan=True), extra
        assert np.array_equal(one["rgb8"], two["rgb8"]) and counts(one) == counts(two), extra
    one = gpu_ctx.render(device.make_params(53, 41, 2, flags=abi.FLAG_PERSISTENT, **kw))  # fewer samples than slices: none are taken
    two = gpu_ctx.render(device.make_params(53, 41, 2, **kw))
    assert np.array_equal(one["slice_sums"], two["slice_sums"]) and two["counters"]["samples"] == 0
    one = gpu_ctx.render(device.make_params(53, 41, 60, flags=abi.FLAG_PERSISTENT, **kw))
    monkeypatch.setenv("RR_WF_WORKSPACE_BYTES", str(16 * 3 * 53 * 41 * 3))  # three samples per slot and batch: 7 batches of 20
    capped = device.Context(0)  # (a context reads the switch when it is created)
    try:
        capped.upload(flat)
        two = capped.render(device.make_params(53, 41, 60, **kw))
    finally:
        capped.close()
    assert np.array_equal(one["slice_sums"], two["slice_sums"], equal_nan=True) and counts(one) == counts(two)


@pytest.mark.parametrize("name,side,spp", [("suzanne", 640, 96), ("final", 512, 136)])
def test_wavefront_large_batch_equals_persistent_kernel(gpu_ctx, name, side, spp):
    """A batch of more than 2^25 paths, deep enough for many lanes of the kernel that finishes the paths to be parked
    in a BVH and to come back to it: same bits as the persistent kernel."""
    b, flat, _ = config_scene(name)
    gpu_ctx.upload(flat)
    kw = dict(n_slices=8, max_depth=25, seed=41)
    assert side * side * spp >= 1 << 25
    one = gpu_ctx.render(device.make_params(side, side, spp, flags=abi.FLAG_PERSISTENT, **kw))
    for flags in (abi.FLAG_WAVEFRONT, abi.FLAG_PATHS):
        two = gpu_ctx.render(device.make_params(side, side, spp, flags=flags, **kw))
        assert np.array_equal(one["slice_sums"], two["slice_sums"], equal_nan=True)
        assert np.array_equal(one["rgb8"], two["rgb8"])
        assert counts(one) == counts(two)


# --------------------------------------------------- size-independent properties
def test_render_is_deterministic_and_slices_compose(gpu_ctx):
    """Bit-identical reruns; rendering slice ranges separately equals rendering them together; a
    sample_offset render equals the matching slices of a larger render (how GPUs split one image)."""
    b, flat, _ = config_scene("cornell")
    gpu_ctx.upload(flat)
    K, spt, W, H = 4, 3, 120, 72
    full = gpu_ctx.render(device.make_params(W, H, K * spt, n_slices=K, max_depth=30, seed=5))
    again = gpu_ctx.render(device.make_params(W, H, K * spt, n_slices=K, max_depth=30, seed=5))
    assert np.array_equal(full["slice_sums"], again["slice_sums"], equal_nan=True)
    lo = gpu_ctx.render(device.make_params(W, H, K * spt, n_slices=K, max_depth=30, seed=5, slice_begin=0, slice_end=1))
    hi = gpu_ctx.render(device.make_params(W, H, K * spt, n_slices=K, max_depth=30, seed=5, slice_begin=1, slice_end=4))
    assert np.array_equal(np.concatenate([lo["slice_sums"], hi["slice_sums"]]), full["slice_sums"], equal_nan=True)
    assert np.allclose(lo["resolved"] + hi["resolved"], full["resolved"], rtol=1e-6)
    assert lo["counters"]["rays"] + hi["counters"]["rays"] == full["counters"]["rays"]
    half = K // 2
    off = gpu_ctx.render(device.make_params(W, H, half * spt, n_slices=half, max_depth=30, seed=5, sample_offset=half * spt))
    assert np.array_equal(off["slice_sums"], full["slice_sums"][half:], equal_nan=True)
    other_seed = gpu_ctx.render(device.make_params(W, H, K * spt, n_slices=K, max_depth=30, seed=6))
    assert not np.array_equal(other_seed["slice_sums"], full["slice_sums"])


@pytest.mark.parametrize("name", ["cornell", "suzanne", "spheres"])
def test_bands_of_rows_and_slice_halves_recompose_the_image(gpu_ctx, name):
    """How several GPUs share one image (rust_raytracer_amd.multi), on one GPU: bands of rows rendered on their own are
    bit for bit the rows of the whole render in every kernel organisation (random streams are keyed by the pixel's
    place in the whole image), and so are slice ranges; bands x slice halves together give back the 8-bit image."""
    from rust_raytracer_amd import multi
    b, flat, _ = config_scene(name)
    gpu_ctx.upload(flat)
    W, H, K, spp = 70, 45, 4, 12
    job = multi.job_params(W, H, spp, K, 25, 9)
    whole = gpu_ctx.render(job)
    for flags in (0, abi.FLAG_PERSISTENT, abi.FLAG_WAVEFRONT, abi.FLAG_PATHS, abi.FLAG_COUNTERS):
        for r0, r1 in ((0, 17), (17, 45), (44, 45), (5, 5)):
            band = gpu_ctx.render(device.make_params(W, H, spp, n_slices=K, max_depth=25, seed=9, flags=flags, row_begin=r0, row_end=r1))
            assert band["slice_sums"].shape == (K, r1 - r0, W, 3)
            assert np.array_equal(band["slice_sums"], whole["slice_sums"][:, r0:r1], equal_nan=True), (flags, r0, r1)
            assert np.array_equal(band["rgb8"], whole["rgb8"][r0:r1]) and np.array_equal(band["resolved"], whole["resolved"][r0:r1], equal_nan=True)
            assert band["counters"]["samples"] == (r1 - r0) * W * spp
    for world, partition in ((4, "both"), (3, "rows"), (2, "slices")):
        bands, groups = multi.grid(world, partition)
        resolved = np.zeros((H, W, 3), dtype=np.float32)
        rays = 0
        for rank in range(world):
            p = multi.rank_share(job, rank, world, partition)
            r = gpu_ctx.render(p)
            assert np.array_equal(r["slice_sums"], whole["slice_sums"][p.slice_begin:p.slice_end, p.row_begin:p.row_end], equal_nan=True)
            resolved[p.row_begin:p.row_end] += r["resolved"]  # the reduce over slice groups; bands are disjoint
            rays += r["counters"]["rays"]
        assert rays == whole["counters"]["rays"]
        if groups == 1:
            assert np.array_equal(resolved, whole["resolved"])
        assert np.abs(O.cast_u8(resolved).astype(int) - whole["rgb8"].astype(int)).max() <= 1
    with pytest.raises(device.DeviceError, match="row"):
        gpu_ctx.render(device.make_params(W, H, spp, n_slices=K, row_begin=10, row_end=H + 1))
    with pytest.raises(device.DeviceError, match="row"):
        gpu_ctx.render(device.make_params(W, H, spp, n_slices=K, row_begin=10, row_end=9))


@pytest.mark.parametrize("name", ["cornell", "final"])
def test_in_process_multi_device_entry(gpu_ctx, name):
    """rr_multi_* (several GPUs driven by one host thread) with this box's one GPU listed one, two, three and four times:
    every entry has a context, stream and buffers of its own, so the cuts, the peer copies, the adds and the row gather are the
    ones of a node with that many GPUs.  Bands of rows give back the single-context image bit for bit; slice groups up to the
    association of the f32 sum (the same cut as rust_raytracer_amd.multi makes for one process per GPU)."""
    from rust_raytracer_amd import multi
    b, flat, _ = config_scene(name)
    gpu_ctx.upload(flat)
    job = multi.job_params(90, 61, 24, 8, 30, 5)
    whole = gpu_ctx.render(job)
    for n_dev in (1, 2, 3, 4):
        mc = device.MultiContext([0] * n_dev)
        mc.upload(flat)
        for split, part in ((abi.SPLIT_SLICES, "slices"), (abi.SPLIT_ROWS, "rows"), (abi.SPLIT_BOTH, "both")):
            r = mc.render(job, split)
            assert counts(r) == counts(whole), (n_dev, part)
            if multi.grid(n_dev, part)[1] == 1:
                assert np.array_equal(r["resolved"], whole["resolved"]) and np.array_equal(r["linear"], whole["linear"], equal_nan=True)
                assert np.array_equal(r["rgb8"], whole["rgb8"])
            else:
                assert np.allclose(r["resolved"], whole["resolved"], rtol=1e-6, atol=1e-4) and np.allclose(r["linear"], whole["linear"], rtol=1e-6, atol=1e-6)
                assert np.abs(r["rgb8"].astype(int) - whole["rgb8"].astype(int)).max() <= 1
            again = mc.render(job, split)  # buffers and contexts are reused
            assert np.array_equal(again["resolved"], r["resolved"]) and np.array_equal(again["rgb8"], r["rgb8"])
        small = multi.job_params(33, 20, 8, 4, 10, 6)  # a smaller job on the same object
        assert np.array_equal(mc.render(small, abi.SPLIT_ROWS)["rgb8"], gpu_ctx.render(small)["rgb8"])
        mc.close()
    mc = device.MultiContext([0, 0])
    with pytest.raises(device.DeviceError, match="no scene"):
        mc.render(job)
    mc.upload(flat)
    with pytest.raises(device.DeviceError, match="whole image"):
        mc.render(device.make_params(90, 61, 24, n_slices=8, slice_begin=1, slice_end=8))
    mc.close()
    with pytest.raises(device.DeviceError):
        device.MultiContext([0, 99])


def test_in_process_sum_over_rccl(gpu_ctx):
    """The RCCL route of rr_multi_render (ncclReduce per band, librccl.so opened at run time).  This box has one GPU and a
    communicator cannot hold a device twice, so what runs here is the binding itself — dlopen, ncclCommInitAll, a grouped
    ncclReduce of `resolved` and `linear` in one call on the render's stream, communicator reuse and destruction — on
    bands of ONE device (forced mode sends those through the reduce too: a copy in place), against the single-context image;
    and the decisions: a device listed twice is refused in forced mode and takes the peer-copy sum in auto mode.  The sum of
    two or more GPUs over xGMI is unmeasured on hardware."""
    from rust_raytracer_amd import multi
    b, flat, _ = config_scene("cornell")
    gpu_ctx.upload(flat)
    job = multi.job_params(90, 61, 24, 8, 30, 5)
    whole = gpu_ctx.render(job)
    mc = device.MultiContext([0], reduce="rccl")
    mc.upload(flat)
    for split in (abi.SPLIT_SLICES, abi.SPLIT_ROWS):
        for _ in range(2):  # the second call reuses the communicator
            r = mc.render(job, split)
            assert mc.info()["reduce"] == "rccl"
            assert np.array_equal(r["resolved"], whole["resolved"]) and np.array_equal(r["linear"], whole["linear"], equal_nan=True)
            assert np.array_equal(r["rgb8"], whole["rgb8"]) and counts(r) == counts(whole)
    small = multi.job_params(33, 20, 8, 4, 10, 6)
    assert np.array_equal(mc.render(small, abi.SPLIT_SLICES)["rgb8"], gpu_ctx.render(small)["rgb8"])
    mc.set_reduce("peer")
    assert np.array_equal(mc.render(job, abi.SPLIT_SLICES)["rgb8"], whole["rgb8"]) and mc.info()["reduce"] == "peer"
    mc.close()
    twice = device.MultiContext([0, 0], reduce="rccl")
    twice.upload(flat)
    with pytest.raises(device.DeviceError, match="distinct devices"):
        twice.render(job, abi.SPLIT_SLICES)
    r = twice.render(job, abi.SPLIT_ROWS)  # two bands of one device each: two communicators of one rank
    assert np.array_equal(r["rgb8"], whole["rgb8"]) and twice.info()["reduce"] == "rccl"
    twice.set_reduce("auto")
    r = twice.render(job, abi.SPLIT_SLICES)
    assert twice.info()["reduce"] == "peer" and np.abs(r["rgb8"].astype(int) - whole["rgb8"].astype(int)).max() <= 1
    twice.close()


def test_bench_drives_the_number_of_gpus_it_is_asked_for():
    """bench.py on this one-GPU box: --gpus 2 must fail (one GPU visible) without printing a result line; the in-process
    route over the device listed twice reports two ranks on one GPU."""
    import json
    import subprocess
    import sys
    bench = os.path.join(abi.REPO_DIR, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--workload", "cornell", "--spp", "16"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "refusing to measure fewer GPUs than asked" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    r = subprocess.run([sys.executable, bench, "--inprocess", "--devices", "0,0", "--workload", "cornell", "--spp", "64", "--steps", "2",
                        "--warmup", "1"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_ranks_seen"] == 2 and out["n_gpus"] == 1 and out["scaling"] == "strong"
    assert out["value"] > 100.0 and "slices" in out["config"]["parallelism"]


def test_bench_rank_path_rehearsed_with_ranks_sharing_the_gpu():
    """The one-rank-per-GPU route of bench.py (launcher -> N ranks -> each renders its share -> reduce / gather -> rank 0 casts)
    run end to end with 2 and 4 ranks that share this box's one GPU: --share-gpus over gloo, the images staged through the
    host (RCCL cannot hold one GPU twice, so its transport is the one thing this does not run).  Bands of rows give the
    one-rank image bit for bit; slice groups give it up to the association of the f32 sum (8-bit values within 1)."""
    import json
    import subprocess
    import sys
    bench = os.path.join(abi.REPO_DIR, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    common = ["--workload", "cornell", "--spp", "64", "--steps", "1", "--warmup", "1", "--no-cpu", "--no-count"]

    def run(extra):
        r = subprocess.run([sys.executable, bench] + extra + common, capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])

    one = run([])
    assert one["n_gpus"] == 1 and one["n_ranks_seen"] == 1
    rows = run(["--gpus", "2", "--share-gpus", "--backend", "gloo", "--partition", "rows", "--no-weak"])
    assert rows["n_ranks_seen"] == 2 and rows["n_gpus"] == 1 and "REHEARSAL" in rows["config"]["parallelism"]
    assert rows["image"]["sha256"] == one["image"]["sha256"]
    assert len(rows["kernel_ms_per_rank"]) == 2
    sl = run(["--gpus", "2", "--share-gpus", "--backend", "gloo"])  # the defaults of an N-GPU run: strong over slices, then the weak job
    assert sl["n_ranks_seen"] == 2 and sl["scaling"] == "strong" and "slices x2" in sl["config"]["parallelism"]
    assert sl["config"]["spp"] == 64 and sl["weak"]["value"] > 0 and "128 spp in 16 slices" in sl["weak"]["workload"]
    assert np.allclose(sl["image"]["mean_rgb"], one["image"]["mean_rgb"], atol=0.01)
    both = run(["--gpus", "4", "--share-gpus", "--backend", "gloo", "--partition", "both", "--no-weak"])  # 2 bands x 2 slice groups: sub-groups
    assert both["n_ranks_seen"] == 4 and "2 bands x 2 slice groups" in both["config"]["parallelism"]
    assert np.allclose(both["image"]["mean_rgb"], one["image"]["mean_rgb"], atol=0.01)
    # without --share-gpus the same command is refused (one GPU visible), and gloo is not a backend N GPUs are measured over
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--backend", "gloo"] + common, capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_full_size_cornell_configuration(gpu_ctx):
    """The headline benchmark configuration at full size (Cornell box, 600x600, 1000 spp, depth 200): every sample is traced,
    nothing is NaN, the image energy equals a low-spp oracle render's within Monte-Carlo error."""
    b, flat, _ = config_scene("cornell")
    gpu_ctx.upload(flat)
    p = device.make_params(600, 600, 1000, n_slices=8, max_depth=200, seed=1)
    r = gpu_ctx.render(p, slice_sums=False)
    assert r["counters"]["samples"] == 600 * 600 * 1000
    assert 2.2 < r["counters"]["rays"] / r["counters"]["samples"] < 2.7  # (2.9 with dead paths walked to their end, as the reference does)
    t = gpu_ctx.render(device.make_params(600, 600, 64, n_slices=8, max_depth=200, seed=1, flags=abi.FLAG_TRACE_DEAD_PATHS), slice_sums=False)
    assert 2.7 < t["counters"]["rays"] / t["counters"]["samples"] < 3.2
    assert np.isfinite(r["linear"]).all() and r["rgb8"].max() == 255
    ref = O.render(flat, O.make_params(600, 600, 16, n_slices=8, max_depth=200, seed=77), n_threads=16)
    m_gpu, m_ref = r["linear"].mean() / 1000.0, ref["linear"].mean() / 16.0
    assert abs(m_gpu - m_ref) < 0.01 * m_ref, (m_gpu, m_ref)
    # per pixel: the 1000-spp image is the converged version of the 16-spp one.  The oracle's two 8-spp halves (its slices,
    # independent estimates of every pixel) give the noise of its 16-spp mean: var(mean16) ~ (a - b)^2 / 4, and the GPU
    # image's own noise is 1/62 of that.  The mean squared difference must sit at that level, not merely below a constant.
    a, b2 = ref["slice_sums"][:4].sum(axis=0) / 8.0, ref["slice_sums"][4:].sum(axis=0) / 8.0
    noise_var = float(np.mean((a - b2) ** 2) / 4.0)
    mse = float(np.mean((r["linear"] / 1000.0 - ref["linear"] / 16.0) ** 2))
    assert 0.7 * noise_var < mse < 1.4 * noise_var, (mse, noise_var)


@pytest.mark.parametrize("name,depth,spp", [("suzanne", 100, 1000), ("final", 50, 10000)])
def test_full_size_bvh_configurations(gpu_ctx, name, depth, spp):
    """The BVH configurations at their full size (Suzanne 800x800 at 1000 spp, final_scene 800x800 at 10 000 spp: several
    batches of samples), through properties that do not need the oracle: every sample is traced, reruns are identical,
    two halves of the slices rendered on their own add up to the whole render bit for bit, and the image energy equals
    that of a render with other random numbers within Monte-Carlo error."""
    b, flat, _ = config_scene(name)
    gpu_ctx.upload(flat)
    kw = dict(n_slices=8, max_depth=depth, seed=1)
    whole = gpu_ctx.render(device.make_params(800, 800, spp, **kw), slice_sums=False)
    assert whole["counters"]["samples"] == 800 * 800 * spp
    assert 1.2 < whole["counters"]["rays"] / whole["counters"]["samples"] < 5.0  # (final_scene: 1.29; 3.23 with dead paths walked to their end)
    assert np.isfinite(whole["linear"]).all() and whole["rgb8"].max() > 100
    again = gpu_ctx.render(device.make_params(800, 800, spp, **kw), slice_sums=False)
    assert np.array_equal(whole["rgb8"], again["rgb8"]) and np.array_equal(whole["linear"], again["linear"])
    lo = gpu_ctx.render(device.make_params(800, 800, spp, slice_begin=0, slice_end=4, **kw), slice_sums=False)
    hi = gpu_ctx.render(device.make_params(800, 800, spp, slice_begin=4, slice_end=8, **kw), slice_sums=False)
    assert lo["counters"]["samples"] + hi["counters"]["samples"] == whole["counters"]["samples"]
    assert lo["counters"]["rays"] + hi["counters"]["rays"] == whole["counters"]["rays"]
    assert np.allclose(lo["linear"] + hi["linear"], whole["linear"], rtol=1e-6, atol=1e-6)
    assert np.allclose(lo["resolved"] + hi["resolved"], whole["resolved"], rtol=1e-6, atol=1e-6)
    other = gpu_ctx.render(device.make_params(800, 800, 64, n_slices=8, max_depth=depth, seed=99), slice_sums=False)
    m_whole, m_other = whole["linear"].mean() / spp, other["linear"].mean() / 64.0
    assert abs(m_whole - m_other) < 0.02 * m_other, (m_whole, m_other)
    # the CPU side of the same check: the oracle's render of the same camera at 200x200 (a regular subsample of the 800x800
    # pixels' directions up to the jitter) and 64 spp with its own random numbers has the same image energy
    ref = O.render(flat, O.make_params(200, 200, 64, n_slices=8, max_depth=depth, seed=77), n_threads=16)
    m_ref = float(np.nanmean(ref["linear"])) / 64.0
    assert np.isfinite(ref["linear"]).all()
    assert abs(m_whole - m_ref) < 0.01 * m_ref, (m_whole, m_ref)


@pytest.mark.parametrize("name,side,spp,depth", [("cornell", 96, 512, 50), ("suzanne", 64, 256, 50), ("final", 48, 256, 30)])
def test_converged_images_agree_within_rmse(gpu_ctx, name, side, spp, depth):
    """Per-channel RMSE of the displayed-range linear radiance (mean sample value clamped to [0, 1], as
    render.rs:110-117 clamps before display; image means are 0.10-0.13) between the GPU render and the oracle's:
      * same seed, matched spp: <= 0.002 (measured 1e-5 .. 8e-4: only paths that a last-bit difference in a
        transcendental sent elsewhere differ);
      * different seeds, matched spp: <= 0.06 (measured 0.014 .. 0.036: Monte-Carlo noise of two independent
        estimates), and the image means agree within 1.5 %."""
    b, flat, _ = config_scene(name)
    gpu_ctx.upload(flat)
    K = 4
    got = gpu_ctx.render(device.make_params(side, side, spp, n_slices=K, max_depth=depth, seed=101))
    same = O.render(flat, O.make_params(side, side, spp, n_slices=K, max_depth=depth, seed=101), n_threads=16)
    other = O.render(flat, O.make_params(side, side, spp, n_slices=K, max_depth=depth, seed=202), n_threads=16)
    ok = ~(np.isnan(got["slice_sums"]).any(axis=0) | np.isnan(same["slice_sums"]).any(axis=0) |
           np.isnan(other["slice_sums"]).any(axis=0))
    assert ok.mean() > 0.99
    a, s_, o_ = [np.clip(r["linear"] / spp, 0.0, 1.0) for r in (got, same, other)]
    for ch in range(3):
        sel = ok[..., ch]
        rmse_same = float(np.sqrt(np.mean((a[..., ch][sel] - s_[..., ch][sel]) ** 2)))
        rmse_other = float(np.sqrt(np.mean((a[..., ch][sel] - o_[..., ch][sel]) ** 2)))
        assert rmse_same <= 0.002, (name, ch, rmse_same)
        assert rmse_other <= 0.06, (name, ch, rmse_other)
    assert abs(float(a[ok].mean()) - float(o_[ok].mean())) <= 0.015 * float(o_[ok].mean())


def test_config5_at_full_size(gpu_ctx):
    """The 1 000 000-triangle height field with 100 000 spheres (one BVH of 1.1 M primitives, 20 levels) at 3840x2160:
    closest hits of 20 000 rays (camera rays over the whole image and the adversarial set) identical to the oracle's, which
    makes every visit of the reference (about a million records per ray: triangles accept any t), so the production
    traversal's culling and visit skipping are checked at this density; then a render of the full image through size-independent
    properties: every sample traced, reruns identical, bands of rows and slice halves compose bit for bit."""
    b, flat, d = host.build_config_scene("mesh")
    assert (flat.n_triangles, flat.n_spheres, d["width"], d["height"]) == (1_000_000, 100_000, 3840, 2160)
    info = gpu_ctx.upload(flat)
    assert info["bvh_nodes_by_class"][3] > 1_000_000  # culled against [0, inf): triangles below
    W, H = 3840, 2160
    p = O.make_params(W, H, 4, n_slices=1, seed=11)
    pixels = [(int(i * (W - 1) / 119), int(j * (H - 1) / 99)) for j in range(100) for i in range(120)]
    rays = np.concatenate([O.camera_rays(flat, p, pixels, sample=1), _stress_rays(flat, 8000, seed=21)])
    assert len(rays) == 20_000
    got = gpu_ctx.primary_hits(rays, seed=4)
    want = O.primary_hits(flat, rays, seed=4)
    assert np.array_equal(got["prim_id"], want["prim_id"])
    compare_hits(got, want, exact_t=True)
    assert (want["prim_id"] >= 0).mean() > 0.1
    kw = dict(n_slices=2, max_depth=50, seed=1)
    whole = gpu_ctx.render(device.make_params(W, H, 2, **kw))
    assert whole["counters"]["samples"] == W * H * 2 and whole["counters"]["rays"] >= whole["counters"]["samples"]
    assert np.isfinite(whole["linear"]).all() and whole["rgb8"].max() > 50
    again = gpu_ctx.render(device.make_params(W, H, 2, **kw), slice_sums=False)
    assert np.array_equal(whole["rgb8"], again["rgb8"]) and np.array_equal(whole["linear"], again["linear"])
    top = gpu_ctx.render(device.make_params(W, H, 2, row_begin=0, row_end=1000, **kw))
    rest = gpu_ctx.render(device.make_params(W, H, 2, row_begin=1000, row_end=H, **kw))
    assert np.array_equal(np.concatenate([top["slice_sums"], rest["slice_sums"]], axis=1), whole["slice_sums"], equal_nan=True)
    assert np.array_equal(np.concatenate([top["rgb8"], rest["rgb8"]]), whole["rgb8"])
    lo = gpu_ctx.render(device.make_params(W, H, 2, slice_begin=0, slice_end=1, **kw))
    hi = gpu_ctx.render(device.make_params(W, H, 2, slice_begin=1, slice_end=2, **kw))
    assert np.array_equal(np.concatenate([lo["slice_sums"], hi["slice_sums"]]), whole["slice_sums"], equal_nan=True)
    assert lo["counters"]["rays"] + hi["counters"]["rays"] == whole["counters"]["rays"]
    # the image above came from the wavefront pipeline with the wide node move (both children's boxes tested at their parent):
    # the one-box move and the path kernel (either move) give the same bits on a quarter-size image of the same camera
    assert info["wide_nodes"] and info["wavefront"]
    q = dict(n_slices=2, max_depth=50, seed=2)
    small = gpu_ctx.render(device.make_params(960, 540, 2, **q))
    for flags in (abi.FLAG_NARROW_NODES, abi.FLAG_PATHS, abi.FLAG_PATHS | abi.FLAG_NARROW_NODES):
        other = gpu_ctx.render(device.make_params(960, 540, 2, flags=flags, **q))
        assert np.array_equal(small["slice_sums"], other["slice_sums"], equal_nan=True), flags
        assert other["counters"]["rays"] == small["counters"]["rays"]


@pytest.mark.parametrize("name,kw", [("spheres", {}), ("suzanne", {}), ("mesh", dict(n_triangles=20000, n_spheres=2000))])
def test_sah_tree_gives_the_same_hits_under_the_intended_semantics(gpu_ctx, name, kw):
    """The labelled SAH builder (host.SceneBuilder(bvh="sah"), not the reference's tree): under the intended semantics a BVH
    only accelerates, so the device finds the same closest hits in either tree (the oracle too), with fewer primitive
    tests in the SAH tree; the device's hits in the SAH tree are the oracle's bit for bit."""
    rays = None
    out = {}
    for bvh in ("reference", "sah"):
        b, flat, _ = host.build_config_scene(name, bvh=bvh, **kw)
        flat = intended(flat)
        if rays is None:
            rays = _stress_rays(flat, 40_000, seed=77)
        gpu_ctx.upload(flat)
        got = gpu_ctx.primary_hits(rays, seed=4)
        want, cnt = O.primary_hits(flat, rays, seed=4, counters=True)
        compare_hits(got, want, exact_t=True)
        out[bvh] = (got, cnt)
    (a, ca), (s_, cs) = out["reference"], out["sah"]
    # (rays with a zero direction component can be "hit" at t = NaN by whatever is visited: they have no closest hit to agree on)
    sane = (rays[:, 3:6] != 0).all(axis=1) & np.isfinite(a["t"]) & np.isfinite(s_["t"])
    hit = (a["prim_id"] >= 0) & sane
    assert hit.mean() > 0.05 and np.array_equal((s_["prim_id"] >= 0) & sane, hit)
    assert np.array_equal(a["t"][hit], s_["t"][hit])
    if name == "mesh":  # (what the tree saves on the paths of a render: profiles/sah_cost.sh; this ray set is adversarial)
        assert cs["prim_tests"] < ca["prim_tests"] and cs["box_tests"] < ca["box_tests"]


# ------------------------------------------------- the labelled book estimator
def test_book_semantics_match_the_fixture_the_oracle_and_every_kernel_form(gpu_ctx):
    """RR_SEMANTICS_BOOK (never the default): specular materials keep the ray they scattered, `lights` is a list of three
    members, the traversal is the intended one.  The device against pyoracle's fixture and against the C++ oracle sample
    by sample, every kernel organisation the same bits, and the list of lights against a single light."""
    import golden_paths as GP
    from test_golden_cpu import compare_samples
    fx = np.load(f"{G.GOLDEN_DIR}/paths_box_book.npz")
    b, flat = GP.build_host(book=True)
    info = gpu_ctx.upload(flat)
    R = GP.RENDER
    kw = dict(n_slices=R["spp"], max_depth=R["max_depth"], seed=R["seed"])
    full = gpu_ctx.render(device.make_params(R["W"], R["H"], R["spp"], flags=abi.FLAG_TRACE_DEAD_PATHS, **kw))
    compare_samples(full["slice_sums"], fx["samples"], 0.985)  # (the fixture is the full walk, dead paths included)
    got = gpu_ctx.render(device.make_params(R["W"], R["H"], R["spp"], **kw))
    want = O.render(flat, O.make_params(R["W"], R["H"], R["spp"], **kw), n_threads=8, counters=True)
    compare_samples(got["slice_sums"], want["slice_sums"], 0.985)
    # (16 000 rays: the few paths a last-bit difference in sin/cos sends elsewhere are a percent of them)
    assert abs(got["counters"]["rays"] - want["counters"]["rays"]) < 0.02 * want["counters"]["rays"]
    for flags in (abi.FLAG_PATHS, abi.FLAG_PERSISTENT, abi.FLAG_WAVEFRONT, abi.FLAG_WAVEFRONT | abi.FLAG_NO_FINISH_KERNEL, abi.FLAG_COUNTERS):
        other = gpu_ctx.render(device.make_params(R["W"], R["H"], R["spp"], flags=flags, **kw))
        assert np.array_equal(got["slice_sums"], other["slice_sums"], equal_nan=True), flags
    # a larger render of the same scene: the device against the oracle, and not the image of the reference's semantics
    big = dict(n_slices=4, max_depth=20, seed=8)
    a = gpu_ctx.render(device.make_params(96, 80, 16, **big))
    o = O.render(flat, O.make_params(96, 80, 16, **big), n_threads=16)
    both = ~np.isnan(a["slice_sums"]) & ~np.isnan(o["slice_sums"])
    assert (np.isnan(a["slice_sums"]) == np.isnan(o["slice_sums"])).mean() > 0.995 and both.mean() > 0.3
    assert np.isclose(a["slice_sums"][both], o["slice_sums"][both], rtol=1e-3, atol=1e-4).mean() > 0.97
    b2, flat2 = GP.build_host()
    gpu_ctx.upload(flat2)
    ref = gpu_ctx.render(device.make_params(96, 80, 16, **big))
    assert np.nanmean(a["slice_sums"]) > 1.2 * np.nanmean(ref["slice_sums"])  # glass, metal and smoke carry light now


# ------------------------------------------------------------------ edge cases
@pytest.mark.parametrize("W,H", [(2, 2), (37, 21), (8, 8), (65, 9)])
def test_odd_image_sizes(gpu_ctx, W, H):
    b, flat, _ = config_scene("cornell")
    p = device.make_params(W, H, 4, n_slices=2, max_depth=20, seed=13)
    got, want = render_both(gpu_ctx, flat, p)
    assert got["slice_sums"].shape == (2, H, W, 3)
    assert_render_parity(got, want, min_close=0.95, mean_tol=0.05)


@pytest.mark.parametrize("W,H", [(2, 2), (7, 3), (8, 8), (9, 8), (8, 9), (37, 21), (65, 9), (16, 24), (100, 41)])
def test_tile_order_of_the_path_kernel_changes_no_bit(gpu_ctx, W, H):
    """On scenes with BVH subtrees the path kernel hands out pixels in 8x8 tiles (then the columns and the rows that tiles do not
    cover): the same bits as the row order (RR_FLAG_ROW_ORDER) and as the item-granular kernel, for image sizes with and
    without remainders, for bands of rows that start and end anywhere, slice ranges and sample offsets; and the entry point
    that allocates nothing takes the same order."""
    b, flat, _ = config_scene("final")
    gpu_ctx.upload(flat)
    kw = dict(n_slices=4, max_depth=12, seed=21)
    base = gpu_ctx.render(device.make_params(W, H, 8, **kw))
    for flags in (abi.FLAG_ROW_ORDER, abi.FLAG_PERSISTENT, abi.FLAG_PATHS, abi.FLAG_PATHS | abi.FLAG_ROW_ORDER | abi.FLAG_COUNTERS):
        other = gpu_ctx.render(device.make_params(W, H, 8, flags=flags, **kw))
        assert np.array_equal(base["slice_sums"], other["slice_sums"], equal_nan=True), flags
        assert np.array_equal(base["rgb8"], other["rgb8"])
    if H >= 9:
        r0, r1 = 3, H - 2
        band = gpu_ctx.render(device.make_params(W, H, 8, row_begin=r0, row_end=r1, slice_begin=1, slice_end=3, sample_offset=7, **kw))
        rows = gpu_ctx.render(device.make_params(W, H, 8, row_begin=r0, row_end=r1, slice_begin=1, slice_end=3, sample_offset=7,
                                                 flags=abi.FLAG_ROW_ORDER, **kw))
        assert np.array_equal(band["slice_sums"], rows["slice_sums"], equal_nan=True)
        whole = gpu_ctx.render(device.make_params(W, H, 8, sample_offset=7, **kw))
        assert np.array_equal(band["slice_sums"], whole["slice_sums"][1:3, r0:r1], equal_nan=True)


@pytest.mark.parametrize("W,H,spp,slices", [(64, 64, 8, 4), (61, 37, 8, 4), (7, 50, 6, 2), (33, 8, 12, 4), (130, 17, 3, 3), (2, 16, 8, 8), (96, 80, 1, 1),
                                            (1, 9, 16, 4)])
def test_cell_order_of_the_pipeline_changes_no_bit(gpu_ctx, W, H, spp, slices):
    """The wavefront pipeline numbers its paths strip by strip (8 image rows) and, inside a full strip, cell by cell: all
    samples of a 2x2 cell of pixels before the next cell's, so that a traversal wave's 64 rays leave through a few
    neighbouring pixels; columns that cells do not cover and a last strip of fewer rows follow in row order.  Whatever the
    sizes, bands and slice ranges: the path kernel's bits."""
    b, flat, _ = config_scene("suzanne")
    gpu_ctx.upload(flat)
    kw = dict(n_slices=slices, max_depth=10, seed=33)
    want = gpu_ctx.render(device.make_params(W, H, spp, flags=abi.FLAG_PATHS | abi.FLAG_ROW_ORDER, **kw))
    got = gpu_ctx.render(device.make_params(W, H, spp, flags=abi.FLAG_WAVEFRONT, **kw))
    assert np.array_equal(want["slice_sums"], got["slice_sums"], equal_nan=True)
    assert np.array_equal(want["rgb8"], got["rgb8"])
    assert want["counters"]["rays"] == got["counters"]["rays"] and want["counters"]["samples"] == got["counters"]["samples"]
    if H >= 9 and slices >= 3:
        band = dict(row_begin=2, row_end=H - 3, slice_begin=1, slice_end=slices - 1, sample_offset=5)
        want = gpu_ctx.render(device.make_params(W, H, spp, flags=abi.FLAG_PATHS | abi.FLAG_ROW_ORDER, **band, **kw))
        got = gpu_ctx.render(device.make_params(W, H, spp, flags=abi.FLAG_WAVEFRONT, **band, **kw))
        assert np.array_equal(want["slice_sums"], got["slice_sums"], equal_nan=True)


def test_degenerate_parameters(gpu_ctx):
    b, flat, _ = config_scene("cornell")
    gpu_ctx.upload(flat)
    r = gpu_ctx.render(device.make_params(16, 16, 3, n_slices=4, max_depth=10))        # 3 / 4 = 0 samples per slice
    assert (r["slice_sums"] == 0).all() and np.isnan(r["resolved"]).all() and (r["rgb8"] == 0).all()
    r = gpu_ctx.render(device.make_params(16, 16, 4, n_slices=2, max_depth=0))         # depth 0: black, no rays
    assert (r["slice_sums"] == 0).all() and r["counters"]["rays"] == 0 and r["counters"]["samples"] == 16 * 16 * 4
    r = gpu_ctx.render(device.make_params(16, 16, 4, n_slices=2, max_depth=5, slice_begin=1, slice_end=1))   # empty range
    assert r["slice_sums"].shape[0] == 0 and r["counters"]["samples"] == 0
    # empty scene: every sample returns the background (render.rs:39)
    e = host.SceneBuilder(1)
    e.set_camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 40.0, 0.0, 1.0, 1.0)
    e.set_background((0.25, 0.5, 1.0))
    gpu_ctx.upload(e.flatten())
    r = gpu_ctx.render(device.make_params(8, 8, 6, n_slices=2, max_depth=5))
    assert np.array_equal(r["slice_sums"], np.broadcast_to(np.float32([0.75, 1.5, 3.0]), (2, 8, 8, 3)))
    assert list(r["rgb8"][0, 0]) == [128, 181, 255]   # 2 slices x 256 * sqrt(c) / 2, saturating cast


def test_errors_are_reported_not_swallowed(gpu_ctx):
    fresh = device.Context(0)
    with pytest.raises(device.DeviceError, match="no scene"):
        fresh.render(device.make_params(8, 8, 4))
    b, flat, _ = host.build_config_scene("cornell")
    bad = abi.FlatScene.from_buffer_copy(flat)
    refs = (C.c_uint32 * flat.n_child_refs)(*[flat.child_refs[i] for i in range(flat.n_child_refs)])
    refs[0] = abi.ref(abi.TAG_SPHERE, 5)             # no spheres in this scene
    bad.child_refs = C.cast(refs, C.POINTER(C.c_uint32))
    with pytest.raises(device.DeviceError, match="out of range"):
        fresh.upload(bad)
    refs[0] = flat.root                               # the root list inside itself: a cycle
    with pytest.raises(device.DeviceError, match="tree|twice|deeper"):
        fresh.upload(bad)
    with pytest.raises(device.DeviceError):
        fresh.render(device.make_params(8, 8, 4))     # the failed uploads left no scene behind
    fresh.upload(flat)
    with pytest.raises(device.DeviceError, match="slice"):
        fresh.render(device.make_params(8, 8, 4, n_slices=2, slice_begin=1, slice_end=3))
    d = host.SceneBuilder(1)
    m = d.lambertian(d.solid_color((1, 1, 1)))
    g = d.sphere((0, 0, 0), 1.0, m)
    for _ in range(9):
        g = d.instance_translation(g, (1, 0, 0))
    d.add(g)
    with pytest.raises(device.DeviceError, match="nested deeper"):
        fresh.upload(d.flatten())
    fresh.close()

"""One small scene with every material and texture kind, built twice: for the numpy restatement (oracle/pyoracle.py)
and through the host library (→ rr_flat_scene for the C++ oracle and the GPU).  tests/golden/make_golden.py renders it
sample by sample with the numpy restatement and stores the radiances in tests/golden/paths_box.npz."""
import numpy as np

SEED = 9           # host-side random numbers: Perlin tables first, then the BVH split axes
RENDER = dict(W=24, H=20, spp=8, max_depth=12, seed=5)   # 8 slices of one sample each
CAM = dict(look_from=(278, 278, -800), look_at=(278, 278, 0), vup=(0, 1, 0), vfov=40.0, aperture=12.0, focus_dist=800.0,
           aspect=1.2)
LIGHT = (163.0, 393.0, 177.0, 382.0, 554.0)  # x0, x1, z0, z1, k of the XZRect passed as `lights`
# `lights` of the render under the labelled book estimator (RR_SEMANTICS_BOOK): a ColliderList of the lamp, a second XZRect
# next to it (not part of the scene: sampling it finds whatever lies that way) and a sphere (no XZRect: pdf 0, sample (1,0,0))
LIGHT_LIST = [LIGHT, (60.0, 140.0, 400.0, 480.0, 554.0), None]
BACKGROUND = (0.02, 0.03, 0.05)
IMAGE = (np.arange(4 * 5 * 3, dtype=np.uint32).reshape(4, 5, 3) * 37 % 256).astype(np.uint8)

WHITE, RED, GREEN = (0.73, 0.73, 0.73), (0.65, 0.05, 0.05), (0.12, 0.45, 0.15)


def build_py():
    import pyoracle as P
    rng = P.PCG32(SEED)
    noise = ("noise", P.perlin_tables(rng), 0.05, 4)

    def with_m(g, m):
        g.material = m
        return g

    lam = lambda rgb: ("lambertian", ("solid", rgb))
    small = [with_m(P.sphere((120 + 60 * i, 60 + 25 * i, 120 + 50 * (i % 3)), 28.0),
                    ("lambertian", ("image", IMAGE)) if i == 2 else lam((0.2 + 0.15 * i, 0.6, 0.9 - 0.15 * i))) for i in range(5)]
    objs = [with_m(P.yzrect(0, 555, 0, 555, 555), lam(GREEN)), with_m(P.yzrect(0, 555, 0, 555, 0), lam(RED)),
            with_m(P.xzrect(*LIGHT), ("emissive", ("solid", (15.0, 15.0, 15.0)))),
            with_m(P.xzrect(0, 555, 0, 555, 0), ("lambertian", ("checker", (0.2, 0.3, 0.1), (0.9, 0.9, 0.9)))),
            with_m(P.xzrect(0, 555, 0, 555, 555), lam(WHITE)), with_m(P.xyrect(0, 555, 0, 555, 555), ("lambertian", noise)),
            P.translate(P.rotate(with_m(P.cuboid((0, 0, 0), (165, 330, 165)), lam(WHITE)), 1, 15.0), (265, 0, 295)),
            with_m(P.sphere((190, 90, 190), 90.0), ("dielectric", 1.5)),
            with_m(P.sphere((420, 400, 300), 70.0), ("glossy", (0.8, 0.85, 0.88), 0.3)),
            with_m(P.constant_medium(P.sphere((400, 120, 150), 80.0), 0.01), ("isotropic", ("solid", (0.9, 0.9, 0.9)))),
            with_m(P.moving_sphere((100, 400, 200), (130, 400, 200), 0.0, 1.0, 50.0), lam((0.7, 0.3, 0.1))),
            P.bvh_node(small, 0.0, 1.0, rng)]
    P.number_primitives(objs)
    cam = P.Camera(CAM["look_from"], CAM["look_at"], CAM["vup"], CAM["vfov"], CAM["aperture"], CAM["focus_dist"], CAM["aspect"], 0.0, 1.0)
    return objs, P.materials_by_prim(objs), cam


def build_host(light=True, book=False):
    """light=False: an empty list is passed as `lights` (as random_spheres does, main.rs:111): its pdf is 0 and its
    sample (1,0,0), so many samples are 0/0 = NaN (render.rs:34).  book=True: LIGHT_LIST is passed, and the flattened
    scene asks for RR_SEMANTICS_BOOK."""
    import rust_raytracer_amd  # noqa: F401
    from rust_raytracer_amd import host
    b = host.SceneBuilder(SEED)
    noise = b.lambertian(b.noise(0.05, 4))  # first consumer of the builder's random numbers, as in build_py
    lam = lambda rgb: b.lambertian(b.solid_color(rgb))
    img = b.lambertian(b.load_image_rgb8(IMAGE))
    small = [b.sphere((120 + 60 * i, 60 + 25 * i, 120 + 50 * (i % 3)), 28.0, img if i == 2 else lam((0.2 + 0.15 * i, 0.6, 0.9 - 0.15 * i)))
             for i in range(5)]
    lamp = b.xzrect(*LIGHT, b.emissive((15.0, 15.0, 15.0)))
    for g in [b.yzrect(0, 555, 0, 555, 555, lam(GREEN)), b.yzrect(0, 555, 0, 555, 0, lam(RED)), lamp,
              b.xzrect(0, 555, 0, 555, 0, b.lambertian(b.checkered((0.2, 0.3, 0.1), (0.9, 0.9, 0.9)))),
              b.xzrect(0, 555, 0, 555, 555, lam(WHITE)), b.xyrect(0, 555, 0, 555, 555, noise),
              b.instance_translation(b.instance_rotation(b.cuboid((0, 0, 0), (165, 330, 165), lam(WHITE)), 1, 15.0), (265, 0, 295)),
              b.sphere((190, 90, 190), 90.0, b.dielectric(1.5)), b.sphere((420, 400, 300), 70.0, b.glossy((0.8, 0.85, 0.88), 0.3)),
              b.constant_medium(b.sphere((400, 120, 150), 80.0, lam(WHITE)), 0.01, (0.9, 0.9, 0.9)),
              b.moving_sphere((100, 400, 200), (130, 400, 200), 0.0, 1.0, 50.0, lam((0.7, 0.3, 0.1))),
              b.bvh_node(small, 0.0, 1.0)]:
        b.add(g)
    if book:
        em = b.emissive((15.0, 15.0, 15.0))
        b.set_lights(b.collider_list([b.xzrect(*m, em) if m is not None else b.sphere((278, 300, 278), 40.0, em) for m in LIGHT_LIST]))
    else:
        b.set_lights(b.xzrect(*LIGHT, b.emissive((15.0, 15.0, 15.0))) if light else b.collider_list([]))
    b.set_camera(CAM["look_from"], CAM["look_at"], CAM["vup"], CAM["vfov"], CAM["aperture"], CAM["focus_dist"], CAM["aspect"], 0.0, 1.0)
    b.set_background(BACKGROUND)
    flat = b.flatten()
    if book:
        from rust_raytracer_amd import abi
        flat.semantics = abi.SEMANTICS_BOOK
    return b, flat


NOLIGHT = dict(W=16, H=12, spp=4)  # the second, smaller render: same scene, empty `lights`


def render_py(light=True, book=False):
    """→ float32 [spp, H, W, 3]: the radiance of every sample, laid out like the per-slice sums of a render with one
    sample per slice.  book=True: under RR_SEMANTICS_BOOK (intended traversal + book estimator) with LIGHT_LIST as `lights`."""
    import pyoracle as P
    objs, mats, cam = build_py()
    if book:
        P.INTENDED = P.BOOK_ESTIMATOR = True
        try:
            W, H, spp = RENDER["W"], RENDER["H"], RENDER["spp"]
            out = np.zeros((spp, H, W, 3), dtype=np.float32)
            for s in range(spp):
                for row in range(H):
                    for i in range(W):
                        out[s, row, i] = P.sample_pixel(objs, mats, list(LIGHT_LIST), BACKGROUND, cam, W, H, i, row, s, RENDER["max_depth"], RENDER["seed"])
            return out
        finally:
            P.INTENDED = P.BOOK_ESTIMATOR = False
    W, H, spp = (RENDER if light else NOLIGHT)["W"], (RENDER if light else NOLIGHT)["H"], (RENDER if light else NOLIGHT)["spp"]
    out = np.zeros((spp, H, W, 3), dtype=np.float32)
    for s in range(spp):
        for row in range(H):
            for i in range(W):
                out[s, row, i] = P.sample_pixel(objs, mats, LIGHT if light else None, BACKGROUND, cam, W, H, i, row, s,
                                                RENDER["max_depth"], RENDER["seed"])
    return out

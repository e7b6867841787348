"""pyoracle — a second, independent CPU restatement of the reference's intersection and path code.

TEST INFRASTRUCTURE ONLY (fixture generator and cross-check of oracle/rr_oracle.cpp).

Unlike rr_oracle.cpp, which walks the *flattened* arrays, this file keeps the scene as a
tree of Python objects shaped like the reference's `Geometry` enum (src/collider.rs:34-53)
and recurses through it exactly as `Geometry::intersect` does (collider.rs:682-697), in
numpy float32 scalar arithmetic (every operation rounds to f32, no fused multiply-add).
It also restates the host-side BVH builder (collider.rs:173-212), the Perlin table
generation (perlin.rs:14-46) and the host PCG32 generator, so that a tree built here from
the same seed must flatten to the same primitive ids as the C++ host library's, and
ray_color with the materials, textures and pdfs below it (render.rs:15-41, material.rs:85-181,
pdf.rs:5-70, texture.rs:66-103) over the path random streams rr_oracle.cpp uses.

Pure-Python loops: use for small cases only (a few thousand rays).

ray_color here is the reference's walk: every path to its end, also after its weight has become exactly zero
(render.rs:29-34).  The library and rr_oracle.cpp end such paths by default (RR_FLAG_TRACE_DEAD_PATHS in include/rr_hip.h);
comparisons with the fixtures this file writes set that flag.

PARITY UNPINNED with respect to the Rust binary: the reference has no tests or golden
vectors and cannot be built here; this file and rr_oracle.cpp are two readings of the same
source that are checked against each other and against analytic answers.
"""
import math

import numpy as np

f32 = np.float32
F32_MAX = f32(3.40282347e+38)
PI = f32(math.pi)
TAU = f32(2.0 * math.pi)

np.seterr(all="ignore")

# False: the reference as it is at HEAD.  True: the labelled "intended" semantics (RR_SEMANTICS_INTENDED in include/rr_hip.h):
# a BVH node whose first child misses goes on to its other child (the dead src/bvh.rs:28-44), the slab test clips on all
# three axes, a triangle rejects hits outside [t_min, t_max].  Set by the fixture generator and the tests, nowhere else.
INTENDED = False
# True: the second labelled alternative (RR_SEMANTICS_BOOK_ESTIMATOR): ray_color follows the ray a non-Lambertian material
# scattered itself, weighted by the attenuation alone, and `lights` may be a list of XZRects.
BOOK_ESTIMATOR = False


def v3(x, y, z):
    return (f32(x), f32(y), f32(z))


def add(a, b):
    return (a[0] + b[0], a[1] + b[1], a[2] + b[2])


def sub(a, b):
    return (a[0] - b[0], a[1] - b[1], a[2] - b[2])


def scale(a, s):
    return (a[0] * s, a[1] * s, a[2] * s)


def div(a, s):
    return (a[0] / s, a[1] / s, a[2] / s)


def dot(a, b):
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def cross(a, b):
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def length(a):
    return np.sqrt(dot(a, a))


def normalize(a):
    return div(a, length(a))


def neg(a):
    return scale(a, f32(-1.0))


class Ray:
    def __init__(self, o, d, tm):
        self.o, self.d, self.tm = o, d, f32(tm)

    def at(self, t):
        return add(self.o, scale(self.d, t))


class Hit:
    def __init__(self, p, t):
        self.p, self.t = p, t
        self.n = v3(0, 0, 0)
        self.u = self.v = f32(0)
        self.front = False
        self.prim = -1

    def set_face_normal(self, r, out_n):  # material.rs:30-33
        self.front = bool(dot(r.d, out_n) < f32(0))
        self.n = out_n if self.front else neg(out_n)


# ----------------------------------------------------------------- host rng
class PCG32:
    """Mirror of the host library's generator (rust-raytracer_amd/host/scene.hpp, class Rng)."""
    M = (1 << 64) - 1

    def __init__(self, seed=1, stream=0x5851F42D4C957F2D):
        self.state = 0
        self.inc = ((stream << 1) | 1) & self.M
        self.next_u32()
        self.state = (self.state + seed) & self.M
        self.next_u32()

    def next_u32(self):
        old = self.state
        self.state = (old * 6364136223846793005 + self.inc) & self.M
        xs = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xs >> rot) | (xs << ((32 - rot) & 31))) & 0xFFFFFFFF

    def random(self):
        return f32(self.next_u32() >> 9) * f32(1.0 / 8388608.0)

    def randrange(self, lo, hi):
        lo, hi = f32(lo), f32(hi)
        while True:
            v = self.random() * (hi - lo) + lo
            if v < hi:
                return v

    def randuint(self, lo, hi):
        return lo + ((self.next_u32() * (hi - lo)) >> 32)


def philox4x32_10(key, ctr):
    """Philox4x32-10 (Salmon et al., SC'11): 10 rounds, multipliers 0xD2511F53 / 0xCD9E8D57, Weyl key bumps."""
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(10):
        p0 = 0xD2511F53 * c0
        p1 = 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c3 ^ k1) & 0xFFFFFFFF, p0 & 0xFFFFFFFF
        k0 = (k0 + 0x9E3779B9) & 0xFFFFFFFF
        k1 = (k1 + 0xBB67AE85) & 0xFFFFFFFF
    return c0, c1, c2, c3


class PathStream:
    """The path RNG stream (seed; pixel, sample, bounce): draw n = word n&3 of block n>>2, as f32 on the 2^-23 grid."""

    def __init__(self, seed, pixel, sample, bounce):
        self.key = (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
        self.ctr = (pixel, sample, bounce)
        self.n = 0
        self.buf = None

    def __call__(self):
        if self.n & 3 == 0:
            self.buf = philox4x32_10(self.key, self.ctr + (self.n >> 2,))
        w = self.buf[self.n & 3]
        self.n += 1
        return f32(w >> 9) * f32(1.0 / 8388608.0)


# ------------------------------------------------------------------- tree
class Geometry:
    """kind ∈ sphere, moving_sphere, xyrect, xzrect, yzrect, cuboid, medium, translate, rotate, bvh, list, triangle."""

    def __init__(self, kind, **kw):
        self.kind = kind
        self.__dict__.update(kw)
        self.prim = -1


def sphere(center, radius):
    return Geometry("sphere", center=v3(*center), radius=f32(radius))


def moving_sphere(c0, c1, t0, t1, radius):
    return Geometry("moving_sphere", c0=v3(*c0), c1=v3(*c1), t0=f32(t0), t1=f32(t1), radius=f32(radius))


def xyrect(x0, x1, y0, y1, k):
    return Geometry("xyrect", a0=f32(x0), a1=f32(x1), b0=f32(y0), b1=f32(y1), k=f32(k))


def xzrect(x0, x1, z0, z1, k):
    return Geometry("xzrect", a0=f32(x0), a1=f32(x1), b0=f32(z0), b1=f32(z1), k=f32(k))


def yzrect(y0, y1, z0, z1, k):
    return Geometry("yzrect", a0=f32(y0), a1=f32(y1), b0=f32(z0), b1=f32(z1), k=f32(k))


def cuboid(p0, p1):  # collider.rs:86-101
    p0, p1 = v3(*p0), v3(*p1)
    sides = [xyrect(p0[0], p1[0], p0[1], p1[1], p1[2]), xyrect(p0[0], p1[0], p0[1], p1[1], p0[2]),
             xzrect(p0[0], p1[0], p0[2], p1[2], p1[1]), xzrect(p0[0], p1[0], p0[2], p1[2], p0[1]),
             yzrect(p0[1], p1[1], p0[2], p1[2], p1[0]), yzrect(p0[1], p1[1], p0[2], p1[2], p0[0])]
    return Geometry("cuboid", p0=p0, p1=p1, sides=sides)


def constant_medium(boundary, density):
    return Geometry("medium", boundary=boundary, nid=f32(-1.0) / f32(density))


def translate(g, d):
    return Geometry("translate", child=g, d=v3(*d))


def rotate(g, axis, angle):  # collider.rs:114-172; axis 0=X 1=Y 2=Z
    radians = f32(angle) * PI / f32(180.0)
    # f32 sin/cos of the f32 angle, formed by rounding the double result (what a correctly rounded sinf gives)
    s, c = f32(math.sin(float(radians))), f32(math.cos(float(radians)))
    return Geometry("rotate", child=g, axis=axis, s=s, c=c)


def triangle(p0, p1, p2):
    p0, p1, p2 = v3(*p0), v3(*p1), v3(*p2)
    return Geometry("triangle", p0=p0, p1=p1, p2=p2, n=normalize(cross(sub(p1, p0), sub(p2, p0))))


def collider_list(items):
    return Geometry("list", items=list(items))


def bounding_box(g, t0=0.0, t1=0.0):
    """collider.rs:698-713 for the kinds a BVH can hold here → (min, max) or None."""
    e = f32(0.0001)
    k = g.kind
    if k == "sphere":
        r = (g.radius, g.radius, g.radius)
        return sub(g.center, r), add(g.center, r)
    if k == "moving_sphere":
        r = (g.radius, g.radius, g.radius)
        ca, cb = _msphere_center(g, f32(t0)), _msphere_center(g, f32(t1))
        return _surround((sub(ca, r), add(ca, r)), (sub(cb, r), add(cb, r)))
    if k == "xyrect":
        return (g.a0, g.b0, g.k - e), (g.a1, g.b1, g.k + e)
    if k == "xzrect":
        return (g.a0, g.k - e, g.b0), (g.a1, g.k + e, g.b1)
    if k == "yzrect":
        return (g.k - e, g.a0, g.b0), (g.k + e, g.a1, g.b1)
    if k == "cuboid":
        return g.p0, g.p1
    if k == "triangle":
        hi = tuple(max(g.p0[i], max(g.p1[i], g.p2[i])) for i in range(3))
        lo = tuple(min(g.p0[i], min(g.p1[i], g.p2[i])) for i in range(3))
        return lo, hi
    if k == "bvh":
        return g.box
    if k == "medium":
        return bounding_box(g.boundary, t0, t1)
    if k == "translate":
        b = bounding_box(g.child, t0, t1)
        return None if b is None else (add(b[0], g.d), add(b[1], g.d))
    if k == "list":
        return None
    raise NotImplementedError(k)


def _surround(a, b):
    return (tuple(min(a[0][i], b[0][i]) for i in range(3)), tuple(max(a[1][i], b[1][i]) for i in range(3)))


def bvh_node(items, t0, t1, rng, start=0, end=None):
    """collider.rs:173-212 with the host library's draw order (axis first, then left, then right)."""
    items = list(items)
    if end is None:
        end = len(items)

    def build(start, end):
        axis = rng.randuint(0, 2)
        cmp_axis = 0 if axis == 0 else (1 if axis == 2 else 2)
        span = end - start

        def key(g):
            return bounding_box(g, 0.0, 0.0)[0][cmp_axis]

        if span == 1:
            left = right = items[start]
        elif span == 2:
            if not (key(items[start]) > key(items[start + 1])):
                left, right = items[start], items[start + 1]
            else:
                left, right = items[start + 1], items[start]
        else:
            items[start:end] = sorted(items[start:end], key=key)  # Python's sort is stable, like sort_by
            mid = start + span // 2
            left = build(start, mid)
            right = build(mid, end)
        bl, br = bounding_box(left, t0, t1), bounding_box(right, t0, t1)
        if bl is not None and br is not None:
            box = _surround(bl, br)
        else:
            box = bl if bl is not None else br
        return Geometry("bvh", left=left, right=right, box=box, axis=axis)

    return build(start, end)


def number_primitives(roots):
    """Primitive ids in depth-first order from the first top-level object; a BVH leaf that names
    one node on both sides numbers it once.  Media take an id before their boundary."""
    counter = [0]

    def visit(g):
        k = g.kind
        if k in ("sphere", "moving_sphere", "xyrect", "xzrect", "yzrect", "triangle"):
            g.prim = counter[0]
            counter[0] += 1
        elif k == "cuboid":
            for s in g.sides:
                visit(s)
        elif k == "medium":
            g.prim = counter[0]
            counter[0] += 1
            visit(g.boundary)
        elif k in ("translate", "rotate"):
            visit(g.child)
        elif k == "bvh":
            visit(g.left)
            if g.right is not g.left:
                visit(g.right)
        elif k == "list":
            for it in g.items:
                visit(it)

    for r in roots:
        visit(r)
    return counter[0]


# -------------------------------------------------------------- intersect
def _msphere_center(g, time):
    return add(g.c0, scale(sub(g.c1, g.c0), (time - g.t0) / (g.t1 - g.t0)))


def _hit_sphere(center, radius, r, t_min, t_max, with_uv):  # collider.rs:278-301 / 309-331
    oc = sub(r.o, center)
    a = dot(r.d, r.d)
    half_b = dot(oc, r.d)
    c = dot(oc, oc) - radius * radius
    disc = half_b * half_b - a * c
    if disc < f32(0):
        return None
    sq = np.sqrt(disc)
    root = (-half_b - sq) / a
    if root < t_min or t_max < root:
        root = (-half_b + sq) / a
        if root < t_min or t_max < root:
            return None
    h = Hit(r.at(root), root)
    out_n = div(sub(h.p, center), radius)
    h.set_face_normal(r, out_n)
    if with_uv:  # collider.rs:272-276
        theta = np.arccos(-out_n[1])
        phi = np.arctan2(-out_n[2], out_n[0]) + PI
        h.u, h.v = phi / TAU, theta / PI
    return h


_RECT_AXES = {"xyrect": (0, 1, 2), "xzrect": (0, 2, 1), "yzrect": (1, 2, 0)}
_RECT_NORMAL = {"xyrect": v3(0, 0, 1), "xzrect": v3(0, 1, 0), "yzrect": v3(1, 0, 0)}


def _hit_rect(g, r, t_min, t_max):  # collider.rs:344-359, 367-382, 405-420
    ia, ib, ik = _RECT_AXES[g.kind]
    t = (g.k - r.o[ik]) / r.d[ik]
    if t < t_min or t > t_max:
        return None
    a = r.o[ia] + t * r.d[ia]
    b = r.o[ib] + t * r.d[ib]
    if a < g.a0 or a > g.a1 or b < g.b0 or b > g.b1:
        return None
    h = Hit(r.at(t), t)
    h.u, h.v = (a - g.a0) / (g.a1 - g.a0), (b - g.b0) / (g.b1 - g.b0)
    h.set_face_normal(r, _RECT_NORMAL[g.kind])
    return h


def _hit_triangle(g, r, t_min, t_max):  # collider.rs:627-667 (the reference ignores the range)
    ndd = dot(g.n, r.d)
    if abs(ndd) < f32(0.000001):
        return None
    d = -dot(g.n, g.p0)
    dist = -(dot(g.n, r.o) + d) / ndd
    if dist < f32(0):
        return None
    if INTENDED and (dist < t_min or dist > t_max):
        return None
    hp = add(r.o, scale(r.d, dist))
    for pa, pb in ((g.p0, g.p1), (g.p1, g.p2), (g.p2, g.p0)):
        if dot(g.n, cross(sub(pb, pa), sub(hp, pa))) < f32(0):
            return None
    h = Hit(hp, dist)
    h.set_face_normal(r, g.n)
    return h


def aabb_hit(box, r, t_min, t_max):  # aabb.rs:37-52
    lo, hi = box
    for a in range(3):
        inv_d = f32(1.0) / r.d[a]
        t0 = (lo[a] - r.o[a]) * inv_d
        t1 = (hi[a] - r.o[a]) * inv_d
        if inv_d < f32(0):
            t0, t1 = t1, t0
            t_min = t0 if t0 > t_min else t_min
            t_max = t1 if t1 < t_max else t_max
        elif INTENDED:  # the clipping the reference only does under the swap
            t_min = t0 if t0 > t_min else t_min
            t_max = t1 if t1 < t_max else t_max
        if t_max <= t_min:
            return False
    return True


def _hit_list(items, r, t_min, t_max, rnd):  # collider.rs:751-772 / 586-607
    best, closest = None, t_max
    for g in items:
        h = intersect(g, r, t_min, closest, rnd)
        if h is not None:
            closest = h.t
            best = h
    return best


_OTHERS = {0: (1, 2), 1: (0, 2), 2: (0, 1)}  # utils.rs:27-33


def intersect(g, r, t_min, t_max, rnd=None):
    """Geometry::intersect (collider.rs:682-697).  rnd() supplies random() for constant media."""
    t_min, t_max = f32(t_min), f32(t_max)
    k = g.kind
    if k == "sphere":
        h = _hit_sphere(g.center, g.radius, r, t_min, t_max, True)
    elif k == "moving_sphere":
        h = _hit_sphere(_msphere_center(g, r.tm), g.radius, r, t_min, t_max, False)
    elif k in _RECT_AXES:
        h = _hit_rect(g, r, t_min, t_max)
    elif k == "triangle":
        h = _hit_triangle(g, r, t_min, t_max)
    elif k == "cuboid":
        return _hit_list(g.sides, r, t_min, t_max, rnd)
    elif k == "list":
        return _hit_list(g.items, r, t_min, t_max, rnd)
    elif k == "medium":  # collider.rs:435-475
        rec1 = intersect(g.boundary, r, -F32_MAX, F32_MAX, rnd)
        if rec1 is None:
            return None
        rec2 = intersect(g.boundary, r, rec1.t + f32(0.0001), F32_MAX, rnd)
        if rec2 is None:
            return None
        t1, t2 = rec1.t, rec2.t
        if t1 < t_min:
            t1 = t_min
        if t2 > t_max:
            t2 = t_max
        if t1 >= t2:
            return None
        if t1 < f32(0):
            t1 = f32(0)
        ray_length = length(r.d)
        dist_inside = (t2 - t1) * ray_length
        hit_dist = g.nid * np.log10(f32(rnd()))
        if hit_dist > dist_inside:
            return None
        t = t1 + hit_dist / ray_length
        h = Hit(r.at(t), t)
        h.n = v3(1, 0, 0)
        h.front = True
        h.prim = g.prim
        return h
    elif k == "translate":  # collider.rs:480-490
        moved = Ray(sub(r.o, g.d), r.d, r.tm)
        h = intersect(g.child, moved, t_min, t_max, rnd)
        if h is None:
            return None
        h.p = add(h.p, g.d)
        h.set_face_normal(moved, h.n)
        return h
    elif k == "rotate":  # collider.rs:498-528
        o0, o1 = _OTHERS[g.axis]
        org, dr = list(r.o), list(r.d)
        org[o0] = g.c * r.o[o0] - g.s * r.o[o1]
        org[o1] = g.s * r.o[o0] + g.c * r.o[o1]
        dr[o0] = g.c * r.d[o0] - g.s * r.d[o1]
        dr[o1] = g.s * r.d[o0] + g.c * r.d[o1]
        rot = Ray(tuple(org), tuple(dr), r.tm)
        h = intersect(g.child, rot, t_min, t_max, rnd)
        if h is None:
            return None
        p, n = list(h.p), list(h.n)
        p[o0] = g.c * h.p[o0] + g.s * h.p[o1]
        p[o1] = -g.s * h.p[o0] + g.c * h.p[o1]
        n[o0] = g.c * h.n[o0] + g.s * h.n[o1]
        n[o1] = -g.s * h.n[o0] + g.c * h.n[o1]
        h.p = tuple(p)
        h.set_face_normal(rot, tuple(n))
        return h
    elif k == "bvh":  # collider.rs:536-581
        if g.box is not None and not aabb_hit(g.box, r, t_min, t_max):
            return None
        if r.d[g.axis] >= f32(0):
            lh = intersect(g.left, r, t_min, t_max, rnd)
            if lh is not None:
                rh = intersect(g.right, r, t_min, lh.t, rnd)
                return rh if rh is not None else lh
            return intersect(g.right if INTENDED else g.left, r, t_min, t_max, rnd)  # `left` again, as in the reference
        rh = intersect(g.right, r, t_min, t_max, rnd)
        if rh is not None:
            lh = intersect(g.left, r, t_min, rh.t, rnd)
            return lh if lh is not None else rh
        return intersect(g.left, r, t_min, t_max, rnd)
    else:
        raise NotImplementedError(k)
    if h is not None:
        h.prim = g.prim
    return h


def scene_intersect(objects, r, t_min=0.001, t_max=F32_MAX, rnd=None):
    return _hit_list(objects, r, f32(t_min), f32(t_max), rnd)


# ------------------------------------------------------------------ camera
class Camera:  # camera.rs:34-75
    def __init__(self, look_from, look_at, vup, vfov, aperture, focus_dist, aspect, t_open=0.0, t_close=1.0):
        look_from, look_at, vup = v3(*look_from), v3(*look_at), v3(*vup)
        theta = f32(vfov) * PI / f32(180.0)
        h = f32(math.tan(float(theta / f32(2.0))))
        vh = f32(2.0) * h
        vw = f32(aspect) * vh
        self.w = normalize(sub(look_from, look_at))
        self.u = normalize(cross(vup, self.w))
        self.v = cross(self.w, self.u)
        self.origin = look_from
        self.horizontal = scale(scale(self.u, f32(focus_dist)), vw)
        self.vertical = scale(scale(self.v, f32(focus_dist)), vh)
        self.lens_radius = f32(aperture) / f32(2.0)
        self.lower_left = sub(sub(sub(self.origin, div(self.horizontal, f32(2.0))), div(self.vertical, f32(2.0))),
                              scale(self.w, f32(focus_dist)))
        self.t_open, self.t_close = f32(t_open), f32(t_close)

    def ray_through(self, u, v, lens=(0.0, 0.0), time=0.0):
        """get_ray (camera.rs:76-84) with the random lens sample and time supplied by the caller."""
        rd = scale(v3(lens[0], lens[1], 0), self.lens_radius)
        offset = add(scale(self.u, rd[0]), scale(self.v, rd[1]))
        o = add(self.origin, offset)
        d = sub(sub(add(add(self.lower_left, scale(self.horizontal, f32(u))), scale(self.vertical, f32(v))), self.origin), offset)
        return Ray(o, d, time)


# ------------------------------------------------------------------ perlin
def perlin_tables(rng):
    """Perlin::new (perlin.rs:14-46) with the host draw order → (ranvec[256,3], perm[3,256])."""
    vec = np.zeros((256, 3), dtype=np.float32)
    for i in range(256):
        x = rng.randrange(-1.0, 1.0)
        y = rng.randrange(-1.0, 1.0)
        z = rng.randrange(-1.0, 1.0)
        vec[i] = normalize((x, y, z))
    perms = np.zeros((3, 256), dtype=np.uint8)
    for a in range(3):
        ident = list(range(256))
        out = list(range(256))
        for i in range(255, 0, -1):  # perlin.rs:38-46: reads the ORIGINAL array, not a running shuffle
            target = rng.randuint(0, i)
            out[i] = ident[target]
            out[target] = ident[i]
        perms[a] = out
    return vec, perms


def _as_isize(x):  # Rust `as isize` (64-bit): NaN → 0, saturating
    if x != x:
        return 0
    if x >= 9.2233720368547758e18:
        return (1 << 63) - 1
    if x <= -9.2233720368547758e18:
        return -(1 << 63)
    return int(x)


def perlin_noise(vec, perms, p):  # perlin.rs:47-86
    p = v3(*p)
    fl = [np.floor(c) for c in p]
    u, v, w = p[0] - fl[0], p[1] - fl[1], p[2] - fl[2]
    i, j, k = _as_isize(fl[0]), _as_isize(fl[1]), _as_isize(fl[2])
    uu = u * u * (f32(3) - f32(2) * u)
    vv = v * v * (f32(3) - f32(2) * v)
    ww = w * w * (f32(3) - f32(2) * w)
    accum = f32(0)
    for di in range(2):
        for dj in range(2):
            for dk in range(2):
                idx = int(perms[0][(i + di) & 255]) ^ int(perms[1][(j + dj) & 255]) ^ int(perms[2][(k + dk) & 255])
                c = tuple(vec[idx])
                wv = (u - f32(di), v - f32(dj), w - f32(dk))
                accum = accum + ((f32(di) * uu + (f32(1) - f32(di)) * (f32(1) - uu)) *
                                 (f32(dj) * vv + (f32(1) - f32(dj)) * (f32(1) - vv)) *
                                 (f32(dk) * ww + (f32(1) - f32(dk)) * (f32(1) - ww)) * dot(c, wv))
    return accum


def perlin_turb(vec, perms, p, depth):  # perlin.rs:87-98
    accum, weight = f32(0), f32(1)
    tp = v3(*p)
    for _ in range(depth):
        accum = accum + weight * perlin_noise(vec, perms, tp)
        weight = weight * f32(0.5)
        tp = scale(tp, f32(2))
    return abs(accum)


def xzrect_pdf_value(x0, x1, z0, z1, k, origin, v):  # collider.rs:389-399
    g = xzrect(x0, x1, z0, z1, k)
    r = Ray(v3(*origin), v3(*v), 0.0)
    h = _hit_rect(g, r, f32(0.001), F32_MAX)
    if h is None:
        return f32(0)
    area = (g.a1 - g.a0) * (g.b1 - g.b0)
    dist_sqrd = h.t * h.t * dot(r.d, r.d)
    cosine = abs(dot(r.d, h.n) / length(r.d))
    return dist_sqrd / (cosine * area)


# ------------------------------------------------------------- whole paths
# Second restatement of ray_color and what it calls (render.rs:15-41, material.rs:85-181, pdf.rs:5-70,
# texture.rs:66-103, camera.rs:76-84), written from the Rust text and independent of rr_oracle.cpp.  It uses
# the same random-number protocol (one Philox stream per (pixel, sample, bounce); draws that cannot reach the
# result are not taken), so the two can be compared sample for sample.
#
# Materials are tuples attached to primitives and media as `g.material`:
#   ("lambertian", tex) ("glossy", rgb, fuzz) ("dielectric", ior) ("isotropic", tex) ("emissive", tex)
# Textures: ("solid", rgb) ("checker", odd, even) ("noise", (vec, perms), scale, depth) ("image", HxWx3 uint8)
CAMERA_BOUNCE = 0xFFFFFFFF


def _stream_randrange(rnd, lo, hi):  # utils.rs:45-47
    lo, hi = f32(lo), f32(hi)
    while True:
        v = rnd() * (hi - lo) + lo
        if v < hi:
            return v


def materials_by_prim(roots):
    """prim id → material, walking the tree as number_primitives does; cuboid sides carry the cuboid's material."""
    out = {}

    def visit(g, inherited=None):
        k = g.kind
        m = getattr(g, "material", None) or inherited
        if k in ("sphere", "moving_sphere", "xyrect", "xzrect", "yzrect", "triangle"):
            out[g.prim] = m
        elif k == "cuboid":
            for s in g.sides:
                visit(s, m)
        elif k == "medium":
            out[g.prim] = m
            visit(g.boundary)
        elif k in ("translate", "rotate"):
            visit(g.child)
        elif k == "bvh":
            visit(g.left)
            if g.right is not g.left:
                visit(g.right)
        elif k == "list":
            for it in g.items:
                visit(it)

    for r in roots:
        visit(r)
    return out


def _clamp(x, lo, hi):  # utils.rs:57-65
    if x < lo:
        return lo
    if x > hi:
        return hi
    return x


def _as_u32(x):  # Rust `as u32`: saturating, NaN → 0
    if not (x > 0):
        return 0
    if x >= 4294967296.0:
        return 0xFFFFFFFF
    return int(x)


def texture_color(tex, u, v, p):  # texture.rs:66-103
    kind = tex[0]
    if kind == "solid":
        return v3(*tex[1])
    if kind == "checker":
        sines = np.sin(f32(10) * p[0]) * np.sin(f32(10) * p[1]) * np.sin(f32(10) * p[2])
        return v3(*tex[1]) if sines < f32(0) else v3(*tex[2])
    if kind == "noise":
        (vec, perms), sc, depth = tex[1], f32(tex[2]), tex[3]
        return scale(scale(v3(1, 1, 1), f32(0.5)), f32(1) + np.sin(sc * p[2] + f32(10) * perlin_turb(vec, perms, p, depth)))
    img = tex[1]
    h, w = img.shape[0], img.shape[1]
    uc = _clamp(f32(u), f32(0), f32(1))
    vc = f32(1) - _clamp(f32(v), f32(0), f32(1))
    i, j = _as_u32(uc * f32(w)), _as_u32(vc * f32(h))
    i, j = min(i, w - 1), min(j, h - 1)
    cs = f32(1.0) / f32(255.0)
    px = img[j, i]
    return (f32(px[0]) * cs, f32(px[1]) * cs, f32(px[2]) * cs)


def _random_in_unit_sphere(rnd):  # vec3.rs:223-231
    while True:
        p = (_stream_randrange(rnd, -1, 1), _stream_randrange(rnd, -1, 1), _stream_randrange(rnd, -1, 1))
        if dot(p, p) >= f32(1):
            continue
        return p


def _random_cosine_direction(rnd):  # vec3.rs:257-266
    r1, r2 = rnd(), rnd()
    z = np.sqrt(f32(1) - r2)
    phi = TAU * r1
    return (np.cos(phi) * np.sqrt(r2), np.sin(phi) * np.sqrt(r2), z)


def _onb(n):  # vec3.rs:197-219
    w = normalize(n)
    a = v3(0, 1, 0) if abs(w[0]) > f32(0.9) else v3(1, 0, 0)
    v = normalize(cross(w, a))
    u = cross(w, v)
    return u, v, w


def _mul(a, b):
    return (a[0] * b[0], a[1] * b[1], a[2] * b[2])


def _reflect(v, n):  # vec3.rs:268-270: v - n * 2 * dot(v, n)
    return sub(v, scale(scale(n, f32(2)), dot(v, n)))


def _refract(uv, n, etai_over_etat):  # vec3.rs:272-277
    cos_theta = -min(dot(uv, n), f32(1))
    perp = scale(add(uv, scale(n, cos_theta)), f32(etai_over_etat))
    parallel = scale(n, -np.sqrt(abs(f32(1) - dot(perp, perp))))
    return add(perp, parallel)


def _reflectance(cosin, ref_idx):  # material.rs:116-120
    r0 = (f32(1) - ref_idx) / (f32(1) + ref_idx)
    r0 = r0 * r0
    x = f32(1) - cosin
    return r0 + (f32(1) - r0) * (x * x * x * x * x)


def _lights(light):
    """`light` as ray_color takes it -> list of members (None for a member that is no XZRect), or None for a single geometry."""
    if isinstance(light, list):
        return light
    return None


def _light_sample(light, origin, rnd):  # collider.rs:720-725, 400-403
    if light is None:
        return v3(1, 0, 0)
    x = _stream_randrange(rnd, light[0], light[1])
    z = _stream_randrange(rnd, light[2], light[3])
    return sub((x, f32(light[4]), z), origin)


def _light_pdf(light, origin, direction):  # collider.rs:714-719
    if light is None:
        return f32(0)
    return xzrect_pdf_value(light[0], light[1], light[2], light[3], light[4], origin, direction)


def ray_color(objects, materials, light, background, r, depth, bounce, key):
    """render.rs:15-41.  light = (x0, x1, z0, z1, k) of a bare XZRect passed as `lights`, or None for anything
    else (pdf 0, sample (1,0,0): collider.rs:714-725); under BOOK_ESTIMATOR also a list of such members.
    key = (seed, pixel, sample)."""
    if depth <= 0:
        return v3(0, 0, 0)
    rnd = PathStream(key[0], key[1], key[2], bounce)
    rec = scene_intersect(objects, r, f32(0.001), F32_MAX, rnd)
    if rec is None:
        return v3(*background)
    m = materials[rec.prim]
    kind = m[0]
    emitted = texture_color(m[1], rec.u, rec.v, rec.p) if kind == "emissive" else v3(0, 0, 0)  # material.rs:176-181
    # scatter, material.rs:85-169: under the reference's semantics the scattered ray is overwritten at render.rs:32, so of
    # each variant only the return value and the attenuation matter, and only the draws that can change those are taken;
    # under BOOK_ESTIMATOR the ray of every material but the Lambertian is the one ray_color follows
    sdir = None
    if kind == "lambertian":
        attenuation = texture_color(m[1], rec.u, rec.v, rec.p)
    elif kind == "glossy":
        ud = normalize(r.d)
        reflected = _reflect(ud, rec.n)
        sd = add(reflected, scale(_random_in_unit_sphere(rnd), f32(m[2])))
        attenuation = v3(*m[1])
        if not (dot(sd, rec.n) > f32(0)):
            return emitted
        sdir = sd
    elif kind == "dielectric":
        attenuation = v3(1, 1, 1)
        if BOOK_ESTIMATOR:  # material.rs:128-148
            ratio = f32(1) / f32(m[1]) if rec.front else f32(m[1])
            ud = normalize(r.d)
            cos_theta = min(dot(neg(ud), rec.n), f32(1))
            sin_theta = np.sqrt(f32(1) - cos_theta * cos_theta)
            if ratio * sin_theta > f32(1) or _reflectance(cos_theta, ratio) > rnd():
                sdir = _reflect(ud, rec.n)
            else:
                sdir = _refract(ud, rec.n, ratio)
    elif kind == "isotropic":
        if BOOK_ESTIMATOR:
            sdir = _random_in_unit_sphere(rnd)  # material.rs:151 (before the texture lookup, which draws nothing)
        attenuation = texture_color(m[1], rec.u, rec.v, rec.p)
    else:
        return emitted
    if BOOK_ESTIMATOR and kind != "lambertian":
        li = ray_color(objects, materials, light, background, Ray(rec.p, sdir, r.tm), depth - 1, bounce + 1, key)
        return add(emitted, _mul(attenuation, li))
    # render.rs:29-33: mixture of the light pdf and the cosine pdf about the hit normal (pdf.rs:33-53)
    u, v, w = _onb(rec.n)
    members = _lights(light) if BOOK_ESTIMATOR else None
    if rnd() < f32(0.5):
        if members is not None:
            k = min(int(rnd() * f32(len(members))), len(members) - 1)
            direction = _light_sample(members[k], rec.p, rnd)
        else:
            direction = _light_sample(light, rec.p, rnd)
    else:
        cd = _random_cosine_direction(rnd)
        direction = add(add(scale(u, cd[0]), scale(v, cd[1])), scale(w, cd[2]))
    cosin = dot(normalize(direction), w)
    cos_pdf = f32(0) if cosin <= f32(0) else cosin / PI
    if members is not None:
        weight = f32(1) / f32(len(members))
        light_pdf = f32(0)
        for mem in members:
            light_pdf = light_pdf + weight * _light_pdf(mem, rec.p, direction)
    else:
        light_pdf = _light_pdf(light, rec.p, direction)
    with np.errstate(all="ignore"):
        pdf_val = f32(0.5) * light_pdf + f32(0.5) * cos_pdf
        sp = f32(0)  # material.rs:170-175
        if kind == "lambertian":
            cs = dot(rec.n, normalize(direction))
            sp = f32(0) if cs < f32(0) else cs / PI
        li = ray_color(objects, materials, light, background, Ray(rec.p, direction, r.tm), depth - 1, bounce + 1, key)
        return add(emitted, div(_mul(scale(attenuation, sp), li), pdf_val))


def sample_pixel(objects, materials, light, background, cam, W, H, i, row, sample, max_depth, seed):
    """One sample of render.rs:95-100 for column i of image row `row` (0 = top) → its radiance."""
    j = H - 1 - row
    key = (seed, row * W + i, sample)
    rnd = PathStream(seed, key[1], sample, CAMERA_BOUNCE)
    u = (f32(i) + rnd()) / (f32(W) - f32(1))
    v = (f32(j) + rnd()) / (f32(H) - f32(1))
    lens = (f32(0), f32(0))
    if cam.lens_radius != f32(0):  # vec3.rs:247-255; with aperture 0 the sample is multiplied by zero: not drawn
        while True:
            lens = (_stream_randrange(rnd, -1, 1), _stream_randrange(rnd, -1, 1))
            if lens[0] * lens[0] + lens[1] * lens[1] + f32(0) >= f32(1):
                continue
            break
    tm = _stream_randrange(rnd, cam.t_open, cam.t_close)
    r = cam.ray_through(u, v, lens, tm)
    return ray_color(objects, materials, light, background, r, max_depth, 0, key)

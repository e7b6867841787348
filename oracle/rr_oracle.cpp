// rr_oracle.cpp — CPU restatement of the reference's path-tracing hot path.
//
// TEST INFRASTRUCTURE ONLY.  This file is the parity checker and the timed CPU
// baseline; nothing in the product path may call it.  It restates, in plain f32
// C++ and with real recursion, what these reference functions compute:
//   src/render.rs   ray_color (15-41), render_worker (89-124), render_multi (128-165)
//   src/camera.rs   get_ray (76-84)
//   src/pdf.rs      cosine / collidable / mixture value + generate (33-70)
//   src/collider.rs every intersect_* (278-667), SceneColliders::intersect (751-772),
//                   pdf_value_xzrect / random_xzrect (389-403)
//   src/aabb.rs     AABB::intersect (37-52)
//   src/material.rs HitRecord (8-38), scatter_* / scattering_pdf / emitted (85-181)
//   src/texture.rs  get_*_color (66-103)
//   src/perlin.rs   perlin_interp / noise / turb (47-98)
//   src/vec3.rs     Vec3 ops, ONB, sampling helpers, reflect / refract (8-277)
//   src/utils.rs    random / randrange distributions, clamp (41-65)
// It reads the same flattened scene (include/rr_hip.h) the device reads, walks it by
// direct recursion exactly as the reference walks its Box<Geometry> tree, and keeps
// the behaviours that look like bugs but are the live semantics (one-sided slab test,
// the BVH miss branch that re-tests `left`, triangles ignoring the t range, log10 free
// paths, zero scattering pdf for non-Lambertian materials, ...).
//
// PARITY UNPINNED: the reference has no tests, golden vectors or fixtures, draws from
// an unseedable thread_rng(), and cannot be compiled in this environment (no Rust
// toolchain).  Fidelity of this file to the Rust source rests on line-by-line review
// against the citations, on analytic known-answer tests, and on an independent
// tree-walking Python restatement (oracle/pyoracle.py).
//
// Random numbers: the reference's thread_rng() is replaced by Philox4x32-10 streams
// keyed by (seed; pixel, sample, bounce): draw n of a stream is word n&3 of block n>>2;
// a u32 becomes an f32 on the 2^-23 grid of [0,1) (rand 0.8.5's gen_range(0f32..1f32)).
// The camera draws of one sample use bounce 0xFFFFFFFF; ray_color call number b
// (b = 0 for the camera ray) uses bounce b, drawing in the reference's own order:
// traversal (constant media), then scatter, then mixture.generate.  One rule departs
// from "draw wherever the reference draws": a draw whose value cannot influence the
// result (the scattered direction that render.rs:32 overwrites, the lens sample of an
// aperture-0 camera) is not taken.  Streams are counter-based and draws independent, so
// this changes no distribution; it only keeps the streams short.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/rr_hip.h"

namespace {

// ------------------------------------------------------------------ vec3.rs
struct V3 {
    float x, y, z;
};
inline V3 mk(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator/(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
inline V3 neg(V3 a) { return a * -1.0f; }  // vec3.rs:80-85
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline float len2(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
inline float len(V3 a) { return std::sqrt(len2(a)); }
inline V3 unit(V3 a) { return a / len(a); }
inline float get(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
inline void set(V3& a, int i, float v) { (i == 0 ? a.x : (i == 1 ? a.y : a.z)) = v; }
inline V3 from(const float* p) { return mk(p[0], p[1], p[2]); }

const float PI = 3.14159265358979323846f;
const float TAU = 6.28318530717958647692f;
const float F32_MAX = 3.40282347e+38f;

struct Ray {  // ray.rs:6-27
    V3 o, d;
    float tm;
    V3 at(float t) const { return o + d * t; }
};

// ------------------------------------------------------------------ rng
struct Philox {
    uint32_t key[2];
    uint32_t ctr[3];  // pixel, sample, bounce
    uint32_t n = 0;   // draws taken from this stream
    uint32_t buf[4];

    static void block(const uint32_t key_in[2], const uint32_t ctr_in[4], uint32_t out[4]) {
        uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3];
        uint32_t k0 = key_in[0], k1 = key_in[1];
        for (int r = 0; r < 10; ++r) {
            uint64_t p0 = static_cast<uint64_t>(0xD2511F53u) * c0;
            uint64_t p1 = static_cast<uint64_t>(0xCD9E8D57u) * c2;
            uint32_t n0 = static_cast<uint32_t>(p1 >> 32) ^ c1 ^ k0;
            uint32_t n1 = static_cast<uint32_t>(p1);
            uint32_t n2 = static_cast<uint32_t>(p0 >> 32) ^ c3 ^ k1;
            uint32_t n3 = static_cast<uint32_t>(p0);
            c0 = n0;
            c1 = n1;
            c2 = n2;
            c3 = n3;
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        out[0] = c0;
        out[1] = c1;
        out[2] = c2;
        out[3] = c3;
    }
    void start(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce) {
        key[0] = static_cast<uint32_t>(seed);
        key[1] = static_cast<uint32_t>(seed >> 32);
        ctr[0] = pixel;
        ctr[1] = sample;
        ctr[2] = bounce;
        n = 0;
    }
    uint32_t next_u32() {
        if ((n & 3u) == 0) {
            uint32_t c[4] = {ctr[0], ctr[1], ctr[2], n >> 2};
            block(key, c, buf);
        }
        return buf[n++ & 3u];
    }
    float random() { return static_cast<float>(next_u32() >> 9) * (1.0f / 8388608.0f); }  // utils.rs:41-43
    float randrange(float lo, float hi) {                                                 // utils.rs:45-47
        for (;;) {
            float v = random() * (hi - lo) + lo;
            if (v < hi) return v;
        }
    }
};

const uint32_t CAMERA_BOUNCE = 0xFFFFFFFFu;

V3 random_in_unit_sphere(Philox& g) {  // vec3.rs:223-231
    for (;;) {
        float x = g.randrange(-1.0f, 1.0f);
        float y = g.randrange(-1.0f, 1.0f);
        float z = g.randrange(-1.0f, 1.0f);
        V3 p = mk(x, y, z);
        if (len2(p) >= 1.0f) continue;
        return p;
    }
}
V3 random_in_unit_disk(Philox& g) {  // vec3.rs:247-255
    for (;;) {
        float x = g.randrange(-1.0f, 1.0f);
        float y = g.randrange(-1.0f, 1.0f);
        V3 p = mk(x, y, 0.0f);
        if (len2(p) >= 1.0f) continue;
        return p;
    }
}
V3 random_cosine_direction(Philox& g) {  // vec3.rs:257-266
    float r1 = g.random();
    float r2 = g.random();
    float z = std::sqrt(1.0f - r2);
    float phi = TAU * r1;
    float x = std::cos(phi) * std::sqrt(r2);
    float y = std::sin(phi) * std::sqrt(r2);
    return mk(x, y, z);
}
V3 reflect(V3 v, V3 n) { return v - n * 2.0f * dot(v, n); }  // vec3.rs:268-270
V3 refract(V3 uv, V3 n, float etai_over_etat) {               // vec3.rs:272-277
    float cos_theta = -std::fmin(dot(uv, n), 1.0f);
    V3 perp = (uv + n * cos_theta) * etai_over_etat;
    V3 parallel = n * -(std::sqrt(std::fabs(1.0f - len2(perp))));
    return perp + parallel;
}
struct ONB {  // vec3.rs:197-219
    V3 u, v, w;
    static ONB from_w(V3 n) {
        ONB o;
        o.w = unit(n);
        V3 a = std::fabs(o.w.x) > 0.9f ? mk(0, 1, 0) : mk(1, 0, 0);
        o.v = unit(cross(o.w, a));
        o.u = cross(o.w, o.v);
        return o;
    }
    V3 local(V3 a) const { return u * a.x + v * a.y + w * a.z; }
};
float clampf(float x, float lo, float hi) {  // utils.rs:57-65
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}

// -------------------------------------------------------------- hit record
struct Hit {  // material.rs:8-38
    V3 p = {0, 0, 0}, n = {0, 0, 0};
    float t = 0, u = 0, v = 0;
    bool front = false;
    uint32_t mat = 0;
    int32_t prim = -1;
    void set_face_normal(const Ray& r, V3 out_normal) {
        front = dot(r.d, out_normal) < 0.0f;
        n = front ? out_normal : neg(out_normal);
    }
};

struct Ctx {
    const rr_flat_scene* s;
    Philox* rng;
    rr_counters* cnt;
};

bool hit_ref(Ctx& c, uint32_t ref, const Ray& r, float t_min, float t_max, Hit& out);

// ---------------------------------------------------------------- aabb.rs
bool aabb_hit(const float bmin[3], const float bmax[3], const Ray& r, float t_min, float t_max) {  // aabb.rs:37-52
    for (int a = 0; a < 3; ++a) {
        float inv_d = 1.0f / get(r.d, a);
        float t0 = (bmin[a] - get(r.o, a)) * inv_d;
        float t1 = (bmax[a] - get(r.o, a)) * inv_d;
        if (inv_d < 0.0f) {
            // after the swap the near plane is t1, the far plane t0
            t_min = t1 > t_min ? t1 : t_min;
            t_max = t0 < t_max ? t0 : t_max;
        }
        if (t_max <= t_min) return false;
    }
    return true;
}

// The labelled "intended" semantics (RR_SEMANTICS_INTENDED): the slab test on all three axes, as the test of aabb.rs:37-52
// reads once the `if inv_d < 0.0` around the clipping is the swap it was meant to guard.
bool aabb_hit_all_axes(const float bmin[3], const float bmax[3], const Ray& r, float t_min, float t_max) {
    for (int a = 0; a < 3; ++a) {
        float inv_d = 1.0f / get(r.d, a);
        float t0 = (bmin[a] - get(r.o, a)) * inv_d;
        float t1 = (bmax[a] - get(r.o, a)) * inv_d;
        if (inv_d < 0.0f) std::swap(t0, t1);
        t_min = t0 > t_min ? t0 : t_min;
        t_max = t1 < t_max ? t1 : t_max;
        if (t_max <= t_min) return false;
    }
    return true;
}

// ------------------------------------------------------------ collider.rs
bool hit_sphere_at(V3 center, float radius, const Ray& r, float t_min, float t_max, Hit& h, bool with_uv) {
    // collider.rs:278-301 (static) and 309-331 (moving: same roots, no uv)
    V3 oc = r.o - center;
    float a = len2(r.d);
    float half_b = dot(oc, r.d);
    float cc = len2(oc) - radius * radius;
    float disc = half_b * half_b - a * cc;
    if (disc < 0.0f) return false;
    float sqrtd = std::sqrt(disc);
    float root = (-half_b - sqrtd) / a;
    if (root < t_min || t_max < root) {
        root = (-half_b + sqrtd) / a;
        if (root < t_min || t_max < root) return false;
    }
    h = Hit();
    h.p = r.at(root);
    h.t = root;
    V3 out_normal = (h.p - center) / radius;
    h.set_face_normal(r, out_normal);
    if (with_uv) {  // collider.rs:272-276
        float theta = std::acos(-out_normal.y);
        float phi = std::atan2(-out_normal.z, out_normal.x) + PI;
        h.u = phi / TAU;
        h.v = theta / PI;
    }
    return true;
}

bool hit_rect(const rr_rect& q, const Ray& r, float t_min, float t_max, Hit& h) {
    // collider.rs:344-359 (xy), 367-382 (xz), 405-420 (yz)
    int ia, ib, ik;
    V3 nrm;
    if (q.axis == RR_RECT_XY) {
        ia = 0, ib = 1, ik = 2;
        nrm = mk(0, 0, 1);
    } else if (q.axis == RR_RECT_XZ) {
        ia = 0, ib = 2, ik = 1;
        nrm = mk(0, 1, 0);
    } else {
        ia = 1, ib = 2, ik = 0;
        nrm = mk(1, 0, 0);
    }
    float t = (q.k - get(r.o, ik)) / get(r.d, ik);
    if (t < t_min || t > t_max) return false;
    float a = get(r.o, ia) + t * get(r.d, ia);
    float b = get(r.o, ib) + t * get(r.d, ib);
    if (a < q.a0 || a > q.a1 || b < q.b0 || b > q.b1) return false;
    h = Hit();
    h.p = r.at(t);
    h.t = t;
    h.u = (a - q.a0) / (q.a1 - q.a0);
    h.v = (b - q.b0) / (q.b1 - q.b0);
    h.set_face_normal(r, nrm);
    h.mat = q.material;
    h.prim = static_cast<int32_t>(q.prim_id);
    return true;
}

bool hit_triangle(const rr_triangle& tr, const Ray& r, float t_min, float t_max, bool intended, Hit& h) {  // collider.rs:627-667 (t range ignored)
    V3 n = from(tr.normal), p0 = from(tr.p0), p1 = from(tr.p1), p2 = from(tr.p2);
    float n_dot_dir = dot(n, r.d);
    if (std::fabs(n_dot_dir) < 0.000001f) return false;
    float d = -dot(n, p0);
    float dist = -(dot(n, r.o) + d) / n_dot_dir;
    if (dist < 0.0f) return false;
    if (intended && (dist < t_min || dist > t_max)) return false;  // RR_SEMANTICS_INTENDED only: the range test the other intersectors make
    V3 hp = r.o + r.d * dist;
    if (dot(n, cross(p1 - p0, hp - p0)) < 0.0f) return false;
    if (dot(n, cross(p2 - p1, hp - p1)) < 0.0f) return false;
    if (dot(n, cross(p0 - p2, hp - p2)) < 0.0f) return false;
    h = Hit();
    h.p = hp;
    h.t = dist;
    h.set_face_normal(r, n);
    h.mat = tr.material;
    h.prim = static_cast<int32_t>(tr.prim_id);
    return true;
}

// SceneColliders::intersect (collider.rs:751-772) == intersect_collider_list (586-607)
bool hit_children(Ctx& c, const uint32_t* refs, uint32_t n, const Ray& r, float t_min, float t_max, Hit& out) {
    bool any = false;
    float closest = t_max;
    for (uint32_t i = 0; i < n; ++i) {
        Hit h;
        if (hit_ref(c, refs[i], r, t_min, closest, h)) {
            any = true;
            closest = h.t;
            out = h;
        }
    }
    return any;
}

bool hit_cuboid(Ctx& c, const rr_cuboid& cb, const Ray& r, float t_min, float t_max, Hit& out) {  // collider.rs:428-430
    bool any = false;
    float closest = t_max;
    for (uint32_t i = 0; i < 6; ++i) {
        Hit h;
        if (c.cnt) c.cnt->prim_tests++, c.cnt->rect_tests++;
        if (hit_rect(c.s->rects[cb.first_rect + i], r, t_min, closest, h)) {
            any = true;
            closest = h.t;
            out = h;
        }
    }
    return any;
}

bool hit_medium(Ctx& c, const rr_medium& m, const Ray& r, float t_min, float t_max, Hit& out) {  // collider.rs:435-475
    Hit rec1, rec2;
    if (!hit_ref(c, m.boundary, r, -F32_MAX, F32_MAX, rec1)) return false;
    if (!hit_ref(c, m.boundary, r, rec1.t + 0.0001f, F32_MAX, rec2)) return false;
    if (rec1.t < t_min) rec1.t = t_min;
    if (rec2.t > t_max) rec2.t = t_max;
    if (rec1.t >= rec2.t) return false;
    if (rec1.t < 0.0f) rec1.t = 0.0f;
    float ray_length = len(r.d);
    float dist_inside = (rec2.t - rec1.t) * ray_length;
    float hit_dist = m.neg_inv_density * std::log10(c.rng->random());
    if (hit_dist > dist_inside) return false;
    float t = rec1.t + hit_dist / ray_length;
    out = Hit();
    out.p = r.at(t);
    out.t = t;
    out.n = mk(1, 0, 0);
    out.front = true;
    out.mat = m.material;
    out.prim = static_cast<int32_t>(m.prim_id);
    return true;
}

bool hit_instance(Ctx& c, const rr_instance& in, const Ray& r, float t_min, float t_max, Hit& out) {
    if (in.kind == RR_INST_TRANSLATE) {  // collider.rs:480-490
        V3 disp = from(in.v);
        Ray moved{r.o - disp, r.d, r.tm};
        if (!hit_ref(c, in.child, moved, t_min, t_max, out)) return false;
        out.p = out.p + disp;
        out.set_face_normal(moved, out.n);
        return true;
    }
    // collider.rs:498-528; Axis::others() (utils.rs:27-33)
    int o0, o1;
    if (in.kind == RR_INST_ROTATE_X) {
        o0 = 1, o1 = 2;
    } else if (in.kind == RR_INST_ROTATE_Y) {
        o0 = 0, o1 = 2;
    } else {
        o0 = 0, o1 = 1;
    }
    const float s = in.v[0], co = in.v[1];
    V3 org = r.o, dir = r.d;
    set(org, o0, co * get(r.o, o0) - s * get(r.o, o1));
    set(org, o1, s * get(r.o, o0) + co * get(r.o, o1));
    set(dir, o0, co * get(r.d, o0) - s * get(r.d, o1));
    set(dir, o1, s * get(r.d, o0) + co * get(r.d, o1));
    Ray rot{org, dir, r.tm};
    if (!hit_ref(c, in.child, rot, t_min, t_max, out)) return false;
    V3 p = out.p, nrm = out.n;
    set(p, o0, co * get(out.p, o0) + s * get(out.p, o1));
    set(p, o1, -s * get(out.p, o0) + co * get(out.p, o1));
    set(nrm, o0, co * get(out.n, o0) + s * get(out.n, o1));
    set(nrm, o1, -s * get(out.n, o0) + co * get(out.n, o1));
    out.p = p;
    out.set_face_normal(rot, nrm);  // object-space ray against the world-space normal, as in the reference
    return true;
}

bool hit_bvh(Ctx& c, const rr_bvh_node& nd, const Ray& r, float t_min, float t_max, Hit& out) {  // collider.rs:536-581
    const bool intended = (c.s->semantics & RR_SEMANTICS_INTENDED) != 0u;
    if (nd.has_box) {
        if (c.cnt) c.cnt->box_tests++;
        if (!(intended ? aabb_hit_all_axes(nd.bmin, nd.bmax, r, t_min, t_max) : aabb_hit(nd.bmin, nd.bmax, r, t_min, t_max))) return false;
    }
    if (get(r.d, static_cast<int>(nd.axis)) >= 0.0f) {
        Hit left_hit;
        if (hit_ref(c, nd.left, r, t_min, t_max, left_hit)) {
            Hit right_hit;
            if (hit_ref(c, nd.right, r, t_min, left_hit.t, right_hit)) {
                out = right_hit;
            } else {
                out = left_hit;
            }
            return true;
        }
        // the reference re-tests `left` here, never `right`; the intended semantics go on to `right` (src/bvh.rs:28-34)
        return hit_ref(c, intended ? nd.right : nd.left, r, t_min, t_max, out);
    } else {
        Hit right_hit;
        if (hit_ref(c, nd.right, r, t_min, t_max, right_hit)) {
            Hit left_hit;
            if (hit_ref(c, nd.left, r, t_min, right_hit.t, left_hit)) {
                out = left_hit;
            } else {
                out = right_hit;
            }
            return true;
        }
        return hit_ref(c, nd.left, r, t_min, t_max, out);
    }
}

bool hit_ref(Ctx& c, uint32_t ref, const Ray& r, float t_min, float t_max, Hit& out) {  // collider.rs:682-697
    const rr_flat_scene& s = *c.s;
    const uint32_t idx = RR_REF_INDEX(ref);
    switch (RR_REF_TAG(ref)) {
        case RR_TAG_SPHERE: {
            const rr_sphere& sp = s.spheres[idx];
            if (c.cnt) c.cnt->prim_tests++, c.cnt->sphere_tests++;
            if (!hit_sphere_at(from(sp.center), sp.radius, r, t_min, t_max, out, true)) return false;
            out.mat = sp.material;
            out.prim = static_cast<int32_t>(sp.prim_id);
            return true;
        }
        case RR_TAG_MOVING_SPHERE: {
            const rr_moving_sphere& ms = s.moving_spheres[idx];
            if (c.cnt) c.cnt->prim_tests++, c.cnt->sphere_tests++;
            V3 c0 = from(ms.center0), c1 = from(ms.center1);
            V3 center = c0 + (c1 - c0) * ((r.tm - ms.time0) / (ms.time1 - ms.time0));  // collider.rs:269-271
            if (!hit_sphere_at(center, ms.radius, r, t_min, t_max, out, false)) return false;
            out.mat = ms.material;
            out.prim = static_cast<int32_t>(ms.prim_id);
            return true;
        }
        case RR_TAG_RECT:
            if (c.cnt) c.cnt->prim_tests++, c.cnt->rect_tests++;
            return hit_rect(s.rects[idx], r, t_min, t_max, out);
        case RR_TAG_TRIANGLE:
            if (c.cnt) c.cnt->prim_tests++, c.cnt->tri_tests++;
            return hit_triangle(s.triangles[idx], r, t_min, t_max, (s.semantics & RR_SEMANTICS_INTENDED) != 0u, out);
        case RR_TAG_CUBOID:
            return hit_cuboid(c, s.cuboids[idx], r, t_min, t_max, out);
        case RR_TAG_MEDIUM:
            if (c.cnt) c.cnt->inst_visits++;
            return hit_medium(c, s.media[idx], r, t_min, t_max, out);
        case RR_TAG_INSTANCE:
            if (c.cnt) c.cnt->inst_visits++;
            return hit_instance(c, s.instances[idx], r, t_min, t_max, out);
        case RR_TAG_BVH:
            return hit_bvh(c, s.bvh_nodes[idx], r, t_min, t_max, out);
        case RR_TAG_LIST: {
            const rr_list& l = s.lists[idx];
            return hit_children(c, s.child_refs + l.first_child, l.n_children, r, t_min, t_max, out);
        }
        default:
            return false;
    }
}

// ---------------------------------------------------- perlin.rs / texture.rs
float perlin_noise(const rr_perlin& pn, V3 p) {  // perlin.rs:47-86
    float fx = std::floor(p.x), fy = std::floor(p.y), fz = std::floor(p.z);
    float u = p.x - fx, v = p.y - fy, w = p.z - fz;
    // `as isize` saturates; & 255 then keeps the low bits of the (possibly saturated) value
    auto to_i = [](float f) -> int64_t {
        if (f != f) return 0;
        if (f >= 9.2233720368547758e18f) return INT64_MAX;
        if (f <= -9.2233720368547758e18f) return INT64_MIN;
        return static_cast<int64_t>(f);
    };
    int64_t i = to_i(fx), j = to_i(fy), k = to_i(fz);
    float uu = u * u * (3.0f - 2.0f * u);
    float vv = v * v * (3.0f - 2.0f * v);
    float ww = w * w * (3.0f - 2.0f * w);
    float accum = 0.0f;
    for (int di = 0; di < 2; ++di)
        for (int dj = 0; dj < 2; ++dj)
            for (int dk = 0; dk < 2; ++dk) {
                uint32_t idx = pn.perm_x[(static_cast<uint64_t>(i) + di) & 255u] ^ pn.perm_y[(static_cast<uint64_t>(j) + dj) & 255u] ^
                               pn.perm_z[(static_cast<uint64_t>(k) + dk) & 255u];
                V3 cvec = from(pn.ranvec[idx]);
                V3 weight = mk(u - float(di), v - float(dj), w - float(dk));
                accum += (float(di) * uu + (1.0f - float(di)) * (1.0f - uu)) * (float(dj) * vv + (1.0f - float(dj)) * (1.0f - vv)) *
                         (float(dk) * ww + (1.0f - float(dk)) * (1.0f - ww)) * dot(cvec, weight);
            }
    return accum;
}
float perlin_turb(const rr_perlin& pn, V3 p, uint32_t depth) {  // perlin.rs:87-98
    float accum = 0.0f, weight = 1.0f;
    V3 tp = p;
    for (uint32_t i = 0; i < depth; ++i) {
        accum += weight * perlin_noise(pn, tp);
        weight *= 0.5f;
        tp = tp * 2.0f;
    }
    return std::fabs(accum);
}

V3 texture_color(const rr_flat_scene& s, uint32_t tex, float u, float v, V3 p) {  // texture.rs:66-103
    const rr_texture& t = s.textures[tex];
    switch (t.type) {
        case RR_TEX_SOLID:
            return from(t.color0);
        case RR_TEX_CHECKERED: {
            float sines = std::sin(10.0f * p.x) * std::sin(10.0f * p.y) * std::sin(10.0f * p.z);
            return sines < 0.0f ? from(t.color0) : from(t.color1);
        }
        case RR_TEX_NOISE: {
            float tv = perlin_turb(s.perlin[t.index], p, t.depth);
            return mk(1.0f, 1.0f, 1.0f) * 0.5f * (1.0f + std::sin(t.scale * p.z + 10.0f * tv));
        }
        case RR_TEX_IMAGE: {
            const rr_image& im = s.images[t.index];
            float uc = clampf(u, 0.0f, 1.0f);
            float vc = 1.0f - clampf(v, 0.0f, 1.0f);
            // `as u32` saturates and maps NaN to 0
            auto to_u = [](float f) -> uint32_t {
                if (!(f > 0.0f)) return 0u;
                if (f >= 4294967296.0f) return 0xFFFFFFFFu;
                return static_cast<uint32_t>(f);
            };
            uint32_t i = to_u(uc * float(im.width));
            uint32_t j = to_u(vc * float(im.height));
            if (i >= im.width) i = im.width - 1;
            if (j >= im.height) j = im.height - 1;
            const float color_scale = 1.0f / 255.0f;
            const uint8_t* px = s.image_bytes + im.offset + (static_cast<uint64_t>(j) * im.width + i) * 3;
            return mk(float(px[0]) * color_scale, float(px[1]) * color_scale, float(px[2]) * color_scale);
        }
    }
    return mk(0, 0, 0);
}

// ------------------------------------------------------------ material.rs
float reflectance(float cosin, float ref_idx) {  // material.rs:116-120
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 *= r0;
    float x = 1.0f - cosin;
    float x5 = x * x * x * x * x;  // powi(5)
    return r0 + (1.0f - r0) * x5;
}

// scatter (material.rs:85-169).  Under the reference's semantics the scattered direction itself is dropped: ray_color
// overwrites it with mixture.generate() (render.rs:32); what survives is the return value, the attenuation, and the
// number of random draws consumed.  `want_dir` (RR_SEMANTICS_BOOK_ESTIMATOR): the direction is what ray_color follows
// for every material but the Lambertian, so it is formed, with the draws the reference's text takes for it.
bool scatter(const rr_flat_scene& s, const rr_material& m, const Ray& r_in, const Hit& rec, Philox& g, V3& attenuation, bool want_dir, V3& sdir) {
    switch (m.type) {
        case RR_MAT_LAMBERTIAN: {
            // material.rs:85-92 draws a cosine direction here that render.rs:32 then overwrites.  Draws
            // that can influence neither the return value nor the attenuation are not taken from the
            // counter-based stream (they are independent of every other draw, so the distribution of
            // the result is unchanged); the same holds for the dielectric coin and the isotropic direction.
            attenuation = texture_color(s, m.texture, rec.u, rec.v, rec.p);
            return true;
        }
        case RR_MAT_GLOSSY: {
            V3 reflected = reflect(unit(r_in.d), rec.n);
            V3 sd = reflected + random_in_unit_sphere(g) * m.param;
            attenuation = from(m.color);
            sdir = sd;
            return dot(sd, rec.n) > 0.0f;
        }
        case RR_MAT_DIELECTRIC: {
            // material.rs:128-148: reflect or refract by Schlick's approximation — into a direction that the reference's
            // ray_color discards; what remains is attenuation 1 and `true`
            attenuation = mk(1, 1, 1);
            if (want_dir) {
                const float refraction_ratio = rec.front ? 1.0f / m.param : m.param;
                const V3 unit_dir = unit(r_in.d);
                const float cos_theta = std::fmin(dot(neg(unit_dir), rec.n), 1.0f);
                const float sin_theta = std::sqrt(1.0f - cos_theta * cos_theta);
                const bool cannot_refract = refraction_ratio * sin_theta > 1.0f;
                // (`||` does not evaluate its right side, and so does not draw, when the left side holds)
                if (cannot_refract || reflectance(cos_theta, refraction_ratio) > g.random()) sdir = reflect(unit_dir, rec.n);
                else sdir = refract(unit_dir, rec.n, refraction_ratio);
            }
            return true;
        }
        case RR_MAT_ISOTROPIC: {  // material.rs:150-154
            if (want_dir) sdir = random_in_unit_sphere(g);
            attenuation = texture_color(s, m.texture, rec.u, rec.v, rec.p);
            return true;
        }
        default:  // emissive, material.rs:156-158
            return false;
    }
}

float light_pdf_value(const rr_light& l, V3 origin, V3 v) {  // collider.rs:714-719, 389-399
    if (!l.is_xzrect) return 0.0f;
    rr_rect q{l.x0, l.x1, l.z0, l.z1, l.k, RR_RECT_XZ, 0, 0};
    Ray r{origin, v, 0.0f};
    Hit rec;
    if (!hit_rect(q, r, 0.001f, F32_MAX, rec)) return 0.0f;
    float area = (l.x1 - l.x0) * (l.z1 - l.z0);
    float dist_sqrd = rec.t * rec.t * len2(v);
    float cosine = std::fabs(dot(v, rec.n) / len(v));
    return dist_sqrd / (cosine * area);
}
V3 light_random(const rr_light& l, V3 origin, Philox& g) {  // collider.rs:720-725, 400-403
    if (!l.is_xzrect) return mk(1.0f, 0.0f, 0.0f);
    float x = g.randrange(l.x0, l.x1);
    float z = g.randrange(l.z0, l.z1);
    return mk(x, l.k, z) - origin;
}
// RR_SEMANTICS_BOOK_ESTIMATOR with `lights` a ColliderList: the mean of the members' pdf values, and the sample of one
// member chosen uniformly (the reference's text has neither: collider.rs:714-725 gives a list pdf 0 and sample (1,0,0)).
bool light_list(const rr_flat_scene& s) { return (s.semantics & RR_SEMANTICS_BOOK_ESTIMATOR) != 0u && s.n_lights != 0u; }
float lights_pdf_value(const rr_flat_scene& s, V3 origin, V3 v) {
    if (!light_list(s)) return light_pdf_value(s.light, origin, v);
    const float weight = 1.0f / static_cast<float>(s.n_lights);
    float sum = 0.0f;
    for (uint32_t i = 0; i < s.n_lights; ++i) sum += weight * light_pdf_value(s.lights[i], origin, v);
    return sum;
}
V3 lights_random(const rr_flat_scene& s, V3 origin, Philox& g) {
    if (!light_list(s)) return light_random(s.light, origin, g);
    const uint32_t k = std::min(static_cast<uint32_t>(g.random() * static_cast<float>(s.n_lights)), s.n_lights - 1u);
    return light_random(s.lights[k], origin, g);
}

// ---------------------------------------------------------------- render.rs
struct PathKey {
    uint64_t seed;
    uint32_t pixel, sample;
    bool trace_dead;  // RR_FLAG_TRACE_DEAD_PATHS: the reference's walk; without it a call whose weight for what follows is exactly zero returns at once
};

// The library's default (include/rr_hip.h, RR_FLAG_TRACE_DEAD_PATHS): once the factor a ray_color call multiplies the rest of its
// path by is exactly zero in every channel, the rest is not walked and the call returns what it emitted.  The reference walks
// on (render.rs:29-34) and adds 0 * li: the same value unless li is NaN or infinite.  A factor that is NaN in every channel
// ends the walk too, exactly: NaN * li is NaN whatever li is, and that is what the call returns.
static bool dead(const V3& f) { return f.x == 0.0f && f.y == 0.0f && f.z == 0.0f; }
static bool all_nan(const V3& f) { return f.x != f.x && f.y != f.y && f.z != f.z; }

V3 ray_color(const rr_flat_scene& s, const Ray& r, uint32_t depth, uint32_t bounce, const PathKey& key, rr_counters* cnt) {
    if (depth == 0) return mk(0, 0, 0);  // render.rs:16-18
    Philox g;
    g.start(key.seed, key.pixel, key.sample, bounce);
    Ctx c{&s, &g, cnt};
    if (cnt) cnt->rays++;
    const rr_list& root = s.lists[RR_REF_INDEX(s.root)];
    Hit rec;
    if (!hit_children(c, s.child_refs + root.first_child, root.n_children, r, 0.001f, F32_MAX, rec)) return from(s.background);
    const rr_material& m = s.materials[rec.mat];
    V3 emitted = m.type == RR_MAT_EMISSIVE ? texture_color(s, m.texture, rec.u, rec.v, rec.p) : mk(0, 0, 0);  // material.rs:176-181
    V3 attenuation = mk(0, 0, 0), sdir = mk(0, 0, 0);
    const bool book = (s.semantics & RR_SEMANTICS_BOOK_ESTIMATOR) != 0u;
    if (!scatter(s, m, r, rec, g, attenuation, book, sdir)) return emitted;
    if (book && m.type != RR_MAT_LAMBERTIAN) {  // RR_SEMANTICS_BOOK_ESTIMATOR: a specular material's own ray, weighted by its attenuation alone
        Ray specular{rec.p, sdir, r.tm};
        if (!key.trace_dead && dead(attenuation)) return emitted;
        if (!key.trace_dead && all_nan(attenuation)) return attenuation;
        return emitted + attenuation * ray_color(s, specular, depth - 1, bounce + 1, key, cnt);
    }
    // render.rs:29-34
    ONB uvw = ONB::from_w(rec.n);
    V3 dir = g.random() < 0.5f ? lights_random(s, rec.p, g) : uvw.local(random_cosine_direction(g));  // pdf.rs:51-53
    Ray scattered{rec.p, dir, r.tm};
    float cosin = dot(unit(dir), uvw.w);  // pdf.rs:33-36
    float cos_pdf = cosin <= 0.0f ? 0.0f : cosin / PI;
    float pdf_val = 0.5f * lights_pdf_value(s, rec.p, dir) + 0.5f * cos_pdf;  // pdf.rs:48-50
    float sp = 0.0f;                                                              // material.rs:170-175
    if (m.type == RR_MAT_LAMBERTIAN) {                                            // material.rs:111-114
        float cs = dot(rec.n, unit(dir));
        sp = cs < 0.0f ? 0.0f : cs / PI;
    }
    if (!key.trace_dead && dead(attenuation * sp / pdf_val)) return emitted;
    if (!key.trace_dead && all_nan(attenuation * sp / pdf_val)) return attenuation * sp / pdf_val;  // (emitted + NaN)
    V3 li = ray_color(s, scattered, depth - 1, bounce + 1, key, cnt);
    return emitted + attenuation * sp * li / pdf_val;
}

Ray camera_ray(const rr_camera& cam, float u, float v, Philox& g) {  // camera.rs:76-84
    // with aperture 0 the lens sample is multiplied by zero (camera.rs:77): not drawn
    V3 rd = (cam.lens_radius != 0.0f ? random_in_unit_disk(g) : mk(0, 0, 0)) * cam.lens_radius;
    V3 offset = from(cam.u) * rd.x + from(cam.v) * rd.y;
    V3 origin = from(cam.origin);
    Ray r;
    r.o = origin + offset;
    r.d = from(cam.lower_left) + from(cam.horizontal) * u + from(cam.vertical) * v - origin - offset;
    r.tm = g.randrange(cam.shutter_open, cam.shutter_close);
    return r;
}

// One sample of pixel (i, j) with j counted up from the bottom row (render.rs:95-100).
V3 sample_pixel(const rr_flat_scene& s, const rr_render_params& p, uint32_t i, uint32_t j, uint32_t sample, rr_counters* cnt) {
    const uint32_t row = p.height - 1 - j;
    PathKey key{p.seed, row * p.width + i, sample, (p.flags & RR_FLAG_TRACE_DEAD_PATHS) != 0u};
    Philox g;
    g.start(key.seed, key.pixel, key.sample, CAMERA_BOUNCE);
    float u = (float(i) + g.random()) / (float(p.width) - 1.0f);
    float v = (float(j) + g.random()) / (float(p.height) - 1.0f);
    Ray r = camera_ray(s.camera, u, v, g);
    if (cnt) cnt->samples++;
    return ray_color(s, r, p.max_depth, 0, key, cnt);
}

uint8_t cast_u8(float f) {  // Rust `as u8`: saturating, NaN → 0 (render.rs:158)
    if (!(f > 0.0f)) return 0;
    if (f >= 255.0f) return 255;
    return static_cast<uint8_t>(f);
}

void add_counters(rr_counters& a, const rr_counters& b) {
    a.samples += b.samples;
    a.rays += b.rays;
    a.box_tests += b.box_tests;
    a.prim_tests += b.prim_tests;
    a.sphere_tests += b.sphere_tests;
    a.rect_tests += b.rect_tests;
    a.tri_tests += b.tri_tests;
    a.inst_visits += b.inst_visits;
}

}  // namespace

extern "C" {

// Philox4x32-10 block, exposed for known-answer tests.
void rro_philox4x32_10(const uint32_t key[2], const uint32_t ctr[4], uint32_t out[4]) { Philox::block(key, ctr, out); }

// The f32 draws a stream hands out, in order (tests pin the u32→f32 mapping with it).
void rro_stream_floats(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, uint32_t n, float* out) {
    Philox g;
    g.start(seed, pixel, sample, bounce);
    for (uint32_t i = 0; i < n; ++i) out[i] = g.random();
}

int rro_aabb_hit(const float bmin[3], const float bmax[3], const float ray[7], float t_min, float t_max) {
    Ray r{from(ray), from(ray + 3), ray[6]};
    return aabb_hit(bmin, bmax, r, t_min, t_max) ? 1 : 0;
}

float rro_perlin_noise(const rr_perlin* pn, const float p[3]) { return perlin_noise(*pn, from(p)); }
float rro_perlin_turb(const rr_perlin* pn, const float p[3], uint32_t depth) { return perlin_turb(*pn, from(p), depth); }

void rro_texture_color(const rr_flat_scene* s, uint32_t tex, float u, float v, const float p[3], float out[3]) {
    V3 c = texture_color(*s, tex, u, v, from(p));
    out[0] = c.x;
    out[1] = c.y;
    out[2] = c.z;
}

// vec3.rs:268-277 and material.rs:116-120.  The reference evaluates these inside scatter_glossy /
// scatter_dielectric; the dielectric's result is discarded by render.rs:32, so nothing on the
// rendering path reads refract or reflectance — they are kept restated and testable here.
void rro_reflect(const float v[3], const float n[3], float out[3]) {
    V3 r = reflect(from(v), from(n));
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
void rro_refract(const float uv[3], const float n[3], float etai_over_etat, float out[3]) {
    V3 r = refract(from(uv), from(n), etai_over_etat);
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
float rro_reflectance(float cosin, float ref_idx) { return reflectance(cosin, ref_idx); }

float rro_light_pdf_value(const rr_flat_scene* s, const float origin[3], const float v[3]) {
    return light_pdf_value(s->light, from(origin), from(v));
}

// Camera ray of (pixel i, row-from-bottom j, sample) exactly as the renderer draws it: out[7].
void rro_camera_ray(const rr_flat_scene* s, const rr_render_params* p, uint32_t i, uint32_t j, uint32_t sample, float out[7]) {
    const uint32_t row = p->height - 1 - j;
    Philox g;
    g.start(p->seed, row * p->width + i, sample, CAMERA_BOUNCE);
    float u = (float(i) + g.random()) / (float(p->width) - 1.0f);
    float v = (float(j) + g.random()) / (float(p->height) - 1.0f);
    Ray r = camera_ray(s->camera, u, v, g);
    out[0] = r.o.x, out[1] = r.o.y, out[2] = r.o.z, out[3] = r.d.x, out[4] = r.d.y, out[5] = r.d.z, out[6] = r.tm;
}

// Same contract as rr_primary_hits (include/rr_hip.h), computed on the CPU.  Rays are independent (a medium's draws are
// keyed by the ray's index), so they are shared out over the host's cores, up to 16.
int rro_primary_hits(const rr_flat_scene* s, uint32_t n, const float* rays, float t_min, float t_max, uint64_t seed,
                     rr_hit* out, rr_counters* counters) {
    if (!s || (!rays && n) || (!out && n)) return RR_E_INVALID;
    const rr_list& root = s->lists[RR_REF_INDEX(s->root)];
    const unsigned hw = std::thread::hardware_concurrency();
    const uint32_t n_threads = n < 256 ? 1u : std::max(1u, std::min(16u, hw ? hw : 1u));
    std::vector<rr_counters> tc(n_threads);
    std::atomic<uint32_t> next{0};
    auto worker = [&](uint32_t tid) {
        rr_counters local{};
        for (;;) {
            const uint32_t begin = next.fetch_add(64);
            if (begin >= n) break;
            for (uint32_t i = begin; i < std::min(n, begin + 64u); ++i) {
                const float* q = rays + static_cast<size_t>(i) * 7;
                Ray r{from(q), from(q + 3), q[6]};
                Philox g;
                g.start(seed, i, 0, 0);
                Ctx c{s, &g, counters ? &local : nullptr};
                Hit h;
                rr_hit& o = out[i];
                std::memset(&o, 0, sizeof(o));
                if (hit_children(c, s->child_refs + root.first_child, root.n_children, r, t_min, t_max, h)) {
                    o.prim_id = h.prim;
                    o.material = h.mat;
                    o.t = h.t;
                    o.point[0] = h.p.x, o.point[1] = h.p.y, o.point[2] = h.p.z;
                    o.normal[0] = h.n.x, o.normal[1] = h.n.y, o.normal[2] = h.n.z;
                    o.u = h.u, o.v = h.v;
                    o.front_face = h.front ? 1u : 0u;
                } else {
                    o.prim_id = -1;
                }
            }
        }
        tc[tid] = local;
    };
    std::vector<std::thread> pool;
    for (uint32_t t = 1; t < n_threads; ++t) pool.emplace_back(worker, t);
    worker(0);
    for (auto& t : pool) t.join();
    if (counters)
        for (auto& c : tc) add_counters(*counters, c);
    return RR_OK;
}

// Same outputs as rr_render_device, in host memory.  n_threads workers share the
// (slice, row) tasks; the result does not depend on n_threads.  With n_threads equal
// to n_slices this is render_multi's own decomposition: one full-image slice per thread.
int rro_render(const rr_flat_scene* s, const rr_render_params* p, float* slice_sums, float* resolved, float* linear,
               rr_counters* counters, int n_threads) {
    if (!s || !p || !slice_sums) return RR_E_INVALID;
    if (p->n_slices == 0 || p->slice_begin > p->slice_end || p->slice_end > p->n_slices || p->width == 0 || p->height == 0)
        return RR_E_INVALID;
    const uint32_t W = p->width, H = p->height;
    // the band of image rows [row_begin, row_end) (row_end = 0: the image height); outputs have the band's shape
    const uint32_t row_end = p->row_end == 0 ? H : p->row_end;
    if (p->row_begin > row_end || row_end > H) return RR_E_INVALID;
    const uint32_t R0 = p->row_begin, R = row_end - p->row_begin;
    const uint32_t spt = p->samples_per_pixel / p->n_slices;  // render.rs:130
    const uint32_t n_sl = p->slice_end - p->slice_begin;
    const size_t npix = static_cast<size_t>(W) * R;
    if (n_threads < 1) n_threads = 1;
    std::atomic<uint64_t> next{0};
    const uint64_t n_tasks = static_cast<uint64_t>(n_sl) * R;
    std::vector<rr_counters> tc(static_cast<size_t>(n_threads));
    auto worker = [&](int tid) {
        rr_counters* cnt = counters ? &tc[tid] : nullptr;
        for (;;) {
            uint64_t task = next.fetch_add(1);
            if (task >= n_tasks) break;
            const uint32_t sl = static_cast<uint32_t>(task / R);
            const uint32_t row = R0 + static_cast<uint32_t>(task % R);  // rows top to bottom, as render.rs:91
            const uint32_t j = H - 1 - row;
            const uint32_t k = p->slice_begin + sl;
            float* dst = slice_sums + (static_cast<size_t>(sl) * npix + static_cast<size_t>(row - R0) * W) * 3;
            for (uint32_t i = 0; i < W; ++i) {
                V3 pixel = mk(0, 0, 0);
                for (uint32_t q = 0; q < spt; ++q) {
                    uint32_t sample = p->sample_offset + k * spt + q;
                    pixel = pixel + sample_pixel(*s, *p, i, j, sample, cnt);
                }
                dst[i * 3 + 0] = pixel.x;
                dst[i * 3 + 1] = pixel.y;
                dst[i * 3 + 2] = pixel.z;
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; ++t) pool.emplace_back(worker, t);
    worker(0);
    for (auto& t : pool) t.join();
    if (counters)
        for (auto& c : tc) add_counters(*counters, c);

    // render.rs:102-119 per slice, then the sum of render.rs:141-150 in slice order
    if (resolved || linear) {
        const float scale = 1.0f / float(spt);
        const float nthr = float(p->n_slices);
        for (size_t e = 0; e < npix * 3; ++e) {
            float acc_res = 0.0f, acc_lin = 0.0f;
            for (uint32_t sl = 0; sl < n_sl; ++sl) {
                float c = slice_sums[static_cast<size_t>(sl) * npix * 3 + e];
                if (c != c) c = 0.0f;
                acc_lin += c;
                float g = std::sqrt(scale * c);
                acc_res += 256.0f * clampf(g / nthr, 0.0f, 0.999f);
            }
            if (resolved) resolved[e] = acc_res;
            if (linear) linear[e] = acc_lin;
        }
    }
    return RR_OK;
}

void rro_cast_u8(const float* resolved, uint8_t* out, uint64_t n) {
    for (uint64_t i = 0; i < n; ++i) out[i] = cast_u8(resolved[i]);
}

}  // extern "C"

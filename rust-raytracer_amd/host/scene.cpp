// Constructors, bounding boxes, the BVH builder and the .obj reader of the host
// scene API.  See scene.hpp for the mapping to the reference's files.
#include "scene.hpp"

#include <algorithm>
#include <cstring>
#include <fstream>
#include <numeric>
#include <sstream>

namespace rr {

// ---------------------------------------------------------------- perlin
// perlin.rs:14-46.  `permute` is NOT a Fisher-Yates shuffle: entry i and a random
// earlier entry are filled from the ORIGINAL identity array, so the output may
// repeat values; the tables are reproduced exactly as that loop leaves them.
static void perlin_generate_perm(Rng& rng, uint8_t* out) {
    uint8_t ident[RR_PERLIN_POINTS];
    for (int i = 0; i < RR_PERLIN_POINTS; ++i) ident[i] = out[i] = static_cast<uint8_t>(i);
    for (int i = RR_PERLIN_POINTS - 1; i >= 1; --i) {
        uint32_t target = rng.randuint(0, static_cast<uint32_t>(i));
        out[i] = ident[target];
        out[target] = ident[i];
    }
}

std::shared_ptr<const Perlin> Perlin::make(Rng& rng) {
    auto p = std::make_shared<Perlin>();
    for (int i = 0; i < RR_PERLIN_POINTS; ++i) {
        Vec3 v = normalize(rng.vec3_randrange(-1.0f, 1.0f));
        p->ranvec[i][0] = v.x;
        p->ranvec[i][1] = v.y;
        p->ranvec[i][2] = v.z;
    }
    perlin_generate_perm(rng, p->perm_x);
    perlin_generate_perm(rng, p->perm_y);
    perlin_generate_perm(rng, p->perm_z);
    return p;
}

// -------------------------------------------------------------- textures
Texture Texture::solid_color(Rgb color) {
    auto d = std::make_shared<TextureData>();
    d->type = RR_TEX_SOLID;
    d->color0 = color;
    return Texture(std::move(d));
}
Texture Texture::checkered(Rgb odd_color, Rgb even_color) {
    auto d = std::make_shared<TextureData>();
    d->type = RR_TEX_CHECKERED;
    d->color0 = odd_color;
    d->color1 = even_color;
    return Texture(std::move(d));
}
Texture Texture::noise(float scale, uint32_t turb, Rng& rng) {
    auto d = std::make_shared<TextureData>();
    d->type = RR_TEX_NOISE;
    d->perlin = Perlin::make(rng);
    d->scale = scale;
    d->depth = turb;
    return Texture(std::move(d));
}
Texture Texture::image(uint32_t atlas_index) {
    auto d = std::make_shared<TextureData>();
    d->type = RR_TEX_IMAGE;
    d->image = atlas_index;
    return Texture(std::move(d));
}

// ------------------------------------------------------------- materials
static Material make_material(uint32_t type, Texture tex, Rgb color, float param);

Material Material::lambertian(Texture albedo) { return make_material(RR_MAT_LAMBERTIAN, albedo, {}, 0.f); }
Material Material::glossy(Rgb albedo, float fuzz) {
    return make_material(RR_MAT_GLOSSY, Texture::solid_color(albedo), albedo, fuzz < 1.0f ? fuzz : 1.0f);
}
Material Material::dielectric(float ri) {
    return make_material(RR_MAT_DIELECTRIC, Texture::solid_color({1, 1, 1}), {1, 1, 1}, ri);
}
Material Material::isotropic(Rgb color) {
    return make_material(RR_MAT_ISOTROPIC, Texture::solid_color(color), color, 0.f);
}
Material Material::emissive(Rgb color) {
    return make_material(RR_MAT_EMISSIVE, Texture::solid_color(color), color, 0.f);
}

static Material make_material(uint32_t type, Texture tex, Rgb color, float param) {
    auto d = std::make_shared<MaterialData>();
    d->type = type;
    d->texture = tex;
    d->color = color;
    d->param = param;
    return Material(std::move(d));
}

// ------------------------------------------------------------------ aabb
AABB AABB::surrounding_box(const AABB& a, const AABB& b) {
    AABB r;
    r.minimum = {std::fmin(a.minimum.x, b.minimum.x), std::fmin(a.minimum.y, b.minimum.y),
                 std::fmin(a.minimum.z, b.minimum.z)};
    r.maximum = {std::fmax(a.maximum.x, b.maximum.x), std::fmax(a.maximum.y, b.maximum.y),
                 std::fmax(a.maximum.z, b.maximum.z)};
    return r;
}

// ---------------------------------------------------------- constructors
static std::shared_ptr<Geometry> new_geom(GeometryType t, Material m) {
    auto g = std::make_shared<Geometry>();
    g->type = t;
    g->material = std::move(m);
    return g;
}

GeometryPtr Geometry::sphere(Point3 center, float radius, Material m) {
    auto g = new_geom(GeometryType::Sphere, std::move(m));
    g->p[0] = center;
    g->f[0] = radius;
    return g;
}
GeometryPtr Geometry::moving_sphere(Point3 c0, Point3 c1, float time0, float time1, float radius, Material m) {
    auto g = new_geom(GeometryType::MovingSphere, std::move(m));
    g->p[0] = c0;
    g->p[1] = c1;
    g->f[0] = time0;
    g->f[1] = time1;
    g->f[2] = radius;
    return g;
}
static GeometryPtr make_rect(GeometryType t, float a0, float a1, float b0, float b1, float k, Material m) {
    auto g = new_geom(t, std::move(m));
    g->f[0] = a0;
    g->f[1] = a1;
    g->f[2] = b0;
    g->f[3] = b1;
    g->f[4] = k;
    return g;
}
GeometryPtr Geometry::xyrect(float x0, float x1, float y0, float y1, float k, Material m) {
    return make_rect(GeometryType::XYRect, x0, x1, y0, y1, k, std::move(m));
}
GeometryPtr Geometry::xzrect(float x0, float x1, float z0, float z1, float k, Material m) {
    return make_rect(GeometryType::XZRect, x0, x1, z0, z1, k, std::move(m));
}
GeometryPtr Geometry::yzrect(float y0, float y1, float z0, float z1, float k, Material m) {
    // a = y range, b = z range, k = x (collider.rs:80-85 stores the two pairs swapped and
    // un-swaps them again at both use sites, collider.rs:688,704).
    return make_rect(GeometryType::YZRect, y0, y1, z0, z1, k, std::move(m));
}
GeometryPtr Geometry::cuboid(Point3 p0, Point3 p1, Material m) {
    auto g = new_geom(GeometryType::Cuboid, m);
    g->p[0] = p0;
    g->p[1] = p1;
    g->children = {xyrect(p0.x, p1.x, p0.y, p1.y, p1.z, m), xyrect(p0.x, p1.x, p0.y, p1.y, p0.z, m),
                   xzrect(p0.x, p1.x, p0.z, p1.z, p1.y, m), xzrect(p0.x, p1.x, p0.z, p1.z, p0.y, m),
                   yzrect(p0.y, p1.y, p0.z, p1.z, p1.x, m), yzrect(p0.y, p1.y, p0.z, p1.z, p0.x, m)};
    return g;
}
GeometryPtr Geometry::constant_medium(GeometryPtr boundary, float density, Rgb color) {
    auto g = new_geom(GeometryType::ConstantMedium, Material::isotropic(color));
    g->children = {std::move(boundary)};
    g->f[0] = -1.0f / density;
    return g;
}
GeometryPtr Geometry::instance_translation(GeometryPtr child, Vec3 displacement) {
    auto g = new_geom(GeometryType::TranslateInstance, child->material);
    g->p[0] = displacement;
    g->children = {std::move(child)};
    return g;
}
GeometryPtr Geometry::instance_rotation(GeometryPtr child, Axis axis, float angle) {
    // collider.rs:114-172: world box = box of the 8 rotated corners of the child's box
    // over shutter [0,1]; a child without a box gives has_box = false.
    auto g = new_geom(GeometryType::RotationInstance, child->material);
    const float radians = to_radians(angle);
    const float s = std::sin(radians), c = std::cos(radians);
    auto cb = child->bounding_box(0.0f, 1.0f);
    AABB bb;
    bb.minimum = bb.maximum = Vec3(0, 0, 0);
    if (cb) bb = *cb;
    Vec3 lo(3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f);
    Vec3 hi(-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f);
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int k = 0; k < 2; ++k) {
                float x = float(i) * bb.maximum.x + float(1 - i) * bb.minimum.x;
                float y = float(j) * bb.maximum.y + float(1 - j) * bb.minimum.y;
                float z = float(k) * bb.maximum.z + float(1 - k) * bb.minimum.z;
                Vec3 t;
                if (axis == Axis::Y)
                    t = Vec3(c * x + s * z, y, -s * x + c * z);
                else if (axis == Axis::X)
                    t = Vec3(x, c * y + s * z, -s * y + c * z);
                else
                    t = Vec3(c * x + s * y, -s * x + c * y, z);
                for (int a = 0; a < 3; ++a) {
                    lo[a] = std::fmin(lo[a], t[a]);
                    hi[a] = std::fmax(hi[a], t[a]);
                }
            }
    g->axis = axis;
    g->f[0] = angle;
    g->f[1] = s;
    g->f[2] = c;
    g->has_box = cb.has_value();
    g->box.minimum = lo;
    g->box.maximum = hi;
    g->children = {std::move(child)};
    return g;
}
GeometryPtr Geometry::collider_list(std::vector<GeometryPtr> colliders) {
    auto g = new_geom(GeometryType::ColliderList, Material::lambertian(Texture::solid_color({0, 0, 0})));
    g->children = std::move(colliders);
    return g;
}
GeometryPtr Geometry::triangle(Point3 p0, Point3 p1, Point3 p2, Material m) {
    auto g = new_geom(GeometryType::Triangle, std::move(m));
    g->p[0] = p0;
    g->p[1] = p1;
    g->p[2] = p2;
    g->p[3] = normalize(cross(p1 - p0, p2 - p0));
    return g;
}

// --------------------------------------------------------- bounding boxes
static Point3 moving_center(const Geometry& g, float time) {  // collider.rs:269-271
    return g.p[0] + (g.p[1] - g.p[0]) * ((time - g.f[0]) / (g.f[1] - g.f[0]));
}

// collider.rs:608-625: returns None as soon as a child HAS a box (the test is
// inverted in the reference), so only an empty list or boxless children get through —
// and then out_box stays None too.  Net effect: a list never has a box.
static std::optional<AABB> list_bounding_box(const std::vector<GeometryPtr>& items, float t0, float t1) {
    if (items.empty()) return std::nullopt;
    for (const auto& it : items) {
        if (it->bounding_box(t0, t1).has_value()) return std::nullopt;
    }
    return std::nullopt;
}

std::optional<AABB> Geometry::bounding_box(float t0, float t1) const {
    const float e = 0.0001f;
    switch (type) {
        case GeometryType::Sphere: {
            Vec3 r(f[0], f[0], f[0]);
            return AABB{p[0] - r, p[0] + r};
        }
        case GeometryType::MovingSphere: {
            Vec3 r(f[2], f[2], f[2]);
            AABB b0{moving_center(*this, t0) - r, moving_center(*this, t0) + r};
            AABB b1{moving_center(*this, t1) - r, moving_center(*this, t1) + r};
            return AABB::surrounding_box(b0, b1);
        }
        case GeometryType::XYRect:
            return AABB{Point3(f[0], f[2], f[4] - e), Point3(f[1], f[3], f[4] + e)};
        case GeometryType::XZRect:
            return AABB{Point3(f[0], f[4] - e, f[2]), Point3(f[1], f[4] + e, f[3])};
        case GeometryType::YZRect:
            return AABB{Point3(f[4] - e, f[0], f[2]), Point3(f[4] + e, f[1], f[3])};
        case GeometryType::Cuboid:
            return AABB{p[0], p[1]};
        case GeometryType::ConstantMedium:
            return children[0]->bounding_box(t0, t1);
        case GeometryType::TranslateInstance: {
            auto b = children[0]->bounding_box(t0, t1);
            if (!b) return std::nullopt;
            return AABB{b->minimum + p[0], b->maximum + p[0]};
        }
        case GeometryType::RotationInstance:
            if (!has_box) return std::nullopt;
            return box;
        case GeometryType::BVHNode:
            if (!has_box) return std::nullopt;
            return box;
        case GeometryType::ColliderList:
            return list_bounding_box(children, t0, t1);
        case GeometryType::Triangle: {
            Point3 hi(std::fmax(p[0].x, std::fmax(p[1].x, p[2].x)), std::fmax(p[0].y, std::fmax(p[1].y, p[2].y)),
                      std::fmax(p[0].z, std::fmax(p[1].z, p[2].z)));
            Point3 lo(std::fmin(p[0].x, std::fmin(p[1].x, p[2].x)), std::fmin(p[0].y, std::fmin(p[1].y, p[2].y)),
                      std::fmin(p[0].z, std::fmin(p[1].z, p[2].z)));
            return AABB{lo, hi};
        }
    }
    return std::nullopt;
}

// ------------------------------------------------------------ BVH builder
// collider.rs:173-212.  Per node, in pre-order: draw axis = randuint(0,2) ∈ {0,1};
// the comparator orders by box minimum along x when axis == 0 and along z otherwise
// (the y comparator needs axis == 2, which is never drawn); the drawn axis value is
// what the node stores and what traversal later uses to index ray.direction.  Spans
// of 1 put the same node on both sides; spans of 2 order the pair with the comparator;
// longer spans stable-sort their sub-range (Rust's sort_by is stable) and split at
// start + span/2.  The comparator reads bounding_box(0,0) (collider.rs:10-11); the
// node box is the union of the children's bounding_box(time_0, time_1).
namespace {
struct BvhBuild {
    std::vector<GeometryPtr>& items;
    std::vector<std::array<float, 3>>& keys;  // bounding_box(0,0).minimum per item, permuted with items
    float t0, t1;
    Rng& rng;

    GeometryPtr node(size_t start, size_t end) {
        const uint32_t axis = rng.randuint(0, 2);
        const int cmp_axis = axis == 0 ? 0 : (axis == 2 ? 1 : 2);
        const size_t span = end - start;
        GeometryPtr left, right;
        if (span == 1) {
            left = right = items[start];
        } else if (span == 2) {
            if (!(keys[start][cmp_axis] > keys[start + 1][cmp_axis])) {  // Less or Equal
                left = items[start];
                right = items[start + 1];
            } else {
                left = items[start + 1];
                right = items[start];
            }
        } else {
            sort_range(start, end, cmp_axis);
            const size_t mid = start + span / 2;
            left = node(start, mid);
            right = node(mid, end);
        }
        auto g = std::make_shared<Geometry>();
        g->type = GeometryType::BVHNode;
        g->material = left->material;
        g->axis_index = axis;
        auto bl = left->bounding_box(t0, t1);
        auto br = right->bounding_box(t0, t1);
        if (bl && br) {
            g->has_box = true;
            g->box = AABB::surrounding_box(*bl, *br);
        } else if (bl) {
            g->has_box = true;
            g->box = *bl;
        } else if (br) {
            g->has_box = true;
            g->box = *br;
        } else {
            g->has_box = false;
        }
        g->children = {std::move(left), std::move(right)};
        return g;
    }

    void sort_range(size_t start, size_t end, int cmp_axis) {
        const size_t n = end - start;
        std::vector<uint32_t> order(n);
        std::iota(order.begin(), order.end(), 0u);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
            return keys[start + a][cmp_axis] < keys[start + b][cmp_axis];
        });
        std::vector<GeometryPtr> ti(n);
        std::vector<std::array<float, 3>> tk(n);
        for (size_t i = 0; i < n; ++i) {
            ti[i] = std::move(items[start + order[i]]);
            tk[i] = keys[start + order[i]];
        }
        for (size_t i = 0; i < n; ++i) {
            items[start + i] = std::move(ti[i]);
            keys[start + i] = tk[i];
        }
    }
};
}  // namespace

namespace {
GeometryPtr make_bvh_node(GeometryPtr left, GeometryPtr right, uint32_t axis, float t0, float t1) {
    auto g = std::make_shared<Geometry>();
    g->type = GeometryType::BVHNode;
    g->material = left->material;
    g->axis_index = axis;
    auto bl = left->bounding_box(t0, t1);
    auto br = right->bounding_box(t0, t1);
    g->has_box = true;
    g->box = AABB::surrounding_box(*bl, *br);
    g->children = {std::move(left), std::move(right)};
    return g;
}

struct SahBuild {
    struct Item {
        GeometryPtr g;
        AABB box;
        float c[3];
    };
    std::vector<Item> items;
    float t0, t1;
    static constexpr int BINS = 16;

    static float half_area(const float lo[3], const float hi[3]) {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
    GeometryPtr node(size_t start, size_t end) {
        const size_t span = end - start;
        if (span == 1) return make_bvh_node(items[start].g, items[start].g, 0, t0, t1);
        if (span == 2) {
            // the axis on which the two centres are farthest apart; the smaller coordinate on the left
            int axis = 0;
            for (int a = 1; a < 3; ++a)
                if (std::fabs(items[start].c[a] - items[start + 1].c[a]) > std::fabs(items[start].c[axis] - items[start + 1].c[axis])) axis = a;
            const bool swap = items[start].c[axis] > items[start + 1].c[axis];
            return make_bvh_node(items[start + (swap ? 1 : 0)].g, items[start + (swap ? 0 : 1)].g, static_cast<uint32_t>(axis), t0, t1);
        }
        float clo[3] = {3e38f, 3e38f, 3e38f}, chi[3] = {-3e38f, -3e38f, -3e38f};
        for (size_t i = start; i < end; ++i)
            for (int a = 0; a < 3; ++a) clo[a] = std::min(clo[a], items[i].c[a]), chi[a] = std::max(chi[a], items[i].c[a]);
        int best_axis = -1, best_bin = 0;
        float best_cost = 3e38f;
        for (int a = 0; a < 3; ++a) {
            const float extent = chi[a] - clo[a];
            if (!(extent > 0.0f)) continue;
            size_t count[BINS] = {};
            float lo[BINS][3], hi[BINS][3];
            for (int b = 0; b < BINS; ++b)
                for (int k = 0; k < 3; ++k) lo[b][k] = 3e38f, hi[b][k] = -3e38f;
            for (size_t i = start; i < end; ++i) {
                const int b = std::min(BINS - 1, static_cast<int>((items[i].c[a] - clo[a]) / extent * BINS));
                count[b]++;
                const float mn[3] = {items[i].box.minimum.x, items[i].box.minimum.y, items[i].box.minimum.z};
                const float mx[3] = {items[i].box.maximum.x, items[i].box.maximum.y, items[i].box.maximum.z};
                for (int k = 0; k < 3; ++k) lo[b][k] = std::min(lo[b][k], mn[k]), hi[b][k] = std::max(hi[b][k], mx[k]);
            }
            // cost of splitting after bin b: area(left) * n(left) + area(right) * n(right)
            float rarea[BINS];
            size_t rcount[BINS];
            float rl[3] = {3e38f, 3e38f, 3e38f}, rh[3] = {-3e38f, -3e38f, -3e38f};
            size_t rc = 0;
            for (int b = BINS - 1; b > 0; --b) {
                for (int k = 0; k < 3; ++k) rl[k] = std::min(rl[k], lo[b][k]), rh[k] = std::max(rh[k], hi[b][k]);
                rc += count[b];
                rarea[b] = rc ? half_area(rl, rh) : 0.0f;
                rcount[b] = rc;
            }
            float ll[3] = {3e38f, 3e38f, 3e38f}, lh[3] = {-3e38f, -3e38f, -3e38f};
            size_t lc = 0;
            for (int b = 0; b + 1 < BINS; ++b) {
                for (int k = 0; k < 3; ++k) ll[k] = std::min(ll[k], lo[b][k]), lh[k] = std::max(lh[k], hi[b][k]);
                lc += count[b];
                if (lc == 0 || rcount[b + 1] == 0) continue;
                const float cost = half_area(ll, lh) * static_cast<float>(lc) + rarea[b + 1] * static_cast<float>(rcount[b + 1]);
                if (cost < best_cost) best_cost = cost, best_axis = a, best_bin = b;
            }
        }
        size_t mid;
        if (best_axis < 0) {  // every centre in one point: halve the span as it stands
            best_axis = 0;
            mid = start + span / 2;
        } else {
            const float extent = chi[best_axis] - clo[best_axis];
            auto first_right = std::partition(items.begin() + static_cast<std::ptrdiff_t>(start), items.begin() + static_cast<std::ptrdiff_t>(end), [&](const Item& it) {
                return std::min(BINS - 1, static_cast<int>((it.c[best_axis] - clo[best_axis]) / extent * BINS)) <= best_bin;
            });
            mid = static_cast<size_t>(first_right - items.begin());
        }
        GeometryPtr left = node(start, mid), right = node(mid, end);
        return make_bvh_node(std::move(left), std::move(right), static_cast<uint32_t>(best_axis), t0, t1);
    }
};
}  // namespace

GeometryPtr Geometry::bvh_node_sah(const std::vector<GeometryPtr>& colliders, float time_0, float time_1) {
    if (colliders.empty()) throw std::invalid_argument("bvh_node_sah: no elements");
    SahBuild build{{}, time_0, time_1};
    build.items.reserve(colliders.size());
    for (const GeometryPtr& g : colliders) {
        auto b = g->bounding_box(time_0, time_1);
        if (!b) throw std::invalid_argument("bvh_node_sah: element without a bounding box");
        SahBuild::Item it{g, *b, {0.5f * (b->minimum.x + b->maximum.x), 0.5f * (b->minimum.y + b->maximum.y), 0.5f * (b->minimum.z + b->maximum.z)}};
        build.items.push_back(std::move(it));
    }
    return build.node(0, build.items.size());
}

GeometryPtr Geometry::bvh_node(const std::vector<GeometryPtr>& colliders, float time_0, float time_1, size_t start,
                               size_t end, Rng& rng) {
    if (start >= end || end > colliders.size()) throw std::invalid_argument("bvh_node: empty or out-of-range span");
    if (rng.sah_builder)  // the labelled option: not the reference's tree
        return bvh_node_sah(std::vector<GeometryPtr>(colliders.begin() + static_cast<std::ptrdiff_t>(start), colliders.begin() + static_cast<std::ptrdiff_t>(end)), time_0, time_1);
    std::vector<GeometryPtr> items(colliders);
    std::vector<std::array<float, 3>> keys(items.size());
    for (size_t i = start; i < end; ++i) {
        auto b = items[i]->bounding_box(0.0f, 0.0f);
        if (!b) throw std::invalid_argument("bvh_node: element without a bounding box (the reference panics here)");
        keys[i] = {b->minimum.x, b->minimum.y, b->minimum.z};
    }
    BvhBuild build{items, keys, time_0, time_1, rng};
    return build.node(start, end);
}

GeometryPtr Geometry::bvh_node_explicit(GeometryPtr left, GeometryPtr right, uint32_t axis, std::optional<AABB> box) {
    if (axis > 2) throw std::invalid_argument("bvh_node_explicit: axis must be 0..2");
    auto g = std::make_shared<Geometry>();
    g->type = GeometryType::BVHNode;
    g->material = left->material;
    g->axis_index = axis;
    g->has_box = box.has_value();
    if (box) g->box = *box;
    g->children = {std::move(left), std::move(right)};
    return g;
}

// ------------------------------------------------------------- .obj reader
// collider.rs:227-267: whitespace-split lines; a first token starting with "vt" is
// skipped, any other token starting with "v" (so "vn" too) appends a scaled vertex;
// a token starting with "f" reads exactly three 1-based indices, taking the part
// before the first '/' when the first index has slashes.  Everything becomes one
// BVH over the triangles with shutter [0,1].
//
// hardened = true is NOT the reference's reader: only "v" lines are vertices ("vn",
// "vp" and the rest are skipped), an index may be negative (relative to the vertices
// read so far), and a face of n > 3 corners becomes the fan (0, i, i+1).  Files the
// reference reads correctly (triangles, positive indices, no "vn" lines before the
// faces that use later vertices) give the same triangles in the same order either way.
GeometryPtr Geometry::load_obj(const std::string& path, float scale, Material m, Rng& rng, bool hardened) {
    std::ifstream in(path);
    if (!in) throw std::runtime_error("load_obj: cannot open " + path);
    std::vector<Vec3> vertices;
    std::vector<std::array<size_t, 3>> faces;
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream ss(line);
        std::string action;
        if (!(ss >> action)) continue;
        if (action.rfind("vt", 0) == 0) {
            continue;
        } else if (hardened ? action == "v" : action.rfind("v", 0) == 0) {
            float a, b, c;
            if (!(ss >> a >> b >> c)) throw std::runtime_error("load_obj: bad vertex line in " + path);
            vertices.push_back(Vec3(a * scale, b * scale, c * scale));
        } else if (hardened ? action == "f" : action.rfind("f", 0) == 0) {
            if (!hardened) {
                std::array<size_t, 3> idx{};
                for (int i = 0; i < 3; ++i) {
                    std::string tok;
                    if (!(ss >> tok)) throw std::runtime_error("load_obj: face with fewer than 3 indices in " + path);
                    idx[i] = static_cast<size_t>(std::stoull(tok.substr(0, tok.find('/'))));
                }
                faces.push_back(idx);
                continue;
            }
            std::vector<size_t> corners;
            std::string tok;
            while (ss >> tok) {
                long long i = 0;
                try {
                    i = std::stoll(tok.substr(0, tok.find('/')));
                } catch (const std::exception&) {
                    throw std::runtime_error("load_obj: bad face index '" + tok + "' in " + path);
                }
                if (i < 0) i += static_cast<long long>(vertices.size()) + 1;  // -1 = the vertex read last
                if (i < 1 || i > static_cast<long long>(vertices.size())) throw std::runtime_error("load_obj: face index out of range in " + path);
                corners.push_back(static_cast<size_t>(i));
            }
            if (corners.size() < 3) throw std::runtime_error("load_obj: face with fewer than 3 indices in " + path);
            for (size_t i = 1; i + 1 < corners.size(); ++i) faces.push_back({corners[0], corners[i], corners[i + 1]});
        }
    }
    std::vector<GeometryPtr> tris;
    tris.reserve(faces.size());
    for (const auto& fc : faces) {
        for (size_t i : fc)
            if (i < 1 || i > vertices.size()) throw std::runtime_error("load_obj: face index out of range in " + path);
        tris.push_back(Geometry::triangle(vertices[fc[0] - 1], vertices[fc[1] - 1], vertices[fc[2] - 1], m));
    }
    if (tris.empty()) throw std::runtime_error("load_obj: no faces in " + path);
    return Geometry::bvh_node(tris, 0.f, 1.f, 0, tris.size(), rng);
}

// ------------------------------------------------------------ image atlas
Texture SceneColliders::load_image_rgb8(uint32_t width, uint32_t height, const uint8_t* rgb) {
    ImageRGB8 im;
    im.width = width;
    im.height = height;
    im.rgb.assign(rgb, rgb + static_cast<size_t>(width) * height * 3);
    atlas.push_back(std::move(im));
    return Texture::image(static_cast<uint32_t>(atlas.size() - 1));
}

Texture SceneColliders::load_image(const std::string& ppm_path) {
    std::ifstream in(ppm_path, std::ios::binary);
    if (!in) throw std::runtime_error("load_image: cannot open " + ppm_path);
    std::string magic;
    in >> magic;
    if (magic != "P6") throw std::runtime_error("load_image: only binary PPM (P6) is read natively: " + ppm_path);
    auto next_int = [&]() -> uint32_t {
        for (;;) {
            int ch = in.peek();
            if (ch == '#') {
                std::string skip;
                std::getline(in, skip);
            } else if (std::isspace(ch)) {
                in.get();
            } else
                break;
        }
        uint32_t v = 0;
        in >> v;
        return v;
    };
    uint32_t w = next_int(), h = next_int(), maxv = next_int();
    in.get();
    if (w == 0 || h == 0 || maxv != 255) throw std::runtime_error("load_image: unsupported PPM header: " + ppm_path);
    std::vector<uint8_t> buf(static_cast<size_t>(w) * h * 3);
    in.read(reinterpret_cast<char*>(buf.data()), static_cast<std::streamsize>(buf.size()));
    if (static_cast<size_t>(in.gcount()) != buf.size()) throw std::runtime_error("load_image: truncated PPM: " + ppm_path);
    return load_image_rgb8(w, h, buf.data());
}

// ---------------------------------------------------------------- camera
Camera::Camera(Point3 look_from, Point3 look_at, Vec3 vup, float vfov, float aperture, float focus_dist,
               float aspect_ratio, float shutter_open_, float shutter_close_) {
    const float theta = to_radians(vfov);
    const float h = std::tan(theta / 2.0f);
    const float viewport_height = 2.0f * h;
    const float viewport_width = aspect_ratio * viewport_height;
    w = normalize(look_from - look_at);
    u = normalize(cross(vup, w));
    v = cross(w, u);
    origin = look_from;
    horizontal = u * focus_dist * viewport_width;
    vertical = v * focus_dist * viewport_height;
    lens_radius = aperture / 2.0f;
    lower_left = origin - horizontal / 2.0f - vertical / 2.0f - w * focus_dist;
    shutter_open = shutter_open_;
    shutter_close = shutter_close_;
}

uint32_t image_height_for(uint32_t width, float aspect_ratio) {
    return static_cast<uint32_t>(static_cast<float>(width) / aspect_ratio);
}

}  // namespace rr

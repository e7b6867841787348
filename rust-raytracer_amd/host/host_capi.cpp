// C bindings (include/rr_host.h) over the C++ scene API.
#include <cstring>
#include <string>

#include "../../include/rr_host.h"
#include "scene.hpp"

using namespace rr;

struct rrh_builder {
    Rng rng;
    SceneColliders scene;
    std::vector<Texture> textures;
    std::vector<Material> materials;
    std::vector<GeometryPtr> geoms;
    GeometryPtr lights;
    Camera cam;
    Rgb background;
    std::unique_ptr<FlatScene> flat;
    explicit rrh_builder(uint64_t seed) : rng(seed) { lights = Geometry::collider_list({}); }
};

namespace {
thread_local std::string g_err;
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
template <class F>
int guarded(F&& f) {
    try {
        return f();
    } catch (const std::exception& e) {
        return fail(RR_E_INVALID, e.what());
    } catch (...) {
        return fail(RR_E_INVALID, "unknown error");
    }
}
int push_tex(rrh_builder* b, Texture t) {
    b->textures.push_back(std::move(t));
    return static_cast<int>(b->textures.size() - 1);
}
int push_mat(rrh_builder* b, Material m) {
    b->materials.push_back(std::move(m));
    return static_cast<int>(b->materials.size() - 1);
}
int push_geom(rrh_builder* b, GeometryPtr g) {
    b->geoms.push_back(std::move(g));
    return static_cast<int>(b->geoms.size() - 1);
}
const Material& mat(rrh_builder* b, int h) {
    if (h < 0 || static_cast<size_t>(h) >= b->materials.size()) throw std::invalid_argument("bad material handle");
    return b->materials[h];
}
const Texture& tex(rrh_builder* b, int h) {
    if (h < 0 || static_cast<size_t>(h) >= b->textures.size()) throw std::invalid_argument("bad texture handle");
    return b->textures[h];
}
const GeometryPtr& geom(rrh_builder* b, int h) {
    if (h < 0 || static_cast<size_t>(h) >= b->geoms.size()) throw std::invalid_argument("bad geometry handle");
    return b->geoms[h];
}
Vec3 v3(const float* p) { return Vec3(p[0], p[1], p[2]); }
}  // namespace

extern "C" {

rrh_builder* rrh_builder_new(uint64_t seed) {
    try {
        return new rrh_builder(seed);
    } catch (...) {
        g_err = "rrh_builder_new: out of memory";
        return nullptr;
    }
}
void rrh_builder_free(rrh_builder* b) { delete b; }
const char* rrh_last_error(void) { return g_err.c_str(); }

float rrh_random(rrh_builder* b) { return b->rng.random(); }
float rrh_randrange(rrh_builder* b, float lo, float hi) { return b->rng.randrange(lo, hi); }
uint32_t rrh_randuint(rrh_builder* b, uint32_t lo, uint32_t hi) { return b->rng.randuint(lo, hi); }

int rrh_tex_solid_color(rrh_builder* b, float r, float g, float bl) {
    return guarded([&] { return push_tex(b, Texture::solid_color(Rgb(r, g, bl))); });
}
int rrh_tex_checkered(rrh_builder* b, const float odd[3], const float even[3]) {
    return guarded([&] { return push_tex(b, Texture::checkered(v3(odd), v3(even))); });
}
int rrh_tex_noise(rrh_builder* b, float scale, uint32_t turb) {
    return guarded([&] { return push_tex(b, Texture::noise(scale, turb, b->rng)); });
}
int rrh_load_image_rgb8(rrh_builder* b, uint32_t w, uint32_t h, const uint8_t* rgb) {
    return guarded([&] {
        if (!rgb || w == 0 || h == 0) throw std::invalid_argument("rrh_load_image_rgb8: empty image");
        return push_tex(b, b->scene.load_image_rgb8(w, h, rgb));
    });
}
int rrh_load_image_ppm(rrh_builder* b, const char* path) {
    return guarded([&] { return push_tex(b, b->scene.load_image(path)); });
}

int rrh_mat_lambertian(rrh_builder* b, int texture) {
    return guarded([&] { return push_mat(b, Material::lambertian(tex(b, texture))); });
}
int rrh_mat_glossy(rrh_builder* b, float r, float g, float bl, float fuzz) {
    return guarded([&] { return push_mat(b, Material::glossy(Rgb(r, g, bl), fuzz)); });
}
int rrh_mat_dielectric(rrh_builder* b, float ri) {
    return guarded([&] { return push_mat(b, Material::dielectric(ri)); });
}
int rrh_mat_isotropic(rrh_builder* b, float r, float g, float bl) {
    return guarded([&] { return push_mat(b, Material::isotropic(Rgb(r, g, bl))); });
}
int rrh_mat_emissive(rrh_builder* b, float r, float g, float bl) {
    return guarded([&] { return push_mat(b, Material::emissive(Rgb(r, g, bl))); });
}

int rrh_sphere(rrh_builder* b, const float c[3], float radius, int m) {
    return guarded([&] { return push_geom(b, Geometry::sphere(v3(c), radius, mat(b, m))); });
}
int rrh_moving_sphere(rrh_builder* b, const float c0[3], const float c1[3], float t0, float t1, float radius, int m) {
    return guarded([&] { return push_geom(b, Geometry::moving_sphere(v3(c0), v3(c1), t0, t1, radius, mat(b, m))); });
}
int rrh_xyrect(rrh_builder* b, float x0, float x1, float y0, float y1, float k, int m) {
    return guarded([&] { return push_geom(b, Geometry::xyrect(x0, x1, y0, y1, k, mat(b, m))); });
}
int rrh_xzrect(rrh_builder* b, float x0, float x1, float z0, float z1, float k, int m) {
    return guarded([&] { return push_geom(b, Geometry::xzrect(x0, x1, z0, z1, k, mat(b, m))); });
}
int rrh_yzrect(rrh_builder* b, float y0, float y1, float z0, float z1, float k, int m) {
    return guarded([&] { return push_geom(b, Geometry::yzrect(y0, y1, z0, z1, k, mat(b, m))); });
}
int rrh_cuboid(rrh_builder* b, const float p0[3], const float p1[3], int m) {
    return guarded([&] { return push_geom(b, Geometry::cuboid(v3(p0), v3(p1), mat(b, m))); });
}
int rrh_constant_medium(rrh_builder* b, int boundary, float density, float r, float g, float bl) {
    return guarded([&] { return push_geom(b, Geometry::constant_medium(geom(b, boundary), density, Rgb(r, g, bl))); });
}
int rrh_instance_translation(rrh_builder* b, int g, const float d[3]) {
    return guarded([&] { return push_geom(b, Geometry::instance_translation(geom(b, g), v3(d))); });
}
int rrh_instance_rotation(rrh_builder* b, int g, int axis, float angle) {
    return guarded([&] {
        if (axis < 0 || axis > 2) throw std::invalid_argument("rrh_instance_rotation: axis must be 0, 1 or 2");
        return push_geom(b, Geometry::instance_rotation(geom(b, g), static_cast<Axis>(axis), angle));
    });
}
int rrh_bvh_node(rrh_builder* b, const int* gs, uint32_t n, float t0, float t1) {
    return guarded([&] {
        std::vector<GeometryPtr> items;
        items.reserve(n);
        for (uint32_t i = 0; i < n; ++i) items.push_back(geom(b, gs[i]));
        return push_geom(b, Geometry::bvh_node(items, t0, t1, 0, items.size(), b->rng));
    });
}
int rrh_set_bvh_builder(rrh_builder* b, uint32_t kind) {
    return guarded([&] {
        if (kind > 1) throw std::invalid_argument("rrh_set_bvh_builder: kind must be RRH_BVH_REFERENCE or RRH_BVH_SAH");
        b->rng.sah_builder = kind == 1;
        return 0;
    });
}
int rrh_bvh_node_explicit(rrh_builder* b, int left, int right, uint32_t axis, int has_box, const float bmin[3],
                          const float bmax[3]) {
    return guarded([&] {
        std::optional<AABB> box;
        if (has_box) box = AABB{v3(bmin), v3(bmax)};
        return push_geom(b, Geometry::bvh_node_explicit(geom(b, left), geom(b, right), axis, box));
    });
}
int rrh_collider_list(rrh_builder* b, const int* gs, uint32_t n) {
    return guarded([&] {
        std::vector<GeometryPtr> items;
        for (uint32_t i = 0; i < n; ++i) items.push_back(geom(b, gs[i]));
        return push_geom(b, Geometry::collider_list(std::move(items)));
    });
}
int rrh_triangle(rrh_builder* b, const float p0[3], const float p1[3], const float p2[3], int m) {
    return guarded([&] { return push_geom(b, Geometry::triangle(v3(p0), v3(p1), v3(p2), mat(b, m))); });
}
int rrh_load_obj(rrh_builder* b, const char* path, float scale, int m) {
    return guarded([&] { return push_geom(b, Geometry::load_obj(path, scale, mat(b, m), b->rng)); });
}
int rrh_load_obj_hardened(rrh_builder* b, const char* path, float scale, int m) {
    return guarded([&] { return push_geom(b, Geometry::load_obj(path, scale, mat(b, m), b->rng, true)); });
}

int rrh_bounding_box(rrh_builder* b, int g, float t0, float t1, float out[6]) {
    return guarded([&] {
        auto bb = geom(b, g)->bounding_box(t0, t1);
        if (!bb) return 0;
        for (int a = 0; a < 3; ++a) {
            out[a] = bb->minimum[a];
            out[3 + a] = bb->maximum[a];
        }
        return 1;
    });
}

int rrh_scene_add(rrh_builder* b, int g) {
    return guarded([&] {
        b->scene.add(geom(b, g));
        return 0;
    });
}
int rrh_set_lights(rrh_builder* b, int g) {
    return guarded([&] {
        b->lights = geom(b, g);
        return 0;
    });
}
int rrh_set_camera(rrh_builder* b, const float from[3], const float at[3], const float vup[3], float vfov, float aperture,
                   float focus_dist, float aspect, float t0, float t1) {
    return guarded([&] {
        b->cam = Camera(v3(from), v3(at), v3(vup), vfov, aperture, focus_dist, aspect, t0, t1);
        return 0;
    });
}
int rrh_set_background(rrh_builder* b, float r, float g, float bl) {
    b->background = Rgb(r, g, bl);
    return 0;
}

int rrh_build_preset(rrh_builder* b, const char* name_c, const rrh_preset_options* opt, rrh_preset_defaults* defaults) {
    return guarded([&] {
        const std::string name = name_c ? name_c : "";
        rrh_preset_options o{};
        o.image = -1;
        if (opt) o = *opt;
        const ImageRGB8* earthmap = nullptr;
        if (o.image >= 0) {
            const Texture& t = tex(b, o.image);
            if (t.data()->type != RR_TEX_IMAGE) throw std::invalid_argument("preset option `image` is not an image texture");
            earthmap = &b->scene.atlas.at(t.data()->image);
        }
        SceneBundle s;
        if (name == "default_scene") s = default_scene(b->rng);
        else if (name == "random_spheres") s = random_spheres(b->rng);
        else if (name == "random_moving_spheres") s = random_moving_spheres(b->rng);
        else if (name == "two_spheres") s = two_spheres(b->rng);
        else if (name == "two_perlin_spheres") s = two_perlin_spheres(b->rng);
        else if (name == "earth") s = earth(b->rng, earthmap);
        else if (name == "rect_light") s = rect_light(b->rng);
        else if (name == "cornell_box") s = cornell_box(b->rng);
        else if (name == "cornell_smoke") s = cornell_smoke(b->rng, o.repair != 0);
        else if (name == "cornell_pedestal") {
            if (!o.obj_path) throw std::invalid_argument("cornell_pedestal needs obj_path");
            s = cornell_pedestal(b->rng, o.obj_path, o.repair != 0);
        } else if (name == "final_scene") s = final_scene(b->rng, earthmap, o.repair != 0, o.moving_sphere != 0);
        else if (name == "synthetic_mesh") {
            if (o.n_triangles == 0) throw std::invalid_argument("synthetic_mesh needs n_triangles > 0");
            s = synthetic_mesh(b->rng, o.n_triangles, o.n_spheres);
        } else
            throw std::invalid_argument("unknown preset scene: " + name);
        b->scene = std::move(s.scene);
        b->lights = s.lights;
        b->cam = s.cam;
        b->background = s.background;
        if (defaults) {
            defaults->image_width = s.image_width;
            defaults->image_height = s.image_height;
            defaults->samples_per_pixel = s.samples_per_pixel;
            defaults->max_depth = s.max_depth;
            defaults->aspect_ratio = s.aspect_ratio;
        }
        return 0;
    });
}

int rrh_flatten(rrh_builder* b, rr_flat_scene* out) {
    return guarded([&] {
        if (!out) throw std::invalid_argument("rrh_flatten: out is NULL");
        b->flat = flatten(b->scene, b->lights, b->cam, b->background);
        *out = b->flat->view();
        return 0;
    });
}

}  // extern "C"

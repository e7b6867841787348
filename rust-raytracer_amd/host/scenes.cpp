// The scene functions of src/main.rs:8-541 as seedable C++ (inputs to the device
// path, not part of it), plus the synthetic large-mesh scene.  Each keeps the
// reference's order of random draws so that one seed fixes the whole scene.
#include "scene.hpp"

namespace rr {

namespace {
SceneBundle make_bundle(float aspect, uint32_t width, uint32_t spp, uint32_t depth, Rgb background) {
    SceneBundle b;
    b.aspect_ratio = aspect;
    b.image_width = width;
    b.image_height = image_height_for(width, aspect);
    b.samples_per_pixel = spp;
    b.max_depth = depth;
    b.background = background;
    return b;
}
const Rgb SKY(0.7f, 0.8f, 1.0f);
const Rgb BLACK(0.f, 0.f, 0.f);
GeometryPtr no_lights() { return Geometry::collider_list({}); }

Camera book_camera(float aspect, float aperture) {  // main.rs:60-72 and siblings
    return Camera(Point3(13, 2, 3), Point3(0, 0, 0), Vec3(0, 1, 0), 20.0f, aperture, 10.0f, aspect, 0.0f, 1.0f);
}
Camera cornell_camera(Point3 look_from) {  // main.rs:335-347
    return Camera(look_from, Point3(278, 278, 0), Vec3(0, 1, 0), 40.0f, 0.0f, 10.0f, 1.0f, 0.0f, 1.0f);
}
}  // namespace

SceneBundle default_scene(Rng&) {  // main.rs:8-48
    SceneBundle b = make_bundle(16.0f / 9.0f, 800, 500, 200, SKY);
    Point3 from(3, 3, 2), at(0, 0, -1);
    b.cam = Camera(from, at, Vec3(0, 1, 0), 30.0f, 0.1f, length(from - at), b.aspect_ratio, 0.0f, 1.0f);
    b.scene.add(Geometry::sphere(Point3(0, -100.5f, -1), 100.0f, Material::lambertian(Texture::solid_color(Rgb(0.8f, 0.8f, 0.0f)))));
    b.scene.add(Geometry::sphere(Point3(0, 0, -1), 0.5f, Material::lambertian(Texture::solid_color(Rgb(0.1f, 0.2f, 0.5f)))));
    b.scene.add(Geometry::sphere(Point3(-1, 0, -1), 0.5f, Material::dielectric(1.5f)));
    b.scene.add(Geometry::sphere(Point3(-1, 0, -1), -0.4f, Material::dielectric(1.5f)));
    b.scene.add(Geometry::sphere(Point3(1, 0, -1), 0.5f, Material::glossy(Rgb(0.8f, 0.6f, 0.2f), 0.4f)));
    b.lights = no_lights();
    return b;
}

SceneBundle random_spheres(Rng& rng) {  // main.rs:50-115
    SceneBundle b = make_bundle(3.0f / 2.0f, 600, 250, 50, SKY);
    b.cam = book_camera(b.aspect_ratio, 0.1f);
    std::vector<GeometryPtr> spheres;
    b.scene.add(Geometry::sphere(Point3(0, -1000, 0), 1000.0f,
                                 Material::lambertian(Texture::checkered(Rgb(0.2f, 0.3f, 0.1f), Rgb(0.9f, 0.9f, 0.9f)))));
    for (int a = -11; a < 11; ++a) {
        for (int c = -11; c < 11; ++c) {
            float choose_mat = rng.random();
            float cx = float(a) + 0.9f * rng.random();
            float cz = float(c) + 0.9f * rng.random();
            Point3 center(cx, 0.2f, cz);
            if (length(center - Point3(4.0f, 0.2f, 0.0f)) > 0.9f) {
                if (choose_mat < 0.8f) {
                    Rgb r0 = rng.vec3_random();
                    Rgb r1 = rng.vec3_random();
                    spheres.push_back(Geometry::sphere(center, 0.2f, Material::lambertian(Texture::solid_color(r0 * r1))));
                } else if (choose_mat < 0.95f) {
                    Rgb albedo = rng.vec3_randrange(0.5f, 1.0f);
                    float fuzz = rng.randrange(0.0f, 0.5f);
                    spheres.push_back(Geometry::sphere(center, 0.2f, Material::glossy(albedo, fuzz)));
                } else {
                    spheres.push_back(Geometry::sphere(center, 0.2f, Material::dielectric(1.5f)));
                }
            }
        }
    }
    spheres.push_back(Geometry::sphere(Point3(0, 1, 0), 1.0f, Material::dielectric(1.5f)));
    spheres.push_back(Geometry::sphere(Point3(-4, 1, 0), 1.0f, Material::lambertian(Texture::solid_color(Rgb(0.4f, 0.2f, 0.1f)))));
    spheres.push_back(Geometry::sphere(Point3(4, 1, 0), 1.0f, Material::glossy(Rgb(0.7f, 0.6f, 0.5f), 0.1f)));
    b.scene.add(Geometry::bvh_node(spheres, 0.0f, 1.0f, 0, spheres.size(), rng));
    b.lights = no_lights();
    return b;
}

SceneBundle random_moving_spheres(Rng& rng) {  // main.rs:117-182
    SceneBundle b = make_bundle(3.0f / 2.0f, 400, 100, 50, SKY);
    b.cam = book_camera(b.aspect_ratio, 0.1f);
    std::vector<GeometryPtr> spheres;
    b.scene.add(Geometry::sphere(Point3(0, -1000, 0), 1000.0f, Material::lambertian(Texture::solid_color(Rgb(0.5f, 0.5f, 0.5f)))));
    for (int a = -21; a < 21; ++a) {
        for (int c = -21; c < 21; ++c) {
            float choose_mat = rng.random();
            float cx = float(a) / 2.0f + 0.9f * rng.random();
            float cz = float(c) / 2.0f + 0.9f * rng.random();
            Point3 center(cx, 0.2f, cz);
            if (length(center - Point3(4.0f, 0.2f, 0.0f)) > 0.9f) {
                if (choose_mat < 0.8f) {
                    Rgb r0 = rng.vec3_random();
                    Rgb r1 = rng.vec3_random();
                    Material mat = Material::lambertian(Texture::solid_color(r0 * r1));
                    Point3 center2 = center + Vec3(0.0f, rng.randrange(0.0f, 0.5f), 0.0f);
                    spheres.push_back(Geometry::moving_sphere(center, center2, 0.0f, 1.0f, 0.1f, mat));
                } else if (choose_mat < 0.95f) {
                    Rgb albedo = rng.vec3_randrange(0.5f, 1.0f);
                    float fuzz = rng.randrange(0.0f, 0.5f);
                    spheres.push_back(Geometry::sphere(center, 0.1f, Material::glossy(albedo, fuzz)));
                } else {
                    spheres.push_back(Geometry::sphere(center, 0.1f, Material::dielectric(1.5f)));
                }
            }
        }
    }
    b.scene.add(Geometry::bvh_node(spheres, 0.0f, 1.0f, 0, spheres.size(), rng));
    b.scene.add(Geometry::sphere(Point3(0, 1, 0), 1.0f, Material::dielectric(1.5f)));
    b.scene.add(Geometry::sphere(Point3(-4, 1, 0), 1.0f, Material::lambertian(Texture::solid_color(Rgb(0.4f, 0.2f, 0.1f)))));
    b.scene.add(Geometry::sphere(Point3(4, 1, 0), 1.0f, Material::glossy(Rgb(0.7f, 0.6f, 0.5f), 0.1f)));
    b.lights = no_lights();
    return b;
}

SceneBundle two_spheres(Rng&) {  // main.rs:184-216
    SceneBundle b = make_bundle(16.0f / 9.0f, 800, 100, 50, SKY);
    b.cam = book_camera(b.aspect_ratio, 0.0f);
    Material checker = Material::lambertian(Texture::checkered(Rgb(0.2f, 0.3f, 0.1f), Rgb(0.9f, 0.9f, 0.9f)));
    b.scene.add(Geometry::sphere(Point3(0, -8, 0), 8.0f, checker));
    b.scene.add(Geometry::sphere(Point3(0, 8, 0), 8.0f, checker));
    b.lights = no_lights();
    return b;
}

SceneBundle two_perlin_spheres(Rng& rng) {  // main.rs:218-250
    SceneBundle b = make_bundle(16.0f / 9.0f, 600, 100, 50, SKY);
    b.cam = book_camera(b.aspect_ratio, 0.0f);
    Material pertext = Material::lambertian(Texture::noise(4.0f, 7, rng));
    b.scene.add(Geometry::sphere(Point3(0, -1000, 0), 1000.0f, pertext));
    b.scene.add(Geometry::sphere(Point3(0, 2, 0), 2.0f, pertext));
    b.lights = no_lights();
    return b;
}

static Texture earth_texture(SceneColliders& scene, const ImageRGB8* earthmap) {
    if (!earthmap || earthmap->rgb.empty()) throw std::invalid_argument("this scene needs the decoded earth map");
    return scene.load_image_rgb8(earthmap->width, earthmap->height, earthmap->rgb.data());
}

SceneBundle earth(Rng&, const ImageRGB8* earthmap) {  // main.rs:252-283
    SceneBundle b = make_bundle(16.0f / 9.0f, 800, 100, 50, SKY);
    b.cam = book_camera(b.aspect_ratio, 0.0f);
    Material surface = Material::lambertian(earth_texture(b.scene, earthmap));
    b.scene.add(Geometry::sphere(Point3(0, 0, 0), 2.0f, surface));
    b.lights = no_lights();
    return b;
}

SceneBundle rect_light(Rng&) {  // main.rs:285-323 (the lights are never added to the scene)
    SceneBundle b = make_bundle(16.0f / 9.0f, 800, 800, 50, BLACK);
    b.cam = Camera(Point3(26, 3, 6), Point3(0, 2, 0), Vec3(0, 1, 0), 20.0f, 0.0f, 10.0f, b.aspect_ratio, 0.0f, 1.0f);
    b.scene.add(Geometry::sphere(Point3(0, -1000, 0), 1000.0f, Material::lambertian(Texture::solid_color(Rgb(0.9f, 0.9f, 0.9f)))));
    b.scene.add(Geometry::sphere(Point3(0, 2, 0), 2.0f, Material::lambertian(Texture::solid_color(Rgb(0.2f, 0.8f, 1.0f)))));
    Material difflight = Material::emissive(Rgb(5, 2, 2));
    b.lights = Geometry::collider_list({Geometry::xyrect(3, 5, 1, 3, -2, difflight), Geometry::sphere(Point3(0, 7, 0), 1.5f, difflight)});
    return b;
}

namespace {
struct CornellMats {
    Material red = Material::lambertian(Texture::solid_color(Rgb(0.65f, 0.05f, 0.05f)));
    Material white = Material::lambertian(Texture::solid_color(Rgb(0.73f, 0.73f, 0.73f)));
    Material green = Material::lambertian(Texture::solid_color(Rgb(0.12f, 0.45f, 0.15f)));
};
}  // namespace

SceneBundle cornell_box(Rng&) {  // main.rs:325-368
    SceneBundle b = make_bundle(1.0f, 600, 5000, 200, BLACK);
    b.cam = cornell_camera(Point3(278, 278, -800));
    CornellMats m;
    Material light = Material::emissive(Rgb(25, 25, 25));
    b.scene.add(Geometry::yzrect(0, 555, 0, 555, 555, m.green));
    b.scene.add(Geometry::yzrect(0, 555, 0, 555, 0, m.red));
    b.scene.add(Geometry::xzrect(213, 343, 227, 332, 554, light));
    b.scene.add(Geometry::xzrect(0, 555, 0, 555, 0, m.white));
    b.scene.add(Geometry::xzrect(0, 555, 0, 555, 555, m.white));
    b.scene.add(Geometry::xyrect(0, 555, 0, 555, 555, m.white));
    b.scene.add(Geometry::instance_translation(
        Geometry::instance_rotation(Geometry::cuboid(Point3(0, 0, 0), Point3(165, 330, 165), m.white), Axis::Y, 15.0f),
        Vec3(265, 0, 295)));
    b.scene.add(Geometry::instance_translation(
        Geometry::instance_rotation(Geometry::cuboid(Point3(0, 0, 0), Point3(165, 165, 165), m.white), Axis::Y, -18.0f),
        Vec3(130, 0, 65)));
    b.lights = Geometry::xzrect(213, 343, 227, 332, 554, m.white);
    return b;
}

SceneBundle cornell_smoke(Rng&, bool repair) {  // main.rs:370-414
    SceneBundle b = make_bundle(1.0f, 600, 200, 200, BLACK);
    b.cam = cornell_camera(Point3(278, 278, -800));
    CornellMats m;
    Material light = Material::emissive(Rgb(7, 7, 7));
    GeometryPtr light_rect = Geometry::xzrect(113, 443, 127, 432, 554, light);
    b.scene.add(Geometry::yzrect(0, 555, 0, 555, 555, m.green));
    b.scene.add(Geometry::yzrect(0, 555, 0, 555, 0, m.red));
    if (repair) b.scene.add(light_rect);
    b.scene.add(Geometry::xzrect(0, 555, 0, 555, 0, m.white));
    b.scene.add(Geometry::xzrect(0, 555, 0, 555, 555, m.white));
    b.scene.add(Geometry::xyrect(0, 555, 0, 555, 555, m.white));
    b.scene.add(Geometry::constant_medium(
        Geometry::instance_translation(
            Geometry::instance_rotation(Geometry::cuboid(Point3(0, 0, 0), Point3(165, 330, 165), m.white), Axis::Y, 15.0f),
            Vec3(265, 0, 295)),
        0.01f, Rgb(0, 0, 0)));
    b.scene.add(Geometry::constant_medium(
        Geometry::instance_translation(
            Geometry::instance_rotation(Geometry::cuboid(Point3(0, 0, 0), Point3(165, 165, 165), m.white), Axis::Y, -18.0f),
            Vec3(130, 0, 65)),
        0.01f, Rgb(1, 1, 1)));
    b.lights = repair ? light_rect : Geometry::collider_list({light_rect});
    return b;
}

SceneBundle cornell_pedestal(Rng& rng, const std::string& obj_path, bool repair) {  // main.rs:416-464
    SceneBundle b = make_bundle(1.0f, 600, 200, 100, BLACK);
    b.cam = cornell_camera(Point3(278, 278, -800));
    CornellMats m;
    Material light = Material::emissive(Rgb(2, 2, 2));
    GeometryPtr light_rect = Geometry::xzrect(113, 443, 127, 432, 554, light);
    b.scene.add(Geometry::yzrect(0, 555, 0, 555, 555, m.green));
    b.scene.add(Geometry::yzrect(0, 555, 0, 555, 0, m.red));
    if (repair) b.scene.add(light_rect);
    b.scene.add(Geometry::xzrect(0, 555, 0, 555, 0, m.white));
    b.scene.add(Geometry::xzrect(0, 555, 0, 555, 555, m.white));
    b.scene.add(Geometry::xyrect(0, 555, 0, 555, 555, m.white));
    b.scene.add(Geometry::instance_translation(
        Geometry::instance_rotation(Geometry::cuboid(Point3(0, 0, 0), Point3(125, 125, 125), m.white), Axis::Y, 45.0f),
        Vec3(188, 0, 178)));
    b.scene.add(Geometry::instance_translation(
        Geometry::instance_rotation(
            Geometry::instance_rotation(Geometry::load_obj(obj_path, 80.0f, m.white, rng), Axis::Y, 145.0f), Axis::Z, -30.0f),
        Vec3(270, 200, 178)));
    b.lights = repair ? light_rect : Geometry::collider_list({light_rect});
    return b;
}

SceneBundle final_scene(Rng& rng, const ImageRGB8* earthmap, bool repair, bool moving_sphere) {  // main.rs:466-541
    SceneBundle b = make_bundle(1.0f, 800, 50, 50, BLACK);
    b.cam = cornell_camera(Point3(478, 278, -600));
    Material ground = Material::lambertian(Texture::solid_color(Rgb(0.48f, 0.83f, 0.53f)));
    std::vector<GeometryPtr> boxes;
    const int boxes_per_side = 20;
    for (int i = 0; i < boxes_per_side; ++i) {
        for (int j = 0; j < boxes_per_side; ++j) {
            const float w = 100.0f;
            float x0 = -1000.0f + float(i) * w;
            float z0 = -1000.0f + float(j) * w;
            float y1 = rng.randrange(1.0f, 101.0f);
            boxes.push_back(Geometry::cuboid(Point3(x0, 0.0f, z0), Point3(x0 + w, y1, z0 + w), ground));
        }
    }
    b.scene.add(Geometry::bvh_node(boxes, 0.0f, 1.0f, 0, boxes.size(), rng));

    Material light = Material::emissive(Rgb(7, 7, 7));
    GeometryPtr light_rect = Geometry::xzrect(123, 423, 147, 412, 554, light);
    if (repair) b.scene.add(light_rect);

    if (moving_sphere) {
        // The book's motion-blurred sphere; HEAD's final_scene has dropped it, the constructor
        // (collider.rs:62) and the camera shutter (main.rs:487) are still there.
        Point3 c1(400, 400, 200);
        b.scene.add(Geometry::moving_sphere(c1, c1 + Vec3(30, 0, 0), 0.0f, 1.0f, 50.0f,
                                            Material::lambertian(Texture::solid_color(Rgb(0.7f, 0.3f, 0.1f)))));
    }
    b.scene.add(Geometry::sphere(Point3(260, 150, 45), 50.0f, Material::dielectric(1.5f)));
    b.scene.add(Geometry::sphere(Point3(0, 150, 145), 50.0f, Material::glossy(Rgb(0.8f, 0.8f, 0.9f), 1.0f)));

    GeometryPtr boundary = Geometry::sphere(Point3(360, 150, 145), 70.0f, Material::dielectric(1.5f));
    b.scene.add(boundary);
    b.scene.add(Geometry::constant_medium(boundary, 0.2f, Rgb(0.2f, 0.4f, 0.9f)));
    b.scene.add(Geometry::constant_medium(Geometry::sphere(Point3(0, 0, 0), 5000.0f, Material::dielectric(1.5f)), 0.0001f, Rgb(1, 1, 1)));

    b.scene.add(Geometry::sphere(Point3(400, 200, 400), 100.0f, Material::lambertian(earth_texture(b.scene, earthmap))));
    b.scene.add(Geometry::sphere(Point3(220, 280, 300), 80.0f, Material::lambertian(Texture::noise(0.1f, 7, rng))));

    std::vector<GeometryPtr> spheres;
    Material white = Material::lambertian(Texture::solid_color(Rgb(0.73f, 0.73f, 0.73f)));
    for (int i = 0; i < 1000; ++i) spheres.push_back(Geometry::sphere(rng.vec3_randrange(0.0f, 165.0f), 10.0f, white));
    b.scene.add(Geometry::instance_translation(
        Geometry::instance_rotation(Geometry::bvh_node(spheres, 0.0f, 1.0f, 0, spheres.size(), rng), Axis::Y, 15.0f),
        Vec3(-100, 270, 395)));
    b.lights = repair ? light_rect : Geometry::collider_list({light_rect});
    return b;
}

// A height field of n_triangles triangles over [0,555]^2 (two per grid cell, vertices
// jittered from the seed) with n_spheres small spheres floating above it, one BVH over
// all of them, and one XZRect light that is both in the scene and the `lights` argument.
SceneBundle synthetic_mesh(Rng& rng, uint32_t n_triangles, uint32_t n_spheres) {
    SceneBundle b = make_bundle(16.0f / 9.0f, 3840, 4096, 50, Rgb(0.05f, 0.06f, 0.09f));
    b.cam = Camera(Point3(278, 420, -560), Point3(278, 60, 278), Vec3(0, 1, 0), 40.0f, 0.0f, 10.0f, b.aspect_ratio, 0.0f, 1.0f);
    const uint32_t cells = (n_triangles + 1) / 2;
    uint32_t side = 1;
    while (static_cast<uint64_t>(side) * side < cells) ++side;
    const float extent = 555.0f;
    const float step = extent / float(side);
    std::vector<Point3> verts(static_cast<size_t>(side + 1) * (side + 1));
    for (uint32_t j = 0; j <= side; ++j)
        for (uint32_t i = 0; i <= side; ++i) {
            float x = float(i) * step, z = float(j) * step;
            float y = 60.0f + 40.0f * std::sin(x * 0.031f) * std::cos(z * 0.023f) + 12.0f * std::sin(x * 0.11f + z * 0.07f);
            float jx = rng.randrange(-0.2f, 0.2f) * step, jy = rng.randrange(-0.2f, 0.2f) * step, jz = rng.randrange(-0.2f, 0.2f) * step;
            verts[static_cast<size_t>(j) * (side + 1) + i] = Point3(x + jx, y + jy, z + jz);
        }
    std::vector<Material> palette;
    for (int i = 0; i < 16; ++i) palette.push_back(Material::lambertian(Texture::solid_color(rng.vec3_randrange(0.2f, 0.9f))));
    std::vector<GeometryPtr> prims;
    prims.reserve(static_cast<size_t>(n_triangles) + n_spheres);
    uint32_t made = 0;
    for (uint32_t j = 0; j < side && made < n_triangles; ++j)
        for (uint32_t i = 0; i < side && made < n_triangles; ++i) {
            const Point3& v00 = verts[static_cast<size_t>(j) * (side + 1) + i];
            const Point3& v10 = verts[static_cast<size_t>(j) * (side + 1) + i + 1];
            const Point3& v01 = verts[static_cast<size_t>(j + 1) * (side + 1) + i];
            const Point3& v11 = verts[static_cast<size_t>(j + 1) * (side + 1) + i + 1];
            const Material& m = palette[((i >> 4) + 3 * (j >> 4)) & 15];
            prims.push_back(Geometry::triangle(v00, v01, v10, m));
            ++made;
            if (made < n_triangles) {
                prims.push_back(Geometry::triangle(v10, v01, v11, m));
                ++made;
            }
        }
    for (uint32_t s = 0; s < n_spheres; ++s) {
        float x = rng.randrange(0.0f, extent), z = rng.randrange(0.0f, extent);
        float y = rng.randrange(130.0f, 330.0f);
        float r = rng.randrange(0.5f, 2.0f);
        prims.push_back(Geometry::sphere(Point3(x, y, z), r, palette[s & 15]));
    }
    b.scene.add(Geometry::bvh_node(prims, 0.0f, 1.0f, 0, prims.size(), rng));
    GeometryPtr light_rect = Geometry::xzrect(123, 423, 147, 412, 554, Material::emissive(Rgb(7, 7, 7)));
    b.scene.add(light_rect);
    b.lights = light_rect;
    return b;
}

}  // namespace rr

// Host-side scene construction API (C++17), mirroring the reference's
// constructors so that scene code reads like src/main.rs:
//   Geometry::{sphere, moving_sphere, xyrect, xzrect, yzrect, cuboid,
//              constant_medium, instance_translation, instance_rotation,
//              bvh_node, collider_list, triangle, load_obj}   (src/collider.rs:56-267)
//   Material::{lambertian, glossy, dielectric, isotropic, emissive} (src/material.rs:58-82)
//   Texture::{solid_color, checkered, noise, image}            (src/texture.rs:41-64)
//   Camera::new                                                (src/camera.rs:34-75)
//   SceneColliders::{new, add, load_image}                     (src/collider.rs:737-750)
// Differences that are deliberate:
//   * randomness comes from an explicit, seedable Rng instead of thread_rng()
//     (src/utils.rs:41-55), so scenes, BVH split axes and Perlin tables are
//     reproducible;
//   * nodes are immutable and shared (shared_ptr) instead of deep-cloned, and
//     bvh_node sorts index sub-ranges in place: same tree, O(n log^2 n) instead of
//     the reference's O(n^2) cloning (src/collider.rs:177,196-197);
//   * nothing here intersects rays: evaluation lives on the device.
#pragma once

#include <array>
#include <cmath>
#include <cstdint>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rr_hip.h"

namespace rr {

// ------------------------------------------------------------------ math
struct Vec3 {
    float x = 0.f, y = 0.f, z = 0.f;
    Vec3() = default;
    Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(Vec3 a, Vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline Vec3 operator*(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator/(Vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 cross(Vec3 a, Vec3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float length(Vec3 a) { return std::sqrt(dot(a, a)); }
inline Vec3 normalize(Vec3 a) { return a / length(a); }
using Rgb = Vec3;
using Point3 = Vec3;

enum class Axis { X = 0, Y = 1, Z = 2 };

constexpr float PI_F = 3.14159265358979323846f;
inline float to_radians(float degrees) { return degrees * PI_F / 180.0f; }  // utils.rs:37-39

// ------------------------------------------------------------------- rng
// PCG32 (XSH-RR).  Stands in for rand's thread_rng(); only the distributions of
// src/utils.rs:41-55 are kept: random() is uniform on the 2^-23 grid of [0,1),
// randrange(lo,hi) is lo + random()*(hi-lo) redrawn while >= hi, randuint has an
// EXCLUSIVE upper bound (gen_range(lo..hi)).
class Rng {
   public:
    explicit Rng(uint64_t seed = 1, uint64_t stream = 0x5851F42D4C957F2DULL) { reseed(seed, stream); }
    void reseed(uint64_t seed, uint64_t stream = 0x5851F42D4C957F2DULL) {
        state_ = 0;
        inc_ = (stream << 1) | 1u;
        next_u32();
        state_ += seed;
        next_u32();
    }
    uint32_t next_u32() {
        uint64_t old = state_;
        state_ = old * 6364136223846793005ULL + inc_;
        uint32_t xs = static_cast<uint32_t>(((old >> 18) ^ old) >> 27);
        uint32_t rot = static_cast<uint32_t>(old >> 59);
        return (xs >> rot) | (xs << ((32u - rot) & 31u));
    }
    float random() { return static_cast<float>(next_u32() >> 9) * (1.0f / 8388608.0f); }
    float randrange(float lo, float hi) {
        for (;;) {
            float v = random() * (hi - lo) + lo;
            if (v < hi) return v;
        }
    }
    // Which tree Geometry::bvh_node builds with this generator: false (default) the reference's (collider.rs:173-212: median
    // split of a random axis: it is what draws from the generator), true a surface-area-heuristic tree (bvh_node_sah), a
    // labelled non-parity option that draws nothing.
    bool sah_builder = false;
    // uniform integer in [lo, hi)
    uint32_t randuint(uint32_t lo, uint32_t hi) {
        uint32_t span = hi - lo;
        return lo + static_cast<uint32_t>((static_cast<uint64_t>(next_u32()) * span) >> 32);
    }
    Vec3 vec3_random() {
        float a = random(), b = random(), c = random();
        return {a, b, c};
    }
    Vec3 vec3_randrange(float lo, float hi) {
        float a = randrange(lo, hi), b = randrange(lo, hi), c = randrange(lo, hi);
        return {a, b, c};
    }

   private:
    uint64_t state_ = 0, inc_ = 1;
};

// -------------------------------------------------------------- textures
struct Perlin {  // perlin.rs:5-46 (table generation only)
    float ranvec[RR_PERLIN_POINTS][3];
    uint8_t perm_x[RR_PERLIN_POINTS], perm_y[RR_PERLIN_POINTS], perm_z[RR_PERLIN_POINTS];
    static std::shared_ptr<const Perlin> make(Rng& rng);
};

struct TextureData {
    uint32_t type = RR_TEX_SOLID;
    Rgb color0, color1;
    std::shared_ptr<const Perlin> perlin;
    float scale = 0.f;
    uint32_t depth = 0;
    uint32_t image = 0;
};

class Texture {
   public:
    Texture() = default;
    explicit Texture(std::shared_ptr<const TextureData> d) : d_(std::move(d)) {}
    static Texture solid_color(Rgb color);
    static Texture checkered(Rgb odd_color, Rgb even_color);
    static Texture noise(float scale, uint32_t turb, Rng& rng);
    static Texture image(uint32_t atlas_index);
    const std::shared_ptr<const TextureData>& data() const { return d_; }

   private:
    std::shared_ptr<const TextureData> d_;
};

// ------------------------------------------------------------- materials
struct MaterialData {
    uint32_t type = RR_MAT_LAMBERTIAN;
    Texture texture;
    Rgb color;
    float param = 0.f;
};

class Material {
   public:
    Material() = default;
    explicit Material(std::shared_ptr<const MaterialData> d) : d_(std::move(d)) {}
    static Material lambertian(Texture albedo);
    static Material glossy(Rgb albedo, float fuzz);
    static Material dielectric(float refraction_index);
    static Material isotropic(Rgb color);
    static Material emissive(Rgb color);
    const std::shared_ptr<const MaterialData>& data() const { return d_; }

   private:
    std::shared_ptr<const MaterialData> d_;
};

// -------------------------------------------------------------- geometry
struct AABB {  // aabb.rs:5-36
    Point3 minimum, maximum;
    static AABB surrounding_box(const AABB& a, const AABB& b);
};

enum class GeometryType : uint8_t {
    Sphere,
    MovingSphere,
    XYRect,
    XZRect,
    YZRect,
    Cuboid,
    ConstantMedium,
    TranslateInstance,
    RotationInstance,
    BVHNode,
    ColliderList,
    Triangle
};

class Geometry;
using GeometryPtr = std::shared_ptr<const Geometry>;

class Geometry {
   public:
    GeometryType type;
    Material material;
    // Parameter block, meaning by type:
    //  Sphere: p[0]=center, f[0]=radius
    //  MovingSphere: p[0],p[1]=centers, f[0]=time0, f[1]=time1, f[2]=radius
    //  XY/XZ/YZRect: f[0..3]=a0,a1,b0,b1 (plane coordinates in axis order), f[4]=k
    //  Cuboid: p[0],p[1]=corners, children = 6 rects
    //  ConstantMedium: children[0]=boundary, f[0]=-1/density
    //  TranslateInstance: children[0], p[0]=displacement
    //  RotationInstance: children[0], axis, f[0]=angle, f[1]=sin, f[2]=cos, has_box, box
    //  BVHNode: children[0]=left, children[1]=right, has_box, box, axis_index
    //  ColliderList: children
    //  Triangle: p[0..2] vertices, p[3]=unit plane normal
    Vec3 p[4];
    float f[5] = {0, 0, 0, 0, 0};
    Axis axis = Axis::X;
    uint32_t axis_index = 0;
    bool has_box = false;
    AABB box;
    std::vector<GeometryPtr> children;

    static GeometryPtr sphere(Point3 center, float radius, Material m);
    static GeometryPtr moving_sphere(Point3 c0, Point3 c1, float time0, float time1, float radius, Material m);
    static GeometryPtr xyrect(float x0, float x1, float y0, float y1, float k, Material m);
    static GeometryPtr xzrect(float x0, float x1, float z0, float z1, float k, Material m);
    static GeometryPtr yzrect(float y0, float y1, float z0, float z1, float k, Material m);
    static GeometryPtr cuboid(Point3 p0, Point3 p1, Material m);
    static GeometryPtr constant_medium(GeometryPtr boundary, float density, Rgb color);
    static GeometryPtr instance_translation(GeometryPtr g, Vec3 displacement);
    static GeometryPtr instance_rotation(GeometryPtr g, Axis axis, float angle_degrees);
    // Same arguments as the reference plus the Rng that supplies the split axes.
    static GeometryPtr bvh_node(const std::vector<GeometryPtr>& colliders, float time_0, float time_1,
                                size_t start, size_t end, Rng& rng);
    // NOT the reference's builder (a labelled option, never the default): a binary tree split by the surface-area heuristic
    // over 16 bins of the bounding boxes' centres along the axis where that is cheapest, `left` the side of the smaller
    // coordinates and `axis` the split axis (0 = x, 1 = y, 2 = z), so that direction[axis] >= 0 visits the nearer child first;
    // the same node shape as the reference's tree (two children per node, a single primitive is both children of its node).
    // What the reference's own tree costs next to a conventional one is measured with it under RR_SEMANTICS_INTENDED (under
    // the reference's semantics the image depends on the tree, collider.rs:555-561).
    static GeometryPtr bvh_node_sah(const std::vector<GeometryPtr>& colliders, float time_0, float time_1);
    // A BVH node with caller-chosen children, axis and box: no random draw.  For hand-built trees.
    static GeometryPtr bvh_node_explicit(GeometryPtr left, GeometryPtr right, uint32_t axis,
                                         std::optional<AABB> box);
    static GeometryPtr collider_list(std::vector<GeometryPtr> colliders);
    static GeometryPtr triangle(Point3 p0, Point3 p1, Point3 p2, Material m);
    static GeometryPtr load_obj(const std::string& path, float scale, Material m, Rng& rng, bool hardened = false);

    std::optional<AABB> bounding_box(float time_0, float time_1) const;  // collider.rs:698-713
};

struct ImageRGB8 {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgb;  // rows top-down
};

class SceneColliders {  // collider.rs:731-750
   public:
    std::vector<GeometryPtr> objects;
    std::vector<ImageRGB8> atlas;
    void add(GeometryPtr g) { objects.push_back(std::move(g)); }
    // The reference decodes any format through the `image` crate (texture.rs:19-21);
    // here the host hands over decoded RGB8, or a binary PPM (P6) file.
    Texture load_image_rgb8(uint32_t width, uint32_t height, const uint8_t* rgb);
    Texture load_image(const std::string& ppm_path);
};

struct Camera {  // camera.rs:5-75
    Point3 origin, lower_left;
    Vec3 horizontal, vertical, u, v, w;
    float lens_radius = 0.f, shutter_open = 0.f, shutter_close = 1.f;
    Camera() = default;
    Camera(Point3 look_from, Point3 look_at, Vec3 vup, float vfov, float aperture, float focus_dist,
           float aspect_ratio, float shutter_open, float shutter_close);
};

// What every scene function of src/main.rs returns (main.rs:8).
struct SceneBundle {
    Camera cam;
    SceneColliders scene;
    GeometryPtr lights;
    Rgb background;
    float aspect_ratio = 1.f;
    uint32_t image_width = 0, image_height = 0;
    uint32_t samples_per_pixel = 0, max_depth = 0;
};

// ------------------------------------------------------------- flattening
// Owns the arrays an rr_flat_scene points into.
class FlatScene {
   public:
    std::vector<rr_sphere> spheres;
    std::vector<rr_moving_sphere> moving_spheres;
    std::vector<rr_rect> rects;
    std::vector<rr_triangle> triangles;
    std::vector<rr_cuboid> cuboids;
    std::vector<rr_medium> media;
    std::vector<rr_instance> instances;
    std::vector<rr_bvh_node> bvh_nodes;
    std::vector<rr_list> lists;
    std::vector<uint32_t> child_refs;
    std::vector<rr_material> materials;
    std::vector<rr_texture> textures;
    std::vector<rr_perlin> perlin;
    std::vector<rr_image> images;
    std::vector<uint8_t> image_bytes;
    uint32_t root = RR_REF_NONE;
    rr_light light{};
    std::vector<rr_light> lights;  // the members of `lights` when it is a ColliderList (read under RR_SEMANTICS_BOOK_ESTIMATOR only)
    rr_camera camera{};
    float background[3] = {0, 0, 0};
    uint32_t n_prims = 0;

    rr_flat_scene view() const;
};

// Flatten scene + lights + camera + background into tagged arrays.  Primitive ids
// follow the depth-first order of the tree starting at scene.objects[0]; a BVH leaf
// whose two children are the same node (span 1, collider.rs:180-182) references one
// record twice; every other occurrence of a node gets its own records.
std::unique_ptr<FlatScene> flatten(const SceneColliders& scene, const GeometryPtr& lights, const Camera& cam,
                                   Rgb background);

// ---------------------------------------------------------------- scenes
// The scene functions of src/main.rs, with their HEAD parameters.  `repair` adds
// the light rect to the scene and passes a bare XZRect as `lights` where HEAD only
// pushes it to the lights list (main.rs:405,451,513-514 versus main.rs:359,367).
SceneBundle default_scene(Rng& rng);
SceneBundle random_spheres(Rng& rng);
SceneBundle random_moving_spheres(Rng& rng);
SceneBundle two_spheres(Rng& rng);
SceneBundle two_perlin_spheres(Rng& rng);
SceneBundle earth(Rng& rng, const ImageRGB8* earthmap);
SceneBundle rect_light(Rng& rng);
SceneBundle cornell_box(Rng& rng);
SceneBundle cornell_smoke(Rng& rng, bool repair);
SceneBundle cornell_pedestal(Rng& rng, const std::string& obj_path, bool repair);
SceneBundle final_scene(Rng& rng, const ImageRGB8* earthmap, bool repair, bool moving_sphere);
// Not in the reference: n_tri-triangle height field + n_sph spheres + one light.
SceneBundle synthetic_mesh(Rng& rng, uint32_t n_triangles, uint32_t n_spheres);

uint32_t image_height_for(uint32_t width, float aspect_ratio);  // main.rs:12

}  // namespace rr

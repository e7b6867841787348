/*
 * rr_host.h — C bindings of the host-side scene construction library.
 *
 * The reference builds its scene with Rust constructors and hands the tree to
 * render_multi (src/main.rs:546-557).  These bindings expose the same constructors
 * to any host language through integer handles, and produce the rr_flat_scene that
 * rr_scene_upload (rr_hip.h) consumes.  Nothing here touches the GPU.
 *
 * All functions returning `int` handles return a value >= 0, or a negative RR_E_*
 * code with a message in rrh_last_error().  Handles are only valid for the builder
 * that returned them.
 */
#ifndef RR_HOST_H
#define RR_HOST_H

#include "rr_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rrh_builder rrh_builder;

/* seed drives every host-side random draw (BVH split axes, Perlin tables, random
 * scenes): the stand-in for thread_rng(), src/utils.rs:41-55.                     */
rrh_builder* rrh_builder_new(uint64_t seed);
void rrh_builder_free(rrh_builder* b);
const char* rrh_last_error(void);

/* Host random numbers with the reference's distributions (utils.rs:41-55). */
float rrh_random(rrh_builder* b);
float rrh_randrange(rrh_builder* b, float lo, float hi);
uint32_t rrh_randuint(rrh_builder* b, uint32_t lo, uint32_t hi_exclusive);

/* Textures (src/texture.rs:41-64) → texture handle. */
int rrh_tex_solid_color(rrh_builder* b, float r, float g, float bl);
int rrh_tex_checkered(rrh_builder* b, const float odd_rgb[3], const float even_rgb[3]);
int rrh_tex_noise(rrh_builder* b, float scale, uint32_t turb);
/* SceneColliders::load_image (collider.rs:747-750) for already-decoded RGB8 rows (top-down). */
int rrh_load_image_rgb8(rrh_builder* b, uint32_t width, uint32_t height, const uint8_t* rgb);
int rrh_load_image_ppm(rrh_builder* b, const char* path);

/* Materials (src/material.rs:58-82) → material handle. */
int rrh_mat_lambertian(rrh_builder* b, int texture);
int rrh_mat_glossy(rrh_builder* b, float r, float g, float bl, float fuzz);
int rrh_mat_dielectric(rrh_builder* b, float refraction_index);
int rrh_mat_isotropic(rrh_builder* b, float r, float g, float bl);
int rrh_mat_emissive(rrh_builder* b, float r, float g, float bl);

/* Geometry constructors (src/collider.rs:56-267) → geometry handle. */
int rrh_sphere(rrh_builder* b, const float center[3], float radius, int material);
int rrh_moving_sphere(rrh_builder* b, const float center0[3], const float center1[3], float time0, float time1,
                      float radius, int material);
int rrh_xyrect(rrh_builder* b, float x0, float x1, float y0, float y1, float k, int material);
int rrh_xzrect(rrh_builder* b, float x0, float x1, float z0, float z1, float k, int material);
int rrh_yzrect(rrh_builder* b, float y0, float y1, float z0, float z1, float k, int material);
int rrh_cuboid(rrh_builder* b, const float p0[3], const float p1[3], int material);
int rrh_constant_medium(rrh_builder* b, int boundary, float density, float r, float g, float bl);
int rrh_instance_translation(rrh_builder* b, int geometry, const float displacement[3]);
int rrh_instance_rotation(rrh_builder* b, int geometry, int axis /*0=X,1=Y,2=Z*/, float angle_degrees);
int rrh_bvh_node(rrh_builder* b, const int* geometries, uint32_t n, float time_0, float time_1);
/* Which tree rrh_bvh_node, rrh_load_obj and the presets build from here on:
 *   RRH_BVH_REFERENCE (default) the reference's: the median split of a random axis (collider.rs:173-212), by an index-range
 *                     sort (the same tree as the reference's Theta(n^2) cloning gives for the same axis draws);
 *   RRH_BVH_SAH       NOT the reference's, a labelled option: a surface-area-heuristic tree (16 bins), `left` the side of the
 *                     smaller coordinates along the split axis.  Under the reference's semantics the IMAGE depends on the
 *                     tree (collider.rs:555-561), so this is for RR_SEMANTICS_INTENDED / RR_SEMANTICS_BOOK, where a BVH
 *                     only accelerates: it measures what the reference's tree costs next to a conventional one.          */
#define RRH_BVH_REFERENCE 0u
#define RRH_BVH_SAH 1u
int rrh_set_bvh_builder(rrh_builder* b, uint32_t kind);
/* A BVH node with caller-chosen children, axis and box (no random draw): for hand-built test trees. */
int rrh_bvh_node_explicit(rrh_builder* b, int left, int right, uint32_t axis, int has_box, const float bmin[3],
                          const float bmax[3]);
int rrh_collider_list(rrh_builder* b, const int* geometries, uint32_t n);
int rrh_triangle(rrh_builder* b, const float p0[3], const float p1[3], const float p2[3], int material);
int rrh_load_obj(rrh_builder* b, const char* path, float scale, int material);
/* Not the reference's reader (collider.rs:227-267 takes "vn" lines for vertices, reads three corners of
 * every face and no negative indices): only "v" lines are vertices, negative indices count back from the
 * vertex read last, faces of more than three corners become fans.  Same triangles as rrh_load_obj on
 * files the reference reads correctly.                                                                */
int rrh_load_obj_hardened(rrh_builder* b, const char* path, float scale, int material);

/* bounding_box(time_0, time_1) of a geometry (collider.rs:698-713): returns 1 and fills
 * out[6] = min xyz, max xyz; 0 when the geometry has no box.                        */
int rrh_bounding_box(rrh_builder* b, int geometry, float time_0, float time_1, float out[6]);

/* Scene assembly: SceneColliders::add, the `lights` argument, Camera::new
 * (camera.rs:34-75) and the background colour.                                      */
int rrh_scene_add(rrh_builder* b, int geometry);
int rrh_set_lights(rrh_builder* b, int geometry);
int rrh_set_camera(rrh_builder* b, const float look_from[3], const float look_at[3], const float vup[3], float vfov,
                   float aperture, float focus_dist, float aspect_ratio, float shutter_open, float shutter_close);
int rrh_set_background(rrh_builder* b, float r, float g, float bl);

/* The scene functions of src/main.rs by name: "default_scene", "random_spheres",
 * "random_moving_spheres", "two_spheres", "two_perlin_spheres", "earth", "rect_light",
 * "cornell_box", "cornell_smoke", "cornell_pedestal", "final_scene", plus
 * "synthetic_mesh" (not in the reference).  Options:
 *   repair        — add the light to the scene and pass a bare XZRect as lights
 *                   (cornell_smoke / cornell_pedestal / final_scene);
 *   moving_sphere — final_scene only: add the motion-blurred sphere;
 *   obj_path      — cornell_pedestal mesh; image = atlas handle of the earth map
 *                   (earth / final_scene), from rrh_load_image_*;
 *   n_triangles, n_spheres — synthetic_mesh.
 * Fills `defaults` with the scene function's own image size, spp and depth.          */
typedef struct rrh_preset_options {
    int repair;
    int moving_sphere;
    const char* obj_path;
    int image;
    uint32_t n_triangles, n_spheres;
} rrh_preset_options;
typedef struct rrh_preset_defaults {
    uint32_t image_width, image_height, samples_per_pixel, max_depth;
    float aspect_ratio;
} rrh_preset_defaults;
int rrh_build_preset(rrh_builder* b, const char* name, const rrh_preset_options* opt, rrh_preset_defaults* defaults);

/* Flatten what has been assembled.  The returned view points into memory owned by the
 * builder and stays valid until the next rrh_flatten on it or rrh_builder_free.       */
int rrh_flatten(rrh_builder* b, rr_flat_scene* out);

#ifdef __cplusplus
}
#endif
#endif /* RR_HOST_H */

/*
 * rr_hip.h — C ABI of the MI355X (gfx950) path-tracing backend.
 *
 * This is the drop-in boundary for the reference's render seam
 *     pub fn render_multi(scene, lights, cam, background, max_depth,
 *                         samples_per_pixel, img_width, img_height)
 *         -> ImageBuffer<Rgb<u8>>                       (src/render.rs:128)
 * The reference has no FFI of its own; a host (Rust, C++ or Python) flattens
 * its scene tree into the plain arrays below, uploads them once, and calls
 * rr_render / rr_render_device instead of render_multi.
 *
 * Conventions: every function returns 0 on success or a negative RR_E_* code;
 * rr_last_error() returns a thread-local message for the last failure.  No
 * exceptions cross the boundary.  The caller owns every host pointer it passes
 * in; the library owns the device memory behind a context.  A context belongs
 * to ONE device and is used from one thread at a time (one process per GPU).
 * All arithmetic on the path is IEEE f32 (src/vec3.rs:8-12).
 */
#ifndef RR_HIP_H
#define RR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- errors */
#define RR_OK 0
#define RR_E_INVALID -1   /* bad argument / malformed scene               */
#define RR_E_NO_DEVICE -2 /* no usable gfx950 device                      */
#define RR_E_HIP -3       /* a HIP runtime call failed                    */
#define RR_E_NO_SCENE -4  /* render/primary-hit call before scene upload  */
#define RR_E_LIMIT -5     /* a size exceeds what the device layout holds  */

/* ------------------------------------------------- geometry references */
/* A "ref" names one record of the flattened Geometry tree
 * (enum GeometryType, src/collider.rs:34-47): tag in bits 31..28, index into
 * that tag's array in bits 27..0.                                          */
#define RR_TAG_SPHERE 0u        /* GeometryType::Sphere                    */
#define RR_TAG_MOVING_SPHERE 1u /* GeometryType::MovingSphere              */
#define RR_TAG_RECT 2u          /* XYRect / XZRect / YZRect (axis in record) */
#define RR_TAG_TRIANGLE 3u      /* GeometryType::Triangle                  */
#define RR_TAG_CUBOID 4u        /* GeometryType::Cuboid (6 rects)          */
#define RR_TAG_MEDIUM 5u        /* GeometryType::ConstantMedium            */
#define RR_TAG_INSTANCE 6u      /* TranslateInstance / YRotationInstance   */
#define RR_TAG_BVH 7u           /* GeometryType::BVHNode                   */
#define RR_TAG_LIST 8u          /* ColliderList and SceneColliders.objects */
#define RR_TAG_COUNT 9u
#define RR_REF(tag, index) ((((uint32_t)(tag)) << 28) | ((uint32_t)(index)))
#define RR_REF_TAG(ref) (((uint32_t)(ref)) >> 28)
#define RR_REF_INDEX(ref) (((uint32_t)(ref)) & 0x0FFFFFFFu)
#define RR_REF_NONE 0xFFFFFFFFu

/* Rect axis: which plane the rect lies in.
 *   RR_RECT_XY: a = x, b = y, k = z, normal (0,0,1)  (collider.rs:344-359)
 *   RR_RECT_XZ: a = x, b = z, k = y, normal (0,1,0)  (collider.rs:367-382)
 *   RR_RECT_YZ: a = y, b = z, k = x, normal (1,0,0)  (collider.rs:405-420)  */
#define RR_RECT_XY 0u
#define RR_RECT_XZ 1u
#define RR_RECT_YZ 2u

/* Instance kinds (collider.rs:480-528). */
#define RR_INST_TRANSLATE 0u
#define RR_INST_ROTATE_X 1u
#define RR_INST_ROTATE_Y 2u
#define RR_INST_ROTATE_Z 3u

/* Material kinds (enum MaterialType, src/material.rs:43-49). */
#define RR_MAT_LAMBERTIAN 0u
#define RR_MAT_GLOSSY 1u
#define RR_MAT_DIELECTRIC 2u
#define RR_MAT_ISOTROPIC 3u
#define RR_MAT_EMISSIVE 4u

/* Texture kinds (enum TextureType, src/texture.rs:28-33). */
#define RR_TEX_SOLID 0u
#define RR_TEX_CHECKERED 1u
#define RR_TEX_NOISE 2u
#define RR_TEX_IMAGE 3u

/* ------------------------------------------------------- scene records */
/* prim_id is the caller's stable primitive number (depth-first order of the
 * Geometry tree); the library only echoes it in rr_hit.                     */

typedef struct rr_sphere { /* collider.rs:35,278-301 */
    float center[3];
    float radius; /* may be negative (hollow-bubble trick, main.rs:44) */
    uint32_t material;
    uint32_t prim_id;
} rr_sphere;

typedef struct rr_moving_sphere { /* collider.rs:36,269-271,309-331 */
    float center0[3];
    float center1[3];
    float time0, time1;
    float radius;
    uint32_t material;
    uint32_t prim_id;
} rr_moving_sphere;

typedef struct rr_rect { /* collider.rs:37-39 */
    float a0, a1, b0, b1, k;
    uint32_t axis; /* RR_RECT_* */
    uint32_t material;
    uint32_t prim_id;
} rr_rect;

typedef struct rr_triangle { /* collider.rs:46,219-226,627-667 */
    float p0[3], p1[3], p2[3];
    float normal[3]; /* normalize(cross(p1-p0, p2-p0)), precomputed by the host */
    uint32_t material;
    uint32_t prim_id;
} rr_triangle;

typedef struct rr_cuboid { /* collider.rs:40,86-101,428-430 */
    uint32_t first_rect; /* 6 consecutive rr_rect records, constructor order */
} rr_cuboid;

typedef struct rr_medium { /* collider.rs:41,102-107,435-475 */
    uint32_t boundary;     /* ref */
    float neg_inv_density; /* -1/density */
    uint32_t material;     /* isotropic */
    uint32_t prim_id;
} rr_medium;

typedef struct rr_instance { /* collider.rs:42-43,480-528 */
    uint32_t kind;  /* RR_INST_* */
    uint32_t child; /* ref */
    float v[3];     /* translate: displacement; rotate: v[0]=sin, v[1]=cos */
} rr_instance;

typedef struct rr_bvh_node { /* collider.rs:44,536-581 */
    float bmin[3], bmax[3];
    uint32_t left, right; /* refs */
    uint32_t axis;        /* 0..2, indexes ray.direction for child order */
    uint32_t has_box;     /* 0: Option<AABB> is None → no box test */
} rr_bvh_node;

typedef struct rr_list { /* collider.rs:45,586-607,751-772 */
    uint32_t first_child; /* index into rr_flat_scene.child_refs */
    uint32_t n_children;
} rr_list;

typedef struct rr_material { /* material.rs:43-82 */
    uint32_t type;    /* RR_MAT_* */
    uint32_t texture; /* lambertian / isotropic / emissive */
    float color[3];   /* glossy albedo */
    float param;      /* glossy fuzz (already clamped to ≤1) or dielectric index */
} rr_material;

typedef struct rr_texture { /* texture.rs:28-103 */
    uint32_t type;   /* RR_TEX_* */
    float color0[3]; /* solid colour / checker odd colour  */
    float color1[3]; /* checker even colour                */
    uint32_t index;  /* noise: Perlin table; image: image  */
    float scale;     /* noise */
    uint32_t depth;  /* noise turbulence octaves */
} rr_texture;

#define RR_PERLIN_POINTS 256
typedef struct rr_perlin { /* perlin.rs:5-11 */
    float ranvec[RR_PERLIN_POINTS][3];
    uint8_t perm_x[RR_PERLIN_POINTS];
    uint8_t perm_y[RR_PERLIN_POINTS];
    uint8_t perm_z[RR_PERLIN_POINTS];
} rr_perlin;

typedef struct rr_image { /* texture.rs:9-23: decoded to RGB8 by the host */
    uint32_t width, height;
    uint64_t offset; /* byte offset of texel (0,0) in rr_flat_scene.image_bytes; rows top-down, 3 B/texel */
} rr_image;

typedef struct rr_light { /* the `lights: Geometry` argument, collider.rs:714-725 */
    uint32_t is_xzrect; /* 0: any other geometry → pdf 0, direction (1,0,0) */
    float x0, x1, z0, z1, k;
} rr_light;

typedef struct rr_camera { /* the fields get_ray reads, camera.rs:76-84 */
    float origin[3];
    float lower_left[3];
    float horizontal[3];
    float vertical[3];
    float u[3];
    float v[3];
    float lens_radius;
    float shutter_open, shutter_close;
} rr_camera;

typedef struct rr_flat_scene {
    uint32_t n_spheres;        const rr_sphere* spheres;
    uint32_t n_moving_spheres; const rr_moving_sphere* moving_spheres;
    uint32_t n_rects;          const rr_rect* rects;
    uint32_t n_triangles;      const rr_triangle* triangles;
    uint32_t n_cuboids;        const rr_cuboid* cuboids;
    uint32_t n_media;          const rr_medium* media;
    uint32_t n_instances;      const rr_instance* instances;
    uint32_t n_bvh_nodes;      const rr_bvh_node* bvh_nodes;
    uint32_t n_lists;          const rr_list* lists;
    uint32_t n_child_refs;     const uint32_t* child_refs;
    uint32_t n_materials;      const rr_material* materials;
    uint32_t n_textures;       const rr_texture* textures;
    uint32_t n_perlin;         const rr_perlin* perlin;
    uint32_t n_images;         const rr_image* images;
    uint64_t n_image_bytes;    const uint8_t* image_bytes;
    uint32_t root;             /* ref of the list that is SceneColliders.objects */
    rr_light light;
    rr_camera camera;
    float background[3];
    uint32_t semantics;        /* RR_SEMANTICS_* bits; 0 (a zeroed field) = the reference as it is at HEAD */
    uint32_t n_lights;         /* RR_SEMANTICS_BOOK_ESTIMATOR only: `lights` was a ColliderList of this many geometries (0: `light` above */
    const rr_light* lights;    /*   is the `lights` argument, as under every other semantics); is_xzrect = 0 for a member that is no XZRect */
} rr_flat_scene;

/* Which intersection semantics the scene is traversed with.
 *   RR_SEMANTICS_REFERENCE  the live code of src/collider.rs and src/aabb.rs, quirks included: what render_multi computes
 *                           at HEAD, and what every parity claim of this library refers to.
 *   RR_SEMANTICS_INTENDED   a labelled alternative, never the default: the three places where the live code departs from
 *                           what it was written to do are put right -
 *                             a BVH node whose first child misses goes on to its other child (the dead src/bvh.rs:28-44 has
 *                             it; collider.rs:555-561 tests `left` again and never reaches `right`);
 *                             AABB::intersect clips the ray on all three axes (aabb.rs:42 only clips on axes with a negative
 *                             direction component);
 *                             a triangle rejects hits outside [t_min, t_max] (collider.rs:627 ignores both).
 *                           Everything else (materials, pdfs, media, instances) is unchanged.  With these the BVH returns the
 *                           closest hit in range, as a linear list of the same primitives does.
 *   RR_SEMANTICS_BOOK_ESTIMATOR  a second labelled alternative, a bit of its own (never the default either): what the book the
 *                           reference follows ("Ray Tracing: The Rest of Your Life", the chapter that cleans up the PDF
 *                           management) does where the reference's text stops short -
 *                             a material that scatters and is not Lambertian is specular: ray_color follows the ray the
 *                             material itself scattered (material.rs:121-154: the fuzzed reflection of a Glossy, the reflected
 *                             or refracted ray of a Dielectric, the random direction of an Isotropic) and weights it by the
 *                             attenuation alone, instead of replacing it by a sample of the mixture PDF and weighting it by a
 *                             scattering pdf of zero (render.rs:27-34 with material.rs:170-175);
 *                             `lights` may be a ColliderList (rr_flat_scene.lights): its pdf value is the mean of its members'
 *                             and its sample is one member's, chosen uniformly (member floor(random() * n)); members that are
 *                             no XZRect have pdf 0 and sample (1,0,0), as collider.rs:714-725 has it for a single geometry.
 *                           Lambertian surfaces, emission (both faces, material.rs:176-181) and everything else are unchanged.
 *                           RR_SEMANTICS_BOOK = both bits: the images under output/ of the reference (metal, glass and smoke
 *                           lit, the whole BVH visible) are of this kind.                                                   */
#define RR_SEMANTICS_REFERENCE 0u
#define RR_SEMANTICS_INTENDED 1u
#define RR_SEMANTICS_BOOK_ESTIMATOR 2u
#define RR_SEMANTICS_BOOK 3u

/* ---------------------------------------------------------- rendering */
typedef struct rr_render_params {
    uint32_t width, height;
    uint32_t samples_per_pixel; /* as passed to render_multi                           */
    uint32_t n_slices;          /* K = the reference's n_threads (render.rs:129); each slice
                                   renders samples_per_pixel / K samples (render.rs:130)  */
    uint32_t slice_begin, slice_end; /* this call renders slices [begin,end) ⊂ [0,K)   */
    uint32_t max_depth;
    uint64_t seed;
    uint32_t sample_offset;     /* added to the per-pixel sample number in the RNG key:
                                   gives each GPU of a weak-scaling job its own samples   */
    uint32_t flags;             /* RR_FLAG_* */
    uint32_t row_begin, row_end; /* image rows [row_begin,row_end) this call renders, row 0 = top; row_end = 0
                                    stands for `height` (a zeroed pair is the whole image).  A band of rows is
                                    how several GPUs share the PIXELS of one image (slices share its SAMPLES,
                                    render.rs:130-140); every random stream is keyed by the pixel's position in
                                    the whole image, so the bands of an image are bit for bit the rows of the
                                    whole render.  Every output of the call has the band's shape,
                                    (row_end-row_begin) x width x 3, and bands are contiguous in the image.   */
} rr_render_params;

#define RR_FLAG_COUNTERS 1u /* count box tests / primitive tests (rr_counters); this build of the kernels makes
                               every BVH visit the reference makes, in its order (slower, same image) */
/* The library renders with one of three kernel organisations that produce the same bits.  The default is a
 * persistent kernel whose lanes follow single paths from camera ray to termination and leave the sample
 * values for an in-order accumulation.  Scenes that are all BVH (fewer than four things outside the BVH
 * subtrees) with more than 2 MiB of BVH and primitive records go through a wavefront pipeline instead: traversal and shading kernels over a compacted queue of
 * live paths for the camera rays, then a persistent kernel for the rest of every path; RR_FLAG_WAVEFRONT
 * forces that pipeline for any scene.  RR_FLAG_PERSISTENT selects the third, a persistent kernel whose
 * lanes own a (slice, pixel) at a time and add its samples themselves.                                  */
#define RR_FLAG_PERSISTENT 2u
#define RR_FLAG_WAVEFRONT 4u
/* After its first bounces the wavefront pipeline hands the paths still alive to a kernel that runs each of
 * them to its end; this flag keeps it to bounce-by-bounce launches throughout, which makes the call
 * synchronise with the stream after every bounce from the fourth on (for tests: the result is the same). */
#define RR_FLAG_NO_FINISH_KERNEL 8u
/* Forces the path-granular persistent kernel (the default of most scenes) for any scene. */
#define RR_FLAG_PATHS 16u
/* Count what the production kernels do: the same traversal as without flags (redundant BVH visits skipped,
 * subtrees culled), with the counters of rr_counters switched on.  Excludes RR_FLAG_COUNTERS, whose build of
 * the kernels makes (and counts) every visit of the reference instead.  Same image as either.           */
#define RR_FLAG_COUNT_PRODUCTION 32u
/* In scenes whose BVH nodes are all of the culling classes (rr_scene_info.wide_nodes) the traversal kernel of the wavefront
 * pipeline walks the BVHs with a node move that tests the boxes of both children of a node from one 64-byte record and
 * fetches only the children the ray can reach; this flag keeps it to the move that tests one box per visited node, which
 * every other kernel uses (for measurements and tests: same image).                                                     */
#define RR_FLAG_NARROW_NODES 64u
/* Dead paths.  Once a path's throughput is exactly zero in every channel -- after a metal, glass or smoke hit, whose scattering
 * pdf is 0 (material.rs:170-175), or a Lambertian hit whose sampled direction points below the surface (material.rs:111-114) --
 * everything it meets later is multiplied by zero.  The reference follows the path to its end all the same (render.rs:29-34)
 * and adds its 0; by default the kernels end it there and add the 0 at once (final_scene: 57 % of all rays are such, most of
 * them scattering on in the fog; 2.5 x the throughput).  The sample differs only where the rest of the walk would have met a
 * NaN or an infinity (0 * NaN = NaN: the 0/0 of render.rs:34 where a surface's mixture pdf is 0), which the reference turns into
 * one zeroed pixel of one slice (render.rs:102-119); with a light the library can sample that is a few samples in 10^8
 * (tests/test_dead_paths.py).  This flag makes the kernels walk dead paths to their end exactly as the reference
 * does.  The CPU oracle honours the same flag.  A throughput that is NaN in every channel ends a path as well, and that is
 * exact: NaN times whatever follows is NaN, which is what the ended path leaves as its sample.                              */
#define RR_FLAG_TRACE_DEAD_PATHS 128u
/* The path-granular kernel hands the pixels of a scene with BVH subtrees to its lanes in 8x8 tiles (a wave's camera rays then
 * walk the same part of the BVHs), and with two or more subtrees all samples of a pixel together; this flag keeps
 * it to 64 pixels of a row per wave, the order it uses for scenes without subtrees (for measurements and tests: the order
 * changes no result).                                                                                                        */
#define RR_FLAG_ROW_ORDER 256u

typedef struct rr_counters {
    uint64_t samples;     /* camera rays started                                         */
    uint64_t rays;        /* ray_color invocations with depth > 0 (one scene.intersect each) */
    uint64_t box_tests;   /* AABB::intersect calls                (RR_FLAG_COUNTERS)      */
    uint64_t prim_tests;  /* all primitive intersector calls       (RR_FLAG_COUNTERS)     */
    uint64_t inst_visits; /* instance + medium visits              (RR_FLAG_COUNTERS)     */
    uint64_t sphere_tests; /* of prim_tests: spheres and moving spheres                   */
    uint64_t rect_tests;   /* of prim_tests: rects (cuboid sides included)                */
    uint64_t tri_tests;    /* of prim_tests: triangles                                    */
    uint64_t machine_lane_steps; /* steps of the per-lane BVH traversal machine, summed over lanes: a lane with a
                                    subtree to walk takes one step per wave step           (both counting flags) */
    uint64_t machine_wave_steps; /* steps the waves took while any of their lanes walked a subtree; lane_steps /
                                    (64 * wave_steps) is the fraction of lanes at work inside the BVHs          */
} rr_counters;

typedef struct rr_hit {
    int32_t prim_id;   /* -1 = miss */
    uint32_t material;
    float t;
    float point[3];
    float normal[3];
    float u, v;
    uint32_t front_face;
} rr_hit;

typedef struct rr_scene_info {
    uint32_t stack_words;    /* per-lane traversal stack the scene needs */
    uint32_t max_inst_depth; /* deepest instance nesting                 */
    uint64_t device_bytes;   /* bytes of scene data resident in HBM      */
    uint32_t n_prims;        /* spheres + moving spheres + rects + triangles */
    uint32_t wavefront;      /* 1: rendered by the wavefront pipeline unless a flag says otherwise (nearly all of the scene is inside BVHs, and its BVH and primitive records exceed 2 MiB) */
    uint32_t bvh_nodes_by_class[4]; /* how the production kernels treat the BVH nodes: [0] no box, [1] exactly as the
                                       reference (a constant medium below, or a box that does not hold what is below),
                                       [2] redundant visits skipped and subtrees culled against [t_min,t_max],
                                       [3] the same against [0,inf) (triangles below)                              */
    uint32_t subtree_ops[2]; /* BVH subtrees the scene program hands to the per-lane machine: [0] to the moves for BVH nodes over
                                spheres, rects, triangles and cuboids whose root box holds every box below, [1] to the general machine */
    uint32_t wide_nodes;     /* 1: the wavefront pipeline walks the BVHs with the two-boxes-per-node move (every BVH node is of class
                                [2] or [3] and has only BVH nodes, spheres, rects, triangles and cuboids below; reference semantics) */
    uint64_t workspace_bytes; /* device memory the context's render calls have allocated so far (grow-only workspace and work
                                counter of rr_render_device / rr_render); stays 0 on a context that only ever rendered through
                                rr_render_device_noalloc                                                                    */
    uint32_t specialized;    /* 1: rr_scene_specialize has installed a kernel compiled for this scene's program            */
    float specialize_ms;     /* what that took (compile, or the look-up of an earlier compile of the same program, and load) */
} rr_scene_info;

typedef struct rr_context rr_context;

/* Number of visible HIP devices (0 on a host without a GPU; not an error). */
int rr_device_count(int* count);

/* Create / destroy the per-device context.  Fails with RR_E_NO_DEVICE when
 * `device` is not a visible gfx950 GPU: there is no CPU fallback.
 * Measurement and test switches: rr_context_create reads four environment variables, once, and the context keeps what it
 * read (a variable set or changed later does nothing).  None of them changes an image; a production host leaves them unset.
 *   RR_NO_FUSE              the scene program gets one op per transform and per step of a medium instead of the fused ops
 *   RR_PAD_OPS=n            n ops that do nothing are appended to the scene program (prices an op's fetch and dispatch)
 *   RR_DUMP_PROGRAM         rr_scene_upload prints the scene program to stderr
 *   RR_WF_WORKSPACE_BYTES=n caps the workspace of a batch of samples at n bytes (forces many small batches)              */
int rr_context_create(int device, rr_context** out);
int rr_context_destroy(rr_context* ctx);

/* Validate (every ref in range, tree-shaped, bounded depth), pack and copy the
 * scene into HBM.  Replaces the scene, lights, cam and background arguments
 * of render_multi (render.rs:128).  A later call replaces the scene.        */
int rr_scene_upload(rr_context* ctx, const rr_flat_scene* scene);
int rr_scene_info_get(rr_context* ctx, rr_scene_info* out);

/* Compile the path kernel for THIS scene's program.  No counterpart in the reference (its scene is Rust data walked by
 * compiled match arms, collider.rs:751-772); here the top level of the scene is a short program of ops (one per sphere, rect,
 * cuboid, transform, medium or BVH subtree at the root) that the kernels of the library interpret: fetched and dispatched by
 * the scalar unit for every ray.  This call hands the same device source, with the uploaded scene's ops as compile-time
 * constants and the op loop unrolled, to the runtime compiler (libhiprtc.so, opened at the first call; same options as the
 * library's own build, so the arithmetic contract -- one IEEE f32 operation per source operation -- is the same) and installs the
 * result in place of the interpreter for rr_render_device / _noalloc / rr_render calls without counting flags.  Images are bit
 * for bit those of the interpreter.  Measured: Cornell box (10 ops, no BVH) +16 % with the runtime compiler a PyTorch process
 * brings along (ROCm 7.0's) and +24 % with the compiler of the library's own build (ROCm 7.2's, which is what libhiprtc.so is in
 * a process without PyTorch); scenes with BVH subtrees, whose walk dominates, lose 1-13 % (final_scene, Suzanne, random
 * spheres); about 1-3 s of compile per new program, remembered per process.
 *   RR_SPECIALIZE_AUTO    only where it is expected to pay: programs without BVH subtrees (returns RR_OK and leaves
 *                         rr_scene_info.specialized at 0 otherwise)
 *   RR_SPECIALIZE_ALWAYS  whatever the program
 *   RR_SPECIALIZE_OFF     back to the interpreter
 * A later rr_scene_upload drops it.  Fails with RR_E_NO_DEVICE when libhiprtc.so cannot be opened and RR_E_HIP with the
 * compiler's log when the compile fails; the interpreter stays in place then.  Not for callers that cannot allocate or wait:
 * it compiles and loads a code object. */
#define RR_SPECIALIZE_AUTO   0u
#define RR_SPECIALIZE_ALWAYS 1u
#define RR_SPECIALIZE_OFF    2u
int rr_scene_specialize(rr_context* ctx, uint32_t mode);
/* The compiler options of rr_scene_specialize, space separated (a test holds them against the library's own build), and a
 * check that needs no GPU: a two-op program is put through the runtime compiler (ms: what that took).  No counterpart in the
 * reference. */
const char* rr_specialize_options(void);
int rr_specialize_selftest(float* ms);

/* Closest-hit records for caller-supplied rays, exactly as
 * SceneColliders::intersect(r, t_min, t_max) returns them (collider.rs:751).
 * rays: n × 7 f32 (origin xyz, direction xyz, time), host memory.
 * Constant media draw from the counter RNG keyed by (seed, ray index).      */
int rr_primary_hits(rr_context* ctx, uint32_t n, const float* rays, float t_min, float t_max,
                    uint64_t seed, rr_hit* out_hits);

/* Texture::get_color(u, v, p) (texture.rs:96-103) on the device for caller-supplied arguments, or, of a noise
 * texture, Perlin::turb(p, depth) (perlin.rs:87-98) or Perlin::noise(p) (perlin.rs:47-86) alone: the functions the
 * shading kernels evaluate, for parity checks.  textures: n indices into rr_flat_scene.textures; uvp: n x 5 f32
 * (u, v, p.x, p.y, p.z); out_rgb: n x 3 f32 (turbulence and noise in all three).  Host memory.            */
#define RR_EVAL_COLOR 0u
#define RR_EVAL_TURB 1u
#define RR_EVAL_NOISE 2u
int rr_eval_textures(rr_context* ctx, uint32_t n, const uint32_t* textures, const float* uvp, uint32_t what,
                     float* out_rgb);

/* Render slices [slice_begin, slice_end) of the image; replaces the parallel
 * loop of render_multi (render.rs:135-151) and render_worker (render.rs:89-124).
 * All output pointers are DEVICE pointers owned by the caller, any of the
 * optional ones may be NULL; H below is the number of rows the call renders
 * (row_end-row_begin, the image height unless a band of rows is asked for):
 *   d_slice_sums : (slice_end-slice_begin) × H × W × 3 f32, the per-slice pixel sums
 *                  (pixel_color of render.rs:93-100, NaN kept), row 0 = top.  Required.
 *   d_resolved   : H × W × 3 f32, Σ over the rendered slices of the per-slice
 *                  tonemapped value of render.rs:102-119 (what the mutex sum
 *                  of render.rs:141-150 adds up).                  Optional.
 *   d_linear     : H × W × 3 f32, Σ over the rendered slices of the NaN-scrubbed
 *                  pixel sums (linear radiance × samples).          Optional.
 *   d_counters   : one rr_counters, ADDED to (caller zeroes it).   Optional.
 * Work is enqueued on hip_stream (a hipStream_t, NULL = default stream) and the
 * call returns without synchronising, with one exception: the first call, and any call that
 * needs a larger workspace than every call before it, (re)allocates the context's grow-only
 * workspace, which waits for the device (hipDeviceSynchronize + hipFree + hipMalloc).
 * ONE render per context may be in flight at a time: the workspace, the work counters, the
 * library's second stream and the timing events belong to the context, so a second
 * rr_render_device on another stream must be ordered after the first (same stream, or an
 * event); independent concurrent renders take one context each.  Part of the wavefront pipeline's work runs
 * on a stream of the library's own, ordered against hip_stream by events, so
 * whatever the caller enqueues on hip_stream afterwards sees the result.        */
int rr_render_device(rr_context* ctx, const rr_render_params* params, void* d_slice_sums,
                     void* d_resolved, void* d_linear, void* d_counters, void* hip_stream);

/* rr_render_device for callers that cannot allocate device memory.  It has no counterpart in the
 * reference: it is for hosts that call from inside a HIP graph capture, from a tight serving loop or
 * under a harness that forbids hipMalloc.  Same arguments, same outputs and the same bits as
 * rr_render_device, with two differences:
 *   - it calls no hipMalloc, hipFree or other allocating routine: the kernels' workspace (the counter
 *     the waves claim paths from, and 16 bytes of sample value per path in flight) is a static device
 *     array of 64 MiB + 256 bytes inside the library, which every GPU the library runs on holds
 *     until the process ends;
 *   - it always renders with the path-granular persistent kernel (RR_FLAG_PERSISTENT, RR_FLAG_WAVEFRONT and
 *     RR_FLAG_NO_FINISH_KERNEL are rejected with RR_E_INVALID), a group of slices and a batch of samples at a
 *     time, at most RR_NOALLOC_MAX_PATHS paths in flight.
 * Limit: width x (rows of the call) <= RR_NOALLOC_MAX_PATHS, else RR_E_LIMIT (render the image in bands
 * of rows, row_begin/row_end).  All calls on one GPU share that GPU's array: two calls that overlap in
 * time on one GPU (different streams, different contexts, different threads) race and corrupt each
 * other's images; calls on one stream, or ordered by events, are fine.                              */
#define RR_NOALLOC_MAX_PATHS 4194304u
int rr_render_device_noalloc(rr_context* ctx, const rr_render_params* params, void* d_slice_sums,
                             void* d_resolved, void* d_linear, void* d_counters, void* hip_stream);

/* Convenience wrapper for hosts without their own device buffers: allocates,
 * renders all of [slice_begin,slice_end), synchronises and copies to HOST
 * buffers.  out_rgb8 (H×W×3 u8) is the saturating `as u8` cast of render.rs:156-159
 * and is only the final image when the call covers all K slices; out_slice_sums
 * has the layout of d_slice_sums.  Any output may be NULL.                                                             */
int rr_render(rr_context* ctx, const rr_render_params* params, uint8_t* out_rgb8,
              float* out_resolved, float* out_linear, float* out_slice_sums,
              rr_counters* out_counters);

/* f32 → u8 cast of the summed slice images (render.rs:156-159) on the device:
 * d_rgb8[i] = saturating truncation of d_resolved[i], NaN → 0.              */
int rr_cast_u8_device(rr_context* ctx, const void* d_resolved, void* d_rgb8, uint64_t n,
                      void* hip_stream);

/* Milliseconds the dominant (path-tracing) kernel of the last rr_render_device /
 * rr_render call took, measured with HIP events on the stream it ran on.
 * Synchronises with that kernel.                                            */
int rr_last_kernel_ms(rr_context* ctx, float* ms);

/* ------------------------------------------------- several GPUs, one host thread
 * One render_multi job over the GPUs of a node from a host that is ONE process (a Rust or C program; the one-process-per-GPU
 * form over torch.distributed / RCCL is rust_raytracer_amd/multi.py): the job is cut into bands of image rows and groups of
 * sample slices (render.rs:130-140 shares out samples; rows share out pixels), every device renders its share on a stream of
 * its own, the images of a band's slice groups are added on the band's first device (peer copies over xGMI where the devices
 * can reach each other, then one add kernel, or one ncclReduce over RCCL: the mutex sum of render.rs:141-150; see
 * rr_multi_set_reduce), and every band is cast and copied to its rows of the host image.  The scene is replicated.  split:
 *   RR_SPLIT_SLICES  device g renders slices [g*K/N, (g+1)*K/N) of the whole image;
 *   RR_SPLIT_ROWS    device b renders rows [b*H/N, (b+1)*H/N) with every slice: bit for bit the single-device image;
 *   RR_SPLIT_BOTH    B bands x G groups, B the smallest divisor of N with B*B >= N; device r is band r / G, group r % G.
 * `job` describes the whole image (all slices, all rows).  A device may be listed more than once (each entry gets a context
 * of its own), which is how a one-GPU machine tests the arithmetic.  Outputs are host buffers, any may be NULL; the call
 * returns when they are filled.                                                                                          */
#define RR_SPLIT_SLICES 0u
#define RR_SPLIT_ROWS 1u
#define RR_SPLIT_BOTH 2u
typedef struct rr_multi rr_multi;
int rr_multi_create(int n_devices, const int* devices, rr_multi** out);
int rr_multi_destroy(rr_multi* m);
int rr_multi_scene_upload(rr_multi* m, const rr_flat_scene* scene);
int rr_multi_scene_specialize(rr_multi* m, uint32_t mode); /* rr_scene_specialize on every device (the program is compiled once) */
int rr_multi_render(rr_multi* m, const rr_render_params* job, uint32_t split, uint8_t* out_rgb8,
                    float* out_resolved, float* out_linear, rr_counters* out_counters);

/* How rr_multi_render adds the images of a band's slice groups onto the band's first device (the mutex sum of
 * render.rs:141-150):
 *   RR_REDUCE_AUTO  (default) one ncclReduce(sum, f32) per band over RCCL / xGMI when there is more than one slice group, the
 *                   band's devices are distinct GPUs and librccl.so can be opened (it is opened with dlopen at the first
 *                   such call, so a one-GPU host needs no RCCL); the peer-copy sum otherwise;
 *   RR_REDUCE_PEER  hipMemcpyPeerAsync of each group's images to the band's first device and an add kernel there, in
 *                   group order: the association of the f32 sum is that of a loop over the groups;
 *   RR_REDUCE_RCCL  RCCL or fail: rr_multi_set_reduce fails with RR_E_NO_DEVICE when librccl.so cannot be opened, and a
 *                   later rr_multi_render fails when a band lists a device twice or a communicator cannot be made.  Every
 *                   band goes through ncclReduce, also a band of ONE device (a copy in place).
 * RCCL chooses the association of the sum (ring or tree order): with two slice groups the image is bit for bit the
 * peer-copy one (a + b), with more it agrees up to f32 rounding of the sum.
 * rr_multi_reduce_used reports which of RR_REDUCE_PEER / RR_REDUCE_RCCL the last rr_multi_render used.               */
#define RR_REDUCE_AUTO 0u
#define RR_REDUCE_PEER 1u
#define RR_REDUCE_RCCL 2u
int rr_multi_set_reduce(rr_multi* m, uint32_t mode);
int rr_multi_reduce_used(rr_multi* m, uint32_t* mode);

const char* rr_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RR_HIP_H */

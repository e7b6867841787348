/*
 * render_scene.c — the whole path from plain C: build one of the scene functions of
 * src/main.rs with the host library, flatten it, upload it, render it on the GPU with
 * the arguments render_multi takes (src/render.rs:128), write a binary PPM.
 *
 *   render_scene <scene> <out.ppm> [width height spp depth slices seed]
 *
 * <scene> is a name rrh_build_preset knows ("cornell_box", "random_spheres", ...).
 * Omitted numbers come from the scene function's own defaults; `slices` stands for the
 * reference's num_cpus::get() (render.rs:129), default 8.
 *
 * Exit status: 0 on success, 1 on a usage error, 2 when a library call fails (the
 * message of rr_last_error / rrh_last_error goes to stderr).  There is no CPU
 * fallback: without a gfx950 device rr_context_create fails and so does this program.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rr_hip.h"
#include "rr_host.h"

static int fail_host(const char* what) {
    fprintf(stderr, "render_scene: %s: %s\n", what, rrh_last_error());
    return 2;
}
static int fail_dev(const char* what) {
    fprintf(stderr, "render_scene: %s: %s\n", what, rr_last_error());
    return 2;
}

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s <scene> <out.ppm> [width height spp depth slices seed]\n", argv[0]);
        return 1;
    }
    const char* scene = argv[1];
    const char* out_path = argv[2];
    unsigned long long seed = argc > 8 ? strtoull(argv[8], NULL, 10) : 1ull;

    rrh_builder* b = rrh_builder_new(seed);
    if (!b) return fail_host("rrh_builder_new");
    rrh_preset_options opt;
    memset(&opt, 0, sizeof opt);
    opt.repair = 1; /* the scenes whose light is missing from the scene list get it, as cornell_box has it */
    opt.image = -1;
    rrh_preset_defaults def;
    if (rrh_build_preset(b, scene, &opt, &def) < 0) return fail_host("rrh_build_preset");
    rr_flat_scene flat;
    if (rrh_flatten(b, &flat) < 0) return fail_host("rrh_flatten");

    rr_render_params p;
    memset(&p, 0, sizeof p);
    p.width = argc > 3 ? (uint32_t)strtoul(argv[3], NULL, 10) : def.image_width;
    p.height = argc > 4 ? (uint32_t)strtoul(argv[4], NULL, 10) : def.image_height;
    p.samples_per_pixel = argc > 5 ? (uint32_t)strtoul(argv[5], NULL, 10) : def.samples_per_pixel;
    p.max_depth = argc > 6 ? (uint32_t)strtoul(argv[6], NULL, 10) : def.max_depth;
    p.n_slices = argc > 7 ? (uint32_t)strtoul(argv[7], NULL, 10) : 8u;
    p.slice_begin = 0;
    p.slice_end = p.n_slices;
    p.seed = seed;
    if (p.width == 0 || p.height == 0) {
        fprintf(stderr, "render_scene: empty image\n");
        return 1;
    }

    rr_context* ctx = NULL;
    if (rr_context_create(0, &ctx) != RR_OK) return fail_dev("rr_context_create");
    if (rr_scene_upload(ctx, &flat) != RR_OK) return fail_dev("rr_scene_upload");
    /* Optional: a kernel compiled for this scene's program where that pays (programs without BVH subtrees).  Same image;
     * without a runtime compiler on the host the interpreter renders, so a failure here is only reported. */
    if (rr_scene_specialize(ctx, RR_SPECIALIZE_AUTO) != RR_OK) fprintf(stderr, "render_scene: %s (rendering with the interpreter)\n", rr_last_error());
    rr_scene_info info;
    if (rr_scene_info_get(ctx, &info) != RR_OK) return fail_dev("rr_scene_info_get");
    size_t n = (size_t)p.width * p.height * 3u;
    uint8_t* rgb = (uint8_t*)malloc(n);
    if (!rgb) {
        fprintf(stderr, "render_scene: out of memory\n");
        return 2;
    }
    rr_counters counters;
    if (rr_render(ctx, &p, rgb, NULL, NULL, NULL, &counters) != RR_OK) return fail_dev("rr_render");
    float ms = 0.0f;
    rr_last_kernel_ms(ctx, &ms);

    FILE* f = fopen(out_path, "wb");
    if (!f) {
        perror(out_path);
        return 2;
    }
    fprintf(f, "P6\n%u %u\n255\n", p.width, p.height); /* row 0 = top, as the reference's ImageBuffer (render.rs:119) */
    if (fwrite(rgb, 1, n, f) != n || fclose(f) != 0) {
        perror(out_path);
        return 2;
    }
    fprintf(stderr, "render_scene: %s %ux%u, %llu samples, %llu rays, %.3f ms on the GPU%s\n", scene, p.width, p.height,
            (unsigned long long)counters.samples, (unsigned long long)counters.rays, ms,
            info.specialized ? " (kernel compiled for this scene's program)" : "");
    free(rgb);
    rr_context_destroy(ctx);
    rrh_builder_free(b);
    return 0;
}

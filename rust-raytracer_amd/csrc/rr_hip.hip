// librr_hip.so — kernels, scene upload and the C ABI of include/rr_hip.h for gfx950.
//
// One persistent path-tracing kernel does the work of render_worker + ray_color
// (src/render.rs:15-41, 89-124): every lane owns one (slice, pixel) work item at a
// time, runs that pixel's samples one after the other (so the f32 pixel sum is added
// in the reference's order, with no atomics), and starts its next camera ray the
// moment a path ends, so that all 64 lanes of a wave stay busy whatever the path
// lengths are.  Lanes of a wave hold the 64 pixels of one 8x8 tile.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // declarations only: librccl.so is opened at run time, the first time a sum over devices asks for it

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "rr_device.h"

using namespace rrdev;

// The device sources as text inside the library, for rr_scene_specialize: it hands them, unchanged, to the runtime compiler
// together with the ops of one scene.  (The assembler reads the files where the build's -I options point.)
#if !defined(__HIP_DEVICE_COMPILE__)
__asm__(".pushsection .rodata\n"
        ".global rr_src_device_h\n.hidden rr_src_device_h\nrr_src_device_h:\n.incbin \"rr_device.h\"\n.byte 0\n"
        ".global rr_src_wavefront_h\n.hidden rr_src_wavefront_h\nrr_src_wavefront_h:\n.incbin \"rr_wavefront.h\"\n.byte 0\n"
        ".global rr_src_api_h\n.hidden rr_src_api_h\nrr_src_api_h:\n.incbin \"rr_hip.h\"\n.byte 0\n"
        ".popsection\n");
#endif
extern "C" {
extern const char rr_src_device_h[] __attribute__((visibility("hidden")));
extern const char rr_src_wavefront_h[] __attribute__((visibility("hidden")));
extern const char rr_src_api_h[] __attribute__((visibility("hidden")));
}

// ================================================================= kernels
struct RenderArgs {
    uint32_t width, height; // of the whole image
    uint32_t row0, rows;    // the band of image rows [row0, row0 + rows) this launch renders
    uint32_t spt;           // samples per slice and pixel (render.rs:130)
    uint32_t tiles_x;       // 8x8 pixel tiles per row
    uint32_t padded_pixels; // tiles * 64
    uint64_t n_items;       // slices in this launch * padded_pixels
    uint32_t slice_first;   // absolute index of the launch's first slice
    uint32_t slice_out;     // index of that slice inside the output buffer
    uint32_t max_depth;
    uint32_t sample_offset;
    uint32_t trace_dead;    // RR_FLAG_TRACE_DEAD_PATHS: paths go on after their throughput has become zero (see dead_path)
    uint64_t seed;
    unsigned long long* next_item;  // the first work item no wave has claimed yet (zero at launch)
};

#ifndef RR_RENDER_CHUNK
#define RR_RENDER_CHUNK 64u  // work items a wave claims at a time
#endif
template <int COUNT, bool VEC>
__global__ void __launch_bounds__(256, RR_RENDER_WAVES_PER_SIMD) k_render(DevScene sc, RenderArgs a, float* __restrict__ slice_sums,
                                                unsigned long long* counters) {
    extern __shared__ uint32_t lds_stack[];
    uint32_t* stk = lds_stack + threadIdx.x;
    const uint32_t stride = blockDim.x;
    // Work items (slice, pixel) are handed out dynamically, a chunk per wave and claim: pixels differ in how long their
    // paths are, and with a fixed share per lane the waves over cheap regions would sit idle at the end (measured: a
    // fifth of the wave slots of the Cornell box render).  A chunk is 64 consecutive items, one 8x8 tile of one slice.
    uint64_t chunk_pos = 0, chunk_end = 0;  // wave-uniform: the unclaimed rest of this wave's chunk

    Counts cnt = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    bool have_item = false, path_active = false;
    uint32_t px = 0, py = 0, sl = 0, s = 0;
    f3 acc = mk3(0.f, 0.f, 0.f);
    f3 ro = mk3(0.f, 0.f, 0.f), rd = mk3(0.f, 0.f, 0.f), thr = mk3(1.f, 1.f, 1.f);
    float time = 0.f;
    uint32_t depth_left = 0, bounce = 0, sample = 0, pixel_key = 0;

    // The loop below is uniform per wave (its only exit is a wave vote); lanes differ through
    // flags.  That keeps the scene-program interpreter inside traverse() in uniform control
    // flow, where its program counter, op words and branches stay on the scalar unit.
    bool lane_done = false;
    for (;;) {
        // ---- lanes whose path ended: pixel_color of render.rs:93-100 is complete for this slice once the item's samples are through
        auto finish_item = [&]() {
            float* dst = slice_sums + ((static_cast<uint64_t>(a.slice_out + sl) * a.rows + py) * a.width + px) * 3u;
            dst[0] = acc.x;
            dst[1] = acc.y;
            dst[2] = acc.z;
            have_item = false;
        };
        if (!path_active && have_item && s == a.spt) finish_item();
        // ---- lanes without an item take the next unclaimed ones
        for (;;) {
            const bool need = !path_active && !have_item && !lane_done;
            const uint64_t mask = RR_BALLOT(need);
            if (mask == 0) break;
            if (chunk_pos == chunk_end) {
                unsigned long long base = 0;
                if ((threadIdx.x & 63u) == 0) base = atomicAdd(a.next_item, static_cast<unsigned long long>(RR_RENDER_CHUNK));
                base = (static_cast<unsigned long long>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(base >> 32))) << 32) |
                       __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(base));
                chunk_pos = min(static_cast<uint64_t>(base), a.n_items);
                chunk_end = min(static_cast<uint64_t>(base) + RR_RENDER_CHUNK, a.n_items);
            }
            if (chunk_pos == chunk_end) {  // the counter only grows: nothing is left for anyone
                if (need) lane_done = true;
                break;
            }
            const uint32_t take = static_cast<uint32_t>(min(static_cast<uint64_t>(__popcll(mask)), chunk_end - chunk_pos));
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
            if (need && rank < take) {
                const uint64_t q = chunk_pos + rank;
                const uint32_t r = static_cast<uint32_t>(q % a.padded_pixels);
                sl = static_cast<uint32_t>(q / a.padded_pixels);
                const uint32_t tile = r >> 6, within = r & 63u;
                px = (tile % a.tiles_x) * 8u + (within & 7u);
                py = (tile / a.tiles_x) * 8u + (within >> 3);
                if (px < a.width && py < a.rows) {  // (items of the padding around the band are dropped)
                    have_item = true;
                    s = 0;
                    acc = mk3(0.f, 0.f, 0.f);
                    pixel_key = (a.row0 + py) * a.width + px;
                    if (a.spt == 0) finish_item();
                }
            }
            chunk_pos += take;
        }
        // ---- lanes with an item and no path take their next sample
        if (!path_active && have_item) {
            sample = a.sample_offset + (a.slice_first + sl) * a.spt + s;
            Rng g;
            g.start(a.seed, pixel_key, sample, CAMERA_BOUNCE);
            camera_ray(sc, a.width, a.height, px, a.height - 1u - (a.row0 + py), g, ro, rd, time);
            thr = mk3(1.f, 1.f, 1.f);
            depth_left = a.max_depth;
            bounce = 0;
            path_active = true;
            cnt.samples++;
        }
        if (!RR_ANY(path_active)) break;
        // ---- one ray_color call (render.rs:15-41) for every lane that has a path
        const bool run = path_active && depth_left != 0;
        if (run) cnt.rays++;
        Rng g;
        g.start(a.seed, pixel_key, sample, bounce);
        float t;
        uint32_t ref;
#ifdef RR_ABLATE_TRAVERSE  // timing-only build: wrong images, used to price the sections of a bounce
        t = ro.x + rd.y;
        ref = NONE;
#else
        traverse<COUNT, VEC>(sc, ro, rd, time, 0.001f, kF32Max, run, g, stk, stride, t, ref, cnt);
#endif
        f3 terminal = mk3(0.f, 0.f, 0.f);
        bool go_on = false;
        if (run) {
            Bounce b;
#ifdef RR_ABLATE_SHADE
            go_on = false;
            b.value = mk3(t, __uint_as_float(ref), g.random());
#else
            go_on = shade(sc, ro, rd, time, t, ref, ref == NONE ? HitMeta{0u, 0u} : hit_meta(sc, ref), g, b);
#endif
            if (go_on) {
                ro = b.o;
                rd = b.d;
                thr = thr * b.value;
                depth_left -= 1;
                bounce += 1;
                if (a.trace_dead == 0u && dead_path(thr)) go_on = false;  // (terminal is 0: the sample's value is 0)
            } else {
                terminal = b.value;
            }
        }
        if (path_active && !go_on) {
            acc = acc + thr * terminal;  // render.rs:99
            s += 1;
            path_active = false;
        }
    }
    flush_counts(cnt, counters);
}

__global__ void __launch_bounds__(256) k_primary_hits(DevScene sc, uint32_t n, const float* __restrict__ rays, float t_min,
                                                      float t_max, uint64_t seed, float* __restrict__ out_f,
                                                      uint32_t* __restrict__ out_ref) {
    extern __shared__ uint32_t lds_stack[];
    uint32_t* stk = lds_stack + threadIdx.x;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = rays + static_cast<uint64_t>(i) * 7u;
    const f3 o = mk3(r[0], r[1], r[2]), d = mk3(r[3], r[4], r[5]);
    const float time = r[6];
    Rng g;
    g.start(seed, i, 0, 0);
    Counts cnt = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    float t;
    uint32_t ref;
    traverse<COUNT_OFF>(sc, o, d, time, t_min, t_max, true, g, stk, blockDim.x, t, ref, cnt);
    float* of = out_f + static_cast<uint64_t>(i) * 12u;
    out_ref[i] = ref;
    if (ref == NONE) {
        for (int k = 0; k < 12; ++k) of[k] = 0.f;
        return;
    }
    const HitRec h = reconstruct_hit(sc, o, d, time, t, ref, hit_meta(sc, ref), true);
    of[0] = h.t;
    of[1] = h.p.x, of[2] = h.p.y, of[3] = h.p.z;
    of[4] = h.n.x, of[5] = h.n.y, of[6] = h.n.z;
    of[7] = h.u, of[8] = h.v;
    of[9] = __uint_as_float(h.front ? 1u : 0u);
    of[10] = __uint_as_float(h.mat);
    of[11] = 0.f;
}

// Texture::get_color (texture.rs:96-103), Perlin::turb (perlin.rs:87-98) or Perlin::noise (perlin.rs:47-86) for caller-supplied
// arguments: what the shading kernels evaluate, on its own.
__global__ void __launch_bounds__(256) k_eval_textures(DevScene sc, uint32_t n, const uint32_t* __restrict__ tex, const float* __restrict__ uvp,
                                                        uint32_t what, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* q = uvp + static_cast<uint64_t>(i) * 5u;
    const f3 p = mk3(q[2], q[3], q[4]);
    f3 c;
    if (what == RR_EVAL_COLOR) {
        c = texture_color(sc, tex[i], q[0], q[1], p);
    } else {
        const DevTexture& t = sc.k->textures[tex[i]];
        const float v = what == RR_EVAL_TURB ? perlin_turb(sc, t.index, p, t.depth) : perlin_noise(sc, t.index, p);
        c = mk3(v, v, v);
    }
    out[3u * i] = c.x, out[3u * i + 1u] = c.y, out[3u * i + 2u] = c.z;
}

// render.rs:102-119 per slice, summed over the slices in slice order (render.rs:141-150).
__global__ void __launch_bounds__(256) k_resolve(const float* __restrict__ slice_sums, uint32_t n_slices_here, uint64_t n_elems,
                                                 float scale, float n_threads, float* __restrict__ resolved,
                                                 float* __restrict__ linear) {
    const uint64_t e = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (e >= n_elems) return;
    float acc_res = 0.f, acc_lin = 0.f;
    for (uint32_t k = 0; k < n_slices_here; ++k) {
        float c = slice_sums[static_cast<uint64_t>(k) * n_elems + e];
        if (c != c) c = 0.f;
        acc_lin += c;
        const float g = sqrtf(scale * c);
        acc_res += 256.0f * clampf(g / n_threads, 0.0f, 0.999f);
    }
    if (resolved) resolved[e] = acc_res;
    if (linear) linear[e] = acc_lin;
}

// render.rs:156-159: Rust's saturating `as u8` (NaN → 0).
__global__ void __launch_bounds__(256) k_cast_u8(const float* __restrict__ src, uint8_t* __restrict__ dst, uint64_t n) {
    const uint64_t e = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const float f = src[e];
    uint8_t v = 0;
    if (f > 0.0f) v = f >= 255.0f ? 255 : static_cast<uint8_t>(f);
    dst[e] = v;
}

// dst += src over n floats: the sum of the slice groups' images on the device that holds the band (render.rs:141-150).
__global__ void __launch_bounds__(256) k_add_images(float* __restrict__ dst, const float* __restrict__ src, uint64_t n) {
    const uint64_t e = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (e < n) dst[e] += src[e];
}

#include "rr_wavefront.h"

// The builds of k_wf_paths that rr_paths_minreg.hip holds (compiled with another instruction scheduler): the host-side handle of
// the one for scenes with one simple subtree under the reference's semantics.
__attribute__((visibility("hidden"))) void (*rr_paths_kernel_one_subtree(bool lean))(DevScene, WfArgs, float4*, unsigned int*, unsigned long long*);

// Device memory of rr_render_device_noalloc, the entry point that allocates nothing: the counter its waves claim paths
// from (first 256 bytes) and the sample values of the paths in flight (16 bytes each).  One copy per GPU, held until
// the process ends.  Host code reaches it only through the address hipGetSymbolAddress returns for the current device
// (rr_context::scratch); the array's name is not a device address on the host.
constexpr size_t RR_SCRATCH_HEADER = 256;
constexpr size_t RR_SCRATCH_PATHS = static_cast<size_t>(RR_NOALLOC_MAX_PATHS);
constexpr size_t RR_SCRATCH_BYTES = RR_SCRATCH_HEADER + RR_SCRATCH_PATHS * 16u;
static __device__ __align__(256) unsigned char scratch[RR_SCRATCH_BYTES];

// ============================================================ host side
namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) return fail(RR_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));        \
    } while (0)

struct DeviceBuffers {
    std::vector<void*> ptrs;
    uint64_t bytes = 0;
    void release() {
        for (void* p : ptrs) (void)hipFree(p);
        ptrs.clear();
        bytes = 0;
    }
};

}  // namespace

struct rr_context {
    int device = 0;
    int n_cus = 0;
    bool has_scene = false;
    DevScene sc{};
    DeviceBuffers scene_mem;
    rr_scene_info info{};
    // ref → prim_id tables for rr_primary_hits (host side)
    std::vector<uint32_t> id_sphere, id_msphere, id_rect, id_tri, id_medium;
    std::vector<uint32_t> tex_types;  // RR_TEX_* per texture of the scene (argument checks of rr_eval_textures)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    int max_dynamic_lds = 0;
    // wavefront renderer: whether the scene has BVH subtrees, its root for the traversal machine,
    // and a grow-only workspace (path state, two queues, queue controls) with a pinned word through
    // which the host reads the live-path count when it has to (RR_FLAG_NO_FINISH_KERNEL)
    bool has_triangles = false;  // (their test needs more registers than any other)
    bool lean_scene = false;  // no OP_MED_BEGIN / MID / END in the program, every cuboid the constructor's, instances at most two deep
    bool has_subtrees = false;
    bool has_general_subtrees = false;  // OP_SUBTREE ops: subtrees the simple machine moves do not cover
    // Measurement and test switches, read from the environment ONCE, when the context is created (include/rr_hip.h documents them)
    bool sw_no_fuse = false;       // RR_NO_FUSE: one op of the scene program per transform / medium step
    int sw_pad_ops = 0;            // RR_PAD_OPS=n: n ops that do nothing appended to the scene program
    bool sw_dump_program = false;  // RR_DUMP_PROGRAM: the scene program to stderr at upload
    size_t sw_workspace_cap = 0;   // RR_WF_WORKSPACE_BYTES: cap of the per-batch workspace (0: none), to force many small batches
    bool wide_scene = false;  // the production kernels walk the BVHs with the wide node move (rr_scene_upload says when)
    bool plain_scene = false;  // no moving spheres, instances or media: k_wf_traverse has a build without their moves
    bool auto_pipeline = false;  // scenes the wavefront pipeline renders unless a flag says otherwise
    bool prog_vector_ops = false;  // the scene program has ops that read records per lane (OP_LEAF_REF, OP_SUBTREE)
    uint32_t root = NONE;
    void* wf_mem = nullptr;
    size_t wf_bytes = 0;
    uint32_t* wf_host_count = nullptr;
    unsigned long long* next_item = nullptr;  // k_render's work-item counter (device memory)
    unsigned char* scratch = nullptr;             // this device's copy of the file-scope array `scratch` (rr_render_device_noalloc)
    hipStream_t wf_aux = nullptr;                 // k_wf_finish and the accumulation of a batch run here, beside the next batch
    hipEvent_t wf_ev_main[2] = {nullptr, nullptr};  // per workspace: the batch's bounce loop is enqueued
    hipEvent_t wf_ev_aux[2] = {nullptr, nullptr};   // per workspace: the batch is accumulated, the workspace is free
    // rr_scene_specialize: the uploaded program's ops (eight words each, without the closing OP_ENDs), and the path kernel
    // compiled for them: its module, its function and the template arguments it was instantiated with
    std::vector<uint32_t> prog_words;
    hipModule_t jit_mod = nullptr;
    hipFunction_t jit_fn = nullptr;
    std::string jit_targs;
};

namespace {

// ------------------------------------------------------- scene validation
// What lies below a ref: the true bounds of every primitive (in the space the ref is evaluated
// in, over the camera's shutter interval) and what kinds of things there are.
struct Below {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    bool bounded = true;   // false: something below has no finite bounds
    bool has_tri = false;  // triangles accept hits outside the caller's t range (collider.rs:627)
    bool has_medium = false;
    bool simple = true;    // nothing but BVH nodes over spheres, rects, triangles and cuboids
    float box_lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, box_hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};  // union of the stored boxes of the BVH nodes below
    bool boxes_inside(const rr_bvh_node& n) const {  // the stored box of `n` holds the stored box of every BVH node below
        for (int a = 0; a < 3; ++a)
            if (box_lo[a] <= box_hi[a] && !(n.bmin[a] <= box_lo[a] && box_hi[a] <= n.bmax[a])) return false;
        return true;
    }
    void add_point(double x, double y, double z) {
        const double p[3] = {x, y, z};
        for (int a = 0; a < 3; ++a) {
            if (!(p[a] == p[a]) || p[a] > 1e30 || p[a] < -1e30) bounded = false;
            lo[a] = std::min(lo[a], p[a]);
            hi[a] = std::max(hi[a], p[a]);
        }
    }
    void merge(const Below& o) {
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], o.lo[a]);
            hi[a] = std::max(hi[a], o.hi[a]);
        }
        for (int a = 0; a < 3; ++a) {
            box_lo[a] = std::min(box_lo[a], o.box_lo[a]);
            box_hi[a] = std::max(box_hi[a], o.box_hi[a]);
        }
        bounded &= o.bounded;
        has_tri |= o.has_tri;
        has_medium |= o.has_medium;
        simple &= o.simple;
    }
    bool empty() const { return lo[0] > hi[0]; }
};

struct Walker {
    const rr_flat_scene& s;
    std::vector<uint8_t> seen[RR_TAG_COUNT];
    std::vector<uint32_t> owner_sphere, owner_msphere, owner_rect, owner_tri, owner_medium;
    std::vector<uint32_t> inst_depth, inst_chain;
    std::vector<Below> bvh_below;  // per BVH node: what its two children hold
    uint32_t max_inst_depth = 0;
    std::string err;
    int err_code = RR_E_INVALID;

    explicit Walker(const rr_flat_scene& sc) : s(sc) {
        const uint32_t counts[RR_TAG_COUNT] = {s.n_spheres, s.n_moving_spheres, s.n_rects,     s.n_triangles, s.n_cuboids,
                                               s.n_media,   s.n_instances,      s.n_bvh_nodes, s.n_lists};
        for (uint32_t t = 0; t < RR_TAG_COUNT; ++t) seen[t].assign(counts[t], 0);
        owner_sphere.assign(s.n_spheres, NONE);
        owner_msphere.assign(s.n_moving_spheres, NONE);
        owner_rect.assign(s.n_rects, NONE);
        owner_tri.assign(s.n_triangles, NONE);
        owner_medium.assign(s.n_media, NONE);
        inst_depth.assign(s.n_instances, 0);
        inst_chain.assign(static_cast<size_t>(s.n_instances) * MAX_CHAIN, NONE);
        bvh_below.assign(s.n_bvh_nodes, Below{});
    }

    bool bad(const std::string& m, int code = RR_E_INVALID) {
        if (err.empty()) {
            err = m;
            err_code = code;
        }
        return false;
    }

    // owner word of a primitive whose innermost enclosing instance is `inst` (NONE: none)
    uint32_t owner(uint32_t inst) const { return inst == NONE ? 0u : owner_word(inst, inst_depth[inst]); }

    bool mat_ok(uint32_t m) { return m < s.n_materials || bad("primitive references a material outside the table"); }

    // Returns the stack words evaluating `ref` can need and what lies below it, or fails.
    // `owner` is the innermost enclosing instance; depth bounds recursion (and rejects cycles
    // together with the visited marks).
    bool walk(uint32_t ref, uint32_t owner, uint32_t level, uint32_t& words, Below& below) {
        if (level > 128) return bad("scene tree deeper than 128 levels (or cyclic)", RR_E_LIMIT);
        const uint32_t tag = RR_REF_TAG(ref), idx = RR_REF_INDEX(ref);
        if (tag >= RR_TAG_COUNT || idx >= seen[tag].size()) return bad("geometry ref out of range");
        if (seen[tag][idx]++) return bad("a record is reachable twice: the flattened scene must be a tree (only a BVH node may name one child on both sides)");
        words = 0;
        below = Below{};
        switch (tag) {
            case RR_TAG_SPHERE: {
                owner_sphere[idx] = owner;
                const rr_sphere& p = s.spheres[idx];
                const double r = std::fabs(static_cast<double>(p.radius));
                below.add_point(p.center[0] - r, p.center[1] - r, p.center[2] - r);
                below.add_point(p.center[0] + r, p.center[1] + r, p.center[2] + r);
                return mat_ok(p.material);
            }
            case RR_TAG_MOVING_SPHERE: {
                owner_msphere[idx] = owner;
                const rr_moving_sphere& p = s.moving_spheres[idx];
                // the centre moves with the ray's time (collider.rs:269-271), which a caller of rr_primary_hits is
                // free to choose: a BVH node over a moving sphere has no bounds to cull with
                below.add_point(p.center0[0], p.center0[1], p.center0[2]);
                below.bounded = false;
                below.simple = false;
                return mat_ok(p.material);
            }
            case RR_TAG_RECT: {
                owner_rect[idx] = owner;
                const rr_rect& p = s.rects[idx];
                if (p.axis > RR_RECT_YZ) return bad("rect axis out of range");
                for (const float a : {p.a0, p.a1})
                    for (const float b : {p.b0, p.b1}) {
                        if (p.axis == RR_RECT_XY) below.add_point(a, b, p.k);
                        else if (p.axis == RR_RECT_XZ) below.add_point(a, p.k, b);
                        else below.add_point(p.k, a, b);
                    }
                return mat_ok(p.material);
            }
            case RR_TAG_TRIANGLE: {
                owner_tri[idx] = owner;
                const rr_triangle& p = s.triangles[idx];
                below.has_tri = true;
                // Where can a ray be accepted?  intersect_triangle (collider.rs:627-661) takes the point of the ray on
                // the plane (normal, p0) and accepts it if three products n . (edge x (point - corner)) are >= 0 in f32.
                // Each product is |edge| times the distance to the edge's line, known to about 16 ulps of
                // |edge| * |point - corner|: points up to delta = 16 * 2^-24 * L * (L / h) outside the triangle pass,
                // L the longest edge and h the smallest altitude (a sliver's tip moves out by 1 / sin of its angle).
                // The bounds grow by delta; a triangle without area, with a corner or a stored normal that is not
                // finite, or with a stored normal that is not its unit normal accepts points that have nothing to do
                // with its corners (every comparison with NaN is false) and has no bounds at all.
                const double q[3][3] = {{p.p0[0], p.p0[1], p.p0[2]}, {p.p1[0], p.p1[1], p.p1[2]}, {p.p2[0], p.p2[1], p.p2[2]}};
                double e[3][3], len[3], cr[3];
                for (int k = 0; k < 3; ++k) {
                    for (int c = 0; c < 3; ++c) e[k][c] = q[(k + 1) % 3][c] - q[k][c];
                    len[k] = std::sqrt(e[k][0] * e[k][0] + e[k][1] * e[k][1] + e[k][2] * e[k][2]);
                }
                cr[0] = e[0][1] * -e[2][2] - e[0][2] * -e[2][1];
                cr[1] = e[0][2] * -e[2][0] - e[0][0] * -e[2][2];
                cr[2] = e[0][0] * -e[2][1] - e[0][1] * -e[2][0];
                const double area2 = std::sqrt(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);  // twice the area
                const double longest = std::max({len[0], len[1], len[2]});
                bool sound = area2 > 0.0 && area2 == area2 && longest < 1e30;
                double delta = 0.0;
                if (sound) {
                    double off = 0.0;  // the stored normal against +- the unit normal of the corners
                    double dotn = 0.0;
                    for (int c = 0; c < 3; ++c) dotn += p.normal[c] * cr[c] / area2;
                    for (int c = 0; c < 3; ++c) off = std::max(off, std::fabs(p.normal[c] - (dotn < 0 ? -1.0 : 1.0) * cr[c] / area2));
                    sound = off == off && off <= 1e-3;
                    delta = 16.0 * 5.9604644775390625e-8 * longest * (longest / (area2 / longest));
                }
                for (int k = 0; k < 3; ++k) {
                    below.add_point(q[k][0] - delta, q[k][1] - delta, q[k][2] - delta);
                    below.add_point(q[k][0] + delta, q[k][1] + delta, q[k][2] + delta);
                }
                if (!sound) below.bounded = false;
                return mat_ok(p.material);
            }
            case RR_TAG_CUBOID: {
                const uint32_t first = s.cuboids[idx].first_rect;
                if (first > s.n_rects || s.n_rects - first < 6) return bad("cuboid rect range out of bounds");
                for (uint32_t i = 0; i < 6; ++i) {
                    uint32_t w;
                    Below side;
                    if (!walk(RR_REF(RR_TAG_RECT, first + i), owner, level + 1, w, side)) return false;
                    below.merge(side);
                }
                return true;
            }
            case RR_TAG_MEDIUM: {
                owner_medium[idx] = owner;
                if (!mat_ok(s.media[idx].material)) return false;
                uint32_t w;
                if (!walk(s.media[idx].boundary, owner, level + 1, w, below)) return false;
                below.has_medium = true;
                below.simple = false;
                words = FRAME_WORDS_MEDIUM + w;
                return true;
            }
            case RR_TAG_INSTANCE: {
                const rr_instance& in = s.instances[idx];
                if (in.kind > RR_INST_ROTATE_Z) return bad("instance kind out of range");
                const uint32_t depth = owner == NONE ? 1u : inst_depth[owner] + 1u;
                if (depth > MAX_CHAIN) return bad("instances nested deeper than 8", RR_E_LIMIT);
                inst_depth[idx] = depth;
                for (uint32_t k = 0; k + 1 < depth; ++k) inst_chain[idx * MAX_CHAIN + k] = inst_chain[owner * MAX_CHAIN + k];
                inst_chain[idx * MAX_CHAIN + depth - 1] = idx;
                max_inst_depth = std::max(max_inst_depth, depth);
                uint32_t w;
                Below inner;
                if (!walk(in.child, idx, level + 1, w, inner)) return false;
                words = FRAME_WORDS_INST + w;
                // bounds of the child's box seen from outside the instance (collider.rs:484, 513-517)
                below = inner;
                below.simple = false;
                if (!inner.empty()) {
                    for (int a = 0; a < 3; ++a) below.lo[a] = 1e300, below.hi[a] = -1e300;
                    for (int c = 0; c < 8; ++c) {
                        double q[3] = {(c & 1) ? inner.hi[0] : inner.lo[0], (c & 2) ? inner.hi[1] : inner.lo[1], (c & 4) ? inner.hi[2] : inner.lo[2]};
                        if (in.kind == RR_INST_TRANSLATE) {
                            for (int a = 0; a < 3; ++a) q[a] += in.v[a];
                        } else {
                            const double sn = in.v[0], cs = in.v[1];
                            const int i0 = in.kind == RR_INST_ROTATE_X ? 1 : 0, i1 = in.kind == RR_INST_ROTATE_Z ? 1 : 2;
                            const double u = cs * q[i0] + sn * q[i1], v = -sn * q[i0] + cs * q[i1];
                            q[i0] = u, q[i1] = v;
                        }
                        below.add_point(q[0], q[1], q[2]);
                    }
                }
                return true;
            }
            case RR_TAG_BVH: {
                const rr_bvh_node& n = s.bvh_nodes[idx];
                if (n.axis > 2) return bad("BVH node axis out of range");
                uint32_t wl = 0, wr = 0;
                Below bl, br;
                if (!walk(n.left, owner, level + 1, wl, bl)) return false;
                if (n.right != n.left) {
                    if (!walk(n.right, owner, level + 1, wr, br)) return false;
                } else {
                    wr = wl;
                }
                below = bl;
                below.merge(br);
                bvh_below[idx] = below;
                if (n.has_box) {  // (what the caller sees below it includes this node's own box; a box with a NaN holds nothing and is held by nothing)
                    for (int a = 0; a < 3; ++a) {
                        const bool finite = n.bmin[a] == n.bmin[a] && n.bmax[a] == n.bmax[a];
                        below.box_lo[a] = finite ? std::min(below.box_lo[a], n.bmin[a]) : -3.0e38f;
                        below.box_hi[a] = finite ? std::max(below.box_hi[a], n.bmax[a]) : 3.0e38f;
                    }
                }
                words = FRAME_WORDS_BVH + std::max(wl, wr);
                return true;
            }
            case RR_TAG_LIST: {
                const rr_list& l = s.lists[idx];
                if (l.first_child > s.n_child_refs || s.n_child_refs - l.first_child < l.n_children)
                    return bad("list child range out of bounds");
                uint32_t wmax = 0;
                for (uint32_t i = 0; i < l.n_children; ++i) {
                    uint32_t w;
                    Below c;
                    if (!walk(s.child_refs[l.first_child + i], owner, level + 1, w, c)) return false;
                    below.merge(c);
                    wmax = std::max(wmax, w);
                }
                words = FRAME_WORDS_LIST + wmax;
                below.simple = false;
                return true;
            }
        }
        return bad("unknown tag");
    }
};

uint32_t to_dref(uint32_t ref, uint32_t extra = 0) { return dref(RR_REF_TAG(ref), RR_REF_INDEX(ref), extra); }

// Corners of a cuboid whose six rects are exactly what Cuboid::new (collider.rs:419-426) makes of two
// corners, in its order; false for any other six rects.
bool cuboid_corners(const rr_flat_scene& s, uint32_t ci, float p0[3], float p1[3]) {
    const rr_rect* r = &s.rects[s.cuboids[ci].first_rect];
    p0[0] = r[0].a0, p0[1] = r[0].b0, p0[2] = r[1].k;
    p1[0] = r[0].a1, p1[1] = r[0].b1, p1[2] = r[0].k;
    auto is = [](const rr_rect& q, uint32_t axis, float a0, float a1, float b0, float b1, float k) {
        const float have[5] = {q.a0, q.a1, q.b0, q.b1, q.k}, want[5] = {a0, a1, b0, b1, k};
        return q.axis == axis && std::memcmp(have, want, sizeof(have)) == 0;
    };
    return is(r[0], RR_RECT_XY, p0[0], p1[0], p0[1], p1[1], p1[2]) && is(r[1], RR_RECT_XY, p0[0], p1[0], p0[1], p1[1], p0[2]) &&
           is(r[2], RR_RECT_XZ, p0[0], p1[0], p0[2], p1[2], p1[1]) && is(r[3], RR_RECT_XZ, p0[0], p1[0], p0[2], p1[2], p0[1]) &&
           is(r[4], RR_RECT_YZ, p0[1], p1[1], p0[2], p1[2], p1[0]) && is(r[5], RR_RECT_YZ, p0[1], p1[1], p0[2], p1[2], p0[0]);
}

// ---------------------------------------------------------- scene program
// Compiles the ray-independent part of the (already validated) tree into the op array that
// rrdev::traverse interprets: lists are unrolled in order, instances become transform ops on
// the current ray space, cuboids become one op (or their six rect ops if they are not the constructor's), a medium becomes begin / boundary /
// mid / boundary / end, and every BVH node (or a medium nested in a medium's boundary) becomes
// one OP_SUBTREE for the per-lane machine.
struct ProgramBuilder {
    const rr_flat_scene& s;
    const std::vector<Below>& bvh_below;     // per BVH node: what lies below it
    const std::vector<uint8_t>& bvh_class;   // per BVH node: BVH_CLASS_*
    std::vector<Op> ops;
    std::vector<uint32_t> chain;  // instances enclosing the node being emitted, outermost first
    std::vector<uint32_t> space;  // instances the ops emitted so far have applied to the ray
    bool in_medium = false;
    bool fuse = true;  // emit the fused ops (OP_MED_SPHERE, OP_XFORM); off under RR_NO_FUSE, for the test that compares the two forms

    static Op make(uint32_t code, uint32_t ref = NONE, float f0 = 0, float f1 = 0, float f2 = 0, float f3 = 0, float f4 = 0,
                   float f5 = 0) {
        return Op{code, ref, f0, f1, f2, f3, f4, f5};
    }
    void xform(uint32_t ii, bool entering) {
        const rr_instance& in = s.instances[ii];
        const bool flag = entering;  // an instance being entered counts as a visit
        // (one-part OP_XFORMs; fuse_transforms merges the runs)
        if (in.kind == RR_INST_TRANSLATE)
            ops.push_back(make(OP_XFORM | OPX_TRANSLATE | (flag ? OPX_COUNT_TRANSLATE : 0u), NONE, in.v[0], in.v[1], in.v[2]));
        else
            ops.push_back(make(OP_XFORM | ((in.kind == RR_INST_ROTATE_X ? 1u : in.kind == RR_INST_ROTATE_Y ? 2u : 3u) << OPX_ROT_SHIFT) |
                                   (flag ? OPX_COUNT_ROT : 0u),
                               NONE, 0.f, 0.f, 0.f, in.v[0], in.v[1]));
    }
    // Make the ray space of the next op equal to `chain`, always recomputing from the world ray.
    void ensure_space() {
        if (space == chain) return;
        size_t common = 0;
        while (common < space.size() && common < chain.size() && space[common] == chain[common]) ++common;
        if (common != space.size()) {  // leaving a space cannot be undone exactly: start again from the world ray
            ops.push_back(make(OP_XFORM | OPX_RESET));
            common = 0;
        }
        for (size_t k = common; k < chain.size(); ++k) xform(chain[k], false);
        space = chain;
    }
    // Runs of one-part OP_XFORMs in the order reset, translate, rotate (each at most once) become one OP_XFORM each.
    void fuse_transforms() {
        std::vector<Op> out;
        auto parts = [](const Op& o) { return o.code & (OPX_RESET | OPX_TRANSLATE | (3u << OPX_ROT_SHIFT)); };
        for (size_t i = 0; i < ops.size();) {
            if ((ops[i].code & 0xFFu) != OP_XFORM) {
                out.push_back(ops[i++]);
                continue;
            }
            Op x = ops[i++];
            // what may still follow in this op: translate after reset; rotate after reset or translate
            while (i < ops.size() && (ops[i].code & 0xFFu) == OP_XFORM) {
                const uint32_t have = parts(x), next = parts(ops[i]);
                const bool t_next = next == OPX_TRANSLATE, r_next = (next & (3u << OPX_ROT_SHIFT)) != 0u && (next & ~(3u << OPX_ROT_SHIFT)) == 0u;
                if (t_next && !(have & (OPX_TRANSLATE | (3u << OPX_ROT_SHIFT)))) {
                    x.code |= ops[i].code & (OPX_TRANSLATE | OPX_COUNT_TRANSLATE);
                    x.f0 = ops[i].f0, x.f1 = ops[i].f1, x.f2 = ops[i].f2;
                } else if (r_next && !(have & (3u << OPX_ROT_SHIFT))) {
                    x.code |= ops[i].code & ((3u << OPX_ROT_SHIFT) | OPX_COUNT_ROT);
                    x.f3 = ops[i].f3, x.f4 = ops[i].f4;
                } else {
                    break;
                }
                ++i;
            }
            out.push_back(x);
        }
        ops.swap(out);
    }
    void rect_op(uint32_t ri) {
        const rr_rect& r = s.rects[ri];
        const uint32_t code = r.axis == RR_RECT_XY ? OP_RECT_XY : (r.axis == RR_RECT_XZ ? OP_RECT_XZ : OP_RECT_YZ);
        ops.push_back(make(code, dref(RR_TAG_RECT, ri), r.a0, r.a1, r.b0, r.b1, r.k));
    }
    void emit(uint32_t ref) {
        const uint32_t tag = RR_REF_TAG(ref), idx = RR_REF_INDEX(ref);
        switch (tag) {
            case RR_TAG_LIST: {
                const rr_list& l = s.lists[idx];
                for (uint32_t i = 0; i < l.n_children; ++i) emit(s.child_refs[l.first_child + i]);
                break;
            }
            case RR_TAG_INSTANCE:
                ensure_space();
                xform(idx, true);
                chain.push_back(idx);
                space = chain;
                emit(s.instances[idx].child);
                chain.pop_back();
                break;
            case RR_TAG_RECT:
                ensure_space();
                rect_op(idx);
                break;
            case RR_TAG_CUBOID: {
                ensure_space();
                float p0[3], p1[3];
                if (cuboid_corners(s, idx, p0, p1))  // all six sides in one op, from the corners
                    ops.push_back(make(OP_CUBOID, dref(RR_TAG_RECT, s.cuboids[idx].first_rect), p0[0], p0[1], p0[2], p1[0], p1[1], p1[2]));
                else
                    for (uint32_t i = 0; i < 6; ++i) rect_op(s.cuboids[idx].first_rect + i);
                break;
            }
            case RR_TAG_SPHERE: {
                ensure_space();
                const rr_sphere& p = s.spheres[idx];
                ops.push_back(make(OP_SPHERE, dref(RR_TAG_SPHERE, idx), p.center[0], p.center[1], p.center[2], p.radius));
                break;
            }
            case RR_TAG_TRIANGLE:
            case RR_TAG_MOVING_SPHERE:
                ensure_space();
                ops.push_back(make(OP_LEAF_REF, dref(tag, idx)));
                break;
            case RR_TAG_MEDIUM:
                ensure_space();
                if (in_medium) {  // a medium inside a medium's boundary: whole subtree to the machine
                    ops.push_back(make(OP_SUBTREE, dref(tag, idx)));
                    break;
                }
                if (fuse && RR_REF_TAG(s.media[idx].boundary) == RR_TAG_SPHERE) {  // both probes of a bare sphere in one op
                    const rr_sphere& p = s.spheres[RR_REF_INDEX(s.media[idx].boundary)];
                    ops.push_back(make(OP_MED_SPHERE, dref(tag, idx), p.center[0], p.center[1], p.center[2], p.radius, s.media[idx].neg_inv_density));
                    break;
                }
                in_medium = true;
                ops.push_back(make(OP_MED_BEGIN));
                emit(s.media[idx].boundary);
                ops.push_back(make(OP_MED_MID));
                emit(s.media[idx].boundary);
                ensure_space();  // the free path is measured along the ray in the medium's own space
                ops.push_back(make(OP_MED_END, dref(tag, idx), s.media[idx].neg_inv_density));
                in_medium = false;
                break;
            default: {  // RR_TAG_BVH
                ensure_space();
                const rr_bvh_node& n = s.bvh_nodes[idx];
                // the root's box and class travel in the op: its test is made before the descent is handed out, and the margin
                // of the culling test is formed once per visit from that box, which must then hold every box below it (the boxes
                // of a tree built bottom-up from surrounding boxes do, collider.rs:205-211; any other tree takes the general machine)
                if (bvh_below[idx].simple && n.has_box && bvh_below[idx].boxes_inside(n))
                    ops.push_back(make(OP_SUBTREE_SIMPLE | (static_cast<uint32_t>(bvh_class[idx]) << 8), dref(tag, idx), n.bmin[0], n.bmin[1], n.bmin[2],
                                       n.bmax[0], n.bmax[1], n.bmax[2]));
                else
                    ops.push_back(make(OP_SUBTREE, dref(tag, idx)));
                break;
            }
        }
    }
};

template <class T>
int upload(DeviceBuffers& mem, const std::vector<T>& host, const T** out) {
    *out = nullptr;
    if (host.empty()) return RR_OK;
    void* p = nullptr;
    const size_t bytes = host.size() * sizeof(T);
    HIP_TRY(hipMalloc(&p, bytes));
    mem.ptrs.push_back(p);
    mem.bytes += bytes;
    HIP_TRY(hipMemcpy(p, host.data(), bytes, hipMemcpyHostToDevice));
    *out = static_cast<const T*>(p);
    return RR_OK;
}

float u2f(uint32_t u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

int set_device(rr_context* ctx) {
    HIP_TRY(hipSetDevice(ctx->device));
    return RR_OK;
}

int launch_dims(rr_context* ctx, size_t lds_bytes, const void* kernel, int* grid, hipFunction_t module_fn = nullptr) {
    int per_cu = 0;
    if (module_fn) HIP_TRY(hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, module_fn, 256, lds_bytes));
    else HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, lds_bytes));
    if (per_cu < 1) {
        const size_t stacks = static_cast<size_t>(ctx->info.stack_words) * 256u * 4u;
        return fail(RR_E_LIMIT, "a workgroup of this kernel needs " + std::to_string(lds_bytes) + " bytes of LDS (" + std::to_string(stacks) +
                                    " for the traversal stacks of " + std::to_string(ctx->info.stack_words) + " words per lane" +
                                    (lds_bytes > stacks ? ", " + std::to_string(lds_bytes - stacks) + " for the slots of parked lanes" : std::string()) +
                                    "), more than one CU gives a workgroup");
    }
    per_cu = std::min(per_cu, 8);
    *grid = ctx->n_cus * per_cu;
    return RR_OK;
}

// The band of image rows a call renders: [row_begin, row_end), row_end = 0 standing for the image height.
bool band_of(const rr_render_params* p, uint32_t* row0, uint32_t* rows) {
    const uint32_t end = p->row_end == 0 ? p->height : p->row_end;
    if (p->row_begin > end || end > p->height) return false;
    *row0 = p->row_begin;
    *rows = end - p->row_begin;
    return true;
}

int resolve_slices(rr_context* ctx, const rr_render_params* p, uint32_t rows, uint32_t n_sl, uint32_t spt, void* d_slice_sums, void* d_resolved,
                   void* d_linear, hipStream_t stream) {
    (void)ctx;
    if ((d_resolved || d_linear) && rows != 0) {
        const uint64_t n_elems = static_cast<uint64_t>(p->width) * rows * 3u;
        const uint32_t g = static_cast<uint32_t>((n_elems + 255u) / 256u);
        hipLaunchKernelGGL(k_resolve, dim3(g), dim3(256), 0, stream, static_cast<const float*>(d_slice_sums), n_sl, n_elems,
                           1.0f / static_cast<float>(spt), static_cast<float>(p->n_slices), static_cast<float*>(d_resolved),
                           static_cast<float*>(d_linear));
        HIP_TRY(hipGetLastError());
    }
    return RR_OK;
}

// The wavefront renderer for the slices of one pass (a.n_slots slots, absolute first slice
// a.slice_first): batches of samples, each a few bounces of traversal and shading launches, one
// launch that runs the remaining paths to their ends, and the in-order accumulation of the batch.
// Everything is enqueued; the function does not wait for the device (except under
// RR_FLAG_NO_FINISH_KERNEL, where the host has to learn when the last path has ended).
int render_wavefront(rr_context* ctx, WfArgs a, int count, uint32_t flags, float* sums, unsigned long long* counters, hipStream_t stream) {
    // Batch size: as many samples per slot as fit the workspace budget (80 bytes per path).  Two workspaces
    // take turns: the last paths of a batch (whose bounces each last as long as their slowest ray while the
    // chip idles) and its accumulation run on a second stream beside the next batch's bounce loop.
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    // per workspace: at most 16 GiB and a quarter of what is free
    const size_t budget = std::max<size_t>(std::min<size_t>(static_cast<size_t>(16) << 30, (free_b + ctx->wf_bytes) / 4), static_cast<size_t>(1) << 20);
    size_t cap = budget;
    if (ctx->sw_workspace_cap) cap = ctx->sw_workspace_cap;  // (for tests)
    const uint64_t most = std::min<uint64_t>((1ull << 31) / a.n_slots, std::max<uint64_t>(1, cap / (80u * static_cast<uint64_t>(a.n_slots))));
    uint64_t batch = std::max<uint64_t>(1, std::min<uint64_t>(budget / (80u * static_cast<uint64_t>(a.n_slots)), most));
    const bool two = batch < a.spt;  // more than one batch: a second workspace, so that they overlap
    batch = std::min<uint64_t>(batch, std::max(a.spt, 1u));
    const size_t n = static_cast<size_t>(a.n_slots) * batch;
    auto align = [](size_t v) { return (v + 255u) & ~static_cast<size_t>(255u); };
    const size_t off_o = 0, off_d = off_o + align(n * 16), off_t = off_d + align(n * 16), off_h = off_t + align(n * 16),
                 off_q0 = off_h + align(n * 16), off_q1 = off_q0 + align(n * 4), off_ctl = off_q1 + align(n * 4),
                 one = off_ctl + align(3 * sizeof(WfCtl)), need = one * (two ? 2u : 1u);
    if (need > ctx->wf_bytes) {
        if (ctx->wf_mem) {
            HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(hipFree(ctx->wf_mem));
            ctx->wf_mem = nullptr;
            ctx->wf_bytes = 0;
        }
        HIP_TRY(hipMalloc(&ctx->wf_mem, need));
        ctx->wf_bytes = need;
    }
    if (!ctx->wf_host_count) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&ctx->wf_host_count), 64, hipHostMallocDefault));
    if (!ctx->wf_aux) {
        HIP_TRY(hipStreamCreateWithFlags(&ctx->wf_aux, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) {
            HIP_TRY(hipEventCreateWithFlags(&ctx->wf_ev_main[k], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&ctx->wf_ev_aux[k], hipEventDisableTiming));
        }
    }
    struct Workspace {
        WfState st;
        uint32_t* queue[2];
        WfCtl* ctl;
    } ws[2];
    for (int k = 0; k < 2; ++k) {
        char* base = static_cast<char*>(ctx->wf_mem) + (two ? k * one : 0);
        ws[k].st = WfState{reinterpret_cast<float4*>(base + off_o), reinterpret_cast<float4*>(base + off_d),
                           reinterpret_cast<float4*>(base + off_t), reinterpret_cast<uint4*>(base + off_h)};
        ws[k].queue[0] = reinterpret_cast<uint32_t*>(base + off_q0);
        ws[k].queue[1] = reinterpret_cast<uint32_t*>(base + off_q1);
        ws[k].ctl = reinterpret_cast<WfCtl*>(base + off_ctl);
    }

#ifndef RR_WF_TRAVERSE_EXTRA_LDS
#define RR_WF_TRAVERSE_EXTRA_LDS 0u  // (measurements: bytes of LDS a traversal workgroup asks for beyond its stacks, to lower its occupancy)
#endif
    const size_t lds = static_cast<size_t>(ctx->info.stack_words) * 256u * 4u + RR_WF_TRAVERSE_EXTRA_LDS;
    using TraverseKernel = void (*)(DevScene, WfArgs, WfState, const uint32_t*, WfCtl*, uint32_t, unsigned long long*);
    using FinishKernel = void (*)(DevScene, WfArgs, WfState, const uint32_t*, WfCtl*, unsigned long long*);
    const bool ref_sem = ctx->sc.intended == 0u && ctx->sc.book == 0u;  // (the production kernels have a build that knows the semantics switches are off)
    const bool wide = ctx->wide_scene && !(flags & RR_FLAG_NARROW_NODES);  // (a wide scene has the reference's semantics and no general subtrees)
    const TraverseKernel k_traverse = count == COUNT_PRODUCTION ? (wide ? k_wf_traverse<COUNT_PRODUCTION, false, true> : k_wf_traverse<COUNT_PRODUCTION>)
                                      : count == COUNT_REFERENCE ? k_wf_traverse<COUNT_REFERENCE>
                                      : wide ? (ctx->plain_scene ? k_wf_traverse<COUNT_OFF, true, true, true> : k_wf_traverse<COUNT_OFF, true, true>)
                                      : ref_sem ? k_wf_traverse<COUNT_OFF, true> : k_wf_traverse<COUNT_OFF>;
    const FinishKernel k_finish = count == COUNT_PRODUCTION ? k_wf_finish<COUNT_PRODUCTION>
                                  : count == COUNT_REFERENCE ? k_wf_finish<COUNT_REFERENCE>
                                  : ctx->has_general_subtrees ? (ref_sem ? k_wf_finish<COUNT_OFF, true, true> : k_wf_finish<COUNT_OFF>)
                                                              : (ref_sem ? k_wf_finish<COUNT_OFF, false, true> : k_wf_finish<COUNT_OFF, false>);
    int grid_full = 0;
    if (int rc = launch_dims(ctx, lds, reinterpret_cast<const void*>(k_traverse), &grid_full)) return rc;
    const uint32_t slot_blocks = (a.n_slots + 255u) / 256u;
    const size_t lds_finish = lds + 4u * PARK_LDS_WORDS_PER_WAVE * 4u;  // the stacks, then each wave's slots for parked lanes
    int grid_finish = 0;
    if (int rc = launch_dims(ctx, lds_finish, reinterpret_cast<const void*>(k_finish), &grid_finish)) return rc;
    hipStream_t aux = ctx->wf_aux;

    // A batch takes its camera rays as one pair of launches over the queue of all its paths, then one launch of
    // k_wf_finish runs what is left to its end: nothing is read back, the whole render is only enqueued.
    // The bounce kernels are the faster way through the camera rays (8 waves per SIMD of pure traversal with
    // lanes refilled in mid-flight, coherent rays); from the second ray on k_wf_finish is: it walks the part of
    // the tree every ray visits with the scene program, pays no launch gaps and tails, and parks the few lanes
    // with deep BVH descents instead of waiting for them.  Measured, Msamples/s with 0/1/2/3 pairs of bounce
    // launches: Suzanne 128 M paths 1975/2137/1965/1851; final_scene 102 M paths 1043/1180/1188/1192; random
    // spheres 36 M paths 2234/2856/2902/2739, 9 M paths 2070/2407/-/-.
    // With RR_FLAG_NO_FINISH_KERNEL it is bounce launches throughout, and the host reads the number of live
    // paths back after every bounce from the fourth on to know when to stop.
#ifndef RR_WF_FINISH_PATHS_PER_LANE
#define RR_WF_FINISH_PATHS_PER_LANE 2u
#endif
    constexpr uint32_t POLL_FROM = 3;
    uint32_t n_batches = 0;
    for (uint32_t s0 = 0; s0 < a.spt || s0 == 0; s0 += static_cast<uint32_t>(batch), ++n_batches) {
        const int w = two ? static_cast<int>(n_batches & 1u) : 0;
        const WfState& st = ws[w].st;
        uint32_t* const* queue = ws[w].queue;
        WfCtl* ctl = ws[w].ctl;
        a.s0 = s0;
        a.n_paths = a.n_slots * static_cast<uint32_t>(std::min<uint64_t>(batch, a.spt - s0));
        a.n_sub = a.pass_slices * static_cast<uint32_t>(std::min<uint64_t>(batch, a.spt - s0));
        // the workspace is free once the batch that used it last has been accumulated
        if (n_batches >= (two ? 2u : 1u)) HIP_TRY(hipStreamWaitEvent(stream, ctx->wf_ev_aux[w], 0));
        uint32_t finish_bound = 0;  // upper bound of the number of paths handed to k_wf_finish (0: none are)
        const uint32_t* finish_queue = nullptr;
        WfCtl* finish_ctl = nullptr;
        if (a.n_paths != 0) {
            const uint32_t path_blocks = (a.n_paths + 255u) / 256u;
            hipLaunchKernelGGL(k_wf_init, dim3(std::min<uint32_t>(path_blocks, static_cast<uint32_t>(ctx->n_cus) * 32u)), dim3(256), 0, stream, ctx->sc, a, st, queue[0], ctl, counters);
            HIP_TRY(hipGetLastError());
            const bool poll = (flags & RR_FLAG_NO_FINISH_KERNEL) != 0;
            const uint32_t bounces = 1;
            uint32_t live = a.n_paths;  // upper bound of the queue length, refreshed at every poll
            // a path ends at the latest in the ray_color call that finds its depth used up (render.rs:16)
            for (uint32_t it = 0; it <= a.max_depth; ++it) {
                if (!poll && it == bounces) {
                    finish_bound = std::max<uint32_t>(a.n_paths >> it, 1u << 16);
                    finish_queue = queue[it & 1u];
                    finish_ctl = ctl + it % 3u;
                    break;
                }
                WfCtl *c_in = ctl + it % 3u, *c_out = ctl + (it + 1u) % 3u, *c_clear = ctl + (it + 2u) % 3u;
                const uint32_t* q_in = queue[it & 1u];
                uint32_t* q_out = queue[(it + 1u) & 1u];
                const uint32_t live_blocks = (live + 255u) / 256u;
                const uint32_t grid_t = std::min<uint32_t>(static_cast<uint32_t>(grid_full), live_blocks);
                const uint32_t grid_s = std::min<uint32_t>(live_blocks, static_cast<uint32_t>(ctx->n_cus) * 16u);
                hipLaunchKernelGGL(k_traverse, dim3(grid_t), dim3(256), lds, stream, ctx->sc, a, st, q_in, c_in, ctx->root, counters);
                hipLaunchKernelGGL(k_wf_shade, dim3(grid_s), dim3(256), 0, stream, ctx->sc, a, st, q_in, c_in, q_out, c_out, c_clear, counters);
                HIP_TRY(hipGetLastError());
                if (poll && it >= POLL_FROM && it != a.max_depth) {
                    HIP_TRY(hipMemcpyAsync(ctx->wf_host_count, &c_out->count, 4, hipMemcpyDeviceToHost, stream));
                    HIP_TRY(hipStreamSynchronize(stream));
                    live = *ctx->wf_host_count;
                    if (live == 0) break;
                }
            }
        }
        // the rest of the batch's paths and its accumulation on the second stream, in batch order (render.rs:99 adds samples in order)
        HIP_TRY(hipEventRecord(ctx->wf_ev_main[w], stream));
        HIP_TRY(hipStreamWaitEvent(aux, ctx->wf_ev_main[w], 0));
        if (finish_bound != 0) {
            // lanes take new paths as theirs end: fewer lanes than paths, so that the long paths do not hold whole waves
            // (a wave that finds the queue empty leaves at once)
            const uint32_t grid_f = std::min<uint32_t>((finish_bound + 256u * RR_WF_FINISH_PATHS_PER_LANE - 1u) / (256u * RR_WF_FINISH_PATHS_PER_LANE),
                                                       static_cast<uint32_t>(grid_finish));
            hipLaunchKernelGGL(k_finish, dim3(grid_f), dim3(256), lds_finish, aux, ctx->sc, a, st, finish_queue, finish_ctl, counters);
            HIP_TRY(hipGetLastError());
        }
        hipLaunchKernelGGL(k_wf_accumulate, dim3(slot_blocks), dim3(256), 0, aux, a, st, sums);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(ctx->wf_ev_aux[w], aux));
        if (a.spt == 0) {
            ++n_batches;
            break;
        }
    }
    // the caller's stream goes on after the last accumulations
    for (uint32_t k = 0; k < std::min(n_batches, two ? 2u : 1u); ++k) HIP_TRY(hipStreamWaitEvent(stream, ctx->wf_ev_aux[two ? k : 0], 0));
    return RR_OK;
}

// Which build of k_wf_paths renders the context's scene, and the template arguments it is an instance of, as text: what
// rr_scene_specialize instantiates in its per-scene compile.
using PathsKernel = void (*)(DevScene, WfArgs, float4*, unsigned int*, unsigned long long*);
struct PathsChoice {
    PathsKernel fn;
    const char* targs;
    bool cells = false;  // a build that can take its paths cell by cell (WfArgs::tile_paths == 2): the guests builds
};
constexpr int W5 = RR_RENDER_WAVES_PER_SIMD + 1;
#define RR_PK(...) PathsChoice{k_wf_paths<__VA_ARGS__>, #__VA_ARGS__}
PathsChoice choose_paths_kernel(const rr_context* ctx, int count) {
    // (a scene without BVH nodes has no visits to skip: its two counting modes are one kernel)
    // The production builds (no counting): without the semantics switch where it is off (REF_SEM), and for lean scenes
    // (rr_context::lean_scene) the builds at 5 waves per SIMD instead of 4 that leave out what such scenes do not need (see k_wf_paths).
    const bool ref_sem = ctx->sc.intended == 0u && ctx->sc.book == 0u, w5 = ctx->lean_scene && !ctx->has_triangles;
    if (ctx->has_subtrees) {
        if (count == COUNT_PRODUCTION) return RR_PK(COUNT_PRODUCTION, true, true);
        if (count == COUNT_REFERENCE) return RR_PK(COUNT_REFERENCE, true, true);
        if (!ref_sem) return ctx->has_general_subtrees ? RR_PK(COUNT_OFF, true, true) : RR_PK(COUNT_OFF, true, true, false);
        if (ctx->has_general_subtrees) return w5 ? RR_PK(COUNT_OFF, true, true, true, true, W5, true) : RR_PK(COUNT_OFF, true, true, true, true);
        // (scenes whose subtrees are all simple take 5 waves whether lean or not: Suzanne +4 % once parked lanes stopped
        // holding two copies of their ray; a lean build for lean scenes with triangles measured no different from the full one)
        // (two or more subtrees: the builds in which parked lanes are guests of the other subtrees' visits)
        if (ctx->info.subtree_ops[0] >= 2u) {
            PathsChoice c = w5 ? RR_PK(COUNT_OFF, true, true, false, true, W5, true, true) : RR_PK(COUNT_OFF, true, true, false, true, W5, false, true);
            c.cells = true;
            return c;
        }
        // (one subtree: the same builds, compiled in rr_paths_minreg.hip with the scheduler that minimises register use)
        return w5 ? PathsChoice{rr_paths_kernel_one_subtree(true), "COUNT_OFF, true, true, false, true, W5, true"}
                  : PathsChoice{rr_paths_kernel_one_subtree(false), "COUNT_OFF, true, true, false, true, W5, false"};
    }
    if (ctx->prog_vector_ops) {
        // (without subtrees REF_SEM only concerns shading: the builds that know the book estimator is off do not carry its code)
        if (count) return RR_PK(COUNT_REFERENCE, true);
        if (ref_sem) return w5 ? RR_PK(COUNT_OFF, true, false, true, true, W5, true) : RR_PK(COUNT_OFF, true, false, true, true);
        return w5 ? RR_PK(COUNT_OFF, true, false, true, false, W5, true) : RR_PK(COUNT_OFF, true);
    }
    if (count) return RR_PK(COUNT_REFERENCE, false);
    if (ref_sem) return w5 ? RR_PK(COUNT_OFF, false, false, true, true, W5, true) : RR_PK(COUNT_OFF, false, false, true, true);
    return w5 ? RR_PK(COUNT_OFF, false, false, true, false, W5, true) : RR_PK(COUNT_OFF, false);
}
#undef RR_PK

// The persistent renderer at path granularity: batches of samples of every slot, each one launch of k_wf_paths that
// leaves every sample's value in `values` (room for `capacity` paths), and the in-order accumulation of the batch.
// Allocates nothing: the caller brings the values' memory and the device word the waves claim their paths from.
int render_paths_in(rr_context* ctx, WfArgs a, int count, float* sums, unsigned long long* counters, hipStream_t stream,
                    float4* values, uint64_t capacity, unsigned int* next_path) {
    const uint64_t most = std::min<uint64_t>((1ull << 31) / a.n_slots, capacity / a.n_slots);
    if (most == 0) return fail(RR_E_LIMIT, "render_paths: the sample values of one sample per (slice, pixel) do not fit the workspace");
    const uint64_t batch = std::min<uint64_t>(most, std::max(a.spt, 1u));
    WfState st{};
    st.thr = values;
    const PathsChoice choice = choose_paths_kernel(ctx, count);
    const PathsKernel kernel = choice.fn;
    if (a.tile_paths == 1u && choice.cells) a.tile_paths = 2u;  // cell by cell: wf_path_key
    // the kernel rr_scene_specialize compiled for this scene's program stands in for the interpreter's build of the same name
    const bool jit = count == COUNT_OFF && ctx->jit_fn != nullptr && ctx->jit_targs == choice.targs;
    // the stacks, then (scenes with subtrees) each wave's slots for parked lanes
    const size_t lds = static_cast<size_t>(ctx->info.stack_words) * 256u * 4u + (ctx->has_subtrees ? 4u * PARK_LDS_WORDS_PER_WAVE * 4u : 0u);
    int grid = 0;
    if (int rc = launch_dims(ctx, lds, reinterpret_cast<const void*>(kernel), &grid, jit ? ctx->jit_fn : nullptr)) return rc;
    const uint32_t slot_blocks = (a.n_slots + 255u) / 256u;
    for (uint32_t s0 = 0; s0 < a.spt || s0 == 0; s0 += static_cast<uint32_t>(batch)) {
        a.s0 = s0;
        a.n_paths = a.n_slots * static_cast<uint32_t>(std::min<uint64_t>(batch, a.spt - s0));
        a.n_sub = a.pass_slices * static_cast<uint32_t>(std::min<uint64_t>(batch, a.spt - s0));
        if (a.n_paths != 0) {
            HIP_TRY(hipMemsetAsync(next_path, 0, sizeof(unsigned long long), stream));
            const uint32_t g = std::min<uint32_t>(static_cast<uint32_t>(grid), (a.n_paths + 255u) / 256u);
#ifndef RR_PATHS_FIXED_CHUNK
            // Small batches in smaller chunks, so that every wave makes several claims and the waves end together: a sixteenth of
            // a wave's share, in steps of 64 paths between 64 and RR_PATHS_CHUNK.
            // (cell by cell a wave's rays are samples of one pixel and cheaper, and the claims' atomics show earlier: twice the
            // claim there.  final_scene at 10 000 spp with 1 024 / 2 048 / 4 096 instead of 512: +1.6 / +2.2 / +1.6 %, at 400 spp
            // +0.5 / -1.9 / -4.1 %; in tile order Suzanne -0.9 / -2.7 / -3.1 %, random spheres at 1000 spp -1.5 / -1.7 / -1.4 %.)
            a.paths_chunk = std::min<uint32_t>(std::max<uint32_t>(a.n_paths / (g * 4u * 16u) / 64u * 64u, 64u), a.tile_paths == 2u ? 2u * RR_PATHS_CHUNK : RR_PATHS_CHUNK);
#endif
            if (jit) {
                DevScene sc_arg = ctx->sc;
                float4* values_arg = st.thr;
                void* params[] = {&sc_arg, &a, &values_arg, &next_path, &counters};
                HIP_TRY(hipModuleLaunchKernel(ctx->jit_fn, g, 1, 1, 256, 1, 1, static_cast<unsigned int>(lds), stream, params, nullptr));
            } else {
                hipLaunchKernelGGL(kernel, dim3(g), dim3(256), lds, stream, ctx->sc, a, st.thr, next_path, counters);
                HIP_TRY(hipGetLastError());
            }
        }
        hipLaunchKernelGGL(k_wf_accumulate, dim3(slot_blocks), dim3(256), 0, stream, a, st, sums);
        HIP_TRY(hipGetLastError());
        if (a.spt == 0) break;
    }
    return RR_OK;
}

// render_paths_in with the context's grow-only workspace: as many samples per slot in flight as min(16 GiB, a quarter of
// free HBM) holds.
int render_paths(rr_context* ctx, WfArgs a, int count, float* sums, unsigned long long* counters, hipStream_t stream) {
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    const size_t budget = std::max<size_t>(std::min<size_t>(static_cast<size_t>(16) << 30, (free_b + ctx->wf_bytes) / 4), static_cast<size_t>(1) << 20);
    size_t cap = budget;
    if (ctx->sw_workspace_cap) cap = ctx->sw_workspace_cap;  // (for tests)
    const uint64_t most = std::min<uint64_t>((1ull << 31) / a.n_slots, std::max<uint64_t>(1, cap / (16u * static_cast<uint64_t>(a.n_slots))));
    const uint64_t batch = std::max<uint64_t>(1, std::min<uint64_t>(most, std::max(a.spt, 1u)));
    const size_t need = static_cast<size_t>(a.n_slots) * batch * 16u + 256u;
    if (need > ctx->wf_bytes) {
        if (ctx->wf_mem) {
            HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(hipFree(ctx->wf_mem));
            ctx->wf_mem = nullptr;
            ctx->wf_bytes = 0;
        }
        HIP_TRY(hipMalloc(&ctx->wf_mem, need));
        ctx->wf_bytes = need;
    }
    if (!ctx->next_item) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->next_item), sizeof(unsigned long long)));
    return render_paths_in(ctx, a, count, sums, counters, stream, static_cast<float4*>(ctx->wf_mem), static_cast<uint64_t>(a.n_slots) * batch,
                           reinterpret_cast<unsigned int*>(ctx->next_item));
}

// ------------------------------------------------------------------------------------ per-scene compile (rr_scene_specialize)
// libhiprtc.so, opened the first time a scene is specialised: a host that never asks needs no runtime compiler.
struct Hiprtc {
    void* lib = nullptr;
    std::string why;
    int (*CreateProgram)(void**, const char*, const char*, int, const char* const*, const char* const*) = nullptr;
    int (*AddNameExpression)(void*, const char*) = nullptr;
    int (*CompileProgram)(void*, int, const char* const*) = nullptr;
    int (*GetProgramLogSize)(void*, size_t*) = nullptr;
    int (*GetProgramLog)(void*, char*) = nullptr;
    int (*GetLoweredName)(void*, const char*, const char**) = nullptr;
    int (*GetCodeSize)(void*, size_t*) = nullptr;
    int (*GetCode)(void*, char*) = nullptr;
    int (*DestroyProgram)(void**) = nullptr;
    bool ok() const { return lib != nullptr; }
};
Hiprtc& hiprtc() {
    static Hiprtc h;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"}) {
            h.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (h.lib) break;
        }
        if (!h.lib) {
            h.why = std::string("libhiprtc.so cannot be opened: ") + (dlerror() ? dlerror() : "unknown error");
            return;
        }
        auto sym = [&](const char* n) { return dlsym(h.lib, n); };
        h.CreateProgram = reinterpret_cast<decltype(h.CreateProgram)>(sym("hiprtcCreateProgram"));
        h.AddNameExpression = reinterpret_cast<decltype(h.AddNameExpression)>(sym("hiprtcAddNameExpression"));
        h.CompileProgram = reinterpret_cast<decltype(h.CompileProgram)>(sym("hiprtcCompileProgram"));
        h.GetProgramLogSize = reinterpret_cast<decltype(h.GetProgramLogSize)>(sym("hiprtcGetProgramLogSize"));
        h.GetProgramLog = reinterpret_cast<decltype(h.GetProgramLog)>(sym("hiprtcGetProgramLog"));
        h.GetLoweredName = reinterpret_cast<decltype(h.GetLoweredName)>(sym("hiprtcGetLoweredName"));
        h.GetCodeSize = reinterpret_cast<decltype(h.GetCodeSize)>(sym("hiprtcGetCodeSize"));
        h.GetCode = reinterpret_cast<decltype(h.GetCode)>(sym("hiprtcGetCode"));
        h.DestroyProgram = reinterpret_cast<decltype(h.DestroyProgram)>(sym("hiprtcDestroyProgram"));
        if (!h.CreateProgram || !h.AddNameExpression || !h.CompileProgram || !h.GetProgramLogSize || !h.GetProgramLog || !h.GetLoweredName ||
            !h.GetCodeSize || !h.GetCode || !h.DestroyProgram) {
            h.why = "libhiprtc.so lacks an entry point of the runtime compiler";
            dlclose(h.lib);
            h.lib = nullptr;
        }
    });
    return h;
}

// The options of the per-scene compile: those of the library's own build that decide what code a source line becomes
// (rust_raytracer_amd/build.py: HIPCC_FLAGS; a test holds the two lists together).
const char* const kJitOptions[] = {"--gpu-architecture=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize"};
constexpr uint32_t RR_JIT_MAX_OPS = 256;      // an unrolled program longer than this is not compiled at all
constexpr uint32_t RR_JIT_AUTO_MAX_OPS = 64;  // RR_SPECIALIZE_AUTO: nor one longer than this

struct JitCode {
    std::vector<char> code;
    std::string lowered;  // the kernel's symbol in the code object
};
std::mutex g_jit_mutex;
std::map<std::string, JitCode> g_jit_cache;  // compiled once per process and (template arguments, program)

// One header of the library without its #include <...> lines (the runtime compiler has no system headers, and needs none).
std::string without_system_includes(const char* text) {
    std::string out;
    for (const char* p = text; *p;) {
        const char* e = std::strchr(p, '\n');
        const size_t n = e ? static_cast<size_t>(e - p) + 1 : std::strlen(p);
        if (!(std::strncmp(p, "#include <", 10) == 0)) out.append(p, n);
        else out.append("\n");
        p += n;
    }
    return out;
}

// Compiles k_wf_paths<targs> for the program `words` (eight words per op).  On failure *log holds the compiler's output.
int jit_compile(const std::string& targs, const std::vector<uint32_t>& words, JitCode* out, std::string* log) {
    Hiprtc& H = hiprtc();
    if (!H.ok()) return fail(RR_E_NO_DEVICE, "rr_scene_specialize: " + H.why);
    const size_t n_ops = words.size() / 8;
    std::string fixed = "#define RR_FIXED_N_OPS " + std::to_string(n_ops) + "\nconstexpr uint32_t kFixedWords[" + std::to_string(n_ops) + "][8] = {\n";
    char buf[128];
    for (size_t i = 0; i < n_ops; ++i) {
        const uint32_t* w = &words[8 * i];
        std::snprintf(buf, sizeof(buf), "    {0x%08xu, 0x%08xu, 0x%08xu, 0x%08xu, 0x%08xu, 0x%08xu, 0x%08xu, 0x%08xu},\n", w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]);
        fixed += buf;
    }
    fixed += "};\n";
    const std::string main_src =
        "typedef unsigned char uint8_t;\ntypedef unsigned short uint16_t;\ntypedef unsigned int uint32_t;\ntypedef unsigned long long uint64_t;\n"
        "typedef signed char int8_t;\ntypedef short int16_t;\ntypedef int int32_t;\ntypedef long long int64_t;\ntypedef unsigned long size_t;\n"
        "#define RR_FIXED_PROGRAM \"rr_fixed_program.inc\"\n"
        "#include \"rr_device.h\"\nusing namespace rrdev;\n#include \"rr_wavefront.h\"\n"
        "constexpr int W5 = RR_RENDER_WAVES_PER_SIMD + 1;\n";
    const std::string dev_h = without_system_includes(rr_src_device_h), wf_h = without_system_includes(rr_src_wavefront_h),
                      api_h = without_system_includes(rr_src_api_h);
    const char* headers[] = {dev_h.c_str(), wf_h.c_str(), api_h.c_str(), fixed.c_str()};
    const char* names[] = {"rr_device.h", "rr_wavefront.h", "rr_hip.h", "rr_fixed_program.inc"};
    void* prog = nullptr;
    if (H.CreateProgram(&prog, main_src.c_str(), "rr_scene_kernel.hip", 4, headers, names) != 0) return fail(RR_E_HIP, "rr_scene_specialize: hiprtcCreateProgram failed");
    const std::string expr = "&k_wf_paths<" + targs + ">";
    int rc = RR_OK;
    if (H.AddNameExpression(prog, expr.c_str()) != 0) rc = fail(RR_E_HIP, "rr_scene_specialize: hiprtcAddNameExpression failed");
    if (rc == RR_OK) {
        const int cr = H.CompileProgram(prog, static_cast<int>(sizeof(kJitOptions) / sizeof(kJitOptions[0])), kJitOptions);
        size_t ls = 0;
        if (H.GetProgramLogSize(prog, &ls) == 0 && ls > 1) {
            log->assign(ls, '\0');
            (void)H.GetProgramLog(prog, &(*log)[0]);
        }
        if (cr != 0) rc = fail(RR_E_HIP, "rr_scene_specialize: the per-scene compile failed:\n" + log->substr(0, 4000));
    }
    if (rc == RR_OK) {
        const char* lowered = nullptr;
        size_t cs = 0;
        if (H.GetLoweredName(prog, expr.c_str(), &lowered) != 0 || !lowered || H.GetCodeSize(prog, &cs) != 0 || cs == 0) {
            rc = fail(RR_E_HIP, "rr_scene_specialize: the compiled program has no kernel " + expr);
        } else {
            out->lowered = lowered;
            out->code.resize(cs);
            if (H.GetCode(prog, out->code.data()) != 0) rc = fail(RR_E_HIP, "rr_scene_specialize: hiprtcGetCode failed");
        }
    }
    (void)H.DestroyProgram(&prog);
    return rc;
}

void jit_unload(rr_context* ctx) {
    if (ctx->jit_mod) (void)hipModuleUnload(ctx->jit_mod);
    ctx->jit_mod = nullptr;
    ctx->jit_fn = nullptr;
    ctx->jit_targs.clear();
    ctx->info.specialized = 0u;
    ctx->info.specialize_ms = 0.0f;
}

}  // namespace

// ================================================================== C ABI
extern "C" {

const char* rr_last_error(void) { return g_err.c_str(); }

int rr_device_count(int* count) {
    if (!count) return fail(RR_E_INVALID, "rr_device_count: count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) n = 0;  // hipErrorNoDevice on a host without a GPU
    *count = n;
    return RR_OK;
}

int rr_context_create(int device, rr_context** out) {
    if (!out) return fail(RR_E_INVALID, "rr_context_create: out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(RR_E_NO_DEVICE, "no HIP device visible; this library has no CPU path");
    if (device < 0 || device >= n) return fail(RR_E_NO_DEVICE, "device index out of range");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RR_E_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    HIP_TRY(hipSetDevice(device));
    rr_context* ctx = new rr_context();
    ctx->sw_no_fuse = std::getenv("RR_NO_FUSE") != nullptr;
    if (const char* pad = std::getenv("RR_PAD_OPS")) ctx->sw_pad_ops = std::max(0, std::min(std::atoi(pad), 1 << 16));
    ctx->sw_dump_program = std::getenv("RR_DUMP_PROGRAM") != nullptr;
    if (const char* env = std::getenv("RR_WF_WORKSPACE_BYTES")) ctx->sw_workspace_cap = std::max<size_t>(1, std::strtoull(env, nullptr, 10));
    ctx->device = device;
    ctx->n_cus = prop.multiProcessorCount;
    ctx->max_dynamic_lds = static_cast<int>(prop.sharedMemPerBlock);
    if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
        delete ctx;
        return fail(RR_E_HIP, "hipEventCreate failed");
    }
    // this device's copy of the static array behind rr_render_device_noalloc (every GPU has its own, at its own address)
    void* scratch_ptr = nullptr;
    const hipError_t se = hipGetSymbolAddress(&scratch_ptr, HIP_SYMBOL(scratch));
    if (se != hipSuccess || !scratch_ptr) {
        (void)hipEventDestroy(ctx->ev0);
        (void)hipEventDestroy(ctx->ev1);
        delete ctx;
        return fail(RR_E_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(se));
    }
    ctx->scratch = static_cast<unsigned char*>(scratch_ptr);
    *out = ctx;
    return RR_OK;
}

int rr_context_destroy(rr_context* ctx) {
    if (!ctx) return RR_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    jit_unload(ctx);
    ctx->scene_mem.release();
    if (ctx->wf_mem) (void)hipFree(ctx->wf_mem);
    if (ctx->wf_host_count) (void)hipHostFree(ctx->wf_host_count);
    if (ctx->next_item) (void)hipFree(ctx->next_item);
    if (ctx->wf_aux) (void)hipStreamDestroy(ctx->wf_aux);
    for (int k = 0; k < 2; ++k) {
        if (ctx->wf_ev_main[k]) (void)hipEventDestroy(ctx->wf_ev_main[k]);
        if (ctx->wf_ev_aux[k]) (void)hipEventDestroy(ctx->wf_ev_aux[k]);
    }
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    delete ctx;
    return RR_OK;
}

int rr_scene_upload(rr_context* ctx, const rr_flat_scene* s) {
    if (!ctx || !s) return fail(RR_E_INVALID, "rr_scene_upload: NULL argument");
    if (int rc = set_device(ctx)) return rc;
    // ---- validate
    const uint32_t counts[] = {s->n_spheres, s->n_moving_spheres, s->n_rects, s->n_triangles, s->n_cuboids,
                               s->n_media,   s->n_instances,      s->n_bvh_nodes, s->n_lists};
    const void* ptrs[] = {s->spheres, s->moving_spheres, s->rects, s->triangles, s->cuboids, s->media, s->instances, s->bvh_nodes, s->lists};
    for (int i = 0; i < 9; ++i) {
        if (counts[i] && !ptrs[i]) return fail(RR_E_INVALID, "rr_scene_upload: a non-empty array has a NULL pointer");
        if (counts[i] > IDX_MASK) return fail(RR_E_LIMIT, "rr_scene_upload: more than 2^26 records of one kind");
    }
    if ((s->n_child_refs && !s->child_refs) || (s->n_materials && !s->materials) || (s->n_textures && !s->textures) ||
        (s->n_perlin && !s->perlin) || (s->n_images && !s->images) || (s->n_image_bytes && !s->image_bytes))
        return fail(RR_E_INVALID, "rr_scene_upload: a non-empty array has a NULL pointer");
    if (s->semantics > RR_SEMANTICS_BOOK) return fail(RR_E_INVALID, "rr_scene_upload: unknown semantics");
    const bool sem_intended = (s->semantics & RR_SEMANTICS_INTENDED) != 0u, sem_book = (s->semantics & RR_SEMANTICS_BOOK_ESTIMATOR) != 0u;
    if (sem_book && s->n_lights != 0u && !s->lights) return fail(RR_E_INVALID, "rr_scene_upload: n_lights without lights");
    if (RR_REF_TAG(s->root) != RR_TAG_LIST || RR_REF_INDEX(s->root) >= s->n_lists)
        return fail(RR_E_INVALID, "rr_scene_upload: root must be a list ref");
    for (uint32_t i = 0; i < s->n_materials; ++i) {
        const rr_material& m = s->materials[i];
        if (m.type > RR_MAT_EMISSIVE) return fail(RR_E_INVALID, "material type out of range");
        if ((m.type == RR_MAT_LAMBERTIAN || m.type == RR_MAT_ISOTROPIC || m.type == RR_MAT_EMISSIVE) && m.texture >= s->n_textures)
            return fail(RR_E_INVALID, "material references a texture outside the table");
    }
    for (uint32_t i = 0; i < s->n_textures; ++i) {
        const rr_texture& t = s->textures[i];
        if (t.type > RR_TEX_IMAGE) return fail(RR_E_INVALID, "texture type out of range");
        if (t.type == RR_TEX_NOISE && t.index >= s->n_perlin) return fail(RR_E_INVALID, "noise texture references a Perlin table outside the array");
        if (t.type == RR_TEX_IMAGE && t.index >= s->n_images) return fail(RR_E_INVALID, "image texture references an image outside the array");
    }
    for (uint32_t i = 0; i < s->n_images; ++i) {
        const rr_image& im = s->images[i];
        const uint64_t need = static_cast<uint64_t>(im.width) * im.height * 3u;
        if (im.width == 0 || im.height == 0 || im.offset > s->n_image_bytes || s->n_image_bytes - im.offset < need)
            return fail(RR_E_INVALID, "image outside image_bytes");
    }
    Walker wk(*s);
    uint32_t stack_words = 0;
    Below everything;
    if (!wk.walk(s->root, NONE, 0, stack_words, everything)) return fail(wk.err_code, "rr_scene_upload: " + wk.err);
    stack_words = std::max(stack_words, 1u);
    const size_t lds_bytes = static_cast<size_t>(stack_words) * 256u * 4u;
    if (lds_bytes > static_cast<size_t>(ctx->max_dynamic_lds))
        return fail(RR_E_LIMIT, "rr_scene_upload: traversal stack of " + std::to_string(stack_words) +
                                    " words per lane exceeds the LDS available to one workgroup");

    // Margin of the culling test.  The reference's intersectors accept hits by rounded arithmetic, so a ray
    // can be accepted by a primitive it misses geometrically, by a distance that must stay inside the margin:
    //   rects, triangles: the hit point and the edge tests are off by some ulps of the coordinates involved,
    //     below 1e-6 * (|coordinates| + distance from the ray origin);
    //   spheres: the discriminant half_b^2 - a*c (collider.rs:282) carries an absolute error of about
    //     21 * 2^-24 * |d|^2 * |oc|^2, which accepts rays passing up to 0.65e-6 * |oc|^2 / r outside a sphere
    //     of radius r seen from distance |oc| >= r.
    // The kernels grow a node's box by cull_eps + cull_k * (largest squared distance from the ray origin to
    // the box): cull_eps = 2e-5 * (largest coordinate below any BVH node), an order of magnitude above the
    // first bound inside the scene; cull_k = max(4e-6 / smallest radius, 1e-6 / that coordinate), six times
    // the second bound and enough for the first at any distance.
    double extent = 1.0;
    for (const Below& b : wk.bvh_below)
        if (b.bounded && !b.empty())
            for (int a = 0; a < 3; ++a) extent = std::max({extent, std::fabs(b.lo[a]), std::fabs(b.hi[a])});
    double r_min = 1e300;
    for (uint32_t i = 0; i < s->n_spheres; ++i) r_min = std::min(r_min, std::fabs(static_cast<double>(s->spheres[i].radius)));
    for (uint32_t i = 0; i < s->n_moving_spheres; ++i) r_min = std::min(r_min, std::fabs(static_cast<double>(s->moving_spheres[i].radius)));
    const double cull_eps = 2e-5 * extent;
    const double cull_k = std::max(r_min > 0.0 ? 4e-6 / r_min : 1e300, 1e-6 / extent);

    // ---- pack
    // The traversal machine reads three 16-byte words at every record it visits, whatever the
    // record's length: the arrays it reads that way end in three words of padding and are never empty.
    constexpr size_t PAD = 3;
    std::vector<float4> bvh(static_cast<size_t>(s->n_bvh_nodes) * 2 + PAD);
    uint32_t class_count[4] = {0, 0, 0, 0};
    std::vector<uint8_t> bvh_class(s->n_bvh_nodes, 0);
    for (uint32_t i = 0; i < s->n_bvh_nodes; ++i) {
        const rr_bvh_node& n = s->bvh_nodes[i];
        bvh[2 * i] = make_float4(n.bmin[0], n.bmin[1], n.bmin[2], u2f(to_dref(n.left, n.axis)));
        // How the production kernels may treat the node (the counting kernels treat every node as the
        // reference does).  If no constant medium lies below and the stored box holds everything below
        // (within half the margin the kernels add), a ray that misses the box cannot be accepted by
        // anything below: the subtree need not be entered, and no visit of it needs repeating.
        const Below& b = wk.bvh_below[i];
        uint32_t cls = BVH_CLASS_NO_BOX;
        if (n.has_box) {
            cls = BVH_CLASS_REFERENCE;
            bool holds = !b.has_medium && b.bounded && !b.empty();
            for (int a = 0; a < 3 && holds; ++a)
                holds = static_cast<double>(n.bmin[a]) - 0.5 * cull_eps <= b.lo[a] && b.hi[a] <= static_cast<double>(n.bmax[a]) + 0.5 * cull_eps;
            if (holds) cls = b.has_tri ? BVH_CLASS_CULL_ANY_T : BVH_CLASS_CULL_RANGE;
            // under the labelled intended semantics the class only says whether a repeated visit could draw random numbers
            if (sem_intended) cls = b.has_medium ? BVH_CLASS_REFERENCE : BVH_CLASS_CULL_RANGE;
        }
        bvh[2 * i + 1] = make_float4(n.bmax[0], n.bmax[1], n.bmax[2], u2f(to_dref(n.right, cls)));
        bvh_class[i] = static_cast<uint8_t>(cls);
        if (wk.seen[RR_TAG_BVH][i]) class_count[cls]++;
    }
    // The wide node move (trav_node_wide, the wavefront pipeline's traversal kernel) tests the boxes of a node's two children
    // from one 64-byte record of the node.  It is for scenes under the reference's semantics in which every BVH node of the
    // tree is of a culling class and has nothing but BVH nodes, spheres, rects, triangles and cuboids below it; such scenes
    // get the records when they are ones the pipeline renders, the others keep the one-box-per-visit moves.
    bool wide_scene = !sem_intended && s->n_bvh_nodes != 0;
    for (uint32_t i = 0; i < s->n_bvh_nodes && wide_scene; ++i)
        if (wk.seen[RR_TAG_BVH][i])
            wide_scene = bvh_class[i] >= BVH_CLASS_CULL_RANGE && wk.bvh_below[i].simple;
    std::vector<float4> wide;
    if (wide_scene) {
        wide.resize(static_cast<size_t>(s->n_bvh_nodes) * 4);
        for (uint32_t i = 0; i < s->n_bvh_nodes; ++i) {
            const rr_bvh_node& n = s->bvh_nodes[i];
            if (!wk.seen[RR_TAG_BVH][i]) {  // not part of the tree: never visited, its refs were never checked
                for (int k = 0; k < 4; ++k) wide[4 * static_cast<size_t>(i) + k] = make_float4(0.f, 0.f, 0.f, 0.f);
                continue;
            }
            const uint32_t child[2] = {n.left, n.right};
            for (int c = 0; c < 2; ++c) {
                float lo[3] = {0.f, 0.f, 0.f}, hi[3] = {0.f, 0.f, 0.f};
                uint32_t cls = 0;  // a primitive: no box to test
                if (RR_REF_TAG(child[c]) == RR_TAG_BVH) {
                    const rr_bvh_node& cn = s->bvh_nodes[RR_REF_INDEX(child[c])];
                    for (int a = 0; a < 3; ++a) lo[a] = cn.bmin[a], hi[a] = cn.bmax[a];
                    cls = bvh_class[RR_REF_INDEX(child[c])];
                }
                wide[4 * static_cast<size_t>(i) + 2 * c] = make_float4(lo[0], lo[1], lo[2], u2f(to_dref(child[c], c == 0 ? n.axis : 0u)));
                wide[4 * static_cast<size_t>(i) + 2 * c + 1] = make_float4(hi[0], hi[1], hi[2], u2f(cls));
            }
        }
    }
    std::vector<float4> sphere(s->n_spheres + PAD);
    std::vector<uint2> sphere_meta(s->n_spheres);
    ctx->id_sphere.resize(s->n_spheres);
    for (uint32_t i = 0; i < s->n_spheres; ++i) {
        const rr_sphere& p = s->spheres[i];
        sphere[i] = make_float4(p.center[0], p.center[1], p.center[2], p.radius);
        sphere_meta[i] = make_uint2(p.material, wk.owner(wk.owner_sphere[i]));
        ctx->id_sphere[i] = p.prim_id;
    }
    std::vector<float4> msphere(static_cast<size_t>(s->n_moving_spheres) * 3 + PAD);
    ctx->id_msphere.resize(s->n_moving_spheres);
    for (uint32_t i = 0; i < s->n_moving_spheres; ++i) {
        const rr_moving_sphere& p = s->moving_spheres[i];
        msphere[3 * i] = make_float4(p.center0[0], p.center0[1], p.center0[2], p.time0);
        msphere[3 * i + 1] = make_float4(p.center1[0], p.center1[1], p.center1[2], p.time1);
        msphere[3 * i + 2] = make_float4(p.radius, u2f(p.material), u2f(wk.owner(wk.owner_msphere[i])), 0.f);
        ctx->id_msphere[i] = p.prim_id;
    }
    std::vector<float4> rect(static_cast<size_t>(s->n_rects) * 2 + PAD);
    ctx->id_rect.resize(s->n_rects);
    for (uint32_t i = 0; i < s->n_rects; ++i) {
        const rr_rect& p = s->rects[i];
        rect[2 * i] = make_float4(p.a0, p.a1, p.b0, p.b1);
        rect[2 * i + 1] = make_float4(p.k, u2f(p.axis), u2f(p.material), u2f(wk.owner(wk.owner_rect[i])));
        ctx->id_rect[i] = p.prim_id;
    }
    std::vector<float4> tri(static_cast<size_t>(s->n_triangles) * 3 + PAD);
    std::vector<uint2> tri_meta(s->n_triangles);
    ctx->id_tri.resize(s->n_triangles);
    for (uint32_t i = 0; i < s->n_triangles; ++i) {
        const rr_triangle& p = s->triangles[i];
        tri[3 * i] = make_float4(p.p0[0], p.p0[1], p.p0[2], p.p1[0]);
        tri[3 * i + 1] = make_float4(p.p1[1], p.p1[2], p.p2[0], p.p2[1]);
        tri[3 * i + 2] = make_float4(p.p2[2], p.normal[0], p.normal[1], p.normal[2]);
        tri_meta[i] = make_uint2(p.material, wk.owner(wk.owner_tri[i]));
        ctx->id_tri[i] = p.prim_id;
    }
    std::vector<float4> cuboid(static_cast<size_t>(s->n_cuboids) * 2 + PAD);
    bool all_canonical = true;
    for (uint32_t i = 0; i < s->n_cuboids; ++i) {
        // Corners of the box, if the six rects are exactly what Cuboid::new (collider.rs:419-426)
        // makes of two corners; the machine then tests the sides from these 32 bytes alone.
        if (!wk.seen[RR_TAG_CUBOID][i]) continue;  // not part of the tree: never visited, range never checked
        float p0[3], p1[3];
        const bool canonical = cuboid_corners(*s, i, p0, p1);
        all_canonical &= canonical;
        cuboid[2 * i] = make_float4(p0[0], p0[1], p0[2], u2f(s->cuboids[i].first_rect));
        cuboid[2 * i + 1] = make_float4(p1[0], p1[1], p1[2], u2f(canonical ? 1u : 0u));
    }
    std::vector<uint4> medium(s->n_media);
    ctx->id_medium.resize(s->n_media);
    for (uint32_t i = 0; i < s->n_media; ++i) {
        const rr_medium& p = s->media[i];
        uint32_t nid;
        std::memcpy(&nid, &p.neg_inv_density, 4);
        medium[i] = make_uint4(to_dref(p.boundary), nid, p.material, wk.owner(wk.owner_medium[i]));
        ctx->id_medium[i] = p.prim_id;
    }
    auto inst_record = [&](uint32_t i) {
        const rr_instance& p = s->instances[i];
        return p.kind == RR_INST_TRANSLATE ? make_float4(u2f(p.kind), p.v[0], p.v[1], p.v[2]) : make_float4(u2f(p.kind), p.v[0], p.v[1], 0.f);
    };
    std::vector<float4> inst(static_cast<size_t>(s->n_instances) * 2);
    // (never empty: reconstruct_hit<true> reads the first two words of a chain whatever the primitive's depth)
    std::vector<float4> chain_rec(static_cast<size_t>(std::max(s->n_instances, 1u)) * MAX_CHAIN, make_float4(0.f, 0.f, 0.f, 0.f));
    for (uint32_t i = 0; i < s->n_instances; ++i) {
        inst[2 * i] = inst_record(i);
        inst[2 * i + 1] = make_float4(u2f(to_dref(s->instances[i].child)), u2f(wk.inst_depth[i]), 0.f, 0.f);
        for (uint32_t k = 0; k < wk.inst_depth[i]; ++k) chain_rec[static_cast<size_t>(i) * MAX_CHAIN + k] = inst_record(wk.inst_chain[i * MAX_CHAIN + k]);
    }
    std::vector<uint2> list(s->n_lists);
    for (uint32_t i = 0; i < s->n_lists; ++i) list[i] = make_uint2(s->lists[i].first_child, s->lists[i].n_children);
    std::vector<uint32_t> child_refs(s->n_child_refs);
    for (uint32_t i = 0; i < s->n_child_refs; ++i) child_refs[i] = to_dref(s->child_refs[i]);
    std::vector<DevMaterial> materials(s->n_materials);
    for (uint32_t i = 0; i < s->n_materials; ++i) {
        const rr_material& m = s->materials[i];
        DevMaterial dm{m.type, m.texture, m.color[0], m.color[1], m.color[2], m.param, 0, 0};
        const bool textured = m.type == RR_MAT_LAMBERTIAN || m.type == RR_MAT_ISOTROPIC || m.type == RR_MAT_EMISSIVE;
        if (textured && s->textures[m.texture].type == RR_TEX_SOLID) {
            const rr_texture& t = s->textures[m.texture];
            dm.r = t.color0[0], dm.g = t.color0[1], dm.b = t.color0[2];
            dm.solid = 1;
        }
        materials[i] = dm;
    }
    std::vector<DevTexture> textures(s->n_textures);
    ctx->tex_types.resize(s->n_textures);
    for (uint32_t i = 0; i < s->n_textures; ++i) {
        const rr_texture& t = s->textures[i];
        ctx->tex_types[i] = t.type;
        DevTexture d{};
        d.type = t.type;
        d.index = t.index;
        d.depth = t.depth;
        d.scale = t.scale;
        for (int k = 0; k < 3; ++k) {
            d.c0[k] = t.color0[k];
            d.c1[k] = t.color1[k];
        }
        textures[i] = d;
    }
    std::vector<float> perlin_vec(static_cast<size_t>(s->n_perlin) * 768);
    std::vector<uint8_t> perlin_perm(static_cast<size_t>(s->n_perlin) * 768);
    for (uint32_t i = 0; i < s->n_perlin; ++i) {
        std::memcpy(&perlin_vec[static_cast<size_t>(i) * 768], s->perlin[i].ranvec, 768 * sizeof(float));
        std::memcpy(&perlin_perm[static_cast<size_t>(i) * 768], s->perlin[i].perm_x, 256);
        std::memcpy(&perlin_perm[static_cast<size_t>(i) * 768 + 256], s->perlin[i].perm_y, 256);
        std::memcpy(&perlin_perm[static_cast<size_t>(i) * 768 + 512], s->perlin[i].perm_z, 256);
    }
    std::vector<DevImage> images(s->n_images);
    for (uint32_t i = 0; i < s->n_images; ++i) images[i] = DevImage{s->images[i].width, s->images[i].height, s->images[i].offset};
    std::vector<uint8_t> image_bytes(s->image_bytes, s->image_bytes + s->n_image_bytes);
    std::vector<float4> light_list;  // (read by the book estimator only)
    if (sem_book)
        for (uint32_t i = 0; i < s->n_lights; ++i) {
            const rr_light& l = s->lights[i];
            light_list.push_back(make_float4(l.x0, l.x1, l.z0, l.z1));
            light_list.push_back(make_float4(l.k, u2f(l.is_xzrect ? 1u : 0u), 0.f, 0.f));
        }
    ProgramBuilder pb{*s, wk.bvh_below, bvh_class, {}, {}, {}, false, !ctx->sw_no_fuse};
    pb.emit(s->root);
    if (pb.fuse) pb.fuse_transforms();
    for (int i = 0; i < ctx->sw_pad_ops; ++i) pb.ops.push_back(ProgramBuilder::make(OP_XFORM));  // (for measurements: prices an op's way through the interpreter)
    const uint32_t n_ops = static_cast<uint32_t>(pb.ops.size());
    pb.ops.push_back(ProgramBuilder::make(OP_END));
    pb.ops.push_back(ProgramBuilder::make(OP_END));  // the interpreter prefetches one op past the one it runs
    if (ctx->sw_dump_program) {  // for whoever tunes the program: its ops, one per line, to stderr
        static const char* const names[] = {"END", "-", "-", "-", "-", "-", "RECT_XY", "RECT_XZ", "RECT_YZ", "SPHERE", "LEAF_REF",
                                            "SUBTREE", "MED_BEGIN", "MED_MID", "MED_END", "CUBOID", "SUBTREE_SIMPLE", "MED_SPHERE", "XFORM"};
        for (size_t i = 0; i < pb.ops.size(); ++i) {
            const Op& o = pb.ops[i];
            uint32_t w[8];
            std::memcpy(w, &o, sizeof(w));
            std::fprintf(stderr, "op %3zu %-14s flags %02x ref %08x  %g %g %g %g %g %g  words %08x %08x %08x %08x %08x %08x %08x %08x\n", i,
                         names[std::min<uint32_t>(o.code & 0xFFu, 18u)], o.code >> 8, o.ref, o.f0, o.f1, o.f2, o.f3, o.f4, o.f5, w[0], w[1], w[2], w[3], w[4], w[5],
                         w[6], w[7]);
        }
    }
    static_assert(sizeof(Op) == 32, "one op is two 16-byte words");
    std::vector<uint4> prog(pb.ops.size() * 2);
    std::memcpy(prog.data(), pb.ops.data(), pb.ops.size() * sizeof(Op));

    // ---- upload (replacing any previous scene)
    (void)hipDeviceSynchronize();
    jit_unload(ctx);  // (a kernel compiled for the previous scene's program)
    ctx->prog_words.assign(reinterpret_cast<const uint32_t*>(pb.ops.data()), reinterpret_cast<const uint32_t*>(pb.ops.data()) + 8u * n_ops);
    ctx->scene_mem.release();
    ctx->has_scene = false;
    DevScene d{};
    DeviceBuffers& mem = ctx->scene_mem;
    int rc = RR_OK;
    std::vector<DevConst> kc(1);
    DevConst& k = kc[0];
    const uint4* prog_dev = nullptr;
    const size_t geometry_bytes = (bvh.size() + sphere.size() + msphere.size() + rect.size() + tri.size() + cuboid.size()) * sizeof(float4);
    if ((rc = upload(mem, bvh, &d.bvh)) || (rc = upload(mem, wide, &d.wide)) || (rc = upload(mem, sphere, &d.sphere)) || (rc = upload(mem, sphere_meta, &k.sphere_meta)) ||
        (rc = upload(mem, msphere, &d.msphere)) || (rc = upload(mem, rect, &d.rect)) || (rc = upload(mem, tri, &d.tri)) ||
        (rc = upload(mem, tri_meta, &k.tri_meta)) || (rc = upload(mem, cuboid, &d.cuboid)) || (rc = upload(mem, medium, &d.medium)) ||
        (rc = upload(mem, inst, &d.inst)) || (rc = upload(mem, chain_rec, &d.chain_rec)) || (rc = upload(mem, list, &d.list)) ||
        (rc = upload(mem, child_refs, &d.child_refs)) || (rc = upload(mem, materials, &k.materials)) ||
        (rc = upload(mem, textures, &k.textures)) || (rc = upload(mem, perlin_vec, &k.perlin_vec)) ||
        (rc = upload(mem, perlin_perm, &k.perlin_perm)) || (rc = upload(mem, images, &k.images)) ||
        (rc = upload(mem, image_bytes, &k.image_bytes)) || (rc = upload(mem, light_list, &k.lights)) || (rc = upload(mem, prog, &prog_dev))) {
        mem.release();
        return rc;
    }
    d.prog = (decltype(d.prog))prog_dev;  // C-style cast: the member is address-space qualified in the device pass
    d.n_ops = n_ops;
    k.light_is_xzrect = s->light.is_xzrect ? 1u : 0u;
    k.light_x0 = s->light.x0;
    k.light_x1 = s->light.x1;
    k.light_z0 = s->light.z0;
    k.light_z1 = s->light.z1;
    k.light_k = s->light.k;
    k.n_lights = static_cast<uint32_t>(light_list.size() / 2);
    k.cam = s->camera;
    for (int a = 0; a < 3; ++a) k.background[a] = s->background[a];
    const DevConst* k_dev = nullptr;
    if ((rc = upload(mem, kc, &k_dev))) {
        mem.release();
        return rc;
    }
    d.k = (decltype(d.k))k_dev;
    ctx->sc = d;
    ctx->info = rr_scene_info{};
    ctx->info.stack_words = stack_words;
    ctx->info.max_inst_depth = wk.max_inst_depth;
    ctx->info.device_bytes = mem.bytes;
    ctx->info.n_prims = s->n_spheres + s->n_moving_spheres + s->n_rects + s->n_triangles;
    ctx->has_triangles = s->n_triangles != 0;
    ctx->wide_scene = wide_scene;
    ctx->plain_scene = s->n_moving_spheres == 0 && s->n_media == 0 && s->n_instances == 0;
    ctx->info.wide_nodes = wide_scene ? 1u : 0u;
    ctx->lean_scene = all_canonical && wk.max_inst_depth <= 2;
    ctx->has_subtrees = false;
    ctx->has_general_subtrees = false;
    ctx->prog_vector_ops = false;
    uint32_t top_ops = 0;  // ops of the scene program that test or transform something outside the BVH subtrees
    for (const Op& op : pb.ops) {
        const uint32_t code = op.code & 0xFFu;
        const bool subtree = code == OP_SUBTREE || code == OP_SUBTREE_SIMPLE;
        ctx->has_subtrees |= subtree;
        ctx->has_general_subtrees |= code == OP_SUBTREE;
        if (subtree) ctx->info.subtree_ops[code == OP_SUBTREE ? 1 : 0]++;
        if (code == OP_MED_BEGIN || code == OP_MED_MID || code == OP_MED_END) ctx->lean_scene = false;
        ctx->prog_vector_ops |= subtree || code == OP_LEAF_REF;
        // (the fused ops count as the ops they stand for)
        if (code == OP_MED_SPHERE) top_ops += 5u;
        else if (code == OP_XFORM) top_ops += ((op.code & OPX_TRANSLATE) ? 1u : 0u) + (((op.code >> OPX_ROT_SHIFT) & 3u) ? 1u : 0u);
        else top_ops += !subtree && code != OP_END;
    }
    // Scenes with BVH subtrees: when nearly everything is inside the subtrees, the bounce kernels of the wavefront
    // pipeline (pure traversal at twice the occupancy) are the faster way through the camera rays; when the scene
    // program has work of its own, the path-granular persistent kernel is (measured: random spheres 2770 against
    // 2580-2610 Msamples/s, final_scene 1200 against 1155-1175, Cornell box with the Suzanne mesh 2175 against 2800-2830).
    // (Since the path kernel stopped spilling and went to 5 waves it is also ahead on small all-BVH scenes: random spheres 4630
    // against 3965 Msamples/s at 1000 spp.  The pipeline keeps the larger ones, whose traversal waits for memory: measured on
    // height fields with spheres of 0.5 / 2 / 5.6 / 20 / 48 / 95 MB (1280x720, 16 spp), path kernel against pipeline: 8.1 / 8.1,
    // 19.9 / 16.5, 54.6 / 40.8, 156 / 100, 358 / 210 ms, and 31.2 against 40.9 Msamples/s on the 1.1 M-primitive mesh.
    // Measured again after the path kernel ended dead paths and took its pixels in tiles and the pipeline's traversal went
    // cohort by cohort (profiles/crossover.py; 0.57 / 2.3 / 5.8 / 20 / 56 / 111 MB on the device, of which the records counted
    // here are 0.25 / 1.0 / 2.8 / 10 / 26 / 51 MB): 5.4 / 8.6, 9.6 / 14.0, 25.2 / 18.8, 61.9 / 27.7, 172 / 63, 371 / 127 ms.)
    // (the threshold is about where the BVHs and primitives stop being cache-resident — with the records of the wide node move,
    // which double the nodes, one die's L2 of 4 MB: textures and tables do not count)
    ctx->auto_pipeline = ctx->has_subtrees && top_ops < 4 && geometry_bytes > (2u << 20);
    if (ctx->has_subtrees && lds_bytes + 4u * PARK_LDS_WORDS_PER_WAVE * 4u > static_cast<size_t>(ctx->max_dynamic_lds)) {
        ctx->scene_mem.release();
        return fail(RR_E_LIMIT, "rr_scene_upload: traversal stacks of " + std::to_string(stack_words) + " words per lane (" + std::to_string(lds_bytes) +
                                    " bytes per workgroup) plus " + std::to_string(4u * PARK_LDS_WORDS_PER_WAVE * 4u) +
                                    " bytes for the slots of parked lanes exceed the " + std::to_string(ctx->max_dynamic_lds) +
                                    " bytes of LDS available to one workgroup");
    }
    ctx->root = to_dref(s->root);
    ctx->sc.intended = sem_intended ? 1u : 0u;
    ctx->sc.book = sem_book ? 1u : 0u;
    ctx->sc.no_tri = ctx->sc.lean = 0u;
    ctx->sc.cull_eps = static_cast<float>(cull_eps);
    ctx->sc.cull_k = static_cast<float>(std::min(cull_k, 1e30));  // a zero radius leaves a margin no box is missed by
    ctx->info.wavefront = ctx->auto_pipeline ? 1u : 0u;
    for (int c = 0; c < 4; ++c) ctx->info.bvh_nodes_by_class[c] = class_count[c];
    ctx->has_scene = true;
    return RR_OK;
}

int rr_scene_info_get(rr_context* ctx, rr_scene_info* out) {
    if (!ctx || !out) return fail(RR_E_INVALID, "rr_scene_info_get: NULL argument");
    if (!ctx->has_scene) return fail(RR_E_NO_SCENE, "no scene uploaded");
    *out = ctx->info;
    out->workspace_bytes = ctx->wf_bytes + (ctx->next_item ? sizeof(unsigned long long) : 0u);
    return RR_OK;
}

int rr_scene_specialize(rr_context* ctx, uint32_t mode) {
    if (!ctx) return fail(RR_E_INVALID, "rr_scene_specialize: NULL argument");
    if (!ctx->has_scene) return fail(RR_E_NO_SCENE, "rr_scene_specialize: no scene uploaded");
    if (mode > RR_SPECIALIZE_OFF) return fail(RR_E_INVALID, "rr_scene_specialize: unknown mode");
    if (int rc = set_device(ctx)) return rc;
    if (mode == RR_SPECIALIZE_OFF) {
        HIP_TRY(hipDeviceSynchronize());
        jit_unload(ctx);
        return RR_OK;
    }
    const uint32_t n_ops = static_cast<uint32_t>(ctx->prog_words.size() / 8);
    if (mode == RR_SPECIALIZE_AUTO && (ctx->has_subtrees || n_ops == 0 || n_ops > RR_JIT_AUTO_MAX_OPS)) return RR_OK;  // not expected to pay
    if (n_ops == 0) return RR_OK;  // (an empty scene has no program to compile)
    if (n_ops > RR_JIT_MAX_OPS)
        return fail(RR_E_LIMIT, "rr_scene_specialize: a program of " + std::to_string(n_ops) + " ops is not unrolled (at most " + std::to_string(RR_JIT_MAX_OPS) + ")");
    const PathsChoice choice = choose_paths_kernel(ctx, COUNT_OFF);
    if (ctx->jit_fn && ctx->jit_targs == choice.targs) return RR_OK;
    const auto t0 = std::chrono::steady_clock::now();
    std::string key = std::string(choice.targs) + "|";
    key.append(reinterpret_cast<const char*>(ctx->prog_words.data()), ctx->prog_words.size() * sizeof(uint32_t));
    JitCode code;
    {
        std::lock_guard<std::mutex> lock(g_jit_mutex);
        auto it = g_jit_cache.find(key);
        if (it == g_jit_cache.end()) {
            std::string log;
            if (int rc = jit_compile(choice.targs, ctx->prog_words, &code, &log)) return rc;
            g_jit_cache[key] = code;
        } else {
            code = it->second;
        }
    }
    HIP_TRY(hipDeviceSynchronize());
    jit_unload(ctx);
    hipModule_t mod = nullptr;
    HIP_TRY(hipModuleLoadData(&mod, code.code.data()));
    hipFunction_t fn = nullptr;
    const hipError_t e = hipModuleGetFunction(&fn, mod, code.lowered.c_str());
    if (e != hipSuccess || !fn) {
        (void)hipModuleUnload(mod);
        return fail(RR_E_HIP, std::string("rr_scene_specialize: the code object has no function ") + code.lowered);
    }
    ctx->jit_mod = mod;
    ctx->jit_fn = fn;
    ctx->jit_targs = choice.targs;
    ctx->info.specialized = 1u;
    ctx->info.specialize_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return RR_OK;
}

/* The options rr_scene_specialize compiles with, space separated (for checks against the library's own build). */
const char* rr_specialize_options(void) {
    static const std::string joined = [] {
        std::string j;
        for (const char* o : kJitOptions) j += (j.empty() ? "" : " ") + std::string(o);
        return j;
    }();
    return joined.c_str();
}

/* Compiles a two-op program (a sphere and a rect) with the runtime compiler, without a device: a check that the device
 * sources inside the library still compile that way.  ms: the time the compile took. */
int rr_specialize_selftest(float* ms) {
    std::vector<uint32_t> words(16, 0u);
    words[0] = OP_SPHERE, words[1] = 0u, words[5] = 0x3f800000u;            // unit sphere at the origin
    words[8] = OP_RECT_XZ, words[9] = 0u, words[11] = 0x3f800000u, words[13] = 0x3f800000u, words[14] = 0x40000000u;
    JitCode code;
    std::string log;
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = jit_compile("COUNT_OFF, false, false, true, true, W5, true", words, &code, &log);
    if (ms) *ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rc == RR_OK && (code.code.empty() || code.lowered.empty())) return fail(RR_E_HIP, "rr_specialize_selftest: empty code object");
    return rc;
}

int rr_primary_hits(rr_context* ctx, uint32_t n, const float* rays, float t_min, float t_max, uint64_t seed, rr_hit* out) {
    if (!ctx || (n && (!rays || !out))) return fail(RR_E_INVALID, "rr_primary_hits: NULL argument");
    if (!ctx->has_scene) return fail(RR_E_NO_SCENE, "rr_primary_hits: no scene uploaded");
    if (n == 0) return RR_OK;
    if (int rc = set_device(ctx)) return rc;
    float *d_rays = nullptr, *d_out = nullptr;
    uint32_t* d_ref = nullptr;
    auto cleanup = [&]() {
        if (d_rays) (void)hipFree(d_rays);
        if (d_out) (void)hipFree(d_out);
        if (d_ref) (void)hipFree(d_ref);
    };
    hipError_t e;
    if ((e = hipMalloc(&d_rays, static_cast<size_t>(n) * 7 * 4)) != hipSuccess || (e = hipMalloc(&d_out, static_cast<size_t>(n) * 12 * 4)) != hipSuccess ||
        (e = hipMalloc(&d_ref, static_cast<size_t>(n) * 4)) != hipSuccess ||
        (e = hipMemcpy(d_rays, rays, static_cast<size_t>(n) * 7 * 4, hipMemcpyHostToDevice)) != hipSuccess) {
        cleanup();
        return fail(RR_E_HIP, std::string("rr_primary_hits: ") + hipGetErrorString(e));
    }
    const size_t lds = static_cast<size_t>(ctx->info.stack_words) * 256u * 4u;
    const uint32_t grid = (n + 255u) / 256u;
    hipLaunchKernelGGL(k_primary_hits, dim3(grid), dim3(256), lds, 0, ctx->sc, n, d_rays, t_min, t_max, seed, d_out, d_ref);
    std::vector<float> hf(static_cast<size_t>(n) * 12);
    std::vector<uint32_t> hr(n);
    if ((e = hipGetLastError()) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess ||
        (e = hipMemcpy(hf.data(), d_out, hf.size() * 4, hipMemcpyDeviceToHost)) != hipSuccess ||
        (e = hipMemcpy(hr.data(), d_ref, hr.size() * 4, hipMemcpyDeviceToHost)) != hipSuccess) {
        cleanup();
        return fail(RR_E_HIP, std::string("rr_primary_hits: ") + hipGetErrorString(e));
    }
    cleanup();
    for (uint32_t i = 0; i < n; ++i) {
        rr_hit& h = out[i];
        std::memset(&h, 0, sizeof(h));
        const uint32_t ref = hr[i];
        if (ref == NONE) {
            h.prim_id = -1;
            continue;
        }
        const uint32_t tag = dtag(ref), idx = didx(ref);
        uint32_t pid = 0;
        switch (tag) {
            case RR_TAG_SPHERE: pid = ctx->id_sphere[idx]; break;
            case RR_TAG_MOVING_SPHERE: pid = ctx->id_msphere[idx]; break;
            case RR_TAG_RECT: pid = ctx->id_rect[idx]; break;
            case RR_TAG_TRIANGLE: pid = ctx->id_tri[idx]; break;
            default: pid = ctx->id_medium[idx]; break;
        }
        const float* f = &hf[static_cast<size_t>(i) * 12];
        h.prim_id = static_cast<int32_t>(pid);
        h.t = f[0];
        std::memcpy(h.point, f + 1, 12);
        std::memcpy(h.normal, f + 4, 12);
        h.u = f[7];
        h.v = f[8];
        std::memcpy(&h.front_face, f + 9, 4);
        std::memcpy(&h.material, f + 10, 4);
    }
    return RR_OK;
}

}  // extern "C"

namespace {
int render_device(rr_context* ctx, const rr_render_params* p, void* d_slice_sums, void* d_resolved, void* d_linear, void* d_counters,
                  void* hip_stream, bool noalloc) {
    if (!ctx || !p || !d_slice_sums) return fail(RR_E_INVALID, "rr_render_device: NULL argument");
    if (!ctx->has_scene) return fail(RR_E_NO_SCENE, "rr_render_device: no scene uploaded");
    if (p->width == 0 || p->height == 0 || p->n_slices == 0 || p->slice_begin > p->slice_end || p->slice_end > p->n_slices)
        return fail(RR_E_INVALID, "rr_render_device: bad image size or slice range");
    if (static_cast<uint64_t>(p->width) * p->height > 0xFFFFFFFFull) return fail(RR_E_LIMIT, "rr_render_device: more than 2^32 pixels");
    uint32_t row0 = 0, rows = 0;
    if (!band_of(p, &row0, &rows)) return fail(RR_E_INVALID, "rr_render_device: bad row range");
    if (int rc = set_device(ctx)) return rc;
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    const uint32_t n_sl = p->slice_end - p->slice_begin;
    const uint32_t spt = p->samples_per_pixel / p->n_slices;
    if ((p->flags & RR_FLAG_COUNTERS) && (p->flags & RR_FLAG_COUNT_PRODUCTION))
        return fail(RR_E_INVALID, "rr_render_device: RR_FLAG_COUNTERS and RR_FLAG_COUNT_PRODUCTION exclude each other");
    const int count = (p->flags & RR_FLAG_COUNTERS) ? COUNT_REFERENCE : (p->flags & RR_FLAG_COUNT_PRODUCTION) ? COUNT_PRODUCTION : COUNT_OFF;
    if (((p->flags & RR_FLAG_PERSISTENT) != 0) + ((p->flags & RR_FLAG_WAVEFRONT) != 0) + ((p->flags & RR_FLAG_PATHS) != 0) > 1)
        return fail(RR_E_INVALID, "rr_render_device: RR_FLAG_PERSISTENT, RR_FLAG_WAVEFRONT and RR_FLAG_PATHS exclude each other");
    // the persistent kernel at path granularity (k_wf_paths) unless the scene is one for the pipeline or a flag asks for something else
    if (noalloc && (p->flags & (RR_FLAG_PERSISTENT | RR_FLAG_WAVEFRONT | RR_FLAG_NO_FINISH_KERNEL)))
        return fail(RR_E_INVALID, "rr_render_device_noalloc renders with the path-granular kernel only: RR_FLAG_PERSISTENT, RR_FLAG_WAVEFRONT and "
                                  "RR_FLAG_NO_FINISH_KERNEL do not apply");
    const bool wavefront = !noalloc && ((p->flags & RR_FLAG_WAVEFRONT) || (ctx->auto_pipeline && !(p->flags & (RR_FLAG_PERSISTENT | RR_FLAG_PATHS))));
    const bool paths = noalloc || (!wavefront && !(p->flags & RR_FLAG_PERSISTENT));
    if (rows == 0) {  // an empty band: nothing to render, nothing to write
        HIP_TRY(hipEventRecord(ctx->ev0, stream));
        HIP_TRY(hipEventRecord(ctx->ev1, stream));
        ctx->timed = true;
        return RR_OK;
    }
    if (wavefront || paths) {
        const uint64_t n_pixels = static_cast<uint64_t>(p->width) * rows;
        if (n_pixels > (1ull << 30)) return fail(RR_E_LIMIT, "rr_render_device: more than 2^30 pixels in one wavefront pass");
        WfArgs w{};
        w.width = p->width;
        w.height = p->height;
        w.n_pixels = static_cast<uint32_t>(n_pixels);
        w.pixel0 = row0 * p->width;
        w.stack_words = ctx->info.stack_words;
        w.spt = spt;
        w.max_depth = p->max_depth;
        w.sample_offset = p->sample_offset;
        w.trace_dead = (p->flags & RR_FLAG_TRACE_DEAD_PATHS) ? 1u : 0u;
        w.seed = p->seed;
        // slices go through in groups of at most 2^27 slots (72 bytes of path state and queues each); without allocations, of
        // as many slots as the static scratch holds sample values for
        if (noalloc && n_pixels > RR_SCRATCH_PATHS)
            return fail(RR_E_LIMIT, "rr_render_device_noalloc: " + std::to_string(n_pixels) + " pixels in one call, the static workspace holds " +
                                        std::to_string(RR_SCRATCH_PATHS) + " (render the image in bands of rows)");
        const uint32_t group = static_cast<uint32_t>(std::max<uint64_t>(1, (noalloc ? RR_SCRATCH_PATHS : (1ull << 27)) / n_pixels));
        HIP_TRY(hipEventRecord(ctx->ev0, stream));
        for (uint32_t s0 = 0; s0 < n_sl; s0 += group) {
            const uint32_t here = std::min(group, n_sl - s0);
            w.n_slots = static_cast<uint32_t>(here * n_pixels);
            w.pass_slices = here;
            w.strip_pixels = 0;
            // (the path kernel on scenes with BVH subtrees hands out pixels in 8x8 tiles: rr_wavefront.h, wf_tile_to_pix; the builds
            // for two or more simple subtrees, all samples of a pixel together: render_paths_in)
            w.tile_paths = (!wavefront && ctx->has_subtrees && !(p->flags & RR_FLAG_ROW_ORDER)) ? 1u : 0u;
            if (wavefront && RR_WF_STRIP_ROWS != 0u)  // (the pipeline numbers its paths strip by strip: rr_wavefront.h, wf_path_key)
                w.strip_pixels = static_cast<uint32_t>(std::min<uint64_t>(static_cast<uint64_t>(p->width) * RR_WF_STRIP_ROWS, n_pixels));
            w.claim_max = ctx->auto_pipeline ? RR_WF_CLAIM_LARGE_SCENE : RR_WF_CLAIM_SMALL_SCENE;
            w.slice_first = p->slice_begin + s0;
            w.slice_out = s0;
            int rc;
            if (noalloc)
                rc = render_paths_in(ctx, w, count, static_cast<float*>(d_slice_sums), static_cast<unsigned long long*>(d_counters), stream,
                                     reinterpret_cast<float4*>(ctx->scratch + RR_SCRATCH_HEADER), RR_SCRATCH_PATHS,
                                     reinterpret_cast<unsigned int*>(ctx->scratch));
            else if (paths)
                rc = render_paths(ctx, w, count, static_cast<float*>(d_slice_sums), static_cast<unsigned long long*>(d_counters), stream);
            else
                rc = render_wavefront(ctx, w, count, p->flags, static_cast<float*>(d_slice_sums), static_cast<unsigned long long*>(d_counters), stream);
            if (rc) return rc;
        }
        HIP_TRY(hipEventRecord(ctx->ev1, stream));
        ctx->timed = true;
        return resolve_slices(ctx, p, rows, n_sl, spt, d_slice_sums, d_resolved, d_linear, stream);
    }
    using RenderKernel = void (*)(DevScene, RenderArgs, float*, unsigned long long*);
    const RenderKernel render_kernel =
        ctx->prog_vector_ops ? (count == COUNT_PRODUCTION  ? k_render<COUNT_PRODUCTION, true>
                                : count == COUNT_REFERENCE ? k_render<COUNT_REFERENCE, true> : k_render<COUNT_OFF, true>)
                             : (count ? k_render<COUNT_REFERENCE, false> : k_render<COUNT_OFF, false>);
    const void* kernel = reinterpret_cast<const void*>(render_kernel);
    const size_t lds = static_cast<size_t>(ctx->info.stack_words) * 256u * 4u;
    int grid = 0;
    if (int rc = launch_dims(ctx, lds, kernel, &grid)) return rc;

    RenderArgs a{};
    a.width = p->width;
    a.height = p->height;
    a.row0 = row0;
    a.rows = rows;
    a.spt = spt;
    a.tiles_x = (p->width + 7u) / 8u;
    a.padded_pixels = a.tiles_x * ((rows + 7u) / 8u) * 64u;
    a.max_depth = p->max_depth;
    a.sample_offset = p->sample_offset;
    a.trace_dead = (p->flags & RR_FLAG_TRACE_DEAD_PATHS) ? 1u : 0u;
    a.seed = p->seed;
    // Slices are rendered in groups so that one launch stays within about 2^31 samples.
    const uint64_t samples_per_slice = static_cast<uint64_t>(p->width) * rows * std::max(spt, 1u);
    uint32_t group = static_cast<uint32_t>(std::max<uint64_t>(1, (1ull << 31) / samples_per_slice));
    if (!ctx->next_item) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->next_item), sizeof(unsigned long long)));
    a.next_item = ctx->next_item;
    HIP_TRY(hipEventRecord(ctx->ev0, stream));
    for (uint32_t s0 = 0; s0 < n_sl; s0 += group) {
        const uint32_t cnt = std::min(group, n_sl - s0);
        HIP_TRY(hipMemsetAsync(ctx->next_item, 0, sizeof(unsigned long long), stream));
        a.n_items = static_cast<uint64_t>(cnt) * a.padded_pixels;
        a.slice_first = p->slice_begin + s0;
        a.slice_out = s0;
        const uint64_t blocks_needed = (a.n_items + 255u) / 256u;
        const uint32_t g = static_cast<uint32_t>(std::min<uint64_t>(static_cast<uint64_t>(grid), std::max<uint64_t>(blocks_needed, 1)));
        hipLaunchKernelGGL(render_kernel, dim3(g), dim3(256), lds, stream, ctx->sc, a, static_cast<float*>(d_slice_sums),
                           static_cast<unsigned long long*>(d_counters));
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(ctx->ev1, stream));
    ctx->timed = true;
    return resolve_slices(ctx, p, rows, n_sl, spt, d_slice_sums, d_resolved, d_linear, stream);
}
}  // namespace

extern "C" {

int rr_eval_textures(rr_context* ctx, uint32_t n, const uint32_t* textures, const float* uvp, uint32_t what, float* out_rgb) {
    if (!ctx || (n && (!textures || !uvp || !out_rgb))) return fail(RR_E_INVALID, "rr_eval_textures: NULL argument");
    if (!ctx->has_scene) return fail(RR_E_NO_SCENE, "rr_eval_textures: no scene uploaded");
    if (what > RR_EVAL_NOISE) return fail(RR_E_INVALID, "rr_eval_textures: unknown quantity");
    if (n == 0) return RR_OK;
    for (uint32_t i = 0; i < n; ++i) {
        if (textures[i] >= ctx->tex_types.size()) return fail(RR_E_INVALID, "rr_eval_textures: texture outside the scene's table");
        if (what != RR_EVAL_COLOR && ctx->tex_types[textures[i]] != RR_TEX_NOISE)
            return fail(RR_E_INVALID, "rr_eval_textures: turbulence and noise are asked of a noise texture");
    }
    if (int rc = set_device(ctx)) return rc;
    uint32_t* d_tex = nullptr;
    float *d_in = nullptr, *d_out = nullptr;
    auto cleanup = [&]() {
        if (d_tex) (void)hipFree(d_tex);
        if (d_in) (void)hipFree(d_in);
        if (d_out) (void)hipFree(d_out);
    };
    hipError_t e;
    if ((e = hipMalloc(&d_tex, static_cast<size_t>(n) * 4)) != hipSuccess || (e = hipMalloc(&d_in, static_cast<size_t>(n) * 20)) != hipSuccess ||
        (e = hipMalloc(&d_out, static_cast<size_t>(n) * 12)) != hipSuccess ||
        (e = hipMemcpy(d_tex, textures, static_cast<size_t>(n) * 4, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(d_in, uvp, static_cast<size_t>(n) * 20, hipMemcpyHostToDevice)) != hipSuccess) {
        cleanup();
        return fail(RR_E_HIP, std::string("rr_eval_textures: ") + hipGetErrorString(e));
    }
    hipLaunchKernelGGL(k_eval_textures, dim3((n + 255u) / 256u), dim3(256), 0, 0, ctx->sc, n, d_tex, d_in, what, d_out);
    if ((e = hipGetLastError()) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess ||
        (e = hipMemcpy(out_rgb, d_out, static_cast<size_t>(n) * 12, hipMemcpyDeviceToHost)) != hipSuccess) {
        cleanup();
        return fail(RR_E_HIP, std::string("rr_eval_textures: ") + hipGetErrorString(e));
    }
    cleanup();
    return RR_OK;
}

int rr_render_device(rr_context* ctx, const rr_render_params* p, void* d_slice_sums, void* d_resolved, void* d_linear,
                     void* d_counters, void* hip_stream) {
    return render_device(ctx, p, d_slice_sums, d_resolved, d_linear, d_counters, hip_stream, false);
}

// rr_render_device for callers that cannot allocate device memory: the same kernels (k_wf_paths, k_wf_accumulate, k_resolve)
// with their workspace in the file-scope array `scratch` instead of a hipMalloc'ed one.  Calls no hipMalloc / hipFree.
int rr_render_device_noalloc(rr_context* ctx, const rr_render_params* p, void* d_slice_sums, void* d_resolved, void* d_linear,
                             void* d_counters, void* hip_stream) {
    return render_device(ctx, p, d_slice_sums, d_resolved, d_linear, d_counters, hip_stream, true);
}

int rr_cast_u8_device(rr_context* ctx, const void* d_resolved, void* d_rgb8, uint64_t n, void* hip_stream) {
    if (!ctx || !d_resolved || !d_rgb8) return fail(RR_E_INVALID, "rr_cast_u8_device: NULL argument");
    if (n == 0) return RR_OK;
    if (int rc = set_device(ctx)) return rc;
    const uint32_t g = static_cast<uint32_t>((n + 255u) / 256u);
    hipLaunchKernelGGL(k_cast_u8, dim3(g), dim3(256), 0, static_cast<hipStream_t>(hip_stream), static_cast<const float*>(d_resolved),
                       static_cast<uint8_t*>(d_rgb8), n);
    HIP_TRY(hipGetLastError());
    return RR_OK;
}

int rr_last_kernel_ms(rr_context* ctx, float* ms) {
    if (!ctx || !ms) return fail(RR_E_INVALID, "rr_last_kernel_ms: NULL argument");
    if (!ctx->timed) return fail(RR_E_INVALID, "rr_last_kernel_ms: nothing has been rendered yet");
    if (int rc = set_device(ctx)) return rc;
    HIP_TRY(hipEventSynchronize(ctx->ev1));
    HIP_TRY(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return RR_OK;
}

int rr_render(rr_context* ctx, const rr_render_params* p, uint8_t* out_rgb8, float* out_resolved, float* out_linear,
              float* out_slice_sums, rr_counters* out_counters) {
    if (!ctx || !p) return fail(RR_E_INVALID, "rr_render: NULL argument");
    if (!ctx->has_scene) return fail(RR_E_NO_SCENE, "rr_render: no scene uploaded");
    if (p->slice_begin > p->slice_end || p->slice_end > p->n_slices || p->n_slices == 0 || p->width == 0 || p->height == 0)
        return fail(RR_E_INVALID, "rr_render: bad image size or slice range");
    uint32_t row0 = 0, rows = 0;
    if (!band_of(p, &row0, &rows)) return fail(RR_E_INVALID, "rr_render: bad row range");
    if (int rc = set_device(ctx)) return rc;
    const uint64_t n_out = static_cast<uint64_t>(p->width) * rows * 3u;  // elements of one output image (a band of rows)
    const uint64_t n_elems = std::max<uint64_t>(1, n_out);                // (an empty band still gets buffers)
    const uint32_t n_sl = p->slice_end - p->slice_begin;
    float *d_sums = nullptr, *d_res = nullptr, *d_lin = nullptr;
    uint8_t* d_u8 = nullptr;
    unsigned long long* d_cnt = nullptr;
    auto cleanup = [&]() {
        if (d_sums) (void)hipFree(d_sums);
        if (d_res) (void)hipFree(d_res);
        if (d_lin) (void)hipFree(d_lin);
        if (d_u8) (void)hipFree(d_u8);
        if (d_cnt) (void)hipFree(d_cnt);
    };
    auto hip_fail = [&](hipError_t e) {
        cleanup();
        return fail(RR_E_HIP, std::string("rr_render: ") + hipGetErrorString(e));
    };
    hipError_t e;
    if ((e = hipMalloc(&d_sums, std::max<uint64_t>(1, n_sl) * n_elems * 4)) != hipSuccess) return hip_fail(e);
    if ((e = hipMalloc(&d_res, n_elems * 4)) != hipSuccess) return hip_fail(e);
    if ((e = hipMalloc(&d_lin, n_elems * 4)) != hipSuccess) return hip_fail(e);
    if ((e = hipMalloc(&d_u8, n_elems)) != hipSuccess) return hip_fail(e);
    if ((e = hipMalloc(&d_cnt, sizeof(rr_counters))) != hipSuccess) return hip_fail(e);
    if ((e = hipMemset(d_cnt, 0, sizeof(rr_counters))) != hipSuccess) return hip_fail(e);
    int rc = rr_render_device(ctx, p, d_sums, d_res, d_lin, d_cnt, nullptr);
    if (rc == RR_OK) rc = rr_cast_u8_device(ctx, d_res, d_u8, n_out, nullptr);
    if (rc != RR_OK) {
        cleanup();
        return rc;
    }
    if ((e = hipDeviceSynchronize()) != hipSuccess) return hip_fail(e);
    if (out_rgb8 && n_out && (e = hipMemcpy(out_rgb8, d_u8, n_out, hipMemcpyDeviceToHost)) != hipSuccess) return hip_fail(e);
    if (out_resolved && n_out && (e = hipMemcpy(out_resolved, d_res, n_out * 4, hipMemcpyDeviceToHost)) != hipSuccess) return hip_fail(e);
    if (out_linear && n_out && (e = hipMemcpy(out_linear, d_lin, n_out * 4, hipMemcpyDeviceToHost)) != hipSuccess) return hip_fail(e);
    if (out_slice_sums && n_sl && n_out && (e = hipMemcpy(out_slice_sums, d_sums, static_cast<uint64_t>(n_sl) * n_out * 4, hipMemcpyDeviceToHost)) != hipSuccess)
        return hip_fail(e);
    if (out_counters && (e = hipMemcpy(out_counters, d_cnt, sizeof(rr_counters), hipMemcpyDeviceToHost)) != hipSuccess) return hip_fail(e);
    cleanup();
    return RR_OK;
}


// ------------------------------------------------------------------ several GPUs from one host thread
// RCCL, bound at run time: a host with one GPU (or without librccl.so) still loads this library and uses the peer-copy sum.
struct Rccl {
    void* handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string why;  // why it is not usable, when it is not
    bool ok() const { return handle != nullptr; }
};
static Rccl& rccl() {
    static Rccl r = [] {
        Rccl x;
        void* h = nullptr;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
        if (!h) {
            const char* e = dlerror();
            x.why = std::string("librccl.so not loadable: ") + (e ? e : "?");
            return x;
        }
        x.CommInitAll = reinterpret_cast<decltype(x.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
        x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        x.Reduce = reinterpret_cast<decltype(x.Reduce)>(dlsym(h, "ncclReduce"));
        x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(dlsym(h, "ncclGroupStart"));
        x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
        x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        if (!x.CommInitAll || !x.CommDestroy || !x.Reduce || !x.GroupStart || !x.GroupEnd || !x.GetErrorString) {
            x.why = "librccl.so lacks one of ncclCommInitAll / ncclCommDestroy / ncclReduce / ncclGroupStart / ncclGroupEnd / ncclGetErrorString";
            dlclose(h);
            return x;
        }
        x.handle = h;
        return x;
    }();
    return r;
}

struct rr_multi {
    struct Dev {
        int device = 0;
        rr_context* ctx = nullptr;
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
        // resolved and linear are the two halves of ONE allocation (linear = resolved + image), so that a band's slice
        // groups are summed with one collective over both
        float *sums = nullptr, *resolved = nullptr, *linear = nullptr, *stage = nullptr;
        uint8_t* u8 = nullptr;
        unsigned long long* counters = nullptr;
        size_t sums_bytes = 0, image_bytes = 0;
    };
    std::vector<Dev> devs;
    uint32_t reduce_mode = RR_REDUCE_AUTO;
    uint32_t reduce_used = RR_REDUCE_PEER;  // what the last rr_multi_render summed its slice groups with
    std::map<uint32_t, std::vector<ncclComm_t>> comms;  // by number of slice groups G: one communicator per device, band b's are [b*G, (b+1)*G)
};

namespace {
void cut(uint32_t n, uint32_t parts, uint32_t i, uint32_t* a, uint32_t* b) {
    *a = static_cast<uint32_t>(static_cast<uint64_t>(i) * n / parts);
    *b = static_cast<uint32_t>(static_cast<uint64_t>(i + 1) * n / parts);
}
int grow(void** p, size_t* have, size_t need) {
    if (need <= *have && *p) return RR_OK;
    if (*p) HIP_TRY(hipFree(*p));
    *p = nullptr;
    *have = 0;
    HIP_TRY(hipMalloc(p, std::max<size_t>(need, 256)));
    *have = need;
    return RR_OK;
}
}  // namespace

namespace {
// Which way the slice groups of every band are summed in this call, and the communicators for it.
// AUTO: RCCL when there is something to sum (G > 1), every band's devices are distinct GPUs (a communicator cannot hold one
// device twice) and librccl.so loads and initialises; the peer-copy sum otherwise.  RCCL (forced): every band goes through
// ncclReduce, also a band of one device, where the reduce is a copy in place: an error if RCCL cannot be used.
int pick_reduce(rr_multi* m, uint32_t bands, uint32_t groups, bool* use_rccl) {
    *use_rccl = false;
    if (m->reduce_mode == RR_REDUCE_PEER) return RR_OK;
    const bool forced = m->reduce_mode == RR_REDUCE_RCCL;
    if (!forced && groups < 2) return RR_OK;
    for (uint32_t b = 0; b < bands; ++b) {
        std::set<int> seen;
        for (uint32_t g = 0; g < groups; ++g)
            if (!seen.insert(m->devs[b * groups + g].device).second)
                return forced ? fail(RR_E_INVALID, "rr_multi_render: RR_REDUCE_RCCL needs distinct devices within a band (a communicator cannot hold a device twice)") : RR_OK;
    }
    Rccl& R = rccl();
    if (!R.ok()) return forced ? fail(RR_E_NO_DEVICE, "rr_multi_render: " + R.why) : RR_OK;
    if (!m->comms.count(groups)) {
        std::vector<ncclComm_t> comms(m->devs.size(), nullptr);
        for (uint32_t b = 0; b < bands; ++b) {
            std::vector<int> list;
            for (uint32_t g = 0; g < groups; ++g) list.push_back(m->devs[b * groups + g].device);
            const ncclResult_t nr = R.CommInitAll(&comms[b * groups], static_cast<int>(groups), list.data());
            if (nr != ncclSuccess) {
                for (ncclComm_t c : comms)
                    if (c) (void)R.CommDestroy(c);
                if (forced) return fail(RR_E_HIP, std::string("rr_multi_render: ncclCommInitAll failed: ") + R.GetErrorString(nr));
                m->reduce_mode = RR_REDUCE_PEER;  // do not try again at every call
                return RR_OK;
            }
        }
        m->comms[groups] = comms;
    }
    *use_rccl = true;
    return RR_OK;
}
}  // namespace

int rr_multi_set_reduce(rr_multi* m, uint32_t mode) {
    if (!m || mode > RR_REDUCE_RCCL) return fail(RR_E_INVALID, "rr_multi_set_reduce: bad argument");
    if (mode == RR_REDUCE_RCCL && !rccl().ok()) return fail(RR_E_NO_DEVICE, "rr_multi_set_reduce: " + rccl().why);
    m->reduce_mode = mode;
    return RR_OK;
}

int rr_multi_reduce_used(rr_multi* m, uint32_t* mode) {
    if (!m || !mode) return fail(RR_E_INVALID, "rr_multi_reduce_used: NULL argument");
    *mode = m->reduce_used;
    return RR_OK;
}

int rr_multi_create(int n_devices, const int* devices, rr_multi** out) {
    if (!out || n_devices < 1 || !devices) return fail(RR_E_INVALID, "rr_multi_create: bad argument");
    *out = nullptr;
    rr_multi* m = new rr_multi();
    m->devs.resize(static_cast<size_t>(n_devices));
    for (int i = 0; i < n_devices; ++i) {
        rr_multi::Dev& d = m->devs[static_cast<size_t>(i)];
        d.device = devices[i];
        int rc = rr_context_create(devices[i], &d.ctx);
        if (rc == RR_OK && (hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking) != hipSuccess ||
                            hipEventCreateWithFlags(&d.done, hipEventDisableTiming) != hipSuccess))
            rc = fail(RR_E_HIP, "rr_multi_create: stream or event creation failed");
        if (rc == RR_OK && hipMalloc(reinterpret_cast<void**>(&d.counters), sizeof(rr_counters)) != hipSuccess)
            rc = fail(RR_E_HIP, "rr_multi_create: hipMalloc failed");
        if (rc != RR_OK) {
            const std::string msg = g_err;
            rr_multi_destroy(m);
            return fail(rc, msg);
        }
    }
    // bands are summed on their first device: let it read its partners' memory directly where the hardware can (xGMI)
    for (const rr_multi::Dev& a : m->devs)
        for (const rr_multi::Dev& b : m->devs)
            if (a.device != b.device) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, a.device, b.device) == hipSuccess && can) {
                    (void)hipSetDevice(a.device);
                    (void)hipDeviceEnablePeerAccess(b.device, 0);  // (already enabled is not an error worth reporting)
                    (void)hipGetLastError();
                }
            }
    *out = m;
    return RR_OK;
}

int rr_multi_destroy(rr_multi* m) {
    if (!m) return RR_OK;
    for (rr_multi::Dev& d : m->devs)
        if (d.ctx && d.stream) {
            (void)hipSetDevice(d.device);
            (void)hipStreamSynchronize(d.stream);
        }
    for (auto& kv : m->comms)
        for (ncclComm_t c : kv.second)
            if (c) (void)rccl().CommDestroy(c);
    m->comms.clear();
    for (rr_multi::Dev& d : m->devs) {
        if (!d.ctx) continue;  // creation stopped before this entry (its device may not even exist)
        (void)hipSetDevice(d.device);
        if (d.stream) (void)hipStreamSynchronize(d.stream);
        for (void* p : {static_cast<void*>(d.sums), static_cast<void*>(d.resolved), static_cast<void*>(d.stage),
                        static_cast<void*>(d.u8), static_cast<void*>(d.counters)})  // (d.linear is the second half of d.resolved)
            if (p) (void)hipFree(p);
        if (d.done) (void)hipEventDestroy(d.done);
        if (d.stream) (void)hipStreamDestroy(d.stream);
        if (d.ctx) rr_context_destroy(d.ctx);
    }
    delete m;
    return RR_OK;
}

int rr_multi_scene_upload(rr_multi* m, const rr_flat_scene* scene) {
    if (!m || !scene) return fail(RR_E_INVALID, "rr_multi_scene_upload: NULL argument");
    for (rr_multi::Dev& d : m->devs)
        if (int rc = rr_scene_upload(d.ctx, scene)) return rc;
    return RR_OK;
}

int rr_multi_scene_specialize(rr_multi* m, uint32_t mode) {
    if (!m) return fail(RR_E_INVALID, "rr_multi_scene_specialize: NULL argument");
    for (rr_multi::Dev& d : m->devs)  // (the first device's call compiles; the others find the code object in the process's table)
        if (int rc = rr_scene_specialize(d.ctx, mode)) return rc;
    return RR_OK;
}

int rr_multi_render(rr_multi* m, const rr_render_params* job, uint32_t split, uint8_t* out_rgb8, float* out_resolved, float* out_linear,
                    rr_counters* out_counters) {
    if (!m || !job) return fail(RR_E_INVALID, "rr_multi_render: NULL argument");
    if (split > RR_SPLIT_BOTH) return fail(RR_E_INVALID, "rr_multi_render: unknown split");
    if (job->width == 0 || job->height == 0 || job->n_slices == 0) return fail(RR_E_INVALID, "rr_multi_render: bad image size or slice count");
    if (job->slice_begin != 0 || job->slice_end != job->n_slices || job->row_begin != 0 || (job->row_end != 0 && job->row_end != job->height))
        return fail(RR_E_INVALID, "rr_multi_render: the job is a whole image (all slices, all rows); the shares are cut here");
    const uint32_t world = static_cast<uint32_t>(m->devs.size());
    // B bands of rows x G slice groups, B * G = world (the cut rust_raytracer_amd/multi.py makes for one process per GPU)
    uint32_t bands = 1, groups = world;
    if (split == RR_SPLIT_ROWS) bands = world, groups = 1;
    if (split == RR_SPLIT_BOTH) {
        for (bands = 1; bands * bands < world || world % bands != 0; ++bands) {}
        groups = world / bands;
    }
    if (bands > job->height) return fail(RR_E_INVALID, "rr_multi_render: more bands than image rows");
    const uint32_t W = job->width;
    // ---- every device renders its share on its own stream; nothing below waits for a device until all are enqueued
    std::vector<rr_render_params> share(world, *job);
    for (uint32_t r = 0; r < world; ++r) {
        rr_multi::Dev& d = m->devs[r];
        rr_render_params& p = share[r];
        cut(job->n_slices, groups, r % groups, &p.slice_begin, &p.slice_end);
        cut(job->height, bands, r / groups, &p.row_begin, &p.row_end);
        const uint32_t rows = p.row_end - p.row_begin, n_sl = p.slice_end - p.slice_begin;
        const size_t image = static_cast<size_t>(W) * rows * 3u * sizeof(float);
        HIP_TRY(hipSetDevice(d.device));
        size_t keep = d.image_bytes;
        if (int rc = grow(reinterpret_cast<void**>(&d.sums), &d.sums_bytes, image * std::max(n_sl, 1u))) return rc;
        if (image > d.image_bytes || !d.resolved) {
            size_t have = keep * 2u;
            if (int rc = grow(reinterpret_cast<void**>(&d.resolved), &have, image * 2u)) return rc;
            have = keep * 2u;
            if (int rc = grow(reinterpret_cast<void**>(&d.stage), &have, image * 2u)) return rc;
            size_t have8 = keep / 4u;
            if (int rc = grow(reinterpret_cast<void**>(&d.u8), &have8, image / 4u)) return rc;
            d.image_bytes = image;
        }
        d.linear = d.resolved + image / sizeof(float);  // the band's two images back to back, whatever the band's size
        HIP_TRY(hipMemsetAsync(d.counters, 0, sizeof(rr_counters), d.stream));
        if (int rc = rr_render_device(d.ctx, &p, d.sums, d.resolved, d.linear, d.counters, d.stream)) return rc;
        HIP_TRY(hipEventRecord(d.done, d.stream));
    }
    // ---- per band: the slice groups' images are added onto the band's first device, in group order (the mutex sum of
    // render.rs:141-150), then cast (render.rs:156-159) and copied to their rows of the host image
    bool use_rccl = false;
    if (int rc = pick_reduce(m, bands, groups, &use_rccl)) return rc;
    m->reduce_used = use_rccl ? RR_REDUCE_RCCL : RR_REDUCE_PEER;
    for (uint32_t b = 0; b < bands; ++b) {
        rr_multi::Dev& root = m->devs[b * groups];
        const rr_render_params& p = share[b * groups];
        const uint64_t n = static_cast<uint64_t>(W) * (p.row_end - p.row_begin) * 3u;
        if (use_rccl) {
            // one ncclReduce(sum, f32) over the band's devices, each on the stream its render was enqueued on, root = the band's
            // first device, in place there: `resolved` and `linear` travel together (they are one allocation)
            const std::vector<ncclComm_t>& comms = m->comms[groups];
            Rccl& R = rccl();
            ncclResult_t nr = R.GroupStart();
            for (uint32_t g = 0; g < groups && nr == ncclSuccess; ++g) {
                rr_multi::Dev& d = m->devs[b * groups + g];
                HIP_TRY(hipSetDevice(d.device));
                nr = R.Reduce(d.resolved, d.resolved, n * 2u, ncclFloat, ncclSum, 0, comms[b * groups + g], d.stream);
            }
            const ncclResult_t ne = R.GroupEnd();
            if (nr == ncclSuccess) nr = ne;
            if (nr != ncclSuccess) return fail(RR_E_HIP, std::string("rr_multi_render: ncclReduce failed: ") + R.GetErrorString(nr));
            HIP_TRY(hipSetDevice(root.device));
        } else {
            HIP_TRY(hipSetDevice(root.device));
            for (uint32_t g = 1; g < groups; ++g) {
                rr_multi::Dev& src = m->devs[b * groups + g];
                HIP_TRY(hipStreamWaitEvent(root.stream, src.done, 0));
                const uint64_t count = out_linear ? n * 2u : n;  // (both images are adjacent on either side)
                HIP_TRY(hipMemcpyPeerAsync(root.stage, root.device, src.resolved, src.device, count * sizeof(float), root.stream));
                hipLaunchKernelGGL(k_add_images, dim3(static_cast<uint32_t>((count + 255u) / 256u)), dim3(256), 0, root.stream, root.resolved, root.stage, count);
                HIP_TRY(hipGetLastError());
            }
        }
        const size_t row0 = static_cast<size_t>(p.row_begin) * W * 3u;
        if (out_rgb8) {
            if (int rc = rr_cast_u8_device(root.ctx, root.resolved, root.u8, n, root.stream)) return rc;
            HIP_TRY(hipMemcpyAsync(out_rgb8 + row0, root.u8, n, hipMemcpyDeviceToHost, root.stream));
        }
        if (out_resolved) HIP_TRY(hipMemcpyAsync(out_resolved + row0, root.resolved, n * sizeof(float), hipMemcpyDeviceToHost, root.stream));
        if (out_linear) HIP_TRY(hipMemcpyAsync(out_linear + row0, root.linear, n * sizeof(float), hipMemcpyDeviceToHost, root.stream));
    }
    rr_counters total{};
    for (rr_multi::Dev& d : m->devs) {
        HIP_TRY(hipSetDevice(d.device));
        HIP_TRY(hipStreamSynchronize(d.stream));
        rr_counters c{};
        HIP_TRY(hipMemcpy(&c, d.counters, sizeof(c), hipMemcpyDeviceToHost));
        uint64_t* t = reinterpret_cast<uint64_t*>(&total);
        const uint64_t* q = reinterpret_cast<const uint64_t*>(&c);
        for (size_t k = 0; k < sizeof(rr_counters) / sizeof(uint64_t); ++k) t[k] += q[k];
    }
    if (out_counters) *out_counters = total;
    return RR_OK;
}

}  // extern "C"

#ifdef RR_SECTION_TIMERS  // timing-only build: reads (and clears) the wave clocks the path kernel has added up by section
extern "C" int rr_debug_section_cycles(unsigned long long* out16) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(rrdev::g_sec), 16 * sizeof(unsigned long long)) != hipSuccess) return 2;
    unsigned long long zero[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(rrdev::g_sec), zero, sizeof(zero)) != hipSuccess) return 3;
    return 0;
}
#endif

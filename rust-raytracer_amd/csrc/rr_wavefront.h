// Wavefront form of the path tracer for scenes with BVH subtrees.  Included only by rr_hip.hip,
// after the definitions of flush_counts and the device code of rr_device.h.
//
// The persistent kernel k_render keeps a lane on its path from camera ray to termination, which
// is ideal while all 64 rays of a wave take the same steps through the scene.  Below a BVH they
// do not: traversal lengths differ by an order of magnitude between neighbouring rays, and a wave
// is as slow as its longest ray.  Here a path's state lives in device memory between bounces, the
// first bounce or two of a batch are two kernels each over a compacted queue of live paths, and
// one more kernel runs what is left of every path:
//
//   k_wf_traverse  persistent waves; a lane takes a ray from the queue, runs the traversal machine
//                  over the whole scene tree, stores (t, primitive) and takes the next ray, so a
//                  wave never waits for its longest ray with idle lanes; the kernel holds nothing
//                  but traversal state and runs at twice the occupancy of k_render
//   k_wf_shade     one thread per live path: ray_color's work after scene.intersect
//                  (render.rs:19-40); a path that goes on is appended to the next bounce's queue,
//                  one that ends leaves its sample value (throughput x terminal colour) behind
//   k_wf_finish    a lane per remaining path, bounce after bounce to the path's end; lanes whose
//                  path has ended take the next one from the queue
//
// Paths are independent, so many samples of a pixel are in flight at once: a pass covers a batch
// of consecutive samples of every (slice, pixel) pair (a "slot"), runs until every path has
// ended, and k_wf_accumulate then adds each slot's sample values to its pixel sum in sample
// order (render.rs:93-100).  Every random stream is keyed by (pixel, sample, bounce) and every
// f32 sum is taken in the reference's order, so this form and k_render produce the same bits.
#pragma once

// Waves per SIMD k_wf_traverse is compiled for: 7, a budget of 72 vector registers.  At 8 (64 registers) the build with the
// wide node move keeps 64 scalar values in vector lanes and moves them at 511 places; at 7 it is 32 values and 200 places
// and a sixth fewer vector instructions in the binary (1.1 M-primitive mesh +7.5 %, whose 23-word stacks leave room for 6
// workgroups per CU either way; the pipeline on Suzanne / final_scene / random spheres, which lose a wave, +3.7 / +3.3 /
// +9.7 %).  6 and 5: the same code as 7.
#ifndef RR_WF_TRAVERSE_WAVES_PER_SIMD
#define RR_WF_TRAVERSE_WAVES_PER_SIMD 7
#endif
#ifndef RR_WF_SHADE_WAVES_PER_SIMD
#define RR_WF_SHADE_WAVES_PER_SIMD 3
#endif
constexpr uint32_t WF_BLOCK = 256;   // threads of a workgroup of every kernel here that keeps per-lane stacks in LDS (the hosts launch them so)
constexpr uint32_t WF_STAGE = 1024;  // queue entries a shading wave collects before it appends them
#ifndef RR_WF_LEAF_LANES
#define RR_WF_LEAF_LANES 24  // lanes waiting at a leaf that make the wave take the leaf move at once
#endif
#ifndef RR_WF_REFILL_IDLE
// A wave fetches new rays once this many of its lanes have none: 64, a whole cohort at a time.  The 64 entries of a claim
// are the samples of a 2x2 cell of pixels (wf_strip_pos), rays that meet the same nodes and leaves in the same steps, so
// that most lanes take each move of the machine together; rays of another cell fed to the lanes that fall idle are at
// other kinds of records in every step and the wave executes both moves for a part of its lanes each (1.1 M-primitive
// mesh, Msamples/s: 16 idle lanes 80.6, 24: 75.8, 32: 74.0, 40-56: 65.6-68.3, 64: 94.8).
#define RR_WF_REFILL_IDLE 64
#endif

struct WfState {     // one entry per path in each array
    float4* ray_o;   // (origin.xyz, time)
    float4* ray_d;   // (direction.xyz, ray_color depth left)
    float4* thr;     // (throughput.xyz, draws the traversal took from the bounce's random stream) while the path lives,
                     // its sample value once it has ended
    uint4* hit;      // (t, primitive ref, material, owner word) of the closest hit, ref NONE for a miss
};
// Queue entries a traversal wave claims at a time, at most (WfArgs::claim_max; at least 64).  A scene that the dies' L2s do
// not hold: 64, one wave's worth — the waves of a die then walk neighbouring cells of the image at the same time, and what
// they fetch of the scene is one small region (1.1 M-primitive mesh: 256 -> 64 entries +10 %, 4096 -> 64 +22 % with rows
// for cells).  A cache-resident scene, whose rays are short: 512, or the claims' atomics take as long as the rays (64
// entries halve Suzanne and final_scene).
#ifndef RR_WF_CLAIM_LARGE_SCENE
#define RR_WF_CLAIM_LARGE_SCENE 64u
#endif
#ifndef RR_WF_CLAIM_SMALL_SCENE
#define RR_WF_CLAIM_SMALL_SCENE 512u
#endif
#ifndef RR_WF_STRIP_ROWS
#define RR_WF_STRIP_ROWS 8u  // image rows per strip of the wavefront pipeline's path numbering (0: sample-major numbering)
#endif
constexpr uint32_t WF_XCDS = 8;  // accelerator dies of an MI355X, each with an L2 of its own
struct WfCtl {
    uint32_t count;   // entries in the queue
    uint32_t cursor;  // next entry no traversal wave has claimed yet
    uint32_t xcur[WF_XCDS];  // k_wf_traverse: per eighth of the queue, how far into it entries have been claimed
};
struct WfArgs {
    uint32_t width, height;  // of the whole image
    uint32_t n_pixels;       // pixels of the band of rows this call renders (the whole image unless rr_render_params asks for a band)
    uint32_t pixel0;         // position of the band's first pixel in the whole image (row_begin * width)
    uint32_t n_slots;      // slices in this pass * n_pixels
    uint32_t n_paths;      // n_slots * samples in this batch; path p is sample s0 + p / n_slots of slot p % n_slots
    uint32_t s0;           // first sample of the batch within its slice
    uint32_t spt;          // samples per slice and pixel (render.rs:130)
    uint32_t slice_first;  // absolute index of the pass's first slice
    uint32_t slice_out;    // index of that slice inside the output buffer
    uint32_t max_depth;
    uint32_t sample_offset;
    uint32_t stack_words;  // words of traversal stack per lane: the wave's slots for parked lanes follow the stacks in LDS
    uint32_t paths_chunk;  // k_wf_paths: paths a wave claims at a time (0: RR_PATHS_CHUNK)
    uint32_t strip_pixels; // wavefront pipeline: paths are numbered strip by strip (see wf_path_key); 0: sample-major over the whole band
    uint32_t pass_slices;  // n_slots / n_pixels: slices in this pass
    uint32_t n_sub;        // (samples in this batch) * pass_slices: the paths of one pixel in this batch
    uint32_t trace_dead;   // RR_FLAG_TRACE_DEAD_PATHS: paths go on after their throughput has become zero (see dead_path)
    uint32_t claim_max;    // k_wf_traverse: queue entries a wave claims at a time, at most (RR_WF_CLAIM_*)
    uint32_t tile_paths;   // k_wf_paths: 1: the pixels of the band are numbered in 8x8 tiles (wf_tile_to_pix) instead of row by row; 2: and the
                           // paths cell by cell (wf_path_key)
    uint64_t seed;
};

// How the paths of a batch are numbered.  The persistent kernel k_wf_paths numbers them sample-major: path p is sample
// s0 + p / n_slots of slot p % n_slots, slot = slice * n_pixels + pixel.  The wavefront pipeline (scenes larger than the
// L2s) numbers them strip by strip (a.strip_pixels != 0): a strip is eight image rows, and all of a strip's paths — every
// slice and batch sample of its pixels — come before the next strip's, cell by cell (wf_strip_pos).  Rays that are neighbours in the queue
// are then neighbours in the image whatever their sample number, and a contiguous range of the queue is a band of the image:
// that is what lets each die keep its part of the scene in its own L2 (k_wf_traverse).  The numbering changes no result:
// every random stream is keyed by (pixel, sample), and k_wf_accumulate adds a pixel's samples in sample order either way.
// STRIPS = false: a build that knows the numbering is sample-major.
// The cells of the path kernel's cell-by-cell numbering (WfArgs::tile_paths == 2, wf_path_key): 2^WF_CELL_LOG2 pixels that follow
// each other in the order of wf_tile_to_pix.  One pixel: a wave's 64 camera rays are 64 samples of it (or all samples of a
// few neighbours).  Measured on final_scene at 2 000 samples per pixel, against cells of 4 pixels (2x2): 8 pixels -1.0 %,
// 2 pixels +0.7 %, 1 pixel +1.5 %.
#ifndef RR_PATHS_CELL_LOG2
#define RR_PATHS_CELL_LOG2 0
#endif
constexpr uint32_t WF_CELL_LOG2 = RR_PATHS_CELL_LOG2;
struct WfPathPos {
    uint32_t pix;  // pixel within the band of rows
    uint32_t sub;  // s_local * pass_slices + slice_local
};
#ifndef RR_WF_CELL_W
#define RR_WF_CELL_W 2u
#endif
#ifndef RR_WF_CELL_H
#define RR_WF_CELL_H 2u
#endif
// Inside a full strip (RR_WF_STRIP_ROWS rows of the image) the paths are numbered cell by cell: all of the batch's paths of
// a cell of RR_WF_CELL_W x RR_WF_CELL_H pixels — every slice and sample, pixel fastest — before the next cell's, the cells
// column by column; then the columns that cells do not cover, sample-major.  64 consecutive paths, what a traversal wave
// takes at a time, are then 16 samples of one cell (or all samples of several cells), camera rays that differ by a pixel.
// Measured on the 1.1 M-primitive mesh at 8 samples per pixel, Msamples/s: rows 73.3; 8x8 cells one sample each 90.1;
// cell-major 8x8 90.6, 8x4 92.0, 4x4 94.6, 4x2 95.4, 2x2 95.3.  A strip of fewer rows (the band's last) keeps the row order.
__device__ __forceinline__ WfPathPos wf_strip_pos(const WfArgs& a, uint32_t idx) {  // position within the strip's paths -> (pixel within the strip, sub)
    constexpr uint32_t CW = RR_WF_CELL_W, CH = RR_WF_CELL_H, CP = CW * CH, ROWS = RR_WF_STRIP_ROWS, CY = ROWS / CH;
    const uint32_t wc = a.width - a.width % CW;
    const uint32_t cell_part = ROWS * wc * a.n_sub;
    if (idx < cell_part) {
        const uint32_t per_cell = CP * a.n_sub, c = idx / per_cell, rem = idx - c * per_cell, sub = rem / CP, in = rem % CP;
        const uint32_t cx = c / CY, cy = c % CY;
        return WfPathPos{(cy * CH + in / CW) * a.width + cx * CW + in % CW, sub};
    }
    const uint32_t q = idx - cell_part, right = a.width - wc, per = ROWS * right, sub = q / per, r = q - sub * per, y = r / right;
    return WfPathPos{y * a.width + wc + (r - y * right), sub};
}
__device__ __forceinline__ uint32_t wf_strip_index(const WfArgs& a, uint32_t r, uint32_t sub) {  // the inverse: pixel r of the strip, sub -> position
    constexpr uint32_t CW = RR_WF_CELL_W, CH = RR_WF_CELL_H, CP = CW * CH, ROWS = RR_WF_STRIP_ROWS, CY = ROWS / CH;
    const uint32_t wc = a.width - a.width % CW;
    const uint32_t y = r / a.width, x = r - y * a.width;
    if (x < wc) return ((x / CW) * CY + y / CH) * (CP * a.n_sub) + sub * CP + (y % CH) * CW + x % CW;
    const uint32_t right = a.width - wc;
    return ROWS * wc * a.n_sub + sub * (ROWS * right) + y * right + (x - wc);
}
template <bool STRIPS = true>
__device__ __forceinline__ WfPathPos wf_path_pos(const WfArgs& a, uint32_t p) {
    if (!STRIPS || a.strip_pixels == 0u) {
        const uint32_t slot = p % a.n_slots, s = p / a.n_slots;
        return WfPathPos{slot % a.n_pixels, s * a.pass_slices + slot / a.n_pixels};
    }
    const uint32_t per_strip = a.strip_pixels * a.n_sub;
    const uint32_t strip = p / per_strip, rem = p - strip * per_strip;
    const uint32_t first = strip * a.strip_pixels;
    const uint32_t len = min(a.strip_pixels, a.n_pixels - first);  // (the last strip may be shorter)
    if (len == RR_WF_STRIP_ROWS * a.width) {  // a full strip: its paths go cell by cell
        const WfPathPos q = wf_strip_pos(a, rem);
        return WfPathPos{first + q.pix, q.sub};
    }
    const uint32_t sub = rem / len;
    return WfPathPos{first + (rem - sub * len), sub};
}
__device__ __forceinline__ uint32_t wf_pix_to_tile(const WfArgs& a, uint32_t pix);
__device__ __forceinline__ uint32_t wf_path_index(const WfArgs& a, uint32_t pix, uint32_t sub) {
    if (a.strip_pixels == 0u && a.tile_paths == 2u) {  // cell by cell: see wf_path_key
        const uint32_t tiled = (a.width & ~7u) * ((a.n_pixels / a.width) & ~7u), q = wf_pix_to_tile(a, pix);
        if (q < tiled) return (q >> WF_CELL_LOG2) * ((1u << WF_CELL_LOG2) * a.n_sub) + (sub << WF_CELL_LOG2) + (q & ((1u << WF_CELL_LOG2) - 1u));
        return tiled * a.n_sub + sub * (a.n_pixels - tiled) + (q - tiled);
    }
    if (a.strip_pixels == 0u)
        return (sub / a.pass_slices) * a.n_slots + (sub % a.pass_slices) * a.n_pixels + (a.tile_paths ? wf_pix_to_tile(a, pix) : pix);
    const uint32_t strip = pix / a.strip_pixels, first = strip * a.strip_pixels;
    const uint32_t len = min(a.strip_pixels, a.n_pixels - first);
    if (len == RR_WF_STRIP_ROWS * a.width) return first * a.n_sub + wf_strip_index(a, pix - first, sub);
    return first * a.n_sub + sub * len + (pix - first);
}
// (pixel, per-pixel sample number) of a path: the key of its random streams.  The pixel is its position in the whole
// image (row * width + column), whichever band of rows is being rendered.
// Tile order of the pixels of a band (WfArgs::tile_paths, the path kernel on scenes with BVH subtrees): 64 consecutive
// positions are an 8x8 tile of pixels, so that the 64 lanes of a wave start on camera rays that walk the same part of a
// BVH (64 pixels of one row spread over eight times the width).  The part of the band that 8x8 tiles cover exactly comes
// first, tile by tile; then the columns to the right of it, row by row; then the rows below it.  Only the order in which
// paths are handed out changes: every random stream is keyed by (pixel, sample), and a pixel's samples are added in sample
// order wherever they lie (k_wf_accumulate).  Measured once four rays in five were camera rays (dead paths ended): final_scene
// 6 060 -> 6 525 Msamples/s; Suzanne the same; the Cornell box, which has no BVH, -1.2 %, and keeps the rows.
__device__ __forceinline__ uint32_t wf_tile_to_pix(const WfArgs& a, uint32_t q) {  // position in tile order -> pixel of the band
    const uint32_t rows = a.n_pixels / a.width, w8 = a.width & ~7u, r8 = rows & ~7u, tiled = w8 * r8;
    if (q < tiled) {
        const uint32_t tile = q >> 6, in = q & 63u, tiles_x = w8 >> 3;
        const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
        if (a.tile_paths == 2u)  // the tile's sixteen 2x2 cells one after the other
            return (ty * 8u + ((in >> 4) << 1) + ((in >> 1) & 1u)) * a.width + tx * 8u + (((in >> 2) & 3u) << 1) + (in & 1u);
        return (ty * 8u + (in >> 3)) * a.width + tx * 8u + (in & 7u);
    }
    uint32_t r = q - tiled;
    const uint32_t right = a.width - w8;  // columns to the right of the tiles
    if (r < right * r8) {
        const uint32_t y = r / right;
        return y * a.width + w8 + (r - y * right);
    }
    r -= right * r8;
    return r8 * a.width + r;  // the rows below the tiles, whole
}
__device__ __forceinline__ uint32_t wf_pix_to_tile(const WfArgs& a, uint32_t pix) {
    const uint32_t rows = a.n_pixels / a.width, w8 = a.width & ~7u, r8 = rows & ~7u;
    const uint32_t y = pix / a.width, x = pix - y * a.width;
    if (x < w8 && y < r8 && a.tile_paths == 2u)
        return (((y >> 3) * (w8 >> 3) + (x >> 3)) << 6) | (((y & 7u) >> 1) << 4) | (((x & 7u) >> 1) << 2) | ((y & 1u) << 1) | (x & 1u);
    if (x < w8 && y < r8) return (((y >> 3) * (w8 >> 3) + (x >> 3)) << 6) | ((y & 7u) << 3) | (x & 7u);
    if (y < r8) return w8 * r8 + y * (a.width - w8) + (x - w8);
    return r8 * a.width + (pix - r8 * a.width);  // (= pix: everything above is counted, in whatever order)
}
// CELLS = false: a build that knows the numbering is not cell by cell (tile_paths != 2).
template <bool STRIPS = true, bool CELLS = true>
__device__ __forceinline__ void wf_path_key(const WfArgs& a, uint32_t p, uint32_t& pixel, uint32_t& sample) {
    if (!STRIPS && CELLS && a.tile_paths == 2u) {
        // Cell by cell (the builds of the path kernel for scenes with two or more simple BVH subtrees: GUESTS): all of the batch's
        // paths of a cell of pixels (WF_CELL_LOG2), every slice and sample, before the next cell's, the cells in the order of
        // wf_tile_to_pix; then the pixels no 8x8 tile covers, sample-major.  A wave's 64 camera rays are then samples of one
        // pixel or of a few neighbours, and the waves of the chip work on one region of the image at a time.  Measured with 2x2 cells
        // against 8x8 tiles of one sample each: final_scene at 8 / 64 / 400 / 2 000 / 10 000 samples per pixel +2.6 / +24 / +5.6 / +2.9 / +2.0 %; scenes with one subtree: random spheres at 1000
        // samples +6.2 %, at 100 samples -7.0 %, Suzanne -1.1 %, which is why they keep the tiles (and their builds do not carry
        // this code: with it Suzanne's build came out 2.3 % slower).
        const uint32_t tiled = (a.width & ~7u) * ((a.n_pixels / a.width) & ~7u), cell_part = tiled * a.n_sub;
        uint32_t q, sub;
        if (p < cell_part) {
            const uint32_t per_cell = (1u << WF_CELL_LOG2) * a.n_sub, c = p / per_cell, rem = p - c * per_cell;
            sub = rem >> WF_CELL_LOG2, q = (c << WF_CELL_LOG2) + (rem & ((1u << WF_CELL_LOG2) - 1u));
        } else {
            const uint32_t r = p - cell_part, rest = a.n_pixels - tiled;
            sub = r / rest, q = tiled + (r - sub * rest);
        }
        pixel = a.pixel0 + wf_tile_to_pix(a, q);
        sample = a.sample_offset + (a.slice_first + sub % a.pass_slices) * a.spt + a.s0 + sub / a.pass_slices;
        return;
    }
    if (!STRIPS) {
        const uint32_t slot = p % a.n_slots, s = a.s0 + p / a.n_slots;
        const uint32_t q = slot % a.n_pixels;
        pixel = a.pixel0 + (a.tile_paths ? wf_tile_to_pix(a, q) : q);
        sample = a.sample_offset + (a.slice_first + slot / a.n_pixels) * a.spt + s;
        return;
    }
    const WfPathPos q = wf_path_pos<true>(a, p);
    pixel = a.pixel0 + q.pix;
    sample = a.sample_offset + (a.slice_first + q.sub % a.pass_slices) * a.spt + a.s0 + q.sub / a.pass_slices;
}

// (RR_WF_TEMPLATES_ONLY: a second translation unit of the library includes this file for builds of the kernel templates and
// leaves the kernels that are no templates to the first.)
#ifndef RR_WF_TEMPLATES_ONLY
// render.rs:95-98: every path of the batch starts at its camera ray with throughput one and full depth.
__global__ void __launch_bounds__(256) k_wf_init(DevScene sc, WfArgs a, WfState st, uint32_t* __restrict__ queue, WfCtl* ctl,
                                                 unsigned long long* counters) {
    // grid-stride: a bounded number of long-lived waves instead of one short wave per 64 paths
    const uint32_t step = gridDim.x * blockDim.x;
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < a.n_paths; p += step) {
        uint32_t pixel, sample;
        wf_path_key(a, p, pixel, sample);
        const uint32_t px = pixel % a.width, py = pixel / a.width;
        Rng g;
        g.start(a.seed, pixel, sample, CAMERA_BOUNCE);
        f3 ro, rd;
        float time;
        camera_ray(sc, a.width, a.height, px, a.height - 1u - py, g, ro, rd, time);
        st.ray_o[p] = make_float4(ro.x, ro.y, ro.z, time);
        st.ray_d[p] = make_float4(rd.x, rd.y, rd.z, __uint_as_float(a.max_depth));
        st.thr[p] = make_float4(1.f, 1.f, 1.f, 0.f);
        queue[p] = p;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctl[0].count = a.n_paths;
        ctl[0].cursor = 0;
        ctl[1].count = ctl[1].cursor = 0;
        ctl[2].count = ctl[2].cursor = 0;
        for (uint32_t x = 0; x < WF_XCDS; ++x) ctl[0].xcur[x] = ctl[1].xcur[x] = ctl[2].xcur[x] = 0;
        if (counters) atomicAdd(&counters[0], static_cast<unsigned long long>(a.n_paths));  // samples started: one per path
    }
}

// pixel_color of render.rs:93-100 for one slot: the batch's sample values added in sample order.
__global__ void __launch_bounds__(256) k_wf_accumulate(WfArgs a, WfState st, float* __restrict__ slice_sums) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.n_slots) return;
    uint32_t pix = t % a.n_pixels, slice = t / a.n_pixels;
    if (a.strip_pixels == 0u && a.tile_paths == 2u) {
        // Paths numbered cell by cell (wf_path_key): neighbouring threads take the (slice, pixel) pairs whose values are
        // neighbours in memory — a cell's pixels, slice by slice — so that a wave reads whole lines in every turn of the
        // loop below (with a thread per pixel of a row it read 16 bytes of every 64: 66 instead of 17 ms on final_scene).
        const uint32_t tiled = (a.width & ~7u) * ((a.n_pixels / a.width) & ~7u), cell_threads = tiled * a.pass_slices;
        if (t < cell_threads) {
            const uint32_t per_cell = (1u << WF_CELL_LOG2) * a.pass_slices, c = t / per_cell, r = t - c * per_cell;
            slice = r >> WF_CELL_LOG2, pix = wf_tile_to_pix(a, (c << WF_CELL_LOG2) + (r & ((1u << WF_CELL_LOG2) - 1u)));
        } else {
            const uint32_t u = t - cell_threads, rest = a.n_pixels - tiled;
            slice = u / rest, pix = wf_tile_to_pix(a, tiled + (u - slice * rest));
        }
    }
    const uint32_t slot = slice * a.n_pixels + pix;
    float* dst = slice_sums + (static_cast<uint64_t>(a.slice_out) * a.n_pixels + slot) * 3u;
    f3 acc = a.s0 == 0 ? mk3(0.f, 0.f, 0.f) : mk3(dst[0], dst[1], dst[2]);
    for (uint32_t sub = slice; sub < a.n_sub; sub += a.pass_slices) {  // this slot's samples of the batch, in sample order
        const float4 c = st.thr[wf_path_index(a, pix, sub)];
        acc = acc + mk3(c.x, c.y, c.z);
    }
    dst[0] = acc.x, dst[1] = acc.y, dst[2] = acc.z;
}
#endif  // RR_WF_TEMPLATES_ONLY

#ifndef RR_WF_FINISH_PARK_BELOW
#define RR_WF_FINISH_PARK_BELOW 20  // k_wf_finish: lanes still descending in a subtree are parked once fewer than this many are left
#endif
#ifndef RR_PATHS_PARK_BELOW
#define RR_PATHS_PARK_BELOW 12  // k_wf_paths: the same (20 until dead paths ended where they die: with the rays that are left, 12 is +1 % on final_scene and the random spheres, +0.3 % on Suzanne; the mesh's k_wf_finish is 12 % slower with it and keeps 20)
#endif
#ifndef RR_WF_FINISH_PARK_MIN_STEPS
#define RR_WF_FINISH_PARK_MIN_STEPS 6  // machine steps a visit of a subtree takes at least, so that parked lanes get on
#endif
#ifndef RR_WF_FINISH_PARK_MIN_FRESH
#define RR_WF_FINISH_PARK_MIN_FRESH 8  // with the queue drained and fewer lanes than this starting a ray, nothing is parked
#endif
#ifndef RR_PATHS_LEAF_WAIT
#define RR_PATHS_LEAF_WAIT 4u  // k_wf_paths: lanes at a leaf of a subtree wait until they are a quarter of the descending lanes (see traverse_parking)
#endif
#ifndef RR_PATHS_CHUNK
#define RR_PATHS_CHUNK 512u  // paths a wave claims at a time
#endif
// The persistent form of the path tracer at the granularity of single paths: a lane follows one path from its camera
// ray to its end like k_render's, but what it claims next is a path, not the rest of a pixel's samples, and it leaves
// the sample's value behind for k_wf_accumulate to add up in sample order (render.rs:93-100).  A (slice, pixel) item
// of k_render is spt samples that one lane must take one after the other; with a few items per lane the waves end
// at very different times (measured on the Cornell box, 125 samples per item, 11 items per lane: 10 % of the run).
// A wave claims RR_PATHS_CHUNK consecutive paths with one atomic (64 lanes next to each other in an image row) and
// hands them to its lanes as their paths end.
// WAVES: waves per SIMD the register allocator leaves room for (4: 128 vector registers, 5: 96).  Measured once the build
// no longer paired f32 operations (which had cost registers): 5 is +9.5 % on the Cornell box and +3.3 % on final_scene, and,
// once parked lanes no longer held a second copy of their ray, +4 % on Suzanne; 6 is worse everywhere (-23 % final_scene,
// -4 % Cornell box).  The host picks 5 for lean scenes and for scenes whose subtrees are all simple.
// LEAN: a build for lean scenes (rr_context::lean_scene) without the code they do not need; with WAVES above the default it
// is for lean scenes without triangles and has no triangle code either.
// GUESTS: a build for scenes with two or more subtrees, all simple: lanes parked at one subtree take the steps of the visits of
// the others (traverse_parking).
template <int COUNT, bool VEC, bool PARK = false, bool GENERAL = true, bool REF_SEM = false, int WAVES = RR_RENDER_WAVES_PER_SIMD, bool LEAN = false,
          bool GUESTS = false>
__global__ void __launch_bounds__(256, WAVES) k_wf_paths(DevScene sc_arg, WfArgs a, float4* __restrict__ values,
                                                                              unsigned int* next_path, unsigned long long* counters) {
    // REF_SEM: a build that is only launched for scenes with the reference's own semantics, so that the switch is a constant
    // to the compiler and not a scalar register (and a lane mask per clipping axis) held through the traversal loops.
    DevScene sc = sc_arg;
    if (REF_SEM) sc.intended = 0u, sc.book = 0u;
    // (the host launches the lean builds only for scenes without media that are not bounded by a bare sphere, without cuboids
    // that are not the constructor's and with instances at most two deep, and those with more waves only if the scene has no
    // triangles either: that code, and the registers it needs, is not in them)
    sc.lean = LEAN ? 1u : 0u;
    sc.no_tri = (LEAN && WAVES > RR_RENDER_WAVES_PER_SIMD) ? 1u : 0u;
    if (PARK) own_registers(sc);
    extern __shared__ uint32_t lds_stack[];
    uint32_t* stk = lds_stack + threadIdx.x;
    constexpr uint32_t stride = WF_BLOCK;  // word w of a lane's stack is at stk[w * stride]: a shift, not a multiplication by a register
    Counts cnt = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t chunk_pos = 0, chunk_end = 0;  // wave-uniform: the unclaimed rest of this wave's chunk
    const uint32_t chunk = a.paths_chunk != 0u ? a.paths_chunk : RR_PATHS_CHUNK;
    bool exhausted = a.n_paths == 0;        // wave-uniform: no unclaimed paths are left for anyone
    bool path_active = false;
    uint32_t p = 0, pixel = 0, sample = 0, depth_left = 0, bounce = 0;
    f3 ro = mk3(0.f, 0.f, 0.f), rd = mk3(0.f, 0.f, 0.f), thr = mk3(1.f, 1.f, 1.f);
    float time = 0.f;
    // PARK (scenes with BVH subtrees): the few lanes with the deepest descents do not hold the wave in a subtree;
    // they keep their traversal state and come back to it in later turns (traverse_parking)
    ParkArea pa;
    uint32_t park = NONE;  // the slot and op this lane's path is parked at
    uint32_t draws = 0;
    if (PARK) pa.init(lds_stack + a.stack_words * WF_BLOCK);
    // The loop is uniform per wave (its only exit is a wave vote); lanes differ through flags.  That keeps the
    // scene-program interpreter inside traverse() in uniform control flow, on the scalar unit.
#ifdef RR_SECTION_TIMERS
    cnt.sec_last = __builtin_amdgcn_s_memtime();
#endif
    for (;;) {
        RR_SEC_MARK(cnt, 7);
        const uint64_t idle = RR_BALLOT(!path_active);
        if (!exhausted && idle != 0) {
            if (chunk_pos == chunk_end) {
                uint32_t base = 0;
                if ((threadIdx.x & 63u) == 0) base = atomicAdd(next_path, chunk);
                base = __builtin_amdgcn_readfirstlane(base);
                chunk_pos = min(base, a.n_paths);
                chunk_end = min(base + chunk, a.n_paths);
            }
            const uint32_t take = min(static_cast<uint32_t>(__popcll(idle)), chunk_end - chunk_pos);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(idle >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(idle), 0u));
            if (!path_active && rank < take) {
                p = chunk_pos + rank;
                wf_path_key<false, GUESTS>(a, p, pixel, sample);  // (the host asks for cells in the guests builds only)
                Rng g;
                g.start(a.seed, pixel, sample, CAMERA_BOUNCE);
                camera_ray(sc, a.width, a.height, pixel % a.width, a.height - 1u - pixel / a.width, g, ro, rd, time);
                thr = mk3(1.f, 1.f, 1.f);
                depth_left = a.max_depth;
                bounce = 0;
                path_active = true;
                cnt.samples++;
                if (PARK) draws = 0;
            }
            chunk_pos += take;
            exhausted = chunk_pos == chunk_end && chunk_end == a.n_paths;  // the counter only grows
        }
        if (!RR_ANY(path_active)) {
            if (exhausted) break;
            continue;
        }
        // ---- one ray_color call (render.rs:15-41) for every lane that has a path
        RR_SEC_MARK(cnt, 0);
        const bool run = path_active && depth_left != 0;
        const bool was_parked = PARK && park != NONE;
        if (run && !was_parked) cnt.rays++;
        RR_SEC_ADD(cnt, 10, 1ull);
        RR_SEC_ADD(cnt, 11, static_cast<unsigned long long>(__popcll(RR_BALLOT(run))));
        Rng g;
        g.start(a.seed, pixel, sample, bounce);
        float t;
        uint32_t ref;
        bool parked = false;
        if (PARK) {
            g.n = draws;
            if (g.n & 3u) g.refill();
            const uint32_t n_fresh = static_cast<uint32_t>(__popcll(RR_BALLOT(run && !was_parked)));
            const uint32_t park_below = (n_fresh >= RR_WF_FINISH_PARK_MIN_FRESH || !exhausted) ? RR_PATHS_PARK_BELOW : 0u;
            traverse_parking<COUNT, GENERAL, GUESTS && !GENERAL>(sc, ro, rd, time, 0.001f, kF32Max, run, park, pa, g, stk, stride, park_below, RR_WF_FINISH_PARK_MIN_STEPS, RR_PATHS_LEAF_WAIT, cnt, t, ref);
            draws = g.n;
            parked = run && park != NONE;
        } else {
            traverse<COUNT, VEC>(sc, ro, rd, time, 0.001f, kF32Max, run, g, stk, stride, t, ref, cnt);
        }
        RR_SEC_MARK(cnt, 1);
        f3 terminal = mk3(0.f, 0.f, 0.f);
        bool go_on = false;
        if (run && !parked) {
            Bounce b;
            HitFetch hf{};
            if (ref != NONE) hf = hit_fetch(sc, ref);
            // (the eager fetch of reconstruct_hit: +1 % in the lean build of final_scene, -0.8 % on Suzanne)
            go_on = shade<!PARK || LEAN, true>(sc, ro, rd, time, t, ref, hf.meta, g, b, hf.r0, hf.r1);
            if (go_on) {
                ro = b.o;
                rd = b.d;
                thr = thr * b.value;
                depth_left -= 1;
                bounce += 1;
                if (PARK) draws = 0;
                if (a.trace_dead == 0u && dead_path(thr)) go_on = false;  // (terminal is 0: the sample's value is 0)
            } else {
                terminal = b.value;
            }
        }
        RR_SEC_MARK(cnt, 5);
        if (path_active && !go_on && !parked) {
            const f3 c = thr * terminal;  // the sample's value (render.rs:99 adds it to the pixel)
            values[p] = make_float4(c.x, c.y, c.z, 0.f);
            path_active = false;
        }
        RR_SEC_MARK(cnt, 6);
    }
#ifdef RR_SECTION_TIMERS
    if ((threadIdx.x & 63u) == 0)
        for (int i = 0; i < 16; ++i) atomicAdd(&g_sec[i], cnt.sec[i]);
#endif
    flush_counts(cnt, counters);
}

// Providers for the traversal machine: what it rarely needs is read from the slot when it does.
struct WfWorld {
    const WfState& st;
    uint32_t p;
    __device__ __forceinline__ void ray(f3& o, f3& d) const {
        const float4 a = st.ray_o[p], b = st.ray_d[p];
        o = mk3(a.x, a.y, a.z);
        d = mk3(b.x, b.y, b.z);
    }
    __device__ __forceinline__ float time() const { return st.ray_o[p].w; }
};
struct WfDraw {
    const WfState& st;
    const WfArgs& a;
    uint32_t p;
    uint32_t n;  // draws taken from this bounce's stream so far
    __device__ __forceinline__ float random() {
        const uint32_t depth_left = __float_as_uint(st.ray_d[p].w);
        uint32_t pixel, sample;
        wf_path_key(a, p, pixel, sample);
        Rng g;
        g.start(a.seed, pixel, sample, a.max_depth - depth_left);
        g.n = n;
        if (n & 3u) g.refill();
        const float v = g.random();
        n = g.n;
        return v;
    }
};

// WIDE: every BVH node is walked with the wide node move (trav_node_wide): the host asks for it only when all of them are of a
// culling class, none has anything but BVH nodes, spheres, rects, triangles and cuboids below, and the semantics are the reference's.
// PLAIN: the scene has no moving spheres, instances or media (trav_leaf).
template <int COUNT, bool REF_SEM = false, bool WIDE = false, bool PLAIN = false>  // REF_SEM: as for k_wf_paths
__global__ void __launch_bounds__(256, RR_WF_TRAVERSE_WAVES_PER_SIMD)
    k_wf_traverse(DevScene sc_arg, WfArgs a, WfState st, const uint32_t* __restrict__ queue, WfCtl* ctl, uint32_t root,
                  unsigned long long* counters) {
    DevScene sc = sc_arg;
    if (REF_SEM) sc.intended = 0u, sc.book = 0u;
    sc.no_tri = sc.lean = 0u;
    extern __shared__ uint32_t lds_stack[];
    uint32_t* stk = lds_stack + threadIdx.x;
    constexpr uint32_t stride = WF_BLOCK;
    const uint32_t total = ctl->count;
    Counts cnt = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    Trav m;
    m.begin(NONE, mk3(0.f, 0.f, 0.f), mk3(1.f, 1.f, 1.f), 0.f, 0.f, NONE);
    bool active = false;
    bool exhausted = total == 0;  // wave-uniform: the queue has no unclaimed entries left
    uint32_t p = 0, draws = 0;
    // A wave claims queue entries a chunk at a time and hands them to its lanes as they fall idle:
    // one atomic on the shared cursor per chunk, not per refill (atomics on one address serialise,
    // and at a few nanoseconds each a claim per refill would cost as much as the traversal itself).
    // Chunks are sized so that every wave makes several claims, which evens out the finish.
    const uint32_t n_waves = gridDim.x * (blockDim.x / 64u);
    uint32_t chunk_pos = 0, chunk_end = 0;  // wave-uniform: the unclaimed rest of this wave's chunk
    // The queue is cut into one contiguous part per die (XCD): a wave claims from the part of the die it runs on and, once that
    // is used up, from the next die's.  Neighbours in the queue are neighbours in the image (wf_path_key), so a die's rays walk
    // the same region of the scene and its L2 (4 MB, not shared between dies) serves more of their fetches than when every die
    // sees every region.  A claim is a fraction of what is left of the part (at least 64 entries, at most WfArgs::claim_max), so the claims
    // shrink towards the end and the waves finish together whatever the rays of the last entries cost.
    const uint32_t xcd = __builtin_amdgcn_s_getreg((3u << 11) | 20u) & (WF_XCDS - 1u);  // HW_REG_XCC_ID, bits 3..0
    uint32_t part = xcd, parts_tried = 0;
    // Every lane leaves through the same wave-wide vote, after the queue is drained and the last
    // ray of the wave has finished.
    for (;;) {
        const uint64_t idle = RR_BALLOT(!active);
        if (!exhausted && __popcll(idle) >= RR_WF_REFILL_IDLE) {
            while (chunk_pos == chunk_end && parts_tried < WF_XCDS) {
                const uint32_t lo = static_cast<uint32_t>(static_cast<uint64_t>(total) * part / WF_XCDS),
                               hi = static_cast<uint32_t>(static_cast<uint64_t>(total) * (part + 1u) / WF_XCDS);
                uint32_t base = 0, chunk = 0;
                if ((threadIdx.x & 63u) == 0) {
                    const uint32_t seen = min(__hip_atomic_load(&ctl->xcur[part], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), hi - lo);
                    chunk = min(max((hi - lo - seen) / (n_waves / WF_XCDS * 2u + 1u), 64u), a.claim_max);
                    base = atomicAdd(&ctl->xcur[part], chunk);
                }
                base = __builtin_amdgcn_readfirstlane(base);
                chunk = __builtin_amdgcn_readfirstlane(chunk);
                if (base >= hi - lo) {  // this part is used up (its counter only grows): on to the next die's
                    part = (part + 1u) & (WF_XCDS - 1u);
                    parts_tried += 1;
                    continue;
                }
                chunk_pos = lo + base;
                chunk_end = min(chunk_pos + chunk, hi);
            }
            exhausted = chunk_pos == chunk_end;  // every part is used up: nothing is left for anyone
            const uint32_t want = static_cast<uint32_t>(__popcll(idle));
            const uint32_t take = min(want, chunk_end - chunk_pos);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(idle >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(idle), 0u));
            const uint32_t entry = chunk_pos + rank;
            chunk_pos += take;
            if (!active && rank < take) {
                p = queue[entry];
                const float4 ro = st.ray_o[p], rd = st.ray_d[p];
                if (__float_as_uint(rd.w) != 0u) {  // ray_color with depth 0 intersects nothing (render.rs:16)
                    m.begin(root, mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), 0.001f, kF32Max, NONE);
                    draws = 0;
                    active = true;
                }
            }
    }
        if (!RR_ANY(active)) {
            if (exhausted) break;
            continue;
        }
        // Scheduling of the lanes' moves: every lane at a BVH node takes it; the (dearer) leaf move is
        // held back until enough lanes wait at a leaf for it to be worth the wave's while, or until
        // waiting would only idle them; returns are taken as they come.
        const WfWorld world{st, p};
        WfDraw draw{st, a, p, draws};
        if (COUNT) cnt.lane_steps += active ? 1u : 0u, cnt.wave_steps += (threadIdx.x & 63u) == 0 ? 1u : 0u;
        if (active) {
            if (WIDE) trav_node_wide<COUNT>(sc, m, stk, stride, cnt);
            else trav_node<COUNT>(sc, m, stk, stride, cnt);
        }
        const bool at_leaf = active && trav_at_leaf(m);
        const uint32_t n_leaf = static_cast<uint32_t>(__popcll(RR_BALLOT(at_leaf)));
        const uint32_t n_node = static_cast<uint32_t>(__popcll(RR_BALLOT(active && trav_at_node(m))));
        if (n_leaf != 0 && (n_leaf >= RR_WF_LEAF_LANES || n_leaf >= n_node)) {
            if (at_leaf) trav_leaf<COUNT, WfWorld, PLAIN>(sc, m, world, stk, stride, cnt);
        }
        if (active) {
            trav_pop<COUNT, WfWorld, WfDraw, WIDE, PLAIN>(sc, m, world, draw, stk, stride, cnt);
            draws = draw.n;
            if (m.finished()) {
                // the hit's material and owner are fetched here, where the latency hides behind other rays' work
                const HitMeta hm = m.cur == NONE ? HitMeta{0u, 0u} : hit_meta(sc, m.cur);
                st.hit[p] = make_uint4(__float_as_uint(m.tmax), m.cur, hm.mat, hm.owner);
                if (draws != 0) reinterpret_cast<uint32_t*>(&st.thr[p])[3] = draws;
                active = false;
            }
        }
    }
    flush_counts(cnt, counters);
}

#ifndef RR_WF_TEMPLATES_ONLY
__global__ void __launch_bounds__(256, RR_WF_SHADE_WAVES_PER_SIMD) k_wf_shade(DevScene sc, WfArgs a, WfState st, const uint32_t* __restrict__ queue_in,
                                                  const WfCtl* ctl_in, uint32_t* __restrict__ queue_out, WfCtl* ctl_out, WfCtl* ctl_clear,
                                                  unsigned long long* counters) {
    const uint32_t total = ctl_in->count;
    const uint32_t step = gridDim.x * blockDim.x;
    Counts cnt = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    // Paths that go on are collected per wave in LDS and appended to the next queue a batch at a time:
    // one atomic on the shared counter per WF_STAGE-64 or more entries instead of one per wave iteration.
    __shared__ uint32_t stage_all[4][WF_STAGE];
    uint32_t* stage = stage_all[threadIdx.x >> 6];
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t staged = 0;  // wave-uniform
    auto flush = [&]() {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&ctl_out->count, staged);
        base = __builtin_amdgcn_readfirstlane(base);
        __builtin_amdgcn_wave_barrier();  // the lanes read what other lanes of the wave have staged
        for (uint32_t k = lane; k < staged; k += 64u) queue_out[base + k] = stage[k];
        __builtin_amdgcn_wave_barrier();
        staged = 0;
    };
    // Whole waves iterate together (step is a multiple of 64), so the votes below are well defined.
    // The loop is a three-stage pipeline per lane — queue entry two iterations ahead, path state one
    // iteration ahead, shading now — so that the loads of the next paths are in flight while this one
    // is shaded: the kernel is bound by the latency of its dependent loads, not by their volume.
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t p1 = i < total ? queue_in[i] : 0u;
    uint32_t p2 = (i + step < total && i + step > i) ? queue_in[i + step] : 0u;
    float4 qo1 = st.ray_o[p1], qd1 = st.ray_d[p1], qt1 = st.thr[p1];
    uint4 qh1 = st.hit[p1];
    for (; (i & ~63u) < total; i += step) {
        bool keep = false;
        const uint32_t p = p1;
        const float4 qo = qo1, qd = qd1, qt = qt1;
        const uint4 h = qh1;
        p1 = p2;
        qo1 = st.ray_o[p1], qd1 = st.ray_d[p1], qt1 = st.thr[p1], qh1 = st.hit[p1];  // entry 0 when there is no next path: harmless
        const uint32_t i2 = i + 2u * step;
        p2 = (i2 < total && i2 > i) ? queue_in[i2] : 0u;
        if (i < total) {
            const f3 ro = mk3(qo.x, qo.y, qo.z), rd = mk3(qd.x, qd.y, qd.z);
            f3 thr = mk3(qt.x, qt.y, qt.z);
            const float time = qo.w;
            uint32_t depth_left = __float_as_uint(qd.w);
            f3 terminal = mk3(0.f, 0.f, 0.f);
            if (depth_left != 0) {  // one ray_color call (render.rs:15-41) from its scene.intersect on
                cnt.rays++;
                uint32_t pixel, sample;
                wf_path_key(a, p, pixel, sample);
                Rng g;
                g.start(a.seed, pixel, sample, a.max_depth - depth_left);
                g.n = __float_as_uint(qt.w);
                if (g.n & 3u) g.refill();
                Bounce b;
#ifdef RR_ABLATE_WF_SHADE  // timing-only builds: wrong images, price the sections of the kernel
                keep = h.y != NONE && (p & 3u) != 0u;
                b.o = ro + rd * __uint_as_float(h.x), b.d = rd, b.value = mk3(0.5f, 0.5f, g.random());
#if RR_ABLATE_WF_SHADE >= 2
                if (h.y != NONE) {
                    const HitRec rec = reconstruct_hit(sc, ro, rd, time, __uint_as_float(h.x), h.y, HitMeta{h.z, h.w}, false);
                    b.o = rec.p, b.d = rec.n;
#if RR_ABLATE_WF_SHADE >= 3
                    const DevMaterial mm = sc.k->materials[h.z];
                    b.value = mk3(mm.r, mm.g, mm.b);
#endif
#if RR_ABLATE_WF_SHADE >= 4
                    const f3 cd = random_cosine_direction(g);
                    b.d = b.d + cd;
#endif
#if RR_ABLATE_WF_SHADE >= 5
                    b.value.x = light_pdf_value(sc, rec.p, b.d);
#endif
                }
#endif
#else
                keep = shade(sc, ro, rd, time, __uint_as_float(h.x), h.y, HitMeta{h.z, h.w}, g, b);
#endif
                if (keep) {
                    thr = thr * b.value;
                    if (a.trace_dead == 0u && dead_path(thr)) {
                        keep = false;  // (terminal stays 0: the sample's value is 0)
                    } else {
                        depth_left -= 1;
                        st.ray_o[p] = make_float4(b.o.x, b.o.y, b.o.z, time);
                        st.ray_d[p] = make_float4(b.d.x, b.d.y, b.d.z, __uint_as_float(depth_left));
                        st.thr[p] = make_float4(thr.x, thr.y, thr.z, 0.f);
                    }
                } else {
                    terminal = b.value;
                }
            }
            if (!keep) {
                const f3 c = thr * terminal;  // the sample's value (render.rs:99 adds it to the pixel)
                st.thr[p] = make_float4(c.x, c.y, c.z, 0.f);
            }
        }
        const uint64_t mask = RR_BALLOT(keep);
        if (mask) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
            if (keep) stage[staged + rank] = p;
            staged += static_cast<uint32_t>(__popcll(mask));
            if (staged > WF_STAGE - 64u) flush();
        }
    }
    if (staged != 0) flush();
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // the queue two bounces back is free again
        ctl_clear->count = 0;
        ctl_clear->cursor = 0;
        for (uint32_t x = 0; x < WF_XCDS; ++x) ctl_clear->xcur[x] = 0;
    }
    flush_counts(cnt, counters);
}
#endif  // RR_WF_TEMPLATES_ONLY

#ifndef RR_WF_FINISH_WAVES_PER_SIMD
#define RR_WF_FINISH_WAVES_PER_SIMD 4
#endif

// The rest of a batch after its first bounces.  With every bounce fewer paths are left, and a bounce costs
// two launches that each last as long as their slowest ray; this kernel runs every remaining path to its end
// instead, a lane per path, bounce after bounce like k_render (same scene program and machine, same shade, same random streams).  Path
// lengths differ widely here (these are the long ones), so a lane whose path has ended takes the next
// unclaimed queue entry at once: the wave's loop is uniform, one bounce per turn for every lane that has a
// path, and the queue is claimed a chunk at a time per wave as in k_wf_traverse.  (Letting the lanes of a
// wave run out of step as well — traversal moves, shading and refills whenever enough lanes wait for one —
// was measured equal on the Suzanne scene and 4-10 % slower on final_scene.)
template <int COUNT, bool GENERAL = true, bool REF_SEM = false>  // GENERAL = false: without the general subtree machine; REF_SEM: as for k_wf_paths
__global__ void __launch_bounds__(256, RR_WF_FINISH_WAVES_PER_SIMD) k_wf_finish(DevScene sc_arg, WfArgs a, WfState st, const uint32_t* __restrict__ queue,
                                                                                WfCtl* ctl, unsigned long long* counters) {
    DevScene sc = sc_arg;
    if (REF_SEM) sc.intended = 0u, sc.book = 0u;
    sc.no_tri = sc.lean = 0u;
    own_registers(sc);
    extern __shared__ uint32_t lds_stack[];
    uint32_t* stk = lds_stack + threadIdx.x;
    constexpr uint32_t stride = WF_BLOCK;
    const uint32_t total = ctl->count;
    Counts cnt = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const uint32_t n_waves = gridDim.x * (blockDim.x / 64u);
    const uint32_t chunk = min(max(total / (n_waves * 8u), 16u), 1024u);
    uint32_t chunk_pos = 0, chunk_end = 0;  // wave-uniform: the unclaimed rest of this wave's chunk
    bool exhausted = total == 0;            // wave-uniform: no unclaimed entries are left for anyone
    bool have = false;
    uint32_t p = 0, depth_left = 0, pixel = 0, sample = 0, draws = 0;
    f3 ro = mk3(0.f, 0.f, 0.f), rd = mk3(1.f, 1.f, 1.f), thr = mk3(0.f, 0.f, 0.f);
    float time = 0.f;
    ParkArea pa;
    pa.init(lds_stack + a.stack_words * WF_BLOCK);
    uint32_t park = NONE;  // the slot and op this lane's path is parked at
    for (;;) {
        const uint64_t idle = RR_BALLOT(!have);
        if (!exhausted && idle != 0) {
            if (chunk_pos == chunk_end) {
                uint32_t base = 0;
                if ((threadIdx.x & 63u) == 0) base = atomicAdd(&ctl->cursor, chunk);
                base = __builtin_amdgcn_readfirstlane(base);
                chunk_pos = min(base, total);
                chunk_end = min(base + chunk, total);
            }
            const uint32_t take = min(static_cast<uint32_t>(__popcll(idle)), chunk_end - chunk_pos);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(idle >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(idle), 0u));
            const uint32_t entry = chunk_pos + rank;
            chunk_pos += take;
            exhausted = chunk_pos == chunk_end && chunk_end == total;  // the cursor only grows
            if (!have && rank < take) {
                p = queue[entry];
                const float4 qo = st.ray_o[p], qd = st.ray_d[p], qt = st.thr[p];
                ro = mk3(qo.x, qo.y, qo.z), rd = mk3(qd.x, qd.y, qd.z), thr = mk3(qt.x, qt.y, qt.z);
                time = qo.w;
                depth_left = __float_as_uint(qd.w);
                wf_path_key(a, p, pixel, sample);
                have = true;
                draws = 0;
            }
        }
        if (!RR_ANY(have)) {
            if (exhausted) break;
            continue;
        }
        // One ray_color call (render.rs:15-41) for every lane that has a path.  The lanes start their rays
        // together, so the part of the tree every ray visits is walked by the scene program, with scalar
        // control flow and scalar loads, and only the BVH subtrees run on the per-lane machine.  The few lanes
        // with the deepest descents are parked in their subtree rather than waited for (traverse_parking):
        // they finish it in later turns, beside the lanes that enter it then, unless the wave has little else to do.
        const bool run = have && depth_left != 0;
        if (run && park == NONE) cnt.rays++;
        Rng g;
        g.start(a.seed, pixel, sample, a.max_depth - depth_left);
        g.n = draws;
        if (g.n & 3u) g.refill();
        const uint32_t n_fresh = static_cast<uint32_t>(__popcll(RR_BALLOT(run && park == NONE)));
        const uint32_t park_below = (n_fresh >= RR_WF_FINISH_PARK_MIN_FRESH || !exhausted) ? RR_WF_FINISH_PARK_BELOW : 0u;
        float t_hit;
        uint32_t ref_hit;
        traverse_parking<COUNT, GENERAL>(sc, ro, rd, time, 0.001f, kF32Max, run, park, pa, g, stk, stride, park_below, RR_WF_FINISH_PARK_MIN_STEPS, 0u, cnt, t_hit, ref_hit);
        draws = g.n;
        f3 terminal = mk3(0.f, 0.f, 0.f);
        bool go_on = false;
        const bool parked = run && park != NONE;
        if (run && !parked) {
            Bounce b;
            go_on = shade(sc, ro, rd, time, t_hit, ref_hit, ref_hit == NONE ? HitMeta{0u, 0u} : hit_meta(sc, ref_hit), g, b);
            if (go_on) {
                ro = b.o;
                rd = b.d;
                thr = thr * b.value;
                depth_left -= 1;
                draws = 0;
                if (a.trace_dead == 0u && dead_path(thr)) go_on = false;  // (terminal is 0: the sample's value is 0)
            } else {
                terminal = b.value;
            }
        }
        if (have && !go_on && !parked) {
            const f3 c = thr * terminal;  // the sample's value (render.rs:99 adds it to the pixel)
            st.thr[p] = make_float4(c.x, c.y, c.z, 0.f);
            have = false;
        }
    }
    flush_counts(cnt, counters);
}

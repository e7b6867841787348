// Wavefront form of the path tracer for scenes with BVH subtrees.  Included only by rr_hip.hip,
// after the definitions of flush_counts and the device code of rr_device.h.
//
// The persistent kernel k_render keeps a lane on its path from camera ray to termination, which
// is ideal while all 64 rays of a wave take the same steps through the scene.  Below a BVH they
// do not: traversal lengths differ by an order of magnitude between neighbouring rays, and a wave
// is as slow as its longest ray.  Here a path's state lives in device memory between bounces and
// each bounce is two kernels over a compacted queue of live paths:
//
//   k_wf_traverse  persistent waves; a lane takes a ray from the queue, runs the traversal machine
//                  over the whole scene tree, stores (t, primitive) and takes the next ray, so a
//                  wave never waits for its longest ray with idle lanes; the kernel holds nothing
//                  but traversal state and runs at twice the occupancy of k_render
//   k_wf_shade     one thread per live path: ray_color's work after scene.intersect
//                  (render.rs:19-40); a path that goes on is appended to the next bounce's queue,
//                  one that ends leaves its sample value (throughput x terminal colour) behind
//
// Paths are independent, so many samples of a pixel are in flight at once: a pass covers a batch
// of consecutive samples of every (slice, pixel) pair (a "slot"), runs until every path has
// ended, and k_wf_accumulate then adds each slot's sample values to its pixel sum in sample
// order (render.rs:93-100).  Every random stream is keyed by (pixel, sample, bounce) and every
// f32 sum is taken in the reference's order, so this form and k_render produce the same bits.
#pragma once

#ifndef RR_WF_TRAVERSE_WAVES_PER_SIMD
#define RR_WF_TRAVERSE_WAVES_PER_SIMD 8
#endif
#ifndef RR_WF_SHADE_WAVES_PER_SIMD
#define RR_WF_SHADE_WAVES_PER_SIMD 3
#endif
#ifndef RR_WF_REFILL_IDLE
#define RR_WF_REFILL_IDLE 16  // a wave fetches new rays once this many of its lanes have none
#endif

struct WfState {     // one entry per path in each array
    float4* ray_o;   // (origin.xyz, time)
    float4* ray_d;   // (direction.xyz, ray_color depth left)
    float4* thr;     // (throughput.xyz, -) while the path lives, its sample value once it has ended
    uint4* hit;      // (t, primitive ref, draws the traversal took from the bounce's stream, -)
};
struct WfCtl {
    uint32_t count;   // entries in the queue
    uint32_t cursor;  // next entry no traversal wave has claimed yet
};
struct WfArgs {
    uint32_t width, height, n_pixels;
    uint32_t n_slots;      // slices in this pass * n_pixels
    uint32_t n_paths;      // n_slots * samples in this batch; path p is sample s0 + p / n_slots of slot p % n_slots
    uint32_t s0;           // first sample of the batch within its slice
    uint32_t spt;          // samples per slice and pixel (render.rs:130)
    uint32_t slice_first;  // absolute index of the pass's first slice
    uint32_t slice_out;    // index of that slice inside the output buffer
    uint32_t max_depth;
    uint32_t sample_offset;
    uint64_t seed;
};

// (pixel, per-pixel sample number) of a path: the key of its random streams.
__device__ __forceinline__ void wf_path_key(const WfArgs& a, uint32_t p, uint32_t& pixel, uint32_t& sample) {
    const uint32_t slot = p % a.n_slots, s = a.s0 + p / a.n_slots;
    pixel = slot % a.n_pixels;
    sample = a.sample_offset + (a.slice_first + slot / a.n_pixels) * a.spt + s;
}

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
        if (counters) atomicAdd(&counters[0], static_cast<unsigned long long>(a.n_paths));  // samples started: one per path
    }
}

// pixel_color of render.rs:93-100 for one slot: the batch's sample values added in sample order.
__global__ void __launch_bounds__(256) k_wf_accumulate(WfArgs a, WfState st, float* __restrict__ slice_sums) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= a.n_slots) return;
    float* dst = slice_sums + (static_cast<uint64_t>(a.slice_out) * a.n_pixels + slot) * 3u;
    f3 acc = a.s0 == 0 ? mk3(0.f, 0.f, 0.f) : mk3(dst[0], dst[1], dst[2]);
    for (uint32_t p = slot; p < a.n_paths; p += a.n_slots) {
        const float4 c = st.thr[p];
        acc = acc + mk3(c.x, c.y, c.z);
    }
    dst[0] = acc.x, dst[1] = acc.y, dst[2] = acc.z;
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

template <bool COUNT>
__global__ void __launch_bounds__(256, RR_WF_TRAVERSE_WAVES_PER_SIMD)
    k_wf_traverse(DevScene sc, WfArgs a, WfState st, const uint32_t* __restrict__ queue, WfCtl* ctl, uint32_t root,
                  unsigned long long* counters) {
    extern __shared__ uint32_t lds_stack[];
    uint32_t* stk = lds_stack + threadIdx.x;
    const uint32_t stride = blockDim.x;
    const uint32_t total = ctl->count;
    Counts cnt = {0, 0, 0, 0, 0, 0, 0};
    Trav m;
    m.begin(NONE, mk3(0.f, 0.f, 0.f), mk3(1.f, 1.f, 1.f), 0.f, 0.f, NONE);
    bool active = false;
    bool exhausted = total == 0;  // wave-uniform: the queue has no unclaimed entries left
    uint32_t p = 0, draws = 0;
    // Every lane leaves through the same wave-wide vote, after the queue is drained and the last
    // ray of the wave has finished.
    for (;;) {
        const uint64_t idle = __ballot(!active);
        if (!exhausted && __popcll(idle) >= RR_WF_REFILL_IDLE) {
            const uint32_t want = static_cast<uint32_t>(__popcll(idle));
            uint32_t base = 0;
            if ((threadIdx.x & 63u) == 0) base = atomicAdd(&ctl->cursor, want);
            base = __builtin_amdgcn_readfirstlane(base);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(idle >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(idle), 0u));
            const uint32_t entry = base + rank;
            if (!active && base < total && entry < total) {
                p = queue[entry];
                const float4 ro = st.ray_o[p], rd = st.ray_d[p];
                if (__float_as_uint(rd.w) != 0u) {  // ray_color with depth 0 intersects nothing (render.rs:16)
                    m.begin(root, mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), 0.001f, kF32Max, NONE);
                    draws = 0;
                    active = true;
                }
            }
            exhausted = base >= total || total - base <= want;
        }
        if (!__any(active)) {
            if (exhausted) break;
            continue;
        }
        if (active) {
            const WfWorld world{st, p};
            WfDraw draw{st, a, p, draws};
            trav_step<COUNT>(sc, m, world, draw, stk, stride, cnt);
            draws = draw.n;
            if (m.finished()) {
                st.hit[p] = make_uint4(__float_as_uint(m.tmax), m.cur, draws, 0u);
                active = false;
            }
        }
    }
    flush_counts(cnt, counters);
}

__global__ void __launch_bounds__(256, RR_WF_SHADE_WAVES_PER_SIMD) k_wf_shade(DevScene sc, WfArgs a, WfState st, const uint32_t* __restrict__ queue_in,
                                                  const WfCtl* ctl_in, uint32_t* __restrict__ queue_out, WfCtl* ctl_out, WfCtl* ctl_clear,
                                                  unsigned long long* counters) {
    const uint32_t total = ctl_in->count;
    const uint32_t step = gridDim.x * blockDim.x;
    Counts cnt = {0, 0, 0, 0, 0, 0, 0};
    // whole waves iterate together (step is a multiple of 64), so the votes below are well defined
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; (i & ~63u) < total; i += step) {
        bool keep = false;
        uint32_t p = 0;
        if (i < total) {
            p = queue_in[i];
            const float4 qo = st.ray_o[p], qd = st.ray_d[p], qt = st.thr[p];
            const f3 ro = mk3(qo.x, qo.y, qo.z), rd = mk3(qd.x, qd.y, qd.z);
            f3 thr = mk3(qt.x, qt.y, qt.z);
            const float time = qo.w;
            uint32_t depth_left = __float_as_uint(qd.w);
            f3 terminal = mk3(0.f, 0.f, 0.f);
            if (depth_left != 0) {  // one ray_color call (render.rs:15-41) from its scene.intersect on
                cnt.rays++;
                const uint4 h = st.hit[p];
                uint32_t pixel, sample;
                wf_path_key(a, p, pixel, sample);
                Rng g;
                g.start(a.seed, pixel, sample, a.max_depth - depth_left);
                g.n = h.z;
                if (h.z & 3u) g.refill();
                Bounce b;
                keep = shade(sc, ro, rd, time, __uint_as_float(h.x), h.y, g, b);
                if (keep) {
                    thr = thr * b.weight;
                    depth_left -= 1;
                    st.ray_o[p] = make_float4(b.o.x, b.o.y, b.o.z, time);
                    st.ray_d[p] = make_float4(b.d.x, b.d.y, b.d.z, __uint_as_float(depth_left));
                    st.thr[p] = make_float4(thr.x, thr.y, thr.z, 0.f);
                } else {
                    terminal = b.terminal;
                }
            }
            if (!keep) {
                const f3 c = thr * terminal;  // the sample's value (render.rs:99 adds it to the pixel)
                st.thr[p] = make_float4(c.x, c.y, c.z, 0.f);
            }
        }
        const uint64_t mask = __ballot(keep);
        if (mask) {
            uint32_t base = 0;
            if ((threadIdx.x & 63u) == 0) base = atomicAdd(&ctl_out->count, static_cast<uint32_t>(__popcll(mask)));
            base = __builtin_amdgcn_readfirstlane(base);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
            if (keep) queue_out[base + rank] = p;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // the queue two bounces back is free again
        ctl_clear->count = 0;
        ctl_clear->cursor = 0;
    }
    flush_counts(cnt, counters);
}

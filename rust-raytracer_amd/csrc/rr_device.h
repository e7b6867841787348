// Device-side scene layout and the gfx950 path-tracing code (traversal machine,
// hit reconstruction, shading, camera, RNG).  Included only by rr_hip.hip.
//
// Arithmetic contract: every f32 operation below is a single IEEE operation in the
// order the reference's Rust evaluates it (build with -ffp-contract=off; hipcc's
// default correctly-rounded f32 divide and sqrt).  Transcendentals (sin, cos, acos,
// atan2, log10) come from the device math library and may differ from a host libm
// in the last bits.
#pragma once

#ifndef __HIPCC_RTC__  // (the runtime compiler brings its own runtime header and fixed-width integer types)
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

#include "rr_hip.h"  // include/rr_hip.h (the build names the directory)

namespace rrdev {

// ------------------------------------------------------------ packed refs
// Device refs: index in bits 25..0, tag in bits 29..26, two spare bits 31..30.
// In a BVH node the spare bits of `left` hold the split axis and those of `right` the node's
// class (BVH_CLASS_*), so a node is exactly two 16-byte loads.
constexpr uint32_t IDX_BITS = 26;
constexpr uint32_t IDX_MASK = (1u << IDX_BITS) - 1u;
constexpr uint32_t NONE = 0xFFFFFFFFu;
__host__ __device__ inline uint32_t dref(uint32_t tag, uint32_t idx, uint32_t extra = 0) {
    return idx | (tag << IDX_BITS) | (extra << 30);
}
__host__ __device__ inline uint32_t dtag(uint32_t r) { return (r >> IDX_BITS) & 15u; }
__host__ __device__ inline uint32_t didx(uint32_t r) { return r & IDX_MASK; }
__host__ __device__ inline uint32_t dextra(uint32_t r) { return r >> 30; }

constexpr uint32_t MAX_CHAIN = 8;  // deepest instance nesting the device layout holds

constexpr uint32_t REF_MASK = (1u << 30) - 1u;
// Spare bits of a node's `right`: how the production kernels may treat the node.
//   NO_BOX      no bounding box: both children are always visited (collider.rs:538-541)
//   REFERENCE   the reference's box test decides, every visit is made in the reference's order
//   CULL_RANGE  no constant medium below and the box holds everything below: a ray whose segment
//               [t_min, t_max] misses the box (grown by the scene's margin) cannot be accepted below
//   CULL_ANY_T  the same with triangles below, which accept any t >= 0 (collider.rs:627): the
//               segment is [0, inf)
// In the last two classes every intersect call below is a pure function of (ray, t range).
constexpr uint32_t BVH_CLASS_NO_BOX = 0, BVH_CLASS_REFERENCE = 1, BVH_CLASS_CULL_RANGE = 2, BVH_CLASS_CULL_ANY_T = 3;

// Stack frames end in a word laid out like a ref.  A geometry tag in bits 29..26 makes it the
// one-word frame of a BVH node whose first child is being evaluated: the low 30 bits are the
// child to evaluate next.  Any other frame carries one of the codes below there and its record
// index in bits 25..0.  Bit 30 keeps the caller's hit flag.
constexpr uint32_t FK_LIST = 9, FK_INST = 10, FK_MED1 = 11, FK_MED2 = 12, FK_BVH2 = 13;
constexpr uint32_t FW_FLAG = 1u << 30;
constexpr uint32_t FW_MISS_NONE = 1u << 31;  // BVH frame: nothing is left to evaluate if the first child missed
__host__ __device__ inline uint32_t frame_word(uint32_t kind, uint32_t idx) { return idx | (kind << IDX_BITS); }
constexpr uint32_t FRAME_WORDS_LIST = 2, FRAME_WORDS_BVH = 1, FRAME_WORDS_INST = 1, FRAME_WORDS_MEDIUM = 5;

struct DevMaterial {  // 32 B
    uint32_t type, texture;
    float r, g, b, param;  // (r, g, b): the material's own colour, or its texture's when that is one solid colour
    uint32_t solid;        // 1: (r, g, b) is what Texture::get_color would return everywhere (texture.rs:67)
    uint32_t pad1;
};
struct DevTexture {  // 48 B
    uint32_t type, index, depth, pad;
    float c0[3], scale;
    float c1[3], pad2;
};
struct DevImage {
    uint32_t width, height;
    uint64_t offset;
};

// Loads through a constant-address-space pointer are always scalar loads when the address is
// wave-uniform, and are issued where they are used instead of being kept in SGPRs for the
// whole kernel.  On the host the qualifier vanishes; the layout is the same.
#if defined(__HIP_DEVICE_COMPILE__)
#define RR_CONST_AS __attribute__((address_space(4)))
#else
#define RR_CONST_AS
#endif

// What only shading and ray generation read (once per bounce or per sample): kept in device
// memory behind one pointer so that its ~60 dwords do not occupy SGPRs during traversal.
struct DevConst {
    const uint2* sphere_meta;     // (material, owner word)
    const uint2* tri_meta;        // (material, owner word)
    const DevMaterial* materials;
    const DevTexture* textures;
    const float* perlin_vec;      // 256*3 per table
    const uint8_t* perlin_perm;   // 3*256 per table
    const DevImage* images;
    const uint8_t* image_bytes;
    uint32_t light_is_xzrect;
    float light_x0, light_x1, light_z0, light_z1, light_k;
    uint32_t n_lights;            // RR_SEMANTICS_BOOK_ESTIMATOR: members of the ColliderList passed as `lights` (0: the single light above)
    const float4* lights;         // 2 per member: (x0, x1, z0, z1), (k, is_xzrect, -, -)
    rr_camera cam;
    float background[3];
};

struct DevScene {  // kernel argument
    const float4* bvh;          // 2 per node: (min.xyz, left), (max.xyz, right)
    const float4* wide;         // 4 per node, same index: the boxes of the node's two children (see trav_node_wide); null unless the scene qualifies
    const float4* sphere;       // (center.xyz, radius)
    const float4* msphere;      // 3 per record: (c0.xyz, time0), (c1.xyz, time1), (radius, material, owner word, -)
    const float4* rect;         // 2 per record: (a0, a1, b0, b1), (k, axis, material, owner word)
    const float4* tri;          // 3 per record: p0.xyz p1.x | p1.yz p2.xy | p2.z n.xyz
    const float4* cuboid;       // 2 per record: (p0.xyz, first rect), (p1.xyz, 1 if the six rects are the constructor's)
    const uint4* medium;        // (boundary, neg_inv_density bits, material, owner word)
    const float4* inst;         // 2 per record: the instance record (see inst_apply_ray), (child, depth, -, -)
    const float4* chain_rec;    // MAX_CHAIN per instance: the records of its enclosing instances, outermost first, itself last
    const uint2* list;          // (first child, n children)
    const uint32_t* child_refs;
    const RR_CONST_AS uint4* prog;  // scene program, 2 uint4 (one 32-byte op) per entry, terminated by OP_END
    uint32_t n_ops;                 // ops before the terminating OP_ENDs
    float cull_eps, cull_k;         // boxes are grown by cull_eps + cull_k * distance^2 in the culling test
    uint32_t intended;              // RR_SEMANTICS_INTENDED: the labelled alternative semantics (see include/rr_hip.h), 0 by default
    uint32_t book;                  // RR_SEMANTICS_BOOK_ESTIMATOR: specular materials keep their own ray, `lights` may be a list; 0 by default
    uint32_t lean;                  // like no_tri: 1 in the kernels the host launches only for scenes whose media are all bounded by a
                                    // bare sphere (no OP_MED_BEGIN / MID / END), whose cuboids are all the constructor's and whose
                                    // instances nest at most two deep
    uint32_t no_tri;                // 0 from the host; a kernel that is only launched for scenes without triangles sets it to 1 in its
                                    // own copy, and the triangle code (and the registers it needs) is gone from that build
    const RR_CONST_AS DevConst* k;
};

// The kernel arguments arrive by wide scalar loads, eight registers at a time, and the compiler keeps such a group together: when
// it runs out of scalar registers it spills and reloads all eight (a v_readlane each, in the traversal loop) for the one
// pointer a move needs.  A move through an instruction the compiler cannot look into makes each pointer the traversal loop
// uses a value of its own.
template <class T>
__device__ __forceinline__ void own_register_pair(const T*& p) {
    uint64_t from = reinterpret_cast<uint64_t>(p), to;
    asm volatile("s_mov_b64 %0, %1" : "=s"(to) : "s"(from));
    // (named a pointer to global memory again: what comes out of the asm would otherwise be loaded from with flat instructions)
    p = (const T*)reinterpret_cast<const __attribute__((address_space(1))) T*>(to);
}
__device__ __forceinline__ void own_registers(DevScene& sc) {
    own_register_pair(sc.bvh);
    own_register_pair(sc.sphere);
    own_register_pair(sc.rect);
    own_register_pair(sc.tri);
    own_register_pair(sc.cuboid);
}

struct Counts {
    uint32_t rays, box_tests, sphere_tests, rect_tests, tri_tests, inst_visits, samples;
    uint32_t lane_steps, wave_steps;  // steps of the per-lane traversal machine: (lane, step) pairs that moved; steps of the wave (counted on one lane)
#ifdef RR_SECTION_TIMERS  // timing-only build: wave clocks by section of a turn of the path kernel
    unsigned long long sec[16], sec_last;
#endif
};
#ifdef RR_SECTION_TIMERS
__device__ unsigned long long g_sec[16];
#define RR_SEC_MARK(cnt, k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); (cnt).sec[k] += now_ - (cnt).sec_last; (cnt).sec_last = now_; } while (0)
#define RR_SEC_ADD(cnt, k, v) do { (cnt).sec[k] += (v); } while (0)
#else
#define RR_SEC_MARK(cnt, k) do { } while (0)
#define RR_SEC_ADD(cnt, k, v) do { } while (0)
#endif
// Minimum waves per SIMD the register allocator must leave room for in the render kernels.
#ifndef RR_RENDER_WAVES_PER_SIMD
#define RR_RENDER_WAVES_PER_SIMD 4
#endif

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned lo = __shfl_down(static_cast<unsigned>(v), off, 64);
        unsigned hi = __shfl_down(static_cast<unsigned>(v >> 32), off, 64);
        v += (static_cast<unsigned long long>(hi) << 32) | lo;
    }
    return v;
}

// A wave's counts added to the ten 64-bit counters of rr_counters (one atomic per counter and wave).
__device__ __forceinline__ void flush_counts(const Counts& c, unsigned long long* counters) {
    if (!counters) return;
    unsigned long long v[10] = {c.samples, c.rays, c.box_tests, (unsigned long long)c.sphere_tests + c.rect_tests + c.tri_tests, c.inst_visits, c.sphere_tests, c.rect_tests, c.tri_tests,
                                c.lane_steps, c.wave_steps};
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        unsigned long long s = wave_sum(v[i]);
        if ((threadIdx.x & 63u) == 0 && s) atomicAdd(&counters[i], s);
    }
}

// Counting builds of the kernels (template parameter COUNT): COUNT_REFERENCE makes every BVH visit the reference makes,
// in its order, and counts them; COUNT_PRODUCTION counts what the production traversal (visit skipping, culling) does.
constexpr int COUNT_OFF = 0, COUNT_REFERENCE = 1, COUNT_PRODUCTION = 2;

// Wave votes.  RR_VOTE_BUILTIN: on a bool through the compiler's builtin (the lane mask of the comparison itself) instead of
// HIP's __ballot(int), which first makes an integer of the bool and compares it with zero (two vector instructions a vote).
// Measured twice, before and after the kernels stopped spilling registers: -2 % on final_scene both times, so it is off.
#ifdef RR_VOTE_BUILTIN
#define RR_BALLOT(x) __builtin_amdgcn_ballot_w64(static_cast<bool>(x))
#define RR_ANY(x) (__builtin_amdgcn_ballot_w64(static_cast<bool>(x)) != 0ull)
#else
#define RR_BALLOT(x) __ballot(x)
#define RR_ANY(x) __any(x)
#endif

// ------------------------------------------------------------------ math
struct f3 {
    float x, y, z;
};
__device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross3(f3 a, f3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float len2(f3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
__device__ __forceinline__ float len3(f3 a) { return sqrtf(len2(a)); }
__device__ __forceinline__ f3 unit3(f3 a) { return a / len3(a); }
__device__ __forceinline__ f3 neg3(f3 a) { return a * -1.0f; }
__device__ __forceinline__ float comp(f3 a, uint32_t i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
__device__ __forceinline__ f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }

constexpr float kPi = 3.14159265358979323846f;
constexpr float kTau = 6.28318530717958647692f;
constexpr float kF32Max = 3.40282347e+38f;

// ------------------------------------------------------------------- rng
// Philox4x32-10 stream keyed by (seed; pixel, sample, bounce); draw n is word n&3 of
// block n>>2; u32 → f32 on the 2^-23 grid of [0,1).
struct Rng {
    uint32_t k0, k1, c0, c1, c2, n;
    uint32_t b0, b1, b2, b3;
    __device__ __forceinline__ void start(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce) {
        k0 = static_cast<uint32_t>(seed);
        k1 = static_cast<uint32_t>(seed >> 32);
        c0 = pixel;
        c1 = sample;
        c2 = bounce;
        n = 0;
    }
    __device__ __forceinline__ void refill() {
        uint32_t x0 = c0, x1 = c1, x2 = c2, x3 = n >> 2, ka = k0, kb = k1;
#ifndef RR_PHILOX_HOIST_KEYS
        // The key is the seed, a kernel argument, so the ten round keys are the same for every block of a launch, and the
        // compiler would form them once and hold all twenty in scalar registers from one end of the kernel to the other:
        // registers the traversal loops then lack (what does not fit is spilled to lanes of a vector register and read back with
        // a v_readlane per use).  Behind an empty asm the round keys are twenty scalar additions where a block is made.
        asm volatile("" : "+s"(ka), "+s"(kb));
#endif
#ifndef RR_PHILOX_ROUNDS  // (a build with fewer rounds is a timing-only build: it prices the generator)
#define RR_PHILOX_ROUNDS 10
#endif
#pragma unroll
        for (int r = 0; r < RR_PHILOX_ROUNDS; ++r) {
            // (as one 64-bit product per pair the compiler takes v_mad_u64_u32 for both halves: measured -3.5 % on the Cornell box)
            uint32_t hi0 = __umulhi(0xD2511F53u, x0), lo0 = 0xD2511F53u * x0;
            uint32_t hi1 = __umulhi(0xCD9E8D57u, x2), lo1 = 0xCD9E8D57u * x2;
            uint32_t y0 = hi1 ^ x1 ^ ka, y2 = hi0 ^ x3 ^ kb;
            x0 = y0;
            x1 = lo1;
            x2 = y2;
            x3 = lo0;
            ka += 0x9E3779B9u;
            kb += 0xBB67AE85u;
        }
        b0 = x0;
        b1 = x1;
        b2 = x2;
        b3 = x3;
    }
    __device__ __forceinline__ uint32_t next_u32() {
        uint32_t w = n & 3u;
        if (w == 0) refill();
        ++n;
        return w == 0 ? b0 : (w == 1 ? b1 : (w == 2 ? b2 : b3));
    }
    __device__ __forceinline__ float random() { return static_cast<float>(next_u32() >> 9) * (1.0f / 8388608.0f); }
    __device__ __forceinline__ float randrange(float lo, float hi) {
        for (;;) {
            float v = random() * (hi - lo) + lo;
            if (v < hi) return v;
        }
    }
};
constexpr uint32_t CAMERA_BOUNCE = 0xFFFFFFFFu;

// ------------------------------------------------------------- instances
// Ray into the space of instance ii (collider.rs:481 translate, 499-509 rotate).
// An instance record is (kind, v.x, v.y, v.z) for a translation and (kind, sin, cos, -) for a rotation.
__device__ __forceinline__ void inst_apply_ray(float4 a, f3& o, f3& d) {
    const uint32_t kind = __float_as_uint(a.x);
    if (kind == RR_INST_TRANSLATE) {
        o = o - mk3(a.y, a.z, a.w);
        return;
    }
    const float s = a.y, c = a.z;
    if (kind == RR_INST_ROTATE_Y) {  // others = (X, Z)
        f3 no = o, nd = d;
        no.x = c * o.x - s * o.z;
        no.z = s * o.x + c * o.z;
        nd.x = c * d.x - s * d.z;
        nd.z = s * d.x + c * d.z;
        o = no;
        d = nd;
    } else if (kind == RR_INST_ROTATE_X) {  // others = (Y, Z)
        f3 no = o, nd = d;
        no.y = c * o.y - s * o.z;
        no.z = s * o.y + c * o.z;
        nd.y = c * d.y - s * d.z;
        nd.z = s * d.y + c * d.z;
        o = no;
        d = nd;
    } else {  // ROTATE_Z, others = (X, Y)
        f3 no = o, nd = d;
        no.x = c * o.x - s * o.y;
        no.y = s * o.x + c * o.y;
        nd.x = c * d.x - s * d.y;
        nd.y = s * d.x + c * d.y;
        o = no;
        d = nd;
    }
}
__device__ __forceinline__ uint32_t inst_depth(const DevScene& sc, uint32_t ii) {
    return __float_as_uint(sc.inst[2 * ii + 1].y);
}
// "Owner" words name the innermost instance enclosing a primitive together with how many instances
// enclose it, so that the records of the whole chain can be fetched at once.
constexpr uint32_t OWNER_DEPTH_SHIFT = 28;
__host__ __device__ inline uint32_t owner_word(uint32_t inst, uint32_t depth) { return depth == 0 ? 0u : (inst | (depth << OWNER_DEPTH_SHIFT)); }
__host__ __device__ inline uint32_t owner_depth(uint32_t w) { return w >> OWNER_DEPTH_SHIFT; }
__host__ __device__ inline uint32_t owner_inst(uint32_t w) { return w & ((1u << OWNER_DEPTH_SHIFT) - 1u); }
// World ray → ray inside the first `levels` instances of the chain of instance `inst`.
__device__ __forceinline__ void localize(const DevScene& sc, uint32_t inst, uint32_t levels, f3 wo, f3 wd, f3& o, f3& d) {
    o = wo;
    d = wd;
    for (uint32_t k = 0; k < levels; ++k) inst_apply_ray(sc.chain_rec[inst * MAX_CHAIN + k], o, d);
}

// Hit back out of instance ii; `rd` is the ray direction INSIDE the instance
// (collider.rs:484-485 translate, 513-523 rotate: set_face_normal against that ray).
__device__ __forceinline__ void inst_apply_hit(float4 a, f3 rd, f3& p, f3& nrm, bool& front) {
    const uint32_t kind = __float_as_uint(a.x);
    f3 outn = nrm;
    if (kind == RR_INST_TRANSLATE) {
        p = p + mk3(a.y, a.z, a.w);
    } else {
        const float s = a.y, c = a.z;
        f3 np = p;
        if (kind == RR_INST_ROTATE_Y) {
            np.x = c * p.x + s * p.z;
            np.z = -s * p.x + c * p.z;
            outn.x = c * nrm.x + s * nrm.z;
            outn.z = -s * nrm.x + c * nrm.z;
        } else if (kind == RR_INST_ROTATE_X) {
            np.y = c * p.y + s * p.z;
            np.z = -s * p.y + c * p.z;
            outn.y = c * nrm.y + s * nrm.z;
            outn.z = -s * nrm.y + c * nrm.z;
        } else {
            np.x = c * p.x + s * p.y;
            np.y = -s * p.x + c * p.y;
            outn.x = c * nrm.x + s * nrm.y;
            outn.y = -s * nrm.x + c * nrm.y;
        }
        p = np;
    }
    front = dot3(rd, outn) < 0.0f;
    nrm = front ? outn : neg3(outn);
}

// ------------------------------------------------------------ primitives
// Each test returns the accepted t, or a negative-NaN-free "no hit" through `hit`.
__device__ __forceinline__ bool test_sphere(f3 center, float radius, f3 o, f3 d, float tmin, float tmax, float& t) {
    // collider.rs:278-292
    f3 oc = o - center;
    float a = len2(d);
    float half_b = dot3(oc, d);
    float cc = len2(oc) - radius * radius;
    float disc = half_b * half_b - a * cc;
    if (disc < 0.0f) return false;
    float sq = sqrtf(disc);
    float root = (-half_b - sq) / a;
    if (root < tmin || tmax < root) {
        root = (-half_b + sq) / a;
        if (root < tmin || tmax < root) return false;
    }
    t = root;
    return true;
}

__device__ __forceinline__ bool test_rect(float4 q0, float k, uint32_t axis, f3 o, f3 d, float tmin, float tmax, float& t,
                                          float& ca, float& cb) {
    // collider.rs:344-353 / 367-376 / 405-414; the plane is per-lane data here, so components are
    // picked with selects rather than branches
    const bool xy = axis == RR_RECT_XY, yz = axis == RR_RECT_YZ;
    const float ok = xy ? o.z : (yz ? o.x : o.y), dk = xy ? d.z : (yz ? d.x : d.y);
    const float oa = yz ? o.y : o.x, da = yz ? d.y : d.x;
    const float ob = xy ? o.y : o.z, db = xy ? d.y : d.z;
    float tt = (k - ok) / dk;
    if (tt < tmin || tt > tmax) return false;
    float a = oa + tt * da;
    float b = ob + tt * db;
    if (a < q0.x || a > q0.y || b < q0.z || b > q0.w) return false;
    t = tt;
    ca = a;
    cb = b;
    return true;
}

// Rect test with the plane fixed at compile time (collider.rs:344-353 / 367-376 / 405-414).
template <uint32_t AXIS>
__device__ __forceinline__ bool test_rect_axis(float a0, float a1, float b0, float b1, float k, f3 o, f3 d, float tmin, float tmax,
                                               float& t) {
    const float ok = AXIS == RR_RECT_XY ? o.z : (AXIS == RR_RECT_XZ ? o.y : o.x);
    const float dk = AXIS == RR_RECT_XY ? d.z : (AXIS == RR_RECT_XZ ? d.y : d.x);
    const float oa = AXIS == RR_RECT_YZ ? o.y : o.x, da = AXIS == RR_RECT_YZ ? d.y : d.x;
    const float ob = AXIS == RR_RECT_XY ? o.y : o.z, db = AXIS == RR_RECT_XY ? d.y : d.z;
    // One early exit after the division (measured 7 % faster on the Cornell box than none: most
    // lanes of most waves fail the range test), then bitwise | so the rest is straight-line code.
    // The reference's two `if (.. || ..) return None` let NaN through (every comparison is false),
    // and so does this.
    const float tt = (k - ok) / dk;
    if (tt < tmin || tt > tmax) return false;
    const float a = oa + tt * da;
    const float b = ob + tt * db;
    const bool out_of_range = (tt < tmin) | (tt > tmax);
    const bool outside = (a < a0) | (a > a1) | (b < b0) | (b > b1);
    t = tt;
    return !(out_of_range | outside);
}

__device__ __forceinline__ bool test_triangle(float4 q0, float4 q1, float4 q2, f3 o, f3 d, float tmin, float tmax, bool ranged, float& t) {
    // collider.rs:627-661; the t range is not tested (`ranged`: it is, under the labelled intended semantics only)
    const f3 p0 = mk3(q0.x, q0.y, q0.z), p1 = mk3(q0.w, q1.x, q1.y), p2 = mk3(q1.z, q1.w, q2.x), n = mk3(q2.y, q2.z, q2.w);
    float ndd = dot3(n, d);
    if (fabsf(ndd) < 0.000001f) return false;
    float dd = -dot3(n, p0);
    float dist = -(dot3(n, o) + dd) / ndd;
    if (dist < 0.0f) return false;
    if (ranged && (dist < tmin || dist > tmax)) return false;
    f3 hp = o + d * dist;
    if (dot3(n, cross3(p1 - p0, hp - p0)) < 0.0f) return false;
    if (dot3(n, cross3(p2 - p1, hp - p1)) < 0.0f) return false;
    if (dot3(n, cross3(p0 - p2, hp - p2)) < 0.0f) return false;
    t = dist;
    return true;
}

__device__ __forceinline__ bool test_box(float4 lo, float4 hi, f3 o, f3 d, float tmin, float tmax) {
    // aabb.rs:37-52: only axes with a negative reciprocal direction clip the interval
    {
        float inv = 1.0f / d.x;
        float t0 = (lo.x - o.x) * inv, t1 = (hi.x - o.x) * inv;
        if (inv < 0.0f) {
            tmin = t1 > tmin ? t1 : tmin;
            tmax = t0 < tmax ? t0 : tmax;
        }
        if (tmax <= tmin) return false;
    }
    {
        float inv = 1.0f / d.y;
        float t0 = (lo.y - o.y) * inv, t1 = (hi.y - o.y) * inv;
        if (inv < 0.0f) {
            tmin = t1 > tmin ? t1 : tmin;
            tmax = t0 < tmax ? t0 : tmax;
        }
        if (tmax <= tmin) return false;
    }
    {
        float inv = 1.0f / d.z;
        float t0 = (lo.z - o.z) * inv, t1 = (hi.z - o.z) * inv;
        if (inv < 0.0f) {
            tmin = t1 > tmin ? t1 : tmin;
            tmax = t0 < tmax ? t0 : tmax;
        }
        if (tmax <= tmin) return false;
    }
    return true;
}

// test_box with the three reciprocals of the direction already formed (same values, same tests).
__device__ __forceinline__ bool test_box_inv(float4 lo, float4 hi, f3 o, f3 inv, float tmin, float tmax) {
    if (inv.x < 0.0f) {
        const float t0 = (lo.x - o.x) * inv.x, t1 = (hi.x - o.x) * inv.x;
        tmin = t1 > tmin ? t1 : tmin;
        tmax = t0 < tmax ? t0 : tmax;
    }
    if (tmax <= tmin) return false;
    if (inv.y < 0.0f) {
        const float t0 = (lo.y - o.y) * inv.y, t1 = (hi.y - o.y) * inv.y;
        tmin = t1 > tmin ? t1 : tmin;
        tmax = t0 < tmax ? t0 : tmax;
    }
    if (tmax <= tmin) return false;
    if (inv.z < 0.0f) {
        const float t0 = (lo.z - o.z) * inv.z, t1 = (hi.z - o.z) * inv.z;
        tmin = t1 > tmin ? t1 : tmin;
        tmax = t0 < tmax ? t0 : tmax;
    }
    return !(tmax <= tmin);
}

__device__ __forceinline__ f3 msphere_center(float4 a, float4 b, float time) {  // collider.rs:269-271
    f3 c0 = mk3(a.x, a.y, a.z), c1 = mk3(b.x, b.y, b.z);
    return c0 + (c1 - c0) * ((time - a.w) / (b.w - a.w));
}

// ------------------------------------------------------- traversal machine
// Emulates the reference's recursive intersect (collider.rs:682-697) with an explicit
// per-lane stack.  The recursion's value — not simply the closest hit — is what is
// reproduced: BVH children are visited in the reference's order, a first-child miss
// re-tests `left` (collider.rs:555-561), span-1 leaves are tested twice, triangles
// accept any t >= 0 and every accepted hit replaces the previous one.
//
// State: (tmin, tmax, cur, hitf).  `cur` is the primitive (or medium) of the last accepted
// hit and tmax its t; a call that accepts nothing leaves both alone, so "the call returned
// None" is "hitf is still clear".  Calls whose Some/None the caller branches on (the first
// child of a BVH node, a medium's boundary probes) save the caller's hitf in their frame and
// start with it clear; every other recursive call is a tail call and needs no frame.
// Only (t, primitive) travel up the stack: point, normal, uv and the instance
// post-transforms are recomputed once from them in reconstruct_hit().
//
// stk addresses this lane's stack: word w lives at stk[w * stride].
struct Trav {
    f3 o, d, inv;  // ray in the current space; inv = 1/d, which aabb.rs:39 forms anew in every box test
    float tmin, tmax;
    uint32_t cur, node, sp;
    bool hitf;  // a hit was accepted since the innermost open first-child call or medium probe began
    bool cull;  // subtrees whose box the ray misses may be left out (see BVH_CLASS_CULL_*)
    float margin;  // (visits that set it: see node_box_pass) what the culling test grows every box of the subtree by

    __device__ __forceinline__ void set_ray(f3 o_, f3 d_) {
        o = o_;
        d = d_;
        inv = mk3(1.0f / d_.x, 1.0f / d_.y, 1.0f / d_.z);
        // The intersectors accept a hit whose t is NaN (every comparison with NaN is false): a sphere when
        // |d|^2 is 0 or not finite, a rect when its direction component and plane distance are both 0.
        // Such a hit has nothing to do with geometry, so rays that can produce one are never culled.
        const float a = len2(d_);
        cull = d_.x != 0.0f && d_.y != 0.0f && d_.z != 0.0f && a >= 1e-10f && a <= 1e10f;
    }
    __device__ __forceinline__ void begin(uint32_t start, f3 o_, f3 d_, float tmin_, float tmax_, uint32_t cur_) {
        set_ray(o_, d_);
        tmin = tmin_;
        tmax = tmax_;
        cur = cur_;
        node = start;
        sp = 0;
        hitf = false;
        margin = 0.0f;
    }
    __device__ __forceinline__ bool finished() const { return node == NONE && sp == 0; }
};

// What the machine needs only now and then comes through two small providers, so that a kernel
// can keep it in registers or fetch it on demand:
//   World: ray(wo, wd) — the ray in world space; time() — the ray's time
//   Draw:  random()    — the next f32 of the path's random stream
//
// The machine advances in three kinds of moves, each a function below:
//   trav_node   m.node is a BVH node: box test, frame push, first child
//   trav_leaf   m.node is anything else: primitive tests, or the entry of a list, instance or medium
//   trav_pop    m.node is NONE: return to the innermost caller, which names the next node or has none left
// The machine is finished when m.node is NONE and the stack is empty.  A kernel is
// free to schedule the moves of its 64 lanes as it likes (all lanes at nodes first, leaves when
// enough lanes have one) as long as each lane's own sequence is node/leaf followed by pop.
// All three words of a record are in registers from here on: without this the compiler moves the loads of the words only
// some kinds of record use into the branches that use them, behind the first word's arrival, and a move waits for memory twice.
__device__ __forceinline__ void records_arrived(const float4& q0, const float4& q1, const float4& q2) {
    asm volatile("" ::"v"(q0.x), "v"(q0.y), "v"(q0.z), "v"(q0.w), "v"(q1.x), "v"(q1.y), "v"(q1.z), "v"(q1.w), "v"(q2.x), "v"(q2.y), "v"(q2.z), "v"(q2.w));
}
__device__ __forceinline__ bool trav_at_node(const Trav& m) { return m.node != NONE && dtag(m.node) == RR_TAG_BVH; }
__device__ __forceinline__ bool trav_at_leaf(const Trav& m) { return m.node != NONE && dtag(m.node) != RR_TAG_BVH; }

// The box test of a BVH node of class `cls` (not BVH_CLASS_NO_BOX) for the ray and range of m: the reference's test, and in
// the culling classes the three-sided test on the grown box as well.
// The margin of the culling test for a ray from `o` and a box: it grows with the squared distance from the ray origin to the
// farthest point of the box, as the rounding errors of the intersectors below do.
__device__ __forceinline__ float cull_margin(const DevScene& sc, f3 o, f3 bmin, f3 bmax) {
    const float ax = fmaxf(fabsf(bmin.x - o.x), fabsf(bmax.x - o.x)), ay = fmaxf(fabsf(bmin.y - o.y), fabsf(bmax.y - o.y)),
                az = fmaxf(fabsf(bmin.z - o.z), fabsf(bmax.z - o.z));
    return sc.cull_eps + sc.cull_k * (ax * ax + ay * ay + az * az);
}

// SUBTREE_MARGIN: the margin is m.margin, formed once per visit from the box of the subtree's root, which holds the box of
// every node below (rr_scene_upload grants the culling classes to such nodes only): a larger margin than the node's own, so
// the test lets a few more rays in, which find nothing, and saves fourteen instructions at every node.
template <bool SUBTREE_MARGIN = false>
__device__ __forceinline__ bool node_box_pass(const DevScene& sc, const Trav& m, f3 bmin, f3 bmax, uint32_t cls) {
    // Slab parameters of the three axes, entry and exit ordered by the sign of 1/direction.
    const float tx0 = (bmin.x - m.o.x) * m.inv.x, tx1 = (bmax.x - m.o.x) * m.inv.x;
    const float ty0 = (bmin.y - m.o.y) * m.inv.y, ty1 = (bmax.y - m.o.y) * m.inv.y;
    const float tz0 = (bmin.z - m.o.z) * m.inv.z, tz1 = (bmax.z - m.o.z) * m.inv.z;
    const bool nx = m.inv.x < 0.0f, ny = m.inv.y < 0.0f, nz = m.inv.z < 0.0f;
    const float ex = nx ? tx1 : tx0, fx = nx ? tx0 : tx1;
    const float ey = ny ? ty1 : ty0, fy = ny ? ty0 : ty1;
    const float ez = nz ? tz1 : tz0, fz = nz ? tz0 : tz1;
    // aabb.rs:37-52: only axes with a negative 1/direction clip [t_min, t_max], by
    // `t1 > t_min ? t1 : t_min` and `t0 < t_max ? t0 : t_max` (a NaN never replaces a bound:
    // fmaxf/fminf keep the other operand too), and the node is rejected once t_max <= t_min.
    // Bounds only tighten, so testing once after the third axis rejects the same rays.
    // Under the labelled intended semantics every axis clips, and the box test itself is then the cull.  (The flags
    // below do not change while a ray descends: the compiler forms them once per visit, where a branch on the
    // semantics would cost scalar instructions at every node.)
    const bool all_axes = sc.intended != 0u;
    const bool cx = nx || all_axes, cy = ny || all_axes, cz = nz || all_axes;
    const float lo = fmaxf(fmaxf(m.tmin, cx ? ex : m.tmin), fmaxf(cy ? ey : m.tmin, cz ? ez : m.tmin));
    const float hi = fminf(fminf(m.tmax, cx ? fx : m.tmax), fminf(cy ? fy : m.tmax, cz ? fz : m.tmax));
    // (A bound that is NaN on entry — the t of a degenerate hit — stays NaN in the reference and
    // fails no comparison, where fmaxf/fminf would drop it.)
    bool pass = !(hi <= lo) || __builtin_isunordered(m.tmin, m.tmax);
    if (cls >= BVH_CLASS_CULL_RANGE && m.cull && !all_axes) {
        // All three slabs of the box grown by the margin, against the segment no accepted hit can lie
        // outside of.
        const float g = SUBTREE_MARGIN ? m.margin : cull_margin(sc, m.o, bmin, bmax);
        const float gx = g * fabsf(m.inv.x), gy = g * fabsf(m.inv.y), gz = g * fabsf(m.inv.z);
        const float seg0 = cls == BVH_CLASS_CULL_ANY_T ? 0.0f : m.tmin, seg1 = cls == BVH_CLASS_CULL_ANY_T ? kF32Max : m.tmax;
        const float enter = fmaxf(fmaxf(seg0, ex - gx), fmaxf(ey - gy, ez - gz));
        const float leave = fminf(fminf(seg1, fx + gx), fminf(fy + gy, fz + gz));
        pass = pass && !(leave < enter);
    }
    return pass;
}

// What follows a passed box test (collider.rs:546-580): the first child by the sign of direction[axis], and the frame that
// names what comes after it.  `left` and `right` are the node's words (child ref, and axis or class in the spare bits).
__device__ __forceinline__ void node_descend(const DevScene& sc, Trav& m, uint32_t idx, uint32_t left, uint32_t right, uint32_t cls, uint32_t* stk,
                                             uint32_t stride) {
    const uint32_t l = left & REF_MASK, r = right & REF_MASK;
    const bool dirpos = comp(m.d, dextra(left)) >= 0.0f;
    // In the culling classes evaluating the same child twice over the same or a smaller
    // range cannot change the outcome.  The reference-order counting build evaluates everything.
    // Under the labelled intended semantics (src/bvh.rs:19-44) the other child follows the first after a hit and after a
    // miss alike, and a span-1 node's second visit is only left out where it cannot draw random numbers (its classes say so).
    const bool pure = cls >= BVH_CLASS_CULL_RANGE;
    // what follows `left` when it went first: `right` (after a hit only: the re-test of left after a miss misses again), or the
    // node's frame, which decides after left came back; when `right` went first, left follows after a hit and after a miss alike
    const uint32_t after_left = sc.intended != 0u ? r : (pure ? (r | FW_MISS_NONE) : frame_word(FK_BVH2, idx));
    const uint32_t fw = dirpos ? after_left : l;
    m.node = dirpos ? l : r;
    // (straight-line code with one predicated store: as nested ifs this was three levels of scalar branches per node)
    if (!(pure && l == r)) {  // a span-1 node's second visit repeats the first: no frame
        stk[m.sp * stride] = fw | (m.hitf ? FW_FLAG : 0u);
        m.sp += 1;
        m.hitf = false;
    }
}

// The class a build of the kernels treats a node of stored class `stored` as: the reference-order counting build makes every
// visit the reference makes.
template <int COUNT>
__device__ __forceinline__ uint32_t node_class(uint32_t stored) {
    return COUNT == COUNT_REFERENCE ? (stored != BVH_CLASS_NO_BOX ? BVH_CLASS_REFERENCE : BVH_CLASS_NO_BOX) : stored;
}

template <int COUNT, bool SUBTREE_MARGIN = false>
__device__ __forceinline__ void trav_node(const DevScene& sc, Trav& m, uint32_t* stk, uint32_t stride, Counts& cnt) {
    // ------------------------------------------------ BVH node
    if (m.node != NONE && dtag(m.node) == RR_TAG_BVH) {
        const uint32_t idx = didx(m.node);
        const float4 q0 = sc.bvh[2 * idx], q1 = sc.bvh[2 * idx + 1];
        m.node = NONE;
        const uint32_t left = __float_as_uint(q0.w), right = __float_as_uint(q1.w);
        const uint32_t cls = node_class<COUNT>(dextra(right));
        // The test is evaluated for every node, boxless ones included (their result is not used): were it conditional on
        // the class, the compiler would fetch the class words first and the boxes after them, two round trips per move.
        if (COUNT && cls != BVH_CLASS_NO_BOX) cnt.box_tests++;
        const bool pass = node_box_pass<SUBTREE_MARGIN>(sc, m, mk3(q0.x, q0.y, q0.z), mk3(q1.x, q1.y, q1.z), cls) || cls == BVH_CLASS_NO_BOX;
        if (pass) node_descend(sc, m, idx, left, right, cls, stk, stride);
    }
}

// ------------------------------------------------------------------------------------------------ the wide node move
// trav_node tests a node's own box when the ray arrives at it: every child whose box the ray misses has cost a dependent
// fetch of its record and a step of the machine by then.  The wide move tests the boxes of BOTH children from one 64-byte
// record of the node itself (DevScene::wide: (L.min, L link | axis << 30), (L.max, L class), (R.min, R link), (R.max, R class);
// a child that is a primitive, not a BVH node, has class 0 and no box to test) and goes on only to children that pass, which
// arrive "tested" (NODE_TESTED in Trav::node).  For subtrees whose BVH nodes are all of the culling classes (pure: every
// intersect call below is a function of (ray, t range) alone), under the reference's semantics.  What collider.rs:546-580 does
// at such a node, with `first` the child the sign of direction[axis] puts first:
//   * first's box test is made with the very (t_min, t_max) its visit would see, so its outcome here is the visit's;
//   * if first fails and it is `left` (direction[axis] >= 0) the node returns None (the re-test of `left` misses again and
//     `right` is never reached); if it is `right`, `left` follows with the same range: its test here is exact as well;
//   * if first passes, the other child's visit comes after first's subtree.  When nothing was hit in between the range is
//     what it is now, and the outcome of its test here (kept in the frame) stands.  After a hit the range has changed: the
//     child then arrives untested and its own box is fetched from DevScene::bvh and tested as trav_node would (with
//     direction[axis] >= 0 this is the only way `right` is ever visited);
//   * a second child the ray cannot reach is not even kept: the culling test of a class with triangles below
//     (BVH_CLASS_CULL_ANY_T) does not depend on the range; the tests of a child without triangles below do, but t_max only
//     shrinks while the first child is walked, so a test that fails now fails later — unless the first child has triangles
//     below (they are accepted at any t and can move t_max out again), or a hit with t = NaN is possible, which the rays
//     that are never culled (Trav::cull) stand for: then nothing is pruned on those tests.
// Frames are the one-word BVH frames of trav_node (FW_MISS_NONE: nothing follows a miss of the first child); what differs is
// the return: a child that follows a miss arrives tested (trav_pop<..., WIDE = true>).
// Measured (1.1 M-primitive mesh, wavefront pipeline): 2 034 -> 1 346 machine steps per ray with the same box and primitive
// tests, 43.4 -> 48.3 Msamples/s.  In the path kernel's visits of subtrees (cache-resident scenes, bound by instruction issue,
// not by the latency of dependent fetches) the same move took final_scene from 6.4 to 3.9 machine steps per ray and was 3-7 %
// slower (final_scene 2 500 -> 2 333, Suzanne 4 808 -> 4 392, random spheres 4 630 -> 4 142 Msamples/s): a step that tests two
// boxes costs about twice the instructions, 12 more registers were spilled at 96, and the second box is tested where the
// one-box move never comes to it (direction[axis] >= 0 and `left` missed); that kernel keeps the one-box move.
constexpr uint32_t NODE_TESTED = 1u << 30;  // in Trav::node, BVH refs only: the node's own box test has been made and passed

// The two halves of node_box_pass for a box of class `cls` (2 or 3): the reference's test (aabb.rs:37-52) and the culling
// test on the box grown by m.margin.  Reference semantics only.
__device__ __forceinline__ void box_pass_parts(const Trav& m, f3 bmin, f3 bmax, uint32_t cls, float margin, bool& ref_pass, bool& cull_pass) {
    const float tx0 = (bmin.x - m.o.x) * m.inv.x, tx1 = (bmax.x - m.o.x) * m.inv.x;
    const float ty0 = (bmin.y - m.o.y) * m.inv.y, ty1 = (bmax.y - m.o.y) * m.inv.y;
    const float tz0 = (bmin.z - m.o.z) * m.inv.z, tz1 = (bmax.z - m.o.z) * m.inv.z;
    const bool nx = m.inv.x < 0.0f, ny = m.inv.y < 0.0f, nz = m.inv.z < 0.0f;
    const float ex = nx ? tx1 : tx0, fx = nx ? tx0 : tx1;
    const float ey = ny ? ty1 : ty0, fy = ny ? ty0 : ty1;
    const float ez = nz ? tz1 : tz0, fz = nz ? tz0 : tz1;
    const float lo = fmaxf(fmaxf(m.tmin, nx ? ex : m.tmin), fmaxf(ny ? ey : m.tmin, nz ? ez : m.tmin));
    const float hi = fminf(fminf(m.tmax, nx ? fx : m.tmax), fminf(ny ? fy : m.tmax, nz ? fz : m.tmax));
    ref_pass = !(hi <= lo) || __builtin_isunordered(m.tmin, m.tmax);
    const float g = margin;
    const float gx = g * fabsf(m.inv.x), gy = g * fabsf(m.inv.y), gz = g * fabsf(m.inv.z);
    const float seg0 = cls == BVH_CLASS_CULL_ANY_T ? 0.0f : m.tmin, seg1 = cls == BVH_CLASS_CULL_ANY_T ? kF32Max : m.tmax;
    const float enter = fmaxf(fmaxf(seg0, ex - gx), fmaxf(ey - gy, ez - gz));
    const float leave = fminf(fminf(seg1, fx + gx), fminf(fy + gy, fz + gz));
    cull_pass = !m.cull || !(leave < enter);
}

// Every box is grown by a margin of its own (cull_margin of that box), as trav_node<COUNT, false> does: in a scene of a million
// primitives a margin formed once from a subtree's root box (it grows with the squared distance to the farthest corner) lets
// rays into many boxes they only pass by (measured on the 1.1 M-primitive mesh: +27 % box tests, +37 % primitive tests).
template <int COUNT>
__device__ __forceinline__ void trav_node_wide(const DevScene& sc, Trav& m, uint32_t* stk, uint32_t stride, Counts& cnt) {
    if (m.node != NONE && dtag(m.node) == RR_TAG_BVH) {
        const uint32_t idx = didx(m.node);
        const bool tested = (m.node & NODE_TESTED) != 0u;
        const float4 w0 = sc.wide[4 * idx], w1 = sc.wide[4 * idx + 1], w2 = sc.wide[4 * idx + 2], w3 = sc.wide[4 * idx + 3];
        m.node = NONE;
        bool pass = true;
        if (!tested) {  // the ray arrives with a range that is not the one its parent's move saw (or from outside the subtree)
            const float4 q0 = sc.bvh[2 * idx], q1 = sc.bvh[2 * idx + 1];
            const f3 bmin = mk3(q0.x, q0.y, q0.z), bmax = mk3(q1.x, q1.y, q1.z);
            if (COUNT) cnt.box_tests++;
            bool rp, cp;
            box_pass_parts(m, bmin, bmax, dextra(__float_as_uint(q1.w)), cull_margin(sc, m.o, bmin, bmax), rp, cp);
            pass = rp && cp;
        }
        if (pass) {
            const uint32_t lw = __float_as_uint(w0.w), ll = lw & REF_MASK, rl = __float_as_uint(w2.w) & REF_MASK;
            const uint32_t lc = __float_as_uint(w1.w), rc = __float_as_uint(w3.w);
            bool l_ref, l_cull, r_ref, r_cull;
            const f3 lmin = mk3(w0.x, w0.y, w0.z), lmax = mk3(w1.x, w1.y, w1.z), rmin = mk3(w2.x, w2.y, w2.z), rmax = mk3(w3.x, w3.y, w3.z);
            box_pass_parts(m, lmin, lmax, lc, cull_margin(sc, m.o, lmin, lmax), l_ref, l_cull);
            box_pass_parts(m, rmin, rmax, rc, cull_margin(sc, m.o, rmin, rmax), r_ref, r_cull);
            if (COUNT) cnt.box_tests += (lc != 0u ? 1u : 0u) + (rc != 0u ? 1u : 0u);
            const bool l_pass = lc == 0u || (l_ref && l_cull), r_pass = rc == 0u || (r_ref && r_cull);
            // A child that cannot be reached in this visit of the node, whatever is hit before its turn.  With triangles below it
            // (class CULL_ANY_T) its culling test does not depend on the range.  Without (class CULL_RANGE) both of its tests
            // depend on t_max, which only shrinks while the other child is walked — unless that one has triangles below (or is
            // one): a triangle is accepted at any t (collider.rs:627) and can move t_max out again — and unless t_max can
            // become NaN (the rays that are never culled).
            const bool l_grows = lc == BVH_CLASS_CULL_ANY_T || dtag(ll) == RR_TAG_TRIANGLE, r_grows = rc == BVH_CLASS_CULL_ANY_T || dtag(rl) == RR_TAG_TRIANGLE;
            const bool l_prune = lc != 0u && (lc == BVH_CLASS_CULL_ANY_T ? !l_cull : (!r_grows && m.cull && !(l_ref && l_cull)));
            const bool r_prune = rc != 0u && (rc == BVH_CLASS_CULL_ANY_T ? !r_cull : (!l_grows && m.cull && !(r_ref && r_cull)));
            const bool dirpos = comp(m.d, dextra(lw)) >= 0.0f;
            const bool pass1 = dirpos ? l_pass : r_pass, pass2 = dirpos ? r_pass : l_pass, prune2 = dirpos ? r_prune : l_prune;
            const uint32_t link1 = dirpos ? ll : rl, link2 = dirpos ? rl : ll;
            const uint32_t t1 = dtag(link1) == RR_TAG_BVH ? NODE_TESTED : 0u, t2 = dtag(link2) == RR_TAG_BVH ? NODE_TESTED : 0u;
            if (!pass1) {
                m.node = (dirpos || !pass2) ? NONE : (link2 | t2);
            } else {
                m.node = link1 | t1;
                if (!(prune2 || link1 == link2)) {  // (a span-1 node's second visit repeats the first: no frame)
                    stk[m.sp * stride] = link2 | ((dirpos || !pass2) ? FW_MISS_NONE : 0u) | (m.hitf ? FW_FLAG : 0u);
                    m.sp += 1;
                    m.hitf = false;
                }
            }
        }
    }
}

// The leaf and return moves for subtrees that hold nothing but BVH nodes over spheres, rects, triangles and cuboids ("simple"
// subtrees): no lists, instances, media or moving spheres below, so neither the world ray, the ray's time nor random numbers
// are needed, and every frame on the stack is a BVH node's.
template <int COUNT>
__device__ __forceinline__ void trav_leaf_simple(const DevScene& sc, Trav& m, Counts& cnt) {
    if (m.node != NONE && dtag(m.node) != RR_TAG_BVH) {
        const uint32_t node = m.node;
        const uint32_t tag = dtag(node), idx = didx(node);
        const float4* rp = sc.tri;  // one address, three 16-byte loads and one wait, whatever the record
        uint32_t roff = 3 * idx;
        if (tag == RR_TAG_SPHERE) rp = sc.sphere, roff = idx;
        if (tag == RR_TAG_RECT) rp = sc.rect, roff = 2 * idx;
        if (tag == RR_TAG_CUBOID) rp = sc.cuboid, roff = 2 * idx;
        const float4 q0 = rp[roff], q1 = rp[roff + 1], q2 = rp[roff + 2];
        records_arrived(q0, q1, q2);
        m.node = NONE;
        if (tag == RR_TAG_SPHERE) {
            if (COUNT) cnt.sphere_tests++;
            float t;
            if (test_sphere(mk3(q0.x, q0.y, q0.z), q0.w, m.o, m.d, m.tmin, m.tmax, t)) m.tmax = t, m.cur = node, m.hitf = true;
        } else if (tag == RR_TAG_TRIANGLE && !sc.no_tri) {
            if (COUNT) cnt.tri_tests++;
            float t;
            if (test_triangle(q0, q1, q2, m.o, m.d, m.tmin, m.tmax, sc.intended != 0u, t)) m.tmax = t, m.cur = node, m.hitf = true;
        } else if (tag == RR_TAG_RECT) {
            if (COUNT) cnt.rect_tests++;
            float t, ca, cb;
            if (test_rect(q0, q1.x, __float_as_uint(q1.y), m.o, m.d, m.tmin, m.tmax, t, ca, cb)) m.tmax = t, m.cur = node, m.hitf = true;
        } else {  // RR_TAG_CUBOID: six rects as a linear closest-hit list (collider.rs:428-430 -> 751-772)
            const uint32_t first = __float_as_uint(q0.w);
            if (COUNT) cnt.rect_tests += 6;
            if (__float_as_uint(q1.w) != 0u || sc.lean) {  // the sides are the constructor's (collider.rs:86-101): all six follow from the corners
                float t;
                uint32_t side = NONE;
                if (test_rect_axis<RR_RECT_XY>(q0.x, q1.x, q0.y, q1.y, q1.z, m.o, m.d, m.tmin, m.tmax, t)) m.tmax = t, side = 0;
                if (test_rect_axis<RR_RECT_XY>(q0.x, q1.x, q0.y, q1.y, q0.z, m.o, m.d, m.tmin, m.tmax, t)) m.tmax = t, side = 1;
                if (test_rect_axis<RR_RECT_XZ>(q0.x, q1.x, q0.z, q1.z, q1.y, m.o, m.d, m.tmin, m.tmax, t)) m.tmax = t, side = 2;
                if (test_rect_axis<RR_RECT_XZ>(q0.x, q1.x, q0.z, q1.z, q0.y, m.o, m.d, m.tmin, m.tmax, t)) m.tmax = t, side = 3;
                if (test_rect_axis<RR_RECT_YZ>(q0.y, q1.y, q0.z, q1.z, q1.x, m.o, m.d, m.tmin, m.tmax, t)) m.tmax = t, side = 4;
                if (test_rect_axis<RR_RECT_YZ>(q0.y, q1.y, q0.z, q1.z, q0.x, m.o, m.d, m.tmin, m.tmax, t)) m.tmax = t, side = 5;
                if (side != NONE) m.cur = dref(RR_TAG_RECT, first + side), m.hitf = true;
            } else {
                for (uint32_t i = 0; i < 6; ++i) {
                    const float4 r0 = sc.rect[2 * (first + i)], r1 = sc.rect[2 * (first + i) + 1];
                    float t, ca, cb;
                    if (test_rect(r0, r1.x, __float_as_uint(r1.y), m.o, m.d, m.tmin, m.tmax, t, ca, cb))
                        m.tmax = t, m.cur = dref(RR_TAG_RECT, first + i), m.hitf = true;
                }
            }
        }
    }
}

__device__ __forceinline__ void trav_pop_simple(const DevScene& sc, Trav& m, uint32_t* stk, uint32_t stride) {
    if (m.node == NONE && m.sp != 0) {  // one return per move, as in trav_pop
        const uint32_t fw = stk[(m.sp - 1) * stride];
        m.sp -= 1;
        const bool first_hit = m.hitf;
        m.hitf = first_hit || (fw & FW_FLAG);
        if (dtag(fw) == FK_BVH2) {  // left went first: right with t_max = the hit's t, or left once more after a miss
            const uint32_t idx = didx(fw);
            m.node = __float_as_uint(first_hit ? sc.bvh[2 * idx + 1].w : sc.bvh[2 * idx].w) & REF_MASK;
        } else {  // collider.rs:546-580 after the first child came back: the frame names what comes next
            m.node = (!first_hit && (fw & FW_MISS_NONE)) ? NONE : (fw & REF_MASK);
        }
    }
}

// PLAIN (trav_leaf, trav_pop): a build for scenes without moving spheres, instances and media (rr_scene_upload knows), which
// leaves their moves out: fewer instructions in every move and, in k_wf_traverse, a fifth fewer scalar registers moved to
// and from vector lanes around them (1.1 M-primitive mesh: +5.4 %).
template <int COUNT, class World, bool PLAIN = false>
__device__ __forceinline__ void trav_leaf(const DevScene& sc, Trav& m, const World& world, uint32_t* stk, uint32_t stride, Counts& cnt) {
    // ------------------------------------------------ primitive, list, instance or medium
    if (m.node != NONE && dtag(m.node) != RR_TAG_BVH) {
        const uint32_t node = m.node;
        const uint32_t tag = dtag(node), idx = didx(node);
        // One address, three 16-byte loads and one wait for whatever kind of record each lane
        // is at (every record array is padded by two words, so reading past a short record
        // is harmless).
        const float4* rp = sc.tri;
        uint32_t roff = 3 * idx;
        if (tag == RR_TAG_SPHERE) rp = sc.sphere, roff = idx;
        if (tag == RR_TAG_RECT) rp = sc.rect, roff = 2 * idx;
        if (tag == RR_TAG_CUBOID) rp = sc.cuboid, roff = 2 * idx;
        if (!PLAIN && tag == RR_TAG_MOVING_SPHERE) rp = sc.msphere;
        if (tag == RR_TAG_LIST || (!PLAIN && (tag == RR_TAG_INSTANCE || tag == RR_TAG_MEDIUM))) roff = 0;
        const float4 q0 = rp[roff], q1 = rp[roff + 1], q2 = rp[roff + 2];
        records_arrived(q0, q1, q2);
        m.node = NONE;
        if (tag == RR_TAG_SPHERE) {
            if (COUNT) cnt.sphere_tests++;
            float t;
            if (test_sphere(mk3(q0.x, q0.y, q0.z), q0.w, m.o, m.d, m.tmin, m.tmax, t)) {
                m.tmax = t;
                m.cur = node;
                m.hitf = true;
            }
        } else if (tag == RR_TAG_TRIANGLE && !sc.no_tri) {
            if (COUNT) cnt.tri_tests++;
            float t;
            if (test_triangle(q0, q1, q2, m.o, m.d, m.tmin, m.tmax, sc.intended != 0u, t)) {
                m.tmax = t;
                m.cur = node;
                m.hitf = true;
            }
        } else if (tag == RR_TAG_RECT) {
            if (COUNT) cnt.rect_tests++;
            float t, ca, cb;
            if (test_rect(q0, q1.x, __float_as_uint(q1.y), m.o, m.d, m.tmin, m.tmax, t, ca, cb)) {
                m.tmax = t;
                m.cur = node;
                m.hitf = true;
            }
        } else if (tag == RR_TAG_CUBOID) {
            // six rects as a linear closest-hit list (collider.rs:428-430 → 751-772)
            const uint32_t first = __float_as_uint(q0.w);
            if (COUNT) cnt.rect_tests += 6;
            if (__float_as_uint(q1.w) != 0u) {
                // the sides are the constructor's (collider.rs:419-426): all six follow from the corners
                float t;
                uint32_t side = NONE;
                if (test_rect_axis<RR_RECT_XY>(q0.x, q1.x, q0.y, q1.y, q1.z, m.o, m.d, m.tmin, m.tmax, t)) m.tmax = t, side = 0;
                if (test_rect_axis<RR_RECT_XY>(q0.x, q1.x, q0.y, q1.y, q0.z, m.o, m.d, m.tmin, m.tmax, t)) m.tmax = t, side = 1;
                if (test_rect_axis<RR_RECT_XZ>(q0.x, q1.x, q0.z, q1.z, q1.y, m.o, m.d, m.tmin, m.tmax, t)) m.tmax = t, side = 2;
                if (test_rect_axis<RR_RECT_XZ>(q0.x, q1.x, q0.z, q1.z, q0.y, m.o, m.d, m.tmin, m.tmax, t)) m.tmax = t, side = 3;
                if (test_rect_axis<RR_RECT_YZ>(q0.y, q1.y, q0.z, q1.z, q1.x, m.o, m.d, m.tmin, m.tmax, t)) m.tmax = t, side = 4;
                if (test_rect_axis<RR_RECT_YZ>(q0.y, q1.y, q0.z, q1.z, q0.x, m.o, m.d, m.tmin, m.tmax, t)) m.tmax = t, side = 5;
                if (side != NONE) {
                    m.cur = dref(RR_TAG_RECT, first + side);
                    m.hitf = true;
                }
            } else {
                for (uint32_t i = 0; i < 6; ++i) {
                    const float4 r0 = sc.rect[2 * (first + i)], r1 = sc.rect[2 * (first + i) + 1];
                    float t, ca, cb;
                    if (test_rect(r0, r1.x, __float_as_uint(r1.y), m.o, m.d, m.tmin, m.tmax, t, ca, cb)) {
                        m.tmax = t;
                        m.cur = dref(RR_TAG_RECT, first + i);
                        m.hitf = true;
                    }
                }
            }
        } else if (!PLAIN && tag == RR_TAG_MOVING_SPHERE) {
            if (COUNT) cnt.sphere_tests++;
            float t;
            if (test_sphere(msphere_center(q0, q1, world.time()), q2.x, m.o, m.d, m.tmin, m.tmax, t)) {
                m.tmax = t;
                m.cur = node;
                m.hitf = true;
            }
        } else if (tag == RR_TAG_LIST) {
            const uint2 l = sc.list[idx];
            if (l.y != 0) {
                stk[m.sp * stride] = l.x + 1;
                stk[(m.sp + 1) * stride] = frame_word(FK_LIST, idx);
                m.sp += 2;
                m.node = sc.child_refs[l.x];
            }
        } else if (!PLAIN && tag == RR_TAG_INSTANCE) {
            if (COUNT) cnt.inst_visits++;
            stk[m.sp * stride] = frame_word(FK_INST, idx);
            m.sp += 1;
            f3 o = m.o, d = m.d;
            inst_apply_ray(sc.inst[2 * idx], o, d);
            m.set_ray(o, d);
            m.node = __float_as_uint(sc.inst[2 * idx + 1].x);
        } else if (!PLAIN) {  // RR_TAG_MEDIUM: first boundary probe over (f32::MIN, f32::MAX), collider.rs:437
            if (COUNT) cnt.inst_visits++;
            stk[m.sp * stride] = __float_as_uint(m.tmin);
            stk[(m.sp + 1) * stride] = __float_as_uint(m.tmax);
            stk[(m.sp + 2) * stride] = m.cur;
            stk[(m.sp + 3) * stride] = 0u;
            stk[(m.sp + 4) * stride] = frame_word(FK_MED1, idx) | (m.hitf ? FW_FLAG : 0u);
            m.sp += 5;
            m.tmin = -kF32Max;
            m.tmax = kF32Max;
            m.hitf = false;
            m.node = sc.medium[idx].x;
        }
    }
}

template <int COUNT, class World, class Draw, bool WIDE = false, bool PLAIN = false>
__device__ __forceinline__ void trav_pop(const DevScene& sc, Trav& m, const World& world, Draw& draw, uint32_t* stk, uint32_t stride,
                                         Counts& cnt) {
    // ------------------------------------------- return to the callers
    // One return per move: a lane whose caller has nothing left for it either returns again in its next
    // move.  (Returning in a loop here keeps the whole wave waiting for its few lanes with long chains
    // of returns; measured 15 % slower.)
    if (m.node == NONE && m.sp != 0) {
        const uint32_t fw = stk[(m.sp - 1) * stride];
        const uint32_t kind = dtag(fw), idx = didx(fw);
        if (kind <= RR_TAG_LIST) {
            // collider.rs:546-580 after the first child came back: the frame names what comes next
            m.sp -= 1;
            const bool first_hit = m.hitf;
            m.hitf = first_hit || (fw & FW_FLAG);
            if (WIDE) {  // trav_node_wide's frames: the child that follows a miss of the first child has had its box test
                const uint32_t ref = fw & REF_MASK;
                const uint32_t tested = dtag(ref) == RR_TAG_BVH ? NODE_TESTED : 0u;
                m.node = first_hit ? ref : ((fw & FW_MISS_NONE) ? NONE : (ref | tested));
            } else {
                m.node = (!first_hit && (fw & FW_MISS_NONE)) ? NONE : (fw & REF_MASK);
            }
        } else if (kind == FK_BVH2) {
            // left went first: right with t_max = the hit's t, or left once more after a miss
            m.sp -= 1;
            const bool first_hit = m.hitf;
            m.hitf = first_hit || (fw & FW_FLAG);
            m.node = __float_as_uint(first_hit ? sc.bvh[2 * idx + 1].w : sc.bvh[2 * idx].w) & REF_MASK;
        } else if (kind == FK_LIST) {
            const uint32_t pos = stk[(m.sp - 2) * stride];
            const uint2 l = sc.list[idx];
            if (pos < l.x + l.y) {
                stk[(m.sp - 2) * stride] = pos + 1;
                m.node = sc.child_refs[pos];
            } else {
                m.sp -= 2;
            }
        } else if (!PLAIN && kind == FK_INST) {
            m.sp -= 1;
            const uint32_t depth = inst_depth(sc, idx);
            f3 wo, wd, o, d;
            world.ray(wo, wd);
            localize(sc, idx, depth - 1, wo, wd, o, d);  // back to the enclosing space, recomputed from the world ray
            m.set_ray(o, d);
        } else if (!PLAIN && kind == FK_MED1) {
            if (!m.hitf) {
                m.tmin = __uint_as_float(stk[(m.sp - 5) * stride]);
                m.tmax = __uint_as_float(stk[(m.sp - 4) * stride]);
                m.cur = stk[(m.sp - 3) * stride];
                m.hitf = (fw & FW_FLAG) != 0u;
                m.sp -= 5;
            } else {  // second probe over (t1 + 0.0001, f32::MAX), collider.rs:443
                const float t1 = m.tmax;
                stk[(m.sp - 2) * stride] = __float_as_uint(t1);
                stk[(m.sp - 1) * stride] = frame_word(FK_MED2, idx) | (fw & FW_FLAG);
                m.tmin = t1 + 0.0001f;
                m.tmax = kF32Max;
                m.hitf = false;
                m.node = sc.medium[idx].x;
            }
        } else if (!PLAIN) {  // FK_MED2
            const bool found = m.hitf;
            float t2 = m.tmax;
            float t1 = __uint_as_float(stk[(m.sp - 2) * stride]);
            m.tmin = __uint_as_float(stk[(m.sp - 5) * stride]);
            m.tmax = __uint_as_float(stk[(m.sp - 4) * stride]);
            m.cur = stk[(m.sp - 3) * stride];
            m.hitf = (fw & FW_FLAG) != 0u;
            m.sp -= 5;
            if (found) {  // collider.rs:448-474
                if (t1 < m.tmin) t1 = m.tmin;
                if (t2 > m.tmax) t2 = m.tmax;
                if (!(t1 >= t2)) {  // collider.rs:454 returns on `t1 >= t2`: a NaN bound goes on
                    if (t1 < 0.0f) t1 = 0.0f;
                    const float ray_length = len3(m.d);
                    const float dist_inside = (t2 - t1) * ray_length;
                    const float hit_dist = __uint_as_float(sc.medium[idx].y) * log10f(draw.random());
                    if (!(hit_dist > dist_inside)) {
                        m.tmax = t1 + hit_dist / ray_length;
                        m.cur = dref(RR_TAG_MEDIUM, idx);
                        m.hitf = true;
                    }
                }
            }
        }
    }
}

// One step of a single lane's sequence: a node if the lane is at one, then whatever it is at
// after that, then the returns.
template <int COUNT, class World, class Draw>
__device__ __forceinline__ void trav_step(const DevScene& sc, Trav& m, const World& world, Draw& draw, uint32_t* stk,
                                          uint32_t stride, Counts& cnt) {
    trav_node<COUNT>(sc, m, stk, stride, cnt);
    trav_leaf<COUNT>(sc, m, world, stk, stride, cnt);
    trav_pop<COUNT>(sc, m, world, draw, stk, stride, cnt);
}

// Providers for a kernel that holds the world ray and the path's generator in registers.
struct WorldInRegs {
    f3 wo, wd;
    float t;
    __device__ __forceinline__ void ray(f3& o, f3& d) const {
        o = wo;
        d = wd;
    }
    __device__ __forceinline__ float time() const { return t; }
};

// The whole recursive call for `start`, to completion.
template <int COUNT>
__device__ __forceinline__ void machine(const DevScene& sc, uint32_t start, f3 wo, f3 wd, f3 o, f3 d, float time, float tmin,
                                        float& tmax_io, uint32_t& cur_io, Rng& rng, uint32_t* stk, uint32_t stride, Counts& cnt) {
    Trav m;
    m.begin(start, o, d, tmin, tmax_io, cur_io);
    const WorldInRegs world{wo, wd, time};
    do {
        if (COUNT) cnt.lane_steps++;
        trav_step<COUNT>(sc, m, world, rng, stk, stride, cnt);
    } while (!m.finished());
    tmax_io = m.tmax;
    cur_io = m.cur;
}

// ------------------------------------------------- wave-uniform traversal
// Lists, instances, cuboids and media are visited by EVERY ray that reaches them: only
// BVH nodes decide anything from the ray.  The part of the tree above the BVHs is
// therefore walked once per wave with scalar control flow and scalar loads (the record
// of the node all 64 lanes are testing sits in SGPRs), and only BVH subtrees — or
// shapes this layer does not special-case — run on the per-lane machine above.
__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// Primitive or cuboid named by the wave-uniform ref `n`, against the lanes' rays.
template <int COUNT>
__device__ __forceinline__ void eval_leaf(const DevScene& sc, uint32_t n, f3 o, f3 d, float time, float tmin, float& tmax,
                                          uint32_t& cur, Counts& cnt) {
    const uint32_t tag = dtag(n), idx = didx(n);
    if (tag == RR_TAG_RECT) {
        if (COUNT) cnt.rect_tests++;
        const float4 q0 = sc.rect[2 * idx], q1 = sc.rect[2 * idx + 1];
        float t, ca, cb;
        if (test_rect(q0, q1.x, __float_as_uint(q1.y), o, d, tmin, tmax, t, ca, cb)) {
            tmax = t;
            cur = n;
        }
    } else if (tag == RR_TAG_CUBOID) {  // collider.rs:428-430 → 751-772
        const uint32_t first = uni(__float_as_uint(sc.cuboid[2 * idx].w));
#pragma unroll
        for (uint32_t i = 0; i < 6; ++i) {
            if (COUNT) cnt.rect_tests++;
            const float4 q0 = sc.rect[2 * (first + i)], q1 = sc.rect[2 * (first + i) + 1];
            float t, ca, cb;
            if (test_rect(q0, q1.x, __float_as_uint(q1.y), o, d, tmin, tmax, t, ca, cb)) {
                tmax = t;
                cur = dref(RR_TAG_RECT, first + i);
            }
        }
    } else if (tag == RR_TAG_SPHERE) {
        if (COUNT) cnt.sphere_tests++;
        const float4 s = sc.sphere[idx];
        float t;
        if (test_sphere(mk3(s.x, s.y, s.z), s.w, o, d, tmin, tmax, t)) {
            tmax = t;
            cur = n;
        }
    } else if (tag == RR_TAG_TRIANGLE && !sc.no_tri) {
        if (COUNT) cnt.tri_tests++;
        float t;
        if (test_triangle(sc.tri[3 * idx], sc.tri[3 * idx + 1], sc.tri[3 * idx + 2], o, d, tmin, tmax, sc.intended != 0u, t)) {
            tmax = t;
            cur = n;
        }
    } else {  // RR_TAG_MOVING_SPHERE
        if (COUNT) cnt.sphere_tests++;
        const float4 a = sc.msphere[3 * idx], b = sc.msphere[3 * idx + 1], c = sc.msphere[3 * idx + 2];
        float t;
        if (test_sphere(msphere_center(a, b, time), c.x, o, d, tmin, tmax, t)) {
            tmax = t;
            cur = n;
        }
    }
}

// The scene program: the part of the tree that every ray walks in the same order (root list,
// nested lists, instances, cuboids, media, top-level primitives), compiled at upload time
// into a linear array of 32-byte ops.  One wave executes it with scalar control flow and one
// scalar load per op; BVH subtrees and anything else that is data dependent are single ops
// that hand a subtree to the per-lane machine.  Op words: w0 = code | flags << 8, w1 = ref,
// w2..w7 = six f32 operands.
constexpr uint32_t OP_END = 0, /* 1..5: the single transforms OP_XFORM replaced */ OP_RECT_XY = 6,
                   OP_RECT_XZ = 7, OP_RECT_YZ = 8, OP_SPHERE = 9, OP_LEAF_REF = 10, OP_SUBTREE = 11, OP_MED_BEGIN = 12,
                   OP_MED_MID = 13, OP_MED_END = 14, OP_CUBOID = 15, OP_SUBTREE_SIMPLE = 16, OP_MED_SPHERE = 17, OP_XFORM = 18;
// OP_XFORM: back to the world ray (reset), a translation, a rotation about an axis, in that order and each at most once, as
// one op: the ops of an instance chain being entered or left.  An op of three instructions costs a wave as much to fetch
// and dispatch as one of a hundred.  Translation in f0..f2, (sin, cos) in f3, f4.
constexpr uint32_t OPX_RESET = 1u << 8, OPX_TRANSLATE = 1u << 9, OPX_COUNT_TRANSLATE = 1u << 10, OPX_COUNT_ROT = 1u << 11,
                   OPX_ROT_SHIFT = 12;  // two bits: 0 no rotation, 1 + axis otherwise

struct Op {
    uint32_t code, ref;
    float f0, f1, f2, f3, f4, f5;
};
// RR_FIXED_PROGRAM names a file with the ops of ONE scene's program as constants (RR_FIXED_N_OPS, kFixedWords[n][8]: the eight
// words of every op): traverse and traverse_parking then run an unrolled copy of the interpreter's loop in which every op's
// code and operands are known to the compiler.  That is the build rr_scene_specialize makes at run time for an uploaded
// scene; the library's own kernels are built without it.
#ifdef RR_FIXED_PROGRAM
#include RR_FIXED_PROGRAM
#endif

// The transforms of OP_XFORM on the ray (o, d) (collider.rs:481, 499-509); returns the instance visits it counts.
__device__ __forceinline__ uint32_t xform_op(const Op& op, uint32_t flags, f3 wo, f3 wd, f3& o, f3& d) {
    uint32_t visits = 0;
    if (flags & OPX_RESET) o = wo, d = wd;
    if (flags & OPX_TRANSLATE) {
        visits += (flags & OPX_COUNT_TRANSLATE) ? 1u : 0u;
        o = o - mk3(op.f0, op.f1, op.f2);
    }
    const uint32_t rot = (flags >> OPX_ROT_SHIFT) & 3u;
    if (rot != 0u) {
        visits += (flags & OPX_COUNT_ROT) ? 1u : 0u;
        const float sn = op.f3, c = op.f4;
        const f3 po = o, pd = d;
        if (rot == 2u) {  // about y
            o.x = c * po.x - sn * po.z, o.z = sn * po.x + c * po.z;
            d.x = c * pd.x - sn * pd.z, d.z = sn * pd.x + c * pd.z;
        } else if (rot == 1u) {  // about x
            o.y = c * po.y - sn * po.z, o.z = sn * po.y + c * po.z;
            d.y = c * pd.y - sn * pd.z, d.z = sn * pd.y + c * pd.z;
        } else {  // about z
            o.x = c * po.x - sn * po.y, o.y = sn * po.x + c * po.y;
            d.x = c * pd.x - sn * pd.y, d.y = sn * pd.x + c * pd.y;
        }
    }
    return visits;
}

// OP_MED_SPHERE: a constant medium whose boundary is a bare sphere (collider.rs:435-475 over collider.rs:278-292), in one op:
// centre in f0..f2, radius in f3, -1/density in f4, ref = the medium.  The two boundary probes solve the same quadratic for the
// same ray and differ only in the range a root must lie in, so the discriminant, its square root and the two quotients are
// formed once; every value is the one the separate ops (OP_MED_BEGIN, OP_SPHERE, OP_MED_MID, OP_SPHERE, OP_MED_END) form.
template <int COUNT>
__device__ __forceinline__ void medium_in_sphere(const Op& op, f3 o, f3 d, float tmin, float& tmax, uint32_t& cur, bool live, Rng& rng,
                                                 Counts& cnt) {
    if (COUNT && live) cnt.inst_visits++, cnt.sphere_tests++;
    const f3 oc = o - mk3(op.f0, op.f1, op.f2);
    const float a = len2(d);
    const float half_b = dot3(oc, d);
    const float cc = len2(oc) - op.f3 * op.f3;
    const float disc = half_b * half_b - a * cc;
    if (disc < 0.0f) return;  // the first probe misses (a NaN discriminant goes on, as in test_sphere)
    const float sq = sqrtf(disc);
    const float ra = (-half_b - sq) / a, rb = (-half_b + sq) / a;
    float t1 = ra;  // first probe over (f32::MIN, f32::MAX), collider.rs:437
    if (t1 < -kF32Max || kF32Max < t1) {
        t1 = rb;
        if (t1 < -kF32Max || kF32Max < t1) return;
    }
    if (!live) return;
    if (COUNT) cnt.sphere_tests++;
    const float lo = t1 + 0.0001f;  // second probe over (t1 + 0.0001, f32::MAX), collider.rs:443
    float t2 = ra;
    if (t2 < lo || kF32Max < t2) {
        t2 = rb;
        if (t2 < lo || kF32Max < t2) return;
    }
    if (t1 < tmin) t1 = tmin;  // collider.rs:448-474
    if (t2 > tmax) t2 = tmax;
    if (t1 >= t2) return;  // (a NaN bound goes on)
    if (t1 < 0.0f) t1 = 0.0f;
    const float ray_length = len3(d);
    const float dist_inside = (t2 - t1) * ray_length;
    const float hit_dist = op.f4 * log10f(rng.random());
    if (!(hit_dist > dist_inside)) {
        tmax = t1 + hit_dist / ray_length;
        cur = op.ref;
    }
}

// SceneColliders::intersect (collider.rs:751-772) for the world ray (wo, wd).  (tmax, cur) carry
// the closest hit so far through the fold; `live` is false only for lanes whose first medium
// probe missed, until that medium's end op.
// VEC = false builds an interpreter for programs without OP_LEAF_REF and OP_SUBTREE (scenes made of rects,
// spheres, cuboids, instances and media of those): no vector memory access, no traversal machine.
template <int COUNT, bool VEC = true>
__device__ __forceinline__ void traverse(const DevScene& sc, f3 wo, f3 wd, float time, float tmin, float tmax, bool enabled,
                                         Rng& rng, uint32_t* stk, uint32_t stride, float& t_hit, uint32_t& ref_hit, Counts& cnt) {
    // `enabled` is false for lanes that only ride along (the caller keeps this call in wave-uniform
    // control flow): they execute the program but record no hit and count nothing.
    uint32_t cur = NONE;
    f3 o = wo, d = wd;
    bool live = enabled;
    float m_tmin = 0.0f, m_tmax = 0.0f, m_t1 = 0.0f;  // the enclosing call's range and probe 1 of the open medium
    uint32_t m_cur = NONE;
    const RR_CONST_AS Op* prog = reinterpret_cast<const RR_CONST_AS Op*>(sc.prog);
    Op next = prog[0];
    asm volatile("" ::"s"(next.code));  // the first op has arrived before the loop: inside it only the fetch of the op after the current one is in flight
    // A counted loop over a moving pointer: the scalar unit is shared by the sixteen waves of a CU, and every scalar instruction
    // of an op's fetch and dispatch (index arithmetic, a test for OP_END with its exit bookkeeping) is paid sixteen times.
#ifdef RR_FIXED_PROGRAM  // (a per-scene build: see struct Op)
#pragma unroll
    for (uint32_t fpc = 0; fpc < RR_FIXED_N_OPS; ++fpc) {
        Op op;
        op.code = kFixedWords[fpc][0], op.ref = kFixedWords[fpc][1];
        op.f0 = __builtin_bit_cast(float, kFixedWords[fpc][2]), op.f1 = __builtin_bit_cast(float, kFixedWords[fpc][3]);
        op.f2 = __builtin_bit_cast(float, kFixedWords[fpc][4]), op.f3 = __builtin_bit_cast(float, kFixedWords[fpc][5]);
        op.f4 = __builtin_bit_cast(float, kFixedWords[fpc][6]), op.f5 = __builtin_bit_cast(float, kFixedWords[fpc][7]);
        const uint32_t code = op.code & 0xFFu;
        (void)prog, (void)next;
#else
    const uint32_t n_ops = uni(sc.n_ops);
    for (uint32_t pc = 0; pc != n_ops; ++pc) {
        // the op after this one is fetched while this one executes (the program ends in two OP_ENDs)
        const Op op = next;
        ++prog;
        next = *prog;
        const uint32_t code = uni(op.code) & 0xFFu;
#endif
        if (code >= OP_RECT_XY && code <= OP_RECT_YZ) {
            if (COUNT && live) cnt.rect_tests++;
            float t;
            bool hit;
            if (code == OP_RECT_XY)
                hit = test_rect_axis<RR_RECT_XY>(op.f0, op.f1, op.f2, op.f3, op.f4, o, d, tmin, tmax, t);
            else if (code == OP_RECT_XZ)
                hit = test_rect_axis<RR_RECT_XZ>(op.f0, op.f1, op.f2, op.f3, op.f4, o, d, tmin, tmax, t);
            else
                hit = test_rect_axis<RR_RECT_YZ>(op.f0, op.f1, op.f2, op.f3, op.f4, o, d, tmin, tmax, t);
            hit &= live;
            tmax = hit ? t : tmax;
            cur = hit ? op.ref : cur;
        } else if (code == OP_CUBOID) {  // collider.rs:428-430: the six sides in the constructor's order (collider.rs:419-426)
            if (COUNT && live) cnt.rect_tests += 6;
            float t;
            uint32_t side = NONE;
            if (test_rect_axis<RR_RECT_XY>(op.f0, op.f3, op.f1, op.f4, op.f5, o, d, tmin, tmax, t)) tmax = live ? t : tmax, side = 0;
            if (test_rect_axis<RR_RECT_XY>(op.f0, op.f3, op.f1, op.f4, op.f2, o, d, tmin, tmax, t)) tmax = live ? t : tmax, side = 1;
            if (test_rect_axis<RR_RECT_XZ>(op.f0, op.f3, op.f2, op.f5, op.f4, o, d, tmin, tmax, t)) tmax = live ? t : tmax, side = 2;
            if (test_rect_axis<RR_RECT_XZ>(op.f0, op.f3, op.f2, op.f5, op.f1, o, d, tmin, tmax, t)) tmax = live ? t : tmax, side = 3;
            if (test_rect_axis<RR_RECT_YZ>(op.f1, op.f4, op.f2, op.f5, op.f3, o, d, tmin, tmax, t)) tmax = live ? t : tmax, side = 4;
            if (test_rect_axis<RR_RECT_YZ>(op.f1, op.f4, op.f2, op.f5, op.f0, o, d, tmin, tmax, t)) tmax = live ? t : tmax, side = 5;
            if (live && side != NONE) cur = op.ref + side;
        } else if (code == OP_XFORM) {
            const uint32_t visits = xform_op(op, uni(op.code), wo, wd, o, d);
            if (COUNT && live) cnt.inst_visits += visits;
        } else if (code == OP_SPHERE) {
            if (COUNT && live) cnt.sphere_tests++;
            float t;
            if (test_sphere(mk3(op.f0, op.f1, op.f2), op.f3, o, d, tmin, tmax, t) && live) {
                tmax = t;
                cur = op.ref;
            }
        } else if (VEC && code == OP_LEAF_REF) {  // triangles and moving spheres at program level: record by scalar load
            if (live) eval_leaf<COUNT>(sc, uni(op.ref), o, d, time, tmin, tmax, cur, cnt);
        } else if (VEC && (code == OP_SUBTREE || code == OP_SUBTREE_SIMPLE)) {  // BVH (or any shape the program does not flatten): per-lane machine
#ifndef RR_ABLATE_SUBTREE  // timing-only build: wrong images, prices the BVH part of a bounce
            if (live) machine<COUNT>(sc, uni(op.ref), wo, wd, o, d, time, tmin, tmax, cur, rng, stk, stride, cnt);
#endif
        } else if (code == OP_MED_SPHERE) {
            medium_in_sphere<COUNT>(op, o, d, tmin, tmax, cur, live, rng, cnt);
        } else if (sc.lean) {  // (a build for scenes without the three ops below)
        } else if (code == OP_MED_BEGIN) {  // collider.rs:437: first boundary probe over (f32::MIN, f32::MAX)
            if (COUNT && live) cnt.inst_visits++;
            m_tmin = tmin, m_tmax = tmax, m_cur = cur;
            tmin = -kF32Max, tmax = kF32Max, cur = NONE;
        } else if (code == OP_MED_MID) {  // collider.rs:443: second probe over (t1 + 0.0001, f32::MAX)
            live = cur != NONE;
            m_t1 = tmax;
            tmin = m_t1 + 0.0001f, tmax = kF32Max, cur = NONE;
        } else {  // OP_MED_END, collider.rs:448-474
            const bool found = live && cur != NONE;
            float t1 = m_t1, t2 = tmax;
            tmin = m_tmin, tmax = m_tmax, cur = m_cur;
            live = enabled;
            if (found) {
                if (t1 < tmin) t1 = tmin;
                if (t2 > tmax) t2 = tmax;
                if (!(t1 >= t2)) {  // collider.rs:454 returns on `t1 >= t2`: a NaN bound goes on
                    if (t1 < 0.0f) t1 = 0.0f;
                    const float ray_length = len3(d);
                    const float dist_inside = (t2 - t1) * ray_length;
                    const float hit_dist = op.f0 * log10f(rng.random());
                    if (!(hit_dist > dist_inside)) {
                        tmax = t1 + hit_dist / ray_length;
                        cur = op.ref;
                    }
                }
            }
        }
    }
    t_hit = tmax;
    ref_hit = cur;
}

// The same interpreter for a kernel whose lanes may leave a BVH subtree unfinished and come back to it in
// a later call ("parking").  A wave waits in OP_SUBTREE for the deepest descent among its lanes; when only
// a few lanes are still descending, waiting wastes the others.  Here those few keep their traversal state
// (machine state, LDS stack, the interpreter state below) and sit out the rest of the call; in the next
// call of the wave they skip the ops before the subtree they are parked at, join the lanes that enter it
// then, and go on with the rest of the program once their descent is complete.  Every lane still executes
// its own ops in program order on its own state, so the result is the one traverse() computes.
//
// Between calls a parked lane's state lives in LDS, in one of a few slots per wave, not in registers: only
// a handful of lanes are parked at a time, and thirty registers held for them by every lane of the kernel
// (through shading, where the kernel needs most) were what made it spill.
struct ProgState {
    f3 o, d;  // the ray in the space the program is in
    float tmin, tmax;
    uint32_t cur;
    bool live;
    float m_tmin, m_tmax, m_t1;  // the enclosing call's range and probe 1 of the open medium
    uint32_t m_cur;
    uint32_t park_pc;  // the OP_SUBTREE this lane is parked at
    bool parked;
    __device__ __forceinline__ void start(f3 wo, f3 wd, float tmin_, float tmax_) {
        o = wo, d = wd;
        tmin = tmin_, tmax = tmax_;
        cur = NONE;
        live = true;
        m_tmin = m_tmax = m_t1 = 0.0f;
        m_cur = NONE;
        parked = false;
        park_pc = 0;
    }
};

#ifndef RR_PARK_SLOTS
#define RR_PARK_SLOTS 24
#endif
constexpr uint32_t PARK_SLOTS = RR_PARK_SLOTS;  // parked lanes a wave can hold (at most 31: one bit each of a 32-bit mask)
constexpr uint32_t PARK_WORDS = 22;  // machine: o, d, tmin, tmax, cur, node, sp and flags; program: o, d, tmin, the open medium's four
constexpr uint32_t PARK_LDS_WORDS_PER_WAVE = PARK_SLOTS * PARK_WORDS;
constexpr uint32_t PARK_PC_MASK = 0xFFFFFFu;
constexpr uint32_t TRIANGLE_MIN_STEPS = 3;  // see traverse_parking
struct ParkArea {        // wave-uniform
    uint32_t* slots;     // word w of slot k at slots[w * PARK_SLOTS + k]
    uint32_t free_mask;  // bit k: slot k is free
    __device__ __forceinline__ void init(uint32_t* lds_after_stacks) {
        slots = lds_after_stacks + (threadIdx.x >> 6) * PARK_LDS_WORDS_PER_WAVE;
        free_mask = (1u << PARK_SLOTS) - 1u;
    }
};
// `open_medium`: the program can be between OP_MED_BEGIN and OP_MED_END at a subtree; false in the lean builds, whose programs
// have no such ops: the four words of the open medium are then neither saved nor loaded.
// `simple`: the visit is of a simple subtree, inside which the machine's ray and t_min are the program's at that op (nothing
// below changes the ray space or the range's lower end): they are saved once, and after a load the machine's are the
// program's *registers*, not copies the compiler must keep beside them through the loop of the machine.
// `guests`: simple visits also keep the visit's culling margin (in a word a simple visit does not use): a lane that takes steps
// as a guest of another visit needs its own.
__device__ __forceinline__ void park_save(const ParkArea& pa, uint32_t slot, const ProgState& s, const Trav& m, bool open_medium, bool simple,
                                          bool guests = false) {
    uint32_t* w = pa.slots + slot;
    if (!simple) {
        const float v[7] = {m.o.x, m.o.y, m.o.z, m.d.x, m.d.y, m.d.z, m.tmin};
#pragma unroll
        for (uint32_t k = 0; k < 7; ++k) w[k * PARK_SLOTS] = __float_as_uint(v[k]);
    } else if (guests) {
        w[0] = __float_as_uint(m.margin);
    }
    const float v[8] = {m.tmax, s.o.x, s.o.y, s.o.z, s.d.x, s.d.y, s.d.z, s.tmin};
#pragma unroll
    for (uint32_t k = 0; k < 8; ++k) w[(7 + k) * PARK_SLOTS] = __float_as_uint(v[k]);
    if (open_medium) {
        w[15 * PARK_SLOTS] = __float_as_uint(s.m_tmin), w[16 * PARK_SLOTS] = __float_as_uint(s.m_tmax), w[17 * PARK_SLOTS] = __float_as_uint(s.m_t1);
        w[21 * PARK_SLOTS] = s.m_cur;
    }
    w[18 * PARK_SLOTS] = m.cur;
    w[19 * PARK_SLOTS] = m.node;
    w[20 * PARK_SLOTS] = m.sp | (m.hitf ? 1u << 31 : 0u) | (s.live ? 1u << 30 : 0u);
}
__device__ __forceinline__ void park_load(const ParkArea& pa, uint32_t slot, ProgState& s, Trav& m, bool open_medium, bool simple,
                                          bool guests = false) {
    const uint32_t* w = pa.slots + slot;
    float v[8];
#pragma unroll
    for (uint32_t k = 0; k < 8; ++k) v[k] = __uint_as_float(w[(7 + k) * PARK_SLOTS]);
    m.tmax = v[0];
    s.o = mk3(v[1], v[2], v[3]), s.d = mk3(v[4], v[5], v[6]);
    s.tmin = v[7];
    if (!simple) {  // (simple: the caller makes the program's ray and t_min the machine's, for every lane alike)
        float u[7];
#pragma unroll
        for (uint32_t k = 0; k < 7; ++k) u[k] = __uint_as_float(w[k * PARK_SLOTS]);
        m.set_ray(mk3(u[0], u[1], u[2]), mk3(u[3], u[4], u[5]));  // (1/direction and the culling flag follow from the ray)
        m.tmin = u[6];
    } else if (guests) {
        m.margin = __uint_as_float(w[0]);
    }
    if (open_medium) {
        s.m_tmin = __uint_as_float(w[15 * PARK_SLOTS]), s.m_tmax = __uint_as_float(w[16 * PARK_SLOTS]), s.m_t1 = __uint_as_float(w[17 * PARK_SLOTS]);
        s.m_cur = w[21 * PARK_SLOTS];
    }
    m.cur = w[18 * PARK_SLOTS];
    m.node = w[19 * PARK_SLOTS];
    const uint32_t f = w[20 * PARK_SLOTS];
    m.sp = f & 0xFFFFu, m.hitf = (f >> 31) != 0u, s.live = ((f >> 30) & 1u) != 0u;
}

// After a visit of a subtree: lanes whose descent is complete take its result and give their slot back; the others
// are parked at `this_pc`, in the slot they have or in a free one.  Wave-uniform control flow.
// `guest` (only in builds with guests): the lane is parked at another op and took steps of its own descent in this visit: it
// stays parked where it was, finished or not, and only its slot is brought up to date.
__device__ __forceinline__ void park_settle(ParkArea& pa, ProgState& s, const Trav& m, uint32_t& slot, bool in, uint32_t this_pc, bool& act,
                                            bool open_medium, bool simple, bool guests = false, bool guest = false) {
    const bool done = in && m.finished() && !guest;
    uint64_t fm = RR_BALLOT(done && s.parked);  // slots coming back
    while (fm != 0) {
        const uint32_t l = static_cast<uint32_t>(__builtin_ctzll(fm));
        fm &= fm - 1;
        pa.free_mask |= 1u << __builtin_amdgcn_readlane(slot, l);
    }
    uint64_t nm = RR_BALLOT(in && !done && !s.parked);  // lanes parked for the first time
    while (nm != 0) {
        const uint32_t l = static_cast<uint32_t>(__builtin_ctzll(nm));
        nm &= nm - 1;
        const uint32_t k = static_cast<uint32_t>(__builtin_ctz(pa.free_mask));  // (the visit did not end with more such lanes than free slots)
        pa.free_mask &= ~(1u << k);
        if ((threadIdx.x & 63u) == l) slot = k;
    }
    if (in) {
        if (done) {
            s.tmax = m.tmax;
            s.cur = m.cur;
            s.parked = false;
            act = true;
        } else {
            s.parked = true;
            s.park_pc = guest ? s.park_pc : this_pc;
            act = false;
            park_save(pa, slot, s, m, open_medium, simple, guests);
        }
    }
}

// `enabled`: the lane has a ray: a new one (park == NONE: the program starts for it with the range [tmin0, tmax0]) or one
// parked by an earlier call (`park` names its slot and the op it is parked at).
// `park_below`: lanes still descending are parked once fewer than this many are left (0: never park), but not
// before `min_steps` machine steps have been taken in this visit, so that parked lanes always make progress.
// On return a lane with park != NONE has no result yet; the others have theirs in (t_hit, ref_hit).
// GENERAL = false builds the interpreter without the general machine (OP_SUBTREE: subtrees with instances, media, lists or
// moving spheres below BVH nodes), for scenes whose subtrees are all simple: the code a kernel never runs still costs it
// registers (measured: final_scene +1.4 %, Suzanne +1.7 %).
// GUESTS (builds for scenes with two or more subtrees, all of them simple): lanes parked at ANOTHER subtree take the steps of a
// visit as well, on their own descent -- their machine state is complete in their slot and their stack is their own; the lanes
// of a visit that are not in it have nothing to do while it lasts, and a parked path is a lane that sits out the ops and the
// shading of every turn until its descent is complete.  A guest never prolongs a visit, and its result is taken at its own
// op, in program order; visits may then be left after RR_GUEST_MIN_STEPS steps, since parked lanes get on elsewhere too.
// Measured on final_scene (the one benchmark scene with two subtrees; bit-identical): 5 890 -> 5 990 / 5 970 / 5 940 / 5 910
// Msamples/s with visits left after at least 1 / 2 / 4 / 6 steps; the sphere cluster's visits, 14 % of the kernel's time at 16
// lanes, shrink, and its rays finish inside the box tree's visits.  As code that every build carried (a first version) it cost
// the scenes with one subtree 1.6-2.9 %: hence a build of its own.
#ifndef RR_GUEST_MIN_STEPS
#define RR_GUEST_MIN_STEPS 1u
#endif
template <int COUNT, bool GENERAL = true, bool GUESTS = false>
__device__ __forceinline__ void traverse_parking(const DevScene& sc, f3 wo, f3 wd, float time, float tmin0, float tmax0, bool enabled,
                                                 uint32_t& park, ParkArea& pa, Rng& rng, uint32_t* stk, uint32_t stride, uint32_t park_below,
                                                 uint32_t min_steps, uint32_t leaf_wait, Counts& cnt, float& t_hit, uint32_t& ref_hit) {
    ProgState s;
    s.start(wo, wd, tmin0, tmax0);
    uint32_t slot = park >> 24;
    if (enabled && park != NONE) {  // (its state is fetched from its slot when the program comes to the op it is parked at)
        s.parked = true;
        s.park_pc = park & PARK_PC_MASK;
    }
    bool act = enabled && !s.parked;  // the lane executes the ops it comes to
    const RR_CONST_AS Op* prog = reinterpret_cast<const RR_CONST_AS Op*>(sc.prog);
    Op next = prog[0];
    asm volatile("" ::"s"(next.code));  // the first op has arrived before the loop: inside it only the fetch of the op after the current one is in flight
    uint32_t pc = 0;  // wave-uniform by construction; uni() keeps it in an SGPR
    const WorldInRegs world{wo, wd, time};
    Counts& pcnt = cnt;
    // a visit may end with lanes still descending only while there are slots for the ones that have none yet
    auto may_park = [&](bool going, uint32_t n, uint32_t steps) {
        if (!(n < park_below && steps >= (GUESTS ? min(min_steps, RR_GUEST_MIN_STEPS) : min_steps))) return false;
        return static_cast<uint32_t>(__popcll(RR_BALLOT(going && !s.parked))) <= static_cast<uint32_t>(__popc(pa.free_mask));
    };
#ifdef RR_FIXED_PROGRAM  // (a per-scene build: see struct Op)
#pragma unroll
    for (uint32_t fpc = 0; fpc < RR_FIXED_N_OPS; ++fpc) {
        Op op;
        op.code = kFixedWords[fpc][0], op.ref = kFixedWords[fpc][1];
        op.f0 = __builtin_bit_cast(float, kFixedWords[fpc][2]), op.f1 = __builtin_bit_cast(float, kFixedWords[fpc][3]);
        op.f2 = __builtin_bit_cast(float, kFixedWords[fpc][4]), op.f3 = __builtin_bit_cast(float, kFixedWords[fpc][5]);
        op.f4 = __builtin_bit_cast(float, kFixedWords[fpc][6]), op.f5 = __builtin_bit_cast(float, kFixedWords[fpc][7]);
        const uint32_t this_pc = fpc;
        const uint32_t code = op.code & 0xFFu;
        (void)prog, (void)next, (void)pc;
#elif !defined(RR_PARKING_INDEXED_LOOP)
    // A counted loop over a moving pointer, as in traverse.  (Measured slower here while the kernel was short of scalar registers,
    // -1.5 % final_scene and -4.6 % Suzanne; once own_registers and the Philox round keys had given scalar registers back: +1.7 % and +5.6 %.)
    const uint32_t n_ops = uni(sc.n_ops);
    for (; pc != n_ops; ++pc) {
        const Op op = next;
        const uint32_t this_pc = pc;
        ++prog;
        next = *prog;
        const uint32_t code = uni(op.code) & 0xFFu;
#else
    for (;;) {
        const Op op = next;
        const uint32_t this_pc = pc;
        pc = uni(pc + 1);
        next = prog[pc];
        const uint32_t code = uni(op.code) & 0xFFu;
        if (code == OP_END) break;
#endif
        if (code == OP_SUBTREE_SIMPLE) {
#ifdef RR_ABLATE_SUBTREE  // timing-only build: wrong images, prices the BVH part of a bounce
            continue;
#endif
            // A subtree of BVH nodes over spheres, rects, triangles and cuboids.  The root's box and class travel in the op:
            // its test (collider.rs:537-545) is made here for all lanes at once, with scalar operands, and only the lanes
            // that pass descend, on the moves that know nothing but BVH frames and those four kinds of leaves.
            RR_SEC_MARK(cnt, 1);
            const bool rejoin = enabled && s.parked && s.park_pc == this_pc;
            bool guest = GUESTS && enabled && s.parked && s.park_pc != this_pc;
            Trav m;
            m.begin(NONE, s.o, s.d, s.tmin, s.tmax, s.cur);
            if (rejoin || guest) park_load(pa, slot, s, m, sc.lean == 0u, true, GUESTS);
            if (sc.lean) s.tmin = tmin0, s.live = true;  // (only the medium ops a lean program does not have change them: constants, not registers)
            m.set_ray(s.o, s.d);  // (the lanes that rejoin have their ray from their slot: one set of registers for the program's ray and the machine's)
            m.tmin = s.tmin;
#ifndef RR_NODE_MARGIN
            constexpr bool SUBTREE_MARGIN = true;
            {
                const float visit_margin = cull_margin(sc, m.o, mk3(op.f0, op.f1, op.f2), mk3(op.f3, op.f4, op.f5));
                m.margin = guest ? m.margin : visit_margin;  // (a guest's is the one of its own visit, from its slot)
            }
#else
            constexpr bool SUBTREE_MARGIN = false;
#endif
            const uint32_t root = uni(op.ref), ridx = didx(root);
            const uint32_t cls = node_class<COUNT>((uni(op.code) >> 8) & 3u);  // the root has a box (or the op would not be this one)
            // the root's child words: scalar loads that are under way while the box test is made
            const RR_CONST_AS uint32_t* rw = (const RR_CONST_AS uint32_t*)(sc.bvh) + 8u * ridx;
            const uint32_t left = rw[3], right = rw[7];
            bool enter = false;
            if (act && s.live) {
                if (COUNT) cnt.box_tests++;
                enter = node_box_pass<SUBTREE_MARGIN>(sc, m, mk3(op.f0, op.f1, op.f2), mk3(op.f3, op.f4, op.f5), cls);
            }
            asm volatile("" ::"s"(left), "s"(right));
            if (enter) node_descend(sc, m, ridx, left, right, cls, stk, stride);
            const bool local = enter || rejoin;
            guest = guest && !m.finished();  // (a guest whose descent is complete waits for its own op)
            const bool in = local || guest;
            bool moved = false;  // wave-uniform: the visit took at least one step
            RR_SEC_MARK(cnt, 4);
            RR_SEC_ADD(cnt, this_pc < 4u ? 12 : 13, static_cast<unsigned long long>(__popcll(RR_BALLOT(in))));
            // (a visit of a subtree over triangles counts from a head start: at least TRIANGLE_MIN_STEPS instead of min_steps.
            // Most rays are done with a mesh in two or three moves, so that the steps a visit is made to take run with few lanes:
            // Suzanne +2.9 %, the 1.1 M-primitive mesh +3 %)
            uint32_t steps = (cls == BVH_CLASS_CULL_ANY_T && min_steps > TRIANGLE_MIN_STEPS) ? min_steps - TRIANGLE_MIN_STEPS : 0u;
            for (;;) {
                const bool going = in && !m.finished();
                const uint32_t n = static_cast<uint32_t>(__popcll(RR_BALLOT(going)));
                if (n == 0 || may_park(going, n, steps)) break;
                if (GUESTS && !RR_ANY(going && local)) break;  // only guests are left
                moved = true;
                // A leaf move is the dearest of the three and usually only a few lanes are at a leaf: those wait, while other
                // lanes can still move, until they are at least one in `leaf_wait` of the lanes descending (0: nobody waits).
                if (going) trav_node<COUNT, SUBTREE_MARGIN>(sc, m, stk, stride, cnt);
                const bool at_leaf = going && trav_at_leaf(m);
                const uint32_t n_leaf = static_cast<uint32_t>(__popcll(RR_BALLOT(at_leaf)));
                // (when every lane that descends is at a leaf, n_leaf == n, the test holds whatever the fraction)
                if (leaf_wait == 0u || n_leaf * leaf_wait >= n) {
                    if (at_leaf) trav_leaf_simple<COUNT>(sc, m, cnt);
                }
                // (a second return in the same step for lanes whose first one found nothing to do: final_scene +0.1 %, Suzanne -0.8 %;
                // the top word of the stack in a register, so that a return does not wait for LDS: -3 % everywhere, it costs a
                // register and a store per push)
                if (going) trav_pop_simple(sc, m, stk, stride);
                if (COUNT) cnt.lane_steps += going ? 1u : 0u, cnt.wave_steps += (threadIdx.x & 63u) == 0 ? 1u : 0u;
                steps += 1;
                RR_SEC_ADD(cnt, this_pc < 4u ? 8 : 9, 1ull);
                RR_SEC_ADD(cnt, this_pc < 4u ? 14 : 15, static_cast<unsigned long long>(n));
            }
            RR_SEC_MARK(cnt, this_pc < 4u ? 2 : 3);
            guest = guest && moved;  // (nothing to bring up to date in the slot of a guest that took no step)
            park_settle(pa, s, m, slot, local || guest, this_pc, act, sc.lean == 0u, true, GUESTS, guest);
            RR_SEC_MARK(cnt, 4);
        } else if (code == OP_SUBTREE) {
#ifdef RR_ABLATE_SUBTREE  // timing-only build: wrong images, prices the BVH part of a bounce
            continue;
#endif
            if (!GENERAL) continue;  // (a build for scenes without such subtrees: the general machine is not compiled in)
            const bool rejoin = enabled && s.parked && s.park_pc == this_pc;
            const bool enter = act && s.live;
            Trav m;
            m.begin(enter ? uni(op.ref) : NONE, s.o, s.d, s.tmin, s.tmax, s.cur);
            if (rejoin) park_load(pa, slot, s, m, sc.lean == 0u, false);
            if (sc.lean) s.tmin = tmin0, s.live = true;
            const bool in = enter || rejoin;
            uint32_t steps = 0;
            for (;;) {
                const bool going = in && !m.finished();
                const uint32_t n = static_cast<uint32_t>(__popcll(RR_BALLOT(going)));
                if (n == 0 || may_park(going, n, steps)) break;
                if (going) trav_step<COUNT>(sc, m, world, rng, stk, stride, cnt);
                if (COUNT) cnt.lane_steps += going ? 1u : 0u, cnt.wave_steps += (threadIdx.x & 63u) == 0 ? 1u : 0u;
                steps += 1;
            }
            park_settle(pa, s, m, slot, in, this_pc, act, sc.lean == 0u, false);
        } else if (act) {
            if (code >= OP_RECT_XY && code <= OP_RECT_YZ) {
                if (COUNT && s.live) pcnt.rect_tests++;
                float t;
                bool hit;
                if (code == OP_RECT_XY)
                    hit = test_rect_axis<RR_RECT_XY>(op.f0, op.f1, op.f2, op.f3, op.f4, s.o, s.d, s.tmin, s.tmax, t);
                else if (code == OP_RECT_XZ)
                    hit = test_rect_axis<RR_RECT_XZ>(op.f0, op.f1, op.f2, op.f3, op.f4, s.o, s.d, s.tmin, s.tmax, t);
                else
                    hit = test_rect_axis<RR_RECT_YZ>(op.f0, op.f1, op.f2, op.f3, op.f4, s.o, s.d, s.tmin, s.tmax, t);
                hit &= s.live;
                s.tmax = hit ? t : s.tmax;
                s.cur = hit ? op.ref : s.cur;
            } else if (code == OP_CUBOID) {  // collider.rs:428-430: the six sides in the constructor's order (collider.rs:419-426)
                if (COUNT && s.live) pcnt.rect_tests += 6;
                float t;
                uint32_t side = NONE;
                const bool live = s.live;
                if (test_rect_axis<RR_RECT_XY>(op.f0, op.f3, op.f1, op.f4, op.f5, s.o, s.d, s.tmin, s.tmax, t)) s.tmax = live ? t : s.tmax, side = 0;
                if (test_rect_axis<RR_RECT_XY>(op.f0, op.f3, op.f1, op.f4, op.f2, s.o, s.d, s.tmin, s.tmax, t)) s.tmax = live ? t : s.tmax, side = 1;
                if (test_rect_axis<RR_RECT_XZ>(op.f0, op.f3, op.f2, op.f5, op.f4, s.o, s.d, s.tmin, s.tmax, t)) s.tmax = live ? t : s.tmax, side = 2;
                if (test_rect_axis<RR_RECT_XZ>(op.f0, op.f3, op.f2, op.f5, op.f1, s.o, s.d, s.tmin, s.tmax, t)) s.tmax = live ? t : s.tmax, side = 3;
                if (test_rect_axis<RR_RECT_YZ>(op.f1, op.f4, op.f2, op.f5, op.f3, s.o, s.d, s.tmin, s.tmax, t)) s.tmax = live ? t : s.tmax, side = 4;
                if (test_rect_axis<RR_RECT_YZ>(op.f1, op.f4, op.f2, op.f5, op.f0, s.o, s.d, s.tmin, s.tmax, t)) s.tmax = live ? t : s.tmax, side = 5;
                if (live && side != NONE) s.cur = op.ref + side;
            } else if (code == OP_XFORM) {
                const uint32_t visits = xform_op(op, uni(op.code), wo, wd, s.o, s.d);
                if (COUNT && s.live) pcnt.inst_visits += visits;
            } else if (code == OP_SPHERE) {
                if (COUNT && s.live) pcnt.sphere_tests++;
                float t;
                if (test_sphere(mk3(op.f0, op.f1, op.f2), op.f3, s.o, s.d, s.tmin, s.tmax, t) && s.live) {
                    s.tmax = t;
                    s.cur = op.ref;
                }
            } else if (code == OP_LEAF_REF) {  // triangles and moving spheres at program level: record by scalar load
                if (s.live) eval_leaf<COUNT>(sc, uni(op.ref), s.o, s.d, time, s.tmin, s.tmax, s.cur, pcnt);
            } else if (code == OP_MED_SPHERE) {
                medium_in_sphere<COUNT>(op, s.o, s.d, s.tmin, s.tmax, s.cur, s.live, rng, pcnt);
            } else if (sc.lean) {  // (a build for scenes without the three ops below)
            } else if (code == OP_MED_BEGIN) {  // collider.rs:437: first boundary probe over (f32::MIN, f32::MAX)
                if (COUNT && s.live) pcnt.inst_visits++;
                s.m_tmin = s.tmin, s.m_tmax = s.tmax, s.m_cur = s.cur;
                s.tmin = -kF32Max, s.tmax = kF32Max, s.cur = NONE;
            } else if (code == OP_MED_MID) {  // collider.rs:443: second probe over (t1 + 0.0001, f32::MAX)
                s.live = s.cur != NONE;
                s.m_t1 = s.tmax;
                s.tmin = s.m_t1 + 0.0001f, s.tmax = kF32Max, s.cur = NONE;
            } else {  // OP_MED_END, collider.rs:448-474
                const bool found = s.live && s.cur != NONE;
                float t1 = s.m_t1, t2 = s.tmax;
                s.tmin = s.m_tmin, s.tmax = s.m_tmax, s.cur = s.m_cur;
                s.live = true;
                if (found) {
                    if (t1 < s.tmin) t1 = s.tmin;
                    if (t2 > s.tmax) t2 = s.tmax;
                    if (!(t1 >= t2)) {  // collider.rs:454 returns on `t1 >= t2`: a NaN bound goes on
                        if (t1 < 0.0f) t1 = 0.0f;
                        const float ray_length = len3(s.d);
                        const float dist_inside = (t2 - t1) * ray_length;
                        const float hit_dist = op.f0 * log10f(rng.random());
                        if (!(hit_dist > dist_inside)) {
                            s.tmax = t1 + hit_dist / ray_length;
                            s.cur = op.ref;
                        }
                    }
                }
            }
        }
    }
    park = s.parked ? ((slot << 24) | s.park_pc) : NONE;
    t_hit = s.tmax;
    ref_hit = s.cur;
}

// ---------------------------------------------------------- hit records
struct HitRec {
    f3 p, n;
    float t, u, v;
    bool front;
    uint32_t mat;
};

// Material and owner word of the primitive (or medium) a ref names.
struct HitMeta {
    uint32_t mat, owner;
};
__device__ __forceinline__ HitMeta hit_meta(const DevScene& sc, uint32_t ref) {
    const uint32_t tag = dtag(ref), idx = didx(ref);
    HitMeta m;
    if (tag == RR_TAG_SPHERE) {
        const uint2 v = sc.k->sphere_meta[idx];
        m.mat = v.x, m.owner = v.y;
    } else if (tag == RR_TAG_RECT) {
        const float4 v = sc.rect[2 * idx + 1];
        m.mat = __float_as_uint(v.z), m.owner = __float_as_uint(v.w);
    } else if (tag == RR_TAG_TRIANGLE && !sc.no_tri) {
        const uint2 v = sc.k->tri_meta[idx];
        m.mat = v.x, m.owner = v.y;
    } else if (tag == RR_TAG_MOVING_SPHERE) {
        const float4 v = sc.msphere[3 * idx + 2];
        m.mat = __float_as_uint(v.y), m.owner = __float_as_uint(v.z);
    } else {  // medium
        const uint4 v = sc.medium[idx];
        m.mat = v.z, m.owner = v.w;
    }
    return m;
}

// hit_meta and the words of the primitive's record that reconstruct_hit needs, requested together: one wait instead of two
// on the way from the hit to its material.
struct HitFetch {
    HitMeta meta;
    float4 r0, r1;
};
__device__ __forceinline__ HitFetch hit_fetch(const DevScene& sc, uint32_t ref) {
    const uint32_t tag = dtag(ref), idx = didx(ref);
    HitFetch f;
    f.r0 = f.r1 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tag == RR_TAG_SPHERE) {
        const uint2 v = sc.k->sphere_meta[idx];
        f.r0 = sc.sphere[idx];
        f.meta.mat = v.x, f.meta.owner = v.y;
    } else if (tag == RR_TAG_RECT) {
        f.r0 = sc.rect[2 * idx];
        f.r1 = sc.rect[2 * idx + 1];
        f.meta.mat = __float_as_uint(f.r1.z), f.meta.owner = __float_as_uint(f.r1.w);
    } else if (tag == RR_TAG_TRIANGLE && !sc.no_tri) {
        const uint2 v = sc.k->tri_meta[idx];
        f.r0 = sc.tri[3 * idx + 2];
        f.meta.mat = v.x, f.meta.owner = v.y;
    } else if (tag == RR_TAG_MOVING_SPHERE) {
        f.r0 = sc.msphere[3 * idx + 2];
        f.meta.mat = __float_as_uint(f.r0.y), f.meta.owner = __float_as_uint(f.r0.z);
    } else {  // medium
        const uint4 v = sc.medium[idx];
        f.meta.mat = v.z, f.meta.owner = v.w;
    }
    return f;
}

// Rebuild the HitRecord (material.rs:8-38) the reference would have returned for the
// winning (t, primitive): evaluate the primitive in its own space, then walk the
// enclosing instances outwards applying their hit transforms.  Everything fetched here is
// addressed by (ref, meta) alone, so the loads go out together.
// EAGER_CHAIN: the first two instance records of the chain are fetched whatever the primitive's depth, together with the
// primitive's words (the table is never empty; a primitive outside every instance reads the first entry's words without using
// them).  Fetched on condition they arrive word by word in eight branches, after the depth is known; fetched eagerly they
// cost eight registers, which the kernels of scenes with BVH subtrees do not have (measured: Cornell box +1.1 %, final_scene
// -0.6 %, Suzanne -1.4 %), so those keep the conditional form.
template <bool EAGER_CHAIN = false>
__device__ __forceinline__ HitRec reconstruct_hit(const DevScene& sc, f3 wo, f3 wd, float time, float t, uint32_t ref, HitMeta meta,
                                                  float4 r0, float4 r1, bool want_uv) {  // r0, r1: the record words hit_fetch reads
    HitRec h;
    h.t = t;
    h.u = 0.0f;
    h.v = 0.0f;
    h.mat = meta.mat;
    const uint32_t tag = dtag(ref), idx = didx(ref);
    const uint32_t owner = meta.owner;
    // the ray in the primitive's space; the directions inside the two outermost instances are kept for the way back
    const uint32_t depth = sc.lean ? min(owner_depth(owner), 2u) : owner_depth(owner);  // (lean builds: no primitive lies deeper than two instances)
    const float4* chain = sc.chain_rec + owner_inst(owner) * MAX_CHAIN;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 c0 = (EAGER_CHAIN || depth > 0) ? chain[0] : zero4, c1 = (EAGER_CHAIN || depth > 1) ? chain[1] : zero4;
    f3 o = wo, d = wd;
    if (depth > 0) inst_apply_ray(c0, o, d);
    const f3 d1 = d;
    if (depth > 1) inst_apply_ray(c1, o, d);
    const f3 d2 = d;
    for (uint32_t k = 2; k < depth; ++k) inst_apply_ray(chain[k], o, d);
    h.p = o + d * t;
    if (tag == RR_TAG_SPHERE || tag == RR_TAG_MOVING_SPHERE) {
        f3 center;
        float radius;
        if (tag == RR_TAG_SPHERE) {
            center = mk3(r0.x, r0.y, r0.z);
            radius = r0.w;
        } else {
            center = msphere_center(sc.msphere[3 * idx], sc.msphere[3 * idx + 1], time);
            radius = r0.x;
        }
        const f3 outn = (h.p - center) / radius;  // collider.rs:296
        h.front = dot3(d, outn) < 0.0f;
        h.n = h.front ? outn : neg3(outn);
        if (want_uv && tag == RR_TAG_SPHERE) {  // collider.rs:272-276; moving spheres never set uv
            const float theta = acosf(-outn.y);
            const float phi = atan2f(-outn.z, outn.x) + kPi;
            h.u = phi / kTau;
            h.v = theta / kPi;
        }
    } else if (tag == RR_TAG_RECT) {
        const uint32_t axis = __float_as_uint(r1.y);
        float a, b;
        f3 outn;
        // a, b exactly as the intersector formed them (origin + t * direction, not Ray::at)
        if (axis == RR_RECT_XY) {
            a = o.x + t * d.x, b = o.y + t * d.y, outn = mk3(0.0f, 0.0f, 1.0f);
        } else if (axis == RR_RECT_XZ) {
            a = o.x + t * d.x, b = o.z + t * d.z, outn = mk3(0.0f, 1.0f, 0.0f);
        } else {
            a = o.y + t * d.y, b = o.z + t * d.z, outn = mk3(1.0f, 0.0f, 0.0f);
        }
        if (want_uv) {  // only image textures read u, v (texture.rs:84-94)
            h.u = (a - r0.x) / (r0.y - r0.x);
            h.v = (b - r0.z) / (r0.w - r0.z);
        }
        h.front = dot3(d, outn) < 0.0f;
        h.n = h.front ? outn : neg3(outn);
    } else if (tag == RR_TAG_TRIANGLE && !sc.no_tri) {
        const f3 outn = mk3(r0.y, r0.z, r0.w);
        h.front = dot3(d, outn) < 0.0f;
        h.n = h.front ? outn : neg3(outn);
    } else {  // medium, collider.rs:470-473
        h.n = mk3(1.0f, 0.0f, 0.0f);
        h.front = true;
    }
    for (uint32_t k = depth; k > 2; --k) {
        f3 ko, kd;
        localize(sc, owner_inst(owner), k, wo, wd, ko, kd);  // the ray inside instance k-1 of the chain
        inst_apply_hit(chain[k - 1], kd, h.p, h.n, h.front);
    }
    if (depth > 1) inst_apply_hit(c1, d2, h.p, h.n, h.front);
    if (depth > 0) inst_apply_hit(c0, d1, h.p, h.n, h.front);
    return h;
}
// The same for a caller that has the meta words already (the wavefront pipeline's hit records carry them).
template <bool EAGER_CHAIN = false>
__device__ __forceinline__ HitRec reconstruct_hit(const DevScene& sc, f3 wo, f3 wd, float time, float t, uint32_t ref, HitMeta meta,
                                                  bool want_uv) {
    const uint32_t tag = dtag(ref), idx = didx(ref);
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0;
    if (tag == RR_TAG_SPHERE) {
        r0 = sc.sphere[idx];
    } else if (tag == RR_TAG_RECT) {
        r0 = sc.rect[2 * idx];
        r1 = sc.rect[2 * idx + 1];
    } else if (tag == RR_TAG_TRIANGLE && !sc.no_tri) {
        r0 = sc.tri[3 * idx + 2];
    } else if (tag == RR_TAG_MOVING_SPHERE) {
        r0 = sc.msphere[3 * idx + 2];
    }
    return reconstruct_hit<EAGER_CHAIN>(sc, wo, wd, time, t, ref, meta, r0, r1, want_uv);
}

// ------------------------------------------------------- perlin / textures
__device__ __forceinline__ int floor_to_int_sat(float f) {
    // perlin.rs:69-71 casts floor() to isize and keeps the low 8 bits; of a saturated
    // i64 those are 0xFF (positive), 0x00 (negative) and 0 for NaN
    if (f != f) return 0;
    if (f >= 9.2233720368547758e18f) return 0xFF;
    if (f <= -9.2233720368547758e18f) return 0;
    if (f >= 2147483648.0f || f < -2147483648.0f) {
        // beyond i32: every f32 this large is a multiple of 256
        return 0;
    }
    return static_cast<int>(f);
}

__device__ __forceinline__ float perlin_noise(const DevScene& sc, uint32_t table, f3 p) {  // perlin.rs:47-86
    const float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
    const float u = p.x - fx, v = p.y - fy, w = p.z - fz;
    const int i = floor_to_int_sat(fx), j = floor_to_int_sat(fy), k = floor_to_int_sat(fz);
    const float uu = u * u * (3.0f - 2.0f * u);
    const float vv = v * v * (3.0f - 2.0f * v);
    const float ww = w * w * (3.0f - 2.0f * w);
    const uint8_t* px = sc.k->perlin_perm + table * 768u;
    const float* vec = sc.k->perlin_vec + table * 768u;
    float accum = 0.0f;
#pragma unroll
    for (int di = 0; di < 2; ++di)
#pragma unroll
        for (int dj = 0; dj < 2; ++dj)
#pragma unroll
            for (int dk = 0; dk < 2; ++dk) {
                const uint32_t h = px[(i + di) & 255] ^ px[256 + ((j + dj) & 255)] ^ px[512 + ((k + dk) & 255)];
                const f3 c = mk3(vec[3 * h], vec[3 * h + 1], vec[3 * h + 2]);
                const f3 wv = mk3(u - float(di), v - float(dj), w - float(dk));
                accum += (float(di) * uu + (1.0f - float(di)) * (1.0f - uu)) * (float(dj) * vv + (1.0f - float(dj)) * (1.0f - vv)) *
                         (float(dk) * ww + (1.0f - float(dk)) * (1.0f - ww)) * dot3(c, wv);
            }
    return accum;
}

__device__ __forceinline__ float perlin_turb(const DevScene& sc, uint32_t table, f3 p, uint32_t depth) {  // perlin.rs:87-98
    float accum = 0.0f, weight = 1.0f;
    f3 tp = p;
    for (uint32_t i = 0; i < depth; ++i) {
        accum += weight * perlin_noise(sc, table, tp);
        weight *= 0.5f;
        tp = tp * 2.0f;
    }
    return fabsf(accum);
}

__device__ __forceinline__ float clampf(float x, float lo, float hi) {  // utils.rs:57-65
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}
__device__ __forceinline__ uint32_t f32_as_u32_sat(float f) {  // Rust `as u32`
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return static_cast<uint32_t>(f);
}

// Out of line on purpose: materials of one solid colour never come here (DevMaterial::solid), and inlined into the kernels the
// turbulence loop costs every path the registers it needs (measured: 50 fewer spilled registers, +3 % on the BVH scenes).
#ifdef RR_INLINE_TEX
__device__ __forceinline__ f3 texture_color(
#else
__device__ __attribute__((noinline)) f3 texture_color(
#endif
    const DevScene& sc, uint32_t tex, float u, float v, f3 p) {  // texture.rs:66-103
#ifdef RR_ABLATE_TEXTURES
    return mk3(u, v, p.x);
#endif
    const DevTexture& t = sc.k->textures[tex];
    const uint32_t type = t.type;
    if (type == RR_TEX_SOLID) return mk3(t.c0[0], t.c0[1], t.c0[2]);
    if (type == RR_TEX_CHECKERED) {
        const float sines = sinf(10.0f * p.x) * sinf(10.0f * p.y) * sinf(10.0f * p.z);
        return sines < 0.0f ? mk3(t.c0[0], t.c0[1], t.c0[2]) : mk3(t.c1[0], t.c1[1], t.c1[2]);
    }
    if (type == RR_TEX_NOISE) {
        const float tv = perlin_turb(sc, t.index, p, t.depth);
        return mk3(1.0f, 1.0f, 1.0f) * 0.5f * (1.0f + sinf(t.scale * p.z + 10.0f * tv));
    }
    const DevImage im = sc.k->images[t.index];
    const float uc = clampf(u, 0.0f, 1.0f);
    const float vc = 1.0f - clampf(v, 0.0f, 1.0f);
    uint32_t i = f32_as_u32_sat(uc * float(im.width));
    uint32_t j = f32_as_u32_sat(vc * float(im.height));
    if (i >= im.width) i = im.width - 1;
    if (j >= im.height) j = im.height - 1;
    const uint8_t* px = sc.k->image_bytes + im.offset + (static_cast<uint64_t>(j) * im.width + i) * 3u;
    const float cs = 1.0f / 255.0f;
    return mk3(float(px[0]) * cs, float(px[1]) * cs, float(px[2]) * cs);
}

// ---------------------------------------------------------------- sampling
__device__ __forceinline__ f3 random_in_unit_sphere(Rng& g) {  // vec3.rs:223-231
    for (;;) {
        float x = g.randrange(-1.0f, 1.0f);
        float y = g.randrange(-1.0f, 1.0f);
        float z = g.randrange(-1.0f, 1.0f);
        f3 p = mk3(x, y, z);
        if (len2(p) >= 1.0f) continue;
        return p;
    }
}
__device__ __forceinline__ f3 random_cosine_direction(Rng& g) {  // vec3.rs:257-266
    const float r1 = g.random();
    const float r2 = g.random();
    const float z = sqrtf(1.0f - r2);
    const float phi = kTau * r1;
    float sn, cs;  // one argument reduction for both (the values are those of sinf and cosf)
    sincosf(phi, &sn, &cs);
    const float x = cs * sqrtf(r2);
    const float y = sn * sqrtf(r2);
    return mk3(x, y, z);
}
__device__ __forceinline__ float reflectance(float cosin, float ref_idx) {  // material.rs:116-120
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 *= r0;
    const float x = 1.0f - cosin;
    return r0 + (1.0f - r0) * (x * x * x * x * x);
}

// Geometry::pdf_value of a bare XZRect (collider.rs:389-399, 714-719); anything else has pdf 0.
__device__ __forceinline__ float xzrect_pdf_value(bool is_xzrect, float x0, float x1, float z0, float z1, float k, f3 origin, f3 v) {
    if (!is_xzrect) return 0.0f;
    float t, ca, cb;
    if (!test_rect(make_float4(x0, x1, z0, z1), k, RR_RECT_XZ, origin, v, 0.001f, kF32Max, t, ca, cb)) return 0.0f;
    const f3 outn = mk3(0.0f, 1.0f, 0.0f);
    const f3 n = dot3(v, outn) < 0.0f ? outn : neg3(outn);
    const float area = (x1 - x0) * (z1 - z0);
    const float dist_sqrd = t * t * len2(v);
    const float cosine = fabsf(dot3(v, n) / len3(v));
    return dist_sqrd / (cosine * area);
}
__device__ __forceinline__ float light_pdf_value(const DevScene& sc, f3 origin, f3 v) {
    if (sc.book && sc.k->n_lights != 0u) {  // the labelled book estimator with a list of lights: the mean of the members' values
        const uint32_t n = sc.k->n_lights;
        const float weight = 1.0f / static_cast<float>(n);
        float sum = 0.0f;
        for (uint32_t i = 0; i < n; ++i) {
            const float4 a = sc.k->lights[2 * i], b = sc.k->lights[2 * i + 1];
            sum += weight * xzrect_pdf_value(__float_as_uint(b.y) != 0u, a.x, a.y, a.z, a.w, b.x, origin, v);
        }
        return sum;
    }
    return xzrect_pdf_value(sc.k->light_is_xzrect != 0u, sc.k->light_x0, sc.k->light_x1, sc.k->light_z0, sc.k->light_z1, sc.k->light_k, origin, v);
}
// Geometry::random (collider.rs:400-403, 720-725): towards a uniform point of a bare XZRect; (1,0,0) for anything else.
__device__ __forceinline__ f3 light_random(const DevScene& sc, f3 origin, Rng& g) {
    bool is_xz = sc.k->light_is_xzrect != 0u;
    float x0 = sc.k->light_x0, x1 = sc.k->light_x1, z0 = sc.k->light_z0, z1 = sc.k->light_z1, k = sc.k->light_k;
    if (sc.book && sc.k->n_lights != 0u) {  // one member of the list, chosen uniformly
        const uint32_t n = sc.k->n_lights;
        const uint32_t i = min(static_cast<uint32_t>(g.random() * static_cast<float>(n)), n - 1u);
        const float4 a = sc.k->lights[2 * i], b = sc.k->lights[2 * i + 1];
        is_xz = __float_as_uint(b.y) != 0u;
        x0 = a.x, x1 = a.y, z0 = a.z, z1 = a.w, k = b.x;
    }
    if (!is_xz) return mk3(1.0f, 0.0f, 0.0f);
    const float x = g.randrange(x0, x1);
    const float z = g.randrange(z0, z1);
    return mk3(x, k, z) - origin;
}
__device__ __forceinline__ f3 refract3(f3 uv, f3 n, float etai_over_etat) {  // vec3.rs:272-277
    const float cos_theta = -fminf(dot3(uv, n), 1.0f);
    const f3 perp = (uv + n * cos_theta) * etai_over_etat;
    const f3 parallel = n * -(sqrtf(fabsf(1.0f - len2(perp))));
    return perp + parallel;
}

// One ray_color call (render.rs:15-41) minus the recursion.  Returns true when the
// path goes on with ray (o, d) and its throughput multiplied by `value`; returns
// false when it ends with radiance `value` (background, emitted light, or the
// emitted colour of a surface that does not scatter).  One field for both on purpose: with two, the
// compiler merged the stores of the different exits into one store through a selected address, which kept
// the whole struct in scratch memory (three stores and three loads per bounce, and a wait for them).
// A path whose throughput is exactly zero in every channel: whatever it meets from here on is multiplied by zero.  That
// is every path after a metal, glass or smoke hit (scattering pdf 0, material.rs:170-175) and after a Lambertian hit whose
// sampled direction points below the surface (material.rs:111-114) -- in final_scene, where most rays scatter in the fog of
// radius 5000 before they leave, 57 % of all rays.  The reference follows such a path to its end (render.rs:29-34) and adds
// its 0; the kernels end it here and add the 0 at once, unless RR_FLAG_TRACE_DEAD_PATHS asks for the reference's walk.
// The two differ only where the rest of the walk would have met a NaN or an infinity (0 * NaN = NaN: the 0/0 of render.rs:34
// at a surface hit whose mixture pdf is 0), which the reference turns into a zeroed pixel of that slice (render.rs:102-119).
// A throughput that is NaN in every channel ends a path as well, and exactly: NaN times whatever follows is NaN, which is what
// the ended path leaves as its sample (NaN * 0).  Anything else (one channel zero, an infinity) goes on.
__device__ __forceinline__ bool dead_path(f3 thr) {
    return (thr.x == 0.0f && thr.y == 0.0f && thr.z == 0.0f) || (thr.x != thr.x && thr.y != thr.y && thr.z != thr.z);
}

struct Bounce {
    f3 o, d;
    f3 value;
};
template <bool EAGER_CHAIN = false, bool FETCHED = false>
__device__ __forceinline__ bool shade(const DevScene& sc, f3 wo, f3 wd, float time, float t, uint32_t ref, HitMeta meta, Rng& g,
                                      Bounce& out, float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), float4 r1 = make_float4(0.f, 0.f, 0.f, 0.f)) {
    // FETCHED: r0, r1 are the primitive's record words, read with the meta words (hit_fetch)
    if (ref == NONE) {
        out.value = mk3(sc.k->background[0], sc.k->background[1], sc.k->background[2]);
        return false;
    }
    // The material record is fetched first, so that it travels while the hit record is rebuilt.  Only image textures read
    // u, v; the material is not known when reconstruct_hit asks, so it forms them whenever the scene has images at all.
    const DevMaterial m = sc.k->materials[meta.mat];
    const HitRec rec = FETCHED ? reconstruct_hit<EAGER_CHAIN>(sc, wo, wd, time, t, ref, meta, r0, r1, sc.k->images != nullptr)
                               : reconstruct_hit<EAGER_CHAIN>(sc, wo, wd, time, t, ref, meta, sc.k->images != nullptr);
    f3 attenuation, sdir = mk3(0.0f, 0.0f, 0.0f);
    const f3 own = mk3(m.r, m.g, m.b);
    if (m.type == RR_MAT_EMISSIVE) {  // material.rs:156-158, 176-181
        out.value = m.solid ? own : texture_color(sc, m.texture, rec.u, rec.v, rec.p);
        return false;
    }
    // scatter (material.rs:85-169).  ray_color throws the scattered direction away (render.rs:32), so
    // only the return value and the attenuation matter; random draws that can influence neither — the
    // cosine direction of a Lambertian, a dielectric's reflect-or-refract coin, an isotropic direction —
    // are not taken from the stream at all (the CPU oracle follows the same rule).
    if (m.type == RR_MAT_LAMBERTIAN) {  // material.rs:85-92
        attenuation = m.solid ? own : texture_color(sc, m.texture, rec.u, rec.v, rec.p);
    } else if (m.type == RR_MAT_GLOSSY) {  // material.rs:121-126: the fuzz draw decides whether the path ends
        const f3 reflected = unit3(wd) - rec.n * 2.0f * dot3(unit3(wd), rec.n);
        const f3 sd = reflected + random_in_unit_sphere(g) * m.param;
        attenuation = own;
        if (!(dot3(sd, rec.n) > 0.0f)) {
            out.value = mk3(0.0f, 0.0f, 0.0f);
            return false;
        }
        sdir = sd;
    } else if (m.type == RR_MAT_DIELECTRIC) {  // material.rs:128-148: always scatters, attenuation 1
        attenuation = mk3(1.0f, 1.0f, 1.0f);
        if (sc.book) {  // (the direction the reference's ray_color throws away)
            const float ratio = rec.front ? 1.0f / m.param : m.param;
            const f3 ud = unit3(wd);
            const float cos_theta = fminf(dot3(neg3(ud), rec.n), 1.0f);
            const float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
            // (`||` does not evaluate its right side, and so does not draw, when the left side holds)
            if (ratio * sin_theta > 1.0f || reflectance(cos_theta, ratio) > g.random()) sdir = ud - rec.n * 2.0f * dot3(ud, rec.n);
            else sdir = refract3(ud, rec.n, ratio);
        }
    } else {  // isotropic, material.rs:150-154
        if (sc.book) sdir = random_in_unit_sphere(g);
        attenuation = m.solid ? own : texture_color(sc, m.texture, rec.u, rec.v, rec.p);
    }
    if (sc.book && m.type != RR_MAT_LAMBERTIAN) {  // RR_SEMANTICS_BOOK_ESTIMATOR: a specular material's own ray, weighted by its attenuation alone
        out.o = rec.p;
        out.d = sdir;
        out.value = attenuation;
        return true;
    }
    // render.rs:29-33: mixture of the light pdf and the cosine pdf about the hit normal
    const f3 w = unit3(rec.n);  // vec3.rs:205-215
    const f3 a = fabsf(w.x) > 0.9f ? mk3(0.0f, 1.0f, 0.0f) : mk3(1.0f, 0.0f, 0.0f);
    const f3 v = unit3(cross3(w, a));
    const f3 u = cross3(w, v);
    f3 dir;
    if (g.random() < 0.5f) {  // pdf.rs:51-53 → collider.rs:400-403 / 720-725
        dir = light_random(sc, rec.p, g);
    } else {
        const f3 cd = random_cosine_direction(g);
        dir = u * cd.x + v * cd.y + w * cd.z;
    }
    const f3 udir = unit3(dir);
    const float cosin = dot3(udir, w);  // pdf.rs:33-36
    const float cos_pdf = cosin <= 0.0f ? 0.0f : cosin / kPi;
    const float pdf_val = 0.5f * light_pdf_value(sc, rec.p, dir) + 0.5f * cos_pdf;  // pdf.rs:48-50
    float sp = 0.0f;  // material.rs:170-175: every non-Lambertian material has scattering pdf 0
    if (m.type == RR_MAT_LAMBERTIAN) {
        const float cs = dot3(rec.n, udir);
        sp = cs < 0.0f ? 0.0f : cs / kPi;
    }
    out.o = rec.p;
    out.d = dir;
    out.value = (attenuation * sp) / pdf_val;  // render.rs:34, applied front to back
    return true;
}

// Camera::get_ray (camera.rs:76-84) after the pixel jitter of render.rs:96-97.
__device__ __forceinline__ void camera_ray(const DevScene& sc, uint32_t width, uint32_t height, uint32_t i, uint32_t j, Rng& g,
                                           f3& o, f3& d, float& time) {
    const float u = (float(i) + g.random()) / (float(width) - 1.0f);
    const float v = (float(j) + g.random()) / (float(height) - 1.0f);
    float dx = 0.0f, dy = 0.0f;
    if (sc.k->cam.lens_radius != 0.0f) {  // with aperture 0 the lens sample is multiplied by zero: not drawn
        for (;;) {                     // vec3.rs:247-255
            dx = g.randrange(-1.0f, 1.0f);
            dy = g.randrange(-1.0f, 1.0f);
            if (dx * dx + dy * dy + 0.0f * 0.0f >= 1.0f) continue;
            break;
        }
    }
    const f3 rd = mk3(dx, dy, 0.0f) * sc.k->cam.lens_radius;
    const f3 offset = mk3(sc.k->cam.u[0], sc.k->cam.u[1], sc.k->cam.u[2]) * rd.x + mk3(sc.k->cam.v[0], sc.k->cam.v[1], sc.k->cam.v[2]) * rd.y;
    const f3 origin = mk3(sc.k->cam.origin[0], sc.k->cam.origin[1], sc.k->cam.origin[2]);
    o = origin + offset;
    d = mk3(sc.k->cam.lower_left[0], sc.k->cam.lower_left[1], sc.k->cam.lower_left[2]) + mk3(sc.k->cam.horizontal[0], sc.k->cam.horizontal[1], sc.k->cam.horizontal[2]) * u + mk3(sc.k->cam.vertical[0], sc.k->cam.vertical[1], sc.k->cam.vertical[2]) * v - origin - offset;
    time = g.randrange(sc.k->cam.shutter_open, sc.k->cam.shutter_close);
}

}  // namespace rrdev

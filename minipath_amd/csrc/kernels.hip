// gfx950 (CDNA4) kernels for minipath's per-pixel sampling hot path.
//
// Lane mapping (DESIGN.md "Kernels"):
//   * ray generation, shading and accumulation are LANE-PARALLEL: one (pixel, sample) per lane;
//   * BVH traversal is GROUP-PARALLEL: a 64-wide wavefront is 8 groups of 8 lanes, one ray per group, lane i of a
//     group owns child i of an inner node / triangle i of a leaf packet -- the reference's 8-wide AVX2 lane i
//     (ray_bvh_intersection.rs:104-162).  The wavefront keeps a ray queue and eight traversal stacks in LDS;
//     active rays are compacted into the queue with __ballot + mbcnt and groups pull the next ray when they finish.
//
// Arithmetic contract: f32, compiled with -ffp-contract=off; fused multiply-adds only where the reference writes
// mul_add / mul_sub (util/simba.rs:57-67); IEEE division and sqrt; no fast-math.  The results are bit-identical to
// the CPU oracle's.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdlib>

#include "mp_internal.h"

namespace mp {
namespace {

constexpr uint32_t kNoPrim = 0xFFFFFFFFu;
constexpr int kQueueFloats = 6 * 64;  // ox,oy,oz,dx,dy,dz (unit direction) x 64 slots ; hit (t,prim,u,v) aliases rows 0..3

__device__ __forceinline__ float as_f(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t as_u(float f) { return __float_as_uint(f); }
// set bits of a wave mask below this lane (v_mbcnt: no per-lane mask register to keep)
__device__ __forceinline__ int rank_below(uint64_t m) {
    return static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u)));
}

// LDS traffic between lanes of ONE wavefront: DS operations of a wave execute in issue order, so only the
// compiler has to be kept from reordering them.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Global memory written by some lanes of ONE wavefront and read by others (the pooled path kernel's ray queue): the wave's
// outstanding stores are waited for (workgroup-scope release / acquire; the vector L1 of a CU is coherent for its own waves).
__device__ __forceinline__ void wave_mem_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// ---- 8-lane group collectives on DPP (no LDS traffic) ---------------------------------------------------------
// row_half_mirror (0x141) pairs lane i with 7-i inside every group of 8; quad_perm [1,0,3,2] (0xB1) and
// [2,3,0,1] (0x4E) finish the butterfly inside each quad.  All 8 lanes end with the group's reduction.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u(uint32_t v) {
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, 0xF, 0xF, false));
}
// Minimum of hit distances (>= 0, possibly -0.0, never NaN): on those the signed-integer order of the bit patterns is the float
// order (-0.0 below +0.0, as v_min_f32 has it), and v_min_i32 takes the DPP operand directly (no canonicalising v_max).
__device__ __forceinline__ float group_min_t(float v) {
    int k = __float_as_int(v);
    k = min(k, __builtin_amdgcn_update_dpp(0, k, 0x141, 0xF, 0xF, false));
    k = min(k, __builtin_amdgcn_update_dpp(0, k, 0xB1, 0xF, 0xF, false));
    k = min(k, __builtin_amdgcn_update_dpp(0, k, 0x4E, 0xF, 0xF, false));
    return __int_as_float(k);
}
__device__ __forceinline__ uint32_t group_min_u(uint32_t v) {
    v = min(v, dpp_u<0x141>(v));
    v = min(v, dpp_u<0xB1>(v));
    v = min(v, dpp_u<0x4E>(v));
    return v;
}

// ---- RNG: rand 0.9.3 SmallRng = Xoshiro256++ (seeded mode, include/minipath_hip.h) -----------------------------
struct Rng {
    uint64_t s0, s1, s2, s3;
};
__device__ __forceinline__ uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
__device__ __forceinline__ uint64_t splitmix(uint64_t& state) {
    state += 0x9e3779b97f4a7c15ull;
    uint64_t z = state;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ void rng_seed(Rng& r, uint64_t state) {
    r.s0 = splitmix(state);
    r.s1 = splitmix(state);
    r.s2 = splitmix(state);
    r.s3 = splitmix(state);
}
__device__ __forceinline__ uint32_t rng_next_u32(Rng& r) {
    uint64_t result = rotl64(r.s0 + r.s3, 23) + r.s0;
    uint64_t t = r.s1 << 17;
    r.s2 ^= r.s0;
    r.s3 ^= r.s1;
    r.s1 ^= r.s2;
    r.s0 ^= r.s3;
    r.s2 ^= t;
    r.s3 = rotl64(r.s3, 45);
    return static_cast<uint32_t>(result >> 32);
}
__device__ __forceinline__ float rng_value0_1(Rng& r) { return as_f(0x3F800000u | (rng_next_u32(r) >> 9)) - 1.0f; }

struct RayGen {
    mp_camera_sampler s;
    float jitter_scale;  // UniformFloat::new_inclusive(-0.5, 0.5).scale, computed on the host
    uint32_t width, spp;
    uint64_t seed;  // mix(settings.seed), see mixed_seed()
};

struct Ray {
    float ox, oy, oz, dx, dy, dz, ix, iy, iz;
};

// geometry/mod.rs:45-54
__device__ __forceinline__ void ray_new(float ox, float oy, float oz, float dx, float dy, float dz, Ray& r) {
    float n = sqrtf(dx * dx + dy * dy + dz * dz);
    r.ox = ox; r.oy = oy; r.oz = oz;
    r.dx = dx / n; r.dy = dy / n; r.dz = dz / n;
    r.ix = (r.dx == 0.0f) ? INFINITY : 1.0f / r.dx;
    r.iy = (r.dy == 0.0f) ? INFINITY : 1.0f / r.dy;
    r.iz = (r.dz == 0.0f) ? INFINITY : 1.0f / r.dz;
}

// rand_distr::UnitDisc: rejection on two Uniform(-1,1) draws (camera.rs:184)
__device__ __forceinline__ void unit_disc(Rng& rng, float& x1, float& x2) {
    for (;;) {
        x1 = rng_value0_1(rng) * 2.0f + (-1.0f);
        x2 = rng_value0_1(rng) * 2.0f + (-1.0f);
        if (x1 * x1 + x2 * x2 <= 1.0f) break;
    }
}

// CameraSampler::sample_ray camera.rs:176-191 on an already seeded stream
__device__ __forceinline__ void sample_ray_rng(const RayGen& P, uint32_t x, uint32_t y, Rng& rng, Ray& r) {
    float film_u = static_cast<float>(x) + (rng_value0_1(rng) * P.jitter_scale + (-0.5f));
    float film_v = static_cast<float>(y) + (rng_value0_1(rng) * P.jitter_scale + (-0.5f));
    float fv = film_v * P.s.pixel_scale, fu = film_u * P.s.pixel_scale;
    float fx = P.s.film_origin_offset[0] + P.s.up[0] * fv - P.s.right[0] * fu;
    float fy = P.s.film_origin_offset[1] + P.s.up[1] * fv - P.s.right[1] * fu;
    float fz = P.s.film_origin_offset[2] + P.s.up[2] * fv - P.s.right[2] * fu;
    float x1, x2;
    unit_disc(rng, x1, x2);
    float a = P.s.lens_radius * x1, b = P.s.lens_radius * x2;
    float lx = P.s.right[0] * a + P.s.up[0] * b;
    float ly = P.s.right[1] * a + P.s.up[1] * b;
    float lz = P.s.right[2] * a + P.s.up[2] * b;
    ray_new(P.s.center[0] + lx, P.s.center[1] + ly, P.s.center[2] + lz, lx * P.s.lens_weight - fx,
            ly * P.s.lens_weight - fy, lz * P.s.lens_weight - fz, r);
}

// key = mix(seed) + sample index; `P.seed` already holds mix(seed) (mixed_seed(), on the host)
__device__ __forceinline__ uint64_t sample_key(const RayGen& P, uint32_t x, uint32_t y, uint32_t sample) {
    return P.seed + ((static_cast<uint64_t>(y) * P.width + x) * P.spp + sample);
}

// seeded per (pixel, sample): include/minipath_hip.h "Seeded mode"
__device__ __forceinline__ void sample_ray(const RayGen& P, uint32_t x, uint32_t y, uint32_t sample, Ray& r) {
    Rng rng;
    rng_seed(rng, sample_key(P, x, y, sample));
    sample_ray_rng(P, x, y, rng, r);
}

// ---- traversal --------------------------------------------------------------------------------------------------
// util/simba.rs:57-59
__device__ __forceinline__ float fma_dot(float ax, float ay, float az, float bx, float by, float bz) {
    return __builtin_fmaf(az, bz, __builtin_fmaf(ay, by, ax * bx));
}
// util/simba.rs:61-67 : mul_sub(a, b, c) = a*b - c, c rounded first
__device__ __forceinline__ float fms(float a, float b, float c) { return __builtin_fmaf(a, b, -c); }

// Lane masks straight from the compare unit (v_cmp writes the 64-bit mask; no bool -> int -> ballot round trip, which costs two
// VALU per use): LLVM CmpInst predicate numbers.
constexpr int kFcmpOGT = 2, kFcmpOGE = 3, kFcmpOLT = 4, kFcmpOLE = 5;
__device__ __forceinline__ uint64_t mask_gt(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, kFcmpOGT); }
__device__ __forceinline__ uint64_t mask_lt(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, kFcmpOLT); }
__device__ __forceinline__ uint64_t mask_le(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, kFcmpOLE); }
__device__ __forceinline__ uint64_t mask_ge(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, kFcmpOGE); }

// Traces `nrays` rays held in the wave's LDS queue `q` (rows ox,oy,oz,dx,dy,dz; slot = column).
// On return rows 0..3 of each slot hold the closest hit: t (f32::MAX on miss), prim (bits), u, v.
// impl Object for TriangleBvh::intersect, ray_bvh_intersection.rs:26-96, for 8 rays at a time.
// BFE: the count of passing children below a lane from v_bfe_u32 instead of a loop-invariant mask register (one VGPR less, one
// VALU more per node step): pays in trace_rays_kernel, which sits at the 64-VGPR cap, and costs 0.5 % in render_paths_kernel.
// QS = queue stride = ray slots.  QS == 64: the wave's LDS queue (rows ox,oy,oz,dx,dy,dz; the hit aliases rows 0..3; inverse
// directions computed here, one slot per lane).  QS > 64: the pooled path kernel's queue in global memory (rows 0..5 origin and
// direction, 6..8 inverse direction as Ray::new has it, 9..12 the hit: the rays are needed again for shading).
template <bool PATCH_NAN, bool BFE, int QS = 64>
__device__ __forceinline__ void trace_wave_impl(const DevScene& sc, float* __restrict__ q, uint2* __restrict__ stack_base,
                                                int nrays) {
    constexpr int HB = QS == 64 ? 0 : 9;  // first hit row
    const int lane = static_cast<int>(threadIdx.x) & 63;
    const int g = lane >> 3, li = lane & 7;
    uint2* stack = stack_base + g * sc.stack_cap;
    const uint32_t lanes_below = (1u << li) - 1u;

    int slot = -1, sp = 0, head = 0;
    uint32_t pk = 0, pk_end = 0, seq = 0;
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0, ix = 0, iy = 0, iz = 0;
    // Lane li fetches child li / triangle li as whole 16-byte words from the AoS copies (two loads per node, three per packet)
    // PATCH_NAN = some queued ray has an infinite inverse direction component: literal reference tree; otherwise the wide tree
    // (thin nodes absorbed into their parents: same hits for finite inverses, device_tree.cpp)
    const float4* __restrict__ nodes4 = reinterpret_cast<const float4*>(PATCH_NAN ? sc.nodes_lit : sc.nodes_aos);
    const uint32_t root = PATCH_NAN ? sc.root_lit : sc.root;
    const float* __restrict__ tris = sc.tris_aos;
    float best_t = FLT_MAX;                 // best.t, group-uniform (ray_bvh_intersection.rs:34-37)
    float tl = FLT_MAX, ul = 0, vl = 0;     // this lane's earliest closest candidate
    uint32_t pkl = kNoPrim, seql = 0;
    // inv_direction as Ray::new has it (geometry/mod.rs:49-53) for the ray in queue slot `lane`, computed ONCE here for all 64
    // slots in parallel (three IEEE divisions per call instead of three per refill step); a group fetches its ray's inverse with
    // ds_bpermute when it pulls the ray.  Not queued in LDS: three rows more per wave would cost a resident wave on deep trees.
    float pix = 0.0f, piy = 0.0f, piz = 0.0f;
    if (QS == 64) {
        const float qx = q[3 * 64 + lane], qy = q[4 * 64 + lane], qz = q[5 * 64 + lane];
        pix = (qx == 0.0f) ? INFINITY : 1.0f / qx; piy = (qy == 0.0f) ? INFINITY : 1.0f / qy; piz = (qz == 0.0f) ? INFINITY : 1.0f / qz;
    }

    for (;;) {
        // -- ray finished: resolve the winner among the 8 lanes.  The reference takes candidates in (packet visit
        //    order, ascending lane) and replaces on strict `<` (:118-136, :59): the winner is the earliest candidate
        //    that attains the minimum t.  Every lane kept its own earliest minimum; ties across lanes go to the
        //    smaller (visit sequence, lane).
        if (slot >= 0 && sp == 0 && pk == pk_end) {
            float tmin = group_min_t(tl);
            uint32_t key = (pkl != kNoPrim && tl == tmin) ? ((seql << 3) | static_cast<uint32_t>(li)) : 0xFFFFFFFFu;
            uint32_t kmin = group_min_u(key);
            // winner's lane, recomputed from the lane id where it is needed (ds_bpermute takes a byte address: lane * 4)
            const int wl4 = (((static_cast<int>(threadIdx.x) & 56) | static_cast<int>(kmin & 7u))) << 2;
            const float wu = __int_as_float(__builtin_amdgcn_ds_bpermute(wl4, __float_as_int(ul)));
            const float wv = __int_as_float(__builtin_amdgcn_ds_bpermute(wl4, __float_as_int(vl)));
            const uint32_t wp = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(wl4, static_cast<int>(pkl)));
            if (li == 0) {
                bool hit = kmin != 0xFFFFFFFFu;
                q[(HB + 0) * QS + slot] = hit ? tmin : FLT_MAX;
                q[(HB + 1) * QS + slot] = as_f(hit ? (wp * 8u + (kmin & 7u)) : kNoPrim);
                q[(HB + 2) * QS + slot] = wu;
                q[(HB + 3) * QS + slot] = wv;
            }
            slot = -1;
        }
        // -- idle groups pull the next rays of the queue (wave-uniform head, no atomics)
        uint64_t idle = __ballot(slot < 0);
        if (idle != 0) {
            // the mask of the groups below this one is needed once per ray: recomputed here (3 VALU) rather than kept in -- or spilled
            // from -- two registers for the whole walk
            int g_ = g;
            asm volatile("" : "+v"(g_));
            const uint64_t groups_below = (1ull << (g_ * 8)) - 1ull;
            uint64_t gm = idle & 0x0101010101010101ull;
            int mine = head + __popcll(gm & groups_below);
            head += __popcll(gm);
            // (ds_bpermute reads the source lane's register whether or not that lane is enabled; an out-of-range `mine` wraps)
            float fix = 0.0f, fiy = 0.0f, fiz = 0.0f;
            if (QS == 64) { fix = __shfl(pix, mine & 63); fiy = __shfl(piy, mine & 63); fiz = __shfl(piz, mine & 63); }
            if (slot < 0 && mine < nrays) {
                slot = mine;
                ox = q[0 * QS + mine]; oy = q[1 * QS + mine]; oz = q[2 * QS + mine];
                dx = q[3 * QS + mine]; dy = q[4 * QS + mine]; dz = q[5 * QS + mine];
                if (QS != 64) { fix = q[6 * QS + mine]; fiy = q[7 * QS + mine]; fiz = q[8 * QS + mine]; }
                ix = fix; iy = fiy; iz = fiz;
                best_t = FLT_MAX; tl = FLT_MAX; ul = 0; vl = 0; pkl = kNoPrim; seql = 0; seq = 0;
                pk = pk_end = 0;
                sp = 1;
                if (li == 0) stack[0] = make_uint2(root, as_u(-INFINITY));  // :28-32
            }
            if (__ballot(slot >= 0) == 0) break;
        }
        wave_lds_sync();
        // -- A: groups with no pending packet pop one stack entry (:39-62)
        if (slot >= 0 && pk == pk_end) {
            sp--;
            uint2 e = stack[sp];
            float node_t1 = as_f(e.y);
            if (!(node_t1 > best_t)) {  // :40-44
                uint32_t link = e.x;
                if ((link & 63u) == 0u) {  // device link (mp_internal.h): inner = index << 6
                    // InnerNode::intersect :149-162 ; lane li = child li.  Child boxes are stored decompressed
                    // (SURVEY A.4 box chain evaluated once on the host), so the slab test starts directly.
                    // (a uniform base + a 32-bit byte offset per lane: link = node << 6, a node is 256 bytes, a child record 32)
                    const float4* cp = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(nodes4) + ((link << 2) | (static_cast<uint32_t>(li) << 5)));
                    const float4 c0 = cp[0], c1 = cp[1];  // {min.xyz, max.x} {max.yz, link, -}
                    const uint32_t child = as_u(c1.z);
                    // aabb.rs:254-284
                    float ax = (c0.x - ox) * ix, ay = (c0.y - oy) * iy, az = (c0.z - oz) * iz;
                    float cx = (c0.w - ox) * ix, cy = (c1.x - oy) * iy, cz = (c1.y - oz) * iz;
                    // NaN (0 * inf) -> -inf on the min side, +inf on the max side (aabb.rs:262-267): maxNum / minNum return the
                    // other operand for a NaN and the value itself otherwise, one instruction each and no branch.  Only a ray with
                    // a zero direction component (infinite inverse) can produce one: queues without such a ray run the variant
                    // without the six patches (trace_wave)
                    if (PATCH_NAN) {
                        ax = fmaxf(ax, -INFINITY); ay = fmaxf(ay, -INFINITY); az = fmaxf(az, -INFINITY);
                        cx = fminf(cx, INFINITY); cy = fminf(cy, INFINITY); cz = fminf(cz, INFINITY);
                    }
                    const float t1 = fmaxf(fmaxf(fminf(ax, cx), 0.0f), fmaxf(fminf(ay, cy), fminf(az, cz)));
                    const float t2 = fminf(fminf(fmaxf(ax, cx), best_t), fminf(fmaxf(ay, cy), fmaxf(az, cz)));
                    // Null links are skipped at pop in the reference (:49).  The lane mask comes straight from the compare unit;
                    // lanes outside this branch contribute garbage bits to other groups' bytes, which are never looked at
                    const uint64_t okm = mask_le(t1, t2) & __builtin_amdgcn_uicmp(child, MP_LINK_NULL, 33);  // 33 = ICMP_NE
                    const bool ok = __builtin_amdgcn_inverse_ballot_w64(okm);
                    uint32_t m = static_cast<uint32_t>(okm >> (g * 8)) & 0xFFu;
                    const uint32_t below = BFE ? __builtin_amdgcn_ubfe(m, 0u, static_cast<uint32_t>(li)) : (m & lanes_below);
                    if (ok) stack[sp + __popc(below)] = make_uint2(child, as_u(t1));  // ascending lane :161
                    sp += __popc(m);
                } else {
                    pk = link >> 6;  // Leaf :56-62 ; device link = first packet << 6 | real triangles
                    pk_end = pk + (((link & 63u) + 7u) >> 3);
                }
            }
        }
        // -- B: one leaf packet per iteration (:104-140) ; lane li = triangle li
        if (slot >= 0 && pk < pk_end) {
            // 36-byte records: v0, e1, e2 ; a uniform base + a 32-bit byte offset per lane (fewer than 2^32 / 288 packets: upload_scene)
            const uint32_t toff = ((pk * 9u) << 5) + static_cast<uint32_t>(li) * 36u;
            const float* tp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(tris) + toff);
            const float v0x = tp[0], v0y = tp[1], v0z = tp[2], e1x = tp[3], e1y = tp[4], e1z = tp[5], e2x = tp[6], e2y = tp[7], e2z = tp[8];
            // triangle.rs:183-217
            float hx = fms(dy, e2z, dz * e2y), hy = fms(dz, e2x, dx * e2z), hz = fms(dx, e2y, dy * e2x);
            float det = fma_dot(e1x, e1y, e1z, hx, hy, hz);
            float inv_det = 1.0f / det;
            float sx = ox - v0x, sy = oy - v0y, sz = oz - v0z;
            float u = inv_det * fma_dot(sx, sy, sz, hx, hy, hz);
            float qx = fms(sy, e1z, sz * e1y), qy = fms(sz, e1x, sx * e1z), qz = fms(sx, e1y, sy * e1x);
            float v = inv_det * fma_dot(dx, dy, dz, qx, qy, qz);
            float t = inv_det * fma_dot(e2x, e2y, e2z, qx, qy, qz);
            bool valid = (u >= 0.0f) & (v >= 0.0f) & ((u + v) <= 1.0f) & (t >= 0.0f) & (t <= best_t);  // :125
            if (valid && t < tl) { tl = t; ul = u; vl = v; pkl = pk; seql = seq; }
            seq++;
            pk++;
            if (pk == pk_end) best_t = group_min_t(tl);  // `if hit.t < best.t { best = hit }` :59-61
        }
    }
    wave_lds_sync();
}

// A queue without a ray that has an infinite inverse direction component (a zero -- or, denormals being kept, a tiny --
// direction component: geometry/mod.rs:49-53) cannot produce 0 * inf in the slab test: it runs the variant without the NaN
// patches (six VALU per node step less) on the wide tree.  Wave-uniform choice per call.
template <bool BFE = false>
__device__ __forceinline__ void trace_wave(const DevScene& sc, float* __restrict__ q, uint2* __restrict__ stack_base, int nrays) {
    const int lane = static_cast<int>(threadIdx.x) & 63;
    const float qx = q[3 * 64 + lane], qy = q[4 * 64 + lane], qz = q[5 * 64 + lane];
    const float ix = (qx == 0.0f) ? INFINITY : 1.0f / qx, iy = (qy == 0.0f) ? INFINITY : 1.0f / qy, iz = (qz == 0.0f) ? INFINITY : 1.0f / qz;
    const bool inf = lane < nrays && (fabsf(ix) == INFINITY || fabsf(iy) == INFINITY || fabsf(iz) == INFINITY);
    if (__ballot(inf) != 0) trace_wave_impl<true, BFE>(sc, q, stack_base, nrays);
    else trace_wave_impl<false, BFE>(sc, q, stack_base, nrays);
}

// ... and for the pooled path kernel's queue of QS > 64 slots in global memory (inverse directions in rows 6..8)
template <int QS>
__device__ __forceinline__ void trace_wave_pool(const DevScene& sc, float* __restrict__ q, uint2* __restrict__ stack_base, int nrays) {
    int lane = static_cast<int>(threadIdx.x) & 63;
    asm volatile("" : "+v"(lane));  // (re-derived per call: the twelve row addresses below are not worth registers -- or spill slots -- across the kernel)
    bool inf = false;
#pragma unroll
    for (int k = 0; k < QS / 64; k++) {
        const int slot = lane + 64 * k;
        if (slot < nrays)
            inf = inf || fabsf(q[6 * QS + slot]) == INFINITY || fabsf(q[7 * QS + slot]) == INFINITY || fabsf(q[8 * QS + slot]) == INFINITY;
    }
    if (__ballot(inf) != 0) trace_wave_impl<true, true, QS>(sc, q, stack_base, nrays);
    else trace_wave_impl<false, true, QS>(sc, q, stack_base, nrays);
}

// Exact conservative pre-test against the union of the root node's child boxes (DevScene::pre_min/pre_max): every
// child box C lies inside that union U, and the slab interval of C is contained in the slab interval of U in
// floating point too (IEEE subtraction and multiplication are monotone, and the NaN patches of aabb.rs:262-267 map
// to the containing infinities), so a ray whose U interval is empty pushes no child of the root and is a miss.
__device__ __forceinline__ bool may_hit_scene(const DevScene& sc, const Ray& r) {
    if (!sc.has_pre) return true;
    float ax = (sc.pre_min[0] - r.ox) * r.ix, ay = (sc.pre_min[1] - r.oy) * r.iy, az = (sc.pre_min[2] - r.oz) * r.iz;
    float cx = (sc.pre_max[0] - r.ox) * r.ix, cy = (sc.pre_max[1] - r.oy) * r.iy, cz = (sc.pre_max[2] - r.oz) * r.iz;
    ax = (ax != ax) ? -INFINITY : ax; ay = (ay != ay) ? -INFINITY : ay; az = (az != az) ? -INFINITY : az;
    cx = (cx != cx) ? INFINITY : cx; cy = (cy != cy) ? INFINITY : cy; cz = (cz != cz) ? INFINITY : cz;
    float lox = fminf(ax, cx), loy = fminf(ay, cy), loz = fminf(az, cz);
    float hix = fmaxf(ax, cx), hiy = fmaxf(ay, cy), hiz = fmaxf(az, cz);
    float t1 = fmaxf(fmaxf(lox, 0.0f), fmaxf(loy, loz));
    float t2 = fminf(fminf(hix, FLT_MAX), fminf(hiy, hiz));
    return t1 <= t2;
}

// Hit resolve + shade: tail of intersect (ray_bvh_intersection.rs:66-95) and render_sample (worker.rs:59-65).
// Returns |dot(ray.direction, normal)|.
// Returns TriangleShadingData.material of the triangle (mod.rs:44; 0 for everything the reference builds).
__device__ __forceinline__ uint32_t resolve_normal(const DevScene& sc, uint32_t prim, float u, float v, float n[3]) {
    const float4* sh = reinterpret_cast<const float4*>(sc.shade) + static_cast<size_t>(prim) * 3;
    float4 a = sh[0], b = sh[1], c = sh[2];
    float nx, ny, nz;
    if (as_u(c.y) != 0u) {
        // flat: Triangle::normal (triangle.rs:141-144), unfused cross of the decompressed edges
        const float* tp = sc.tris_aos + static_cast<size_t>(prim) * kTriDwords;
        float e1x = tp[3], e1y = tp[4], e1z = tp[5], e2x = tp[6], e2y = tp[7], e2z = tp[8];
        nx = e1y * e2z - e1z * e2y;
        ny = e1z * e2x - e1x * e2z;
        nz = e1x * e2y - e1y * e2x;
    } else {
        float w = 1.0f - u - v;  // triangle.rs:235-236
        nx = a.x * w + a.w * u + b.z * v;
        ny = a.y * w + b.x * u + b.w * v;
        nz = a.z * w + b.y * u + c.x * v;
    }
    float len = sqrtf(nx * nx + ny * ny + nz * nz);
    n[0] = nx / len; n[1] = ny / len; n[2] = nz / len;
    return as_u(c.z);
}

// impl Object for Sphere::intersect, scene/primitives.rs:16-48 (lane-parallel; n = unit normal at the hit)
__device__ __forceinline__ bool sphere_intersect(const DevScene& sc, const Ray& r, float& t, float n[3]) {
    const float ocx = r.ox - sc.sphere_center[0], ocy = r.oy - sc.sphere_center[1], ocz = r.oz - sc.sphere_center[2];
    const float b = ocx * r.dx + ocy * r.dy + ocz * r.dz;
    const float c = (ocx * ocx + ocy * ocy + ocz * ocz) - sc.sphere_radius * sc.sphere_radius;
    const float disc = b * b - c;
    if (disc < 0.0f) return false;
    const float sq = sqrtf(disc);
    const float t1 = -b - sq, t2 = -b + sq;
    if (t1 > 0.0f) t = t1;
    else if (t2 > 0.0f) t = t2;
    else return false;
    const float px = r.ox + r.dx * t, py = r.oy + r.dy * t, pz = r.oz + r.dz * t;
    const float nx = px - sc.sphere_center[0], ny = py - sc.sphere_center[1], nz = pz - sc.sphere_center[2];
    const float len = sqrtf(nx * nx + ny * ny + nz * nz);
    n[0] = nx / len; n[1] = ny / len; n[2] = nz / len;
    return true;
}

// Build-defined object group (mp_scene_group / mp_scene_instances): the scene seen as member k -- the member's own traversal
// arrays under the group's material table and stack bound.  k is wave-uniform in the pass over member k (scalar loads) and per
// lane when a lane shades the member its ray hit.
__device__ __forceinline__ void object_scene(const DevScene& sc, uint32_t k, DevScene& sk) {
    sk = sc;
    const DevObject& o = sc.objects[k];
    sk.shade = o.shade; sk.nodes_aos = o.nodes_aos; sk.nodes_lit = o.nodes_lit; sk.tris_aos = o.tris_aos; sk.vidx = o.vidx; sk.vtex = o.vtex;
    sk.root = o.root; sk.root_lit = o.root_lit; sk.has_pre = o.has_pre;
    sk.kind = o.kind; sk.sphere_radius = o.sphere_radius;
    for (int i = 0; i < 3; i++) { sk.pre_min[i] = o.pre_min[i]; sk.pre_max[i] = o.pre_max[i]; sk.sphere_center[i] = o.sphere_center[i]; }
}
// UnitQuaternion * Vector3 as the oracle (and nalgebra) evaluate it: t = 2 * (qv x v); v' = t * w + (qv x t) + v, unfused
__device__ __forceinline__ void quat_rotate(float qi, float qj, float qk, float qw, float vx, float vy, float vz, float& ox,
                                            float& oy, float& oz) {
    const float tx = (qj * vz - qk * vy) * 2.0f, ty = (qk * vx - qi * vz) * 2.0f, tz = (qi * vy - qj * vx) * 2.0f;
    const float cx = qj * tz - qk * ty, cy = qk * tx - qi * tz, cz = qi * ty - qj * tx;
    ox = tx * qw + cx + vx; oy = ty * qw + cy + vy; oz = tz * qw + cz + vz;
}
// ... and the ray in member k's frame: origin - translation, then (rotated members) origin and direction through the inverse
// rotation; the direction is not re-normalised (t stays the world ray's), inv_direction by Ray::new's rule (geometry/mod.rs:49-53)
__device__ __forceinline__ void object_ray(const DevScene& sc, uint32_t k, const Ray& r, Ray& rk) {
    const DevObject& o = sc.objects[k];
    rk = r;
    rk.ox = r.ox - o.t[0]; rk.oy = r.oy - o.t[1]; rk.oz = r.oz - o.t[2];
    if (o.rotated) {
        const float qi = -o.q[0], qj = -o.q[1], qk = -o.q[2], qw = o.q[3];
        const float vx = rk.ox, vy = rk.oy, vz = rk.oz;
        quat_rotate(qi, qj, qk, qw, vx, vy, vz, rk.ox, rk.oy, rk.oz);
        quat_rotate(qi, qj, qk, qw, r.dx, r.dy, r.dz, rk.dx, rk.dy, rk.dz);
        rk.ix = (rk.dx == 0.0f) ? INFINITY : 1.0f / rk.dx;
        rk.iy = (rk.dy == 0.0f) ? INFINITY : 1.0f / rk.dy;
        rk.iz = (rk.dz == 0.0f) ? INFINITY : 1.0f / rk.dz;
    }
}

struct GroupHit {
    float t, u, v;
    uint32_t prim, inst;
};

// Closest hit of the wave's rays (one per lane, `act` = lane holds a ray) with the group walk: compaction of the lanes whose ray
// can reach the object into the wave's ray queue, trace_wave, results back to the lanes.  OBJ: once per member of the object
// group, in order, closest wins with a strict `<` (the first member keeps ties).
template <bool OBJ, bool BFE = false>
__device__ __forceinline__ void trace_objects(const DevScene& sc, const Ray& r, bool act, float* __restrict__ q,
                                              uint2* __restrict__ stack, GroupHit& h) {
    h.t = FLT_MAX; h.u = h.v = 0.0f; h.prim = kNoPrim; h.inst = 0u;
    auto pass = [&](const DevScene& sk, const Ray& rk, uint32_t k) {
        if (OBJ && sk.kind == 1u) {  // a Sphere member (scene/primitives.rs:16-48): lane-parallel, no walk; prim 0
            float ts, nn[3];
            if (act && sphere_intersect(sk, rk, ts, nn) && ts < h.t) { h.t = ts; h.prim = 0u; h.u = h.v = 0.0f; h.inst = k; }
            return;
        }
        const bool queued = act && may_hit_scene(sk, rk);
        const uint64_t am = __ballot(queued);
        const int n = __popcll(am), rank = rank_below(am);
        if (queued) {
            q[0 * 64 + rank] = rk.ox; q[1 * 64 + rank] = rk.oy; q[2 * 64 + rank] = rk.oz;
            q[3 * 64 + rank] = rk.dx; q[4 * 64 + rank] = rk.dy; q[5 * 64 + rank] = rk.dz;
        }
        wave_lds_sync();
        trace_wave<BFE>(sk, q, stack, n);
        if (queued) {
            const uint32_t prim = as_u(q[1 * 64 + rank]);
            const float t = q[0 * 64 + rank];
            if (prim != kNoPrim && t < h.t) { h.t = t; h.prim = prim; h.u = q[2 * 64 + rank]; h.v = q[3 * 64 + rank]; h.inst = k; }
        }
        wave_lds_sync();
    };
    if (!OBJ) {
        pass(sc, r, 0u);
        return;
    }
    for (uint32_t k = 0; k < sc.inst_count; k++) {
        DevScene sk;
        Ray rk;
        object_scene(sc, k, sk);
        object_ray(sc, k, r, rk);
        pass(sk, rk, k);
    }
}

// Normal and material id of a hit on member `inst` of an object group (r = the world ray).  A Sphere member's normal is that of
// its intersect, evaluated again on the ray in the member's frame (same operations, same bits); material 0 (primitives.rs:40-46).
__device__ __forceinline__ uint32_t object_normal(const DevScene& sc, uint32_t inst, const Ray& r, uint32_t prim, float u, float v,
                                                  float n[3]) {
    DevScene so;
    object_scene(sc, inst, so);
    uint32_t mat = 0u;
    if (so.kind == 1u) {
        Ray rk;
        object_ray(sc, inst, r, rk);
        float t;
        sphere_intersect(so, rk, t, n);
    } else {
        mat = resolve_normal(so, prim, u, v, n);
    }
    const DevObject& o = sc.objects[inst];
    if (o.rotated) quat_rotate(o.q[0], o.q[1], o.q[2], o.q[3], n[0], n[1], n[2], n[0], n[1], n[2]);  // back into the world frame
    return mat;
}

// ---- fused tile render: Worker::render_tile (worker.rs:32-49) for a list of tiles -----------------------------
struct RenderParams {
    DevScene scene;
    RayGen gen;
    const mp_block* tiles;
    uint32_t n_tiles, tile_size;
    float* out;           // tile-major RGBA f32
    uint32_t* counter;    // work-queue head
    float inv_spp;        // 1.0 / spp as f32 (worker.rs:44)
    uint32_t s_begin, s_end;  // samples of this launch (progressive accumulation: a sub-range of [0, spp))
    uint32_t carry_in;    // out holds the running sums / hit counts of samples [0, s_begin)
    uint32_t finalize;    // write the means (worker.rs:44); otherwise the running sums
    uint32_t chunked;     // MP_FLAG_CHUNKED_SUM: f32 sums over 256-sample chunks, f64 total kept in the tile buffer
    const uint32_t* tile_order;      // optional hand-out order: work slot k renders tile tile_order[k] (into that tile's own slot)
    unsigned long long* tile_cost;   // optional: += shader-clock cycles the waves spent on each tile
    uint32_t lds_per_wave;
    uint32_t max_depth;   // path extension only
    unsigned long long* segments;  // path extension: ray segments traced (Object::intersect calls), may be null
    float* pool;           // pooled path kernel: per-wave workspace (ray queue + parked path state), pool_stride floats per wave
    uint32_t pool_stride;
};

// worker.rs:40-44 with the running state of a pixel carried across launches (MP_FLAG_ACCUMULATE): rgb = sequential sample sum,
// a = hit count; the launch that draws the last sample writes the means.  Every lane of the pixel loads the same state.
//
// MP_FLAG_CHUNKED_SUM (build-defined, include/minipath_hip.h): the pixel's slot holds {x: f32 sum of the current 256-sample
// chunk, y: hit count, (z,w): f64 total of the finished chunks}.  The total stays in memory (one read-modify-write per pixel and
// chunk by the pixel's first lane), so the rule costs the render kernels no registers.
constexpr uint32_t kSumChunk = 256;
__device__ __forceinline__ void pixel_state_load(const RenderParams& P, size_t off, bool inpix, bool writer, float& acc, float& cnt) {
    acc = 0.0f;
    cnt = 0.0f;
    if (P.carry_in && inpix) {
        const float4 prev = *reinterpret_cast<const float4*>(P.out + off);
        acc = prev.x;
        cnt = P.chunked ? prev.y : prev.w;
    } else if (P.chunked && inpix && writer) {
        *reinterpret_cast<double*>(P.out + off + 2) = 0.0;
    }
}
// end of a chunk: total += (f64)chunk sum; the next chunk starts from +0
__device__ __forceinline__ void chunk_flush(const RenderParams& P, size_t off, bool writer, float& acc) {
    if (writer) {
        double* t = reinterpret_cast<double*>(P.out + off + 2);
        *t = *t + static_cast<double>(acc);
    }
    acc = 0.0f;
}
// `s_next` = index of the first sample NOT yet added (the launch's s_end)
__device__ __forceinline__ void pixel_state_store(const RenderParams& P, size_t off, float acc, float cnt) {
    if (P.chunked) {
        if (!P.finalize) {
            *reinterpret_cast<float2*>(P.out + off) = make_float2(acc, cnt);
            return;
        }
        double t = *reinterpret_cast<const double*>(P.out + off + 2);
        if ((P.s_end & (kSumChunk - 1u)) != 0u) t = t + static_cast<double>(acc);  // the last, partial chunk
        const double inv = 1.0 / static_cast<double>(P.gen.spp);
        const float m = static_cast<float>(t * inv), a = static_cast<float>(static_cast<double>(cnt) * inv);
        *reinterpret_cast<float4*>(P.out + off) = make_float4(m, m, m, a);
        return;
    }
    const float m = P.finalize ? acc * P.inv_spp : acc;  // worker.rs:44
    const float a = P.finalize ? cnt * P.inv_spp : cnt;
    *reinterpret_cast<float4*>(P.out + off) = make_float4(m, m, m, a);
}

// The render kernels take RenderParams by value (328 bytes of kernel arguments) but read it through the kernarg segment pointer,
// one short-lived VIEW per phase (unit setup, ray generation, walk, shading, accumulation): a compiler barrier on the pointer
// ends the life of everything loaded through the previous view, so a field costs an s_load from the constant cache where it is
// used instead of an SGPR for the whole kernel -- the walks need the scalar registers for their record double-buffers, and what
// did not fit was spilled to VGPR lanes and scratch (round 2: 78 SGPR + 4 VGPR spills in the packet kernel, 67 + 29 and 104 B of
// scratch in the path kernel).
typedef const __attribute__((address_space(4))) RenderParams* kparams_t;
template <class T>
__device__ __forceinline__ const T& kernarg_view(const __attribute__((address_space(4))) T* kp) {
    asm volatile("" : "+s"(kp));
    return *(const T*)kp;
}
__device__ __forceinline__ const RenderParams& params_view(kparams_t kp) { return kernarg_view<RenderParams>(kp); }
#define MP_KERNEL_PARAMS kparams_t KP = (kparams_t)__builtin_amdgcn_kernarg_segment_ptr()

// get_next_tile (machinery.rs:206-208) for wavefronts: kWorkQueues interleaved queues (mp_internal.h), home queue = this
// workgroup's XCD, the others once it is empty.  Wave-uniform; `state` = queue | consecutive dry queues << 8 (one SGPR: the
// packet kernel has none to spare).  Used as:  for (;;) { MP_NEXT_UNIT(unit); ... }
#define MP_NEXT_UNIT(unit)                                                                              \
    uint32_t unit;                                                                                      \
    {                                                                                                   \
        const uint32_t q_ = qstate & 0xFFu;                                                             \
        uint32_t idx_ = 0;                                                                              \
        if (lane == 0) idx_ = atomicAdd(P.counter + q_ * kWorkQueueStride, 1u);                         \
        idx_ = __builtin_amdgcn_readfirstlane(idx_);                                                    \
        unit = idx_ * kWorkQueues + q_;                                                                 \
        if (unit >= total || idx_ >= 0x10000000u) {                                                     \
            if ((qstate >> 8) + 1u == kWorkQueues) break;                                               \
            qstate = ((q_ + 1u) % kWorkQueues) | (((qstate >> 8) + 1u) << 8);                           \
            continue;                                                                                   \
        }                                                                                               \
        qstate = q_;                                                                                    \
    }

// S = samples of one pixel in flight in a wavefront (64/S pixels x S consecutive samples per pass).
template <int S, bool OBJ>
__global__ __launch_bounds__(256) void render_tiles_kernel(RenderParams) {
    extern __shared__ __align__(16) unsigned char smem[];
    MP_KERNEL_PARAMS;
    const RenderParams& P0 = params_view(KP);
    constexpr int BW = (S == 1) ? 8 : (S == 2) ? 8 : (S == 4) ? 4 : 4;
    constexpr int BH = 64 / S / BW;
    const int lane = static_cast<int>(threadIdx.x) & 63, wave = static_cast<int>(threadIdx.x) >> 6;
    float* q = reinterpret_cast<float*>(smem + static_cast<size_t>(wave) * P0.lds_per_wave);
    uint2* stack = reinterpret_cast<uint2*>(q + kQueueFloats);
    const uint32_t ts = P0.tile_size;
    const uint32_t bx = (ts + BW - 1) / BW, by = (ts + BH - 1) / BH, upt = bx * by;
    const uint32_t total = P0.n_tiles * upt;
    const int pix = lane / S, sub = lane % S;

    uint32_t qstate = blockIdx.x % kWorkQueues;
    for (;;) {
        const RenderParams& P = params_view(KP);  // one view per work unit
        MP_NEXT_UNIT(unit)
        const uint32_t b = unit % upt;
        const uint32_t tile_i = P.tile_order ? P.tile_order[unit / upt] : unit / upt;
        const uint64_t t_unit = P.tile_cost ? __builtin_readcyclecounter() : 0;
        const mp_block T = P.tiles[tile_i];
        const uint32_t px = T.min_x + (b % bx) * BW + static_cast<uint32_t>(pix % BW);
        const uint32_t py = T.min_y + (b / bx) * BH + static_cast<uint32_t>(pix / BW);
        const bool inpix = px < T.max_x && py < T.max_y;
        if (__ballot(inpix) == 0) continue;
        const size_t off = (static_cast<size_t>(tile_i) * ts * ts + static_cast<size_t>(py - T.min_y) * ts + (px - T.min_x)) * 4;
        float acc, cnt;  // pixel_sum (r=g=b) and alpha (worker.rs:40)
        pixel_state_load(P, off, inpix, sub == 0, acc, cnt);
        // passes are aligned to multiples of S in the absolute sample index, so that a chunk boundary (MP_FLAG_CHUNKED_SUM) never
        // falls inside a pass; lanes outside [s_begin, s_end) add +0.0, which is exact
        for (uint32_t s0 = P.s_begin & ~static_cast<uint32_t>(S - 1); s0 < P.s_end; s0 += S) {
            const uint32_t s = s0 + static_cast<uint32_t>(sub);
            const bool act = inpix && s >= P.s_begin && s < P.s_end;
            Ray r;
            r.dx = r.dy = r.dz = 0.0f;
            if (act) sample_ray(P.gen, px, py, s, r);
            if (P.scene.kind == 1u) {  // Scene<Sphere>: no traversal, everything is lane-parallel
                float c = 0.0f, h = 0.0f, ts_, nn[3];
                if (act && sphere_intersect(P.scene, r, ts_, nn)) { c = fabsf(r.dx * nn[0] + r.dy * nn[1] + r.dz * nn[2]); h = 1.0f; }
#pragma unroll
                for (int j = 0; j < S; j++) {
                    acc += __shfl(c, (lane & ~(S - 1)) + j);
                    cnt += __shfl(h, (lane & ~(S - 1)) + j);
                }
                if (P.chunked && ((s0 + S) & (kSumChunk - 1u)) == 0u && s0 + S <= P.s_end) chunk_flush(P, off, inpix && sub == 0, acc);
                continue;
            }
            // group walk (once per member of an object group): closest hit, then the shading of worker.rs:59-65
            GroupHit gh;
            trace_objects<OBJ>(P.scene, r, act, q, stack, gh);
            float c = 0.0f, h = 0.0f;
            if (gh.prim != kNoPrim) {
                float nn[3];
                if (OBJ) object_normal(P.scene, gh.inst, r, gh.prim, gh.u, gh.v, nn);
                else resolve_normal(P.scene, gh.prim, gh.u, gh.v, nn);
                c = fabsf(r.dx * nn[0] + r.dy * nn[1] + r.dz * nn[2]);  // worker.rs:60
                h = 1.0f;
            }
            // pixel_sum += sample, strictly in sample order (worker.rs:41-43); inactive samples add +0.0 (exact)
#pragma unroll
            for (int j = 0; j < S; j++) {
                acc += __shfl(c, (lane & ~(S - 1)) + j);
                cnt += __shfl(h, (lane & ~(S - 1)) + j);
            }
            if (P.chunked && ((s0 + S) & (kSumChunk - 1u)) == 0u && s0 + S <= P.s_end) chunk_flush(P, off, inpix && sub == 0, acc);
        }
        if (inpix && sub == 0) pixel_state_store(P, off, acc, cnt);
        if (P.tile_cost && lane == 0) atomicAdd(P.tile_cost + tile_i, static_cast<unsigned long long>(__builtin_readcyclecounter() - t_unit));
    }
}


// ---- ray-packet traversal (coherent rays) --------------------------------------------------------------------
// One ray per lane, 64 rays share ONE traversal: the wavefront walks the BVH in the reference's canonical order
// (children pushed in ascending index, popped in descending, ray_bvh_intersection.rs:52-54,161 -- the reference
// never orders children by distance, so every ray's own visit sequence is a subsequence of this walk) and each lane
// applies its own ray's tests: the pop-time cull `node_t1 > best.t` (:40), the slab test against its own best.t
// (:149-162) and the triangle acceptance (:118-136).  A node / leaf is visited when at least one lane still needs
// it; lanes that the reference would not take there are masked.  Node and triangle data are wave-uniform and come
// through the scalar unit (s_load into SGPRs) from AoS copies of the scene; the per-lane entry distances of the
// shared stack live in LDS.
typedef const __attribute__((address_space(4))) float* kfp;      // constant address space => scalar loads
typedef const __attribute__((address_space(4))) uint32_t* kup;
typedef float krec8 __attribute__((ext_vector_type(8)));      // one 32-byte child record

constexpr float kTiny = 9.094947017729282e-13f;  // 2^-40
constexpr float kHuge = 1099511627776.0f;        // 2^40

// Early-out argument of the triangle tests (trace_packet_impl): a numerator whose sign differs from det's, with |num| >= 2^-40 and
// |det| <= 2^40, gives a quotient fl(fl(1/det) * num) that is certainly a non-zero negative number (|quotient| >= 2^-80(1-eps): no
// underflow to -0, which would pass `>= 0`).  det = +-0 or subnormal gives an infinite reciprocal whose sign still follows det.
// Flipping num's sign by det's turns "signs differ and |num| >= 2^-40" into one ordered compare against -2^-40; NaN never rejects.

// OCT >= 0: every ray of the wave has a finite inverse direction with the sign pattern OCT (bit 0: x < 0, bit 1: y < 0,
// bit 2: z < 0) and the box is ordered (min <= max, checked at upload).  IEEE subtraction and multiplication by a constant are
// monotone, so (bmin - o) * inv <= (bmax - o) * inv for inv > 0 and >= for inv < 0, and no 0 * inf can arise: the min / max of
// aabb.rs:269-271 select the near / far plane the sign names, and the six min/max drop out of the compiled code.
template <bool PATCH_NAN, int OCT>
__device__ __forceinline__ void slab(float bnx, float bny, float bnz, float bxx, float bxy, float bxz, const Ray& r,
                                     float limit, float& t1, float& t2) {
    // aabb.rs:254-284
    float ax = (bnx - r.ox) * r.ix, ay = (bny - r.oy) * r.iy, az = (bnz - r.oz) * r.iz;
    float cx = (bxx - r.ox) * r.ix, cy = (bxy - r.oy) * r.iy, cz = (bxz - r.oz) * r.iz;
    if (PATCH_NAN) {  // only rays with an infinite inverse direction component can produce 0*inf; maxNum/minNum drop a NaN operand
        ax = fmaxf(ax, -INFINITY); ay = fmaxf(ay, -INFINITY); az = fmaxf(az, -INFINITY);
        cx = fminf(cx, INFINITY); cy = fminf(cy, INFINITY); cz = fminf(cz, INFINITY);
    }
    float lox, loy, loz, hix, hiy, hiz;
    if (OCT >= 0) {
        lox = (OCT & 1) ? cx : ax; hix = (OCT & 1) ? ax : cx;
        loy = (OCT & 2) ? cy : ay; hiy = (OCT & 2) ? ay : cy;
        loz = (OCT & 4) ? cz : az; hiz = (OCT & 4) ? az : cz;
    } else {
        lox = fminf(ax, cx); loy = fminf(ay, cy); loz = fminf(az, cz);
        hix = fmaxf(ax, cx); hiy = fmaxf(ay, cy); hiz = fmaxf(az, cz);
    }
    t1 = fmaxf(fmaxf(lox, 0.0f), fmaxf(loy, loz));
    t2 = fminf(fminf(hix, limit), fminf(hiy, hiz));
}

struct PacketHit {
    float t, u, v;
    uint32_t prim;
};

// The shared traversal stack is wave-uniform and lives in four VGPRs used as 64-entry arrays (entry k = lane k, written with
// v_writelane, read with v_readlane, both with a scalar index and both regardless of EXEC): link, source slot (node*8+child, to
// re-derive the entry distance) and the 64-bit mask of lanes that passed the slab test when the entry was pushed.
// The pop-time cull `node_t1 > best.t` (:40) needs the ray's own entry distance t1: it can only fire for a lane whose best.t
// shrank after the push, so while no lane accepted a hit since then (entry not "stale") the pushed mask is the answer; otherwise
// t1 is recomputed from the child's box (same operations, same bits as at push time).  Staleness needs no per-entry state: the
// entries that predate the last change of any best.t are exactly those below a low-water mark of the stack pointer
// (`stale_top` in trace_packet_impl: set to sp when a leaf changed a best.t, lowered by every pop).
constexpr uint32_t kSrcRoot = 0xFFFFFFFFu;

template <bool PATCH_NAN, int OCT>
__device__ __forceinline__ float slab_entry(float bnx, float bny, float bnz, float bxx, float bxy, float bxz, const Ray& r) {
    float t1, t2;
    slab<PATCH_NAN, OCT>(bnx, bny, bnz, bxx, bxy, bxz, r, FLT_MAX, t1, t2);
    return t1;
}

// Wave-uniform stack in registers: four VGPRs used as 64-entry arrays.
struct RegStack {
    int link, src, mlo, mhi;
    __device__ __forceinline__ RegStack(float*, int) : link(0), src(0), mlo(0), mhi(0) {}
    // v_writelane_b32 ignores EXEC: a push made while only the rays of the current node are enabled still lands in lane `sp`
    // (gfx9 allows one SGPR operand per VALU instruction: the lane select goes through M0, which nothing else in these kernels uses)
    __device__ __forceinline__ void push(int sp, uint32_t l, uint32_t s, uint64_t m) {
        asm volatile(
            "s_mov_b32 m0, %4\n\ts_nop 0\n\t"
            "v_writelane_b32 %0, %5, m0\n\tv_writelane_b32 %1, %6, m0\n\tv_writelane_b32 %2, %7, m0\n\tv_writelane_b32 %3, %8, m0"
            : "+v"(link), "+v"(src), "+v"(mlo), "+v"(mhi)
            : "s"(sp), "s"(l), "s"(s), "s"(static_cast<uint32_t>(m)), "s"(static_cast<uint32_t>(m >> 32))
            : "m0");
    }
    __device__ __forceinline__ void pop(int sp, uint32_t& l, uint32_t& s, uint64_t& m) const {
        l = static_cast<uint32_t>(__builtin_amdgcn_readlane(link, sp));
        s = static_cast<uint32_t>(__builtin_amdgcn_readlane(src, sp));
        m = static_cast<uint32_t>(__builtin_amdgcn_readlane(mlo, sp)) |
            (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(mhi, sp))) << 32);
    }
    __device__ __forceinline__ void sync(int) {}
};

// Trees deeper than 9 levels (7*depth+1 > 64) can in principle need more entries: those above 63 go to LDS (one uint4
// {link, src, mask} per entry, written by lane 0, read back as a broadcast).  Real walks rarely get there
// (the 28-level atrium tree peaks at 16 entries per ray), so the register path stays the common one.
struct HybridStack {
    RegStack reg;
    uint4* ent;
    int lane;
    int nreg;  // entries kept in registers (<= 64; lowered only by the mp_ctx_set_option test knob)
    __device__ __forceinline__ HybridStack(float* lds, int lane_, uint32_t, uint32_t nreg_)
        : reg(nullptr, lane_), ent(reinterpret_cast<uint4*>(lds)), lane(lane_), nreg(static_cast<int>(nreg_)) {}
    __device__ __forceinline__ void push(int sp, uint32_t l, uint32_t s, uint64_t m) {
        if (sp < nreg) {
            reg.push(sp, l, s, m);
        } else if (lane == 0) {
            ent[sp - nreg] = make_uint4(l, s, static_cast<uint32_t>(m), static_cast<uint32_t>(m >> 32));
        }
    }
    __device__ __forceinline__ void pop(int sp, uint32_t& l, uint32_t& s, uint64_t& m) const {
        if (sp < nreg) {
            reg.pop(sp, l, s, m);
        } else {
            const uint4 e = ent[sp - nreg];
            l = __builtin_amdgcn_readfirstlane(e.x);
            s = __builtin_amdgcn_readfirstlane(e.y);
            m = __builtin_amdgcn_readfirstlane(e.z) | (static_cast<uint64_t>(__builtin_amdgcn_readfirstlane(e.w)) << 32);
        }
    }
    __device__ __forceinline__ void sync(int sp) { if (sp > nreg) wave_lds_sync(); }
};

__device__ __forceinline__ uint32_t uniform_u(float f) { return __builtin_amdgcn_readfirstlane(as_u(f)); }
// a wave-uniform 32-bit value the compiler must not fold into 64-bit address arithmetic (it would leave the scalar unit: v_mad_u64_u32)
__device__ __forceinline__ uint32_t scalar_u(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
// record i of a leaf whose first record is tp: base + a 32-bit byte offset (s_load's register-offset form: one s_mul_i32)
__device__ __forceinline__ kfp tri_record(kfp tp, uint32_t i) {
    return reinterpret_cast<kfp>(reinterpret_cast<const __attribute__((address_space(4))) char*>(tp) + scalar_u(i * static_cast<uint32_t>(kTriDwords * 4)));
}

#ifdef MP_PROF_MISSES  // profiling builds only (tools/build_variant.sh): slow-path calls of the mask cache, read by mp_prof_read
__device__ unsigned long long g_prof[4];
#define MP_PROF_COUNT(i) do { if ((threadIdx.x & 63u) == 0u) atomicAdd(&g_prof[i], 1ull); } while (0)
#else
#define MP_PROF_COUNT(i) do { } while (0)
#endif
// ---- packet-level child rejection with a per-unit mask cache (round 3) ----------------------------------------------------------
// Counted on the metric's frame (tools/sim_collapse.py --packet-studies; profiles/r03_notes.md): a 64-ray camera packet tests 110
// child boxes per pass and pushes 19; for 90 of the 91 others NO ray of the packet can pass, and a conservative test on bounds of
// the packet -- componentwise min / max of the origins and inverse directions -- proves it.  Evaluating that test per node visit
// (lane j = child j, the records through a vector load) was built first and ran slower than it saved (profiles/r03_notes.md).
// What pays is doing it ONCE PER WORK UNIT: a unit shoots 8-16 passes through the same 2-4 pixels, and one set of bounds B that
// contains every pass's rays gives one 8-bit "children that may be hit" mask per node, cached in LDS (the packet kernel uses no
// other LDS).  Every pass checks, lane by lane, that its ray lies inside B (no reduction), and while that holds and the sign
// pattern is the same, a node visit costs one LDS lookup and the exact per-ray slab tests of the surviving children only.  B starts
// as the bounds of the first pass (wave reductions) widened by MP_MCACHE_PAD of their extent; a pass that does not fit widens B and
// clears the cache.
// Why skipping a child whose bit is clear is exact.  Only in the sign-specialised walks (OCT >= 0: every active ray has finite
// inverse directions of one sign pattern) and only if every active origin and inverse component is finite.  Axis with inv > 0
// (aabb.rs:257-271 gives lo = fl(fl(bmin - o) * inv), hi = fl(fl(bmax - o) * inv)): with omax >= o for every ray,
// y = fl(bmin - omax) <= fl(bmin - o) (IEEE subtraction is monotone), so lo >= fl(y * inv) (multiplication by a positive number is
// monotone) >= min(fl(y * imin), fl(y * imax)) =: L (x -> fl(y * x) is monotone on [imin, imax]: its minimum sits at an end);
// likewise hi <= max(fl(z * imin), fl(z * imax)) =: U with z = fl(bmax - omin).  Axis with inv < 0: lo comes from bmax and products
// decrease with the first factor: L from z, U from y.  Every ray's t1 = max(lo.x, 0, lo.y, lo.z) >= T1 = max(L.x, L.y, L.z, 0) and
// its t2 = min(hi.x, limit, hi.y, hi.z) <= T2 = min(U.x, U.y, U.z); T1 > T2 therefore means t1 > t2 for every ray whose origin and
// inverse direction lie in B: the reference pushes the child for none of them (ray_bvh_intersection.rs:158).  No NaN can arise: all
// inputs are finite, inv is never 0, and B's inverse bounds keep the sign of the pattern.
#ifndef MP_MCACHE_PAD
#define MP_MCACHE_PAD 0.25f  // widening of the unit's bounds on either side, in extents of the pass that sets them (A/B: 0.0625 .. 1, profiles/r03_notes.md)
#endif
// Table sizes (powers of two).  Counted on the metric's frame (tools/cache_miss_count.py, 16 passes per unit): 17.4 node-mask and 6.1
// leaf-mask slow paths per unit, 1.08 bounds (re)sets; 256 / 256 entries: 22.8 and 5.1 (20.05 against 20.14 ms), 256 / 128: 20.23 ms.
#ifndef MP_NODE_ENTRIES
#define MP_NODE_ENTRIES 512
#endif
#ifndef MP_LEAF_ENTRIES
#define MP_LEAF_ENTRIES 128
#endif
constexpr int kMaskCacheEntries = MP_NODE_ENTRIES;                      // direct-mapped: node index & 511 ; entry = node << 8 | mask
constexpr int kMaskCacheHeader = 32;                        // B: [0..11] origin / inverse-direction bounds, [12] = sign pattern | 0x100 when valid (0xFFFFFFFF: none), [16..21] direction bounds
constexpr int kLeafCacheEntries = MP_LEAF_ENTRIES;                      // direct-mapped: first packet of the leaf & 127 ; tag = first packet, mask = 64 bits (triangle i of the leaf)
constexpr int kLeafTagBase = kMaskCacheHeader + kMaskCacheEntries;
constexpr int kLeafMaskBase = kLeafTagBase + kLeafCacheEntries;   // uint2 per entry (8-byte aligned)
constexpr int kMaskCacheDwords = kLeafMaskBase + 2 * kLeafCacheEntries;
static_assert((kLeafMaskBase % 2) == 0 && (kMaskCacheDwords % 4) == 0, "LDS alignment of the leaf masks / of the next wave's header");
constexpr float kCoordCap = 1073741824.0f;                  // 2^30: magnitude bound of ray origins and triangle vertices for the triangle masks (see tri_may_hit)
struct MaskCache {
    uint32_t* lds;  // this wave's header + entries, or nullptr: no packet-level rejection
};
// Wave-wide minima of N values and maxima of N values at once (every lane takes part; inactive rays hold the neutral element):
// four DPP steps inside each row of 16, row_bcast15 / row_bcast31 across the rows, the totals end in LANE 63's registers (the
// caller goes on in the vector domain and takes lane 63's verdict: the scalar registers would not fit beside the walk's).  The
// independent chains are interleaved step by step, so no instruction reads a register the previous two instructions wrote (the
// DPP read-after-VALU-write hazard needs two wait states) and no s_nop is spent.
#define MP_DPP_STEP6(CTRL)                                                                                              \
    "v_min_f32_dpp %0, %0, %0 " CTRL "\n\tv_min_f32_dpp %1, %1, %1 " CTRL "\n\tv_min_f32_dpp %2, %2, %2 " CTRL "\n\t"         \
    "v_max_f32_dpp %3, %3, %3 " CTRL "\n\tv_max_f32_dpp %4, %4, %4 " CTRL "\n\tv_max_f32_dpp %5, %5, %5 " CTRL "\n\t"
__device__ __forceinline__ void wave_min3_max3(float (&mn)[3], float (&mx)[3]) {
    asm volatile("s_nop 1\n\t"  // the operands may come straight out of VALU instructions
                 MP_DPP_STEP6("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 MP_DPP_STEP6("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 MP_DPP_STEP6("row_half_mirror row_mask:0xf bank_mask:0xf")
                 MP_DPP_STEP6("row_mirror row_mask:0xf bank_mask:0xf")
                 MP_DPP_STEP6("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 MP_DPP_STEP6("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 : "+v"(mn[0]), "+v"(mn[1]), "+v"(mn[2]), "+v"(mx[0]), "+v"(mx[1]), "+v"(mx[2]));
}
// Called once per pass, before a sign-specialised walk with pattern `oct`: makes the cache's bounds B contain this pass's rays
// (widening B and clearing the masks if they do not).  Returns false when the pass cannot use the cache: a non-finite inverse
// direction, an origin beyond 2^30 or a direction component beyond 2 in magnitude (tri_may_hit's no-overflow argument).
// B = origin, inverse-direction and direction bounds; three groups of (3 minima, 3 maxima), header slots g*6 .. g*6+5 for the first
// two and 16..21 for the directions.
__device__ __forceinline__ bool mask_cache_begin_pass(const MaskCache& mc, const Ray& r, bool active, uint32_t oct) {
    const int lane = static_cast<int>(threadIdx.x) & 63;
    const bool fin = fabsf(r.ox) <= kCoordCap && fabsf(r.oy) <= kCoordCap && fabsf(r.oz) <= kCoordCap && fabsf(r.ix) < INFINITY &&
                     fabsf(r.iy) < INFINITY && fabsf(r.iz) < INFINITY && fabsf(r.dx) <= 2.0f && fabsf(r.dy) <= 2.0f && fabsf(r.dz) <= 2.0f;
    if (__ballot(active && !fin) != 0) return false;
    float* hdr = reinterpret_cast<float*>(mc.lds);
    const uint32_t state = __builtin_amdgcn_readfirstlane(mc.lds[12]);
    const bool same = state == (oct | 0x100u);
    const float val[3][3] = {{r.ox, r.oy, r.oz}, {r.ix, r.iy, r.iz}, {r.dx, r.dy, r.dz}};
    // P inside B ?  Every active lane compares its own ray with the header: no reduction in the common case
    bool viol = !same;
    if (same) {
#pragma unroll
        for (int g = 0; g < 3; g++) {
            const int base = g == 2 ? 16 : g * 6;
#pragma unroll
            for (int k = 0; k < 3; k++) viol = viol || val[g][k] < hdr[base + k] || val[g][k] > hdr[base + 3 + k];
        }
    }
    if (__ballot(active && viol) != 0) {
        float pmin[3][3], pmax[3][3];  // after the reductions: lane 63 holds the wave's bounds
#pragma unroll
        for (int g = 0; g < 3; g++) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                pmin[g][k] = active ? val[g][k] : INFINITY;
                pmax[g][k] = active ? val[g][k] : -INFINITY;
            }
            wave_min3_max3(pmin[g], pmax[g]);
        }
        // new bounds: this pass's, united with the old ones when they belong to the same sign pattern, widened by MP_MCACHE_PAD of the extent
        // (an inverse-direction bound never crosses zero: the sign pattern is part of the node masks' meaning; origin bounds stay
        // within 2^31 and direction bounds within 2: every pass that gets here lies well inside)
#pragma unroll
        for (int g = 0; g < 3; g++) {
            const int base = g == 2 ? 16 : g * 6;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                float lo = pmin[g][k], hi = pmax[g][k];
                if (same) { lo = fminf(lo, hdr[base + k]); hi = fmaxf(hi, hdr[base + 3 + k]); }
                const float pad = (hi - lo) * MP_MCACHE_PAD;
                float wlo = lo - pad, whi = hi + pad;
                if (g == 1) {  // same sign as the pass's inverse directions (all of one sign, finite, non-zero)
                    if ((wlo < 0.0f) != (lo < 0.0f) || wlo == 0.0f) wlo = lo;
                    if ((whi < 0.0f) != (hi < 0.0f) || whi == 0.0f) whi = hi;
                    if (!(fabsf(wlo) < INFINITY)) wlo = lo;
                    if (!(fabsf(whi) < INFINITY)) whi = hi;
                } else {
                    const float cap = g == 0 ? 2.0f * kCoordCap : 2.0f;
                    wlo = fmaxf(wlo, -cap); whi = fminf(whi, cap);
                }
                pmin[g][k] = wlo; pmax[g][k] = whi;
            }
        }
        MP_PROF_COUNT(3);
        wave_lds_sync();  // the reads above before the header is rewritten
        if (lane == 63) {
#pragma unroll
            for (int g = 0; g < 3; g++) {
                const int base = g == 2 ? 16 : g * 6;
#pragma unroll
                for (int k = 0; k < 3; k++) { hdr[base + k] = pmin[g][k]; hdr[base + 3 + k] = pmax[g][k]; }
            }
            mc.lds[12] = oct | 0x100u;
        }
        int l_ = lane;  // (re-derived here: the clear runs once per unit, its address is not worth a register across the walk)
        asm volatile("" : "+v"(l_));
#pragma unroll
        for (int i = 0; i < (kMaskCacheEntries + kLeafCacheEntries) / 64; i++) mc.lds[kMaskCacheHeader + i * 64 + l_] = 0xFFFFFFFFu;  // no node / leaf has this tag
        wave_lds_sync();
    }
    return true;
}
// lane j (0..7): can any ray with origin / inverse direction inside the bounds `b` (omin[3], omax[3], imin[3], imax[3]) pass child
// j's box {bmn, bmx}?  (see above)
template <int OCT>
__device__ __forceinline__ bool bounds_may_hit(const float* b, const float bmn[3], const float bmx[3]) {
    float L[3], U[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float y = bmn[k] - b[3 + k], z = bmx[k] - b[k];
        const float lo_src = ((OCT >> k) & 1) ? z : y, hi_src = ((OCT >> k) & 1) ? y : z;
        L[k] = fminf(lo_src * b[6 + k], lo_src * b[9 + k]);
        U[k] = fmaxf(hi_src * b[6 + k], hi_src * b[9 + k]);
    }
    const float t1 = fmaxf(fmaxf(L[0], 0.0f), fmaxf(L[1], L[2])), t2 = fminf(U[0], fminf(U[1], U[2]));
    return !(t1 > t2);
}

// ---- ... and packet-level TRIANGLE rejection with the same bounds --------------------------------------------------------------------
// Counted on the metric's frame (tools/sim_tri_reject.py): a pass of 64 rays through two pixels tests 53 triangles and 3.8 of them are
// hit by some ray; with the unit's bounds B (origins, directions) 41 of the 53 can be PROVEN missed by every ray inside B, once per
// work unit and leaf: the leaf's 64-bit mask "triangle i may be hit" is cached next to the node masks and a pass tests the survivors.
// The proof is the Moeller-Trumbore expression sequence of triangle.rs:183-217 itself, evaluated on intervals: every operation of
// it (fl(a*b), fl(a*b+c), fl(a-b), fl(1/x) on an interval without zero) is monotone in each operand while the others are fixed, so
// the same f32 operation evaluated at the corners of the operand intervals bounds the operation's result for every ray inside B --
// no error analysis, the bounds contain the very f32 values the per-ray test computes.  A triangle is skipped when the bounds show
// u < 0, v < 0, u + v > 1 or t < 0 for all of them (:125 then accepts for no ray).  No NaN can arise on the way: every origin bound
// is within 2^31, every direction bound within 2, every vertex within 2^30 and every edge within 2^31 (DevScene::tris_bounded,
// mask_cache_begin_pass), which keeps all intermediate bounds finite up to the reciprocal (|t numerator| < 2^98); a determinant
// interval that touches zero or has an infinite reciprocal keeps the triangle; after that every value is one product of finite
// numbers (never NaN), and the only sum (u + v) can at worst be inf - inf = NaN, which compares false and keeps the triangle.
struct Iv {
    float lo, hi;
};
__device__ __forceinline__ Iv iv_neg(const Iv a) { return Iv{-a.hi, -a.lo}; }
__device__ __forceinline__ Iv iv_mul_c(const Iv a, const float c) {  // fl(a * c)
    const float p = a.lo * c, q = a.hi * c;
    return Iv{fminf(p, q), fmaxf(p, q)};
}
__device__ __forceinline__ Iv iv_mul(const Iv a, const Iv b) {  // fl(a * b)
    const float p1 = a.lo * b.lo, p2 = a.lo * b.hi, p3 = a.hi * b.lo, p4 = a.hi * b.hi;
    return Iv{fminf(fminf(p1, p2), fminf(p3, p4)), fmaxf(fmaxf(p1, p2), fmaxf(p3, p4))};
}
__device__ __forceinline__ Iv iv_fma_c(const Iv a, const float c, const Iv z) {  // fl(a * c + z)
    return Iv{fminf(__builtin_fmaf(a.lo, c, z.lo), __builtin_fmaf(a.hi, c, z.lo)), fmaxf(__builtin_fmaf(a.lo, c, z.hi), __builtin_fmaf(a.hi, c, z.hi))};
}
__device__ __forceinline__ Iv iv_fma(const Iv a, const Iv b, const Iv z) {  // fl(a * b + z)
    const float l1 = __builtin_fmaf(a.lo, b.lo, z.lo), l2 = __builtin_fmaf(a.lo, b.hi, z.lo), l3 = __builtin_fmaf(a.hi, b.lo, z.lo), l4 = __builtin_fmaf(a.hi, b.hi, z.lo);
    const float h1 = __builtin_fmaf(a.lo, b.lo, z.hi), h2 = __builtin_fmaf(a.lo, b.hi, z.hi), h3 = __builtin_fmaf(a.hi, b.lo, z.hi), h4 = __builtin_fmaf(a.hi, b.hi, z.hi);
    return Iv{fminf(fminf(l1, l2), fminf(l3, l4)), fmaxf(fmaxf(h1, h2), fmaxf(h3, h4))};
}
// can any ray with origin / direction inside the bounds `b` (mask-cache header) hit the triangle {v0, e1, e2}?
__device__ __forceinline__ bool tri_may_hit(const float* b, const float (&v0)[3], const float (&e1)[3], const float (&e2)[3]) {
    const Iv d[3] = {Iv{b[16], b[19]}, Iv{b[17], b[20]}, Iv{b[18], b[21]}};
    // h = (fms(dy, e2z, dz * e2y), fms(dz, e2x, dx * e2z), fms(dx, e2y, dy * e2x)) ; fms(a, b, c) = fma(a, b, -c)
    const Iv h[3] = {iv_fma_c(d[1], e2[2], iv_neg(iv_mul_c(d[2], e2[1]))), iv_fma_c(d[2], e2[0], iv_neg(iv_mul_c(d[0], e2[2]))),
                     iv_fma_c(d[0], e2[1], iv_neg(iv_mul_c(d[1], e2[0])))};
    // fma_dot(a, b) = fma(az, bz, fma(ay, by, ax * bx))
    const Iv det = iv_fma_c(h[2], e1[2], iv_fma_c(h[1], e1[1], iv_mul_c(h[0], e1[0])));
    if (!(det.lo > 0.0f || det.hi < 0.0f)) return true;
    const Iv inv = Iv{1.0f / det.hi, 1.0f / det.lo};
    if (!(fabsf(inv.lo) < INFINITY && fabsf(inv.hi) < INFINITY)) return true;
    const Iv s[3] = {Iv{b[0] - v0[0], b[3] - v0[0]}, Iv{b[1] - v0[1], b[4] - v0[1]}, Iv{b[2] - v0[2], b[5] - v0[2]}};
    const Iv u = iv_mul(inv, iv_fma(s[2], h[2], iv_fma(s[1], h[1], iv_mul(s[0], h[0]))));
    const Iv q[3] = {iv_fma_c(s[1], e1[2], iv_neg(iv_mul_c(s[2], e1[1]))), iv_fma_c(s[2], e1[0], iv_neg(iv_mul_c(s[0], e1[2]))),
                     iv_fma_c(s[0], e1[1], iv_neg(iv_mul_c(s[1], e1[0])))};
    const Iv v = iv_mul(inv, iv_fma(d[2], q[2], iv_fma(d[1], q[1], iv_mul(d[0], q[0]))));
    const Iv t = iv_mul(inv, iv_fma_c(q[2], e2[2], iv_fma_c(q[1], e2[1], iv_mul_c(q[0], e2[0]))));
    const bool miss = u.hi < 0.0f || v.hi < 0.0f || (u.lo + v.lo) > 1.0f || t.hi < 0.0f;
    return !miss;
}

// The triangle masks of one leaf (first packet `first`, n_real triangles), computed and stored in the unit's cache: lane j = triangle
// j, one coalesced read of the leaf's records.  A real call: its registers are needed once per unit and leaf, and inlined they
// would push the walk's per-ray state out of the 64 registers an 8-wave kernel has.
__device__ __noinline__ uint64_t leaf_mask_slow(const float* __restrict__ tris_aos, uint32_t* mcache, uint32_t first, uint32_t n_real) {
    const uint32_t j = threadIdx.x & 63u;
    bool keep = false;
    if (j < n_real) {
        const float* tv = tris_aos + (static_cast<size_t>(first) * 8 + j) * kTriDwords;
        const float v0[3] = {tv[0], tv[1], tv[2]}, e1[3] = {tv[3], tv[4], tv[5]}, e2[3] = {tv[6], tv[7], tv[8]};
        keep = tri_may_hit(reinterpret_cast<const float*>(mcache), v0, e1, e2);
    }
    const uint64_t todo = __ballot(keep);
    if (j == 0u) {
        const uint32_t ls = first & static_cast<uint32_t>(kLeafCacheEntries - 1);
        mcache[kLeafTagBase + ls] = first;
        reinterpret_cast<uint2*>(mcache + kLeafMaskBase)[ls] = make_uint2(static_cast<uint32_t>(todo), static_cast<uint32_t>(todo >> 32));
    }
    return todo;
}

// MODE 1: every active ray has finite inverse directions (no 0*inf, so the NaN patches of aabb.rs:262-267 are dead code).
// MODE 2: literal aabb.rs:254-284.
//
// The walk is fed by the scalar unit (wave-uniform node / triangle records through s_load), and measured on MI355X it is the
// SCALAR ALU, one per CU, that saturates first (profiles/r02_notes.md: an extra SALU instruction per triangle costs 4x what an
// extra VALU instruction costs).  So per-ray predicates that used to be combined as lane masks with s_and / s_andn2 / s_cmp are
// folded into the vector domain: the rays a popped entry is NOT live for are disabled through two per-lane operands -- `lim`
// (best.t, or -1 for a disabled ray: no slab interval and no hit distance passes) and `thr` (the early-out threshold -2^-40, or
// +inf: a disabled ray is always "surely rejected") -- so every ballot below is already the masked result and "no ray left" is a
// branch on VCC; det's magnitude guard is a v_cndmask, the three sign tests one minNum chain; loops are single-exit pair loops
// with a scalar countdown; pushes are v_writelane; staleness is a low-water mark instead of a 64-bit mask.
template <int MODE, int OCT, class Stack>
__device__ __forceinline__ void trace_packet_impl(const DevScene& sc, const Ray& r, bool active, Stack& st, PacketHit& hit) {
    constexpr bool PATCH_NAN = MODE == 2;
    // MODE 2 walks the literal reference tree, MODE 1 the wide tree (thin nodes absorbed into their parents: bit-identical hits
    // for rays with finite inverse directions, device_tree.cpp)
    kfp nodes = (kfp)(uintptr_t)(PATCH_NAN ? sc.nodes_lit : sc.nodes_aos);
    kfp tris = (kfp)(uintptr_t)sc.tris_aos;
    float best_t = FLT_MAX, bu = 0.0f, bv = 0.0f;  // best (:34-37)
    uint32_t bprim = kNoPrim;
    // entry 0 = root (:28-32), t1 = -inf: never culled
    st.push(0, PATCH_NAN ? sc.root_lit : sc.root, kSrcRoot, __ballot(active));
    st.sync(1);
    int sp = 1;
    int stale_top = 0;  // entries [0, stale_top) were pushed before some ray's best.t last changed
    while (sp > 0) {
        sp--;
        uint32_t link, src;
        uint64_t onm;  // rays this entry is still live for
        st.pop(sp, link, src, onm);
        if (sp < stale_top) {  // :40-44, per ray (the root is popped first, before anything can be stale)
            stale_top = sp;
            const krec8 bx = *reinterpret_cast<const __attribute__((address_space(4))) krec8*>(nodes + static_cast<size_t>(src) * 8);  // one load
            const float node_t1 = slab_entry<PATCH_NAN, OCT>(bx[0], bx[1], bx[2], bx[3], bx[4], bx[5], r);
            onm &= ~mask_gt(node_t1, best_t);
        }
        if (onm == 0) continue;
        const bool on = __builtin_amdgcn_inverse_ballot_w64(onm);
        float lim = on ? best_t : -1.0f;          // slab limit / hit-distance bound: nothing passes for a disabled ray
        if ((link & 63u) == 0u) {  // device link (mp_internal.h): inner = index << 6, leaf = first packet << 6 | real triangles
            // InnerNode::intersect :149-162, children ascending.  Child record = {min.xyz, max.xyz, link, n}: n (record 0 only)
            // = index of the node's last real child + 1.  Two SGPR sets alternate: B is fetched while A is tested.
            const uint32_t node = link >> 6;
            kfp nd = nodes + static_cast<size_t>(node) * 64;
            auto child = [&](const float b0, const float b1, const float b2, const float b3, const float b4, const float b5,
                             const float blink, const uint32_t cslot) {
                const uint32_t cl = uniform_u(blink);
                if (cl == MP_LINK_NULL) return;  // Null links are skipped at pop in the reference (:49)
                float t1, t2;
                if (PATCH_NAN) {
                    slab<PATCH_NAN, OCT>(b0, b1, b2, b3, b4, b5, r, lim, t1, t2);
                } else {
                    // aabb.rs:254-284 in two stages (round 3): t1 = max(lo.x, 0, lo.y, lo.z) and t2 = min(hi.x, limit, hi.y, hi.z) are
                    // the same numbers in any order of evaluation (max / min of the same operands), and max(lo.x, lo.y, 0) >
                    // min(hi.x, hi.y, limit) already decides t1 > t2.  Counted on the metric's frame (profiles/r03_notes.md): for
                    // 60 % of the child boxes a packet tests, no ray survives the x and y slabs -- the z slab and the rest are
                    // skipped for the whole wave (11 instead of 17 VALU).  Which two axes go first is a compile-time choice.
#ifndef MP_SLAB_LAST
#define MP_SLAB_LAST 2  // the axis whose slab is tested last (A/B-measured: profiles/r03_notes.md)
#endif
                    constexpr int A2 = MP_SLAB_LAST, A0 = (A2 + 1) % 3, A1 = (A2 + 2) % 3;
                    const float bn[3] = {b0, b1, b2}, bx3[3] = {b3, b4, b5}, ro[3] = {r.ox, r.oy, r.oz}, ri[3] = {r.ix, r.iy, r.iz};
                    const float a0_ = (bn[A0] - ro[A0]) * ri[A0], c0_ = (bx3[A0] - ro[A0]) * ri[A0];
                    const float a1_ = (bn[A1] - ro[A1]) * ri[A1], c1_ = (bx3[A1] - ro[A1]) * ri[A1];
                    float lo0, hi0, lo1, hi1;
                    if (OCT >= 0) {
                        lo0 = ((OCT >> A0) & 1) ? c0_ : a0_; hi0 = ((OCT >> A0) & 1) ? a0_ : c0_;
                        lo1 = ((OCT >> A1) & 1) ? c1_ : a1_; hi1 = ((OCT >> A1) & 1) ? a1_ : c1_;
                    } else {
                        lo0 = fminf(a0_, c0_); hi0 = fmaxf(a0_, c0_); lo1 = fminf(a1_, c1_); hi1 = fmaxf(a1_, c1_);
                    }
                    const float t1p = fmaxf(fmaxf(lo0, 0.0f), lo1), t2p = fminf(fminf(hi0, lim), hi1);
                    if (__ballot(t1p <= t2p) == 0) return;
                    const float a2_ = (bn[A2] - ro[A2]) * ri[A2], c2_ = (bx3[A2] - ro[A2]) * ri[A2];
                    float lo2, hi2;
                    if (OCT >= 0) { lo2 = ((OCT >> A2) & 1) ? c2_ : a2_; hi2 = ((OCT >> A2) & 1) ? a2_ : c2_; }
                    else { lo2 = fminf(a2_, c2_); hi2 = fmaxf(a2_, c2_); }
                    t1 = fmaxf(t1p, lo2);
                    t2 = fminf(t2p, hi2);
                }
                const uint64_t okm = __ballot(t1 <= t2);
                if (okm != 0) {
                    st.push(sp, cl, cslot, okm);
                    sp++;
                }
            };
            float a0 = nd[0], a1 = nd[1], a2 = nd[2], a3 = nd[3], a4 = nd[4], a5 = nd[5], a6 = nd[6];
            const uint32_t nchild = uniform_u(nd[7]);
            uint32_t slot = node * 8u;
            for (uint32_t p = nchild >> 1; p > 0; p--) {
                const float b0 = nd[8], b1 = nd[9], b2 = nd[10], b3 = nd[11], b4 = nd[12], b5 = nd[13], b6 = nd[14];
                child(a0, a1, a2, a3, a4, a5, a6, slot);
                a0 = nd[16]; a1 = nd[17]; a2 = nd[18]; a3 = nd[19]; a4 = nd[20]; a5 = nd[21]; a6 = nd[22];
                child(b0, b1, b2, b3, b4, b5, b6, slot + 1u);
                slot += 2u;
                nd += 16;
            }
            if (nchild & 1u) child(a0, a1, a2, a3, a4, a5, a6, slot);
            st.sync(sp);
        } else {
            // intersect_triangles :104-140 ; every lane walks the leaf's triangles in (packet, lane) order with a
            // strict `<`.  Padding (only at the tail of the last packet) can never be accepted and is not visited.
            const uint32_t first = link >> 6, n_real = link & 63u;  // the count travels in the link: no dependent load before the first triangle
            kfp tp = tris + static_cast<size_t>(first) * (8 * kTriDwords);
            const float thr = on ? -kTiny : INFINITY;  // early-out threshold: a disabled ray is always "surely rejected"
            uint64_t changed = 0;  // lanes that accepted a hit in this leaf
            auto test = [&](const float v0x, const float v0y, const float v0z, const float e1x, const float e1y, const float e1z,
                            const float e2x, const float e2y, const float e2z, const uint32_t tri) {
                // triangle.rs:183-217
                const float hx = fms(r.dy, e2z, r.dz * e2y), hy = fms(r.dz, e2x, r.dx * e2z), hz = fms(r.dx, e2y, r.dy * e2x);
                const float det = fma_dot(e1x, e1y, e1z, hx, hy, hz);
                const float sx = r.ox - v0x, sy = r.oy - v0y, sz = r.oz - v0z;
                const float un = fma_dot(sx, sy, sz, hx, hy, hz);
                // Exact early-outs.  A numerator whose sign differs from det's, with |num| >= 2^-40 and |det| <= 2^40, gives a
                // non-zero negative quotient fl(fl(1/det) * num) (no underflow to -0, which would pass `>= 0`): u >= 0 (v >= 0,
                // t >= 0) cannot hold.  x = num with its sign flipped by det's (or +1 where |det| is too large to tell):
                // "surely negative" is the single ordered compare x <= -2^-40; NaN never rejects.
                const uint32_t det_sign = as_u(det) & 0x80000000u;
                const bool det_ok = fabsf(det) <= kHuge;
                const float xu = det_ok ? as_f(as_u(un) ^ det_sign) : 1.0f;
                if (__ballot(!(xu <= thr)) == 0) return;  // no live ray can have u >= 0
                const float qx = fms(sy, e1z, sz * e1y), qy = fms(sz, e1x, sx * e1z), qz = fms(sx, e1y, sy * e1x);
                const float vn = fma_dot(r.dx, r.dy, r.dz, qx, qy, qz);
                const float tn = fma_dot(e2x, e2y, e2z, qx, qy, qz);
                const float xv = det_ok ? as_f(as_u(vn) ^ det_sign) : 1.0f, xt = det_ok ? as_f(as_u(tn) ^ det_sign) : 1.0f;
                // minNum drops NaN operands, so the minimum is <= the threshold iff one of the three ordered compares holds
                if (__ballot(!(fminf(fminf(xu, xv), xt) <= thr)) == 0) return;
                const float inv_det = 1.0f / det;
                const float u = inv_det * un, v = inv_det * vn, t = inv_det * tn;
                // mask & t>=0 & t<=max_t (:125), strict `<` vs the leaf best, then vs the global best (:129,:59):
                // equivalent to a running strict `<` against best.t (best.t never exceeds max_t); lim == best.t for live rays
                const bool acc = (u >= 0.0f) & (v >= 0.0f) & ((u + v) <= 1.0f) & (t >= 0.0f) & (t < lim);
                best_t = acc ? t : best_t;
                lim = acc ? t : lim;
                bu = acc ? u : bu;
                bv = acc ? v : bv;
                bprim = acc ? tri : bprim;
                changed |= __ballot(acc);
            };
            // two register sets (A, B) alternate: B is fetched while A is tested and vice versa (the array has tail padding)
            float a0 = tp[0], a1 = tp[1], a2 = tp[2], a3 = tp[3], a4 = tp[4], a5 = tp[5], a6 = tp[6], a7 = tp[7], a8 = tp[8];
            uint32_t tri = first * 8u;
            for (uint32_t p = n_real >> 1; p > 0; p--) {
                const float b0 = tp[9], b1 = tp[10], b2 = tp[11], b3 = tp[12], b4 = tp[13], b5 = tp[14], b6 = tp[15], b7 = tp[16], b8 = tp[17];
                test(a0, a1, a2, a3, a4, a5, a6, a7, a8, tri);
                a0 = tp[18]; a1 = tp[19]; a2 = tp[20]; a3 = tp[21]; a4 = tp[22]; a5 = tp[23]; a6 = tp[24]; a7 = tp[25]; a8 = tp[26];
                test(b0, b1, b2, b3, b4, b5, b6, b7, b8, tri + 1u);
                tri += 2u;
                tp += 2 * kTriDwords;
            }
            if (n_real & 1u) test(a0, a1, a2, a3, a4, a5, a6, a7, a8, tri);
            if (changed != 0) stale_top = sp;  // every entry still on the stack predates this change
        }
    }
    hit.t = best_t; hit.u = bu; hit.v = bv; hit.prim = bprim;
}

// Wave-uniform stack of the cached walk: three VGPRs used as 64-entry arrays.  An entry is a FRAME -- a visited node with the children
// that are still to be popped (node << 8 | 8-bit mask) and the 64-bit mask of the rays that were live at the visit.  The frame on
// top lives in scalar registers: popping a child is three scalar instructions, and the arrays are touched once per node visit
// instead of once per child.
struct RegStack3 {
    int ent, mlo, mhi;
    __device__ __forceinline__ RegStack3() : ent(0), mlo(0), mhi(0) {}
    __device__ __forceinline__ void push(int sp, uint32_t e, uint64_t m) {
        asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "v_writelane_b32 %0, %4, m0\n\tv_writelane_b32 %1, %5, m0\n\tv_writelane_b32 %2, %6, m0"
                     : "+v"(ent), "+v"(mlo), "+v"(mhi)
                     : "s"(sp), "s"(e), "s"(static_cast<uint32_t>(m)), "s"(static_cast<uint32_t>(m >> 32))
                     : "m0");
    }
    __device__ __forceinline__ void pop(int sp, uint32_t& e, uint64_t& m) const {
        e = static_cast<uint32_t>(__builtin_amdgcn_readlane(ent, sp));
        m = static_cast<uint32_t>(__builtin_amdgcn_readlane(mlo, sp)) |
            (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(mhi, sp))) << 32);
    }
};

// The sign-specialised walk of a kernel with a per-unit mask cache (MaskCache, tri_may_hit; `mcache` = this wave's header and
// entries; every active ray has finite inverse directions of the sign pattern OCT and lies inside the cache's bounds).
// Same visits, same per-ray decisions as trace_packet_impl<1, OCT>, organised around the cache: a node visit is ONE lookup and
// opens a frame with the children some ray of the unit may pass, WITHOUT testing them; a child's per-ray slab test runs when it is
// popped from its frame (highest index first: the reference pushes ascending and pops descending, :161), against best.t as it is
// then.  That is the reference's decision: it pushes a child when t1 <= min(hi, best.t at the visit) (:158) and visits it when
// !(t1 > best.t at the pop) (:40); best.t only shrinks and t1 is never NaN here, so both hold exactly when t1 <= min(hi, best.t
// at the pop) -- one test with the later limit -- for a ray that was live at the parent's visit (the frame carries that mask: a
// child box may stick out of its parent's by an ulp).  No entry distance is stored or recomputed, no record is fetched at the
// visit, and the child's record -- box and link -- is one 32-byte scalar load at the pop.  The root is child 0 of a pseudo-node
// behind the last node: its record (device_tree.cpp) is an unbounded box, which every ray passes with t1 = 0 -- never culled (:28-32).
template <int OCT>
__device__ __forceinline__ void trace_packet_cached(const DevScene& sc, const Ray& r, bool active, PacketHit& hit, uint32_t* mcache) {
    kfp nodes = (kfp)(uintptr_t)sc.nodes_aos;
    kfp tris = (kfp)(uintptr_t)sc.tris_aos;
    float best_t = FLT_MAX, bu = 0.0f, bv = 0.0f;  // best (:34-37)
    uint32_t bprim = kNoPrim;
    RegStack3 st;
    int sp = 0;                                   // frames below the current one
    uint32_t cur = (sc.inner_count << 8) | 1u;    // current frame: node << 8 | children still to pop
    uint64_t pm = __ballot(active);               // ... and the rays that were live at its visit
    // (fetching the record of the child that is popped next ahead of its pop was built and measured slower: 27.0 against 24.3 ms)
    for (;;) {
        if ((cur & 0xFFu) == 0u) {  // frame exhausted: back to the one below
            if (sp == 0) break;
            sp--;
            st.pop(sp, cur, pm);
        }
        const uint32_t c = 31u - static_cast<uint32_t>(__builtin_clz(cur & 0xFFu));  // highest child first
        cur &= ~(1u << c);
        const uint32_t src = ((cur >> 8) << 3) | c;
        const bool pon = __builtin_amdgcn_inverse_ballot_w64(pm);
        // (base + a 32-bit byte offset: s_load's register-offset form; fewer than 2^24 nodes, checked at upload)
        const krec8 rec = *reinterpret_cast<const __attribute__((address_space(4))) krec8*>(reinterpret_cast<const __attribute__((address_space(4))) char*>(nodes) + scalar_u(src * 32u));
        float t1, t2;
        slab<false, OCT>(rec[0], rec[1], rec[2], rec[3], rec[4], rec[5], r, pon ? best_t : -1.0f, t1, t2);  // aabb.rs:254-284
        const bool ok = t1 <= t2;
        if (__ballot(ok) == 0) continue;
        const uint32_t link = uniform_u(rec[6]);
        float lim = ok ? best_t : -1.0f;  // best.t for the rays this child is live for, -1 for the others: no slab interval and no hit distance passes
        if ((link & 63u) == 0u) {  // inner node (device link, mp_internal.h)
            MP_PROF_COUNT(2);
            const uint32_t node = link >> 6;
            const uint32_t cslot = static_cast<uint32_t>(kMaskCacheHeader) + (node & static_cast<uint32_t>(kMaskCacheEntries - 1));
            const uint32_t e = __builtin_amdgcn_readfirstlane(mcache[cslot]);
            uint32_t todo;
            if (__builtin_expect((e >> 8) == node, 1)) {
                todo = e & 0xFFu;
            } else {  // first visit of this node under the current bounds: lane j = child j, records through one vector load pair
                MP_PROF_COUNT(0);
                const int cj = static_cast<int>(threadIdx.x) & 7;
                bool keep = false;
                if ((threadIdx.x & 63u) < 8u) {
                    const float4* rec4 = reinterpret_cast<const float4*>(sc.nodes_aos) + (static_cast<size_t>(node) * 8 + static_cast<size_t>(cj)) * 2;
                    const float4 c0 = rec4[0], c1 = rec4[1];  // {min.xyz, max.x} {max.yz, link, n}
                    const float bmn[3] = {c0.x, c0.y, c0.z}, bmx[3] = {c0.w, c1.x, c1.y};
                    keep = as_u(c1.z) != MP_LINK_NULL && bounds_may_hit<OCT>(reinterpret_cast<const float*>(mcache), bmn, bmx);
                }
                todo = static_cast<uint32_t>(__ballot(keep)) & 0xFFu;
                if ((threadIdx.x & 63u) == 0u) mcache[cslot] = (node << 8) | todo;
            }
            if (todo != 0u) {  // a new frame; the one it replaces goes to the arrays if it still has children
                if ((cur & 0xFFu) != 0u) {
                    st.push(sp, cur, pm);
                    sp++;
                }
                cur = (node << 8) | todo;
                pm = __ballot(lim >= 0.0f);
            }
        } else {
            // intersect_triangles :104-140 ; every lane walks the leaf's surviving triangles in (packet, lane) order with a strict `<`
            const uint32_t first = link >> 6, n_real = link & 63u;
            kfp tp = tris + static_cast<size_t>(first) * (8 * kTriDwords);
            const float thr = lim >= 0.0f ? -kTiny : INFINITY;  // early-out threshold: a disabled ray is always "surely rejected"
            auto test = [&](const float v0x, const float v0y, const float v0z, const float e1x, const float e1y, const float e1z,
                            const float e2x, const float e2y, const float e2z, const uint32_t tri) {
                // triangle.rs:183-217 (early-out argument: see trace_packet_impl)
                const float hx = fms(r.dy, e2z, r.dz * e2y), hy = fms(r.dz, e2x, r.dx * e2z), hz = fms(r.dx, e2y, r.dy * e2x);
                const float det = fma_dot(e1x, e1y, e1z, hx, hy, hz);
                const float sx = r.ox - v0x, sy = r.oy - v0y, sz = r.oz - v0z;
                const float un = fma_dot(sx, sy, sz, hx, hy, hz);
                const uint32_t det_sign = as_u(det) & 0x80000000u;
                const bool det_ok = fabsf(det) <= kHuge;
                const float xu = det_ok ? as_f(as_u(un) ^ det_sign) : 1.0f;
                if (__ballot(!(xu <= thr)) == 0) return;  // no live ray can have u >= 0
                const float qx = fms(sy, e1z, sz * e1y), qy = fms(sz, e1x, sx * e1z), qz = fms(sx, e1y, sy * e1x);
                const float vn = fma_dot(r.dx, r.dy, r.dz, qx, qy, qz);
                const float tn = fma_dot(e2x, e2y, e2z, qx, qy, qz);
                const float xv = det_ok ? as_f(as_u(vn) ^ det_sign) : 1.0f, xt = det_ok ? as_f(as_u(tn) ^ det_sign) : 1.0f;
                if (__ballot(!(fminf(fminf(xu, xv), xt) <= thr)) == 0) return;
                const float inv_det = 1.0f / det;
                const float u = inv_det * un, v = inv_det * vn, t = inv_det * tn;
                const bool acc = (u >= 0.0f) & (v >= 0.0f) & ((u + v) <= 1.0f) & (t >= 0.0f) & (t < lim);  // :125, :129, :59
                best_t = acc ? t : best_t;
                lim = acc ? t : lim;
                bu = acc ? u : bu;
                bv = acc ? v : bv;
                bprim = acc ? tri : bprim;
            };
            // per-unit triangle masks (see tri_may_hit): which triangles of this leaf can ANY ray inside the unit's bounds hit?
            const uint32_t ls = first & static_cast<uint32_t>(kLeafCacheEntries - 1);
            const uint32_t tag = mcache[kLeafTagBase + ls];
            const uint2 tm = reinterpret_cast<const uint2*>(mcache + kLeafMaskBase)[ls];
            uint64_t todo;
            if (__builtin_expect(__builtin_amdgcn_readfirstlane(tag) == first, 1)) {
                todo = __builtin_amdgcn_readfirstlane(tm.x) | (static_cast<uint64_t>(__builtin_amdgcn_readfirstlane(tm.y)) << 32);
            } else {  // first visit of this leaf under the current bounds
                MP_PROF_COUNT(1);
                const uint64_t m = leaf_mask_slow(sc.tris_aos, mcache, first, n_real);  // (a call's result is not known to be uniform)
                todo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(m)) | (static_cast<uint64_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(m >> 32))) << 32);
            }
            // the next survivor's record is fetched while this one is tested (one register set rotated through moves: a two-set
            // form without the moves measured 3 % slower -- code size; no prefetch at all: the same time)
            if (todo != 0) {
                uint32_t c = static_cast<uint32_t>(__builtin_ctzll(todo));
                todo &= ~(1ull << c);
                kfp ta = tri_record(tp, c);
                krec8 ra = *reinterpret_cast<const __attribute__((address_space(4))) krec8*>(ta);  // (one block of eight registers: rotated with 64-bit moves)
                float a8 = ta[8];
                while (todo != 0) {
                    const uint32_t cn = static_cast<uint32_t>(__builtin_ctzll(todo));
                    todo &= ~(1ull << cn);
                    kfp tb = tri_record(tp, cn);
                    const krec8 rb = *reinterpret_cast<const __attribute__((address_space(4))) krec8*>(tb);
                    const float b8 = tb[8];
                    test(ra[0], ra[1], ra[2], ra[3], ra[4], ra[5], ra[6], ra[7], a8, first * 8u + c);
                    ra = rb; a8 = b8;
                    c = cn;
                }
                test(ra[0], ra[1], ra[2], ra[3], ra[4], ra[5], ra[6], ra[7], a8, first * 8u + c);
            }
        }
    }
    hit.t = best_t; hit.u = bu; hit.v = bv; hit.prim = bprim;
}

// OCTANTS: also instantiate the eight sign-specialised walks (the production kernels; the rest keep the generic slab).
template <bool OCTANTS, class Stack, bool MC = false>
__device__ __forceinline__ void trace_packet(const DevScene& sc, const Ray& r, bool active, Stack& st, PacketHit& hit,
                                             const MaskCache& mc = MaskCache{nullptr}) {
    const bool slow = active && (fabsf(r.ix) == INFINITY || fabsf(r.iy) == INFINITY || fabsf(r.iz) == INFINITY);
    if (__ballot(slow) != 0) {
        trace_packet_impl<2, -1>(sc, r, active, st, hit);
        return;
    }
    if (OCTANTS && sc.boxes_ordered) {
        const uint64_t am = __ballot(active);
        const uint64_t nx = __ballot(active && r.ix < 0.0f), ny = __ballot(active && r.iy < 0.0f), nz = __ballot(active && r.iz < 0.0f);
        if ((nx == 0 || nx == am) && (ny == 0 || ny == am) && (nz == 0 || nz == am)) {
            const uint32_t oct = (nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u);
            if (MC) {  // kernels with a mask cache: the sign-specialised walks use it; a pass with a non-finite component takes the generic walk
                if (mask_cache_begin_pass(mc, r, active, oct)) {
                    switch (oct) {
                        case 0: trace_packet_cached<0>(sc, r, active, hit, mc.lds); return;
                        case 1: trace_packet_cached<1>(sc, r, active, hit, mc.lds); return;
                        case 2: trace_packet_cached<2>(sc, r, active, hit, mc.lds); return;
                        case 3: trace_packet_cached<3>(sc, r, active, hit, mc.lds); return;
                        case 4: trace_packet_cached<4>(sc, r, active, hit, mc.lds); return;
                        case 5: trace_packet_cached<5>(sc, r, active, hit, mc.lds); return;
                        case 6: trace_packet_cached<6>(sc, r, active, hit, mc.lds); return;
                        default: trace_packet_cached<7>(sc, r, active, hit, mc.lds); return;
                    }
                }
            } else {
                switch (oct) {
                    case 0: trace_packet_impl<1, 0>(sc, r, active, st, hit); return;
                    case 1: trace_packet_impl<1, 1>(sc, r, active, st, hit); return;
                    case 2: trace_packet_impl<1, 2>(sc, r, active, st, hit); return;
                    case 3: trace_packet_impl<1, 3>(sc, r, active, st, hit); return;
                    case 4: trace_packet_impl<1, 4>(sc, r, active, st, hit); return;
                    case 5: trace_packet_impl<1, 5>(sc, r, active, st, hit); return;
                    case 6: trace_packet_impl<1, 6>(sc, r, active, st, hit); return;
                    default: trace_packet_impl<1, 7>(sc, r, active, st, hit); return;
                }
            }
        }
    }
    trace_packet_impl<1, -1>(sc, r, active, st, hit);
}

// Object group (mp_scene_group / mp_scene_instances) on the packet walk: member by member, in order, the 64 rays are moved into
// the member's frame (object_ray) and walk the member's own tree as one packet; closest wins with a strict `<`, so the first
// member keeps ties -- the definition trace_objects (8-lane groups) and the oracle's bvh_intersect_impl implement.  A Sphere
// member is intersected lane-parallel.  k is wave-uniform: the member's descriptor comes through scalar loads.
template <class T>
__device__ __forceinline__ const T* uniform_ptr(const T* p) {  // a wave-uniform pointer that the compiler holds in VGPRs -> SGPRs
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v)), hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
    return reinterpret_cast<const T*>((static_cast<uint64_t>(hi) << 32) | lo);
}

template <bool OCTANTS, class Stack>
__device__ __forceinline__ void trace_packet_objects(const DevScene& sc, const Ray& r, bool act, Stack& st, PacketHit& h, uint32_t& inst) {
    h.t = FLT_MAX; h.u = h.v = 0.0f; h.prim = kNoPrim;
    inst = 0u;
    for (uint32_t k = 0; k < sc.inst_count; k++) {
        DevScene sk;
        Ray rk;
        object_scene(sc, k, sk);
        object_ray(sc, k, r, rk);
        // the packet walk feeds the scalar unit: what it reads through s_load / pushes with v_writelane must sit in SGPRs
        sk.kind = __builtin_amdgcn_readfirstlane(sk.kind);
        sk.root = __builtin_amdgcn_readfirstlane(sk.root);
        sk.root_lit = __builtin_amdgcn_readfirstlane(sk.root_lit);
        sk.nodes_aos = uniform_ptr(sk.nodes_aos);
        sk.nodes_lit = uniform_ptr(sk.nodes_lit);
        sk.tris_aos = uniform_ptr(sk.tris_aos);
        if (sk.kind == 1u) {  // scene/primitives.rs:16-48 ; prim 0
            float ts, nn[3];
            if (act && sphere_intersect(sk, rk, ts, nn) && ts < h.t) { h.t = ts; h.prim = 0u; h.u = h.v = 0.0f; inst = k; }
            continue;
        }
        const bool go = act && may_hit_scene(sk, rk);
        if (__ballot(go) == 0) continue;
        PacketHit hk;
        trace_packet<OCTANTS>(sk, rk, go, st, hk);
        if (hk.prim != kNoPrim && hk.t < h.t) { h = hk; inst = k; }
    }
}

// ---- two rays per lane: 128 rays share one walk (opt-in: mp_ctx_set_option "packet_rays_per_lane" = 2) --------------------
// The walk's scalar work (record fetches, loop control, pops and pushes) does not depend on how many rays ride on it, and the
// scalar ALU is the unit the 64-ray walk saturates first (profiles/r02_notes.md).  Here every lane carries TWO rays (A and B: the
// same sample index in two pixels two columns apart), so the scalar work per ray halves and each fetched record feeds two
// independent vector dependency chains.  Per-ray decisions are exactly those of trace_packet_impl (same operations on the same
// operands: the frame is bit-identical, tested); an entry is visited while any ray of either set needs it.
// MEASURED SLOWER than the 64-ray walk on MI355X (metric's frame: 52.9 ms at 6 waves / 80 VGPRs, 54.2 at 5, 56.5 at 4, against
// 48.6 ms; teapot 13.5 against 12.4): the union of two 2x2-pixel packets is visited by twice the vector work, the wave-level
// early-outs fire less often, and the 80-VGPR budget spills in the pop loop.  Kept as the measured alternative, not the default.
struct RegStack2 {
    int link, src, alo, ahi, blo, bhi;
    __device__ __forceinline__ RegStack2() : link(0), src(0), alo(0), ahi(0), blo(0), bhi(0) {}
    __device__ __forceinline__ void push(int sp, uint32_t l, uint32_t s, uint64_t ma, uint64_t mb) {
        asm volatile(
            "s_mov_b32 m0, %6\n\ts_nop 0\n\t"
            "v_writelane_b32 %0, %7, m0\n\tv_writelane_b32 %1, %8, m0\n\tv_writelane_b32 %2, %9, m0\n\tv_writelane_b32 %3, %10, m0\n\t"
            "v_writelane_b32 %4, %11, m0\n\tv_writelane_b32 %5, %12, m0"
            : "+v"(link), "+v"(src), "+v"(alo), "+v"(ahi), "+v"(blo), "+v"(bhi)
            : "s"(sp), "s"(l), "s"(s), "s"(static_cast<uint32_t>(ma)), "s"(static_cast<uint32_t>(ma >> 32)),
              "s"(static_cast<uint32_t>(mb)), "s"(static_cast<uint32_t>(mb >> 32))
            : "m0");
    }
    __device__ __forceinline__ void pop(int sp, uint32_t& l, uint32_t& s, uint64_t& ma, uint64_t& mb) const {
        l = static_cast<uint32_t>(__builtin_amdgcn_readlane(link, sp));
        s = static_cast<uint32_t>(__builtin_amdgcn_readlane(src, sp));
        ma = static_cast<uint32_t>(__builtin_amdgcn_readlane(alo, sp)) |
             (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(ahi, sp))) << 32);
        mb = static_cast<uint32_t>(__builtin_amdgcn_readlane(blo, sp)) |
             (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(bhi, sp))) << 32);
    }
};

template <int MODE, int OCT>
__device__ __forceinline__ void trace_packet2_impl(const DevScene& sc, const Ray& ra, const Ray& rb, bool active_a, bool active_b,
                                                   PacketHit& hit_a, PacketHit& hit_b) {
    constexpr bool PATCH_NAN = MODE == 2;
    kfp nodes = (kfp)(uintptr_t)(PATCH_NAN ? sc.nodes_lit : sc.nodes_aos);  // literal / wide tree, as in trace_packet_impl
    kfp tris = (kfp)(uintptr_t)sc.tris_aos;
    RegStack2 st;
    float best_a = FLT_MAX, ua = 0.0f, va = 0.0f, best_b = FLT_MAX, ub = 0.0f, vb = 0.0f;  // best (:34-37), per ray
    uint32_t prim_a = kNoPrim, prim_b = kNoPrim;
    st.push(0, PATCH_NAN ? sc.root_lit : sc.root, kSrcRoot, __ballot(active_a), __ballot(active_b));  // entry 0 = root (:28-32), never culled
    int sp = 1;
    int stale_top = 0;  // entries [0, stale_top) were pushed before some ray's best.t last changed
    while (sp > 0) {
        sp--;
        uint32_t link, src;
        uint64_t onm_a, onm_b;  // rays of each set this entry is still live for
        st.pop(sp, link, src, onm_a, onm_b);
        if (sp < stale_top) {  // :40-44, per ray
            stale_top = sp;
            kfp bx = nodes + static_cast<size_t>(src) * 8;
            const float b0 = bx[0], b1 = bx[1], b2 = bx[2], b3 = bx[3], b4 = bx[4], b5 = bx[5];
            onm_a &= ~mask_gt(slab_entry<PATCH_NAN, OCT>(b0, b1, b2, b3, b4, b5, ra), best_a);
            onm_b &= ~mask_gt(slab_entry<PATCH_NAN, OCT>(b0, b1, b2, b3, b4, b5, rb), best_b);
        }
        if ((onm_a | onm_b) == 0) continue;
        const bool on_a = __builtin_amdgcn_inverse_ballot_w64(onm_a), on_b = __builtin_amdgcn_inverse_ballot_w64(onm_b);
        float lim_a = on_a ? best_a : -1.0f, lim_b = on_b ? best_b : -1.0f;  // nothing passes for a disabled ray
        if ((link & 63u) == 0u) {
            // InnerNode::intersect :149-162, children ascending; record = {min.xyz, max.xyz, dlink, n}
            const uint32_t node = link >> 6;
            kfp nd = nodes + static_cast<size_t>(node) * 64;
            float a0 = nd[0], a1 = nd[1], a2 = nd[2], a3 = nd[3], a4 = nd[4], a5 = nd[5], a6 = nd[6];
            const uint32_t nchild = uniform_u(nd[7]);
            uint32_t slot = node * 8u;
            auto child = [&](const float b0, const float b1, const float b2, const float b3, const float b4, const float b5,
                             const float blink, const uint32_t cslot) {
                const uint32_t cl = uniform_u(blink);
                if (cl == MP_LINK_NULL) return;  // Null links are skipped at pop in the reference (:49)
                float t1a, t2a, t1b, t2b;
                slab<PATCH_NAN, OCT>(b0, b1, b2, b3, b4, b5, ra, lim_a, t1a, t2a);
                slab<PATCH_NAN, OCT>(b0, b1, b2, b3, b4, b5, rb, lim_b, t1b, t2b);
                const uint64_t ok_a = __ballot(t1a <= t2a), ok_b = __ballot(t1b <= t2b);
                if ((ok_a | ok_b) != 0) {
                    st.push(sp, cl, cslot, ok_a, ok_b);
                    sp++;
                }
            };
            for (uint32_t p = nchild >> 1; p > 0; p--) {
                const float b0 = nd[8], b1 = nd[9], b2 = nd[10], b3 = nd[11], b4 = nd[12], b5 = nd[13], b6 = nd[14];
                child(a0, a1, a2, a3, a4, a5, a6, slot);
                a0 = nd[16]; a1 = nd[17]; a2 = nd[18]; a3 = nd[19]; a4 = nd[20]; a5 = nd[21]; a6 = nd[22];
                child(b0, b1, b2, b3, b4, b5, b6, slot + 1u);
                slot += 2u;
                nd += 16;
            }
            if (nchild & 1u) child(a0, a1, a2, a3, a4, a5, a6, slot);
        } else {
            // intersect_triangles :104-140 for both rays of every lane
            const uint32_t first = link >> 6, n_real = link & 63u;
            kfp tp = tris + static_cast<size_t>(first) * (8 * kTriDwords);
            const float thr_a = on_a ? -kTiny : INFINITY, thr_b = on_b ? -kTiny : INFINITY;
            uint64_t changed = 0;
            auto test = [&](const float v0x, const float v0y, const float v0z, const float e1x, const float e1y, const float e1z,
                            const float e2x, const float e2y, const float e2z, const uint32_t tri) {
                // triangle.rs:183-217, ray A and ray B side by side (see trace_packet_impl for the early-out argument)
                const float hxa = fms(ra.dy, e2z, ra.dz * e2y), hya = fms(ra.dz, e2x, ra.dx * e2z), hza = fms(ra.dx, e2y, ra.dy * e2x);
                const float hxb = fms(rb.dy, e2z, rb.dz * e2y), hyb = fms(rb.dz, e2x, rb.dx * e2z), hzb = fms(rb.dx, e2y, rb.dy * e2x);
                const float det_a = fma_dot(e1x, e1y, e1z, hxa, hya, hza), det_b = fma_dot(e1x, e1y, e1z, hxb, hyb, hzb);
                const float sxa = ra.ox - v0x, sya = ra.oy - v0y, sza = ra.oz - v0z;
                const float sxb = rb.ox - v0x, syb = rb.oy - v0y, szb = rb.oz - v0z;
                const float un_a = fma_dot(sxa, sya, sza, hxa, hya, hza), un_b = fma_dot(sxb, syb, szb, hxb, hyb, hzb);
                const uint32_t sg_a = as_u(det_a) & 0x80000000u, sg_b = as_u(det_b) & 0x80000000u;
                const bool ok_a = fabsf(det_a) <= kHuge, ok_b = fabsf(det_b) <= kHuge;
                const float xua = ok_a ? as_f(as_u(un_a) ^ sg_a) : 1.0f, xub = ok_b ? as_f(as_u(un_b) ^ sg_b) : 1.0f;
                if ((__ballot(!(xua <= thr_a)) | __ballot(!(xub <= thr_b))) == 0) return;  // no live ray of either set can have u >= 0
                const float qxa = fms(sya, e1z, sza * e1y), qya = fms(sza, e1x, sxa * e1z), qza = fms(sxa, e1y, sya * e1x);
                const float qxb = fms(syb, e1z, szb * e1y), qyb = fms(szb, e1x, sxb * e1z), qzb = fms(sxb, e1y, syb * e1x);
                const float vn_a = fma_dot(ra.dx, ra.dy, ra.dz, qxa, qya, qza), vn_b = fma_dot(rb.dx, rb.dy, rb.dz, qxb, qyb, qzb);
                const float tn_a = fma_dot(e2x, e2y, e2z, qxa, qya, qza), tn_b = fma_dot(e2x, e2y, e2z, qxb, qyb, qzb);
                const float xva = ok_a ? as_f(as_u(vn_a) ^ sg_a) : 1.0f, xta = ok_a ? as_f(as_u(tn_a) ^ sg_a) : 1.0f;
                const float xvb = ok_b ? as_f(as_u(vn_b) ^ sg_b) : 1.0f, xtb = ok_b ? as_f(as_u(tn_b) ^ sg_b) : 1.0f;
                if ((__ballot(!(fminf(fminf(xua, xva), xta) <= thr_a)) | __ballot(!(fminf(fminf(xub, xvb), xtb) <= thr_b))) == 0) return;
                const float inv_a = 1.0f / det_a, inv_b = 1.0f / det_b;
                const float u_a = inv_a * un_a, v_a = inv_a * vn_a, t_a = inv_a * tn_a;
                const float u_b = inv_b * un_b, v_b = inv_b * vn_b, t_b = inv_b * tn_b;
                const bool acc_a = (u_a >= 0.0f) & (v_a >= 0.0f) & ((u_a + v_a) <= 1.0f) & (t_a >= 0.0f) & (t_a < lim_a);
                const bool acc_b = (u_b >= 0.0f) & (v_b >= 0.0f) & ((u_b + v_b) <= 1.0f) & (t_b >= 0.0f) & (t_b < lim_b);
                best_a = acc_a ? t_a : best_a; lim_a = acc_a ? t_a : lim_a; ua = acc_a ? u_a : ua; va = acc_a ? v_a : va; prim_a = acc_a ? tri : prim_a;
                best_b = acc_b ? t_b : best_b; lim_b = acc_b ? t_b : lim_b; ub = acc_b ? u_b : ub; vb = acc_b ? v_b : vb; prim_b = acc_b ? tri : prim_b;
                changed |= __ballot(acc_a) | __ballot(acc_b);
            };
            float a0 = tp[0], a1 = tp[1], a2 = tp[2], a3 = tp[3], a4 = tp[4], a5 = tp[5], a6 = tp[6], a7 = tp[7], a8 = tp[8];
            uint32_t tri = first * 8u;
            for (uint32_t p = n_real >> 1; p > 0; p--) {
                const float b0 = tp[9], b1 = tp[10], b2 = tp[11], b3 = tp[12], b4 = tp[13], b5 = tp[14], b6 = tp[15], b7 = tp[16], b8 = tp[17];
                test(a0, a1, a2, a3, a4, a5, a6, a7, a8, tri);
                a0 = tp[18]; a1 = tp[19]; a2 = tp[20]; a3 = tp[21]; a4 = tp[22]; a5 = tp[23]; a6 = tp[24]; a7 = tp[25]; a8 = tp[26];
                test(b0, b1, b2, b3, b4, b5, b6, b7, b8, tri + 1u);
                tri += 2u;
                tp += 2 * kTriDwords;
            }
            if (n_real & 1u) test(a0, a1, a2, a3, a4, a5, a6, a7, a8, tri);
            if (changed != 0) stale_top = sp;
        }
    }
    hit_a.t = best_a; hit_a.u = ua; hit_a.v = va; hit_a.prim = prim_a;
    hit_b.t = best_b; hit_b.u = ub; hit_b.v = vb; hit_b.prim = prim_b;
}

__device__ __forceinline__ void trace_packet2(const DevScene& sc, const Ray& ra, const Ray& rb, bool act_a, bool act_b, PacketHit& ha,
                                              PacketHit& hb) {
    const bool slow = (act_a && (fabsf(ra.ix) == INFINITY || fabsf(ra.iy) == INFINITY || fabsf(ra.iz) == INFINITY)) ||
                      (act_b && (fabsf(rb.ix) == INFINITY || fabsf(rb.iy) == INFINITY || fabsf(rb.iz) == INFINITY));
    if (__ballot(slow) != 0) {
        trace_packet2_impl<2, -1>(sc, ra, rb, act_a, act_b, ha, hb);
        return;
    }
    if (sc.boxes_ordered) {
        // wave-uniform octant over BOTH ray sets
        const uint64_t am = __ballot(act_a), bm = __ballot(act_b);
        const uint64_t nxa = __ballot(act_a && ra.ix < 0.0f), nya = __ballot(act_a && ra.iy < 0.0f), nza = __ballot(act_a && ra.iz < 0.0f);
        const uint64_t nxb = __ballot(act_b && rb.ix < 0.0f), nyb = __ballot(act_b && rb.iy < 0.0f), nzb = __ballot(act_b && rb.iz < 0.0f);
        const bool xn = (nxa | nxb) != 0, yn = (nya | nyb) != 0, zn = (nza | nzb) != 0;
        const bool xu = !xn || (nxa == am && nxb == bm), yu = !yn || (nya == am && nyb == bm), zu = !zn || (nza == am && nzb == bm);
        if (xu && yu && zu) {
            switch ((xn ? 1 : 0) | (yn ? 2 : 0) | (zn ? 4 : 0)) {
                case 0: trace_packet2_impl<1, 0>(sc, ra, rb, act_a, act_b, ha, hb); return;
                case 1: trace_packet2_impl<1, 1>(sc, ra, rb, act_a, act_b, ha, hb); return;
                case 2: trace_packet2_impl<1, 2>(sc, ra, rb, act_a, act_b, ha, hb); return;
                case 3: trace_packet2_impl<1, 3>(sc, ra, rb, act_a, act_b, ha, hb); return;
                case 4: trace_packet2_impl<1, 4>(sc, ra, rb, act_a, act_b, ha, hb); return;
                case 5: trace_packet2_impl<1, 5>(sc, ra, rb, act_a, act_b, ha, hb); return;
                case 6: trace_packet2_impl<1, 6>(sc, ra, rb, act_a, act_b, ha, hb); return;
                default: trace_packet2_impl<1, 7>(sc, ra, rb, act_a, act_b, ha, hb); return;
            }
        }
    }
    trace_packet2_impl<1, -1>(sc, ra, rb, act_a, act_b, ha, hb);
}

// pixel_sum += sample, strictly in sample order (worker.rs:41-43), for the S samples of a pixel held by S consecutive lanes.
// S = 16 is one DPP row: v_add_f32_dpp with row_newbcast:j adds lane j of the row in ONE instruction (no LDS permute); every
// lane of the row ends with the same sum.  Inactive samples contribute +0.0 (exact).
template <int J, int N>
struct RowSum {
    static __device__ __forceinline__ void run(float& acc, float c) {
        acc += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c), 0x150 + J, 0xF, 0xF, false));
        RowSum<J + 1, N>::run(acc, c);
    }
};
template <int N>
struct RowSum<N, N> {
    static __device__ __forceinline__ void run(float&, float) {}
};
template <int S>
__device__ __forceinline__ void add_samples_in_order(float& acc, float c, int lane) {
    if (S == 1) {
        acc += c;
    } else if (S == 16) {
        RowSum<0, 16>::run(acc, c);
    } else if (S == 32) {
        // a pixel = two rows: every row adds its own 16 samples to the incoming sum (right for the even rows), row_bcast15 hands
        // the even rows' result to the odd rows, which add their 16 samples (the even rows redo theirs: discarded), and the
        // pixel's last lane holds the sum that all 32 lanes take over
        RowSum<0, 16>::run(acc, c);
        acc = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(acc), __float_as_int(acc), 0x142, 0xA, 0xF, false));
        RowSum<0, 16>::run(acc, c);
        const float lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 31));
        const float hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 63));
        acc = lane < 32 ? lo : hi;
    } else {
#pragma unroll
        for (int j = 0; j < S; j++) acc += __shfl(c, (lane & ~(S - 1)) + j);
    }
}

// Fused tile render on ray packets.  A wave owns a block of 64/S pixels and shoots S consecutive samples of each
// pixel per pass (lane = pixel*S + sub-sample): all 64 rays of a pass are neighbours on the film, so the packet stays
// coherent, while the work unit (block x all samples) shrinks with S, which evens out the load.  pixel_sum is
// accumulated strictly in sample order (worker.rs:41-43), redundantly by every lane of the pixel (add_samples_in_order).
// WPE = waves per SIMD the register allocation is held to: 8 (64 VGPRs) hides the scalar-cache misses of scenes that
// outgrow it, 7 (72 VGPRs) schedules slightly better when the scene stays cache resident (profiles/r01_notes.md).
template <int S, bool LDS_STACK, int WPE, bool OBJ = false, bool MCACHE = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, 8))) void render_tiles_packet_kernel(RenderParams) {
    extern __shared__ __align__(16) unsigned char smem[];
    MP_KERNEL_PARAMS;
    constexpr int BW = (S <= 2) ? 8 : (S <= 8) ? 4 : (S <= 32) ? 2 : 1;  // pixel block = BW x BH, BW*BH*S == 64
    constexpr int BH = 64 / S / BW;
    const int lane = static_cast<int>(threadIdx.x) & 63;
    const int pix = lane / S, sub = lane % S;
    uint32_t qstate = blockIdx.x % kWorkQueues;
    // per-unit mask cache of the packet-level child rejection (MaskCache): this wave's header + entries in LDS
    MaskCache mc{nullptr};
    if (MCACHE) mc.lds = reinterpret_cast<uint32_t*>(smem) + static_cast<size_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6)) * kMaskCacheDwords;
    for (;;) {
        const RenderParams& P = params_view(KP);  // unit setup
        const uint32_t ts = P.tile_size;
        const uint32_t bx = (ts + BW - 1) / BW, by = (ts + BH - 1) / BH, upt = bx * by, total = P.n_tiles * upt;
        MP_NEXT_UNIT(unit)
        const uint32_t b = unit % upt;
        const uint32_t tile_i = P.tile_order ? P.tile_order[unit / upt] : unit / upt;
        const uint64_t t_unit = P.tile_cost ? __builtin_readcyclecounter() : 0;
        const mp_block T = P.tiles[tile_i];
        const uint32_t px = T.min_x + (b % bx) * BW + static_cast<uint32_t>(pix % BW);
        const uint32_t py = T.min_y + (b / bx) * BH + static_cast<uint32_t>(pix / BW);
        const bool inpix = px < T.max_x && py < T.max_y;
        if (__ballot(inpix) == 0) continue;
        const size_t off = (static_cast<size_t>(tile_i) * ts * ts + static_cast<size_t>(py - T.min_y) * ts + (px - T.min_x)) * 4;
        float acc, cnt;  // pixel_sum (r=g=b) and alpha (worker.rs:40)
        pixel_state_load(P, off, inpix, sub == 0, acc, cnt);
        if (MCACHE) {  // a new unit: other pixels, other bounds
            if (lane == 0) mc.lds[12] = 0xFFFFFFFFu;
            wave_lds_sync();
        }
        // passes are aligned to multiples of S in the absolute sample index, so that a chunk boundary (MP_FLAG_CHUNKED_SUM) never
        // falls inside a pass; lanes outside [s_begin, s_end) add +0.0, which is exact
        const uint32_t s_begin = P.s_begin, s_end = P.s_end;
        for (uint32_t s0 = s_begin & ~static_cast<uint32_t>(S - 1); s0 < s_end; s0 += S) {
            const uint32_t s = s0 + static_cast<uint32_t>(sub);
            const bool act = inpix && s >= s_begin && s < s_end;
            Ray r;
            r.ox = r.oy = r.oz = r.dx = r.dy = r.dz = r.ix = r.iy = r.iz = 0.0f;
            {
                const RenderParams& G = params_view(KP);  // ray generation
                if (act) sample_ray(G.gen, px, py, s, r);
            }
            PacketHit h;
            h.t = FLT_MAX; h.u = h.v = 0.0f; h.prim = kNoPrim;
            uint32_t hinst = 0u;
            {
                const RenderParams& W = params_view(KP);  // walk
                float* lds = reinterpret_cast<float*>(smem + static_cast<size_t>(static_cast<int>(threadIdx.x) >> 6) * W.lds_per_wave);
                if (OBJ) {  // object group: the packet walks every member's tree in turn
                    if (LDS_STACK) {
                        HybridStack st(lds, lane, W.scene.stack_cap, W.scene.packet_stack_regs);
                        trace_packet_objects<false>(W.scene, r, act, st, h, hinst);
                    } else {
                        RegStack st(lds, lane);
                        trace_packet_objects<(S >= 8 && S <= 32)>(W.scene, r, act, st, h, hinst);
                    }
                } else {
#ifdef MP_PROF_NOWALK  // profiling builds only (tools/build_variant.sh): what a pass costs without its walk
                    const bool go = false;
#else
                    const bool go = act && W.scene.kind == 0u && may_hit_scene(W.scene, r);
#endif
                    if (__ballot(go) != 0) {
                        if (LDS_STACK) {
                            HybridStack st(lds, lane, W.scene.stack_cap, W.scene.packet_stack_regs);
                            trace_packet<false>(W.scene, r, go, st, h);
                        } else {
                            RegStack st(lds, lane);
                            trace_packet<((S >= 8 && S <= 32) || MCACHE), RegStack, MCACHE>(W.scene, r, go, st, h, mc);
                        }
                    }
                }
            }
            float c = 0.0f;
            bool hit = h.prim != kNoPrim;
            const RenderParams& H = params_view(KP);  // shading + accumulation
            if (hit) {
#ifdef MP_PROF_NOSHADE  // profiling builds only
                c = h.u;
#else
                float nn[3];
                if (OBJ) object_normal(H.scene, hinst, r, h.prim, h.u, h.v, nn);
                else resolve_normal(H.scene, h.prim, h.u, h.v, nn);
                c = fabsf(r.dx * nn[0] + r.dy * nn[1] + r.dz * nn[2]);  // worker.rs:60
#endif
            } else if (H.scene.kind == 1u) {  // Scene<Sphere>
                float ts_, nn[3];
                hit = act && sphere_intersect(H.scene, r, ts_, nn);
                if (hit) c = fabsf(r.dx * nn[0] + r.dy * nn[1] + r.dz * nn[2]);
            }
            // alpha sums 1.0 per hit: an exact integer in f32, so the order is irrelevant
            {   // (the mask of this pixel's lanes is rebuilt from the lane id here: two registers less across the walk)
                int l_ = lane;
                asm volatile("" : "+v"(l_));
                const uint64_t pixel_lanes = (S == 64 ? ~0ull : ((1ull << (S & 63)) - 1ull)) << (l_ & ~(S - 1));
                cnt += static_cast<float>(__popcll(__ballot(hit) & pixel_lanes));
            }
            add_samples_in_order<S>(acc, c, lane);  // misses add +0.0 (exact)
            if (H.chunked && ((s0 + S) & (kSumChunk - 1u)) == 0u && s0 + S <= s_end) chunk_flush(H, off, inpix && sub == 0, acc);
        }
        const RenderParams& E = params_view(KP);  // unit end
        if (inpix && sub == 0) pixel_state_store(E, off, acc, cnt);
        if (E.tile_cost && lane == 0) atomicAdd(E.tile_cost + tile_i, static_cast<unsigned long long>(__builtin_readcyclecounter() - t_unit));
    }
}

// Fused tile render on 128-ray packets (two rays per lane, trace_packet2).  A wave owns a 4x2 block of pixels and shoots 16
// consecutive samples of each pixel per pass: lane = (pixel of the left 2x2 half) * 16 + sub-sample carries that sample of its
// pixel (ray A) and of the pixel two columns to the right (ray B).  Reference semantics only (the path kernel has its own loop);
// TriangleBvh scenes whose stack bound fits the register stack.
template <int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, 8))) void render_tiles_packet2_kernel(RenderParams P) {
    constexpr int S = 16;
    const int lane = static_cast<int>(threadIdx.x) & 63;
    const int pix = lane / S, sub = lane % S;
    const uint64_t pixel_lanes = ((1ull << S) - 1ull) << (lane & ~(S - 1));
    const uint32_t ts = P.tile_size;
    const uint32_t bx = (ts + 3) / 4, by = (ts + 1) / 2, upt = bx * by, total = P.n_tiles * upt;
    uint32_t qstate = blockIdx.x % kWorkQueues;
    for (;;) {
        MP_NEXT_UNIT(unit)
        const uint32_t b = unit % upt;
        const uint32_t tile_i = P.tile_order ? P.tile_order[unit / upt] : unit / upt;
        const uint64_t t_unit = P.tile_cost ? __builtin_readcyclecounter() : 0;
        const mp_block T = P.tiles[tile_i];
        const uint32_t px_a = T.min_x + (b % bx) * 4 + static_cast<uint32_t>(pix % 2), px_b = px_a + 2u;
        const uint32_t py = T.min_y + (b / bx) * 2 + static_cast<uint32_t>(pix / 2);
        const bool in_a = px_a < T.max_x && py < T.max_y, in_b = px_b < T.max_x && py < T.max_y;
        if (__ballot(in_a) == 0) continue;  // B lies to the right of A: no A pixel, no B pixel
        const size_t off_a = (static_cast<size_t>(tile_i) * ts * ts + static_cast<size_t>(py - T.min_y) * ts + (px_a - T.min_x)) * 4;
        const size_t off_b = off_a + 8;
        float acc_a, cnt_a, acc_b, cnt_b;  // pixel_sum (r=g=b) and alpha (worker.rs:40)
        pixel_state_load(P, off_a, in_a, sub == 0, acc_a, cnt_a);
        pixel_state_load(P, off_b, in_b, sub == 0, acc_b, cnt_b);
        for (uint32_t s0 = P.s_begin & ~static_cast<uint32_t>(S - 1); s0 < P.s_end; s0 += S) {
            const uint32_t s = s0 + static_cast<uint32_t>(sub);
            const bool in_pass = s >= P.s_begin && s < P.s_end;
            const bool act_a = in_a && in_pass, act_b = in_b && in_pass;
            Ray ra, rb;
            ra.ox = ra.oy = ra.oz = ra.dx = ra.dy = ra.dz = ra.ix = ra.iy = ra.iz = 0.0f;
            rb = ra;
            if (act_a) sample_ray(P.gen, px_a, py, s, ra);
            if (act_b) sample_ray(P.gen, px_b, py, s, rb);
            const bool go_a = act_a && may_hit_scene(P.scene, ra), go_b = act_b && may_hit_scene(P.scene, rb);
            PacketHit ha, hb;
            ha.t = FLT_MAX; ha.u = ha.v = 0.0f; ha.prim = kNoPrim;
            hb = ha;
            if ((__ballot(go_a) | __ballot(go_b)) != 0) trace_packet2(P.scene, ra, rb, go_a, go_b, ha, hb);
            float c_a = 0.0f, c_b = 0.0f;
            const bool hit_a = ha.prim != kNoPrim, hit_b = hb.prim != kNoPrim;
            if (hit_a) {
                float nn[3];
                resolve_normal(P.scene, ha.prim, ha.u, ha.v, nn);
                c_a = fabsf(ra.dx * nn[0] + ra.dy * nn[1] + ra.dz * nn[2]);  // worker.rs:60
            }
            if (hit_b) {
                float nn[3];
                resolve_normal(P.scene, hb.prim, hb.u, hb.v, nn);
                c_b = fabsf(rb.dx * nn[0] + rb.dy * nn[1] + rb.dz * nn[2]);
            }
            cnt_a += static_cast<float>(__popcll(__ballot(hit_a) & pixel_lanes));
            cnt_b += static_cast<float>(__popcll(__ballot(hit_b) & pixel_lanes));
            add_samples_in_order<S>(acc_a, c_a, lane);  // misses add +0.0 (exact)
            add_samples_in_order<S>(acc_b, c_b, lane);
            if (P.chunked && ((s0 + S) & (kSumChunk - 1u)) == 0u && s0 + S <= P.s_end) {
                chunk_flush(P, off_a, in_a && sub == 0, acc_a);
                chunk_flush(P, off_b, in_b && sub == 0, acc_b);
            }
        }
        if (sub == 0) {
            if (in_a) pixel_state_store(P, off_a, acc_a, cnt_a);
            if (in_b) pixel_state_store(P, off_b, acc_b, cnt_b);
        }
        if (P.tile_cost && lane == 0) atomicAdd(P.tile_cost + tile_i, static_cast<unsigned long long>(__builtin_readcyclecounter() - t_unit));
    }
}

// ---- build-defined path extension (MP_FLAG_PATHS; the reference has no bounce loop) -----------------
// Diffuse grey surfaces {albedo, emission} indexed by TriangleShadingData.material under a uniform sky (defaults: one material
// {0.75, 0}, sky 1), at most max_depth segments per path; the operations and their order are those of the oracle's
// render_sample_paths_impl (only + - * / sqrt and compares => bit-identical).
// Camera rays are coherent and go through the packet walk; bounce rays are incoherent: the lanes whose path is still
// alive are compacted (ballot + mbcnt) into the wave's LDS ray queue and traced by the 8-lane-group traversal.
constexpr float kPathEps = 1e-4f;

// HitRecord.texture_coords (geometry/mod.rs:78-79; BarycentricCoordinates::interpolate at ray_bvh_intersection.rs:80-83), x and y
__device__ __forceinline__ void hit_tex(const DevScene& so, uint32_t prim, float u, float v, float& tx, float& ty) {
    const uint32_t* vi = so.vidx + static_cast<size_t>(prim) * 3;
    const float *t0 = so.vtex + 3 * static_cast<size_t>(vi[0]), *t1 = so.vtex + 3 * static_cast<size_t>(vi[1]),
                *t2 = so.vtex + 3 * static_cast<size_t>(vi[2]);
    const float w = 1.0f - u - v;
    tx = t0[0] * w + t1[0] * u + t2[0] * v;
    ty = t0[1] * w + t1[1] * u + t2[1] * v;
}

// One path vertex of the build-defined extension (oracle: render_sample_paths_impl), shared by the fused and the staged kernels so
// that both evaluate the very same operations: `h` is the closest hit of segment `depth` along `r`.  Updates L / thr, and either
// ends the path (returns false) or replaces `r` by the bounce ray drawn from `rng` (returns true).
// N = colour channels carried: 3 when some material of the table is coloured or textured (DevScene::materials_rgb), else 1 -- a grey
// table makes the three channels the same number, so one is computed (same bits).  Material record = mp_material: albedo rgb,
// emission rgb, albedo2 rgb, texture, texture_scale.
template <bool OBJ, int N>
__device__ __forceinline__ bool path_vertex(const DevScene& sc, const PacketHit& h, uint32_t depth, uint32_t max_depth, Rng& rng,
                                            Ray& r, float (&L)[N], float (&thr)[N], bool& primary_hit, uint32_t inst = 0u) {
    if (h.prim == kNoPrim) {
#pragma unroll
        for (int c = 0; c < N; c++) L[c] = L[c] + thr[c] * sc.sky;
        return false;
    }
    if (depth == 1) primary_hit = true;
    float n[3];
    const uint32_t mat = OBJ ? object_normal(sc, inst, r, h.prim, h.u, h.v, n) : resolve_normal(sc, h.prim, h.u, h.v, n);
    const float* m = sc.materials + static_cast<size_t>(mat) * 12;
#pragma unroll
    for (int c = 0; c < N; c++) L[c] = L[c] + thr[c] * m[3 + c];
    const float dn = r.dx * n[0] + r.dy * n[1] + r.dz * n[2];
    if (dn > 0.0f) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
    const float* alb = m;
    if (N == 3 && as_u(m[9]) == MP_TEXTURE_CHECKER) {  // reads HitRecord.texture_coords; a Sphere member's are the origin (primitives.rs:45)
        float tx = 0.0f, ty = 0.0f;
        if (OBJ) {
            DevScene so;
            object_scene(sc, inst, so);
            if (so.kind == 0u) hit_tex(so, h.prim, h.u, h.v, tx, ty);
        } else {
            hit_tex(sc, h.prim, h.u, h.v, tx, ty);
        }
        const float cell = floorf(tx * m[10]) + floorf(ty * m[10]);
        const float half = cell * 0.5f;
        if (half - floorf(half) != 0.0f) alb = m + 6;  // odd cell (NaN counts as odd): albedo2
    }
#pragma unroll
    for (int c = 0; c < N; c++) thr[c] = thr[c] * alb[c];
    if (depth == max_depth) return false;
    const float hx = r.ox + r.dx * h.t, hy = r.oy + r.dy * h.t, hz = r.oz + r.dz * h.t;  // geometry/mod.rs:56-58
    float x1, x2;
    unit_disc(rng, x1, x2);
    const float z = sqrtf(1.0f - (x1 * x1 + x2 * x2));
    const float sign = copysignf(1.0f, n[2]);
    const float a = -1.0f / (sign + n[2]);
    const float bb = n[0] * n[1] * a;
    const float t0 = 1.0f + sign * n[0] * n[0] * a, t1 = sign * bb, t2 = -sign * n[0];
    const float b0 = bb, b1 = sign + n[1] * n[1] * a, b2 = -n[1];
    ray_new(hx + n[0] * kPathEps, hy + n[1] * kPathEps, hz + n[2] * kPathEps, t0 * x1 + b0 * x2 + n[0] * z,
            t1 * x1 + b1 * x2 + n[1] * z, t2 * x1 + b2 * x2 + n[2] * z, r);
    return true;
}

// worker.rs:40-44 for three colour channels (coloured / textured material tables; never with MP_FLAG_CHUNKED_SUM): the slot holds
// {sum r, sum g, sum b, hit count} between MP_FLAG_ACCUMULATE launches
__device__ __forceinline__ void pixel_state_load3(const RenderParams& P, size_t off, bool inpix, float (&acc)[3], float& cnt) {
    acc[0] = acc[1] = acc[2] = 0.0f;
    cnt = 0.0f;
    if (P.carry_in && inpix) {
        const float4 prev = *reinterpret_cast<const float4*>(P.out + off);
        acc[0] = prev.x; acc[1] = prev.y; acc[2] = prev.z;
        cnt = prev.w;
    }
}
__device__ __forceinline__ void pixel_state_store3(const RenderParams& P, size_t off, const float (&acc)[3], float cnt) {
    const float s = P.finalize ? P.inv_spp : 1.0f;  // worker.rs:44 ; x * 1.0f is x
    *reinterpret_cast<float4*>(P.out + off) = make_float4(acc[0] * s, acc[1] * s, acc[2] * s, cnt * s);
}

#ifndef MP_PATHS_WPE
#define MP_PATHS_WPE 6  // waves per SIMD the path kernel's registers are held to (A/B-measured, profiles/r03_notes.md)
#endif
// MCACHE: the camera pass runs the cached packet walk (MaskCache; the wave's header + tables are the last kMaskCacheDwords dwords of
// its LDS region)
template <int S, bool OBJ, bool RGB, bool MCACHE = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MP_PATHS_WPE, 8))) void render_paths_kernel(RenderParams) {
    constexpr int N = RGB ? 3 : 1;
    extern __shared__ __align__(16) unsigned char smem[];
    MP_KERNEL_PARAMS;  // RenderParams through short-lived views (params_view)
    MaskCache mc{nullptr};
    if (MCACHE) {
        const RenderParams& P0 = params_view(KP);
        mc.lds = reinterpret_cast<uint32_t*>(smem + static_cast<size_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6) + 1) * P0.lds_per_wave) - kMaskCacheDwords;
    }
    constexpr int BW = (S <= 2) ? 8 : (S <= 8) ? 4 : 2;
    constexpr int BH = 64 / S / BW;
    const int lane = static_cast<int>(threadIdx.x) & 63, wave = static_cast<int>(threadIdx.x) >> 6;
    const int pix = lane / S, sub = lane % S;
    unsigned long long segs = 0;  // wave-uniform
    uint32_t qstate = blockIdx.x % kWorkQueues;
    for (;;) {
        const RenderParams& P = params_view(KP);  // unit setup
        const uint32_t ts = P.tile_size;
        const uint32_t bx = (ts + BW - 1) / BW, by = (ts + BH - 1) / BH, upt = bx * by, total = P.n_tiles * upt;
        MP_NEXT_UNIT(unit)
        const uint32_t b = unit % upt;
        const uint32_t tile_i = P.tile_order ? P.tile_order[unit / upt] : unit / upt;
        const uint64_t t_unit = P.tile_cost ? __builtin_readcyclecounter() : 0;
        const mp_block T = P.tiles[tile_i];
        const uint32_t px = T.min_x + (b % bx) * BW + static_cast<uint32_t>(pix % BW);
        const uint32_t py = T.min_y + (b / bx) * BH + static_cast<uint32_t>(pix / BW);
        const bool inpix = px < T.max_x && py < T.max_y;
        if (__ballot(inpix) == 0) continue;
        const size_t off = (static_cast<size_t>(tile_i) * ts * ts + static_cast<size_t>(py - T.min_y) * ts + (px - T.min_x)) * 4;
        float acc[N], cnt;  // pixel_sum (r=g=b for a grey table: one channel) and alpha (worker.rs:40)
        if (RGB) pixel_state_load3(P, off, inpix, reinterpret_cast<float (&)[3]>(acc[0]), cnt);
        else pixel_state_load(P, off, inpix, sub == 0, acc[0], cnt);
        if (MCACHE) {  // a new unit: other pixels, other bounds
            if (lane == 0) mc.lds[12] = 0xFFFFFFFFu;
            wave_lds_sync();
        }
        // passes are aligned to multiples of S in the absolute sample index, so that a chunk boundary (MP_FLAG_CHUNKED_SUM) never
        // falls inside a pass; lanes outside [s_begin, s_end) add +0.0, which is exact
        const uint32_t s_begin = P.s_begin, s_end = P.s_end, max_depth = P.max_depth;
        for (uint32_t s0 = s_begin & ~static_cast<uint32_t>(S - 1); s0 < s_end; s0 += S) {
            const uint32_t s = s0 + static_cast<uint32_t>(sub);
            const bool act = inpix && s >= s_begin && s < s_end;
            Rng rng;
            rng.s0 = rng.s1 = rng.s2 = rng.s3 = 0;
            Ray r;
            r.ox = r.oy = r.oz = r.dx = r.dy = r.dz = r.ix = r.iy = r.iz = 0.0f;
            {
                const RenderParams& G = params_view(KP);  // ray generation
                if (act) {
                    rng_seed(rng, sample_key(G.gen, px, py, s));
                    sample_ray_rng(G.gen, px, py, rng, r);
                }
            }
            float L[N], thr[N];
#pragma unroll
            for (int c = 0; c < N; c++) { L[c] = 0.0f; thr[c] = 1.0f; }
            bool alive = act, primary_hit = false;
            PacketHit h;
            for (uint32_t depth = 1; depth <= max_depth; depth++) {
                const uint64_t alive_m = __ballot(alive);
                if (alive_m == 0) break;
                segs += static_cast<unsigned long long>(__popcll(alive_m));
                h.t = FLT_MAX; h.u = h.v = 0.0f; h.prim = kNoPrim;
                uint32_t hinst = 0u;
                {
                    const RenderParams& W = params_view(KP);  // walk
                    float* q = reinterpret_cast<float*>(smem + static_cast<size_t>(wave) * W.lds_per_wave);
                    uint2* stack = reinterpret_cast<uint2*>(q + kQueueFloats);
                    if (depth == 1 && OBJ) {  // camera rays of an object group: one packet walk per member
                        RegStack rst(nullptr, lane);
                        HybridStack hst(reinterpret_cast<float*>(stack), lane, W.scene.stack_cap, W.scene.packet_stack_regs);
                        if (W.scene.stack_cap > W.scene.packet_stack_regs) trace_packet_objects<false>(W.scene, r, alive, hst, h, hinst);
                        else trace_packet_objects<(S >= 8)>(W.scene, r, alive, rst, h, hinst);
                    } else if (depth == 1) {
                        const bool go = alive && may_hit_scene(W.scene, r);
                        if (__ballot(go) != 0) {
                            RegStack rst(nullptr, lane);
                            HybridStack hst(reinterpret_cast<float*>(stack), lane, W.scene.stack_cap, W.scene.packet_stack_regs);
                            if (W.scene.stack_cap > W.scene.packet_stack_regs) trace_packet<false>(W.scene, r, go, hst, h);
                            else trace_packet<(S >= 8), RegStack, MCACHE>(W.scene, r, go, rst, h, mc);
                        }
                    } else {
                        // bounce rays: group walk (once per member of an object group)
                        GroupHit gh;
                        trace_objects<OBJ>(W.scene, r, alive, q, stack, gh);
                        h.t = gh.t; h.u = gh.u; h.v = gh.v; h.prim = gh.prim;
                        hinst = gh.inst;
                    }
                }
                const RenderParams& V = params_view(KP);  // shade + bounce
                if (alive) alive = path_vertex<OBJ, N>(V.scene, h, depth, max_depth, rng, r, L, thr, primary_hit, hinst);
            }
            {   // (the mask of this pixel's lanes is rebuilt from the lane id here: two registers less across the walks)
                int l_ = lane;
                asm volatile("" : "+v"(l_));
                const uint64_t pixel_lanes = (S == 64 ? ~0ull : ((1ull << (S & 63)) - 1ull)) << (l_ & ~(S - 1));
                cnt += static_cast<float>(__popcll(__ballot(primary_hit) & pixel_lanes));
            }
#pragma unroll
            for (int c = 0; c < N; c++) add_samples_in_order<S>(acc[c], L[c], lane);
            const RenderParams& A = params_view(KP);  // accumulation
            if (!RGB && A.chunked && ((s0 + S) & (kSumChunk - 1u)) == 0u && s0 + S <= s_end) chunk_flush(A, off, inpix && sub == 0, acc[0]);
        }
        const RenderParams& E = params_view(KP);  // unit end
        if (inpix && sub == 0) {
            if (RGB) pixel_state_store3(E, off, reinterpret_cast<const float (&)[3]>(acc[0]), cnt);
            else pixel_state_store(E, off, acc[0], cnt);
        }
        if (E.tile_cost && lane == 0) atomicAdd(E.tile_cost + tile_i, static_cast<unsigned long long>(__builtin_readcyclecounter() - t_unit));
    }
    const RenderParams& Z = params_view(KP);
    if (lane == 0 && Z.segments && segs) atomicAdd(Z.segments, segs);
}

// ---- pooled form of render_paths_kernel (round 3) ----------------------------------------------------------------------------
// The fused kernel above traces the bounce rays of ONE pass (8 pixels x 8 samples = at most 64 rays) per call of the 8-lane-group
// walk, and a ray's walk takes 10 to 80 steps: when the queue runs dry the groups idle until the slowest ray of the call finishes
// -- 13 % of the group-iterations of a 64-ray call on the stand-in, 6.5 % of a 256-ray call (tools/sim_collapse.py's traces;
// profiles/r03_notes.md).  Here a wave keeps NSUB passes in flight: per bounce, the passes are shaded one after the other, lane =
// path, and their rays are compacted into ONE queue of up to 64 * NSUB rays that the walk drains in a single call.  The state of
// the passes that are not being shaded -- RNG, radiance, throughput, flags: 11 dwords per path -- and the queue itself (origin,
// direction, inverse direction, hit: 13 dwords per ray) live in a per-wave slab of global memory (RenderParams::pool, 24 dwords
// x 64 * NSUB per wave, L2-resident): the "rays laid out SoA + stream compaction" of the north star, inside one persistent wave.
// Nothing of a path is live in registers across the walk, so the kernel fits 64 VGPRs = 8 waves per SIMD.
// Per path the operations, their order and the RNG stream are those of render_paths_kernel (same path_vertex, same walks, samples
// added in index order): the frame is bit-identical (tests run both kernels on the same cases).
#ifndef MP_POOL_WPE
#define MP_POOL_WPE 8
#endif
constexpr int kPoolQueueRows = 13, kPoolStateRows = 11;
__host__ __device__ constexpr uint32_t pool_floats_per_wave(int nsub) { return static_cast<uint32_t>((kPoolQueueRows + kPoolStateRows) * 64 * nsub); }
constexpr uint32_t kPoolAlive = 1u, kPoolPrimary = 2u, kPoolQueued = 4u;  // flags word: bits 0..2, queue slot << 8

template <int NSUB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MP_POOL_WPE, 8))) void render_paths_pooled_kernel(RenderParams) {
    extern __shared__ __align__(16) unsigned char smem[];
    MP_KERNEL_PARAMS;
    constexpr int S = 8, BW = 4, BH = 2, QN = 64 * NSUB;
    const int lane = static_cast<int>(threadIdx.x) & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);  // uniform: the pool / stack pointers stay in SGPRs
    const int pix = lane / S, sub = lane % S;
    unsigned long long segs = 0;  // wave-uniform
    uint32_t qstate = blockIdx.x % kWorkQueues;
    float* q;        // ray queue: kPoolQueueRows rows x QN slots
    uint32_t* park;  // parked path state: kPoolStateRows rows x QN (slot = pass * 64 + lane): rng 8, L, thr, flags
    uint2* stack;
    {
        const RenderParams& P0 = params_view(KP);
        q = P0.pool + static_cast<size_t>(blockIdx.x * 4u + static_cast<uint32_t>(wave)) * P0.pool_stride;
        park = reinterpret_cast<uint32_t*>(q + kPoolQueueRows * QN);
        stack = reinterpret_cast<uint2*>(smem + static_cast<size_t>(wave) * P0.lds_per_wave);
    }
    auto park_store = [&](int j, const Rng& rng, float L, float thr, uint32_t flags) {
        // streamed (non-temporal): 200 MB of parked state per launch must not push the scene out of the L2
        uint32_t* p = park + j * 64 + lane;
        __builtin_nontemporal_store(static_cast<uint32_t>(rng.s0), p + 0 * QN); __builtin_nontemporal_store(static_cast<uint32_t>(rng.s0 >> 32), p + 1 * QN);
        __builtin_nontemporal_store(static_cast<uint32_t>(rng.s1), p + 2 * QN); __builtin_nontemporal_store(static_cast<uint32_t>(rng.s1 >> 32), p + 3 * QN);
        __builtin_nontemporal_store(static_cast<uint32_t>(rng.s2), p + 4 * QN); __builtin_nontemporal_store(static_cast<uint32_t>(rng.s2 >> 32), p + 5 * QN);
        __builtin_nontemporal_store(static_cast<uint32_t>(rng.s3), p + 6 * QN); __builtin_nontemporal_store(static_cast<uint32_t>(rng.s3 >> 32), p + 7 * QN);
        __builtin_nontemporal_store(as_u(L), p + 8 * QN); __builtin_nontemporal_store(as_u(thr), p + 9 * QN); __builtin_nontemporal_store(flags, p + 10 * QN);
    };
    // bounce ray of a path that is still alive: into the queue (compacted over the lanes whose ray can reach the scene at all)
    auto push_ray = [&](const DevScene& sc, const Ray& r, bool alive, uint32_t& nq, uint32_t& flags) {
        const bool queued = alive && may_hit_scene(sc, r);
        const uint64_t am = __ballot(queued);
        const uint32_t slot = nq + static_cast<uint32_t>(rank_below(am));
        if (queued) {
            q[0 * QN + slot] = r.ox; q[1 * QN + slot] = r.oy; q[2 * QN + slot] = r.oz;
            q[3 * QN + slot] = r.dx; q[4 * QN + slot] = r.dy; q[5 * QN + slot] = r.dz;
            q[6 * QN + slot] = r.ix; q[7 * QN + slot] = r.iy; q[8 * QN + slot] = r.iz;
            flags |= kPoolQueued | (slot << 8);
        }
        nq += static_cast<uint32_t>(__popcll(am));
    };
    for (;;) {
        const RenderParams& P = params_view(KP);  // unit setup
        const uint32_t ts = P.tile_size;
        const uint32_t bx = (ts + BW - 1) / BW, by = (ts + BH - 1) / BH, upt = bx * by, total = P.n_tiles * upt;
        MP_NEXT_UNIT(unit)
        const uint32_t b = unit % upt;
        const uint32_t tile_i = P.tile_order ? P.tile_order[unit / upt] : unit / upt;
        const uint64_t t_unit = P.tile_cost ? __builtin_readcyclecounter() : 0;
        const mp_block T = P.tiles[tile_i];
        const uint32_t px = T.min_x + (b % bx) * BW + static_cast<uint32_t>(pix % BW);
        const uint32_t py = T.min_y + (b / bx) * BH + static_cast<uint32_t>(pix / BW);
        const bool inpix = px < T.max_x && py < T.max_y;
        if (__ballot(inpix) == 0) continue;
        const size_t off = (static_cast<size_t>(tile_i) * ts * ts + static_cast<size_t>(py - T.min_y) * ts + (px - T.min_x)) * 4;
        float acc, cnt;  // pixel_sum (grey table: one channel) and alpha (worker.rs:40)
        pixel_state_load(P, off, inpix, sub == 0, acc, cnt);
        const uint32_t s_begin = P.s_begin, s_end = P.s_end, max_depth = P.max_depth;
        // a batch = NSUB consecutive passes of S samples, aligned in the absolute sample index (so that a 256-sample chunk boundary
        // of MP_FLAG_CHUNKED_SUM never falls inside a batch); samples outside [s_begin, s_end) add +0.0, which is exact
        for (uint32_t s0 = s_begin & ~static_cast<uint32_t>(S * NSUB - 1); s0 < s_end; s0 += S * NSUB) {
            uint32_t nq = 0, na = 0;  // rays in the queue / paths alive, over all passes of the batch (wave-uniform)
            // ---- segment 1: camera rays of every pass as one packet each, then the first vertex
#pragma nounroll
            for (int j = 0; j < NSUB; j++) {
                const uint32_t s = s0 + static_cast<uint32_t>(j * S + sub);
                const bool act = inpix && s >= s_begin && s < s_end;
                Rng rng;
                rng.s0 = rng.s1 = rng.s2 = rng.s3 = 0;
                Ray r;
                r.ox = r.oy = r.oz = r.dx = r.dy = r.dz = r.ix = r.iy = r.iz = 0.0f;
                {
                    const RenderParams& G = params_view(KP);  // ray generation
                    if (act) {
                        rng_seed(rng, sample_key(G.gen, px, py, s));
                        sample_ray_rng(G.gen, px, py, rng, r);
                    }
                }
                segs += static_cast<unsigned long long>(__popcll(__ballot(act)));
                PacketHit h;
                h.t = FLT_MAX; h.u = h.v = 0.0f; h.prim = kNoPrim;
                {
                    const RenderParams& W = params_view(KP);  // walk
                    const bool go = act && may_hit_scene(W.scene, r);
                    if (__ballot(go) != 0) {
                        RegStack rst(nullptr, lane);
                        HybridStack hst(reinterpret_cast<float*>(stack), lane, W.scene.stack_cap, W.scene.packet_stack_regs);
                        if (W.scene.stack_cap > W.scene.packet_stack_regs) trace_packet<false>(W.scene, r, go, hst, h);
                        else trace_packet<true>(W.scene, r, go, rst, h);
                    }
                }
                const RenderParams& V = params_view(KP);  // shade + bounce
                float L[1] = {0.0f}, thr[1] = {1.0f};
                bool alive = act, primary_hit = false;
                if (alive) alive = path_vertex<false, 1>(V.scene, h, 1u, max_depth, rng, r, L, thr, primary_hit);
                uint32_t flags = (alive ? kPoolAlive : 0u) | (primary_hit ? kPoolPrimary : 0u);
                push_ray(V.scene, r, alive, nq, flags);
                na += static_cast<uint32_t>(__popcll(__ballot(alive)));
                park_store(j, rng, L[0], thr[0], flags);
            }
            // ---- segments 2 .. max_depth: one walk over the rays of all passes, then the passes' vertices one after the other
            for (uint32_t depth = 2; depth <= max_depth && na != 0u; depth++) {
                segs += na;
                wave_mem_sync();  // the queue written above is read by other lanes of this wave
                if (nq != 0u) {
                    const RenderParams& W = params_view(KP);
                    trace_wave_pool<QN>(W.scene, q, stack, static_cast<int>(nq));
                }
                wave_mem_sync();
                nq = 0;
                na = 0;
    #pragma nounroll
            for (int j = 0; j < NSUB; j++) {
                    const uint32_t* p = park + j * 64 + lane;
                    uint32_t flags = __builtin_nontemporal_load(p + 10 * QN);
                    bool alive = (flags & kPoolAlive) != 0u;
                    if (__ballot(alive) == 0) continue;  // the whole pass is finished: its parked L / flags stay as they are
                    Rng rng;
                    auto ld = [&](int row) { return static_cast<uint64_t>(__builtin_nontemporal_load(p + row * QN)); };
                    rng.s0 = ld(0) | (ld(1) << 32); rng.s1 = ld(2) | (ld(3) << 32);
                    rng.s2 = ld(4) | (ld(5) << 32); rng.s3 = ld(6) | (ld(7) << 32);
                    float L[1] = {as_f(static_cast<uint32_t>(ld(8)))}, thr[1] = {as_f(static_cast<uint32_t>(ld(9)))};
                    bool primary_hit = (flags & kPoolPrimary) != 0u;
                    Ray r;
                    r.ox = r.oy = r.oz = r.dx = r.dy = r.dz = r.ix = r.iy = r.iz = 0.0f;
                    PacketHit h;
                    h.t = FLT_MAX; h.u = h.v = 0.0f; h.prim = kNoPrim;
                    if (alive && (flags & kPoolQueued)) {  // the ray this path shot and what it hit (a ray that was not queued missed)
                        const uint32_t slot = flags >> 8;
                        r.ox = q[0 * QN + slot]; r.oy = q[1 * QN + slot]; r.oz = q[2 * QN + slot];
                        r.dx = q[3 * QN + slot]; r.dy = q[4 * QN + slot]; r.dz = q[5 * QN + slot];
                        h.t = q[9 * QN + slot]; h.prim = as_u(q[10 * QN + slot]); h.u = q[11 * QN + slot]; h.v = q[12 * QN + slot];
                    }
                    const RenderParams& V = params_view(KP);
                    if (alive) alive = path_vertex<false, 1>(V.scene, h, depth, max_depth, rng, r, L, thr, primary_hit);
                    flags = (alive ? kPoolAlive : 0u) | (primary_hit ? kPoolPrimary : 0u);
                    // every lane of this pass has read its old slot (the loads above feed the stores below); the new slots are no
                    // higher than the old ones of this pass, so later passes' rays and hits are untouched
                    push_ray(V.scene, r, alive, nq, flags);
                    na += static_cast<uint32_t>(__popcll(__ballot(alive)));
                    park_store(j, rng, L[0], thr[0], flags);
                }
            }
            // ---- pixel_sum += sample, strictly in sample order (worker.rs:41-43): the passes in index order
#pragma nounroll
            for (int j = 0; j < NSUB; j++) {
                const uint32_t* p = park + j * 64 + lane;
                const float L = as_f(p[8 * QN]);
                const bool primary_hit = (p[10 * QN] & kPoolPrimary) != 0u;
                {
                    int l_ = lane;
                    asm volatile("" : "+v"(l_));
                    const uint64_t pixel_lanes = ((1ull << S) - 1ull) << (l_ & ~(S - 1));
                    cnt += static_cast<float>(__popcll(__ballot(primary_hit) & pixel_lanes));
                }
                add_samples_in_order<S>(acc, L, lane);
            }
            const RenderParams& A = params_view(KP);  // accumulation
            if (A.chunked && ((s0 + S * NSUB) & (kSumChunk - 1u)) == 0u && s0 + S * NSUB <= s_end) chunk_flush(A, off, inpix && sub == 0, acc);
        }
        const RenderParams& E = params_view(KP);  // unit end
        if (inpix && sub == 0) pixel_state_store(E, off, acc, cnt);
        if (E.tile_cost && lane == 0) atomicAdd(E.tile_cost + tile_i, static_cast<unsigned long long>(__builtin_readcyclecounter() - t_unit));
    }
    const RenderParams& Z = params_view(KP);
    if (lane == 0 && Z.segments && segs) atomicAdd(Z.segments, segs);
}

// ---- staged ("wavefront") evaluation of the path extension: MP_FLAG_WAVEFRONT --------------------------------------
// The north star's formulation, kept beside the fused render_paths_kernel: the paths of a batch (a few tiles x a chunk of
// samples, ~2 M paths) live in HBM as SoA streams (RNG state, ray, throughput, radiance, hit); per segment: a vertex kernel
// (shade + bounce, one thread per path), a counting sort of the live paths by (tile, direction bin) = stream compaction, and a
// trace kernel over the sorted stream.  Per-path RNG stream and operations are those of the fused kernel (path_vertex), every
// ray's result is independent of its neighbours, and the per-pixel sum is taken in sample order at the end: the frame is
// bit-identical to the fused kernel's and the oracle's (the two pipelines cross-check each other in the tests).  Measured slower
// than the fused kernel, which therefore stays the default.
constexpr uint32_t kDirBins = 512;  // 8 octants x 8 x 8 cells of the octahedral map of |d|

struct WfState {
    uint64_t* rng;       // 4 rows x n
    float* ray;          // 6 rows x n : origin, unit direction
    float* thr;          // nchan rows x n (nchan = 3 for a coloured / textured material table, else 1)
    float* L;            // nchan rows x n
    float* hit_t;        // n
    uint32_t* hit_prim;  // n
    float* hit_u;        // n
    float* hit_v;        // n
    uint32_t* hit_inst;  // n : member of the object group that was hit (groups only; otherwise unused)
    uint32_t* flags;     // n : kWfAlive | kWfPrimaryHit | kWfValid
    uint32_t* key;       // n : sort key of a live path
    uint32_t* idx;       // n : live paths in key order
    uint32_t* hist;      // nbins + 1 : per-key counts of the segment being generated ; [nbins] unused
    uint32_t* offs;      // nbins + 1 : exclusive scan of hist ; [nbins] = number of live paths
    uint32_t* cursor;    // nbins
    uint32_t n, nbins, nchan;
};
constexpr uint32_t kWfAlive = 1u, kWfPrimaryHit = 2u, kWfValid = 4u;

struct WfParams {
    DevScene scene;
    RayGen gen;
    const mp_block* tiles;  // this batch's tiles (device)
    uint32_t n_tiles, tile_size;
    uint32_t tile_base;     // index of the batch's first tile in the caller's tile list (output slot)
    uint32_t s0, sc, s_end; // the chunk covers samples [s0, min(s0 + sc, s_end))
    uint32_t depth, max_depth;
    float* out;
    float inv_spp;
    uint32_t carry_in, finalize, chunked;
    uint32_t lds_per_wave;
    unsigned long long* segments;
    WfState st;
};

__device__ __forceinline__ uint32_t direction_bin(float dx, float dy, float dz) {
    const float ax = fabsf(dx), ay = fabsf(dy), az = fabsf(dz);
    const float inv = __builtin_amdgcn_rcpf(ax + ay + az);  // approximate: the bin only steers the sort
    const uint32_t qx = min(7u, static_cast<uint32_t>(ax * inv * 8.0f)), qy = min(7u, static_cast<uint32_t>(ay * inv * 8.0f));
    const uint32_t oct = (dx < 0.0f ? 1u : 0u) | (dy < 0.0f ? 2u : 0u) | (dz < 0.0f ? 4u : 0u);
    return oct * 64u + qx * 8u + qy;
}

// path p = (tile_local * ts*ts + (y - min_y) * ts + (x - min_x)) * sc + (s - s0)
// Stage 1: camera rays, generated and walked as packets of 2x2 pixels x 16 samples like render_tiles_packet_kernel.
template <bool LDS_STACK, bool OBJ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 8))) void wf_camera_kernel(WfParams) {
    extern __shared__ __align__(16) unsigned char smem[];
    typedef const __attribute__((address_space(4))) WfParams* kwf_t;
    kwf_t KP = (kwf_t)__builtin_amdgcn_kernarg_segment_ptr();  // WfParams through short-lived views (see params_view)
    const WfParams& P0 = kernarg_view<WfParams>(KP);
    constexpr int S = 16, BW = 2, BH = 2;
    const int lane = static_cast<int>(threadIdx.x) & 63;
    const int pix = lane / S, sub = lane % S;
    const uint32_t ts = P0.tile_size, bx = (ts + BW - 1) / BW, by = (ts + BH - 1) / BH, upt = bx * by, total = P0.n_tiles * upt;
    const uint32_t n = P0.st.n;
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    unsigned long long segs = 0;  // camera segments of this wave (wave-uniform)
    for (uint32_t unit = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); unit < total; unit += n_waves) {
        const WfParams& Pu = kernarg_view<WfParams>(KP);
        const uint32_t tile_l = unit / upt, b = unit % upt;
        const mp_block T = Pu.tiles[tile_l];
        const uint32_t px = T.min_x + (b % bx) * BW + static_cast<uint32_t>(pix % BW);
        const uint32_t py = T.min_y + (b / bx) * BH + static_cast<uint32_t>(pix / BW);
        const bool inpix = px < T.max_x && py < T.max_y;
        const uint32_t pbase = ((tile_l * ts + (py - T.min_y)) * ts + (px - T.min_x)) * Pu.sc;
        const uint32_t sc_ = Pu.sc;
        for (uint32_t sl0 = 0; sl0 < sc_; sl0 += S) {
            const WfParams& P = kernarg_view<WfParams>(KP);  // one view per pass
            const uint32_t sl = sl0 + static_cast<uint32_t>(sub), s = P.s0 + sl;
            const bool slot = inpix && sl < P.sc;       // a path slot of the batch exists for this lane
            const bool act = slot && s < P.s_end;       // ... and holds a sample of this launch
            Rng rng;
            rng.s0 = rng.s1 = rng.s2 = rng.s3 = 0;
            Ray r;
            r.ox = r.oy = r.oz = r.dx = r.dy = r.dz = r.ix = r.iy = r.iz = 0.0f;
            if (act) {
                rng_seed(rng, sample_key(P.gen, px, py, s));
                sample_ray_rng(P.gen, px, py, rng, r);
            }
            segs += static_cast<unsigned long long>(__popcll(__ballot(act)));
            PacketHit h;
            h.t = FLT_MAX; h.u = h.v = 0.0f; h.prim = kNoPrim;
            uint32_t hinst = 0u;
            float* lds = reinterpret_cast<float*>(smem + static_cast<size_t>(static_cast<int>(threadIdx.x) >> 6) * P.lds_per_wave);
            if (OBJ) {  // object group: one packet walk per member
                if (LDS_STACK) {
                    HybridStack st(lds, lane, P.scene.stack_cap, P.scene.packet_stack_regs);
                    trace_packet_objects<false>(P.scene, r, act, st, h, hinst);
                } else {
                    RegStack st(lds, lane);
                    trace_packet_objects<true>(P.scene, r, act, st, h, hinst);
                }
            } else {
                const bool go = act && may_hit_scene(P.scene, r);
                if (__ballot(go) != 0) {
                    if (LDS_STACK) {
                        HybridStack st(lds, lane, P.scene.stack_cap, P.scene.packet_stack_regs);
                        trace_packet<false>(P.scene, r, go, st, h);
                    } else {
                        RegStack st(lds, lane);
                        trace_packet<true>(P.scene, r, go, st, h);
                    }
                }
            }
            if (slot) {
                const uint32_t p = pbase + sl;
                P.st.flags[p] = act ? (kWfAlive | kWfValid) : 0u;
                if (act) {
                    P.st.rng[0 * static_cast<size_t>(n) + p] = rng.s0; P.st.rng[1 * static_cast<size_t>(n) + p] = rng.s1;
                    P.st.rng[2 * static_cast<size_t>(n) + p] = rng.s2; P.st.rng[3 * static_cast<size_t>(n) + p] = rng.s3;
                    P.st.ray[0 * static_cast<size_t>(n) + p] = r.ox; P.st.ray[1 * static_cast<size_t>(n) + p] = r.oy;
                    P.st.ray[2 * static_cast<size_t>(n) + p] = r.oz; P.st.ray[3 * static_cast<size_t>(n) + p] = r.dx;
                    P.st.ray[4 * static_cast<size_t>(n) + p] = r.dy; P.st.ray[5 * static_cast<size_t>(n) + p] = r.dz;
                    for (uint32_t c = 0; c < P.st.nchan; c++) {
                        P.st.thr[c * static_cast<size_t>(n) + p] = 1.0f;
                        P.st.L[c * static_cast<size_t>(n) + p] = 0.0f;
                    }
                    P.st.hit_t[p] = h.t; P.st.hit_prim[p] = h.prim; P.st.hit_u[p] = h.u; P.st.hit_v[p] = h.v;
                    if (OBJ) P.st.hit_inst[p] = hinst;
                }
            }
        }
    }
    if (lane == 0 && P0.segments && segs) atomicAdd(P0.segments, segs);
}

// Stage 2: one thread per path: shade the hit of segment P.depth, draw the bounce ray, count it under its sort key.
template <int N, bool OBJ>
__global__ __launch_bounds__(256) void wf_vertex_kernel(WfParams P) {
    const uint32_t n = P.st.n;
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
        uint32_t fl = P.st.flags[p];
        if (!(fl & kWfAlive)) continue;
        Rng rng;
        rng.s0 = P.st.rng[0 * static_cast<size_t>(n) + p]; rng.s1 = P.st.rng[1 * static_cast<size_t>(n) + p];
        rng.s2 = P.st.rng[2 * static_cast<size_t>(n) + p]; rng.s3 = P.st.rng[3 * static_cast<size_t>(n) + p];
        Ray r;
        r.ox = P.st.ray[0 * static_cast<size_t>(n) + p]; r.oy = P.st.ray[1 * static_cast<size_t>(n) + p];
        r.oz = P.st.ray[2 * static_cast<size_t>(n) + p]; r.dx = P.st.ray[3 * static_cast<size_t>(n) + p];
        r.dy = P.st.ray[4 * static_cast<size_t>(n) + p]; r.dz = P.st.ray[5 * static_cast<size_t>(n) + p];
        r.ix = r.iy = r.iz = 0.0f;  // not used by path_vertex
        PacketHit h;
        h.t = P.st.hit_t[p]; h.prim = P.st.hit_prim[p]; h.u = P.st.hit_u[p]; h.v = P.st.hit_v[p];
        float L[N], thr[N];
#pragma unroll
        for (int c = 0; c < N; c++) { L[c] = P.st.L[c * static_cast<size_t>(n) + p]; thr[c] = P.st.thr[c * static_cast<size_t>(n) + p]; }
        bool primary = (fl & kWfPrimaryHit) != 0u;
        const bool alive = path_vertex<OBJ, N>(P.scene, h, P.depth, P.max_depth, rng, r, L, thr, primary, OBJ ? P.st.hit_inst[p] : 0u);
        fl = (fl & ~(kWfAlive | kWfPrimaryHit)) | (alive ? kWfAlive : 0u) | (primary ? kWfPrimaryHit : 0u);
        P.st.flags[p] = fl;
#pragma unroll
        for (int c = 0; c < N; c++) { P.st.L[c * static_cast<size_t>(n) + p] = L[c]; P.st.thr[c * static_cast<size_t>(n) + p] = thr[c]; }
        if (alive) {
            P.st.rng[0 * static_cast<size_t>(n) + p] = rng.s0; P.st.rng[1 * static_cast<size_t>(n) + p] = rng.s1;
            P.st.rng[2 * static_cast<size_t>(n) + p] = rng.s2; P.st.rng[3 * static_cast<size_t>(n) + p] = rng.s3;
            P.st.ray[0 * static_cast<size_t>(n) + p] = r.ox; P.st.ray[1 * static_cast<size_t>(n) + p] = r.oy;
            P.st.ray[2 * static_cast<size_t>(n) + p] = r.oz; P.st.ray[3 * static_cast<size_t>(n) + p] = r.dx;
            P.st.ray[4 * static_cast<size_t>(n) + p] = r.dy; P.st.ray[5 * static_cast<size_t>(n) + p] = r.dz;
            const uint32_t tile_l = p / (P.tile_size * P.tile_size * P.sc);
            const uint32_t k = tile_l * kDirBins + direction_bin(r.dx, r.dy, r.dz);
            P.st.key[p] = k;
            atomicAdd(P.st.hist + k, 1u);
        }
    }
}

// Stage 3: exclusive scan of the key histogram (one workgroup), cursors and histogram reset for the next segment, and the
// number of live paths = ray segments of the next stage.
__global__ __launch_bounds__(1024) void wf_scan_kernel(WfParams P) {
    __shared__ uint32_t part[1024];
    const uint32_t nb = P.st.nbins, per = (nb + 1023u) / 1024u, t = threadIdx.x;
    uint32_t sum = 0;
    for (uint32_t i = t * per; i < min(nb, (t + 1u) * per); i++) sum += P.st.hist[i];
    part[t] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {  // Hillis-Steele inclusive scan of the per-thread sums
        const uint32_t v = t >= d ? part[t - d] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;
    for (uint32_t i = t * per; i < min(nb, (t + 1u) * per); i++) {
        const uint32_t c = P.st.hist[i];
        P.st.offs[i] = run;
        P.st.cursor[i] = 0u;
        P.st.hist[i] = 0u;
        run += c;
    }
    if (t == 1023u) {
        P.st.offs[nb] = part[1023];
        if (P.segments) atomicAdd(P.segments, static_cast<unsigned long long>(part[1023]));
    }
}

// Stage 4: scatter the live paths to their key's range (order inside a key is arbitrary: results do not depend on it).
__global__ __launch_bounds__(256) void wf_scatter_kernel(WfParams P) {
    const uint32_t n = P.st.n;
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
        if (!(P.st.flags[p] & kWfAlive)) continue;
        const uint32_t k = P.st.key[p];
        P.st.idx[P.st.offs[k] + atomicAdd(P.st.cursor + k, 1u)] = p;
    }
}

// Stage 5: the sorted rays through the 8-lane-group traversal, 64 per wave.  (Walking 64 sorted rays as ONE packet was measured
// too: diffuse bounce rays do not share enough of the tree even when sorted -- 2 586 VALU per ray against ~500 here; see
// profiles/r01_notes.md.)  The sort serves cache locality.
template <bool OBJ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void wf_trace_groups_kernel(WfParams P) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = static_cast<int>(threadIdx.x) & 63, wave = static_cast<int>(threadIdx.x) >> 6;
    float* q = reinterpret_cast<float*>(smem + static_cast<size_t>(wave) * P.lds_per_wave);
    uint2* stack = reinterpret_cast<uint2*>(q + kQueueFloats);
    const uint32_t n = P.st.n, live = P.st.offs[P.st.nbins];
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6), chunks = (live + 63u) / 64u;
    for (uint32_t c = blockIdx.x * (blockDim.x >> 6) + static_cast<uint32_t>(wave); c < chunks; c += n_waves) {
        const uint32_t slot = c * 64u + static_cast<uint32_t>(lane);
        const bool act = slot < live;
        const uint32_t p = act ? P.st.idx[slot] : 0u;
        Ray r;
        r.ox = r.oy = r.oz = r.dx = r.dy = r.dz = r.ix = r.iy = r.iz = 0.0f;
        if (act) {
            r.ox = P.st.ray[0 * static_cast<size_t>(n) + p]; r.oy = P.st.ray[1 * static_cast<size_t>(n) + p];
            r.oz = P.st.ray[2 * static_cast<size_t>(n) + p]; r.dx = P.st.ray[3 * static_cast<size_t>(n) + p];
            r.dy = P.st.ray[4 * static_cast<size_t>(n) + p]; r.dz = P.st.ray[5 * static_cast<size_t>(n) + p];
            r.ix = (r.dx == 0.0f) ? INFINITY : 1.0f / r.dx;
            r.iy = (r.dy == 0.0f) ? INFINITY : 1.0f / r.dy;
            r.iz = (r.dz == 0.0f) ? INFINITY : 1.0f / r.dz;
        }
        // compaction of the rays that can reach the object into the wave's queue + the 8-lane-group walk (once per member of an
        // object group, closest wins)
        GroupHit gh;
        trace_objects<OBJ>(P.scene, r, act, q, stack, gh);
        if (act) {
            P.st.hit_t[p] = gh.t;
            P.st.hit_prim[p] = gh.prim;
            P.st.hit_u[p] = gh.u;
            P.st.hit_v[p] = gh.v;
            if (OBJ) P.st.hit_inst[p] = gh.inst;
        }
    }
}

// Stage 6: pixel_sum += sample in sample order (worker.rs:41-43), one thread per pixel of the batch.
__global__ __launch_bounds__(256) void wf_accumulate_kernel(WfParams P) {
    const uint32_t ts = P.tile_size, npx = P.n_tiles * ts * ts;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += gridDim.x * blockDim.x) {
        const uint32_t tile_l = i / (ts * ts), q = i % (ts * ts), x = q % ts, y = q / ts;
        const mp_block T = P.tiles[tile_l];
        if (!(T.min_x + x < T.max_x && T.min_y + y < T.max_y)) continue;
        float* o = P.out + (static_cast<size_t>(P.tile_base + tile_l) * ts * ts + q) * 4;
        if (P.st.nchan == 3u) {  // coloured / textured material table: slot = {sum r, sum g, sum b, hit count}; never chunked
            float a3[3] = {0.0f, 0.0f, 0.0f}, c3 = 0.0f;
            if (P.carry_in) {
                const float4 prev = *reinterpret_cast<const float4*>(o);
                a3[0] = prev.x; a3[1] = prev.y; a3[2] = prev.z; c3 = prev.w;
            }
            const uint32_t ns3 = min(P.sc, P.s_end - P.s0);
            const uint32_t pb3 = i * P.sc;
            for (uint32_t sl = 0; sl < ns3; sl++) {
                for (uint32_t c = 0; c < 3u; c++) a3[c] += P.st.L[c * static_cast<size_t>(P.st.n) + pb3 + sl];
                c3 += (P.st.flags[pb3 + sl] & kWfPrimaryHit) ? 1.0f : 0.0f;
            }
            const float sc3 = P.finalize ? P.inv_spp : 1.0f;
            *reinterpret_cast<float4*>(o) = make_float4(a3[0] * sc3, a3[1] * sc3, a3[2] * sc3, c3 * sc3);
            continue;
        }
        float acc = 0.0f, cnt = 0.0f;
        double tot = 0.0;  // MP_FLAG_CHUNKED_SUM: slot = {chunk sum, hit count, f64 total} (pixel_state_load)
        if (P.carry_in) {
            const float4 prev = *reinterpret_cast<const float4*>(o);
            acc = prev.x;
            cnt = P.chunked ? prev.y : prev.w;
            if (P.chunked) tot = *reinterpret_cast<const double*>(o + 2);
        }
        const uint32_t ns = min(P.sc, P.s_end - P.s0);
        const uint32_t pbase = i * P.sc;
        for (uint32_t sl = 0; sl < ns; sl++) {
            acc += P.st.L[pbase + sl];
            cnt += (P.st.flags[pbase + sl] & kWfPrimaryHit) ? 1.0f : 0.0f;
            if (P.chunked && ((P.s0 + sl + 1u) & (kSumChunk - 1u)) == 0u) { tot = tot + static_cast<double>(acc); acc = 0.0f; }
        }
        if (P.chunked) {
            if (!P.finalize) {
                *reinterpret_cast<float2*>(o) = make_float2(acc, cnt);
                *reinterpret_cast<double*>(o + 2) = tot;
                continue;
            }
            if (((P.s0 + ns) & (kSumChunk - 1u)) != 0u) tot = tot + static_cast<double>(acc);
            const double inv = 1.0 / static_cast<double>(P.gen.spp);
            const float m = static_cast<float>(tot * inv), a = static_cast<float>(static_cast<double>(cnt) * inv);
            *reinterpret_cast<float4*>(o) = make_float4(m, m, m, a);
            continue;
        }
        const float m = P.finalize ? acc * P.inv_spp : acc, a = P.finalize ? cnt * P.inv_spp : cnt;
        *reinterpret_cast<float4*>(o) = make_float4(m, m, m, a);
    }
}

// ---- batched Object::intersect over SoA ray streams (ray_bvh_intersection.rs:26-96) -------------------------
struct TraceParams {
    DevScene scene;
    const float *ox, *oy, *oz, *dx, *dy, *dz;
    uint64_t n;
    mp_hits_soa hits;
    uint32_t lds_per_wave;
};

template <bool OBJ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void trace_rays_kernel(TraceParams) {
    extern __shared__ __align__(16) unsigned char smem[];
    typedef const __attribute__((address_space(4))) TraceParams* ktrace_t;
    ktrace_t KP = (ktrace_t)__builtin_amdgcn_kernarg_segment_ptr();  // TraceParams through short-lived views (see params_view)
    const TraceParams& P0 = kernarg_view<TraceParams>(KP);
    const int lane = static_cast<int>(threadIdx.x) & 63, wave = static_cast<int>(threadIdx.x) >> 6;
    float* q = reinterpret_cast<float*>(smem + static_cast<size_t>(wave) * P0.lds_per_wave);
    uint2* stack = reinterpret_cast<uint2*>(q + kQueueFloats);
    const uint64_t chunks = (P0.n + 63) / 64;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * (blockDim.x >> 6);
    for (uint64_t chunk = static_cast<uint64_t>(blockIdx.x) * (blockDim.x >> 6) + wave; chunk < chunks; chunk += stride) {
        const TraceParams& P = kernarg_view<TraceParams>(KP);  // one view per chunk of 64 rays
        const uint64_t i = chunk * 64 + lane;
        const bool act = i < P.n;
        Ray r0;
        r0.ox = r0.oy = r0.oz = r0.dx = r0.dy = r0.dz = r0.ix = r0.iy = r0.iz = 0.0f;
        if (act) ray_new(P.ox[i], P.oy[i], P.oz[i], P.dx[i], P.dy[i], P.dz[i], r0);
        if (P.scene.kind == 1u) {
            const Ray& r = r0;  // Sphere: HitRecord{t, point, normal, material 0, texture_coords origin} (primitives.rs:40-46)
            if (act) {
                float t = FLT_MAX, nn[3] = {0, 0, 0};
                const bool hit = sphere_intersect(P.scene, r, t, nn);
                if (!hit) t = FLT_MAX;
                if (P.hits.d_t) P.hits.d_t[i] = t;
                if (P.hits.d_prim) P.hits.d_prim[i] = hit ? 0u : kNoPrim;
                if (P.hits.d_u) P.hits.d_u[i] = 0.0f;
                if (P.hits.d_v) P.hits.d_v[i] = 0.0f;
                for (int k = 0; k < 3; k++) {
                    const float o = k == 0 ? r.ox : k == 1 ? r.oy : r.oz, d = k == 0 ? r.dx : k == 1 ? r.dy : r.dz;
                    if (P.hits.d_point) P.hits.d_point[i * 3 + k] = hit ? o + d * t : 0.0f;
                    if (P.hits.d_normal) P.hits.d_normal[i * 3 + k] = hit ? nn[k] : 0.0f;
                    if (P.hits.d_tex) P.hits.d_tex[i * 3 + k] = 0.0f;
                }
                if (P.hits.d_material) P.hits.d_material[i] = 0u;
                if (P.hits.d_instance) P.hits.d_instance[i] = 0u;
            }
            continue;
        }
        GroupHit gh;
        trace_objects<OBJ, true>(P.scene, r0, act, q, stack, gh);
        const float t = gh.t, u = gh.u, v = gh.v;
        const uint32_t prim = gh.prim, inst = gh.inst;
        if (act) {
            if (P.hits.d_t) P.hits.d_t[i] = t;
            if (P.hits.d_prim) P.hits.d_prim[i] = prim;
            if (P.hits.d_u) P.hits.d_u[i] = u;
            if (P.hits.d_v) P.hits.d_v[i] = v;
            if (P.hits.d_point || P.hits.d_normal || P.hits.d_tex || P.hits.d_material) {
                float pt[3] = {0, 0, 0}, nn[3] = {0, 0, 0}, tx[3] = {0, 0, 0};
                uint32_t mat = 0;  // HitRecord.material (geometry/mod.rs:78)
                if (prim != kNoPrim) {
                    Ray r;  // rebuilt here so that no ray registers stay live across the walk
                    ray_new(P.ox[i], P.oy[i], P.oz[i], P.dx[i], P.dy[i], P.dz[i], r);
                    DevScene so = P.scene;  // the member the ray hit (its normals, texture coordinates, material ids)
                    if (OBJ) object_scene(P.scene, inst, so);
                    mat = OBJ ? object_normal(P.scene, inst, r, prim, u, v, nn) : resolve_normal(so, prim, u, v, nn);
                    pt[0] = r.ox + r.dx * t; pt[1] = r.oy + r.dy * t; pt[2] = r.oz + r.dz * t;  // geometry/mod.rs:56-58
                    if (!OBJ || so.kind == 0u) {  // a Sphere member's texture_coords are the origin (primitives.rs:45)
                        const uint32_t* vi = so.vidx + static_cast<size_t>(prim) * 3;
                        const float *t0 = so.vtex + 3 * static_cast<size_t>(vi[0]), *t1 = so.vtex + 3 * static_cast<size_t>(vi[1]),
                                    *t2 = so.vtex + 3 * static_cast<size_t>(vi[2]);
                        float w = 1.0f - u - v;
                        for (int k = 0; k < 3; k++) tx[k] = t0[k] * w + t1[k] * u + t2[k] * v;
                    }
                }
                for (int k = 0; k < 3; k++) {
                    if (P.hits.d_point) P.hits.d_point[i * 3 + k] = pt[k];
                    if (P.hits.d_normal) P.hits.d_normal[i * 3 + k] = nn[k];
                    if (P.hits.d_tex) P.hits.d_tex[i * 3 + k] = tx[k];
                }
                if (P.hits.d_material) P.hits.d_material[i] = mat;
            }
            if (P.hits.d_instance) P.hits.d_instance[i] = inst;
        }
    }
}

// ---- CameraSampler::sample_ray batched (camera.rs:176-191) -----------------------------------------------------
__global__ __launch_bounds__(256) void generate_rays_kernel(RayGen G, mp_block blk, uint32_t sample, float* ox, float* oy,
                                                            float* oz, float* dx, float* dy, float* dz) {
    const uint32_t w = blk.max_x - blk.min_x, h = blk.max_y - blk.min_y;
    const uint64_t n = static_cast<uint64_t>(w) * h;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
         i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        Ray r;
        sample_ray(G, blk.min_x + static_cast<uint32_t>(i % w), blk.min_y + static_cast<uint32_t>(i / w), sample, r);
        ox[i] = r.ox; oy[i] = r.oy; oz[i] = r.oz;
        dx[i] = r.dx; dy[i] = r.dy; dz[i] = r.dz;
    }
}

// ---- tile buffer -> image (machinery.rs:78-89) + color_to_image (worker.rs:69-76) ----------------------------
__device__ __forceinline__ uint8_t to_u8(float c) {
    float x = roundf(c * 255.0f);  // half away from zero
    if (x != x) return 0;          // `as u8` on NaN
    x = fminf(fmaxf(x, 0.0f), 255.0f);
    return static_cast<uint8_t>(x);
}

// mode 0: the buffer holds finished pixels (means).  Preview of an unfinished MP_FLAG_ACCUMULATE buffer after k samples (the buffer
// is only read): mode 1: slot = {sum, sum, sum, hits} -> {sum * inv_k, .., hits * inv_k} with inv_k = 1.0f / (float)k (worker.rs:44
// for the samples drawn so far); mode 2 (MP_FLAG_CHUNKED_SUM): slot = {chunk sum, hits, f64 total} ->
// (f32)((total + (f64)chunk sum) * (1.0 / (f64)k)), (f32)((f64)hits * (1.0 / (f64)k)) -- pixel_state_store's rule for k samples.
__global__ __launch_bounds__(256) void untile_kernel(uint32_t width, uint32_t height, uint32_t ts, const mp_block* tiles,
                                                     uint32_t n_tiles, const float* src, float* img_f32, uint8_t* img_u8,
                                                     uint32_t mode, float inv_k, double inv_k64) {
    const uint64_t per_tile = static_cast<uint64_t>(ts) * ts;
    const uint64_t n = per_tile * n_tiles;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
         i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint32_t tile_i = static_cast<uint32_t>(i / per_tile), p = static_cast<uint32_t>(i % per_tile);
        const mp_block T = tiles[tile_i];
        const uint32_t x = T.min_x + p % ts, y = T.min_y + p / ts;
        if (x >= T.max_x || y >= T.max_y || x >= width || y >= height) continue;
        float4 v = *reinterpret_cast<const float4*>(src + i * 4);
        if (mode == 1u) {
            const float m = v.x * inv_k;
            v = make_float4(m, m, m, v.w * inv_k);
        } else if (mode == 2u) {
            const double tot = *reinterpret_cast<const double*>(src + i * 4 + 2) + static_cast<double>(v.x);
            const float m = static_cast<float>(tot * inv_k64);
            v = make_float4(m, m, m, static_cast<float>(static_cast<double>(v.y) * inv_k64));
        }
        const size_t o = (static_cast<size_t>(y) * width + x) * 4;
        if (img_f32) *reinterpret_cast<float4*>(img_f32 + o) = v;
        if (img_u8) {
            uchar4 c = make_uchar4(to_u8(v.x), to_u8(v.y), to_u8(v.z), to_u8(v.w));
            *reinterpret_cast<uchar4*>(img_u8 + o) = c;
        }
    }
}

// color_to_image (worker.rs:69-76) over a tile-major pixel buffer: RGBA f32 -> RGBA u8, same layout (the render() worker reads
// both back and only copies rows on the host).
__global__ __launch_bounds__(256) void quantise_kernel(const float* src, uint8_t* dst, uint64_t n_pixels) {
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n_pixels;
         i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const float4 v = *reinterpret_cast<const float4*>(src + i * 4);
        *reinterpret_cast<uchar4*>(dst + i * 4) = make_uchar4(to_u8(v.x), to_u8(v.y), to_u8(v.z), to_u8(v.w));
    }
}

__global__ void set_u64_kernel(unsigned long long* p, unsigned long long v) { *p = v; }

// UniformFloat::new_inclusive(low, high).scale (rand 0.9): (high-low)/max_rand, reduced by ulps until
// scale*max_rand + low <= high
float uniform_inclusive_scale(float low, float high) {
    const float max_rand = 1.0f - FLT_EPSILON;
    float scale = (high - low) / max_rand;
    for (;;) {
        volatile float top = scale * max_rand;
        top = top + low;
        if (!(top > high)) break;
        uint32_t b;
        __builtin_memcpy(&b, &scale, 4);
        b -= 1;
        __builtin_memcpy(&scale, &b, 4);
    }
    return scale;
}

// Seeded mode (include/minipath_hip.h): the first SplitMix64 output for state `seed`, so that consecutive seeds give unrelated
// sample streams; the per-sample index is added on the device (sample_key).
uint64_t mixed_seed(uint64_t seed) {
    uint64_t z = seed + 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

uint32_t lds_bytes_per_wave(uint32_t stack_cap) { return static_cast<uint32_t>(kQueueFloats * 4 + 8u * stack_cap * 8u); }

int check(hipError_t e, const char* what, std::string& err) {
    if (e == hipSuccess) return MP_OK;
    err = std::string(what) + ": " + hipGetErrorString(e);
    return MP_ERR_HIP;
}

}  // namespace

int launch_render_tiles(const RenderLaunch& L, void* stream, std::string& err) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (L.n_tiles == 0) return MP_OK;
    RenderParams P;
    P.scene = L.scene;
    P.gen.s = L.sampler;
    P.gen.jitter_scale = uniform_inclusive_scale(-0.5f, 0.5f);
    P.gen.width = L.width;
    P.gen.spp = L.spp;
    P.gen.seed = mixed_seed(L.seed);
    P.tiles = L.d_tiles;
    P.n_tiles = L.n_tiles;
    P.tile_size = L.tile_size;
    P.out = L.d_out;
    P.counter = L.d_counter;
    P.inv_spp = 1.0f / static_cast<float>(L.spp);
    P.s_begin = L.pass_begin;
    P.s_end = L.pass_end;
    P.carry_in = L.carry_in ? 1u : 0u;
    P.finalize = L.finalize ? 1u : 0u;
    P.chunked = L.chunked ? 1u : 0u;
    P.tile_order = L.d_tile_order;
    P.tile_cost = L.d_tile_cost;
    P.max_depth = L.max_depth;
    P.segments = L.d_segments;
    P.pool = nullptr;
    P.pool_stride = 0;
    P.lds_per_wave = lds_bytes_per_wave(L.scene.stack_cap);
    const uint32_t lds = P.lds_per_wave * 4;
    if (lds > 160 * 1024) { err = "scene too deep for the LDS traversal stacks"; return MP_ERR_UNSUPPORTED; }
    int rc = check(hipMemsetAsync(L.d_counter, 0, kWorkQueues * kWorkQueueStride * sizeof(uint32_t), st), "hipMemsetAsync(counter)", err);
    if (rc) return rc;
    const uint64_t units = static_cast<uint64_t>(L.n_tiles) * ((L.tile_size + 7) / 8) * ((L.tile_size + 7) / 8);
    const uint64_t want = (units + 3) / 4;
    if (L.max_depth > 0 && L.scene.kind != 0u) { err = "the path extension is defined for TriangleBvh scenes only"; return MP_ERR_UNSUPPORTED; }
    if (L.max_depth > 0) {  // build-defined path extension
        const uint32_t per_cu = std::max<uint32_t>(1, std::min<uint32_t>(8, (160u * 1024u) / lds));
        const uint32_t nspp = L.pass_end - L.pass_begin;  // samples per pixel in this launch
        const int S = nspp >= 8 ? 8 : nspp >= 4 ? 4 : nspp >= 2 ? 2 : 1;  // 16 in flight measured slower here (teapot depth 8: 17.1 vs 15.7 ms)
        const uint32_t grid = static_cast<uint32_t>(std::min<uint64_t>(want * S, static_cast<uint64_t>(L.cu_count) * per_cu));
#define MP_LAUNCH_PATHS(SV)                                                                                                  \
    do {                                                                                                                     \
        if (L.scene.inst_count != 0u) {  /* object group: every segment is walked member by member */                        \
            if (rgb) hipLaunchKernelGGL((render_paths_kernel<SV, true, true>), dim3(grid), dim3(256), lds, st, P);           \
            else hipLaunchKernelGGL((render_paths_kernel<SV, true, false>), dim3(grid), dim3(256), lds, st, P);              \
        } else if (rgb) hipLaunchKernelGGL((render_paths_kernel<SV, false, true>), dim3(grid), dim3(256), lds, st, P);       \
        else hipLaunchKernelGGL((render_paths_kernel<SV, false, false>), dim3(grid), dim3(256), lds, st, P);                 \
    } while (0)
        const bool rgb = L.scene.materials_rgb != 0u;  // a coloured / textured material table: three channels
        if (rgb && L.chunked) { err = "coloured / textured materials are not combined with MP_FLAG_CHUNKED_SUM"; return MP_ERR_UNSUPPORTED; }
        // pooled form (render_paths_pooled_kernel): plain TriangleBvh scenes with a grey table, at least two passes of 8 samples.
        // By default for scenes whose traversal arrays exceed 1 MB -- there the 8-lane-group walk dominates and the longer queue
        // pays (stand-in depth 8: 564 against 628 ms); on the teapot, where most paths end after one or two segments and ray
        // generation and shading dominate, the one-pass kernel is faster (50.2 against 54.3 ms).  paths_pooled: 0 never, 1 auto,
        // 2 always with two passes, 3 always with up to four.
        const bool big_scene = (static_cast<uint64_t>(L.scene.inner_count) * 256u + static_cast<uint64_t>(L.scene.packet_count) * 384u) > (1u << 20);
        const bool pooled = L.paths_pooled >= 2u || (L.paths_pooled == 1u && big_scene);
        if (pooled && !rgb && L.scene.inst_count == 0u && L.max_depth >= 2 && nspp >= 16) {
            const int nsub = (nspp >= 32 && L.paths_pooled != 2u) ? 4 : 2;
            const uint32_t plds_wave = 8u * L.scene.stack_cap * 8u;  // the eight traversal stacks; the ray queue lives in the pool
            P.lds_per_wave = plds_wave;
            const uint32_t plds = plds_wave * 4u;
            if (plds > 160 * 1024) { err = "scene too deep for the LDS traversal stacks"; return MP_ERR_UNSUPPORTED; }
            const uint32_t pper_cu = plds ? std::max<uint32_t>(1, std::min<uint32_t>(8, (160u * 1024u) / plds)) : 8u;
            const uint32_t pgrid = static_cast<uint32_t>(std::min<uint64_t>(want * 8, static_cast<uint64_t>(L.cu_count) * pper_cu));
            P.pool_stride = pool_floats_per_wave(nsub);
            const size_t bytes = static_cast<size_t>(pgrid) * 4u * P.pool_stride * sizeof(float);
            rc = check(hipMallocAsync(reinterpret_cast<void**>(&P.pool), bytes, st), "hipMallocAsync(path pool)", err);
            if (rc) return rc;
            if (nsub == 4) hipLaunchKernelGGL(render_paths_pooled_kernel<4>, dim3(pgrid), dim3(256), plds, st, P);
            else hipLaunchKernelGGL(render_paths_pooled_kernel<2>, dim3(pgrid), dim3(256), plds, st, P);
            rc = check(hipGetLastError(), "render_paths_pooled_kernel launch", err);
            (void)hipFreeAsync(P.pool, st);
            return rc;
        }
        // camera pass on the cached packet walk (MaskCache): plain scenes whose stack fits the registers, units of at least four
        // passes, and only while the cache's 3 712 bytes per wave leave the six resident waves per SIMD their LDS
        const uint32_t clds_wave = P.lds_per_wave + static_cast<uint32_t>(kMaskCacheDwords) * 4u;
        if (L.mask_cache != 0u && S == 8 && L.scene.inst_count == 0u && L.scene.stack_cap <= L.scene.packet_stack_regs && nspp >= 32u &&
            L.scene.inner_count < (1u << 24) && L.scene.tris_bounded != 0u && L.scene.boxes_ordered != 0u && clds_wave * 4u * MP_PATHS_WPE <= 160u * 1024u) {
            P.lds_per_wave = clds_wave;
            if (rgb) hipLaunchKernelGGL((render_paths_kernel<8, false, true, true>), dim3(grid), dim3(256), clds_wave * 4u, st, P);
            else hipLaunchKernelGGL((render_paths_kernel<8, false, false, true>), dim3(grid), dim3(256), clds_wave * 4u, st, P);
            return check(hipGetLastError(), "render_paths_kernel launch", err);
        }
        if (S == 8) MP_LAUNCH_PATHS(8);
        else if (S == 4) MP_LAUNCH_PATHS(4);
        else if (S == 2) MP_LAUNCH_PATHS(2);
        else MP_LAUNCH_PATHS(1);
#undef MP_LAUNCH_PATHS
        return check(hipGetLastError(), "render_paths_kernel launch", err);
    }
    if (L.traversal == 1) {  // MP_FLAG_TRAVERSAL_GROUPS
        const uint32_t grid = static_cast<uint32_t>(std::min<uint64_t>(want, static_cast<uint64_t>(L.cu_count) * 8));
        if (L.scene.inst_count != 0u) hipLaunchKernelGGL((render_tiles_kernel<1, true>), dim3(grid), dim3(256), lds, st, P);
        else hipLaunchKernelGGL((render_tiles_kernel<1, false>), dim3(grid), dim3(256), lds, st, P);
        return check(hipGetLastError(), "render_tiles_kernel launch", err);
    }
    // samples of one pixel in flight per pass: 16 = one DPP row per pixel (ordered sums by row_newbcast), a 2x2 pixel footprint per
    // wave and 4-pixel work units (measured best on MI355X for full frames: profiles/r01_notes.md)
    const uint32_t nspp = L.pass_end - L.pass_begin;  // samples per pixel in this launch
    int S = nspp >= 16 ? 16 : nspp >= 8 ? 8 : nspp >= 4 ? 4 : nspp >= 2 ? 2 : 1;
    // small launches (a rank's shard of a multi-GPU frame): 2-pixel units, so that the tail of the launch is half as long
    const bool small_launch = units * 16u < static_cast<uint64_t>(L.cu_count) * 32u * 24u;
    if (nspp >= 32 && small_launch) S = 32;
    // scenes whose traversal arrays exceed the 16 KB scalar data cache by far run 8 waves per SIMD, and at many samples per pixel
    // 32 samples of a pixel in flight (a 1x2 pixel footprint: measured 42.1 against 42.6 ms on the metric's frame, tools/s_sweep.py)
    const bool big = (static_cast<uint64_t>(L.scene.inner_count) * 256u + static_cast<uint64_t>(L.scene.packet_count) * 384u) > (1u << 20);
    if (big && nspp >= 128) S = 32;
    const bool obj = L.scene.inst_count != 0u;  // object group: one packet walk per member (instantiated for 16 and 1 samples in flight)
    const bool lds_stack = L.scene.stack_cap > L.scene.packet_stack_regs;
    // The per-unit mask cache (MaskCache) wants units of at least four passes: with it, the samples in flight follow the sample
    // count -- 16 from 64 spp on (metric's frame, 256 spp: 21.7 ms against 22.1 with 32; 64 spp: 6.3 against 6.4 with 8), 8 for
    // 32-63 spp (3.5 against 5.5 ms uncached at 32 spp), 4 for 16-31 (2.2 against 2.9 ms at 16 spp) -- and small launches keep
    // their 2-pixel units where those still have four passes
    const bool cache_ok = L.mask_cache != 0u && !lds_stack && !obj && L.scene.kind == 0u && L.scene.inner_count < (1u << 24) && L.scene.tris_bounded != 0u;
    if (cache_ok && nspp >= 16) S = (small_launch && nspp >= 128) ? 32 : nspp >= 64 ? 16 : nspp >= 32 ? 8 : 4;
    if (L.packet_samples) S = static_cast<int>(std::min<uint32_t>(L.packet_samples, 64u));
    if (obj) S = (S >= 16 && nspp >= 16) ? 16 : 1;
    P.lds_per_wave = lds_stack ? (L.scene.stack_cap - L.scene.packet_stack_regs) * 16u : 0u;  // one uint4 per entry beyond the register stack
    const uint32_t plds = P.lds_per_wave * 4;
    if (plds > 160 * 1024) { err = "scene too deep for the LDS traversal stack"; return MP_ERR_UNSUPPORTED; }
    const uint32_t per_cu = plds ? std::max<uint32_t>(1, std::min<uint32_t>(8, (160u * 1024u) / plds)) : 8u;
    if (S == 16 && L.rays_per_lane == 2 && !lds_stack && !obj && L.scene.kind == 0u && L.scene.stack_cap <= 64u) {
        // 128-ray walks: two rays per lane (8-pixel units)
        const uint64_t units2 = static_cast<uint64_t>(L.n_tiles) * ((L.tile_size + 3) / 4) * ((L.tile_size + 1) / 2);
        const uint32_t grid2 = static_cast<uint32_t>(std::min<uint64_t>((units2 + 3) / 4, static_cast<uint64_t>(L.cu_count) * 8));
        hipLaunchKernelGGL((render_tiles_packet2_kernel<6>), dim3(grid2), dim3(256), 0, st, P);
        return check(hipGetLastError(), "render_tiles_packet2_kernel launch", err);
    }
    const uint32_t grid = static_cast<uint32_t>(std::min<uint64_t>(want * S, static_cast<uint64_t>(L.cu_count) * per_cu));
#define MP_LAUNCH_PACKET(SV, W)                                                                                         \
    do {                                                                                                                \
        if (lds_stack) hipLaunchKernelGGL((render_tiles_packet_kernel<SV, true, W>), dim3(grid), dim3(256), plds, st, P); \
        else hipLaunchKernelGGL((render_tiles_packet_kernel<SV, false, W>), dim3(grid), dim3(256), 0, st, P);           \
    } while (0)
    // per-unit mask cache of the packet-level child rejection: units of at least four passes, stack in registers, node indices
    // that fit the cache tag, triangle coordinates within the bound of the triangle masks; 3 712 bytes of LDS per wave
#ifndef MP_MCACHE_WPE
#define MP_MCACHE_WPE 8
#endif
    // (every scene: with the triangle masks the teapot's frame gains too -- 9.9 against 11.8 ms)
    const bool mcache = cache_ok && (S == 4 || S == 8 || S == 16 || S == 32) && nspp >= 4u * static_cast<uint32_t>(S);
    if (mcache) {
        const uint32_t clds = 4u * kMaskCacheDwords * 4u;
        if (S == 32) hipLaunchKernelGGL((render_tiles_packet_kernel<32, false, MP_MCACHE_WPE, false, true>), dim3(grid), dim3(256), clds, st, P);
        else if (S == 8) hipLaunchKernelGGL((render_tiles_packet_kernel<8, false, MP_MCACHE_WPE, false, true>), dim3(grid), dim3(256), clds, st, P);
        else if (S == 4) hipLaunchKernelGGL((render_tiles_packet_kernel<4, false, MP_MCACHE_WPE, false, true>), dim3(grid), dim3(256), clds, st, P);
        else hipLaunchKernelGGL((render_tiles_packet_kernel<16, false, MP_MCACHE_WPE, false, true>), dim3(grid), dim3(256), clds, st, P);
        return check(hipGetLastError(), "render_tiles_packet_kernel launch", err);
    }
    if (obj) {
        if (S == 16 && lds_stack) hipLaunchKernelGGL((render_tiles_packet_kernel<16, true, 6, true>), dim3(grid), dim3(256), plds, st, P);
        else if (S == 16) hipLaunchKernelGGL((render_tiles_packet_kernel<16, false, 6, true>), dim3(grid), dim3(256), 0, st, P);
        else if (lds_stack) hipLaunchKernelGGL((render_tiles_packet_kernel<1, true, 6, true>), dim3(grid), dim3(256), plds, st, P);
        else hipLaunchKernelGGL((render_tiles_packet_kernel<1, false, 6, true>), dim3(grid), dim3(256), 0, st, P);
    } else if (S == 64) MP_LAUNCH_PACKET(64, 7);
    else if (S == 32 && big) MP_LAUNCH_PACKET(32, 8);
    else if (S == 32) MP_LAUNCH_PACKET(32, 7);
    else if (S == 16 && big) MP_LAUNCH_PACKET(16, 8);
    else if (S == 16) MP_LAUNCH_PACKET(16, 7);
    else if (S == 8) MP_LAUNCH_PACKET(8, 7);
    else if (S == 4) MP_LAUNCH_PACKET(4, 7);
    else if (S == 2) MP_LAUNCH_PACKET(2, 7);
    else MP_LAUNCH_PACKET(1, 7);
#undef MP_LAUNCH_PACKET
    return check(hipGetLastError(), "render_tiles_packet_kernel launch", err);
}

int launch_render_paths_wavefront(const RenderLaunch& L, void* stream, std::string& err) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (L.n_tiles == 0) return MP_OK;
    if (L.scene.kind != 0u || L.max_depth == 0) { err = "the staged path evaluation needs MP_FLAG_PATHS and a TriangleBvh scene or an object group"; return MP_ERR_UNSUPPORTED; }
    const bool obj = L.scene.inst_count != 0u;
    WfParams P;
    P.scene = L.scene;
    P.gen.s = L.sampler;
    P.gen.jitter_scale = uniform_inclusive_scale(-0.5f, 0.5f);
    P.gen.width = L.width;
    P.gen.spp = L.spp;
    P.gen.seed = mixed_seed(L.seed);
    P.tile_size = L.tile_size;
    P.max_depth = L.max_depth;
    P.out = L.d_out;
    P.inv_spp = 1.0f / static_cast<float>(L.spp);
    P.segments = L.d_segments;
    P.chunked = L.chunked ? 1u : 0u;
    const bool lds_stack = L.scene.stack_cap > L.scene.packet_stack_regs;
    P.lds_per_wave = lds_stack ? (L.scene.stack_cap - L.scene.packet_stack_regs) * 16u : 0u;  // one uint4 per entry beyond the register stack
    const uint32_t plds = P.lds_per_wave * 4;
    if (plds > 160 * 1024) { err = "scene too deep for the LDS traversal stack"; return MP_ERR_UNSUPPORTED; }
    const uint32_t per_cu = plds ? std::max<uint32_t>(1, std::min<uint32_t>(8, (160u * 1024u) / plds)) : 8u;
    // a batch = tb tiles x sc samples, about two million paths: enough rays per (tile, direction bin) to fill packets
    const uint32_t ts = L.tile_size, nspp = L.pass_end - L.pass_begin;
    const uint32_t sc = std::min<uint32_t>(nspp, 64u);
    const uint64_t per_tile = static_cast<uint64_t>(ts) * ts * sc;
    if (per_tile > (1ull << 28)) { err = "tile_size too large for the staged path evaluation"; return MP_ERR_UNSUPPORTED; }
    const uint32_t tb = static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>(L.n_tiles, (1ull << 21) / per_tile)));
    const uint32_t n_max = static_cast<uint32_t>(per_tile * tb), nbins = tb * kDirBins;
    // workspace: rng 32 + ray 24 + thr, L 8 (24 for three channels) + hit 16 + flags, key, idx 12 = 92 (108) bytes per path,
    // stream-ordered allocation
    const uint32_t nchan = L.scene.materials_rgb ? 3u : 1u;
    if (nchan == 3u && L.chunked) { err = "coloured / textured materials are not combined with MP_FLAG_CHUNKED_SUM"; return MP_ERR_UNSUPPORTED; }
    P.st.nchan = nchan;
    const size_t n64 = (static_cast<size_t>(n_max) + 63) & ~static_cast<size_t>(63);
    const size_t bytes = n64 * (88 + 8 * nchan) + (static_cast<size_t>(nbins) + 64) * 4 * 3;
    unsigned char* ws = nullptr;
    int rc = check(hipMallocAsync(reinterpret_cast<void**>(&ws), bytes, st), "hipMallocAsync(path state)", err);
    if (rc) return rc;
    unsigned char* w = ws;
    auto take = [&](size_t b) { unsigned char* r = w; w += b; return r; };
    P.st.rng = reinterpret_cast<uint64_t*>(take(n64 * 32));
    P.st.ray = reinterpret_cast<float*>(take(n64 * 24));
    P.st.thr = reinterpret_cast<float*>(take(n64 * 4 * nchan));
    P.st.L = reinterpret_cast<float*>(take(n64 * 4 * nchan));
    P.st.hit_t = reinterpret_cast<float*>(take(n64 * 4));
    P.st.hit_prim = reinterpret_cast<uint32_t*>(take(n64 * 4));
    P.st.hit_u = reinterpret_cast<float*>(take(n64 * 4));
    P.st.hit_v = reinterpret_cast<float*>(take(n64 * 4));
    P.st.hit_inst = reinterpret_cast<uint32_t*>(take(n64 * 4));
    P.st.flags = reinterpret_cast<uint32_t*>(take(n64 * 4));
    P.st.key = reinterpret_cast<uint32_t*>(take(n64 * 4));
    P.st.idx = reinterpret_cast<uint32_t*>(take(n64 * 4));
    P.st.hist = reinterpret_cast<uint32_t*>(take((static_cast<size_t>(nbins) + 64) * 4));
    P.st.offs = reinterpret_cast<uint32_t*>(take((static_cast<size_t>(nbins) + 64) * 4));
    P.st.cursor = reinterpret_cast<uint32_t*>(take((static_cast<size_t>(nbins) + 64) * 4));
    const uint32_t cus = static_cast<uint32_t>(L.cu_count);
    for (uint32_t tile_base = 0; tile_base < L.n_tiles && !rc; tile_base += tb) {
        const uint32_t ntb = std::min(tb, L.n_tiles - tile_base);
        P.tiles = L.d_tiles + tile_base;
        P.n_tiles = ntb;
        P.tile_base = tile_base;
        for (uint32_t s0 = L.pass_begin; s0 < L.pass_end && !rc; s0 += sc) {
            P.s0 = s0;
            P.sc = sc;
            P.s_end = L.pass_end;
            P.st.n = static_cast<uint32_t>(per_tile * ntb);
            P.st.nbins = ntb * kDirBins;
            P.carry_in = (L.carry_in || s0 > L.pass_begin) ? 1u : 0u;
            P.finalize = (L.finalize && s0 + sc >= L.pass_end) ? 1u : 0u;
            rc = check(hipMemsetAsync(P.st.hist, 0, (static_cast<size_t>(P.st.nbins) + 1) * 4, st), "hipMemsetAsync(histogram)", err);
            if (rc) break;
            // path slots of pixels outside a clipped tile are never written by the camera stage: they must read as dead
            rc = check(hipMemsetAsync(P.st.flags, 0, static_cast<size_t>(P.st.n) * 4, st), "hipMemsetAsync(path flags)", err);
            if (rc) break;
            const uint32_t units = ntb * ((ts + 1) / 2) * ((ts + 1) / 2);
            const uint32_t cam_grid = std::min<uint32_t>((units + 3) / 4, cus * per_cu);
            if (obj && lds_stack) hipLaunchKernelGGL((wf_camera_kernel<true, true>), dim3(cam_grid), dim3(256), plds, st, P);
            else if (obj) hipLaunchKernelGGL((wf_camera_kernel<false, true>), dim3(cam_grid), dim3(256), 0, st, P);
            else if (lds_stack) hipLaunchKernelGGL((wf_camera_kernel<true, false>), dim3(cam_grid), dim3(256), plds, st, P);
            else hipLaunchKernelGGL((wf_camera_kernel<false, false>), dim3(cam_grid), dim3(256), 0, st, P);
            const uint32_t flat_grid = std::min<uint32_t>((P.st.n + 255u) / 256u, cus * 16u);
            WfParams G = P;  // bounce stage: LDS ray queue + eight traversal stacks per wave
            G.lds_per_wave = lds_bytes_per_wave(L.scene.stack_cap);
            const uint32_t glds = G.lds_per_wave * 4;
            if (glds > 160 * 1024) { err = "scene too deep for the LDS traversal stacks"; rc = MP_ERR_UNSUPPORTED; break; }
            const uint32_t gper = std::max<uint32_t>(1, std::min<uint32_t>(8, (160u * 1024u) / glds));
            for (uint32_t depth = 1; depth <= L.max_depth; depth++) {
                P.depth = depth;
                if (nchan == 3u && obj) hipLaunchKernelGGL((wf_vertex_kernel<3, true>), dim3(flat_grid), dim3(256), 0, st, P);
                else if (nchan == 3u) hipLaunchKernelGGL((wf_vertex_kernel<3, false>), dim3(flat_grid), dim3(256), 0, st, P);
                else if (obj) hipLaunchKernelGGL((wf_vertex_kernel<1, true>), dim3(flat_grid), dim3(256), 0, st, P);
                else hipLaunchKernelGGL((wf_vertex_kernel<1, false>), dim3(flat_grid), dim3(256), 0, st, P);
                if (depth == L.max_depth) break;
                hipLaunchKernelGGL(wf_scan_kernel, dim3(1), dim3(1024), 0, st, P);
                hipLaunchKernelGGL(wf_scatter_kernel, dim3(flat_grid), dim3(256), 0, st, P);
                if (obj) hipLaunchKernelGGL(wf_trace_groups_kernel<true>, dim3(cus * gper), dim3(256), glds, st, G);
                else hipLaunchKernelGGL(wf_trace_groups_kernel<false>, dim3(cus * gper), dim3(256), glds, st, G);
            }
            const uint32_t px_grid = std::min<uint32_t>((ntb * ts * ts + 255u) / 256u, cus * 16u);
            hipLaunchKernelGGL(wf_accumulate_kernel, dim3(px_grid), dim3(256), 0, st, P);
            rc = check(hipGetLastError(), "staged path kernels launch", err);
        }
    }
    (void)hipFreeAsync(ws, st);
    return rc;
}

int launch_trace_rays(const DevScene& sc, const float* ox, const float* oy, const float* oz, const float* dx,
                      const float* dy, const float* dz, uint64_t n, const mp_hits_soa& hits, int cu_count, void* stream,
                      std::string& err) {
    if (n == 0) return MP_OK;
    TraceParams P;
    P.scene = sc;
    P.ox = ox; P.oy = oy; P.oz = oz; P.dx = dx; P.dy = dy; P.dz = dz;
    P.n = n;
    P.hits = hits;
    P.lds_per_wave = lds_bytes_per_wave(sc.stack_cap);
    const uint32_t lds = P.lds_per_wave * 4;
    if (lds > 160 * 1024) { err = "scene too deep for the LDS traversal stacks"; return MP_ERR_UNSUPPORTED; }
    const uint64_t chunks = (n + 63) / 64, want = (chunks + 3) / 4;
    const uint32_t grid = static_cast<uint32_t>(std::min<uint64_t>(want, static_cast<uint64_t>(cu_count) * 8));
    if (sc.inst_count != 0u) hipLaunchKernelGGL(trace_rays_kernel<true>, dim3(grid), dim3(256), lds, static_cast<hipStream_t>(stream), P);
    else hipLaunchKernelGGL(trace_rays_kernel<false>, dim3(grid), dim3(256), lds, static_cast<hipStream_t>(stream), P);
    return check(hipGetLastError(), "trace_rays_kernel launch", err);
}

int launch_set_u64(unsigned long long* d_ptr, unsigned long long value, void* stream, std::string& err) {
    hipLaunchKernelGGL(set_u64_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), d_ptr, value);
    return check(hipGetLastError(), "set_u64_kernel launch", err);
}

int launch_generate_rays(const mp_camera_sampler& s, uint32_t width, uint32_t spp, uint64_t seed, mp_block block,
                         uint32_t sample, float* ox, float* oy, float* oz, float* dx, float* dy, float* dz, void* stream,
                         std::string& err) {
    const uint64_t n = static_cast<uint64_t>(block.max_x - block.min_x) * (block.max_y - block.min_y);
    if (n == 0) return MP_OK;
    RayGen G;
    G.s = s;
    G.jitter_scale = uniform_inclusive_scale(-0.5f, 0.5f);
    G.width = width;
    G.spp = spp;
    G.seed = mixed_seed(seed);
    const uint32_t grid = static_cast<uint32_t>(std::min<uint64_t>((n + 255) / 256, 8192));
    hipLaunchKernelGGL(generate_rays_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), G, block, sample, ox,
                       oy, oz, dx, dy, dz);
    return check(hipGetLastError(), "generate_rays_kernel launch", err);
}

int launch_untile(uint32_t width, uint32_t height, uint32_t tile_size, const mp_block* d_tiles, uint32_t n_tiles,
                  const float* d_tiles_f32, float* d_image_f32, uint8_t* d_image_u8, void* stream, std::string& err,
                  uint32_t preview_mode, uint32_t preview_samples) {
    const uint64_t n = static_cast<uint64_t>(tile_size) * tile_size * n_tiles;
    if (n == 0) return MP_OK;
    const uint32_t grid = static_cast<uint32_t>(std::min<uint64_t>((n + 255) / 256, 8192));
    const uint32_t k = std::max<uint32_t>(preview_samples, 1u);
    hipLaunchKernelGGL(untile_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), width, height, tile_size,
                       d_tiles, n_tiles, d_tiles_f32, d_image_f32, d_image_u8, preview_mode, 1.0f / static_cast<float>(k),
                       1.0 / static_cast<double>(k));
    return check(hipGetLastError(), "untile_kernel launch", err);
}

int launch_quantise(const float* d_rgba_f32, uint8_t* d_rgba_u8, uint64_t n_pixels, void* stream, std::string& err) {
    if (n_pixels == 0) return MP_OK;
    const uint32_t grid = static_cast<uint32_t>(std::min<uint64_t>((n_pixels + 255) / 256, 8192));
    hipLaunchKernelGGL(quantise_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), d_rgba_f32, d_rgba_u8, n_pixels);
    return check(hipGetLastError(), "quantise_kernel launch", err);
}

}  // namespace mp

#ifdef MP_PROF_MISSES
extern "C" int mp_prof_read(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mp::g_prof), sizeof(unsigned long long) * 4) != hipSuccess) return 1;
    if (reset) { unsigned long long z[4] = {0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(mp::g_prof), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif

/*
 * minipath_oracle.c -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY (see header).
 * Compile: gcc -std=c11 -O3 -mavx2 -mfma -ffp-contract=off -fno-fast-math -fPIC -shared -pthread
 *
 * Conventions: f32 everywhere; "lane" loops of 8 mirror the reference's WideF32x8; fmaf() appears only where
 * the reference writes mul_add/mul_sub (util/simba.rs:57-67, compressed_geometry.rs:103-109).
 * (R) = recalled from a crate whose source is not in the reference tree; unverifiable offline.
 */
#define _GNU_SOURCE
#include "minipath_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define L8 8

/* ================================================================================================= */
/* RNG (R)                                                                                            */
/* ================================================================================================= */

static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

/* rand 0.9.3 rngs/xoshiro256plusplus.rs seed_from_u64: four SplitMix64 outputs (R). */
void mpo_rng_seed(mpo_rng *r, uint64_t state) {
    for (int i = 0; i < 4; i++) {
        state += 0x9e3779b97f4a7c15ull;
        uint64_t z = state;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        z = z ^ (z >> 31);
        r->s[i] = z;
    }
}

/* xoshiro256++ next_u64 (R) */
uint64_t mpo_rng_next_u64(mpo_rng *r) {
    uint64_t *s = r->s;
    uint64_t result = rotl64(s[0] + s[3], 23) + s[0];
    uint64_t t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl64(s[3], 45);
    return result;
}

/* "The lowest bits have some linear dependencies, so we use the upper bits instead" (R) */
uint32_t mpo_rng_next_u32(mpo_rng *r) { return (uint32_t)(mpo_rng_next_u64(r) >> 32); }

static inline float bits_to_f32(uint32_t b) {
    float f;
    memcpy(&f, &b, 4);
    return f;
}
static inline uint32_t f32_to_bits(float f) {
    uint32_t b;
    memcpy(&b, &f, 4);
    return b;
}

/* rand 0.9 UniformFloat: value in [0,1) with 23 random mantissa bits (R) */
static inline float rng_value0_1(mpo_rng *r) {
    uint32_t u = mpo_rng_next_u32(r);
    return bits_to_f32(0x3F800000u | (u >> 9)) - 1.0f;
}

/* UniformFloat::new_inclusive scale (R): scale = (high-low)/max_rand, reduced by ulps until
 * scale*max_rand + low <= high.  max_rand = 1 - 2^-23. */
static float uniform_inclusive_scale(float low, float high) {
    const float max_rand = 1.0f - FLT_EPSILON;
    float scale = (high - low) / max_rand;
    for (;;) {
        float top = scale * max_rand + low;
        if (!(top > high)) break;
        scale = bits_to_f32(f32_to_bits(scale) - 1u);
    }
    return scale;
}

/* camera.rs:178-179 : rng.random_range(-0.5..=0.5) */
float mpo_rng_range_pm_half(mpo_rng *r) {
    static float scale = 0.0f;
    if (scale == 0.0f) scale = uniform_inclusive_scale(-0.5f, 0.5f);
    float v = rng_value0_1(r);
    return v * scale + (-0.5f);
}

/* rand_distr 0.5.1 UnitDisc: rejection on Uniform::new(-1,1)^2, accept x1^2+x2^2 <= 1 (R).
 * Uniform::new(-1,1): scale = 2 (2*(1-2^-23) - 1 < 1, no reduction), sample = v01*2 + (-1). */
void mpo_rng_unit_disc(mpo_rng *r, float out[2]) {
    float x1, x2;
    for (;;) {
        x1 = rng_value0_1(r) * 2.0f + (-1.0f);
        x2 = rng_value0_1(r) * 2.0f + (-1.0f);
        if (x1 * x1 + x2 * x2 <= 1.0f) break;
    }
    out[0] = x1;
    out[1] = x2;
}

/* Build-defined seeded mode (the reference's RNG is OS-seeded, worker.rs:25):
 *   key = mix(seed) + ((y*W + x)*spp + s), all u64 wrapping,  mix(seed) = the first SplitMix64 output for state `seed`
 * (the value seed_from_u64(seed) puts into s[0]).  The seed is mixed before the sample index is added so that the streams of
 * consecutive seeds are unrelated: with a plain `seed + index` the frame of seed s+1 would be the frame of seed s shifted by
 * one sample.  A reference user reproduces it with SmallRng::seed_from_u64(key) at the top of render_sample (worker.rs:57). */
uint64_t mpo_seed_mix(uint64_t seed) {
    uint64_t z = seed + 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
uint64_t mpo_sample_key(uint64_t seed, uint32_t width, uint32_t spp, uint32_t x, uint32_t y, uint32_t s) {
    return mpo_seed_mix(seed) + (((uint64_t)y * (uint64_t)width + (uint64_t)x) * (uint64_t)spp + (uint64_t)s);
}

/* ================================================================================================= */
/* Small vector helpers (nalgebra semantics: unfused, left to right) (R)                              */
/* ================================================================================================= */

static inline float dot3(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline float norm3(const float a[3]) { return sqrtf(dot3(a, a)); }
static inline void cross3(const float a[3], const float b[3], float o[3]) {
    float x = a[1] * b[2] - a[2] * b[1];
    float y = a[2] * b[0] - a[0] * b[2];
    float z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline void normalize3(const float a[3], float o[3]) {
    float n = norm3(a);
    o[0] = a[0] / n; o[1] = a[1] / n; o[2] = a[2] / n;
}

/* ================================================================================================= */
/* geometry/mod.rs                                                                                   */
/* ================================================================================================= */

/* geometry/mod.rs:45-54 : Unit::new_normalize (division by the norm), inv = x==0 ? +inf : 1/x */
void mpo_ray_new(const float o[3], const float d[3], mpo_ray *out) {
    float n[3];
    normalize3(d, n);
    for (int k = 0; k < 3; k++) {
        out->o[k] = o[k];
        out->d[k] = n[k];
        out->inv[k] = (n[k] == 0.0f) ? INFINITY : 1.0f / n[k];
    }
}

/* geometry/mod.rs:56-58 */
void mpo_ray_point_at(const mpo_ray *r, float t, float out[3]) {
    for (int k = 0; k < 3; k++) out[k] = r->o[k] + r->d[k] * t;
}

/* wide f32x8::fast_min / fast_max lower to vminps / vmaxps: (a<b)?a:b and (a>b)?a:b -- the second operand is
 * returned when unordered (R).  No NaN reaches them in the slab test (aabb.rs:262-267 patches first). */
static inline float fast_min(float a, float b) { return a < b ? a : b; }
static inline float fast_max(float a, float b) { return a > b ? a : b; }

/* aabb.rs:254-284 */
void mpo_aabb8_intersect(const float bmin[3][8], const float bmax[3][8], const mpo_ray *ray, float max_t,
                         float t1[8], float t2[8]) {
    /* one loop over the 8 lanes with local results (no aliasing checks): gcc emits one AVX2 instruction per line, as f32x8 */
    const float ox = ray->o[0], oy = ray->o[1], oz = ray->o[2];
    const float ix = ray->inv[0], iy = ray->inv[1], iz = ray->inv[2];
    float r1[L8], r2[L8];
    for (int i = 0; i < L8; i++) {
        float ax = (bmin[0][i] - ox) * ix, ay = (bmin[1][i] - oy) * iy, az = (bmin[2][i] - oz) * iz;
        float bx = (bmax[0][i] - ox) * ix, by = (bmax[1][i] - oy) * iy, bz = (bmax[2][i] - oz) * iz;
        ax = (ax != ax) ? -INFINITY : ax; ay = (ay != ay) ? -INFINITY : ay; az = (az != az) ? -INFINITY : az; /* :262-264 */
        bx = (bx != bx) ? INFINITY : bx; by = (by != by) ? INFINITY : by; bz = (bz != bz) ? INFINITY : bz;    /* :265-267 */
        const float lox = fast_min(ax, bx), loy = fast_min(ay, by), loz = fast_min(az, bz); /* :270 */
        const float hix = fast_max(ax, bx), hiy = fast_max(ay, by), hiz = fast_max(az, bz); /* :271 */
        r1[i] = fast_max(fast_max(lox, 0.0f), fast_max(loy, loz));   /* :273-276 */
        r2[i] = fast_min(fast_min(hix, max_t), fast_min(hiy, hiz)); /* :277-280 */
    }
    for (int i = 0; i < L8; i++) { t1[i] = r1[i]; t2[i] = r2[i]; }
}

/* util/simba.rs:57-59 */
#define FMA_DOT(ax, ay, az, bx, by, bz) fmaf((az), (bz), fmaf((ay), (by), (ax) * (bx)))
/* util/simba.rs:61-67 : mul_sub(a,b,c) = a*b - c fused, c rounded first */
#define FMS(a, b, c) fmaf((a), (b), -(c))

/* triangle.rs:183-217 */
unsigned mpo_tri8_intersect(const float v0[3][8], const float v1[3][8], const float v2[3][8], const mpo_ray *ray,
                            float t[8], float u[8], float v[8]) {
    const float ox = ray->o[0], oy = ray->o[1], oz = ray->o[2];
    const float dx = ray->d[0], dy = ray->d[1], dz = ray->d[2];
    /* results go through locals (no aliasing with the inputs) and the lane mask is assembled after the arithmetic loop, so that
     * gcc vectorises the loop: 8 lanes = one AVX2 register, as the reference's f32x8 */
    int m8[L8];
    float t8[L8], u8[L8], v8[L8];
    for (int i = 0; i < L8; i++) {
        float e1x = v1[0][i] - v0[0][i], e1y = v1[1][i] - v0[1][i], e1z = v1[2][i] - v0[2][i];
        float e2x = v2[0][i] - v0[0][i], e2y = v2[1][i] - v0[1][i], e2z = v2[2][i] - v0[2][i];
        /* ray_cross_e2 = fma_cross(direction, e2) :198 */
        float hx = FMS(dy, e2z, dz * e2y);
        float hy = FMS(dz, e2x, dx * e2z);
        float hz = FMS(dx, e2y, dy * e2x);
        float det = FMA_DOT(e1x, e1y, e1z, hx, hy, hz); /* :199 */
        float inv_det = 1.0f / det;                     /* :201, may be inf */
        float sx = ox - v0[0][i], sy = oy - v0[1][i], sz = oz - v0[2][i]; /* :202 */
        float uu = inv_det * FMA_DOT(sx, sy, sz, hx, hy, hz);            /* :203 */
        /* s_cross_e1 = fma_cross(s, e1) :205 */
        float qx = FMS(sy, e1z, sz * e1y);
        float qy = FMS(sz, e1x, sx * e1z);
        float qz = FMS(sx, e1y, sy * e1x);
        float vv = inv_det * FMA_DOT(dx, dy, dz, qx, qy, qz);     /* :206 */
        float tt = inv_det * FMA_DOT(e2x, e2y, e2z, qx, qy, qz);  /* :207 */
        m8[i] = (uu >= 0.0f) & (vv >= 0.0f) & ((uu + vv) <= 1.0f); /* :209-211 */
        t8[i] = tt; u8[i] = uu; v8[i] = vv;
    }
    unsigned mask = 0;
    for (int i = 0; i < L8; i++) { mask |= (unsigned)m8[i] << i; t[i] = t8[i]; u[i] = u8[i]; v[i] = v8[i]; }
    return mask;
}

/* ================================================================================================= */
/* compressed_geometry.rs                                                                            */
/* ================================================================================================= */

static const float INV_U16_MAX = 1.0f / 65535.0f; /* :49 */

/* :48-51 */
float mpo_unit_interval_decompress(uint16_t q) { return (float)(int32_t)q * INV_U16_MAX; }

/* :25-46.  rounding 0 = wide round (vroundps nearest, ties-even) (R), 1 = floor, 2 = ceil.
 * blend -> fast_min(.,65535) -> fast_max(.,0) -> fast_trunc_int (cvttps2dq) -> as u16. */
uint16_t mpo_unit_interval_compress(float v, int rounding, int mask) {
    float x = v * 65535.0f;
    if (rounding == 0) x = nearbyintf(x); /* default FE_TONEAREST == ties-even */
    else if (rounding == 1) x = floorf(x);
    else x = ceilf(x);
    x = mask ? x : 0.0f;
    x = fast_min(x, 65535.0f); /* NaN -> 65535 (second operand) */
    x = fast_max(x, 0.0f);
    return (uint16_t)(int32_t)x;
}

/* RelativePoint8::decompress :95-110 : size.mul_add(relative, min) */
static inline float decompress_coord(uint16_t q, float size, float min) {
    return fmaf(size, mpo_unit_interval_decompress(q), min);
}
/* eight lanes at once (RelativePoint8 / RelativeBox8 decompress): a plain loop gcc turns into vpmovzxwd + vcvtdq2ps + vfmadd */
static inline void decompress8(const uint16_t *restrict q, float size, float min, float *restrict out) {
    for (int i = 0; i < L8; i++) out[i] = fmaf(size, (float)(int32_t)q[i] * INV_U16_MAX, min);
}
/* RelativePoint8::compress_internal :74-93 : relative = (p - min) / size */
static inline uint16_t compress_coord(float p, float min, float size, int rounding, int mask) {
    float rel = (p - min) / size;
    return mpo_unit_interval_compress(rel, rounding, mask);
}

/* util/mod.rs:6-31 */
int mpo_bit_iter(uint64_t bits, int out[64]) {
    int n = 0;
    while (bits) {
        out[n++] = __builtin_ctzll(bits);
        bits &= bits - 1;
    }
    return n;
}

/* triangle_bvh/mod.rs:70-80 */
uint32_t mpo_link_new_leaf(uint32_t index, uint32_t count, int *ok) {
    int good = count >= 1 && count <= MPO_LINK_MAX_COUNT && index <= MPO_LINK_MAX_INDEX;
    if (ok) *ok = good;
    return (index << MPO_LINK_COUNT_BITS) | count;
}
uint32_t mpo_link_new_inner(uint32_t index, int *ok) {
    if (ok) *ok = index <= MPO_LINK_MAX_INDEX;
    return index << MPO_LINK_COUNT_BITS;
}
/* triangle_bvh/mod.rs:89-113 */
int mpo_link_decode(uint32_t link, uint32_t *index, uint32_t *count) {
    if (link == MPO_LINK_NULL) return 0;
    uint32_t c = link & MPO_LINK_COUNT_MASK;
    if (index) *index = link >> MPO_LINK_COUNT_BITS;
    if (count) *count = c;
    return c == 0 ? 1 : 2;
}

/* ================================================================================================= */
/* camera.rs (nalgebra Isometry3 / UnitQuaternion semantics are (R))                                 */
/* ================================================================================================= */

/* UnitQuaternion * Vector3 (R): t = 2*(qv x v); v' = t*w + (qv x t) + v */
static void quat_rotate(const float q[4], const float v[3], float out[3]) {
    float t[3], c[3];
    cross3(q, v, t);
    t[0] *= 2.0f; t[1] *= 2.0f; t[2] *= 2.0f;
    cross3(q, t, c);
    for (int k = 0; k < 3; k++) out[k] = t[k] * q[3] + c[k] + v[k];
}

/* UnitQuaternion::from_rotation_matrix (R); m is row-major m[r][c] */
static void quat_from_matrix(const float m[3][3], float q[4]) {
    float tr = m[0][0] + m[1][1] + m[2][2];
    float w, i, j, k;
    if (tr > 0.0f) {
        float denom = sqrtf(tr + 1.0f) * 2.0f;
        w = 0.25f * denom;
        i = (m[2][1] - m[1][2]) / denom;
        j = (m[0][2] - m[2][0]) / denom;
        k = (m[1][0] - m[0][1]) / denom;
    } else if (m[0][0] > m[1][1] && m[0][0] > m[2][2]) {
        float denom = sqrtf(1.0f + m[0][0] - m[1][1] - m[2][2]) * 2.0f;
        w = (m[2][1] - m[1][2]) / denom;
        i = 0.25f * denom;
        j = (m[0][1] + m[1][0]) / denom;
        k = (m[0][2] + m[2][0]) / denom;
    } else if (m[1][1] > m[2][2]) {
        float denom = sqrtf(1.0f + m[1][1] - m[0][0] - m[2][2]) * 2.0f;
        w = (m[0][2] - m[2][0]) / denom;
        i = (m[0][1] + m[1][0]) / denom;
        j = 0.25f * denom;
        k = (m[1][2] + m[2][1]) / denom;
    } else {
        float denom = sqrtf(1.0f + m[2][2] - m[0][0] - m[1][1]) * 2.0f;
        w = (m[1][0] - m[0][1]) / denom;
        i = (m[0][2] + m[2][0]) / denom;
        j = (m[1][2] + m[2][1]) / denom;
        k = 0.25f * denom;
    }
    q[0] = i; q[1] = j; q[2] = k; q[3] = w;
}

/* Isometry3::look_at_rh(eye, target, up) (R):
 *   rotation = UnitQuaternion::face_towards(-(target-eye), up).inverse(); translation = rotation * (-eye) */
static void isometry_look_at_rh(const float eye[3], const float target[3], const float up[3], float q[4], float t[3]) {
    float dir[3] = {-(target[0] - eye[0]), -(target[1] - eye[1]), -(target[2] - eye[2])};
    float z[3], x[3], y[3], tmp[3];
    normalize3(dir, z);
    cross3(up, z, tmp);
    normalize3(tmp, x);
    cross3(z, x, tmp);
    normalize3(tmp, y);
    float m[3][3] = {{x[0], y[0], z[0]}, {x[1], y[1], z[1]}, {x[2], y[2], z[2]}};
    float qf[4];
    quat_from_matrix(m, qf);
    q[0] = -qf[0]; q[1] = -qf[1]; q[2] = -qf[2]; q[3] = qf[3]; /* inverse of a unit quaternion = conjugate */
    float ne[3] = {-eye[0], -eye[1], -eye[2]};
    quat_rotate(q, ne, t);
}

/* Isometry3::inverse (R): rotation^-1 ; translation = rotation^-1 * (-translation) */
static void isometry_inverse(const float q[4], const float t[3], float qo[4], float to[3]) {
    qo[0] = -q[0]; qo[1] = -q[1]; qo[2] = -q[2]; qo[3] = q[3];
    float nt[3] = {-t[0], -t[1], -t[2]};
    quat_rotate(qo, nt, to);
}

/* camera.rs:42-52 */
void mpo_camera_default(mpo_camera *c) {
    c->q[0] = c->q[1] = c->q[2] = 0.0f; c->q[3] = 1.0f;
    c->t[0] = c->t[1] = c->t[2] = 0.0f;
    c->focus_distance = INFINITY;
    c->sensor_is_width = 0;
    c->sensor_size = 24e-3f;
    c->focal_length = 50e-3f;
    c->f_number = 9.0f;
}

/* camera.rs:93-101 */
void mpo_camera_look_at(mpo_camera *c, const float eye[3], const float at[3], const float up[3]) {
    float q[4], t[3];
    isometry_look_at_rh(eye, at, up, q, t);
    isometry_inverse(q, t, c->q, c->t);
    float d[3] = {at[0] - eye[0], at[1] - eye[1], at[2] - eye[2]};
    c->focus_distance = norm3(d);
}

/* camera.rs:104-116 */
void mpo_camera_look_direction(mpo_camera *c, const float eye[3], const float fwd[3], const float up[3]) {
    float at[3] = {eye[0] + fwd[0], eye[1] + fwd[1], eye[2] + fwd[2]};
    float q[4], t[3];
    isometry_look_at_rh(eye, at, up, q, t);
    isometry_inverse(q, t, c->q, c->t);
}

/* camera.rs:119-121 with transform = Translation3 : (T * iso).translation = T + iso.translation (R) */
void mpo_camera_translate(mpo_camera *c, const float t[3]) {
    for (int k = 0; k < 3; k++) c->t[k] = t[k] + c->t[k];
}

/* camera.rs:148-171 */
void mpo_camera_basis(const mpo_camera *c, float center[3], float fwd[3], float up[3], float right[3]) {
    const float zero[3] = {0, 0, 0}, f[3] = {0, 0, -1}, u[3] = {0, 1, 0}, r[3] = {1, 0, 0};
    float rz[3];
    quat_rotate(c->q, zero, rz);
    for (int k = 0; k < 3; k++) center[k] = rz[k] + c->t[k];
    quat_rotate(c->q, f, fwd);
    quat_rotate(c->q, u, up);
    quat_rotate(c->q, r, right);
}

/* camera.rs:123-146 */
void mpo_camera_build_sampler(const mpo_camera *c, uint32_t w, uint32_t h, mpo_sampler *out) {
    float center[3], fwd[3], up[3], right[3];
    mpo_camera_basis(c, center, fwd, up, right);
    float rx = (float)w, ry = (float)h;
    float ps = c->sensor_is_width ? c->sensor_size / rx : c->sensor_size / ry;
    float uvx = ((rx - 1.0f) * ps) / 2.0f;
    float uvy = ((ry - 1.0f) * ps) / 2.0f;
    for (int k = 0; k < 3; k++) {
        out->center[k] = center[k];
        out->up[k] = up[k];
        out->right[k] = right[k];
        /* -forward * focal_length + right * uv.x - up * uv.y  (left to right) */
        out->film_origin_offset[k] = (-fwd[k]) * c->focal_length + right[k] * uvx - up[k] * uvy;
    }
    out->pixel_scale = ps;
    out->lens_radius = c->focal_length / (2.0f * c->f_number);
    out->lens_weight = c->focal_length / c->focus_distance;
}

/* camera.rs:176-191 */
void mpo_sample_ray(const mpo_sampler *s, uint32_t x, uint32_t y, mpo_rng *rng, mpo_ray *out) {
    float film_u = (float)x + mpo_rng_range_pm_half(rng);
    float film_v = (float)y + mpo_rng_range_pm_half(rng);
    float fv = film_v * s->pixel_scale, fu = film_u * s->pixel_scale;
    float fpo[3];
    for (int k = 0; k < 3; k++) fpo[k] = s->film_origin_offset[k] + s->up[k] * fv - s->right[k] * fu;
    float lens_uv[2];
    mpo_rng_unit_disc(rng, lens_uv);
    float a = s->lens_radius * lens_uv[0], b = s->lens_radius * lens_uv[1];
    float lens[3], dir[3], org[3];
    for (int k = 0; k < 3; k++) {
        lens[k] = s->right[k] * a + s->up[k] * b;
        dir[k] = lens[k] * s->lens_weight - fpo[k];
        org[k] = s->center[k] + lens[k];
    }
    mpo_ray_new(org, dir, out);
}

/* ================================================================================================= */
/* screen_block.rs                                                                                   */
/* ================================================================================================= */

/* :144-160 */
size_t mpo_divide_range(uint32_t start, uint32_t end, uint32_t tile, uint32_t *out, size_t cap) {
    uint32_t total = end - start;
    uint32_t full = total / tile;
    uint32_t n = full + ((full * tile != total) ? 1u : 0u);
    for (uint32_t i = 0; i < n && (size_t)i < cap; i++) {
        uint32_t ts = start + i * tile;
        uint32_t te = ts + tile;
        if (te > end) te = end;
        out[2 * i] = ts;
        out[2 * i + 1] = te;
    }
    return n;
}

typedef struct { float key; uint32_t idx; } tile_key;
static int tile_key_cmp(const void *a, const void *b) {
    const tile_key *x = a, *y = b;
    if (x->key < y->key) return -1;
    if (x->key > y->key) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx); /* stable, like sort_by_cached_key */
}

/* :46-81 */
size_t mpo_tile_ordering(uint32_t minx, uint32_t miny, uint32_t maxx, uint32_t maxy, uint32_t tile,
                         uint64_t shuffle_seed, uint32_t *out, size_t cap) {
    if (!(minx < maxx && miny < maxy)) return 0; /* is_empty :10-12 */
    uint32_t nx = (maxx - minx) / tile + (((maxx - minx) % tile) ? 1 : 0);
    uint32_t ny = (maxy - miny) / tile + (((maxy - miny) % tile) ? 1 : 0);
    size_t n = (size_t)nx * ny;
    if (!out) return n;
    uint32_t *tmp = malloc(n * 4 * sizeof(uint32_t));
    size_t k = 0;
    for (uint32_t j = 0; j < ny; j++) {
        uint32_t y0 = miny + j * tile, y1 = y0 + tile > maxy ? maxy : y0 + tile;
        for (uint32_t i = 0; i < nx; i++) {
            uint32_t x0 = minx + i * tile, x1 = x0 + tile > maxx ? maxx : x0 + tile;
            tmp[4 * k] = x0; tmp[4 * k + 1] = y0; tmp[4 * k + 2] = x1; tmp[4 * k + 3] = y1;
            k++;
        }
    }
    if (shuffle_seed == 0) {
        for (size_t i = 0; i < n && i < cap; i++) memcpy(out + 4 * i, tmp + 4 * i, 16);
    } else {
        /* centre-out + Exp(1/(0.1*|centre|)) noise :48,61-62,74-78 ; integer centre (min+max)/2 cast to f32 */
        float cx = (float)((minx + maxx) / 2), cy = (float)((miny + maxy) / 2);
        float scale = sqrtf(cx * cx + cy * cy) * 0.1f;
        mpo_rng r;
        mpo_rng_seed(&r, shuffle_seed);
        tile_key *keys = malloc(n * sizeof(tile_key));
        for (size_t i = 0; i < n; i++) {
            float tx = (float)((tmp[4 * i] + tmp[4 * i + 2]) / 2), ty = (float)((tmp[4 * i + 1] + tmp[4 * i + 3]) / 2);
            float ddx = cx - tx, ddy = cy - ty;
            /* inverse-CDF exponential; the reference uses rand_distr's ziggurat on a thread RNG */
            float u01 = rng_value0_1(&r);
            float e = -logf(1.0f - u01) * scale;
            keys[i].key = sqrtf(ddx * ddx + ddy * ddy) + e;
            keys[i].idx = (uint32_t)i;
        }
        qsort(keys, n, sizeof(tile_key), tile_key_cmp);
        for (size_t i = 0; i < n && i < cap; i++) memcpy(out + 4 * i, tmp + 4 * keys[i].idx, 16);
        free(keys);
    }
    free(tmp);
    return n;
}

/* :28-39, 104-128 */
size_t mpo_internal_points(uint32_t minx, uint32_t miny, uint32_t maxx, uint32_t maxy, uint32_t *out, size_t cap) {
    if (!(minx < maxx && miny < maxy)) return 0;
    size_t n = 0;
    uint32_t cx = minx, cy = miny;
    while (cy < maxy) {
        if (n < cap) { out[2 * n] = cx; out[2 * n + 1] = cy; }
        n++;
        cx++;
        if (cx >= maxx) { cx = minx; cy++; }
    }
    return n;
}

/* ================================================================================================= */
/* building.rs                                                                                       */
/* ================================================================================================= */

#define MPO_PATH_ALBEDO 0.75f /* default material of the build-defined path extension */
#define MPO_PATH_EPS 1e-4f

struct mpo_bvh {
    float bbox_min[3], bbox_max[3];
    uint32_t root;
    mpo_inner_node *inner; uint32_t n_inner, cap_inner;
    mpo_tri_packet *packets; uint32_t n_packets, cap_packets;
    mpo_tri_shading *shading; /* n_packets*8 */
    uint32_t cap_shading;
    float *vnormal, *vtex; uint32_t nv;
    uint32_t depth;
    uint32_t *material;      /* n_packets*8 : TriangleShadingData.material (mod.rs:44), 0 for padding */
    /* build-defined path extension: material table + sky radiance (see render_sample_paths_impl) */
    mpo_material *mats; uint32_t n_mats;
    float sky;
    /* build-defined Object: translated instances of this BVH (see scene_intersect); 0 = the plain TriangleBvh */
    uint32_t n_inst; float *inst_t;
    const struct mpo_bvh **inst_obj; /* member objects of a group (NULL: every member is this BVH) */
    float *inst_sph;                  /* group: 4 floats per member {center, radius}; radius < 0 = the member is a BVH */
    float *inst_q;                    /* group: unit quaternion (i, j, k, w) per member, NULL = no member is rotated */
};

typedef struct { float mn[3], mx[3]; } box3;

typedef struct {
    const float *pos; /* nv*3 */
    const float *nrm; /* nv*3 (zeros when absent) */
    uint32_t nv;
    mpo_bvh *bvh;
    char *err; size_t errcap; int failed;
} build_ctx;

static void set_err(build_ctx *c, const char *msg) {
    if (!c->failed && c->err && c->errcap) snprintf(c->err, c->errcap, "%s", msg);
    c->failed = 1;
}

static inline void box_empty(box3 *b) { /* AABB::default aabb.rs:139-157 */
    for (int k = 0; k < 3; k++) { b->mn[k] = INFINITY; b->mx[k] = -INFINITY; }
}
/* nalgebra inf/sup on f32 = simd_min/simd_max = f32::min/max (R); no NaN on this path */
static inline void box_extend(box3 *b, const float p[3]) { /* aabb.rs:219-222 */
    for (int k = 0; k < 3; k++) {
        b->mn[k] = fminf(b->mn[k], p[k]);
        b->mx[k] = fmaxf(b->mx[k], p[k]);
    }
}
static inline float box_surface_area(const box3 *b) { /* aabb.rs:247-251 */
    float sx = b->mx[0] - b->mn[0], sy = b->mx[1] - b->mn[1], sz = b->mx[2] - b->mn[2];
    return 2.0f * (sx * (sy + sz) + sy * sz);
}

/* triangle.rs:115-120 : sum of coords (fold from zero) / 3 */
static inline void tri_centroid(const build_ctx *c, const uint32_t t[4], float out[3]) {
    for (int k = 0; k < 3; k++) {
        float s = 0.0f + c->pos[3 * t[0] + k];
        s = s + c->pos[3 * t[1] + k];
        s = s + c->pos[3 * t[2] + k];
        out[k] = s / 3.0f;
    }
}

typedef struct { box3 box; size_t count; size_t parent; float sah; } split_bin;

/* building.rs:358-383 */
static float bin_sah(const box3 *box, size_t count) {
    const float B = 8.0f;
    size_t packet_count = (count + 7) / 8;
    float leaf_cost = (packet_count <= MPO_LINK_MAX_COUNT) ? 0.75f * (float)packet_count : INFINITY;
    float pc = (float)packet_count;
    float depth = floorf(logf(pc) / logf(B)); /* f32::log(self, base) = ln(self)/ln(base) (R) */
    float p = 1.0f;
    for (int i = 0; i < (int)depth; i++) p *= B; /* B.powi(depth): exact powers of 8 */
    float tree_cost = 1.0f * depth + 0.75f * ceilf(pc / p);
    return box_surface_area(box) * fminf(leaf_cost, tree_cost);
}

typedef struct {
    box3 box; float bin_size; size_t counts[3];
} bin_grid;

static inline size_t f32_to_usize(float x) { /* Rust `as usize`: saturating, NaN -> 0 */
    if (!(x > 0.0f)) return 0;
    if (x >= 18446744073709551616.0f) return SIZE_MAX;
    return (size_t)x;
}

/* building.rs:442-449 */
static inline size_t bin_index(const bin_grid *g, const float p[3]) {
    size_t cx = f32_to_usize(floorf((p[0] - g->box.mn[0]) / g->bin_size));
    size_t cy = f32_to_usize(floorf((p[1] - g->box.mn[1]) / g->bin_size));
    size_t cz = f32_to_usize(floorf((p[2] - g->box.mn[2]) / g->bin_size));
    return cx + cy * g->counts[0] + cz * (g->counts[0] * g->counts[1]);
}

typedef struct { size_t lo, hi; box3 box; } child_range;

/* building.rs:238-345.  Returns number of children (<= 8), reorders tris[0..n). */
static int split_triangles(build_ctx *c, uint32_t (*tris)[4], size_t n, child_range out[8]) {
    box3 cb;
    float ctr[3];
    tri_centroid(c, tris[0], ctr);
    for (int k = 0; k < 3; k++) cb.mn[k] = cb.mx[k] = ctr[k]; /* from_points :232-245 */
    for (size_t i = 1; i < n; i++) { tri_centroid(c, tris[i], ctr); box_extend(&cb, ctr); }

    size_t bin_count = n / 64;
    if (bin_count < 128) bin_count = 128;
    if (bin_count > 1024) bin_count = 1024; /* :248 */
    bin_grid g;
    g.box = cb;
    float sx = cb.mx[0] - cb.mn[0], sy = cb.mx[1] - cb.mn[1], sz = cb.mx[2] - cb.mn[2];
    float volume = sx * sy * sz;                         /* aabb.rs:243-245 */
    g.bin_size = cbrtf(volume / (float)bin_count);       /* :424 */
    g.counts[0] = f32_to_usize(ceilf(sx / g.bin_size));  /* :429 */
    g.counts[1] = f32_to_usize(ceilf(sy / g.bin_size));
    g.counts[2] = f32_to_usize(ceilf(sz / g.bin_size));
    size_t nb = g.counts[0] * g.counts[1] * g.counts[2];
    if (nb == 0 || nb > (size_t)1 << 26) { set_err(c, "degenerate centroid box: bin grid empty or too large (reference would panic)"); return 0; }

    split_bin *bins = malloc(nb * sizeof(split_bin));
    for (size_t i = 0; i < nb; i++) { box_empty(&bins[i].box); bins[i].count = 0; bins[i].parent = i; }
    size_t *tri_bin = malloc(n * sizeof(size_t));
    for (size_t i = 0; i < n; i++) {
        tri_centroid(c, tris[i], ctr);
        size_t bi = bin_index(&g, ctr);
        if (bi >= nb) { set_err(c, "centroid bin index out of range (reference would panic)"); free(bins); free(tri_bin); return 0; }
        tri_bin[i] = bi;
        for (int v = 0; v < 3; v++) box_extend(&bins[bi].box, &c->pos[3 * tris[i][v]]);
        bins[bi].count++;
    }
    size_t ng = 0;
    split_bin *groups = malloc(nb * sizeof(split_bin));
    for (size_t i = 0; i < nb; i++)
        if (bins[i].count > 0) { groups[ng] = bins[i]; groups[ng].sah = bin_sah(&bins[i].box, bins[i].count); ng++; }
    if (ng < 2) { set_err(c, "all centroids in a single bin (reference asserts groups.len() >= 2, building.rs:275)"); free(bins); free(groups); free(tri_bin); return 0; }

    while (ng > 2) { /* :278-293 */
        /* find_best_bin_merge :394-414 */
        size_t b1 = 0, b2 = 0;
        float best = -INFINITY;
        for (size_t i1 = 0; i1 < ng; i1++) {
            for (size_t i2 = i1 + 1; i2 < ng; i2++) {
                box3 m;
                for (int k = 0; k < 3; k++) {
                    m.mn[k] = fminf(groups[i1].box.mn[k], groups[i2].box.mn[k]);
                    m.mx[k] = fmaxf(groups[i1].box.mx[k], groups[i2].box.mx[k]);
                }
                float merged_sah = bin_sah(&m, groups[i1].count + groups[i2].count);
                float imp = groups[i1].sah + groups[i2].sah - merged_sah;
                if (imp > best) { b1 = i1; b2 = i2; best = imp; }
            }
        }
        if (best < 0.0f && ng <= MPO_INNER_NODE_CHILDREN) break;
        split_bin *g1 = &groups[b1], *g2 = &groups[b2];
        bins[g2->parent].parent = g1->parent;
        split_bin merged;
        for (int k = 0; k < 3; k++) {
            merged.box.mn[k] = fminf(g1->box.mn[k], g2->box.mn[k]);
            merged.box.mx[k] = fmaxf(g1->box.mx[k], g2->box.mx[k]);
        }
        merged.count = g1->count + g2->count;
        merged.parent = g1->parent;
        merged.sah = bin_sah(&merged.box, merged.count);
        groups[b1] = merged;
        groups[b2] = groups[ng - 1]; /* swap_remove */
        ng--;
    }

    /* :295-313 sort by disjoint-set root.  The reference uses sort_unstable_by_key (pattern-defeating
     * quicksort family, version dependent); topology and every box are independent of the intra-group order
     * (SURVEY A.9), so the restatement uses a STABLE order: groups ascending by root bin index, triangles
     * inside a group in their incoming order.  Only lane/packet positions inside a leaf can differ. */
    size_t *root = malloc(n * sizeof(size_t));
    size_t roots[8]; int nroots = 0;
    for (size_t i = 0; i < n; i++) {
        size_t r = tri_bin[i];
        while (bins[r].parent != r) r = bins[r].parent;
        root[i] = r;
        int found = 0;
        for (int j = 0; j < nroots; j++) if (roots[j] == r) { found = 1; break; }
        if (!found) {
            if (nroots >= 8) { set_err(c, "more than 8 groups after merging"); nroots = -1; break; }
            roots[nroots++] = r;
        }
    }
    int nchild = 0;
    if (nroots > 0) {
        for (int i = 1; i < nroots; i++) { /* ascending root index */
            size_t v = roots[i]; int j = i - 1;
            while (j >= 0 && roots[j] > v) { roots[j + 1] = roots[j]; j--; }
            roots[j + 1] = v;
        }
        uint32_t (*tmp)[4] = malloc(n * sizeof(*tmp));
        size_t w = 0;
        for (int j = 0; j < nroots; j++) {
            child_range *cr = &out[nchild];
            cr->lo = w;
            int first = 1;
            for (size_t i = 0; i < n; i++) {
                if (root[i] != roots[j]) continue;
                memcpy(tmp[w], tris[i], sizeof(*tmp));
                /* :315-344 chunk box = from_points(first triangle) then extend */
                if (first) {
                    for (int k = 0; k < 3; k++) cr->box.mn[k] = cr->box.mx[k] = c->pos[3 * tris[i][0] + k];
                    first = 0;
                }
                for (int v = 0; v < 3; v++) box_extend(&cr->box, &c->pos[3 * tris[i][v]]);
                w++;
            }
            cr->hi = w;
            nchild++;
        }
        memcpy(tris, tmp, n * sizeof(*tmp));
        free(tmp);
    }
    free(root); free(groups); free(bins); free(tri_bin);
    return nchild;
}

static uint32_t build_recursive(build_ctx *c, uint32_t (*tris)[4], size_t n, const box3 *enc, uint32_t depth);

/* building.rs:170-207 */
static uint32_t build_leaf(build_ctx *c, uint32_t (*tris)[4], size_t n, const box3 *enc) {
    mpo_bvh *b = c->bvh;
    if (n == 0) { set_err(c, "empty leaf (reference asserts !triangles.is_empty(), building.rs:178)"); return MPO_LINK_NULL; }
    float emin[3], esize[3];
    for (int k = 0; k < 3; k++) { emin[k] = enc->mn[k]; esize[k] = enc->mx[k] - enc->mn[k]; } /* aabb.rs:292-303 */
    uint32_t packet_count = (uint32_t)((n + 7) / 8);
    uint32_t first = b->n_packets;
    if (b->n_packets + packet_count > b->cap_packets) {
        while (b->n_packets + packet_count > b->cap_packets) b->cap_packets = b->cap_packets ? b->cap_packets * 2 : 64;
        b->packets = realloc(b->packets, (size_t)b->cap_packets * sizeof(mpo_tri_packet));
        b->shading = realloc(b->shading, (size_t)b->cap_packets * 8 * sizeof(mpo_tri_shading));
        b->material = realloc(b->material, (size_t)b->cap_packets * 8 * sizeof(uint32_t));
    }
    int ok = 1;
    uint32_t link = mpo_link_new_leaf(first, packet_count, &ok);
    if (!ok) { set_err(c, "leaf link out of range"); return MPO_LINK_NULL; }
    for (uint32_t p = 0; p < packet_count; p++) {
        mpo_tri_packet *pk = &b->packets[first + p];
        for (int lane = 0; lane < 8; lane++) {
            size_t ti = (size_t)p * 8 + lane;
            int mask = ti < n;
            mpo_tri_shading *sh = &b->shading[(size_t)(first + p) * 8 + lane];
            for (int v = 0; v < 3; v++)
                for (int k = 0; k < 3; k++) {
                    float pv = mask ? c->pos[3 * tris[ti][v] + k] : 0.0f; /* T::default() lanes, simba.rs:42 */
                    pk->v[v][k][lane] = compress_coord(pv, emin[k], esize[k], 0, mask);
                }
            if (mask) {
                int flat = 0;
                for (int v = 0; v < 3; v++) {
                    const float *nn = &c->nrm[3 * tris[ti][v]];
                    if (dot3(nn, nn) == 0.0f) flat = 1; /* :200 */
                    sh->vi[v] = tris[ti][v];
                }
                sh->flat = (uint32_t)flat;
                /* the reference writes `material: 0` (:201); ids given by the caller / `usemtl` travel with the triangle */
                b->material[(size_t)(first + p) * 8 + lane] = tris[ti][3];
            } else {
                sh->vi[0] = sh->vi[1] = sh->vi[2] = 0; sh->flat = 0; /* Default :203-204 */
                b->material[(size_t)(first + p) * 8 + lane] = 0;
            }
        }
    }
    b->n_packets += packet_count;
    return link;
}

/* building.rs:122-168 */
static uint32_t build_inner(build_ctx *c, uint32_t (*tris)[4], size_t n, const box3 *enc, uint32_t depth) {
    mpo_bvh *b = c->bvh;
    child_range ch[8];
    int nchild = split_triangles(c, tris, n, ch);
    if (c->failed || nchild <= 0) return MPO_LINK_NULL;
    if (b->n_inner == b->cap_inner) {
        b->cap_inner = b->cap_inner ? b->cap_inner * 2 : 16;
        b->inner = realloc(b->inner, (size_t)b->cap_inner * sizeof(mpo_inner_node));
    }
    uint32_t node_index = b->n_inner++;
    if (depth + 1 > b->depth) b->depth = depth + 1;
    int ok = 1;
    uint32_t self_link = mpo_link_new_inner(node_index, &ok);
    if (!ok) { set_err(c, "inner link out of range"); return MPO_LINK_NULL; }

    float emin[3], esize[3];
    for (int k = 0; k < 3; k++) { emin[k] = enc->mn[k]; esize[k] = enc->mx[k] - enc->mn[k]; }
    mpo_inner_node node;
    box3 dec[8];
    for (int i = 0; i < 8; i++) {
        int mask = i < nchild;
        for (int k = 0; k < 3; k++) {
            float pmin = mask ? ch[i].box.mn[k] : INFINITY;  /* AABB::default lanes */
            float pmax = mask ? ch[i].box.mx[k] : -INFINITY;
            node.bmin[k][i] = compress_coord(pmin, emin[k], esize[k], 1, mask); /* floor :128 */
            node.bmax[k][i] = compress_coord(pmax, emin[k], esize[k], 2, mask); /* ceil  :129 */
            dec[i].mn[k] = decompress_coord(node.bmin[k][i], esize[k], emin[k]);  /* :146 */
            dec[i].mx[k] = decompress_coord(node.bmax[k][i], esize[k], emin[k]);
        }
        node.link[i] = MPO_LINK_NULL;
    }
    for (int i = 0; i < nchild; i++) {
        node.link[i] = build_recursive(c, tris + ch[i].lo, ch[i].hi - ch[i].lo, &dec[i], depth + 1);
        if (c->failed) return MPO_LINK_NULL;
    }
    b->inner[node_index] = node;
    return self_link;
}

/* building.rs:109-120 */
static uint32_t build_recursive(build_ctx *c, uint32_t (*tris)[4], size_t n, const box3 *enc, uint32_t depth) {
    if (n <= MPO_LEAF_MAX_TRIANGLES) return build_leaf(c, tris, n, enc);
    return build_inner(c, tris, n, enc, depth);
}

static mpo_bvh *bvh_alloc(uint32_t nv, const float *nrm, const float *tex) {
    mpo_bvh *b = calloc(1, sizeof(mpo_bvh));
    b->root = MPO_LINK_NULL;
    b->nv = nv;
    b->vnormal = calloc((size_t)nv * 3 + 1, sizeof(float));
    b->vtex = calloc((size_t)nv * 3 + 1, sizeof(float));
    if (nrm) memcpy(b->vnormal, nrm, (size_t)nv * 3 * sizeof(float));
    if (tex) memcpy(b->vtex, tex, (size_t)nv * 3 * sizeof(float));
    b->mats = malloc(sizeof(mpo_material));
    memset(&b->mats[0], 0, sizeof(mpo_material));
    for (int k = 0; k < 3; k++) b->mats[0].albedo[k] = b->mats[0].albedo2[k] = MPO_PATH_ALBEDO;
    b->n_mats = 1;
    b->sky = 1.0f;
    return b;
}

/* building.rs:83-107 ; tri_mat (nullable) = material id per input triangle (the reference has only material 0, :201) */
mpo_bvh *mpo_bvh_build_mat(const float *pos, const float *nrm, const float *tex, uint32_t nv, const uint32_t *tri_idx,
                           const uint32_t *tri_mat, uint32_t nt, char *err, size_t errcap) {
    if (err && errcap) err[0] = 0;
    mpo_bvh *b = bvh_alloc(nv, nrm, tex);
    build_ctx c = {pos, b->vnormal, nv, b, err, errcap, 0};
    for (uint32_t i = 0; i < nt * 3; i++)
        if (tri_idx[i] >= nv) { set_err(&c, "vertex index out of range"); mpo_bvh_free(b); return NULL; }
    if (nt == 0) { set_err(&c, "no triangles (reference panics in build_leaf, building.rs:178)"); mpo_bvh_free(b); return NULL; }
    uint32_t (*tris)[4] = malloc((size_t)nt * sizeof(*tris));
    for (uint32_t i = 0; i < nt; i++) {
        tris[i][0] = tri_idx[3 * i]; tris[i][1] = tri_idx[3 * i + 1]; tris[i][2] = tri_idx[3 * i + 2];
        tris[i][3] = tri_mat ? tri_mat[i] : 0u;
    }
    box3 bb;
    for (int k = 0; k < 3; k++) bb.mn[k] = bb.mx[k] = pos[3 * tris[0][0] + k];
    for (uint32_t i = 0; i < nt; i++)
        for (int v = 0; v < 3; v++) box_extend(&bb, &pos[3 * tris[i][v]]);
    for (int k = 0; k < 3; k++) { b->bbox_min[k] = bb.mn[k]; b->bbox_max[k] = bb.mx[k]; }
    b->root = build_recursive(&c, tris, nt, &bb, 0);
    free(tris);
    if (c.failed) { mpo_bvh_free(b); return NULL; }
    return b;
}

mpo_bvh *mpo_bvh_build(const float *pos, const float *nrm, const float *tex, uint32_t nv, const uint32_t *tri_idx,
                       uint32_t nt, char *err, size_t errcap) {
    return mpo_bvh_build_mat(pos, nrm, tex, nv, tri_idx, NULL, nt, err, errcap);
}

/* A TriangleBvh handed over as its reference-layout arrays (triangle_bvh/mod.rs:20-53): what a reference user's own tree holds.
 * Arrays are copied.  `material` may be NULL (all 0, building.rs:201).  Links are validated so that traversal cannot leave the
 * arrays; `depth` is recomputed. */
static uint32_t arrays_depth(const mpo_bvh *b, uint32_t link, uint32_t level, int *ok) {
    uint32_t index, count;
    int kind = mpo_link_decode(link, &index, &count);
    if (kind == 0) return 0;
    if (kind == 2) { if ((uint64_t)index + count > b->n_packets) *ok = 0; return 0; }
    if (index >= b->n_inner || level > 64) { *ok = 0; return 0; }
    uint32_t d = 0;
    for (int i = 0; i < 8 && *ok; i++) {
        uint32_t l = b->inner[index].link[i];
        if ((l & MPO_LINK_COUNT_MASK) == 0 && l != MPO_LINK_NULL && (l >> 3) <= index) { *ok = 0; break; } /* children follow parents */
        uint32_t s = arrays_depth(b, l, level + 1, ok);
        if (s > d) d = s;
    }
    return d + 1;
}

mpo_bvh *mpo_bvh_from_arrays(const mpo_inner_node *inner, uint32_t n_inner, const mpo_tri_packet *packets, uint32_t n_packets,
                             const mpo_tri_shading *shading, const uint32_t *material, const float *vnormal, const float *vtex,
                             uint32_t nv, uint32_t root, const float bmin[3], const float bmax[3], char *err, size_t errcap) {
    if (err && errcap) err[0] = 0;
    mpo_bvh *b = bvh_alloc(nv, vnormal, vtex);
    b->n_inner = b->cap_inner = n_inner;
    b->n_packets = b->cap_packets = n_packets;
    b->inner = malloc(((size_t)n_inner + 1) * sizeof(mpo_inner_node));
    b->packets = malloc(((size_t)n_packets + 1) * sizeof(mpo_tri_packet));
    b->shading = malloc(((size_t)n_packets * 8 + 1) * sizeof(mpo_tri_shading));
    b->material = calloc((size_t)n_packets * 8 + 1, sizeof(uint32_t));
    if (n_inner) memcpy(b->inner, inner, (size_t)n_inner * sizeof(mpo_inner_node));
    if (n_packets) memcpy(b->packets, packets, (size_t)n_packets * sizeof(mpo_tri_packet));
    if (n_packets) memcpy(b->shading, shading, (size_t)n_packets * 8 * sizeof(mpo_tri_shading));
    if (material && n_packets) memcpy(b->material, material, (size_t)n_packets * 8 * sizeof(uint32_t));
    for (int k = 0; k < 3; k++) { b->bbox_min[k] = bmin[k]; b->bbox_max[k] = bmax[k]; }
    b->root = root;
    int ok = 1;
    for (size_t i = 0; i < (size_t)n_packets * 8 && ok; i++)
        for (int v = 0; v < 3; v++) if (b->shading[i].vi[v] >= (nv ? nv : 1u)) ok = 0;
    if (ok) b->depth = arrays_depth(b, root, 0, &ok);
    if (!ok) {
        if (err && errcap) snprintf(err, errcap, "%s", "arrays do not form a TriangleBvh (link or vertex index out of range)");
        mpo_bvh_free(b);
        return NULL;
    }
    return b;
}

int mpo_bvh_set_materials(mpo_bvh *b, const mpo_material *table, uint32_t n, float sky) {
    if (!b || !table || n == 0) return 0;
    for (size_t i = 0; i < (size_t)b->n_packets * 8; i++) if (b->material[i] >= n) return 0;
    free(b->mats);
    b->mats = malloc((size_t)n * sizeof(mpo_material));
    memcpy(b->mats, table, (size_t)n * sizeof(mpo_material));
    b->n_mats = n;
    b->sky = sky;
    return 1;
}
const uint32_t *mpo_bvh_tri_material(const mpo_bvh *b) { return b->material; }
uint32_t mpo_bvh_material_count(const mpo_bvh *b) {
    uint32_t m = 0;
    for (size_t i = 0; i < (size_t)b->n_packets * 8; i++) if (b->material[i] > m) m = b->material[i];
    return m + 1;
}

void mpo_bvh_free(mpo_bvh *b) {
    if (!b) return;
    free(b->inner); free(b->packets); free(b->shading); free(b->vnormal); free(b->vtex); free(b->material); free(b->mats); free(b->inst_t); free(b->inst_obj); free(b->inst_sph); free(b->inst_q);
    free(b);
}

/* ---- OBJ loading: building.rs:36-81 over obj 0.10.2 (R: triangles only, file order, f32 parse) --------- */

typedef struct { int64_t p, t, n; uint32_t index; int used; } vkey;

static uint64_t vkey_hash(int64_t p, int64_t t, int64_t n) {
    uint64_t h = (uint64_t)p * 0x9e3779b97f4a7c15ull;
    h ^= ((uint64_t)t + 0x7f4a7c15ull) * 0xbf58476d1ce4e5b9ull;
    h ^= ((uint64_t)n + 0x1ce4e5b9ull) * 0x94d049bb133111ebull;
    return h ^ (h >> 29);
}

typedef struct { float *d; size_t n, cap; } fvec;
static void fvec_push(fvec *v, float x) {
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 256; v->d = realloc(v->d, v->cap * sizeof(float)); }
    v->d[v->n++] = x;
}
typedef struct { uint32_t *d; size_t n, cap; } uvec;
static void uvec_push(uvec *v, uint32_t x) {
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 256; v->d = realloc(v->d, v->cap * sizeof(uint32_t)); }
    v->d[v->n++] = x;
}

/* parse "a", "a/b", "a//c", "a/b/c" ; 1-based, negative = relative to the current count. -1 => absent */
static int parse_index_tuple(const char *tok, size_t np, size_t nt, size_t nn, int64_t out[3]) {
    out[0] = out[1] = out[2] = -1;
    const char *s = tok;
    for (int f = 0; f < 3; f++) {
        if (*s == 0) break;
        if (*s == '/') { s++; continue; }
        char *end;
        long long v = strtoll(s, &end, 10);
        if (end == s) return 0;
        size_t cnt = f == 0 ? np : (f == 1 ? nt : nn);
        int64_t idx = v > 0 ? (int64_t)v - 1 : (int64_t)cnt + v;
        if (idx < 0 || (size_t)idx >= cnt) return 0;
        out[f] = idx;
        s = end;
        if (*s == '/') s++;
        else break;
    }
    return out[0] >= 0;
}

mpo_bvh *mpo_bvh_from_obj(const char *path, char *err, size_t errcap) {
    if (err && errcap) err[0] = 0;
    FILE *f = fopen(path, "r");
    if (!f) { if (err) snprintf(err, errcap, "Failed to read file: %s", path); return NULL; }
    fvec P = {0}, T = {0}, N = {0};
    fvec vpos = {0}, vnrm = {0}, vtex = {0};
    uvec tri = {0}, tmat = {0};
    /* `usemtl NAME`: material ids in first-seen order of the names, starting at 1; faces before any usemtl keep the
     * reference's `material: 0` (building.rs:201 -- the reference ignores usemtl; ids feed the build-defined extension only) */
    char **mnames = NULL; uint32_t n_mnames = 0, cur_mat = 0;
    size_t hcap = 1 << 12, hn = 0;
    vkey *ht = calloc(hcap, sizeof(vkey));
    char *line = NULL; size_t lcap = 0;
    int bad = 0;
    while (getline(&line, &lcap, f) >= 0) {
        char *s = line;
        while (*s == ' ' || *s == '\t') s++;
        if (s[0] == 'v' && (s[1] == ' ' || s[1] == '\t')) {
            char *e = s + 1;
            for (int k = 0; k < 3; k++) fvec_push(&P, strtof(e, &e));
        } else if (s[0] == 'v' && s[1] == 'n' && (s[2] == ' ' || s[2] == '\t')) {
            char *e = s + 2;
            for (int k = 0; k < 3; k++) fvec_push(&N, strtof(e, &e));
        } else if (s[0] == 'v' && s[1] == 't' && (s[2] == ' ' || s[2] == '\t')) {
            char *e = s + 2;
            for (int k = 0; k < 2; k++) fvec_push(&T, strtof(e, &e));
        } else if (strncmp(s, "usemtl", 6) == 0 && (s[6] == ' ' || s[6] == '\t')) {
            char *nm = s + 6;
            while (*nm == ' ' || *nm == '\t') nm++;
            size_t L = strlen(nm);
            while (L > 0 && (nm[L - 1] == '\n' || nm[L - 1] == '\r' || nm[L - 1] == ' ' || nm[L - 1] == '\t')) nm[--L] = 0;
            uint32_t id = 0;
            for (uint32_t i = 0; i < n_mnames; i++) if (strcmp(mnames[i], nm) == 0) { id = i + 1; break; }
            if (id == 0) {
                mnames = realloc(mnames, ((size_t)n_mnames + 1) * sizeof(char *));
                mnames[n_mnames] = malloc(L + 1);
                memcpy(mnames[n_mnames], nm, L + 1);
                id = ++n_mnames;
            }
            cur_mat = id;
        } else if (s[0] == 'f' && (s[1] == ' ' || s[1] == '\t')) {
            int64_t tup[64][3]; int nvtx = 0;
            char *save = NULL;
            for (char *tok = strtok_r(s + 1, " \t\r\n", &save); tok; tok = strtok_r(NULL, " \t\r\n", &save)) {
                if (nvtx >= 64) break;
                if (!parse_index_tuple(tok, P.n / 3, T.n / 2, N.n / 3, tup[nvtx])) { bad = 1; break; }
                nvtx++;
            }
            if (bad) break;
            if (nvtx != 3) continue; /* "non-triangle primitive!" :43-46 */
            for (int v = 0; v < 3; v++) {
                /* IndexMap entry: first-seen order :48-67 */
                if ((hn + 1) * 2 > hcap) {
                    size_t ncap = hcap * 2;
                    vkey *nh = calloc(ncap, sizeof(vkey));
                    for (size_t i = 0; i < hcap; i++) if (ht[i].used) {
                        size_t h = vkey_hash(ht[i].p, ht[i].t, ht[i].n) & (ncap - 1);
                        while (nh[h].used) h = (h + 1) & (ncap - 1);
                        nh[h] = ht[i];
                    }
                    free(ht); ht = nh; hcap = ncap;
                }
                size_t h = vkey_hash(tup[v][0], tup[v][1], tup[v][2]) & (hcap - 1);
                while (ht[h].used && !(ht[h].p == tup[v][0] && ht[h].t == tup[v][1] && ht[h].n == tup[v][2])) h = (h + 1) & (hcap - 1);
                if (!ht[h].used) {
                    ht[h].used = 1; ht[h].p = tup[v][0]; ht[h].t = tup[v][1]; ht[h].n = tup[v][2];
                    ht[h].index = (uint32_t)hn++;
                    for (int k = 0; k < 3; k++) fvec_push(&vpos, P.d[3 * tup[v][0] + k]);
                    if (tup[v][1] >= 0) { fvec_push(&vtex, T.d[2 * tup[v][1]]); fvec_push(&vtex, T.d[2 * tup[v][1] + 1]); fvec_push(&vtex, 0.0f); }
                    else { for (int k = 0; k < 3; k++) fvec_push(&vtex, 0.0f); } /* TexturePoint::origin */
                    if (tup[v][2] >= 0) {
                        float nn[3];
                        normalize3(&N.d[3 * tup[v][2]], nn); /* .normalize() :62 */
                        for (int k = 0; k < 3; k++) fvec_push(&vnrm, nn[k]);
                    } else { for (int k = 0; k < 3; k++) fvec_push(&vnrm, 0.0f); } /* WorldVector::zeros */
                }
                uvec_push(&tri, ht[h].index);
            }
            uvec_push(&tmat, cur_mat);
        }
    }
    free(line);
    fclose(f);
    mpo_bvh *b = NULL;
    if (bad) { if (err) snprintf(err, errcap, "Failed to parse file: %s", path); }
    else b = mpo_bvh_build_mat(vpos.d, vnrm.d, vtex.d, (uint32_t)hn, tri.d, tmat.d, (uint32_t)(tri.n / 3), err, errcap);
    for (uint32_t i = 0; i < n_mnames; i++) free(mnames[i]);
    free(mnames); free(tmat.d);
    free(P.d); free(T.d); free(N.d); free(vpos.d); free(vnrm.d); free(vtex.d); free(tri.d); free(ht);
    return b;
}

uint32_t mpo_bvh_root(const mpo_bvh *b) { return b->root; }
void mpo_bvh_bbox(const mpo_bvh *b, float bmin[3], float bmax[3]) {
    for (int k = 0; k < 3; k++) { bmin[k] = b->bbox_min[k]; bmax[k] = b->bbox_max[k]; }
}
uint32_t mpo_bvh_inner_count(const mpo_bvh *b) { return b->n_inner; }
uint32_t mpo_bvh_packet_count(const mpo_bvh *b) { return b->n_packets; }
uint32_t mpo_bvh_vertex_count(const mpo_bvh *b) { return b->nv; }
uint32_t mpo_bvh_depth(const mpo_bvh *b) { return b->depth; }
const mpo_inner_node *mpo_bvh_inner_nodes(const mpo_bvh *b) { return b->inner; }
const mpo_tri_packet *mpo_bvh_packets(const mpo_bvh *b) { return b->packets; }
const mpo_tri_shading *mpo_bvh_tri_shading(const mpo_bvh *b) { return b->shading; }
const float *mpo_bvh_vertex_normals(const mpo_bvh *b) { return b->vnormal; }
const float *mpo_bvh_vertex_tex(const mpo_bvh *b) { return b->vtex; }

/* ================================================================================================= */
/* ray_bvh_intersection.rs                                                                           */
/* ================================================================================================= */

typedef struct { uint32_t link; float mn[3], size[3]; float t1; } stack_entry; /* :19-23 */
typedef struct { stack_entry *e; size_t n, cap; } stack_cache;

static inline void stack_push(stack_cache *s, const stack_entry *x) {
    if (s->n == s->cap) { s->cap = s->cap ? s->cap * 2 : 64; s->e = realloc(s->e, s->cap * sizeof(stack_entry)); }
    s->e[s->n++] = *x;
}

typedef struct { float t, u, v; float gn[3]; uint64_t prim; } leaf_hit; /* :165-171 */

/* optional per-ray operation trace for scheduling studies (tools/sim_sched.py): one byte per stack pop:
 * 0 = culled/null, 1 = inner node tested, 8+k = leaf with k packets tested.  Not part of the restatement. */
static _Thread_local uint8_t *g_ops = NULL;
static _Thread_local uint32_t *g_links = NULL;
static _Thread_local size_t g_ops_cap = 0, g_ops_n = 0;
static _Thread_local uint32_t g_cur_link = 0;
static inline void op_rec(uint8_t v) {
    if (g_ops && g_ops_n < g_ops_cap) { if (g_links) g_links[g_ops_n] = g_cur_link; g_ops[g_ops_n++] = v; }
}

static void bvh_intersect_one(const mpo_bvh *b, const mpo_ray *ray, stack_cache *st, mpo_hit *out, mpo_counters *cnt) {
    st->n = 0;
    stack_entry root;
    root.link = b->root;
    for (int k = 0; k < 3; k++) { root.mn[k] = b->bbox_min[k]; root.size[k] = b->bbox_max[k] - b->bbox_min[k]; }
    root.t1 = -INFINITY;
    stack_push(st, &root); /* :28-32 */

    leaf_hit best = {FLT_MAX, 0, 0, {0, 0, 0}, MPO_NO_TRIANGLE}; /* :34-37 */
    if (cnt) cnt->rays++;

    while (st->n > 0) { /* :39 */
        if (cnt && st->n > cnt->max_stack) cnt->max_stack = st->n;
        stack_entry e = st->e[--st->n];
        if (cnt) cnt->stack_pops++;
        g_cur_link = e.link;
        if (e.t1 > best.t) { op_rec(0); continue; } /* :40-44 */
        uint32_t index, count;
        int kind = mpo_link_decode(e.link, &index, &count);
        if (kind == 0) { op_rec(0); continue; } /* Null :49 */
        op_rec(kind == 1 ? 1 : (uint8_t)(8 + count));
        if (kind == 1) {
            /* InnerNode::intersect :149-162 */
            const mpo_inner_node *node = &b->inner[index];
            if (cnt) cnt->inner_visited++;
            float bmin[3][8], bmax[3][8], t1[8], t2[8];
            for (int k = 0; k < 3; k++) {
                decompress8(node->bmin[k], e.size[k], e.mn[k], bmin[k]);
                decompress8(node->bmax[k], e.size[k], e.mn[k], bmax[k]);
            }
            mpo_aabb8_intersect(bmin, bmax, ray, best.t, t1, t2);
            for (int i = 0; i < L8; i++) { /* bit_iter ascending :158-161 */
                if (!(t1[i] <= t2[i])) continue;
                stack_entry c;
                c.link = node->link[i];
                for (int k = 0; k < 3; k++) { c.mn[k] = bmin[k][i]; c.size[k] = bmax[k][i] - bmin[k][i]; } /* :157 */
                c.t1 = t1[i];
                stack_push(st, &c);
            }
        } else {
            /* intersect_triangles :104-140 */
            const float max_t = best.t;
            leaf_hit lb = {INFINITY, 0, 0, {0, 0, 0}, MPO_NO_TRIANGLE};
            for (uint32_t j = index; j < index + count; j++) {
                const mpo_tri_packet *pk = &b->packets[j];
                if (cnt) cnt->packets_tested++;
                float v[3][3][8], t[8], u[8], vv[8];
                for (int a = 0; a < 3; a++)
                    for (int k = 0; k < 3; k++) decompress8(pk->v[a][k], e.size[k], e.mn[k], v[a][k]);
                unsigned mask = mpo_tri8_intersect(v[0], v[1], v[2], ray, t, u, vv);
                for (int i = 0; i < L8; i++) {
                    if (!((mask >> i) & 1u)) continue;
                    if (!(t[i] >= 0.0f && t[i] <= max_t)) continue; /* :125 */
                    if (t[i] < lb.t) { /* :129-135 */
                        lb.t = t[i];
                        lb.prim = (uint64_t)j * 8 + (uint64_t)i;
                        lb.u = u[i]; lb.v = vv[i];
                        /* Triangle::normal triangle.rs:141-144 : unfused cross of the decompressed edges */
                        float e1[3], e2[3];
                        for (int k = 0; k < 3; k++) { e1[k] = v[1][k][i] - v[0][k][i]; e2[k] = v[2][k][i] - v[0][k][i]; }
                        cross3(e1, e2, lb.gn);
                    }
                }
            }
            if (lb.t < best.t) best = lb; /* :59-61 */
        }
    }

    memset(out, 0, sizeof(*out));
    out->prim = best.prim;
    if (best.prim == MPO_NO_TRIANGLE) { out->hit = 0; out->t = best.t; return; } /* :66-67 */
    /* :69-94 */
    const mpo_tri_shading *sh = &b->shading[best.prim];
    float n[3], tx[3];
    float w = 1.0f - best.u - best.v; /* triangle.rs:235 */
    if (sh->flat) {
        n[0] = best.gn[0]; n[1] = best.gn[1]; n[2] = best.gn[2];
    } else {
        const float *n0 = &b->vnormal[3 * sh->vi[0]], *n1 = &b->vnormal[3 * sh->vi[1]], *n2 = &b->vnormal[3 * sh->vi[2]];
        for (int k = 0; k < 3; k++) n[k] = n0[k] * w + n1[k] * best.u + n2[k] * best.v; /* triangle.rs:236 */
    }
    {
        const float *t0 = &b->vtex[3 * sh->vi[0]], *t1 = &b->vtex[3 * sh->vi[1]], *t2 = &b->vtex[3 * sh->vi[2]];
        for (int k = 0; k < 3; k++) tx[k] = t0[k] * w + t1[k] * best.u + t2[k] * best.v;
    }
    out->hit = 1;
    out->t = best.t; out->u = best.u; out->v = best.v;
    for (int k = 0; k < 3; k++) { out->gn[k] = best.gn[k]; out->tex[k] = tx[k]; }
    normalize3(n, out->normal);
    mpo_ray_point_at(ray, best.t, out->point);
    out->material = b->material[best.prim]; /* always 0 in the reference (building.rs:201) */
}

/* BUILD-DEFINED Object (scene/mod.rs:7-10 `trait Object`; the reference has one object per Scene and no transforms): a list of
 * members {TriangleBvh, translation} (mpo_bvh_set_group; mpo_bvh_set_instances = every member is this BVH).  intersect = for
 * every member in order: the member's own intersect with the ray moved into the member's frame (origin - translation;
 * direction, hence t, unchanged), closest wins with a strict `<` (the first member keeps ties); HitRecord.point = point_at(t)
 * of the WORLD ray; normal / tex / material id are the member's, prim is the triangle index inside the member.  The material
 * table and the sky radiance of the path extension are the container's (this BVH's). */
static void bvh_intersect_impl(const mpo_bvh *b, const mpo_ray *ray, stack_cache *st, mpo_hit *out, mpo_counters *cnt) {
    if (b->n_inst == 0) { bvh_intersect_one(b, ray, st, out, cnt); return; }
    mpo_hit best;
    memset(&best, 0, sizeof(best));
    best.prim = MPO_NO_TRIANGLE;
    best.t = FLT_MAX;
    for (uint32_t k = 0; k < b->n_inst; k++) {
        mpo_ray r2 = *ray;
        for (int i = 0; i < 3; i++) r2.o[i] = ray->o[i] - b->inst_t[3 * k + i];
        const float *q = b->inst_q ? &b->inst_q[4 * k] : NULL;
        if (q) { /* world = q * local + translation: the ray goes through the inverse rotation; the direction is NOT re-normalised
                  * (t keeps the world ray's scale), inv_direction by Ray::new's rule (geometry/mod.rs:49-53) */
            const float qc[4] = {-q[0], -q[1], -q[2], q[3]};
            float v[3] = {r2.o[0], r2.o[1], r2.o[2]};
            quat_rotate(qc, v, r2.o);
            quat_rotate(qc, ray->d, r2.d);
            for (int i = 0; i < 3; i++) r2.inv[i] = (r2.d[i] == 0.0f) ? INFINITY : 1.0f / r2.d[i];
        }
        mpo_hit h;
        if (b->inst_sph && b->inst_sph[4 * k + 3] >= 0.0f) { /* a Sphere member (primitives.rs:16-48): material 0, tex = origin */
            mpo_sphere_intersect(&b->inst_sph[4 * k], b->inst_sph[4 * k + 3], &r2, &h);
            if (cnt) cnt->rays++;
        } else
            bvh_intersect_one(b->inst_obj ? b->inst_obj[k] : b, &r2, st, &h, cnt);
        if (cnt) cnt->rays--; /* one Object::intersect call of the scene's object, however many members it holds */
        if (h.hit && h.t < best.t) {
            best = h;
            best.instance = k;
            if (q) quat_rotate(q, h.normal, best.normal); /* the normal back into the world frame */
        }
    }
    if (cnt) cnt->rays++;
    if (best.hit) mpo_ray_point_at(ray, best.t, best.point);
    *out = best;
}

int mpo_bvh_set_group(mpo_bvh *b, const mpo_bvh *const *objects, const float *spheres, const float *translations, uint32_t n) {
    if (n && objects)
        for (uint32_t k = 0; k < n; k++) {
            const int sphere = spheres && spheres[4 * k + 3] >= 0.0f;
            if (sphere) continue;
            if (!objects[k] || (objects[k] != b && objects[k]->n_inst)) return 0; /* members are plain BVHs (or the container) */
        }
    if (!mpo_bvh_set_instances(b, translations, n)) return 0;
    if (n && objects) {
        b->inst_obj = malloc((size_t)n * sizeof(*b->inst_obj));
        memcpy(b->inst_obj, objects, (size_t)n * sizeof(*b->inst_obj));
    }
    if (n && spheres) {
        b->inst_sph = malloc((size_t)n * 4 * sizeof(float));
        memcpy(b->inst_sph, spheres, (size_t)n * 4 * sizeof(float));
    }
    return 1;
}

int mpo_bvh_set_group_rotations(mpo_bvh *b, const float *quaternions) {
    if (!b) return 0;
    free(b->inst_q);
    b->inst_q = NULL;
    if (quaternions && b->n_inst) {
        b->inst_q = malloc((size_t)b->n_inst * 4 * sizeof(float));
        memcpy(b->inst_q, quaternions, (size_t)b->n_inst * 4 * sizeof(float));
    }
    return 1;
}

int mpo_bvh_set_instances(mpo_bvh *b, const float *translations, uint32_t n) {
    if (!b || (n && !translations)) return 0;
    free(b->inst_t);
    free(b->inst_obj);
    free(b->inst_sph);
    free(b->inst_q);
    b->inst_t = NULL;
    b->inst_obj = NULL;
    b->inst_sph = NULL;
    b->inst_q = NULL;
    b->n_inst = n;
    if (n) {
        b->inst_t = malloc((size_t)n * 3 * sizeof(float));
        memcpy(b->inst_t, translations, (size_t)n * 3 * sizeof(float));
    }
    return 1;
}

size_t mpo_bvh_intersect_ops(const mpo_bvh *b, const mpo_ray *ray, uint8_t *ops, uint32_t *links, size_t cap) {
    stack_cache st = {0};
    mpo_hit h;
    g_ops = ops; g_links = links; g_ops_cap = cap; g_ops_n = 0;
    bvh_intersect_one(b, ray, &st, &h, NULL);
    size_t n = g_ops_n;
    g_ops = NULL; g_links = NULL; g_ops_cap = 0; g_ops_n = 0;
    free(st.e);
    return n;
}

void mpo_bvh_intersect(const mpo_bvh *b, const mpo_ray *ray, mpo_hit *out, mpo_counters *cnt) {
    stack_cache st = {0};
    bvh_intersect_impl(b, ray, &st, out, cnt);
    free(st.e);
}

static _Thread_local uint32_t *g_trace_inst = NULL;
void mpo_trace_rays_inst(const mpo_bvh *b, const float *ox, const float *oy, const float *oz, const float *dx,
                         const float *dy, const float *dz, uint64_t n, float *t, uint32_t *prim, float *u, float *v,
                         uint32_t *inst) {
    g_trace_inst = inst;
    mpo_trace_rays(b, ox, oy, oz, dx, dy, dz, n, t, prim, u, v, NULL);
    g_trace_inst = NULL;
}

void mpo_trace_rays(const mpo_bvh *b, const float *ox, const float *oy, const float *oz, const float *dx,
                    const float *dy, const float *dz, uint64_t n, float *t, uint32_t *prim, float *u, float *v,
                    mpo_counters *cnt) {
    stack_cache st = {0};
    for (uint64_t i = 0; i < n; i++) {
        float o[3] = {ox[i], oy[i], oz[i]}, d[3] = {dx[i], dy[i], dz[i]};
        mpo_ray r;
        mpo_ray_new(o, d, &r);
        mpo_hit h;
        bvh_intersect_impl(b, &r, &st, &h, cnt);
        t[i] = h.t;
        prim[i] = h.hit ? (uint32_t)h.prim : 0xFFFFFFFFu;
        u[i] = h.u; v[i] = h.v;
        if (g_trace_inst) g_trace_inst[i] = h.hit ? h.instance : 0u;
    }
    free(st.e);
}

/* ================================================================================================= */
/* renderer/worker.rs                                                                                */
/* ================================================================================================= */

static void render_sample_impl(const mpo_bvh *b, const mpo_sampler *s, uint32_t width, uint32_t spp, uint64_t seed,
                               uint32_t x, uint32_t y, uint32_t sample, stack_cache *st, float rgba[4], mpo_counters *cnt) {
    mpo_rng rng;
    mpo_rng_seed(&rng, mpo_sample_key(seed, width, spp, x, y, sample)); /* seeded mode, SURVEY 8c */
    mpo_ray ray;
    mpo_sample_ray(s, x, y, &rng, &ray); /* :57 */
    mpo_hit h;
    bvh_intersect_impl(b, &ray, st, &h, cnt); /* :59 */
    if (h.hit) {
        float d = fabsf(ray.d[0] * h.normal[0] + ray.d[1] * h.normal[1] + ray.d[2] * h.normal[2]); /* :60 */
        rgba[0] = rgba[1] = rgba[2] = d; rgba[3] = 1.0f;
    } else {
        rgba[0] = rgba[1] = rgba[2] = rgba[3] = 0.0f; /* :63 */
    }
}

/* ---- scene/primitives.rs:15-56 : Sphere ------------------------------------------------------------------------ */
void mpo_sphere_intersect(const float center[3], float radius, const mpo_ray *ray, mpo_hit *out) {
    memset(out, 0, sizeof(*out));
    out->prim = MPO_NO_TRIANGLE;
    out->t = FLT_MAX;
    float oc[3] = {ray->o[0] - center[0], ray->o[1] - center[1], ray->o[2] - center[2]}; /* :17 */
    float b = dot3(oc, ray->d);                                                          /* :18 */
    float c = dot3(oc, oc) - radius * radius;                                            /* :19 */
    float disc = b * b - c;                                                              /* :20 */
    if (disc < 0.0f) return;                                                             /* :22-24 */
    float sq = sqrtf(disc);
    float t1 = -b - sq, t2 = -b + sq, t;                                                 /* :27-28 */
    if (t1 > 0.0f) t = t1;
    else if (t2 > 0.0f) t = t2;
    else return;                                                                         /* :29-35 */
    out->hit = 1;
    out->t = t;
    out->prim = 0;
    mpo_ray_point_at(ray, t, out->point);                                                /* :37 */
    float n[3] = {out->point[0] - center[0], out->point[1] - center[1], out->point[2] - center[2]};
    normalize3(n, out->normal);                                                          /* :38 */
}

/* render_tile (worker.rs:32-49) over Scene<Sphere> */
void mpo_render_tile_sphere(const float center[3], float radius, const mpo_sampler *s, uint32_t width, uint32_t spp,
                            uint64_t seed, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, float *rgba_f32, uint8_t *rgba_u8) {
    float inv = 1.0f / (float)spp;
    size_t tw = x1 - x0;
    for (uint32_t y = y0; y < y1; y++)
        for (uint32_t x = x0; x < x1; x++) {
            float sum[4] = {0, 0, 0, 0};
            for (uint32_t i = 0; i < spp; i++) {
                mpo_rng rng;
                mpo_rng_seed(&rng, mpo_sample_key(seed, width, spp, x, y, i));
                mpo_ray ray;
                mpo_sample_ray(s, x, y, &rng, &ray);
                mpo_hit h;
                mpo_sphere_intersect(center, radius, &ray, &h);
                if (h.hit) {
                    float d = fabsf(ray.d[0] * h.normal[0] + ray.d[1] * h.normal[1] + ray.d[2] * h.normal[2]);
                    sum[0] += d; sum[1] += d; sum[2] += d; sum[3] += 1.0f;
                } else {
                    for (int k = 0; k < 4; k++) sum[k] += 0.0f;
                }
            }
            float px[4];
            for (int k = 0; k < 4; k++) px[k] = sum[k] * inv;
            size_t o = ((size_t)(y - y0) * tw + (x - x0)) * 4;
            if (rgba_f32) memcpy(rgba_f32 + o, px, 16);
            if (rgba_u8) mpo_color_to_image(px, rgba_u8 + o);
        }
}

/* ---- build-defined path extension (NO reference counterpart: the reference has no bounce loop) --------
 * Diffuse surfaces with a material table {albedo rgb, emission rgb, checker texture} indexed by TriangleShadingData.material
 * (mod.rs:44; always 0 in the reference, building.rs:201) under a uniform sky of radiance `sky`; defaults: one grey material
 * {0.75, 0}, sky = 1.  Every operation below is per colour channel c (r, g, b); a table of grey, untextured materials makes the
 * three channels the same number, which is the round-1/2 definition bit for bit.
 * Paths of at most max_depth segments:
 *   L = 0, throughput = 1; for depth = 1..max_depth: trace; miss -> L[c] = L[c] + throughput[c] * sky, stop;
 *   hit -> m = material of the triangle; L[c] = L[c] + throughput[c] * emission[m][c]; n = shading normal turned against the ray;
 *   a = albedo[m], or -- texture == MPO_TEXTURE_CHECKER, reading HitRecord.texture_coords (geometry/mod.rs:78-79, interpolated
 *   at ray_bvh_intersection.rs:80-83; a Sphere's are the origin, primitives.rs:45) -- albedo2[m] on odd cells:
 *   cell = floor(tex.x * tex_scale) + floor(tex.y * tex_scale), odd iff (cell * 0.5 - floor(cell * 0.5)) != 0 (so NaN counts as odd);
 *   throughput[c] *= a[c]; at depth == max_depth stop;
 *   next direction = cosine-weighted about n: (x, y) = UnitDisc rejection sample from the SAME Xoshiro stream,
 *   z = sqrt(1 - (x*x + y*y)); orthonormal basis of Duff et al. 2017 (branchless, copysign); origin = point + n * 1e-4.
 * Only + - * / sqrt floor and comparisons, so the GPU reproduces it bit for bit.  rgba = (L.r, L.g, L.b, primary hit ? 1 : 0).
 * With the defaults this is exactly the round-1 definition (L = 0 + throughput * 1 at the miss, + 0 at every hit). */
static _Thread_local uint32_t g_max_depth = 0;      /* 0 = reference semantics (worker.rs:51-66) */
static _Thread_local uint64_t g_segments = 0;

static void render_sample_paths_impl(const mpo_bvh *b, const mpo_sampler *s, uint32_t width, uint32_t spp, uint64_t seed,
                                     uint32_t x, uint32_t y, uint32_t sample, uint32_t max_depth, stack_cache *st, float rgba[4],
                                     mpo_counters *cnt, uint64_t *segments) {
    mpo_rng rng;
    mpo_rng_seed(&rng, mpo_sample_key(seed, width, spp, x, y, sample));
    mpo_ray ray;
    mpo_sample_ray(s, x, y, &rng, &ray);
    float L[3] = {0.0f, 0.0f, 0.0f}, thr[3] = {1.0f, 1.0f, 1.0f}, alpha = 0.0f;
    for (uint32_t depth = 1; depth <= max_depth; depth++) {
        mpo_hit h;
        bvh_intersect_impl(b, &ray, st, &h, cnt);
        if (segments) (*segments)++;
        if (!h.hit) { for (int c = 0; c < 3; c++) L[c] = L[c] + thr[c] * b->sky; break; }
        if (depth == 1) alpha = 1.0f;
        const mpo_material *m = &b->mats[h.material];
        for (int c = 0; c < 3; c++) L[c] = L[c] + thr[c] * m->emission[c];
        float n[3] = {h.normal[0], h.normal[1], h.normal[2]};
        float dn = ray.d[0] * n[0] + ray.d[1] * n[1] + ray.d[2] * n[2];
        if (dn > 0.0f) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
        const float *alb = m->albedo;
        if (m->texture == MPO_TEXTURE_CHECKER) {
            const float cell = floorf(h.tex[0] * m->tex_scale) + floorf(h.tex[1] * m->tex_scale);
            const float half = cell * 0.5f;
            if (half - floorf(half) != 0.0f) alb = m->albedo2;
        }
        for (int c = 0; c < 3; c++) thr[c] = thr[c] * alb[c];
        if (depth == max_depth) break;
        float d2[2];
        mpo_rng_unit_disc(&rng, d2);
        float z = sqrtf(1.0f - (d2[0] * d2[0] + d2[1] * d2[1]));
        float sign = copysignf(1.0f, n[2]);
        float a = -1.0f / (sign + n[2]);
        float bb = n[0] * n[1] * a;
        float t[3] = {1.0f + sign * n[0] * n[0] * a, sign * bb, -sign * n[0]};
        float bt[3] = {bb, sign + n[1] * n[1] * a, -n[1]};
        float dir[3], org[3];
        for (int k = 0; k < 3; k++) {
            dir[k] = t[k] * d2[0] + bt[k] * d2[1] + n[k] * z;
            org[k] = h.point[k] + n[k] * MPO_PATH_EPS;
        }
        mpo_ray_new(org, dir, &ray);
    }
    rgba[0] = L[0]; rgba[1] = L[1]; rgba[2] = L[2];
    rgba[3] = alpha;
}

void mpo_render_sample_paths(const mpo_bvh *b, const mpo_sampler *s, uint32_t width, uint32_t spp, uint64_t seed, uint32_t x,
                             uint32_t y, uint32_t sample, uint32_t max_depth, float rgba[4], uint64_t *segments) {
    stack_cache st = {0};
    render_sample_paths_impl(b, s, width, spp, seed, x, y, sample, max_depth, &st, rgba, NULL, segments);
    free(st.e);
}

void mpo_render_sample(const mpo_bvh *b, const mpo_sampler *s, uint32_t width, uint32_t spp, uint64_t seed,
                       uint32_t x, uint32_t y, uint32_t sample, float rgba[4], mpo_counters *cnt) {
    stack_cache st = {0};
    render_sample_impl(b, s, width, spp, seed, x, y, sample, &st, rgba, cnt);
    free(st.e);
}

/* worker.rs:69-76 : (c*255).round() half away from zero, clamp, `as u8` (NaN -> 0) */
void mpo_render_tile_paths(const mpo_bvh *b, const mpo_sampler *s, uint32_t width, uint32_t height, uint32_t spp,
                           uint64_t seed, uint32_t max_depth, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                           float *rgba_f32, uint8_t *rgba_u8, uint64_t *segments) {
    g_max_depth = max_depth;
    g_segments = 0;
    mpo_render_tile(b, s, width, height, spp, seed, x0, y0, x1, y1, rgba_f32, rgba_u8, NULL);
    if (segments) *segments = g_segments;
    g_max_depth = 0;
}

void mpo_color_to_image(const float rgba[4], uint8_t out[4]) {
    for (int k = 0; k < 4; k++) {
        float x = roundf(rgba[k] * 255.0f);
        if (x != x) { out[k] = 0; continue; }
        x = x < 0.0f ? 0.0f : (x > 255.0f ? 255.0f : x);
        out[k] = (uint8_t)x;
    }
}

/* BUILD-DEFINED accumulation rule for very long sample chains (BASELINE configs[4]: 65 536 spp; SURVEY 7 "hard parts"): the
 * reference sums every sample of a pixel into one f32 (worker.rs:40-43), whose rounding error grows with the chain.  With the
 * chunked rule the samples are summed in f32, in index order, in chunks of MPO_SUM_CHUNK (chunk c = samples
 * [256c, 256c+256)), each chunk sum is added to an f64 total, and pixel = (f32)(total * (1.0 / (f64)spp)).  Off by default
 * (reference semantics); process-wide switch, set before rendering. */
#define MPO_SUM_CHUNK 256u
static int g_chunked_sum = 0;
void mpo_set_chunked_sum(int on) { g_chunked_sum = on; }

static void render_tile_impl(const mpo_bvh *b, const mpo_sampler *s, uint32_t width, uint32_t spp, uint64_t seed,
                             uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, float *f32_out, size_t f32_row_stride,
                             uint8_t *u8_out, size_t u8_row_stride, stack_cache *st, mpo_counters *cnt) {
    float inv = 1.0f / (float)spp; /* :44 */
    const double inv64 = 1.0 / (double)spp;
    for (uint32_t y = y0; y < y1; y++)
        for (uint32_t x = x0; x < x1; x++) { /* internal_points, x fastest :39 */
            float sum[4] = {0, 0, 0, 0};
            double total[4] = {0, 0, 0, 0};
            for (uint32_t i = 0; i < spp; i++) { /* :41-43 */
                float c[4];
                if (g_max_depth == 0) { render_sample_impl(b, s, width, spp, seed, x, y, i, st, c, cnt); g_segments++; }
                else render_sample_paths_impl(b, s, width, spp, seed, x, y, i, g_max_depth, st, c, cnt, &g_segments);
                for (int k = 0; k < 4; k++) sum[k] += c[k];
                if (g_chunked_sum && ((i + 1) % MPO_SUM_CHUNK == 0 || i + 1 == spp))
                    for (int k = 0; k < 4; k++) { total[k] += (double)sum[k]; sum[k] = 0.0f; }
            }
            float px[4];
            for (int k = 0; k < 4; k++) px[k] = g_chunked_sum ? (float)(total[k] * inv64) : sum[k] * inv;
            size_t lx = x - x0, ly = y - y0;
            if (f32_out) memcpy(f32_out + ly * f32_row_stride + lx * 4, px, 16);
            if (u8_out) mpo_color_to_image(px, u8_out + ly * u8_row_stride + lx * 4);
        }
}

void mpo_render_tile(const mpo_bvh *b, const mpo_sampler *s, uint32_t width, uint32_t height, uint32_t spp,
                     uint64_t seed, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, float *rgba_f32,
                     uint8_t *rgba_u8, mpo_counters *cnt) {
    (void)height;
    stack_cache st = {0};
    size_t tw = x1 - x0;
    render_tile_impl(b, s, width, spp, seed, x0, y0, x1, y1, rgba_f32, tw * 4, rgba_u8, tw * 4, &st, cnt);
    free(st.e);
}

/* ---- renderer/machinery.rs:20-123 : worker threads + atomic tile queue ------------------------------- */

typedef struct {
    const mpo_bvh *b; const mpo_sampler *s;
    uint32_t width, height, spp, tile; uint64_t seed;
    uint32_t *tiles; size_t ntiles, limit, stride;
    atomic_size_t next;
    float *f32; uint8_t *u8;
    pthread_mutex_t mu; mpo_counters total; int want_cnt;
    atomic_ullong rays;
    uint32_t max_depth; atomic_ullong segments;
} mt_state;

static void *mt_worker(void *arg) {
    mt_state *S = arg;
    stack_cache st = {0};
    mpo_counters cnt = {0};
    unsigned long long rays = 0;
    g_max_depth = S->max_depth;
    g_segments = 0;
    for (;;) {
        size_t id = atomic_fetch_add_explicit(&S->next, 1, memory_order_acq_rel); /* get_next_tile :205-208 */
        size_t ti = id * S->stride;
        if (id >= S->limit || ti >= S->ntiles) break;
        const uint32_t *t = S->tiles + 4 * ti;
        size_t row = (size_t)S->width * 4;
        render_tile_impl(S->b, S->s, S->width, S->spp, S->seed, t[0], t[1], t[2], t[3],
                         S->f32 ? S->f32 + (size_t)t[1] * row + (size_t)t[0] * 4 : NULL, row,
                         S->u8 ? S->u8 + (size_t)t[1] * row + (size_t)t[0] * 4 : NULL, row, &st,
                         S->want_cnt ? &cnt : NULL);
        rays += (unsigned long long)(t[2] - t[0]) * (t[3] - t[1]) * S->spp;
    }
    free(st.e);
    atomic_fetch_add(&S->rays, rays);
    atomic_fetch_add(&S->segments, g_segments);
    g_max_depth = 0;
    if (S->want_cnt) {
        pthread_mutex_lock(&S->mu);
        S->total.rays += cnt.rays; S->total.inner_visited += cnt.inner_visited;
        S->total.packets_tested += cnt.packets_tested; S->total.stack_pops += cnt.stack_pops;
        if (cnt.max_stack > S->total.max_stack) S->total.max_stack = cnt.max_stack;
        pthread_mutex_unlock(&S->mu);
    }
    return NULL;
}

static double render_image_mt_impl(const mpo_bvh *b, const mpo_sampler *s, uint32_t width, uint32_t height, uint32_t spp,
                                   uint64_t seed, uint32_t tile, int nthreads, size_t max_tiles, size_t tile_stride,
                                   float *rgba_f32, uint8_t *rgba_u8, uint64_t *rays_out, mpo_counters *cnt, uint32_t max_depth,
                                   uint64_t *segments_out) {
    mt_state S;
    memset(&S, 0, sizeof(S));
    S.max_depth = max_depth;
    atomic_init(&S.segments, 0);
    S.b = b; S.s = s; S.width = width; S.height = height; S.spp = spp; S.tile = tile; S.seed = seed;
    S.ntiles = mpo_tile_ordering(0, 0, width, height, tile, 0, NULL, 0);
    S.tiles = malloc(S.ntiles * 16 + 16);
    mpo_tile_ordering(0, 0, width, height, tile, 0, S.tiles, S.ntiles);
    S.stride = tile_stride ? tile_stride : 1;
    S.limit = max_tiles ? max_tiles : S.ntiles;
    atomic_init(&S.next, 0);
    atomic_init(&S.rays, 0);
    S.f32 = rgba_f32; S.u8 = rgba_u8;
    S.want_cnt = cnt != NULL;
    pthread_mutex_init(&S.mu, NULL);
    if (nthreads < 1) nthreads = 1;
    pthread_t *th = malloc((size_t)nthreads * sizeof(pthread_t));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = 0; i < nthreads; i++) pthread_create(&th[i], NULL, mt_worker, &S);
    for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(th); free(S.tiles);
    pthread_mutex_destroy(&S.mu);
    if (rays_out) *rays_out = atomic_load(&S.rays);
    if (segments_out) *segments_out = atomic_load(&S.segments);
    if (cnt) *cnt = S.total;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

double mpo_render_image_mt(const mpo_bvh *b, const mpo_sampler *s, uint32_t width, uint32_t height, uint32_t spp,
                           uint64_t seed, uint32_t tile, int nthreads, size_t max_tiles, size_t tile_stride,
                           float *rgba_f32, uint8_t *rgba_u8, uint64_t *rays_out, mpo_counters *cnt) {
    return render_image_mt_impl(b, s, width, height, spp, seed, tile, nthreads, max_tiles, tile_stride, rgba_f32, rgba_u8, rays_out,
                                cnt, 0, NULL);
}

/* path extension: *segments_out = Object::intersect calls (ray segments) */
double mpo_render_image_paths_mt(const mpo_bvh *b, const mpo_sampler *s, uint32_t width, uint32_t height, uint32_t spp,
                                 uint64_t seed, uint32_t max_depth, uint32_t tile, int nthreads, size_t max_tiles,
                                 size_t tile_stride, float *rgba_f32, uint8_t *rgba_u8, uint64_t *segments_out) {
    return render_image_mt_impl(b, s, width, height, spp, seed, tile, nthreads, max_tiles, tile_stride, rgba_f32, rgba_u8, NULL, NULL,
                                max_depth, segments_out);
}

/* same with traversal counters (diagnostics) */
double mpo_render_image_paths_mt_cnt(const mpo_bvh *b, const mpo_sampler *s, uint32_t width, uint32_t height, uint32_t spp,
                                     uint64_t seed, uint32_t max_depth, uint32_t tile, int nthreads, float *rgba_f32,
                                     uint64_t *segments_out, mpo_counters *cnt) {
    return render_image_mt_impl(b, s, width, height, spp, seed, tile, nthreads, 0, 1, rgba_f32, NULL, NULL, cnt, max_depth, segments_out);
}

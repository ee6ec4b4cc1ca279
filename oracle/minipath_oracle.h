/*
 * minipath_oracle.h -- CPU restatement of bluecube/minipath's per-pixel sampling hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product (minipath_amd/) never links, imports or calls
 * anything in oracle/.
 *
 * It restates, in plain C11 with scalar-per-lane loops that mirror the reference's 8-wide AVX2 SIMD, the
 * functions of SURVEY.md section 8(a).  Every function cites the reference file:line it follows (paths are
 * relative to the reference checkout).  FMAs appear only where the reference writes mul_add / mul_sub;
 * the file must be compiled with -ffp-contract=off and without -ffast-math.
 *
 * PARITY PINNING.  The Rust reference cannot be compiled in this pipeline (no cargo/rustc, crates not
 * vendored), so the restatement is pinned by the reference's own known-answer unit tests
 * (tests/test_oracle_known_answers.py restates them: aabb.rs:374-471, triangle_bvh/mod.rs:189-237,
 * compressed_geometry.rs:190-200, util/mod.rs:40-57, util/simba.rs:85-112, screen_block.rs:215-254,
 * camera.rs:201-247).  The reference holds NO fixture for Triangle::intersect, TriangleBvh::intersect, the
 * builder, the RNG stream or a rendered image: for those the oracle is "parity unpinned" (see DESIGN.md).
 * Items marked (R) below are recalled from third-party crates whose sources are not in the reference tree
 * (rand 0.9.3, rand_distr 0.5.1, nalgebra 0.33.2, wide 0.7.32, obj 0.10.2).
 */
#ifndef MINIPATH_ORACLE_H
#define MINIPATH_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- triangle_bvh/mod.rs:14-17, 57-114 ---------------------------------------------------------- */
#define MPO_INNER_NODE_CHILDREN 8
#define MPO_LEAF_PACKET_SIZE 8
#define MPO_LINK_COUNT_BITS 3u
#define MPO_LINK_COUNT_MASK 7u
#define MPO_LINK_NULL 0xFFFFFFF8u
#define MPO_LINK_MAX_INDEX 536870910u /* (u32::MAX >> 3) - 1 */
#define MPO_LINK_MAX_COUNT 7u
#define MPO_LEAF_MAX_TRIANGLES 56
#define MPO_NO_TRIANGLE UINT64_MAX /* TriangleIdx::default(), mod.rs:143-147 */

/* RNG: rand 0.9.3 SmallRng == Xoshiro256++ on 64-bit targets (R). */
typedef struct { uint64_t s[4]; } mpo_rng;

/* geometry/mod.rs:33-43 */
typedef struct { float o[3]; float d[3]; float inv[3]; } mpo_ray;

/* camera.rs:26-39 -- the 15 floats of CameraSampler, in declaration order. */
typedef struct {
    float center[3];
    float up[3];
    float right[3];
    float film_origin_offset[3];
    float pixel_scale;
    float lens_radius;
    float lens_weight;
} mpo_sampler;

/* camera.rs:9-18.  camera_to_world is an Isometry3: unit quaternion (i,j,k,w) + translation. */
typedef struct {
    float q[4]; /* i, j, k, w */
    float t[3];
    float focus_distance;
    int sensor_is_width; /* SensorSize::Width / ::Height, camera.rs:20-24 */
    float sensor_size;
    float focal_length;
    float f_number;
} mpo_camera;

/* triangle_bvh/mod.rs:33-38 : RelativeBox8 {min{x,y,z}, max{x,y,z}} (6 x u16x8) + 8 links = 128 B */
typedef struct {
    uint16_t bmin[3][8];
    uint16_t bmax[3][8];
    uint32_t link[8];
} mpo_inner_node;

/* compressed_geometry.rs:139 : Triangle<RelativePoint8> = 3 vertices x 3 coords x u16x8 = 144 B */
typedef struct { uint16_t v[3][3][8]; } mpo_tri_packet;

/* triangle_bvh/mod.rs:41-45 (usize narrowed to u32; material is always 0, building.rs:201) */
typedef struct { uint32_t vi[3]; uint32_t flat; } mpo_tri_shading;

/* ray_bvh_intersection.rs:165-171 + geometry/mod.rs:71-80 */
typedef struct {
    int hit;            /* 0 = None */
    uint64_t prim;      /* LeafHitRecord.triangle_index = packet*8+lane */
    float t, u, v;      /* LeafHitRecord.t, .uv */
    float gn[3];        /* LeafHitRecord.geometric_normal (unnormalised, unfused cross) */
    float point[3];     /* HitRecord.point */
    float normal[3];    /* HitRecord.normal (unit) */
    float tex[3];       /* HitRecord.texture_coords */
    uint64_t material;
    uint32_t instance;  /* build-defined instanced Object: which instance was hit (0 otherwise) */
} mpo_hit;

typedef struct {
    uint64_t rays;
    uint64_t inner_visited;   /* InnerNode::intersect calls */
    uint64_t packets_tested;  /* RelativeTriangle8 decompress+intersect calls */
    uint64_t stack_pops;
    uint64_t max_stack;
} mpo_counters;

typedef struct mpo_bvh mpo_bvh;

/* build-defined path extension: grey diffuse material {albedo, emission}, indexed by TriangleShadingData.material */
/* BUILD-DEFINED material of the path extension (no reference counterpart; see render_sample_paths_impl): diffuse reflectance and
 * emitted radiance per colour channel; texture = MPO_TEXTURE_CHECKER swaps in albedo2 on the odd cells of a checkerboard over
 * HitRecord.texture_coords (geometry/mod.rs:78-79), tex_scale cells per unit of texture coordinate. */
#define MPO_TEXTURE_NONE 0u
#define MPO_TEXTURE_CHECKER 1u
typedef struct {
    float albedo[3];
    float emission[3];
    float albedo2[3];
    uint32_t texture;
    float tex_scale;
    uint32_t pad;
} mpo_material;

/* ---- RNG (R): rand 0.9.3 / rand_distr 0.5.1 ------------------------------------------------------- */
void mpo_rng_seed(mpo_rng *r, uint64_t state);          /* Xoshiro256PlusPlus::seed_from_u64 (SplitMix64) */
uint64_t mpo_rng_next_u64(mpo_rng *r);
uint32_t mpo_rng_next_u32(mpo_rng *r);                  /* upper 32 bits of next_u64 */
float mpo_rng_range_pm_half(mpo_rng *r);                /* rng.random_range(-0.5..=0.5), camera.rs:178-179 */
void mpo_rng_unit_disc(mpo_rng *r, float out[2]);       /* rand_distr::UnitDisc, camera.rs:184 */
uint64_t mpo_seed_mix(uint64_t seed);                  /* first SplitMix64 output for state `seed` */
uint64_t mpo_sample_key(uint64_t seed, uint32_t width, uint32_t spp, uint32_t x, uint32_t y, uint32_t s);

/* ---- geometry ------------------------------------------------------------------------------------- */
void mpo_ray_new(const float o[3], const float d[3], mpo_ray *out);   /* geometry/mod.rs:45-54 */
void mpo_ray_point_at(const mpo_ray *r, float t, float out[3]);        /* geometry/mod.rs:56-58 */
/* aabb.rs:254-284, 8 boxes vs one ray */
void mpo_aabb8_intersect(const float bmin[3][8], const float bmax[3][8], const mpo_ray *ray, float max_t,
                         float t1[8], float t2[8]);
/* triangle.rs:183-217, returns 8-bit mask (u>=0 & v>=0 & u+v<=1) */
unsigned mpo_tri8_intersect(const float v0[3][8], const float v1[3][8], const float v2[3][8], const mpo_ray *ray,
                            float t[8], float u[8], float v[8]);
/* compressed_geometry.rs:25-51 ; rounding: 0 = round (ties-even), 1 = floor, 2 = ceil */
uint16_t mpo_unit_interval_compress(float v, int rounding, int mask);
float mpo_unit_interval_decompress(uint16_t q);
/* util/mod.rs:6-31 : writes ascending set-bit indices, returns count */
int mpo_bit_iter(uint64_t bits, int out[64]);
/* triangle_bvh/mod.rs:70-113 */
uint32_t mpo_link_new_leaf(uint32_t index, uint32_t count, int *ok);
uint32_t mpo_link_new_inner(uint32_t index, int *ok);
/* returns 0 = Null, 1 = Inner, 2 = Leaf */
int mpo_link_decode(uint32_t link, uint32_t *index, uint32_t *count);

/* ---- camera.rs ------------------------------------------------------------------------------------ */
void mpo_camera_default(mpo_camera *c);                                                    /* :42-52 */
void mpo_camera_look_at(mpo_camera *c, const float eye[3], const float at[3], const float up[3]); /* :93-101 */
void mpo_camera_look_direction(mpo_camera *c, const float eye[3], const float fwd[3], const float up[3]); /* :104-116 */
void mpo_camera_translate(mpo_camera *c, const float t[3]);   /* Camera::transformed(Translation3), :119-121 */
void mpo_camera_basis(const mpo_camera *c, float center[3], float fwd[3], float up[3], float right[3]); /* :148-171 */
void mpo_camera_build_sampler(const mpo_camera *c, uint32_t w, uint32_t h, mpo_sampler *out); /* :123-146 */
void mpo_sample_ray(const mpo_sampler *s, uint32_t x, uint32_t y, mpo_rng *rng, mpo_ray *out); /* :176-191 */

/* ---- screen_block.rs ------------------------------------------------------------------------------ */
/* divide_range :144-160 ; writes (start,end) pairs, returns n */
size_t mpo_divide_range(uint32_t start, uint32_t end, uint32_t tile, uint32_t *out_pairs, size_t cap);
/* tile_ordering :46-81.  out = n x {minx,miny,maxx,maxy}.  shuffle_seed==0: plain row-major grid order
 * (deterministic bench order); otherwise centre-out order with seeded Exp noise (the reference draws the
 * noise from a thread-local OS-seeded RNG, so no particular order is reproducible). */
size_t mpo_tile_ordering(uint32_t minx, uint32_t miny, uint32_t maxx, uint32_t maxy, uint32_t tile,
                         uint64_t shuffle_seed, uint32_t *out, size_t cap);
/* internal_points :28-39,104-128 ; writes (x,y) pairs in C order, returns n */
size_t mpo_internal_points(uint32_t minx, uint32_t miny, uint32_t maxx, uint32_t maxy, uint32_t *out, size_t cap);

/* ---- building.rs ---------------------------------------------------------------------------------- */
mpo_bvh *mpo_bvh_from_obj(const char *path, char *err, size_t errcap);                     /* :28-81 */
mpo_bvh *mpo_bvh_build(const float *pos, const float *nrm, const float *tex, uint32_t nv,
                       const uint32_t *tri_idx, uint32_t nt, char *err, size_t errcap);     /* :83-207 */
/* same with a material id per input triangle (nullable = all 0, the reference's `material: 0`, building.rs:201) */
mpo_bvh *mpo_bvh_build_mat(const float *pos, const float *nrm, const float *tex, uint32_t nv, const uint32_t *tri_idx,
                           const uint32_t *tri_mat, uint32_t nt, char *err, size_t errcap);
/* a TriangleBvh given as its reference-layout arrays (triangle_bvh/mod.rs:20-53); arrays are copied, links validated */
mpo_bvh *mpo_bvh_from_arrays(const mpo_inner_node *inner, uint32_t n_inner, const mpo_tri_packet *packets, uint32_t n_packets,
                             const mpo_tri_shading *shading, const uint32_t *material, const float *vnormal, const float *vtex,
                             uint32_t nv, uint32_t root, const float bmin[3], const float bmax[3], char *err, size_t errcap);
/* material table + sky radiance of the build-defined path extension; returns 0 if a triangle's id is >= n */
int mpo_bvh_set_materials(mpo_bvh *b, const mpo_material *table, uint32_t n, float sky);
const uint32_t *mpo_bvh_tri_material(const mpo_bvh *b);       /* packet_count*8 entries */
uint32_t mpo_bvh_material_count(const mpo_bvh *b);            /* max id + 1 */
/* BUILD-DEFINED Object: n translated instances of this BVH (n*3 floats; n = 0 restores the plain TriangleBvh).  Every
 * intersect / render entry point then treats the list as the scene's object (see scene semantics in the .c file). */
int mpo_bvh_set_instances(mpo_bvh *b, const float *translations, uint32_t n);
/* ... n members {object k, translation k}: objects[k] = a plain BVH (may be b itself), or -- where spheres (4 floats per member:
 * center, radius; NULL = no sphere members) has radius >= 0 -- a Sphere (scene/primitives.rs:10-56; objects[k] is then ignored).
 * b is the container: its material table and sky apply, its own triangles are only reachable through members that name it.  The
 * members must outlive b's use. */
int mpo_bvh_set_group(mpo_bvh *b, const mpo_bvh *const *objects, const float *spheres, const float *translations, uint32_t n);
/* ... and a rotation per member of the current group (n_members * 4 floats, unit quaternions (i, j, k, w) as nalgebra stores
 * them; NULL = none): world = q * local + translation.  The ray enters a member's frame as q^-1 * (origin - translation),
 * q^-1 * direction (not re-normalised: t stays the world ray's), the hit normal leaves it as q * normal. */
int mpo_bvh_set_group_rotations(mpo_bvh *b, const float *quaternions);
void mpo_bvh_free(mpo_bvh *b);
uint32_t mpo_bvh_root(const mpo_bvh *b);
void mpo_bvh_bbox(const mpo_bvh *b, float bmin[3], float bmax[3]);
uint32_t mpo_bvh_inner_count(const mpo_bvh *b);
uint32_t mpo_bvh_packet_count(const mpo_bvh *b);
uint32_t mpo_bvh_vertex_count(const mpo_bvh *b);
uint32_t mpo_bvh_depth(const mpo_bvh *b); /* max number of inner nodes on a root-to-leaf path */
const mpo_inner_node *mpo_bvh_inner_nodes(const mpo_bvh *b);
const mpo_tri_packet *mpo_bvh_packets(const mpo_bvh *b);
const mpo_tri_shading *mpo_bvh_tri_shading(const mpo_bvh *b); /* packet_count*8 entries */
const float *mpo_bvh_vertex_normals(const mpo_bvh *b);        /* nv*3 */
const float *mpo_bvh_vertex_tex(const mpo_bvh *b);            /* nv*3 */

/* ---- ray_bvh_intersection.rs ---------------------------------------------------------------------- */
void mpo_bvh_intersect(const mpo_bvh *b, const mpo_ray *ray, mpo_hit *out, mpo_counters *cnt);  /* :26-96 */
/* diagnostics only: per-pop operation trace of one ray (0 culled, 1 inner, 8+k leaf of k packets) */
size_t mpo_bvh_intersect_ops(const mpo_bvh *b, const mpo_ray *ray, uint8_t *ops, uint32_t *links, size_t cap);
/* batched: rays as SoA ox,oy,oz,dx,dy,dz (directions need not be normalised: Ray::new is applied) */
void mpo_trace_rays(const mpo_bvh *b, const float *ox, const float *oy, const float *oz, const float *dx,
                    const float *dy, const float *dz, uint64_t n, float *t, uint32_t *prim, float *u, float *v,
                    mpo_counters *cnt);

void mpo_trace_rays_inst(const mpo_bvh *b, const float *ox, const float *oy, const float *oz, const float *dx,
                         const float *dy, const float *dz, uint64_t n, float *t, uint32_t *prim, float *u, float *v,
                         uint32_t *inst);

/* ---- renderer/worker.rs --------------------------------------------------------------------------- */
/* render_sample :51-66 with the build-defined seeded RNG: returns rgba */
void mpo_render_sample(const mpo_bvh *b, const mpo_sampler *s, uint32_t width, uint32_t spp, uint64_t seed,
                       uint32_t x, uint32_t y, uint32_t sample, float rgba[4], mpo_counters *cnt);
/* render_tile :32-49.  rgba_f32 = tile_w*tile_h*4 pre-quantisation means (x fastest); rgba_u8 nullable */
void mpo_render_tile(const mpo_bvh *b, const mpo_sampler *s, uint32_t width, uint32_t height, uint32_t spp,
                     uint64_t seed, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, float *rgba_f32,
                     uint8_t *rgba_u8, mpo_counters *cnt);
void mpo_color_to_image(const float rgba[4], uint8_t out[4]); /* worker.rs:69-76 */
/* BUILD-DEFINED (no reference counterpart): chunked accumulation for very long sample chains -- f32 sums over chunks of 256
 * consecutive samples, chunk sums added in f64, pixel = (f32)(total * (1.0 / (f64)spp)).  Process-wide, off by default. */
void mpo_set_chunked_sum(int on);
/* renderer/machinery.rs:20-123 -- threads pulling tiles from an atomic queue.  Image-major f32/u8 output.
 * max_tiles = 0 renders all tiles, otherwise only the first max_tiles of the row-major order (bounded
 * baseline sample).  Returns wall seconds of the tile loop; *rays_out = samples rendered. */
double mpo_render_image_mt(const mpo_bvh *b, const mpo_sampler *s, uint32_t width, uint32_t height, uint32_t spp,
                           uint64_t seed, uint32_t tile, int nthreads, size_t max_tiles, size_t tile_stride,
                           float *rgba_f32, uint8_t *rgba_u8, uint64_t *rays_out, mpo_counters *cnt);

/* ---- scene/primitives.rs : Sphere (Object::intersect :16-48) ----------------------------------------------------- */
void mpo_sphere_intersect(const float center[3], float radius, const mpo_ray *ray, mpo_hit *out);
void mpo_render_tile_sphere(const float center[3], float radius, const mpo_sampler *s, uint32_t width, uint32_t spp,
                            uint64_t seed, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, float *rgba_f32, uint8_t *rgba_u8);

/* ---- build-defined path extension (no reference counterpart; see the .c file) ------------------------------ */
void mpo_render_sample_paths(const mpo_bvh *b, const mpo_sampler *s, uint32_t width, uint32_t spp, uint64_t seed, uint32_t x,
                             uint32_t y, uint32_t sample, uint32_t max_depth, float rgba[4], uint64_t *segments);
void mpo_render_tile_paths(const mpo_bvh *b, const mpo_sampler *s, uint32_t width, uint32_t height, uint32_t spp,
                           uint64_t seed, uint32_t max_depth, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                           float *rgba_f32, uint8_t *rgba_u8, uint64_t *segments);
double mpo_render_image_paths_mt(const mpo_bvh *b, const mpo_sampler *s, uint32_t width, uint32_t height, uint32_t spp,
                                 uint64_t seed, uint32_t max_depth, uint32_t tile, int nthreads, size_t max_tiles,
                                 size_t tile_stride, float *rgba_f32, uint8_t *rgba_u8, uint64_t *segments_out);

#ifdef __cplusplus
}
#endif
#endif

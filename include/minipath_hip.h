/*
 * minipath_hip.h -- C ABI of the MI355X (gfx950) implementation of minipath's per-pixel sampling hot path.
 *
 * This is the drop-in boundary.  The reference has no FFI; its seam is the Rust API of
 * src/renderer (+ src/screen_block.rs, src/camera.rs, src/scene/triangle_bvh).  Every entry point below names
 * the reference interface it replaces (file:line in the reference checkout).  A Rust `src/renderer`-shaped shim
 * binds exactly these symbols (INTEGRATION.md shows it).
 *
 * Conventions
 *   - plain C types, pointers and sizes only; no C++/torch types cross the boundary;
 *   - every function returns an int status (MP_OK == 0); no exception or abort crosses the ABI (every entry point's body
 *     runs inside a catch-all: std::bad_alloc -> MP_ERR_NOMEM, anything else -> MP_ERR_INVALID); the message of the last
 *     failure on the calling thread is mp_last_error();
 *   - "d_" pointers are device (HBM) pointers on the context's GPU, everything else is host memory;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *   - one context drives ONE GPU.  Multi-GPU, two ways: (a) one process and one context PER GPU, tiles sharded across
 *     ranks by the caller, framebuffer gathered with RCCL by the caller (bench.py / minipath_amd.distributed);
 *     (b) ONE process driving several contexts behind this ABI: mp_render_begin_multi (render()'s worker pool over
 *     devices) and mp_render_frame_multi (device-resident frame, shards gathered by peer copies over xGMI).
 *
 * Seeded mode.  The reference seeds every worker's RNG from the OS (worker.rs:25), so it has no reproducible
 * sample stream.  This library defines one: sample s of pixel (x,y) uses
 *   Xoshiro256++::seed_from_u64(mix(seed) + ((y*W + x)*spp + s)),   mix(seed) = first SplitMix64 output for state `seed`
 * followed by the draw order of CameraSampler::sample_ray (camera.rs:176-191).  The seed is mixed before the sample index is
 * added, so frames rendered with consecutive seeds share no sample streams.
 */
#ifndef MINIPATH_HIP_H
#define MINIPATH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MP_OK 0
#define MP_ERR_INVALID 1      /* bad argument */
#define MP_ERR_IO 2           /* ObjOpenError::ReadError / ParseError, building.rs:209-216 */
#define MP_ERR_BUILD 3        /* a condition on which the reference builder panics (building.rs:178,275) */
#define MP_ERR_HIP 4          /* HIP runtime error */
#define MP_ERR_UNSUPPORTED 5
#define MP_ERR_ABORTED 6
#define MP_ERR_NOMEM 7        /* host allocation failed (std::bad_alloc caught at the boundary) */

#define MP_NO_PRIM 0xFFFFFFFFu /* TriangleIdx::default() (usize::MAX) narrowed to u32, triangle_bvh/mod.rs:143-147 */
#define MP_LINK_NULL 0xFFFFFFF8u /* CompressedNodeLink::NULL, triangle_bvh/mod.rs:63 */

typedef struct mp_ctx mp_ctx;
typedef struct mp_scene mp_scene;
typedef struct mp_render mp_render;

/* geometry/mod.rs:15 ScreenBlock = AABB<Point2<u32>> : [min, max) */
typedef struct { uint32_t min_x, min_y, max_x, max_y; } mp_block;

/* camera.rs:9-18 Camera.  camera_to_world (Isometry3<f32>) = unit quaternion (i,j,k,w) + translation. */
typedef struct {
    float q[4];
    float t[3];
    float focus_distance;
    int32_t sensor_is_width; /* camera.rs:20-24 SensorSize::Width(1) / ::Height(0) */
    float sensor_size;
    float focal_length;
    float f_number;
} mp_camera;

/* camera.rs:26-39 CameraSampler: the 15 floats, in declaration order. */
typedef struct {
    float center[3];
    float up[3];
    float right[3];
    float film_origin_offset[3];
    float pixel_scale;
    float lens_radius;
    float lens_weight;
} mp_camera_sampler;

/* renderer/mod.rs:7-13 RenderSettings, plus the build-defined seed (see "Seeded mode"). */
typedef struct {
    uint32_t tile_size;    /* NonZeroU32 */
    uint32_t sample_count; /* NonZeroU32 */
    uint32_t width, height;
    uint64_t seed;
    uint32_t flags;        /* MP_FLAG_* */
    uint32_t max_depth;    /* only with MP_FLAG_PATHS: ray segments per path (>= 1); otherwise ignored (0) */
    uint32_t pass_begin;   /* only with MP_FLAG_ACCUMULATE: the launch renders samples [pass_begin, pass_begin + pass_count) */
    uint32_t pass_count;   /*   of sample_count; pass_count 0 = through the last sample.  Otherwise both 0. */
} mp_settings;

#define MP_FLAG_SHUFFLE_TILES 1u /* centre-out tile order with random noise (screen_block.rs:74-78); default: row-major */
#define MP_FLAG_TRAVERSAL_GROUPS 2u /* camera rays on the 8-lane-group traversal instead of 64-ray packets (same results) */
/* BUILD-DEFINED EXTENSION, no reference counterpart (the reference has no bounce loop: worker.rs:51-66 is one primary ray and
 * |d.n|).  Diffuse grey surfaces (albedo 0.75) under a uniform white sky, paths of at most max_depth segments, cosine-weighted
 * bounces drawn from the same per-sample Xoshiro stream (UnitDisc rejection + sqrt, Duff et al. basis), origin offset 1e-4
 * along the normal.  rgba = (L, L, L, primary hit ? 1 : 0).  Defined operation by operation in oracle/minipath_oracle.c. */
#define MP_FLAG_PATHS 4u
/* Progressive accumulation / checkpoint-resume (SURVEY "aux subsystems"; BASELINE configs[4] is a 65 536-spp progressive
 * render): with this flag the tile buffer of mp_render_tiles_device[_counted] carries the running per-pixel state of
 * worker.rs:40-43 between launches -- rgb = the sequential f32 sample sum, a = the hit count.  A launch reads it when
 * pass_begin > 0, adds its samples in index order, and writes it back; the launch that reaches sample_count writes the means
 * of worker.rs:44 instead.  Sample s of a pixel is the same ray whichever launch draws it (its stream is keyed by
 * sample_count and s), so any split of the samples over launches is bit-identical to one launch; the buffer plus the next
 * pass_begin is the checkpoint (minipath_amd.io.save_checkpoint). */
#define MP_FLAG_ACCUMULATE 8u
/* With MP_FLAG_PATHS: staged ("wavefront") evaluation.  The paths of a batch of tiles live in HBM as SoA streams (RNG state, ray,
 * throughput, hit); between two segments the live paths are counting-sorted by (tile, direction bin) -- stream compaction plus
 * cache locality -- and traced 64 per wavefront by the same eight-lanes-per-ray traversal the fused kernel uses (walking the 64
 * sorted rays as ONE packet was measured 5x slower: DESIGN.md 4.4).  Same frame, bit for bit, as without the flag (each ray's
 * result is independent of its neighbours; the per-pixel sum is taken in sample order at the end).  tile_order / d_tile_cost of
 * mp_launch_extras are ignored in this mode. */
#define MP_FLAG_WAVEFRONT 16u
/* BUILD-DEFINED accumulation rule for very long sample chains (BASELINE configs[4]: 65 536 spp progressive).  The reference adds
 * every sample of a pixel into one f32 (worker.rs:40-43); that chain's rounding error grows with its length.  With this flag the
 * samples are summed in f32, in index order, over chunks of 256 consecutive samples (chunk c = samples [256c, 256c+256)), every
 * chunk sum is added to an f64 total, and the pixel is (f32)(total * (1.0 / (f64)sample_count)).  Any split of the samples over
 * MP_FLAG_ACCUMULATE passes gives the same bits (the tile buffer then carries {f32 chunk sum, f64 total (2 floats), hit count}
 * per pixel between launches).  Defined the same way in oracle/minipath_oracle.c (mpo_set_chunked_sum). */
#define MP_FLAG_CHUNKED_SUM 32u
/* render() only (mp_render_begin / mp_render_begin_multi): keep just the reference's image -- the u8 RgbaImage of machinery.rs:34 --
 * on the host.  Without the flag the library also reads back and files the pre-quantisation f32 means (mp_render_image_f32, a
 * build-defined extra: 16 bytes per pixel over PCIe and through the filing thread); with it mp_render_image_f32 is
 * MP_ERR_UNSUPPORTED.  The u8 image is the same either way. */
#define MP_FLAG_IMAGE_U8_ONLY 64u

/* machinery.rs:180-189 RenderProgressSnapshot */
typedef struct { size_t finished, total; } mp_progress;

/* triangle_bvh/mod.rs:20-30 TriangleBvh, as counts */
typedef struct {
    uint32_t root_link;
    uint32_t inner_count;    /* InnerNode, 128 B each in the reference layout */
    uint32_t packet_count;   /* RelativeTriangle8, 144 B each */
    uint32_t vertex_count;
    uint32_t triangle_count; /* real (unpadded) triangles */
    uint32_t depth;          /* max inner nodes on a root-to-leaf path */
    uint32_t stack_bound;    /* exact upper bound of the traversal stack for any ray (<= 7*depth+1) */
    float bbox_min[3], bbox_max[3];
    uint64_t device_bytes;
    uint32_t material_count; /* max TriangleShadingData.material + 1 (1 for everything the reference builds, building.rs:201) */
    uint32_t reserved;
} mp_scene_info;

/* triangle_bvh/mod.rs:20-53 TriangleBvh as arrays in the reference's own layout -- what mp_scene_export emits and
 * mp_scene_from_arrays takes:
 *   inner_nodes : inner_count x 128 B  InnerNode       = RelativeBox8 {min.x,min.y,min.z,max.x,max.y,max.z : u16[8]} + u32 links[8]
 *   packets     : packet_count x 144 B RelativeTriangle8 = 3 vertices x 3 coords x u16[8]
 *   tri_shading : packet_count*8 x 16 B {u32 vertex_indices[3]; u32 flat_shading}         (TriangleShadingData, usize -> u32)
 *   tri_material: packet_count*8 x u32  TriangleShadingData.material, may be NULL (= 0, building.rs:201)
 *   vertex_normals / vertex_tex : vertex_count x 3 f32 (VertexShadingData); vertex_tex may be NULL (origin)
 *   root_link, bbox : TriangleBvh.root / .bounding_box */
typedef struct {
    const void *inner_nodes;
    const void *packets;
    const void *tri_shading;
    const uint32_t *tri_material;
    const float *vertex_normals;
    const float *vertex_tex;
    uint32_t inner_count, packet_count, vertex_count;
    uint32_t root_link;
    float bbox_min[3], bbox_max[3];
} mp_bvh_desc;

/* BUILD-DEFINED path extension (MP_FLAG_PATHS): diffuse material, indexed by TriangleShadingData.material.  The reference carries
 * `material: usize` and `texture_coords` in every HitRecord (geometry/mod.rs:78-79, interpolated at ray_bvh_intersection.rs:80-83)
 * but only ever writes material 0 (building.rs:201) and reads neither.  Reflectance and emitted radiance per colour channel;
 * texture = MP_TEXTURE_CHECKER reads HitRecord.texture_coords: albedo2 replaces albedo on the odd cells of a checkerboard,
 * cell = floor(tex.x * texture_scale) + floor(tex.y * texture_scale), odd iff cell * 0.5 has a fractional part (NaN counts as odd).
 * Operation by operation in oracle/minipath_oracle.c (render_sample_paths_impl).  A table of grey (r = g = b), untextured
 * materials renders exactly as the scalar {albedo, emission} table of rounds 1-2 did.  Coloured / textured tables are not
 * combined with MP_FLAG_CHUNKED_SUM (its 16-byte pixel state carries one channel): MP_ERR_UNSUPPORTED. */
#define MP_TEXTURE_NONE 0u
#define MP_TEXTURE_CHECKER 1u
typedef struct {
    float albedo[3];
    float emission[3];
    float albedo2[3];       /* MP_TEXTURE_CHECKER: reflectance of the odd cells */
    uint32_t texture;       /* MP_TEXTURE_* */
    float texture_scale;    /* checker cells per unit of texture coordinate */
    uint32_t reserved;      /* 0 */
} mp_material;
/* Material ids (mp_scene_from_triangles_mat, mp_bvh_desc.tri_material, OBJ `usemtl`) must be below this: a scene's material
 * table has max id + 1 entries, so an unbounded id would size a table of gigabytes (MP_ERR_INVALID otherwise). */
#define MP_MAX_MATERIALS 65536u

/* geometry/mod.rs:71-80 HitRecord, batched SoA on the device (any pointer may be NULL to skip that output) */
typedef struct {
    float *d_t;             /* f32::MAX on miss (ray_bvh_intersection.rs:35) */
    uint32_t *d_prim;       /* packet*8+lane, MP_NO_PRIM on miss */
    float *d_u, *d_v;       /* LeafHitRecord.uv */
    float *d_point;         /* n*3, HitRecord.point   (optional) */
    float *d_normal;        /* n*3, HitRecord.normal  (optional) */
    float *d_tex;           /* n*3, HitRecord.texture_coords (optional) */
    uint32_t *d_material;   /* HitRecord.material (geometry/mod.rs:78), 0 on miss (optional) */
    uint32_t *d_instance;   /* object groups (mp_scene_group / mp_scene_instances): index of the member that was hit, 0 otherwise (optional) */
} mp_hits_soa;

typedef void (*mp_tile_started_cb)(void *user, mp_block tile);                       /* F1, machinery.rs:22 */
typedef void (*mp_tile_finished_cb)(void *user, mp_block tile, mp_progress snapshot); /* F2, machinery.rs:23 */

const char *mp_last_error(void);
const char *mp_version(void);

/* ---- context ---------------------------------------------------------------------------------------- */
int mp_ctx_create(int device_id, mp_ctx **out);
void mp_ctx_destroy(mp_ctx *ctx);
int mp_ctx_device(const mp_ctx *ctx, int *device_id, int *cu_count);
/* Tuning knobs.  "packet_stack_registers" (1..64, default 64): entries of the ray-packet walk's shared stack kept in registers; the
 * rest of the scene's stack bound lives in LDS.  "packet_samples_in_flight" (0 = automatic, or 1, 2, 4, ... 64): samples of one pixel
 * a wavefront traces per pass (64 / value pixels side by side); sets the size of a work unit.  "packet_rays_per_lane" (1 default,
 * or 2): 2 = 128-ray walks, two rays per lane (measured slower on MI355X; kept as the measured alternative).  "blocks_per_cu"
 * (0 = as many as fit, or 1..8): resident workgroups per CU, a diagnostic knob for occupancy studies.  "packet_mask_cache" (1
 * default): the packet walk's per-work-unit cache of "children / triangles no ray of the unit can hit" masks (exact: interval
 * evaluation of the reference's own tests on bounds of the unit's rays): 0 = off, 1 or 2 = on wherever a work unit has at least
 * four passes (2 is kept for callers of earlier builds, where 1 meant big scenes only).  "paths_pooled" (MP_FLAG_PATHS
 * without MP_FLAG_WAVEFRONT; 1 default): 0 = one pass of 8 samples per walk of the bounce rays, 2 / 3 = two / up to four passes share
 * one walk over a per-wave ray queue in global memory (fewer idle lane groups at the end of every walk), 1 = the latter for scenes
 * whose traversal arrays exceed 1 MB.  "render_batch_tiles" (0 = automatic, default): tiles per launch of mp_render_begin's workers.
 * Results never depend on any of them (tests sweep them). */
int mp_ctx_set_option(mp_ctx *ctx, const char *key, int value);

/* ---- camera.rs ------------------------------------------------------------------------------------------ */
int mp_camera_default(mp_camera *cam);                                                            /* :42-52 */
int mp_camera_look_at(mp_camera *cam, const float eye[3], const float at[3], const float up[3]);   /* :93-101 */
int mp_camera_look_direction(mp_camera *cam, const float eye[3], const float fwd[3], const float up[3]); /* :104-116 */
int mp_camera_translate(mp_camera *cam, const float t[3]);                                        /* :119-121 */
int mp_camera_basis(const mp_camera *cam, float center[3], float fwd[3], float up[3], float right[3]); /* :148-171 */
int mp_camera_build_sampler(const mp_camera *cam, uint32_t width, uint32_t height, mp_camera_sampler *out); /* :123-146 */

/* ---- screen_block.rs ------------------------------------------------------------------------------------ */
/* ScreenBlock::tile_ordering :46-81.  out==NULL returns the count in *n.  shuffle_seed 0 = row-major grid. */
int mp_tile_ordering(mp_block block, uint32_t tile_size, uint64_t shuffle_seed, mp_block *out, size_t cap, size_t *n);

/* ---- scene/triangle_bvh ---------------------------------------------------------------------------------- */
/* TriangleBvh::with_obj building.rs:28-34 : load + dedupe + build + upload to HBM.
 * ctx may be NULL: the scene is then host-only (mp_scene_info_get / mp_scene_export work, rendering does not). */
int mp_scene_from_obj(mp_ctx *ctx, const char *path, mp_scene **out);
/* TriangleBvh::build building.rs:83-107 over caller-supplied indexed triangles (arrays are copied).
 * normals/tex may be NULL (=> zero normals => flat shading, building.rs:200). */
int mp_scene_from_triangles(mp_ctx *ctx, const float *positions, const float *normals, const float *tex,
                            uint32_t vertex_count, const uint32_t *indices, uint32_t triangle_count, mp_scene **out);
/* Same, with a material id per triangle (NULL = all 0, which is what the reference writes, building.rs:201). */
int mp_scene_from_triangles_mat(mp_ctx *ctx, const float *positions, const float *normals, const float *tex,
                                uint32_t vertex_count, const uint32_t *indices, const uint32_t *tri_material,
                                uint32_t triangle_count, mp_scene **out);
/* A TriangleBvh the caller already holds (e.g. the Rust reference's own tree, fields of triangle_bvh/mod.rs:20-30): arrays are
 * copied, links and indices validated (inner nodes must be in pre-order: children after their parent, as building.rs emits
 * them), nothing is rebuilt -- triangle_index values and the lane order inside leaves are the caller's.  Inverse of
 * mp_scene_export.  ctx may be NULL (host-only). */
int mp_scene_from_arrays(mp_ctx *ctx, const mp_bvh_desc *desc, mp_scene **out);
/* Material table (n >= material_count entries, copied) and sky radiance of the build-defined path extension.  Defaults: one
 * material {0.75, 0}, sky 1.  Not used by the reference semantics (depth 1). */
int mp_scene_set_materials(mp_scene *scene, const mp_material *table, uint32_t n, float sky_radiance);
/* `usemtl` name of material id (OBJ scenes; "" for id 0 = faces before any usemtl); NULL if id >= material_count. */
const char *mp_scene_material_name(const mp_scene *scene, uint32_t id);
/* scene/primitives.rs:10-56 Sphere as the scene's Object (analytic intersection, no BVH).  ctx may be NULL (host-only). */
int mp_scene_sphere(mp_ctx *ctx, const float center[3], float radius, mp_scene **out);
/* BUILD-DEFINED Object (scene/mod.rs:7-10 `trait Object`; the reference's Scene holds ONE object and has no transforms): a
 * top-level list of n members {object, rigid transform}.  objects: n TriangleBvh or Sphere scenes of this context (the
 * reference's two Object implementations), repeats allowed, groups do not nest.  Transform of member k: world = q_k * local +
 * t_k with translations = n*3 floats and rotations = n*4 floats, unit quaternions (i, j, k, w) as nalgebra stores them (the type
 * of the reference's camera isometry, camera.rs:9-19), or NULL for translations only; both copied.
 * intersect = for every member in order the member's own intersect with the ray moved into the member's frame -- origin -
 * translation, then origin and direction through the inverse rotation; the direction is not re-normalised, so t stays the
 * world ray's -- closest wins with a strict `<` (the first member keeps ties).  HitRecord.point = point_at(t) of the world ray,
 * normal = q * the member's normal, tex / material id are the member's; mp_hits_soa.d_instance = index of the member that was
 * hit, prim = triangle index inside that member (0, material 0 and texture_coords 0 for a Sphere member, primitives.rs:40-46).
 * The group has ONE material table indexed by the members' material ids (initially the first member's, padded with the default
 * material; mp_scene_set_materials replaces it) and the first member's sky radiance.  get_bounding_box = union of the members'
 * boxes in the world frame (rotated member: the box of its box's rotated corners); mp_scene_info counts are sums over the
 * members.  The group SHARES its members' device arrays and holds a reference on each member: mp_scene_destroy on a member only
 * drops the caller's handle, the arrays are freed when the last group using them is destroyed too.  Rendered by every kernel: camera rays
 * walk each TriangleBvh member as a 64-ray packet, bounce rays on the 8-lane-group traversal, and the staged MP_FLAG_WAVEFRONT
 * pipeline takes groups too; defined operation by operation in oracle/minipath_oracle.c (bvh_intersect_impl). */
int mp_scene_group(mp_ctx *ctx, const mp_scene *const *objects, const float *rotations, const float *translations, uint32_t n,
                   mp_scene **out);
/* n members that are all `object` (instancing): as mp_scene_group, except that mp_scene_info reports the object's own counts and
 * mp_scene_export exports the object's arrays. */
int mp_scene_instances(mp_ctx *ctx, const mp_scene *object, const float *translations, uint32_t n, mp_scene **out);
void mp_scene_destroy(mp_scene *scene);
int mp_scene_info_get(const mp_scene *scene, mp_scene_info *out);
/* Export of the reference-layout arrays (for parity checks and for a Rust caller that wants to rebuild a
 * TriangleBvh): inner nodes 128 B (6 x u16[8] min.xyz,max.xyz + 8 x u32 links), packets 144 B (3 verts x 3
 * coords x u16[8]), tri shading 16 B (3 x u32 vertex index + u32 flat), vertex normals / tex (n*3 f32).
 * Any pointer may be NULL. */
int mp_scene_export(const mp_scene *scene, void *inner_nodes, void *packets, void *tri_shading, float *vertex_normals,
                    float *vertex_tex, uint32_t *tri_material);
/* Diagnostics / tests (no reference counterpart): the traversal-format node array the kernels walk, rebuilt from
 * the host tree by the code the upload uses; works on host-only scenes.  which = 0: the WIDE tree -- thin inner nodes of the
 * reference tree absorbed into their parents where every child box is contained, in floating point, in its parent's box, which
 * makes the hits of rays with finite inverse directions bit-identical (argument: minipath_amd/csrc/device_tree.cpp); which = 1:
 * the literal reference tree (InnerNode n = node n), walked by rays with an infinite inverse direction component.
 * nodes (nullable): count x 64 dwords = 8 child records {min.xyz, max.xyz (absolute f32), link, n}; link (u32 bits): inner =
 * node index << 6, leaf = first packet << 6 | real triangles (1..56), null = 0xFFFFFFF8; n (u32 bits, record 0 only) = index of
 * the last real child + 1.  absorbed = reference nodes that are no longer nodes of their own.  Any out pointer may be NULL. */
int mp_scene_device_tree(const mp_scene *scene, int which, float *nodes, uint32_t *count, uint32_t *root_link,
                         uint32_t *stack_bound, uint32_t *absorbed);

/* ---- impl Object for TriangleBvh :: intersect, batched (ray_bvh_intersection.rs:26-96) -------------------- */
/* d_o/d_d: SoA device arrays of n floats each (ox,oy,oz / dx,dy,dz).  Directions need not be unit: Ray::new
 * (geometry/mod.rs:45-54) is applied on the device. */
int mp_trace_rays(mp_ctx *ctx, const mp_scene *scene, const float *d_ox, const float *d_oy, const float *d_oz,
                  const float *d_dx, const float *d_dy, const float *d_dz, uint64_t n, const mp_hits_soa *hits,
                  void *stream);
/* CameraSampler::sample_ray (camera.rs:176-191) batched, seeded mode: writes unit-direction rays for sample s of
 * every pixel of `block` (x fastest) -- the ray stream of the staged (wavefront) pipeline. */
int mp_generate_rays(mp_ctx *ctx, const mp_camera_sampler *sampler, const mp_settings *settings, mp_block block,
                     uint32_t sample, float *d_ox, float *d_oy, float *d_oz, float *d_dx, float *d_dy, float *d_dz,
                     void *stream);

/* ---- Worker::render_tile (worker.rs:32-49) ---------------------------------------------------------------- */
/* Synchronous, host output.  rgba_f32 = w*h*4 pre-quantisation means (worker.rs:44), x fastest; rgba_u8 =
 * color_to_image (worker.rs:69-76).  Either may be NULL. */
int mp_render_tile(mp_ctx *ctx, const mp_scene *scene, const mp_camera_sampler *sampler, const mp_settings *settings,
                   mp_block tile, float *rgba_f32, uint8_t *rgba_u8);
/* Device output, asynchronous on `stream`: renders `n_tiles` tiles in ONE launch.  Tile i's pixels go to
 * d_rgba_f32 + i*tile_size*tile_size*4 (tile-major, row stride = tile_size, clipped tiles leave the rest
 * untouched).  This is the bench hot path and the per-rank shard of the multi-GPU render. */
int mp_render_tiles_device(mp_ctx *ctx, const mp_scene *scene, const mp_camera_sampler *sampler,
                           const mp_settings *settings, const mp_block *tiles, size_t n_tiles, float *d_rgba_f32,
                           void *stream);
/* Same, and *d_ray_segments (device u64, overwritten) receives the number of Object::intersect calls of the launch:
 * pixels*spp for the reference semantics, the traced path segments with MP_FLAG_PATHS. */
int mp_render_tiles_device_counted(mp_ctx *ctx, const mp_scene *scene, const mp_camera_sampler *sampler,
                                   const mp_settings *settings, const mp_block *tiles, size_t n_tiles, float *d_rgba_f32,
                                   uint64_t *d_ray_segments, void *stream);
/* Same, with optional extras (any member may be NULL).  The waves of a launch take tiles from one queue, like the reference's
 * workers (machinery.rs:206-208 get_next_tile), in the order of the `tiles` array -- or, with `tile_order` (host, a permutation
 * of 0..n_tiles-1), in that order while tile i still renders to slot i.  `d_tile_cost` (device u64[n_tiles]) is incremented by the
 * shader-clock cycles the waves spent on each tile: handing out the expensive tiles first (order = argsort of the previous
 * frame's or progressive pass's cost, descending) shortens the tail of a launch, most of all for the small per-rank launches of a
 * multi-GPU frame.  Results do not depend on the order. */
typedef struct {
    uint64_t *d_ray_segments;   /* as in mp_render_tiles_device_counted */
    uint64_t *d_tile_cost;      /* device u64[n_tiles], accumulated (+=) */
    const uint32_t *tile_order; /* host u32[n_tiles] */
} mp_launch_extras;
int mp_render_tiles_device_ex(mp_ctx *ctx, const mp_scene *scene, const mp_camera_sampler *sampler,
                              const mp_settings *settings, const mp_block *tiles, size_t n_tiles, float *d_rgba_f32,
                              const mp_launch_extras *extras, void *stream);
/* machinery.rs:78-89 (tile buffer -> image copy) on the device: scatters tile-major tiles into an image-major
 * f32 frame and/or its color_to_image u8 frame (either may be NULL). */
int mp_untile(mp_ctx *ctx, const mp_settings *settings, const mp_block *tiles, size_t n_tiles,
              const float *d_tiles_f32, float *d_image_f32, uint8_t *d_image_u8, void *stream);
/* Preview of an unfinished MP_FLAG_ACCUMULATE tile buffer after samples_done (< sample_count) samples, scattered into an
 * image-major frame like mp_untile; the buffer is only read.  BUILD-DEFINED (the reference's progressive consumer, gui.rs:216-224,
 * re-renders with a smaller sample_count instead): pixel = running sum * (1.0f / (f32)samples_done), alpha = hits * the same
 * (worker.rs:44 for the samples drawn so far); under MP_FLAG_CHUNKED_SUM (f32)((total + (f64)chunk sum) * (1.0 / (f64)samples_done)).
 * samples_done == sample_count: the buffer already holds the means, plain mp_untile. */
int mp_untile_preview(mp_ctx *ctx, const mp_settings *settings, const mp_block *tiles, size_t n_tiles,
                      const float *d_tiles_f32, uint32_t samples_done, float *d_image_f32, uint8_t *d_image_u8, void *stream);
/* One whole frame over n GPUs of this process, device-resident (SURVEY 8e in one process): rank r renders tiles r, r+n, ... of
 * the row-major tile grid in ONE launch on ctxs[r]'s own stream into its shard and sends the shard to ctxs[0]'s device on that
 * same stream (hipMemcpyPeerAsync: n copies on n streams, every peer over its own xGMI link, overlapping the other ranks'
 * renders); `stream` (a stream of ctxs[0]'s device) waits for the n copies and scatters the gathered tiles into d_image_f32 /
 * d_image_u8 (image-major, either may be NULL).  Peer access is checked once per device pair (hipDeviceCanAccessPeer, both
 * directions); without it the shard is staged through pinned host memory.  Asynchronous: the frame is complete when `stream`
 * is.  All samples of a pixel stay on one device, so the image does not depend on n.  *ray_segments (host, optional) receives
 * the Object::intersect calls of the frame for the reference semantics.  Contexts must be distinct; two may share a device.
 * Frames on the same ctxs[0] are serialised; a context takes part in one multi-device frame at a time. */
int mp_render_frame_multi(mp_ctx *const *ctxs, const mp_scene *const *scenes, int n, const mp_camera_sampler *sampler,
                          const mp_settings *settings, float *d_image_f32, uint8_t *d_image_u8, uint64_t *ray_segments,
                          void *stream);
/* Progressive form (BASELINE configs[4]; SURVEY 8e: "accumulators stay sharded; gather only per displayed pass / at end"):
 * settings carries MP_FLAG_ACCUMULATE and names the pass [pass_begin, pass_begin + pass_count).  Every rank adds the pass to the
 * running state of ITS shard, which stays on its device between calls (pass_begin must continue the previous call's pass with
 * the same settings, ranks and scenes; pass_begin 0 starts over).  gather = 0: nothing leaves the devices.  gather != 0: the
 * shards are gathered and scattered as in mp_render_frame_multi -- after the last pass the finished frame (bit-identical to one
 * mp_render_frame_multi), after an earlier pass a preview as mp_untile_preview defines it (the shards keep their state). */
int mp_render_pass_multi(mp_ctx *const *ctxs, const mp_scene *const *scenes, int n, const mp_camera_sampler *sampler,
                         const mp_settings *settings, int gather, float *d_image_f32, uint8_t *d_image_u8,
                         uint64_t *ray_segments, void *stream);
/* Rays traced (Object::intersect calls) and their wall time inside the last mp_render_tiles_device launch is
 * NOT measured here: the caller brackets the stream with events. */

/* ---- render() / RenderProgress (machinery.rs:20-178) ------------------------------------------------------ */
/* Asynchronous like render(): returns after starting a library-owned host thread that walks the tile ordering,
 * launching batches of tiles on the GPU; callbacks (may be NULL) are invoked from that thread once per tile
 * start and once per tile end. */
int mp_render_begin(mp_ctx *ctx, const mp_scene *scene, const mp_camera *camera, const mp_settings *settings,
                    mp_tile_started_cb started, mp_tile_finished_cb finished, void *user, mp_render **out);
/* render() over n GPUs of this process (machinery.rs:51-116 with devices in place of cores): one library-owned host thread
 * per (ctxs[i], scenes[i]) pair pulls batches of tiles from ONE shared atomic queue (get_next_tile, :205-208), renders them on
 * its device and files them into the one host image under its lock; callbacks come from all n threads (concurrently, like the
 * reference's worker threads).  scenes[i] is the same scene uploaded to ctxs[i].  mp_render_begin == n = 1.  Several contexts
 * may share a device (this is how the path is tested on a one-GPU box). */
int mp_render_begin_multi(mp_ctx *const *ctxs, const mp_scene *const *scenes, int n, const mp_camera *camera,
                          const mp_settings *settings, mp_tile_started_cb started, mp_tile_finished_cb finished, void *user,
                          mp_render **out);
int mp_render_progress(const mp_render *r, mp_progress *out);        /* RenderProgress::progress :133-142 */
int mp_render_is_finished(const mp_render *r, int *finished);        /* ::is_finished :144-146 */
int mp_render_elapsed_ns(const mp_render *r, uint64_t *ns);          /* ::elapsed :150-157 */
int mp_render_abort(mp_render *r);                                   /* ::abort :161-165 */
int mp_render_wait(mp_render *r);                                    /* ::wait :169-173 ; returns the worker's status */
int mp_render_image_u8(mp_render *r, uint8_t *dst /* w*h*4 */);      /* ::image :175 (copy under the lock) */
int mp_render_image_f32(mp_render *r, float *dst /* w*h*4 */);       /* pre-quantisation means (worker.rs:44) */
void mp_render_destroy(mp_render *r);

#ifdef __cplusplus
}
#endif
#endif

// Internal declarations shared by the host sources and the HIP kernels of libminipath_hip.so.
// Nothing here is part of the C ABI (include/minipath_hip.h is).
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/minipath_hip.h"

namespace mp {

// ---- reference-layout host structures (triangle_bvh/mod.rs:33-45, compressed_geometry.rs) -----------------
struct InnerNodeRef {       // 128 B
    uint16_t bmin[3][8];
    uint16_t bmax[3][8];
    uint32_t link[8];
};
struct TriPacketRef {       // 144 B
    uint16_t v[3][3][8];    // [vertex][coord][lane]
};
struct TriShadingRef {      // 16 B
    uint32_t vi[3];
    uint32_t flat;
};
static_assert(sizeof(InnerNodeRef) == 128, "InnerNode layout");
static_assert(sizeof(TriPacketRef) == 144, "RelativeTriangle8 layout");

struct Box3 {
    float mn[3], mx[3];
};

// Result of the host pre-pass (building.rs).  `node_box` / `leaf_box` hold the decompressed enclosing box of
// every inner node / of every leaf (keyed by the leaf's first packet), i.e. the box chain of SURVEY A.4.
struct HostBvh {
    Box3 bbox{};
    uint32_t root = MP_LINK_NULL;
    uint32_t depth = 0;
    uint32_t triangle_count = 0;
    std::vector<InnerNodeRef> inner;
    std::vector<Box3> inner_box;          // enclosing box each inner node was built against
    std::vector<TriPacketRef> packets;
    std::vector<Box3> packet_box;         // enclosing (leaf) box of each packet
    std::vector<TriShadingRef> shading;   // packets*8
    std::vector<uint32_t> material;       // packets*8 : TriangleShadingData.material (mod.rs:44); 0 for padding
    std::vector<float> vnormal, vtex;     // nv*3
    uint32_t vertex_count = 0;
    std::vector<std::string> material_names;  // [0] = "" (faces before any usemtl), then `usemtl` names in first-seen order
};

// building.rs:83-107.  Returns MP_OK or an error code with `err` filled.  tri_mat (nullable): material id per input triangle
// (the reference writes `material: 0`, building.rs:201).
int build_bvh(const float* pos, const float* nrm, const float* tex, uint32_t nv, const uint32_t* tri, const uint32_t* tri_mat,
              uint32_t nt, HostBvh& out, std::string& err);
// building.rs:28-81 (+ `usemtl` -> per-triangle material ids, which the reference ignores)
int load_obj(const char* path, std::vector<float>& pos, std::vector<float>& nrm, std::vector<float>& tex,
             std::vector<uint32_t>& tri, std::vector<uint32_t>& tri_mat, std::vector<std::string>& material_names, std::string& err);
// A TriangleBvh handed over as its reference-layout arrays (mp_scene_from_arrays): validates the links, recomputes the box
// chain, depth and triangle count.
int bvh_from_arrays(const mp_bvh_desc& d, HostBvh& out, std::string& err);

// Traversal-format node array (device_tree.cpp): `count` nodes x 8 child records x 8 dwords (+ two records of tail padding; in the
// wide tree the first of them is the ROOT's record: an unbounded box and the root's dlink, slot count * 8), the dlink of the
// root, the exact traversal-stack bound and whether every real child box has min <= max.  wide = thin nodes absorbed into
// their parents (the tree the walks use for rays with finite inverse directions); otherwise the literal reference tree.
struct DeviceTree {
    std::vector<float> nodes;
    uint32_t count = 0;
    uint32_t root = MP_LINK_NULL;
    uint32_t stack_bound = 1;
    uint32_t absorbed = 0;       // reference nodes that no longer exist as nodes of their own (wide tree)
    bool boxes_ordered = true;
};
std::vector<uint32_t> packet_real_counts(const HostBvh& h);  // real (unpadded) triangles of each packet
int build_device_tree(const HostBvh& h, const std::vector<uint32_t>& pkt_valid, bool wide, DeviceTree& out, std::string& err);

// ---- device scene ("traversal format") -------------------------------------------------------
// nodes_aos : node_count x 8 children x 8 dwords {minx,miny,minz,maxx,maxy,maxz,dlink,n}: absolute decompressed child boxes;
//             n (record 0 only) = index of the node's last real child + 1.  This is the WIDE tree (device_tree.cpp: thin nodes
//             absorbed into their parents, bit-identical hits for rays with finite inverse directions); nodes_lit / root_lit are
//             the literal reference tree in the same format, walked by rays with an infinite inverse direction component.
// dlink     : device-private link (the reference's CompressedNodeLink idx<<3|count, mod.rs:57-114, re-encoded so that a leaf needs
//             no side lookup): inner = node index << 6 ; leaf = first packet << 6 | real (unpadded) triangles of the leaf (1..56) ;
//             null = MP_LINK_NULL unchanged (checked before decoding; scenes are limited to 2^24 nodes and 14.9 M packets:
//             the walks address records with 32-bit byte offsets).
// tris_aos  : packet_count x 8 triangles x kTriDwords (9) dwords {v0.xyz,e1.xyz,e2.xyz}: decompressed v0 and the edges e1=v1-v0,
//             e2=v2-v0 of triangle.rs:195-196 (+ 3 records of tail padding: the packet walk fetches two triangles ahead).
//             The packet walk reads both through the scalar unit (wave-uniform), the 8-lane-group walk with per-lane vector loads
//             (lane i = child i / triangle i).
// pkt_valid : real (unpadded) triangles of each packet (host staging for dlink; kept on the device for diagnostics).
// shade     : packet_count*8 x 3 float4 : n0.xyz n1.xyz n2.xyz flat(u32 bits) material(u32 bits) pad.  48 B per triangle slot.
// vidx/vtex : for the full HitRecord (texture_coords).
// One member {object, translation} of a build-defined object group (mp_scene_group / mp_scene_instances): the member's own
// traversal arrays (borrowed from its scene) or its sphere, and the rigid transform that places it.  136 bytes.
struct DevObject {
    const float* shade;
    const float* nodes_aos;
    const float* tris_aos;
    const uint32_t* vidx;
    const float* vtex;
    const float* nodes_lit;
    uint32_t root;
    uint32_t has_pre;
    float pre_min[3], pre_max[3];
    float t[3];
    uint32_t kind;              // 0 = TriangleBvh, 1 = Sphere (scene/primitives.rs:10-13)
    float sphere_center[3];
    float sphere_radius;
    float q[4];                 // rotation (unit quaternion i, j, k, w): world = q * local + t
    uint32_t rotated;           // 0 = translation only (q is not applied: the ray keeps its exact direction)
    uint32_t root_lit;
};
static_assert(sizeof(DevObject) == 136, "DevObject layout");

struct DevScene {
    uint32_t kind = 0;               // 0 = TriangleBvh, 1 = Sphere (scene/primitives.rs:10-13)
    float sphere_center[3] = {0, 0, 0};
    float sphere_radius = 0;
    const float* shade = nullptr;
    const float* nodes_aos = nullptr;  // wide tree
    const float* nodes_lit = nullptr;  // literal tree
    const float* tris_aos = nullptr;
    const uint32_t* pkt_valid = nullptr;
    const uint32_t* vidx = nullptr;  // packets*8*3
    const float* vtex = nullptr;     // nv*3
    const float* materials = nullptr;  // build-defined path extension: mp_material records (12 dwords) per material id (device)
    uint32_t materials_rgb = 0;        // ... some material is coloured or textured: three-channel path kernels
    float sky = 1.0f;                  // ... and the sky radiance
    uint32_t inst_count = 0;           // build-defined object group: number of members; 0 = plain BVH
    const struct DevObject* objects = nullptr;  // ... and their descriptors (device)
    uint32_t root = MP_LINK_NULL;    // dlink of the root (wide tree)
    uint32_t root_lit = MP_LINK_NULL;  // ... in the literal tree
    uint32_t inner_count = 0;        // nodes of the wide tree
    uint32_t packet_count = 0;
    uint32_t stack_cap = 1;          // exact traversal-stack bound (upload_scene)
    uint32_t packet_stack_regs = 64; // entries of the packet walk's stack held in registers (test knob, <= 64)
    uint32_t boxes_ordered = 0;      // every real child box has min <= max on all axes (sign-specialised slab test allowed)
    uint32_t tris_bounded = 0;       // every triangle record is finite with |v0| <= 2^30, |e1|, |e2| <= 2^31 (packet-level triangle masks allowed: tri_may_hit)
    // union of the root inner node's (non-null) decompressed child boxes: exact conservative ray pre-test
    uint32_t has_pre = 0;
    float pre_min[3] = {0, 0, 0}, pre_max[3] = {0, 0, 0};
};

// A launch's work units are dealt to kWorkQueues interleaved queues (unit u belongs to queue u % kWorkQueues), one per XCD: a wave
// draws from the queue of its XCD (workgroups go to XCDs round-robin) and moves on to the others when it runs dry.  One shared
// counter saturates on same-address atomics from 8192 waves once units get small.  Each head sits on its own 128-byte line.
constexpr uint32_t kWorkQueues = 8;
constexpr uint32_t kWorkQueueStride = 32;  // dwords

constexpr int kTriDwords = 9;      // device triangle record: v0.xyz, e1.xyz, e2.xyz, densely packed (36 B: 25 % fewer cache lines than a 48-B padded record)
constexpr int kNodeDwords = 56;    // host staging rows: minx..maxz (8 floats each) + link[8]
constexpr int kPacketDwords = 72;  // host staging rows: v0.xyz, e1.xyz, e2.xyz (8 floats each)

// ---- kernel launchers (kernels.hip) ---------------------------------------------------------------------
struct RenderLaunch {
    DevScene scene;
    mp_camera_sampler sampler;
    uint32_t width, height, spp, tile_size;
    uint64_t seed;
    const mp_block* d_tiles;   // device copy of the tile list
    uint32_t n_tiles;
    float* d_out;              // tile-major f32 RGBA
    uint32_t* d_counter;       // kWorkQueues work-queue heads, kWorkQueueStride dwords apart, zeroed by the launcher
    int cu_count;
    int traversal;             // 0 = ray packets (coherent camera rays), 1 = 8-lane groups
    uint32_t max_depth;        // 0 = reference semantics (worker.rs:51-66); > 0 = build-defined path extension
    unsigned long long* d_segments;  // optional device counter of traced ray segments (path extension)
    // progressive accumulation (MP_FLAG_ACCUMULATE): samples [pass_begin, pass_end) of spp; carry_in: d_out holds the running
    // sums of the earlier passes; finalize: write means (worker.rs:44) instead of sums
    uint32_t pass_begin = 0, pass_end = 0;
    bool carry_in = false, finalize = true;
    bool chunked = false;         // MP_FLAG_CHUNKED_SUM
    uint32_t packet_samples = 0;  // samples of a pixel in flight per pass of the packet kernel (0 = automatic)
    uint32_t rays_per_lane = 1;   // 2: 128-ray walks (render_tiles_packet2_kernel) where applicable
    uint32_t mask_cache = 1;      // packet kernel: per-unit mask cache of the packet-level child / triangle rejection: 0 off, 1 / 2 on (units of >= 4 passes)
    uint32_t paths_pooled = 1;    // path extension: 0 = one pass per walk (render_paths_kernel), 1 = pooled passes for big scenes (auto), 2 / 3 = always pooled (two / up to four passes)
    const uint32_t* d_tile_order = nullptr;      // optional: hand-out order of the tiles (device, n_tiles)
    unsigned long long* d_tile_cost = nullptr;   // optional: += shader-clock cycles spent per tile (device, n_tiles)
};

int launch_render_tiles(const RenderLaunch& L, void* stream, std::string& err);
int launch_render_paths_wavefront(const RenderLaunch& L, void* stream, std::string& err);
int launch_trace_rays(const DevScene& sc, const float* ox, const float* oy, const float* oz, const float* dx,
                      const float* dy, const float* dz, uint64_t n, const mp_hits_soa& hits, int cu_count, void* stream,
                      std::string& err);
int launch_set_u64(unsigned long long* d_ptr, unsigned long long value, void* stream, std::string& err);
int launch_generate_rays(const mp_camera_sampler& s, uint32_t width, uint32_t spp, uint64_t seed, mp_block block,
                         uint32_t sample, float* ox, float* oy, float* oz, float* dx, float* dy, float* dz, void* stream,
                         std::string& err);
// preview_mode 0: finished pixels; 1 / 2: preview of an unfinished MP_FLAG_ACCUMULATE buffer (plain / MP_FLAG_CHUNKED_SUM state)
// after preview_samples samples (untile_kernel)
int launch_untile(uint32_t width, uint32_t height, uint32_t tile_size, const mp_block* d_tiles, uint32_t n_tiles,
                  const float* d_tiles_f32, float* d_image_f32, uint8_t* d_image_u8, void* stream, std::string& err,
                  uint32_t preview_mode = 0, uint32_t preview_samples = 0);

int launch_quantise(const float* d_rgba_f32, uint8_t* d_rgba_u8, uint64_t n_pixels, void* stream, std::string& err);

// camera / tiles (host_camera.cpp)
void camera_default(mp_camera& c);
void camera_look_at(mp_camera& c, const float eye[3], const float at[3], const float up[3]);
void camera_look_direction(mp_camera& c, const float eye[3], const float fwd[3], const float up[3]);
void camera_basis(const mp_camera& c, float center[3], float fwd[3], float up[3], float right[3]);
void camera_build_sampler(const mp_camera& c, uint32_t w, uint32_t h, mp_camera_sampler& out);
std::vector<mp_block> tile_ordering(mp_block block, uint32_t tile_size, uint64_t shuffle_seed);

void set_last_error(const std::string& s);

}  // namespace mp

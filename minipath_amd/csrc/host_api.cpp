// C ABI (include/minipath_hip.h): context, scene upload, synchronous tile rendering and the asynchronous
// render()/RenderProgress machinery (reference: src/renderer/machinery.rs).
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <mutex>
#include <thread>

#include "mp_internal.h"

namespace mp {

static thread_local std::string g_last_error;
void set_last_error(const std::string& s) { g_last_error = s; }

namespace {

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}
int hip_fail(hipError_t e, const char* what) { return fail(MP_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); }

// No exception crosses the C ABI: every extern "C" body runs inside guarded().
template <class F>
int guarded(F&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        try { g_last_error = "out of host memory (std::bad_alloc)"; } catch (...) {}
        return MP_ERR_NOMEM;
    } catch (const std::exception& ex) {
        try { g_last_error = std::string("unexpected exception: ") + ex.what(); } catch (...) {}
        return MP_ERR_INVALID;
    } catch (...) {
        try { g_last_error = "unexpected exception"; } catch (...) {}
        return MP_ERR_INVALID;
    }
}

#define MP_HIP(call)                                  \
    do {                                              \
        hipError_t e_ = (call);                       \
        if (e_ != hipSuccess) return hip_fail(e_, #call); \
    } while (0)

constexpr float kInvU16Max = 1.0f / 65535.0f;
inline float dequantise(uint16_t q, float size, float mn) {  // compressed_geometry.rs:48-51,95-110
    return std::fmaf(size, static_cast<float>(static_cast<int32_t>(q)) * kInvU16Max, mn);
}

// worker.rs:69-76 on the host (used by the synchronous / async host-image paths)
inline uint8_t to_u8(float c) {
    float x = std::round(c * 255.0f);
    if (x != x) return 0;
    x = x < 0.0f ? 0.0f : (x > 255.0f ? 255.0f : x);
    return static_cast<uint8_t>(x);
}

}  // namespace
}  // namespace mp

using namespace mp;

struct mp_ctx {
    int device = 0;
    int cu_count = 0;
    // work-queue heads: one per launch, handed out round-robin so that launches on different streams never share one
    static constexpr uint32_t kCounters = 256;
    uint32_t* d_counters = nullptr;
    std::atomic<uint32_t> next_counter{0};
    std::atomic<uint32_t> packet_stack_regs{64};
    std::atomic<uint32_t> packet_samples{0};  // 0 = chosen by the launcher
    std::atomic<uint32_t> rays_per_lane{1};   // packet kernel: 1 = 64-ray walks, 2 = 128-ray walks (two rays per lane)
    std::atomic<uint32_t> blocks_per_cu{0};   // 0 = as many as fit (diagnostic knob: resident workgroups per CU)
    std::atomic<uint32_t> render_batch{0};    // tiles per launch of render() (0 = automatic)
    std::atomic<uint32_t> mask_cache{1};      // packet kernel: per-unit mask cache of the packet-level child rejection
    std::atomic<uint32_t> paths_pooled{1};    // path extension: 0 = render_paths_kernel, 1 = auto, 2 / 3 = always pooled (RenderLaunch::paths_pooled)
    uint32_t* take_counter() {  // one set of work-queue heads per launch in flight
        return d_counters + static_cast<size_t>(next_counter.fetch_add(1, std::memory_order_relaxed) % kCounters) * kWorkQueues * kWorkQueueStride;
    }
    // Device copies of the tile lists (and hand-out orders) callers pass as host arrays.  A frame loop passes the same list every
    // frame: uploading it per launch would put a pageable host-to-device copy in the stream, which blocks the calling thread
    // until the stream has drained (measured: a full frame) and so serialises host and GPU.  Entries are compared by content.
    struct TileList {
        std::vector<mp_block> tiles;
        std::vector<uint32_t> order;
        unsigned char* d_buf = nullptr;  // tiles, then order
        uint64_t last_use = 0;
        int device = 0;
        TileList() = default;
        TileList(const TileList&) = delete;
        TileList& operator=(const TileList&) = delete;
        ~TileList() {  // runs when the cache AND every caller that still holds the list have let go; hipFree waits for the device
            if (!d_buf) return;
            int prev = -1;
            (void)hipGetDevice(&prev);
            if (prev != device) (void)hipSetDevice(device);
            (void)hipFree(d_buf);
            if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
        }
    };
    using TileListRef = std::shared_ptr<TileList>;
    // Buffers of the render() worker (stream, device tile-major f32/u8 + tile list, pinned host mirrors), kept between calls:
    // creating and freeing them costs several milliseconds per render() otherwise.
    struct WorkerSlot {
        hipStream_t stream = nullptr;
        float* d_f32 = nullptr;
        uint8_t* d_u8 = nullptr;
        mp_block* d_tiles = nullptr;
        float* h_f32 = nullptr;   // pinned
        uint8_t* h_u8 = nullptr;  // pinned
        size_t tiles = 0, per_tile = 0;  // capacity: `tiles` tile slots of per_tile floats
        size_t first = 0, n = 0;  // batch in flight
        bool busy = false;
    };
    std::mutex slot_mutex;
    std::vector<WorkerSlot> free_slots;
    int acquire_slot(size_t tiles, size_t per_tile, WorkerSlot& out);
    void release_slot(const WorkerSlot& s);
    static void destroy_slot(WorkerSlot& s);
    // Buffers of mp_render_frame_multi / mp_render_pass_multi.  As a rank: this device's shard (tile-major; between
    // MP_FLAG_ACCUMULATE passes it holds the running per-pixel state and never leaves the device), the rank's stream -- render and
    // the copy of the shard towards the gathering device are issued on it in order, so every rank's copy runs on its own stream
    // (its own xGMI link) and overlaps the others -- and the event recorded after the copy.  As the gathering context (ctxs[0]):
    // the rank-major gather buffer, the event after which it may be overwritten by the next frame's copies, and per peer device
    // whether direct peer access works (checked once) or the shard is staged through pinned host memory.
    struct MultiBuf {
        float* d_shard = nullptr;
        size_t shard_cap = 0;  // floats
        hipStream_t stream = nullptr;
        hipEvent_t copied = nullptr;
        float* h_stage = nullptr;  // pinned; only without peer access to the gathering device
        size_t stage_cap = 0;
        float* d_gather = nullptr;
        size_t gather_cap = 0;
        hipEvent_t untiled = nullptr;
        bool untiled_valid = false;
        std::vector<int> peer;  // by device id: 0 = not asked yet, 1 = direct peer copies, 2 = staged through the host
        // progressive accumulation: what the shard's running state belongs to, and the next sample it expects
        uint64_t acc_seed = 0;
        uint32_t acc_w = 0, acc_h = 0, acc_ts = 0, acc_spp = 0, acc_flags = 0, acc_depth = 0, acc_n = 0, acc_rank = 0, acc_next = 0;
        const mp_scene* acc_scene = nullptr;
    } multi;
    std::mutex multi_mu;
    static constexpr size_t kTileLists = 32;
    std::mutex tile_mutex;
    std::vector<TileListRef> tile_lists;
    uint64_t tile_clock = 0;
    // Returns device pointers and `keep`, a reference on the cache entry: the caller holds it until its launch is enqueued, so a
    // concurrent caller that evicts the entry (LRU, 32 lists) cannot free the buffer in between; the buffer is freed when the
    // last reference goes (hipFree then waits for the kernels already enqueued).
    int device_tiles(const mp_block* tiles, size_t n, const uint32_t* order, TileListRef& keep, const mp_block** d_tiles,
                     const uint32_t** d_order);
};

// default material of the build-defined path extension: grey 0.75, no emission, no texture
inline mp_material default_material() {
    mp_material m{};
    for (int c = 0; c < 3; c++) m.albedo[c] = m.albedo2[c] = 0.75f;
    return m;
}
// a table of grey (r = g = b), untextured materials runs the one-channel kernels (bit-identical to the three-channel ones)
inline bool table_is_grey(const std::vector<mp_material>& t) {
    for (const mp_material& m : t)
        if (m.texture != MP_TEXTURE_NONE || std::memcmp(&m.albedo[0], &m.albedo[1], 4) != 0 || std::memcmp(&m.albedo[0], &m.albedo[2], 4) != 0 ||
            std::memcmp(&m.emission[0], &m.emission[1], 4) != 0 || std::memcmp(&m.emission[0], &m.emission[2], 4) != 0)
            return false;
    return true;
}

struct mp_scene {
    mp_ctx* ctx = nullptr;
    HostBvh host;
    DevScene dev;
    void* d_shade = nullptr;
    void* d_vidx = nullptr;
    void* d_vtex = nullptr;
    void* d_nodes_aos = nullptr;  // wide tree
    void* d_nodes_lit = nullptr;  // literal tree
    uint32_t wide_nodes = 0, absorbed_nodes = 0;
    void* d_tris_aos = nullptr;
    void* d_pkt_valid = nullptr;
    // material table of the path extension: shared by reference with the instanced scenes made from this scene (each keeps the
    // table it was created with alive; mp_scene_set_materials gives a scene a new table of its own)
    struct DevTable {
        void* d = nullptr;
        int device = 0;
        ~DevTable() {
            if (!d) return;
            int prev = -1;
            (void)hipGetDevice(&prev);
            if (prev != device) (void)hipSetDevice(device);
            (void)hipFree(d);
            if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
        }
    };
    std::shared_ptr<DevTable> mat_table;
    void* d_inst = nullptr;               // object group (mp_scene_group / mp_scene_instances): DevObject per member
    const mp_scene* inst_of = nullptr;    // ... first member; the group borrows its members' device arrays
    std::vector<const mp_scene*> members;
    bool one_object = false;              // mp_scene_instances: every member is inst_of
    std::vector<float> inst_t;
    std::vector<mp_material> materials{default_material()};  // build-defined path extension defaults
    float sky = 1.0f;
    uint32_t material_count = 1;  // max TriangleShadingData.material + 1
    uint64_t device_bytes = 0;
    // An object group borrows its members' device arrays and host trees: every group holds one reference on each of its members,
    // so mp_scene_destroy on a member only drops the caller's reference and the arrays live until the last group has gone too.
    std::atomic<int> refs{1};
};

#ifndef MP_RENDER_SLOTS
#define MP_RENDER_SLOTS 3  // batches of tiles in flight per render() worker (measured on the metric's frame: 2 -> 50.4-51.1 ms, 3 -> 49.1-49.5, 4 -> 49.4-49.7; bare launch 48.8)
#endif

namespace {

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace

int mp_ctx::device_tiles(const mp_block* tiles, size_t n, const uint32_t* order, TileListRef& keep, const mp_block** d_tiles,
                         const uint32_t** d_order) {
    TileListRef evicted;  // released after the lock is dropped: its hipFree may wait for the device
    {
        std::lock_guard<std::mutex> lock(tile_mutex);
        const size_t tb = n * sizeof(mp_block), ob = order ? n * sizeof(uint32_t) : 0;
        TileListRef hit;
        for (TileListRef& e : tile_lists)
            if (e->tiles.size() == n && e->order.size() == (order ? n : 0) && std::memcmp(e->tiles.data(), tiles, tb) == 0 &&
                (!order || std::memcmp(e->order.data(), order, ob) == 0)) {
                hit = e;
                break;
            }
        if (!hit) {
            if (tile_lists.size() >= kTileLists) {  // evict the least recently used list
                size_t lru = 0;
                for (size_t i = 1; i < tile_lists.size(); i++)
                    if (tile_lists[i]->last_use < tile_lists[lru]->last_use) lru = i;
                evicted = std::move(tile_lists[lru]);
                tile_lists.erase(tile_lists.begin() + static_cast<std::ptrdiff_t>(lru));
            }
            auto e = std::make_shared<TileList>();
            e->device = device;
            e->tiles.assign(tiles, tiles + n);
            if (order) e->order.assign(order, order + n);
            MP_HIP(hipMalloc(reinterpret_cast<void**>(&e->d_buf), tb + ob));
            hipError_t err = hipMemcpy(e->d_buf, tiles, tb, hipMemcpyHostToDevice);
            if (err == hipSuccess && order) err = hipMemcpy(e->d_buf + tb, order, ob, hipMemcpyHostToDevice);
            if (err != hipSuccess) return hip_fail(err, "hipMemcpy(tile list)");
            tile_lists.push_back(e);
            hit = e;
        }
        hit->last_use = ++tile_clock;
        *d_tiles = reinterpret_cast<const mp_block*>(hit->d_buf);
        if (d_order) *d_order = order ? reinterpret_cast<const uint32_t*>(hit->d_buf + tb) : nullptr;
        keep = std::move(hit);
    }
    return MP_OK;
}

void mp_ctx::destroy_slot(WorkerSlot& s) {
    if (s.stream) (void)hipStreamSynchronize(s.stream);
    if (s.d_f32) (void)hipFree(s.d_f32);
    if (s.d_u8) (void)hipFree(s.d_u8);
    if (s.d_tiles) (void)hipFree(s.d_tiles);
    if (s.h_f32) (void)hipHostFree(s.h_f32);
    if (s.h_u8) (void)hipHostFree(s.h_u8);
    if (s.stream) (void)hipStreamDestroy(s.stream);
    s = WorkerSlot{};
}

int mp_ctx::acquire_slot(size_t tiles, size_t per_tile, WorkerSlot& out) {
    {
        std::lock_guard<std::mutex> lock(slot_mutex);
        for (size_t i = 0; i < free_slots.size(); i++)
            if (free_slots[i].tiles * free_slots[i].per_tile >= tiles * per_tile && free_slots[i].tiles >= tiles) {
                out = free_slots[i];
                free_slots.erase(free_slots.begin() + static_cast<std::ptrdiff_t>(i));
                out.busy = false;
                return MP_OK;
            }
    }
    WorkerSlot s;
    s.tiles = tiles;
    s.per_tile = per_tile;
    hipError_t e = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s.d_f32), tiles * per_tile * 4);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s.d_u8), tiles * per_tile);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s.d_tiles), tiles * sizeof(mp_block));
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&s.h_f32), tiles * per_tile * 4, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&s.h_u8), tiles * per_tile, hipHostMallocDefault);
    if (e != hipSuccess) {
        destroy_slot(s);
        return hip_fail(e, "render worker buffers");
    }
    out = s;
    return MP_OK;
}

void mp_ctx::release_slot(const WorkerSlot& s) {
    if (!s.stream) return;
    std::lock_guard<std::mutex> lock(slot_mutex);
    if (free_slots.size() < 8) {
        free_slots.push_back(s);
    } else {
        WorkerSlot d = s;
        destroy_slot(d);
    }
}

namespace {

int upload_scene(mp_scene* s) {
    const HostBvh& h = s->host;
    const size_t np = h.packets.size();
    std::vector<float> tris(np * kPacketDwords, 0.0f);
    std::vector<float> shade(np * 8 * 12, 0.0f);
    std::vector<uint32_t> vidx(np * 8 * 3, 0);
    for (size_t p = 0; p < np; p++) {
        const TriPacketRef& pk = h.packets[p];
        const Box3& e = h.packet_box[p];
        float size[3] = {e.mx[0] - e.mn[0], e.mx[1] - e.mn[1], e.mx[2] - e.mn[2]};
        float* o = &tris[p * kPacketDwords];
        for (int i = 0; i < 8; i++) {
            float v[3][3];
            for (int a = 0; a < 3; a++)
                for (int k = 0; k < 3; k++) v[a][k] = dequantise(pk.v[a][k][i], size[k], e.mn[k]);  // :160-162
            for (int k = 0; k < 3; k++) {
                o[k * 8 + i] = v[0][k];
                o[(3 + k) * 8 + i] = v[1][k] - v[0][k];  // e1, triangle.rs:195
                o[(6 + k) * 8 + i] = v[2][k] - v[0][k];  // e2, triangle.rs:196
            }
            const TriShadingRef& sh = h.shading[p * 8 + i];
            float* so = &shade[(p * 8 + i) * 12];
            for (int a = 0; a < 3; a++) {
                for (int k = 0; k < 3; k++) so[a * 3 + k] = h.vnormal[3 * static_cast<size_t>(sh.vi[a]) + k];
                vidx[(p * 8 + i) * 3 + a] = sh.vi[a];
            }
            uint32_t flat = sh.flat, mat = h.material.empty() ? 0u : h.material[p * 8 + i];
            std::memcpy(&so[9], &flat, 4);
            std::memcpy(&so[10], &mat, 4);
        }
    }
    // node arrays (device_tree.cpp): the wide tree the walks normally use and the literal reference tree for rays with an
    // infinite inverse direction component.  The sign-specialised slab test of the packet walk relies on min <= max for every
    // real child box of either.
    const std::vector<uint32_t> pkt_valid = packet_real_counts(h);
    DeviceTree wide, lit;
    {
        std::string err;
        int trc = build_device_tree(h, pkt_valid, true, wide, err);
        if (!trc) trc = build_device_tree(h, pkt_valid, false, lit, err);
        if (trc) return fail(trc, err);
    }
    s->dev.boxes_ordered = (wide.boxes_ordered && lit.boxes_ordered) ? 1u : 0u;
    s->wide_nodes = wide.count;
    s->absorbed_nodes = wide.absorbed;
    std::vector<float> tris_aos(np * 8 * kTriDwords + 4 * kTriDwords, 0.0f);  // + tail padding: the triangle loop prefetches up to two ahead
    for (size_t p = 0; p < np; p++)
        for (int i = 0; i < 8; i++)
            for (int k = 0; k < 9; k++) tris_aos[(p * 8 + i) * kTriDwords + k] = tris[p * kPacketDwords + k * 8 + i];
    // magnitude bound behind the packet walk's triangle masks (kernels.hip, tri_may_hit): no overflow in the interval evaluation
    s->dev.tris_bounded = 1u;
    for (size_t i = 0; i < np * 8 && s->dev.tris_bounded; i++)
        for (int k = 0; k < 9; k++) {
            const float v = tris_aos[i * kTriDwords + k];
            if (!(std::fabs(v) <= (k < 3 ? 1073741824.0f : 2147483648.0f))) { s->dev.tris_bounded = 0u; break; }
        }
    auto up = [&](void** dst, const void* src, size_t bytes) -> int {
        bytes = std::max<size_t>(bytes, 16);
        MP_HIP(hipMalloc(dst, bytes));
        MP_HIP(hipMemset(*dst, 0, bytes));
        s->device_bytes += bytes;
        return MP_OK;
    };
    int rc;
    if ((rc = up(&s->d_shade, shade.data(), shade.size() * 4))) return rc;
    if ((rc = up(&s->d_vidx, vidx.data(), vidx.size() * 4))) return rc;
    if ((rc = up(&s->d_vtex, h.vtex.data(), h.vtex.size() * 4))) return rc;
    if ((rc = up(&s->d_nodes_aos, wide.nodes.data(), wide.nodes.size() * 4))) return rc;
    if ((rc = up(&s->d_nodes_lit, lit.nodes.data(), lit.nodes.size() * 4))) return rc;
    if ((rc = up(&s->d_tris_aos, tris_aos.data(), tris_aos.size() * 4))) return rc;
    if ((rc = up(&s->d_pkt_valid, pkt_valid.data(), pkt_valid.size() * 4))) return rc;
    {
        auto tb = std::make_shared<mp_scene::DevTable>();
        tb->device = s->ctx->device;
        if ((rc = up(&tb->d, s->materials.data(), s->materials.size() * sizeof(mp_material)))) return rc;
        MP_HIP(hipMemcpy(tb->d, s->materials.data(), s->materials.size() * sizeof(mp_material), hipMemcpyHostToDevice));
        s->mat_table = std::move(tb);
    }
    s->dev.materials = static_cast<const float*>(s->mat_table->d);
    s->dev.sky = s->sky;
    MP_HIP(hipMemcpy(s->d_nodes_aos, wide.nodes.data(), wide.nodes.size() * 4, hipMemcpyHostToDevice));
    MP_HIP(hipMemcpy(s->d_nodes_lit, lit.nodes.data(), lit.nodes.size() * 4, hipMemcpyHostToDevice));
    if (!tris_aos.empty()) MP_HIP(hipMemcpy(s->d_tris_aos, tris_aos.data(), tris_aos.size() * 4, hipMemcpyHostToDevice));
    if (!pkt_valid.empty()) MP_HIP(hipMemcpy(s->d_pkt_valid, pkt_valid.data(), pkt_valid.size() * 4, hipMemcpyHostToDevice));
    if (!shade.empty()) MP_HIP(hipMemcpy(s->d_shade, shade.data(), shade.size() * 4, hipMemcpyHostToDevice));
    if (!vidx.empty()) MP_HIP(hipMemcpy(s->d_vidx, vidx.data(), vidx.size() * 4, hipMemcpyHostToDevice));
    if (!h.vtex.empty()) MP_HIP(hipMemcpy(s->d_vtex, h.vtex.data(), h.vtex.size() * 4, hipMemcpyHostToDevice));
    s->dev.shade = static_cast<const float*>(s->d_shade);
    s->dev.vidx = static_cast<const uint32_t*>(s->d_vidx);
    s->dev.vtex = static_cast<const float*>(s->d_vtex);
    s->dev.nodes_aos = static_cast<const float*>(s->d_nodes_aos);
    s->dev.nodes_lit = static_cast<const float*>(s->d_nodes_lit);
    s->dev.tris_aos = static_cast<const float*>(s->d_tris_aos);
    s->dev.pkt_valid = static_cast<const uint32_t*>(s->d_pkt_valid);
    s->dev.root = wide.root;
    s->dev.root_lit = lit.root;
    s->dev.inner_count = wide.count;
    s->dev.packet_count = static_cast<uint32_t>(np);
    // exact bound of the traversal stack (device_tree.cpp), over both trees: the walks share their stack storage
    s->dev.stack_cap = std::max(wide.stack_bound, lit.stack_bound);
    // union of the literal root's child boxes (the wide root's slots lie inside it: absorbed children are FP-nested)
    s->dev.has_pre = 0;
    if ((h.root & 7u) == 0u && h.root != MP_LINK_NULL) {  // root is an inner node
        const float* o = &lit.nodes[static_cast<size_t>(h.root >> 3) * 64];
        bool any = false;
        for (int i = 0; i < 8; i++) {
            if (h.inner[h.root >> 3].link[i] == MP_LINK_NULL) continue;
            for (int k = 0; k < 3; k++) {
                float mn = o[i * 8 + k], mx = o[i * 8 + 3 + k];
                s->dev.pre_min[k] = any ? std::fmin(s->dev.pre_min[k], mn) : mn;
                s->dev.pre_max[k] = any ? std::fmax(s->dev.pre_max[k], mx) : mx;
            }
            any = true;
        }
        s->dev.has_pre = any ? 1u : 0u;
    }
    return MP_OK;
}

int finish_scene(mp_ctx* ctx, std::unique_ptr<mp_scene> s, mp_scene** out) {
    s->ctx = ctx;
    uint32_t mc = 0;
    for (uint32_t m : s->host.material) mc = std::max(mc, m);
    if (mc >= MP_MAX_MATERIALS) return fail(MP_ERR_INVALID, "material id out of range (MP_MAX_MATERIALS)");  // build_bvh / bvh_from_arrays reject these already
    s->material_count = mc + 1;
    s->materials.assign(s->material_count, default_material());
    if (!ctx) {  // host-only scene: build + export work, nothing is uploaded and nothing can be rendered
        *out = s.release();
        return MP_OK;
    }
    DeviceGuard g(ctx->device);
    if (!g.ok) return fail(MP_ERR_HIP, "hipSetDevice failed");
    int rc = upload_scene(s.get());
    if (rc) {
        mp_scene_destroy(s.release());
        return rc;
    }
    *out = s.release();
    return MP_OK;
}

bool valid_settings(const mp_settings* st) {
    if (!(st && st->tile_size > 0 && st->sample_count > 0 && st->width > 0 && st->height > 0 &&
          (!(st->flags & MP_FLAG_PATHS) || st->max_depth >= 1)))
        return false;
    if ((st->flags & MP_FLAG_WAVEFRONT) && !(st->flags & MP_FLAG_PATHS)) return false;
    if (!(st->flags & MP_FLAG_ACCUMULATE)) return st->pass_begin == 0 && st->pass_count == 0;
    return st->pass_begin < st->sample_count && st->pass_count <= st->sample_count - st->pass_begin;
}
// samples per pixel drawn by one launch
uint32_t pass_samples(const mp_settings& st) {
    if (!(st.flags & MP_FLAG_ACCUMULATE)) return st.sample_count;
    return st.pass_count ? st.pass_count : st.sample_count - st.pass_begin;
}

// Renders `tiles` into a tile-major device buffer (launch only).
int render_tiles_device(mp_ctx* ctx, const mp_scene* scene, const mp_camera_sampler& sampler, const mp_settings& st,
                        const mp_block* d_tiles, size_t n, float* d_out, void* stream, uint64_t* d_segments = nullptr,
                        const uint32_t* d_tile_order = nullptr, uint64_t* d_tile_cost = nullptr) {
    if ((st.flags & MP_FLAG_PATHS) && (st.flags & MP_FLAG_CHUNKED_SUM) && scene->dev.materials_rgb)
        return fail(MP_ERR_UNSUPPORTED, "coloured / textured materials are not combined with MP_FLAG_CHUNKED_SUM (its pixel state carries one channel)");
    RenderLaunch L;
    L.scene = scene->dev;
    L.scene.packet_stack_regs = ctx->packet_stack_regs.load();
    L.packet_samples = ctx->packet_samples.load();
    L.rays_per_lane = ctx->rays_per_lane.load();
    L.paths_pooled = ctx->paths_pooled.load();
    L.mask_cache = ctx->mask_cache.load();
    L.sampler = sampler;
    L.width = st.width;
    L.height = st.height;
    L.spp = st.sample_count;
    L.tile_size = st.tile_size;
    L.seed = st.seed;
    L.d_tiles = d_tiles;
    L.n_tiles = static_cast<uint32_t>(n);
    L.d_out = d_out;
    L.d_counter = ctx->take_counter();
    {   // diagnostic knob: fewer resident workgroups per CU (the launchers size their grids as cu_count x blocks that fit)
        const uint32_t bpc = ctx->blocks_per_cu.load();
        L.cu_count = bpc ? std::max(1, ctx->cu_count * static_cast<int>(bpc) / 8) : ctx->cu_count;
    }
    L.traversal = (st.flags & MP_FLAG_TRAVERSAL_GROUPS) ? 1 : 0;
    L.max_depth = (st.flags & MP_FLAG_PATHS) ? st.max_depth : 0u;
    L.d_segments = reinterpret_cast<unsigned long long*>(d_segments);
    L.d_tile_order = d_tile_order;
    L.d_tile_cost = reinterpret_cast<unsigned long long*>(d_tile_cost);
    L.pass_begin = (st.flags & MP_FLAG_ACCUMULATE) ? st.pass_begin : 0u;
    L.pass_end = L.pass_begin + pass_samples(st);
    L.carry_in = L.pass_begin > 0;
    L.finalize = L.pass_end == st.sample_count;
    L.chunked = (st.flags & MP_FLAG_CHUNKED_SUM) != 0;
    std::string err;
    int rc = (st.flags & MP_FLAG_WAVEFRONT) ? launch_render_paths_wavefront(L, stream, err) : launch_render_tiles(L, stream, err);
    if (rc) return fail(rc, err);
    return MP_OK;
}

}  // namespace

extern "C" {

const char* mp_last_error(void) { return g_last_error.c_str(); }
const char* mp_version(void) { return "minipath_hip 0.2 (gfx950)"; }
const char* mp_scene_material_name(const mp_scene* scene, uint32_t id) {
    if (!scene || id >= scene->material_count) return nullptr;
    return id < scene->host.material_names.size() ? scene->host.material_names[id].c_str() : "";
}

int mp_ctx_create(int device_id, mp_ctx** out) {
    return guarded([&]() -> int {
    if (!out) return fail(MP_ERR_INVALID, "out is NULL");
    int count = 0;
    MP_HIP(hipGetDeviceCount(&count));
    if (device_id < 0 || device_id >= count) return fail(MP_ERR_INVALID, "device_id out of range");
    DeviceGuard g(device_id);
    if (!g.ok) return fail(MP_ERR_HIP, "hipSetDevice failed");
    hipDeviceProp_t prop;
    MP_HIP(hipGetDeviceProperties(&prop, device_id));
    auto ctx = std::make_unique<mp_ctx>();
    ctx->device = device_id;
    ctx->cu_count = prop.multiProcessorCount;
    const size_t counter_bytes = static_cast<size_t>(mp_ctx::kCounters) * kWorkQueues * kWorkQueueStride * sizeof(uint32_t);
    MP_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->d_counters), counter_bytes));
    MP_HIP(hipMemset(ctx->d_counters, 0, counter_bytes));
    *out = ctx.release();
    return MP_OK;
    });
}

void mp_ctx_destroy(mp_ctx* ctx) {
    if (!ctx) return;
    DeviceGuard g(ctx->device);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    ctx->tile_lists.clear();
    for (auto& sl : ctx->free_slots) mp_ctx::destroy_slot(sl);
    if (ctx->multi.stream) (void)hipStreamSynchronize(ctx->multi.stream);
    if (ctx->multi.d_shard) (void)hipFree(ctx->multi.d_shard);
    if (ctx->multi.d_gather) (void)hipFree(ctx->multi.d_gather);
    if (ctx->multi.h_stage) (void)hipHostFree(ctx->multi.h_stage);
    if (ctx->multi.copied) (void)hipEventDestroy(ctx->multi.copied);
    if (ctx->multi.untiled) (void)hipEventDestroy(ctx->multi.untiled);
    if (ctx->multi.stream) (void)hipStreamDestroy(ctx->multi.stream);
    delete ctx;
}

int mp_ctx_set_option(mp_ctx* ctx, const char* key, int value) {
    return guarded([&]() -> int {
    if (!ctx || !key) return fail(MP_ERR_INVALID, "NULL argument");
    if (std::strcmp(key, "packet_stack_registers") == 0) {
        if (value < 1 || value > 64) return fail(MP_ERR_INVALID, "packet_stack_registers must be in 1..64");
        ctx->packet_stack_regs.store(static_cast<uint32_t>(value));
        return MP_OK;
    }
    if (std::strcmp(key, "packet_mask_cache") == 0) {
        if (value < 0 || value > 2) return fail(MP_ERR_INVALID, "packet_mask_cache must be 0 (off), 1 or 2 (on)");
        ctx->mask_cache.store(static_cast<uint32_t>(value));
        return MP_OK;
    }
    if (std::strcmp(key, "paths_pooled") == 0) {
        if (value < 0 || value > 3) return fail(MP_ERR_INVALID, "paths_pooled must be 0 (one pass per walk), 1 (auto), 2 (always, two passes) or 3 (always, up to four passes)");
        ctx->paths_pooled.store(static_cast<uint32_t>(value));
        return MP_OK;
    }
    if (std::strcmp(key, "packet_rays_per_lane") == 0) {
        if (value != 1 && value != 2) return fail(MP_ERR_INVALID, "packet_rays_per_lane must be 1 or 2");
        ctx->rays_per_lane.store(static_cast<uint32_t>(value));
        return MP_OK;
    }
    if (std::strcmp(key, "render_batch_tiles") == 0) {
        if (value < 0 || value > 65536) return fail(MP_ERR_INVALID, "render_batch_tiles must be in 0..65536 (0 = automatic)");
        ctx->render_batch.store(static_cast<uint32_t>(value));
        return MP_OK;
    }
    if (std::strcmp(key, "blocks_per_cu") == 0) {
        if (value < 0 || value > 8) return fail(MP_ERR_INVALID, "blocks_per_cu must be in 0..8 (0 = as many as fit)");
        ctx->blocks_per_cu.store(static_cast<uint32_t>(value));
        return MP_OK;
    }
    if (std::strcmp(key, "packet_samples_in_flight") == 0) {
        if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8 && value != 16 && value != 32 && value != 64)
            return fail(MP_ERR_INVALID, "packet_samples_in_flight must be 0 (automatic) or a power of two up to 64");
        ctx->packet_samples.store(static_cast<uint32_t>(value));
        return MP_OK;
    }
    return fail(MP_ERR_INVALID, std::string("unknown option: ") + key);
    });
}

int mp_ctx_device(const mp_ctx* ctx, int* device_id, int* cu_count) {
    return guarded([&]() -> int {
    if (!ctx) return fail(MP_ERR_INVALID, "ctx is NULL");
    if (device_id) *device_id = ctx->device;
    if (cu_count) *cu_count = ctx->cu_count;
    return MP_OK;
    });
}

// ---- camera ---------------------------------------------------------------------------------------------------
int mp_camera_default(mp_camera* cam) {
    return guarded([&]() -> int {
    if (!cam) return fail(MP_ERR_INVALID, "cam is NULL");
    camera_default(*cam);
    return MP_OK;
    });
}
int mp_camera_look_at(mp_camera* cam, const float eye[3], const float at[3], const float up[3]) {
    return guarded([&]() -> int {
    if (!cam || !eye || !at || !up) return fail(MP_ERR_INVALID, "NULL argument");
    camera_look_at(*cam, eye, at, up);
    return MP_OK;
    });
}
int mp_camera_look_direction(mp_camera* cam, const float eye[3], const float fwd[3], const float up[3]) {
    return guarded([&]() -> int {
    if (!cam || !eye || !fwd || !up) return fail(MP_ERR_INVALID, "NULL argument");
    camera_look_direction(*cam, eye, fwd, up);
    return MP_OK;
    });
}
int mp_camera_translate(mp_camera* cam, const float t[3]) {
    return guarded([&]() -> int {
    if (!cam || !t) return fail(MP_ERR_INVALID, "NULL argument");
    for (int k = 0; k < 3; k++) cam->t[k] = t[k] + cam->t[k];  // (Translation3 * Isometry3).translation
    return MP_OK;
    });
}
int mp_camera_basis(const mp_camera* cam, float center[3], float fwd[3], float up[3], float right[3]) {
    return guarded([&]() -> int {
    if (!cam || !center || !fwd || !up || !right) return fail(MP_ERR_INVALID, "NULL argument");
    camera_basis(*cam, center, fwd, up, right);
    return MP_OK;
    });
}
int mp_camera_build_sampler(const mp_camera* cam, uint32_t width, uint32_t height, mp_camera_sampler* out) {
    return guarded([&]() -> int {
    if (!cam || !out) return fail(MP_ERR_INVALID, "NULL argument");
    if (width == 0 || height == 0) return fail(MP_ERR_INVALID, "empty resolution");
    camera_build_sampler(*cam, width, height, *out);
    return MP_OK;
    });
}

int mp_tile_ordering(mp_block block, uint32_t tile_size, uint64_t shuffle_seed, mp_block* out, size_t cap, size_t* n) {
    return guarded([&]() -> int {
    if (tile_size == 0) return fail(MP_ERR_INVALID, "tile_size must be non-zero (NonZeroU32)");
    if (!n) return fail(MP_ERR_INVALID, "n is NULL");
    std::vector<mp_block> t = tile_ordering(block, tile_size, shuffle_seed);
    *n = t.size();
    if (out && !t.empty() && cap) std::memcpy(out, t.data(), std::min(cap, t.size()) * sizeof(mp_block));
    return MP_OK;
    });
}

// ---- scene ----------------------------------------------------------------------------------------------------
int mp_scene_from_obj(mp_ctx* ctx, const char* path, mp_scene** out) {
    return guarded([&]() -> int {
    if (!path || !out) return fail(MP_ERR_INVALID, "NULL argument");
    std::vector<float> pos, nrm, tex;
    std::vector<uint32_t> tri, tri_mat;
    std::vector<std::string> names;
    std::string err;
    int rc = load_obj(path, pos, nrm, tex, tri, tri_mat, names, err);
    if (rc) return fail(rc, err);
    auto s = std::make_unique<mp_scene>();
    rc = build_bvh(pos.data(), nrm.data(), tex.data(), static_cast<uint32_t>(pos.size() / 3), tri.data(), tri_mat.data(),
                   static_cast<uint32_t>(tri.size() / 3), s->host, err);
    if (rc) return fail(rc, err);
    s->host.material_names = std::move(names);
    return finish_scene(ctx, std::move(s), out);
    });
}

int mp_scene_from_triangles(mp_ctx* ctx, const float* positions, const float* normals, const float* tex,
                            uint32_t vertex_count, const uint32_t* indices, uint32_t triangle_count, mp_scene** out) {
    return mp_scene_from_triangles_mat(ctx, positions, normals, tex, vertex_count, indices, nullptr, triangle_count, out);
}

int mp_scene_from_triangles_mat(mp_ctx* ctx, const float* positions, const float* normals, const float* tex,
                                uint32_t vertex_count, const uint32_t* indices, const uint32_t* tri_material,
                                uint32_t triangle_count, mp_scene** out) {
    return guarded([&]() -> int {
    if (!positions || !indices || !out) return fail(MP_ERR_INVALID, "NULL argument");
    auto s = std::make_unique<mp_scene>();
    std::string err;
    int rc = build_bvh(positions, normals, tex, vertex_count, indices, tri_material, triangle_count, s->host, err);
    if (rc) return fail(rc, err);
    return finish_scene(ctx, std::move(s), out);
    });
}

int mp_scene_from_arrays(mp_ctx* ctx, const mp_bvh_desc* desc, mp_scene** out) {
    return guarded([&]() -> int {
    if (!desc || !out) return fail(MP_ERR_INVALID, "NULL argument");
    auto s = std::make_unique<mp_scene>();
    std::string err;
    int rc = bvh_from_arrays(*desc, s->host, err);
    if (rc) return fail(rc, err);
    return finish_scene(ctx, std::move(s), out);
    });
}

namespace {

// UnitQuaternion * Vector3, the operation order of the oracle and of the device code (kernels.hip: quat_rotate)
void quat_rotate_host(const float q[4], const float v[3], float out[3]) {
    const float tx = (q[1] * v[2] - q[2] * v[1]) * 2.0f, ty = (q[2] * v[0] - q[0] * v[2]) * 2.0f, tz = (q[0] * v[1] - q[1] * v[0]) * 2.0f;
    const float cx = q[1] * tz - q[2] * ty, cy = q[2] * tx - q[0] * tz, cz = q[0] * ty - q[1] * tx;
    out[0] = tx * q[3] + cx + v[0];
    out[1] = ty * q[3] + cy + v[1];
    out[2] = tz * q[3] + cz + v[2];
}

// {object, rigid transform} x n behind one Object (include/minipath_hip.h: mp_scene_group).  The group borrows its members' device
// arrays; it owns the descriptor array and its own material table.
int make_group(mp_ctx* ctx, const mp_scene* const* objects, const float* rotations, const float* translations, uint32_t n,
               bool one_object, mp_scene** out) {
    if (!objects || !translations || !out || n == 0) return fail(MP_ERR_INVALID, "bad argument");
    if (n > (1u << 20)) return fail(MP_ERR_INVALID, "too many members");
    for (uint32_t i = 0; i < n; i++) {
        const mp_scene* o = objects[i];
        if (!o) return fail(MP_ERR_INVALID, "NULL member");
        if (o->inst_of) return fail(MP_ERR_UNSUPPORTED, "the members of an object group are TriangleBvh or Sphere scenes, not groups");
        if (one_object && o->dev.kind != 0u) return fail(MP_ERR_UNSUPPORTED, "instances are made of a TriangleBvh scene");
        if (o->ctx != ctx) return fail(MP_ERR_INVALID, "member belongs to another context");
    }
    auto s = std::make_unique<mp_scene>();
    s->ctx = ctx;
    s->inst_of = objects[0];
    s->one_object = one_object;
    s->members.assign(objects, objects + n);
    s->inst_t.assign(translations, translations + static_cast<size_t>(n) * 3);
    s->host.root = objects[0]->host.root;
    s->host.material_names = objects[0]->host.material_names;
    s->material_count = 1;
    s->dev = DevScene{};  // kind 0; every geometry field comes from the member descriptors
    s->dev.packet_stack_regs = objects[0]->dev.packet_stack_regs;
    s->dev.materials = objects[0]->dev.materials;
    s->dev.sky = objects[0]->sky;
    for (uint32_t i = 0; i < n; i++) {
        const mp_scene* o = objects[i];
        s->host.depth = std::max(s->host.depth, o->host.depth);
        if (!one_object || i == 0) {  // instances of one object report the object's own counts
            s->host.vertex_count += o->host.vertex_count;
            s->host.triangle_count += o->host.triangle_count;
        }
        s->material_count = std::max(s->material_count, o->material_count);
        s->dev.stack_cap = std::max(s->dev.stack_cap, o->dev.stack_cap);
    }
    // get_bounding_box: union of the members' boxes in the world frame (a rotated member: the box of its box's eight rotated corners)
    for (uint32_t i = 0; i < n; i++) {
        const Box3& ob = objects[i]->host.bbox;
        float lo[3], hi[3];
        if (rotations) {
            const float* q = rotations + 4 * static_cast<size_t>(i);
            for (int c = 0; c < 8; c++) {
                const float v[3] = {(c & 1) ? ob.mx[0] : ob.mn[0], (c & 2) ? ob.mx[1] : ob.mn[1], (c & 4) ? ob.mx[2] : ob.mn[2]};
                float w[3];
                quat_rotate_host(q, v, w);
                for (int k = 0; k < 3; k++) {
                    lo[k] = c ? std::fmin(lo[k], w[k]) : w[k];
                    hi[k] = c ? std::fmax(hi[k], w[k]) : w[k];
                }
            }
        } else {
            for (int k = 0; k < 3; k++) { lo[k] = ob.mn[k]; hi[k] = ob.mx[k]; }
        }
        for (int k = 0; k < 3; k++) {
            const float a = lo[k] + translations[3 * i + k], b = hi[k] + translations[3 * i + k];
            s->host.bbox.mn[k] = i ? std::fmin(s->host.bbox.mn[k], a) : a;
            s->host.bbox.mx[k] = i ? std::fmax(s->host.bbox.mx[k], b) : b;
        }
    }
    // material table of the group: the first member's, padded with the default material up to the largest id any member uses
    s->materials = objects[0]->materials;
    if (s->materials.size() < s->material_count) s->materials.resize(s->material_count, default_material());
    s->sky = objects[0]->sky;
    s->dev.materials_rgb = table_is_grey(s->materials) ? 0u : 1u;
    s->dev.inst_count = n;
    s->dev.has_pre = 0;
    if (ctx) {
        DeviceGuard g(ctx->device);
        std::vector<DevObject> desc(n);
        for (uint32_t i = 0; i < n; i++) {
            const DevScene& d = objects[i]->dev;
            DevObject& o = desc[i];
            std::memset(&o, 0, sizeof(o));
            o.shade = d.shade; o.nodes_aos = d.nodes_aos; o.nodes_lit = d.nodes_lit; o.tris_aos = d.tris_aos; o.vidx = d.vidx; o.vtex = d.vtex;
            o.root = d.root; o.root_lit = d.root_lit; o.has_pre = d.has_pre;
            o.kind = d.kind; o.sphere_radius = d.sphere_radius;
            for (int k = 0; k < 3; k++) {
                o.pre_min[k] = d.pre_min[k]; o.pre_max[k] = d.pre_max[k]; o.t[k] = translations[3 * i + k];
                o.sphere_center[k] = d.sphere_center[k];
            }
            if (rotations) {
                for (int k = 0; k < 4; k++) o.q[k] = rotations[4 * static_cast<size_t>(i) + k];
                o.rotated = 1u;
            }
        }
        MP_HIP(hipMalloc(&s->d_inst, desc.size() * sizeof(DevObject)));
        hipError_t e = hipMemcpy(s->d_inst, desc.data(), desc.size() * sizeof(DevObject), hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(s->d_inst); return hip_fail(e, "hipMemcpy(object group)"); }
        s->dev.objects = static_cast<const DevObject*>(s->d_inst);
        s->device_bytes = desc.size() * sizeof(DevObject);
        {  // the group's own copy of the table: a later mp_scene_set_materials on a member does not reach the group
            auto tb = std::make_shared<mp_scene::DevTable>();  // (its destructor frees the table on the error paths below)
            tb->device = ctx->device;
            e = hipMalloc(&tb->d, std::max<size_t>(16, s->materials.size() * sizeof(mp_material)));
            if (e == hipSuccess) e = hipMemcpy(tb->d, s->materials.data(), s->materials.size() * sizeof(mp_material), hipMemcpyHostToDevice);
            if (e != hipSuccess) { (void)hipFree(s->d_inst); return hip_fail(e, "material table of the object group"); }
            s->mat_table = std::move(tb);
            s->dev.materials = static_cast<const float*>(s->mat_table->d);
        }
    }
    // the group is complete: one reference per member it borrows from (instances of one object: one in all)
    for (size_t i = 0; i < (one_object ? 1 : static_cast<size_t>(n)); i++) const_cast<mp_scene*>(objects[i])->refs.fetch_add(1, std::memory_order_relaxed);
    *out = s.release();
    return MP_OK;
}

}  // namespace

int mp_scene_group(mp_ctx* ctx, const mp_scene* const* objects, const float* rotations, const float* translations, uint32_t n,
                   mp_scene** out) {
    return guarded([&]() -> int { return make_group(ctx, objects, rotations, translations, n, false, out); });
}

int mp_scene_instances(mp_ctx* ctx, const mp_scene* object, const float* translations, uint32_t n, mp_scene** out) {
    return guarded([&]() -> int {
    if (!object || n == 0 || n > (1u << 20)) return fail(MP_ERR_INVALID, "bad argument");
    std::vector<const mp_scene*> objs(n, object);
    return make_group(ctx, objs.data(), nullptr, translations, n, true, out);
    });
}

int mp_scene_set_materials(mp_scene* scene, const mp_material* table, uint32_t n, float sky_radiance) {
    return guarded([&]() -> int {
    if (!scene || !table) return fail(MP_ERR_INVALID, "NULL argument");
    if (scene->dev.kind != 0u) return fail(MP_ERR_UNSUPPORTED, "materials belong to TriangleBvh scenes");
    if (n < scene->material_count) return fail(MP_ERR_INVALID, "material table shorter than the scene's material_count");
    for (uint32_t i = 0; i < n; i++)
        if (table[i].texture != MP_TEXTURE_NONE && table[i].texture != MP_TEXTURE_CHECKER) return fail(MP_ERR_INVALID, "unknown texture kind in the material table");
    scene->materials.assign(table, table + n);
    scene->sky = sky_radiance;
    scene->dev.materials_rgb = table_is_grey(scene->materials) ? 0u : 1u;
    if (scene->ctx) {  // a new table for this scene; instanced scenes made earlier keep the one they were created with
        DeviceGuard g(scene->ctx->device);
        auto tb = std::make_shared<mp_scene::DevTable>();
        tb->device = scene->ctx->device;
        MP_HIP(hipMalloc(&tb->d, std::max<size_t>(16, n * sizeof(mp_material))));
        MP_HIP(hipMemcpy(tb->d, scene->materials.data(), n * sizeof(mp_material), hipMemcpyHostToDevice));
        scene->mat_table = std::move(tb);  // the old table is freed when its last holder lets go (hipFree waits for the device)
        scene->dev.materials = static_cast<const float*>(scene->mat_table->d);
        scene->dev.sky = sky_radiance;
    }
    return MP_OK;
    });
}

int mp_scene_sphere(mp_ctx* ctx, const float center[3], float radius, mp_scene** out) {
    return guarded([&]() -> int {
    if (!center || !out) return fail(MP_ERR_INVALID, "NULL argument");
    auto s = std::make_unique<mp_scene>();
    s->ctx = ctx;
    s->dev.kind = 1;
    for (int k = 0; k < 3; k++) {
        s->dev.sphere_center[k] = center[k];
        s->host.bbox.mn[k] = center[k] - radius;  // get_bounding_box, primitives.rs:50-56
        s->host.bbox.mx[k] = center[k] + radius;
    }
    s->dev.sphere_radius = radius;
    s->dev.stack_cap = 1;
    *out = s.release();
    return MP_OK;
    });
}

namespace {
void scene_release(mp_scene* s) {
    if (!s || s->refs.fetch_sub(1, std::memory_order_acq_rel) != 1) return;
    if (s->inst_of) {  // an object group owns its descriptor array (and, if re-set, its material table) and a reference per member
        if (s->ctx) {
            DeviceGuard g(s->ctx->device);
            if (s->d_inst) (void)hipFree(s->d_inst);
        }
        const size_t held = s->one_object ? 1 : s->members.size();
        for (size_t i = 0; i < held; i++) scene_release(const_cast<mp_scene*>(s->members[i]));
        delete s;
        return;
    }
    if (s->ctx) {
        DeviceGuard g(s->ctx->device);
        for (void* p : {s->d_shade, s->d_vidx, s->d_vtex, s->d_nodes_aos, s->d_nodes_lit, s->d_tris_aos, s->d_pkt_valid})
            if (p) (void)hipFree(p);
    }
    delete s;
}
}  // namespace

void mp_scene_destroy(mp_scene* s) { scene_release(s); }

int mp_scene_info_get(const mp_scene* s, mp_scene_info* out) {
    return guarded([&]() -> int {
    if (!s || !out) return fail(MP_ERR_INVALID, "NULL argument");
    out->root_link = s->host.root;
    out->inner_count = static_cast<uint32_t>(s->host.inner.size());
    out->packet_count = static_cast<uint32_t>(s->host.packets.size());
    if (s->inst_of) {  // object group: sums over the members (instances of one object: the object's own counts)
        uint64_t ni = 0, np = 0;
        for (size_t i = 0; i < (s->one_object ? 1 : s->members.size()); i++) {
            ni += s->members[i]->host.inner.size();
            np += s->members[i]->host.packets.size();
        }
        out->inner_count = static_cast<uint32_t>(ni);
        out->packet_count = static_cast<uint32_t>(np);
    }
    out->vertex_count = s->host.vertex_count;
    out->triangle_count = s->host.triangle_count;
    out->depth = s->host.depth;
    out->stack_bound = s->ctx ? s->dev.stack_cap : 0;
    for (int k = 0; k < 3; k++) {
        out->bbox_min[k] = s->host.bbox.mn[k];
        out->bbox_max[k] = s->host.bbox.mx[k];
    }
    out->device_bytes = s->device_bytes;
    out->material_count = s->material_count;
    out->reserved = 0;
    return MP_OK;
    });
}

int mp_scene_export(const mp_scene* s, void* inner_nodes, void* packets, void* tri_shading, float* vertex_normals,
                    float* vertex_tex, uint32_t* tri_material) {
    return guarded([&]() -> int {
    if (!s) return fail(MP_ERR_INVALID, "scene is NULL");
    if (s->inst_of && !s->one_object) return fail(MP_ERR_UNSUPPORTED, "an object group has no arrays of its own: export its members");
    const HostBvh& h = s->inst_of ? s->inst_of->host : s->host;
    if (inner_nodes && !h.inner.empty()) std::memcpy(inner_nodes, h.inner.data(), h.inner.size() * sizeof(InnerNodeRef));
    if (packets && !h.packets.empty()) std::memcpy(packets, h.packets.data(), h.packets.size() * sizeof(TriPacketRef));
    if (tri_shading && !h.shading.empty()) std::memcpy(tri_shading, h.shading.data(), h.shading.size() * sizeof(TriShadingRef));
    if (vertex_normals && !h.vnormal.empty()) std::memcpy(vertex_normals, h.vnormal.data(), h.vnormal.size() * 4);
    if (vertex_tex && !h.vtex.empty()) std::memcpy(vertex_tex, h.vtex.data(), h.vtex.size() * 4);
    if (tri_material && !h.material.empty()) std::memcpy(tri_material, h.material.data(), h.material.size() * 4);
    return MP_OK;
    });
}

int mp_scene_device_tree(const mp_scene* s, int which, float* nodes, uint32_t* count, uint32_t* root_dlink, uint32_t* stack_bound,
                         uint32_t* absorbed) {
    return guarded([&]() -> int {
    if (!s) return fail(MP_ERR_INVALID, "scene is NULL");
    if (which != 0 && which != 1) return fail(MP_ERR_INVALID, "which must be 0 (wide tree) or 1 (literal tree)");
    if (s->dev.kind != 0u || (s->inst_of && !s->one_object)) return fail(MP_ERR_UNSUPPORTED, "only a TriangleBvh scene has a node tree of its own");
    const HostBvh& h = s->inst_of ? s->inst_of->host : s->host;
    DeviceTree t;
    std::string err;
    const int rc = build_device_tree(h, packet_real_counts(h), which == 0, t, err);  // rebuilt from the host tree: same code, same floats as the upload
    if (rc) return fail(rc, err);
    if (nodes && t.count) std::memcpy(nodes, t.nodes.data(), static_cast<size_t>(t.count) * 64 * sizeof(float));
    if (count) *count = t.count;
    if (root_dlink) *root_dlink = t.root;
    if (stack_bound) *stack_bound = t.stack_bound;
    if (absorbed) *absorbed = t.absorbed;
    return MP_OK;
    });
}

// ---- rays -------------------------------------------------------------------------------------------------------
int mp_trace_rays(mp_ctx* ctx, const mp_scene* scene, const float* d_ox, const float* d_oy, const float* d_oz,
                  const float* d_dx, const float* d_dy, const float* d_dz, uint64_t n, const mp_hits_soa* hits,
                  void* stream) {
    return guarded([&]() -> int {
    if (!ctx || !scene || !hits) return fail(MP_ERR_INVALID, "NULL argument");
    if (n && (!d_ox || !d_oy || !d_oz || !d_dx || !d_dy || !d_dz)) return fail(MP_ERR_INVALID, "NULL ray array");
    if (scene->ctx != ctx) return fail(MP_ERR_INVALID, "scene belongs to another context");
    DeviceGuard g(ctx->device);
    std::string err;
    const uint32_t bpc = ctx->blocks_per_cu.load();
    int rc = launch_trace_rays(scene->dev, d_ox, d_oy, d_oz, d_dx, d_dy, d_dz, n, *hits, ctx->cu_count * (bpc ? static_cast<int>(bpc) : 8) / 8, stream, err);
    if (rc) return fail(rc, err);
    return MP_OK;
    });
}

int mp_generate_rays(mp_ctx* ctx, const mp_camera_sampler* sampler, const mp_settings* settings, mp_block block,
                     uint32_t sample, float* d_ox, float* d_oy, float* d_oz, float* d_dx, float* d_dy, float* d_dz,
                     void* stream) {
    return guarded([&]() -> int {
    if (!ctx || !sampler || !valid_settings(settings)) return fail(MP_ERR_INVALID, "bad argument");
    if (!(block.min_x <= block.max_x && block.min_y <= block.max_y)) return fail(MP_ERR_INVALID, "inverted block");
    if (!d_ox || !d_oy || !d_oz || !d_dx || !d_dy || !d_dz) return fail(MP_ERR_INVALID, "NULL ray array");
    DeviceGuard g(ctx->device);
    std::string err;
    int rc = launch_generate_rays(*sampler, settings->width, settings->sample_count, settings->seed, block, sample, d_ox,
                                  d_oy, d_oz, d_dx, d_dy, d_dz, stream, err);
    if (rc) return fail(rc, err);
    return MP_OK;
    });
}

// ---- tiles ------------------------------------------------------------------------------------------------------
int mp_render_tiles_device(mp_ctx* ctx, const mp_scene* scene, const mp_camera_sampler* sampler,
                           const mp_settings* settings, const mp_block* tiles, size_t n_tiles, float* d_rgba_f32,
                           void* stream) {
    return guarded([&]() -> int {
    return mp_render_tiles_device_counted(ctx, scene, sampler, settings, tiles, n_tiles, d_rgba_f32, nullptr, stream);
    });
}

int mp_render_tiles_device_counted(mp_ctx* ctx, const mp_scene* scene, const mp_camera_sampler* sampler,
                                   const mp_settings* settings, const mp_block* tiles, size_t n_tiles, float* d_rgba_f32,
                                   uint64_t* d_ray_segments, void* stream) {
    return guarded([&]() -> int {
    mp_launch_extras ex{d_ray_segments, nullptr, nullptr};
    return mp_render_tiles_device_ex(ctx, scene, sampler, settings, tiles, n_tiles, d_rgba_f32, &ex, stream);
    });
}

int mp_render_tiles_device_ex(mp_ctx* ctx, const mp_scene* scene, const mp_camera_sampler* sampler, const mp_settings* settings,
                              const mp_block* tiles, size_t n_tiles, float* d_rgba_f32, const mp_launch_extras* extras,
                              void* stream) {
    return guarded([&]() -> int {
    if (!ctx || !scene || !sampler || !valid_settings(settings)) return fail(MP_ERR_INVALID, "bad argument");
    if (n_tiles && (!tiles || !d_rgba_f32)) return fail(MP_ERR_INVALID, "NULL tiles/output");
    if (scene->ctx != ctx) return fail(MP_ERR_INVALID, "scene belongs to another context");
    if (n_tiles == 0) return MP_OK;
    if (n_tiles > 0xFFFFFFFFull) return fail(MP_ERR_INVALID, "too many tiles");
    uint64_t* d_ray_segments = extras ? extras->d_ray_segments : nullptr;
    uint64_t* d_tile_cost = extras ? extras->d_tile_cost : nullptr;
    const uint32_t* tile_order = extras ? extras->tile_order : nullptr;
    for (size_t i = 0; i < n_tiles; i++) {
        const mp_block& t = tiles[i];
        if (!(t.min_x < t.max_x && t.min_y < t.max_y) || t.max_x - t.min_x > settings->tile_size ||
            t.max_y - t.min_y > settings->tile_size || t.max_x > settings->width || t.max_y > settings->height)
            return fail(MP_ERR_INVALID, "tile empty, larger than tile_size, or outside the resolution");
    }
    if (tile_order) {  // must be a permutation: every tile is rendered exactly once
        std::vector<bool> seen(n_tiles, false);
        for (size_t i = 0; i < n_tiles; i++) {
            if (tile_order[i] >= n_tiles || seen[tile_order[i]]) return fail(MP_ERR_INVALID, "tile_order is not a permutation of 0..n_tiles-1");
            seen[tile_order[i]] = true;
        }
    }
    DeviceGuard g(ctx->device);
    const mp_block* d_tiles = nullptr;
    const uint32_t* d_order = nullptr;
    mp_ctx::TileListRef keep;  // held until the launch below is enqueued
    int rc = ctx->device_tiles(tiles, n_tiles, tile_order, keep, &d_tiles, &d_order);
    if (rc) return rc;
    if (!rc && d_ray_segments) {
        uint64_t init = 0;
        if (!(settings->flags & MP_FLAG_PATHS))  // reference semantics: one Object::intersect per sample
            for (size_t i = 0; i < n_tiles; i++)
                init += static_cast<uint64_t>(tiles[i].max_x - tiles[i].min_x) * (tiles[i].max_y - tiles[i].min_y) * pass_samples(*settings);
        std::string err;  // the value travels as a kernel argument: no host memory is read after this call returns
        rc = launch_set_u64(reinterpret_cast<unsigned long long*>(d_ray_segments), init, stream, err);
        if (rc) fail(rc, err);
    }
    if (!rc) rc = render_tiles_device(ctx, scene, *sampler, *settings, d_tiles, n_tiles, d_rgba_f32, stream, d_ray_segments, d_order, d_tile_cost);
    return rc;
    });
}

int mp_untile(mp_ctx* ctx, const mp_settings* settings, const mp_block* tiles, size_t n_tiles, const float* d_tiles_f32,
              float* d_image_f32, uint8_t* d_image_u8, void* stream) {
    return guarded([&]() -> int {
    if (!ctx || !valid_settings(settings)) return fail(MP_ERR_INVALID, "bad argument");
    if (n_tiles == 0) return MP_OK;
    if (!tiles || !d_tiles_f32) return fail(MP_ERR_INVALID, "NULL tiles/input");
    if (n_tiles > 0xFFFFFFFFull) return fail(MP_ERR_INVALID, "too many tiles");
    DeviceGuard g(ctx->device);
    const mp_block* d_tiles = nullptr;
    mp_ctx::TileListRef keep;
    int rc = ctx->device_tiles(tiles, n_tiles, nullptr, keep, &d_tiles, nullptr);  // cached: no per-frame upload
    if (rc) return rc;
    std::string err;
    rc = launch_untile(settings->width, settings->height, settings->tile_size, d_tiles, static_cast<uint32_t>(n_tiles), d_tiles_f32,
                       d_image_f32, d_image_u8, stream, err);
    if (rc) fail(rc, err);
    return rc;
    });
}

namespace {

// One pass (or whole frame) over n ranks, and optionally the gather + un-tile on ctxs[0]'s device.
int render_multi_impl(mp_ctx* const* ctxs, const mp_scene* const* scenes, int n, const mp_camera_sampler* sampler,
                      const mp_settings* settings, bool gather, float* d_image_f32, uint8_t* d_image_u8, uint64_t* ray_segments,
                      void* stream) {
    if (!ctxs || !scenes || n < 1 || n > 64 || !sampler || !valid_settings(settings)) return fail(MP_ERR_INVALID, "bad argument");
    if (settings->flags & MP_FLAG_WAVEFRONT) return fail(MP_ERR_INVALID, "the multi-device frame uses the fused kernels");
    for (int i = 0; i < n; i++) {
        if (!ctxs[i] || !scenes[i] || scenes[i]->ctx != ctxs[i]) return fail(MP_ERR_INVALID, "scenes[i] must live on ctxs[i]");
        for (int j = 0; j < i; j++)
            if (ctxs[j] == ctxs[i]) return fail(MP_ERR_INVALID, "every rank needs its own context (two contexts may share a device)");
    }
    const bool acc = (settings->flags & MP_FLAG_ACCUMULATE) != 0;
    const uint32_t p_begin = acc ? settings->pass_begin : 0u, p_end = p_begin + pass_samples(*settings);
    const bool final_pass = p_end == settings->sample_count;
    const uint32_t ts = settings->tile_size;
    const std::vector<mp_block> all = tile_ordering(mp_block{0, 0, settings->width, settings->height}, ts, 0);
    const size_t per_rank = (all.size() + static_cast<size_t>(n) - 1) / static_cast<size_t>(n);
    const size_t tile_floats = static_cast<size_t>(ts) * ts * 4;
    hipStream_t st0 = static_cast<hipStream_t>(stream);
    mp_ctx* c0 = ctxs[0];
    // Frames on the same gathering context are serialised: its gather buffer, events and peer table are one set.
    std::lock_guard<std::mutex> lk0(c0->multi_mu);
    // gather buffer on device 0, rank-major: slot r*per_rank + k = k-th tile of rank r (tiles r, r+n, r+2n, ... of the grid)
    std::vector<mp_block> order(per_rank * static_cast<size_t>(n), mp_block{0, 0, 0, 0});
    std::vector<std::vector<mp_block>> shard(static_cast<size_t>(n));
    for (size_t t = 0; t < all.size(); t++) {
        const size_t r = t % static_cast<size_t>(n), k = t / static_cast<size_t>(n);
        order[r * per_rank + k] = all[t];
        shard[r].push_back(all[t]);
    }
    if (gather) {
        DeviceGuard g(c0->device);
        if (c0->multi.gather_cap < order.size() * tile_floats) {
            if (c0->multi.untiled_valid) (void)hipEventSynchronize(c0->multi.untiled);  // the last un-tile still reads the old buffer
            if (c0->multi.d_gather) (void)hipFree(c0->multi.d_gather);
            c0->multi.d_gather = nullptr;
            c0->multi.gather_cap = 0;
            MP_HIP(hipMalloc(reinterpret_cast<void**>(&c0->multi.d_gather), order.size() * tile_floats * 4));
            c0->multi.gather_cap = order.size() * tile_floats;
        }
        if (!c0->multi.untiled) MP_HIP(hipEventCreateWithFlags(&c0->multi.untiled, hipEventDisableTiming));
        // direct peer copies where the devices can reach each other (xGMI), staged through pinned host memory otherwise; asked once
        int ndev = 0;
        MP_HIP(hipGetDeviceCount(&ndev));
        if (c0->multi.peer.size() < static_cast<size_t>(ndev)) c0->multi.peer.resize(static_cast<size_t>(ndev), 0);
        for (int r = 1; r < n; r++) {
            const int d = ctxs[r]->device;
            if (d == c0->device || d < 0 || d >= ndev || c0->multi.peer[static_cast<size_t>(d)] != 0) continue;
            int can01 = 0, can10 = 0;
            bool direct = hipDeviceCanAccessPeer(&can01, c0->device, d) == hipSuccess && hipDeviceCanAccessPeer(&can10, d, c0->device) == hipSuccess &&
                          can01 && can10;
            if (direct) {  // both directions: the copy runs on the source device's stream and writes device 0's memory
                hipError_t e1 = hipDeviceEnablePeerAccess(d, 0);
                if (e1 == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); e1 = hipSuccess; }
                hipError_t e2;
                {
                    DeviceGuard gd(d);
                    e2 = hipDeviceEnablePeerAccess(c0->device, 0);
                    if (e2 == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); e2 = hipSuccess; }
                }
                if (e1 != hipSuccess || e2 != hipSuccess) { (void)hipGetLastError(); direct = false; }
            }
            c0->multi.peer[static_cast<size_t>(d)] = direct ? 1 : 2;
        }
    }
    uint64_t segs = 0;
    // every rank renders its shard in one launch on its own stream (machinery.rs:51-116: the workers own their tiles) and, when
    // the frame is gathered, sends it to device 0 on that same stream: n copies on n streams, each peer over its own link
    for (int r = 0; r < n; r++) {
        mp_ctx* c = ctxs[r];
        const size_t nt = shard[static_cast<size_t>(r)].size();
        if (nt == 0) continue;
        DeviceGuard g(c->device);
        std::unique_lock<std::mutex> lk(c->multi_mu, std::defer_lock);
        if (c != c0) lk.lock();
        mp_ctx::MultiBuf& m = c->multi;
        if (!m.stream) MP_HIP(hipStreamCreateWithFlags(&m.stream, hipStreamNonBlocking));
        if (!m.copied) MP_HIP(hipEventCreateWithFlags(&m.copied, hipEventDisableTiming));
        if (m.shard_cap < per_rank * tile_floats) {
            MP_HIP(hipStreamSynchronize(m.stream));  // its render and its copy (issued in order on this stream) are done with the old buffer
            if (m.d_shard) (void)hipFree(m.d_shard);
            m.d_shard = nullptr;
            m.shard_cap = 0;
            m.acc_next = 0;
            MP_HIP(hipMalloc(reinterpret_cast<void**>(&m.d_shard), per_rank * tile_floats * 4));
            m.shard_cap = per_rank * tile_floats;
        }
        if (acc) {  // the running state in the shard must be this render's, at this sample
            const bool same = m.acc_seed == settings->seed && m.acc_w == settings->width && m.acc_h == settings->height && m.acc_ts == ts &&
                              m.acc_spp == settings->sample_count && m.acc_flags == (settings->flags & ~0u) && m.acc_depth == settings->max_depth &&
                              m.acc_n == static_cast<uint32_t>(n) && m.acc_rank == static_cast<uint32_t>(r) && m.acc_scene == scenes[r];
            if (p_begin != 0 && !(same && m.acc_next == p_begin))
                return fail(MP_ERR_INVALID, "progressive pass does not continue the state this rank's shard holds (pass_begin must be the previous pass's end, same settings, same ranks)");
            m.acc_seed = settings->seed; m.acc_w = settings->width; m.acc_h = settings->height; m.acc_ts = ts;
            m.acc_spp = settings->sample_count; m.acc_flags = settings->flags; m.acc_depth = settings->max_depth;
            m.acc_n = static_cast<uint32_t>(n); m.acc_rank = static_cast<uint32_t>(r); m.acc_scene = scenes[r];
            m.acc_next = final_pass ? 0u : p_end;
        } else {
            m.acc_next = 0;
        }
        const mp_block* d_tiles = nullptr;
        mp_ctx::TileListRef keep;
        int rc = c->device_tiles(shard[static_cast<size_t>(r)].data(), nt, nullptr, keep, &d_tiles, nullptr);
        if (rc) return rc;
        rc = render_tiles_device(c, scenes[r], *sampler, *settings, d_tiles, nt, m.d_shard, m.stream);
        if (rc) return rc;
        if (!(settings->flags & MP_FLAG_PATHS))
            for (const mp_block& b : shard[static_cast<size_t>(r)]) segs += static_cast<uint64_t>(b.max_x - b.min_x) * (b.max_y - b.min_y) * (p_end - p_begin);
        if (!gather) continue;
        // the previous frame's un-tile must have read the gather buffer before this frame's copy overwrites its slot
        if (c0->multi.untiled_valid) MP_HIP(hipStreamWaitEvent(m.stream, c0->multi.untiled, 0));
        float* dst = c0->multi.d_gather + static_cast<size_t>(r) * per_rank * tile_floats;
        const size_t bytes = nt * tile_floats * 4;
        const bool staged = c->device != c0->device && c0->multi.peer[static_cast<size_t>(c->device)] == 2;
        if (!staged) {
            MP_HIP(hipMemcpyPeerAsync(dst, c0->device, m.d_shard, c->device, bytes, m.stream));
        } else {
            if (m.stage_cap < per_rank * tile_floats) {
                if (m.h_stage) (void)hipHostFree(m.h_stage);
                m.h_stage = nullptr;
                m.stage_cap = 0;
                MP_HIP(hipHostMalloc(reinterpret_cast<void**>(&m.h_stage), per_rank * tile_floats * 4, hipHostMallocDefault));
                m.stage_cap = per_rank * tile_floats;
            }
            MP_HIP(hipMemcpyAsync(m.h_stage, m.d_shard, bytes, hipMemcpyDeviceToHost, m.stream));
        }
        MP_HIP(hipEventRecord(m.copied, m.stream));
    }
    if (ray_segments) *ray_segments = segs;  // reference semantics only (0 with MP_FLAG_PATHS: use mp_render_tiles_device_counted)
    if (!gather) return MP_OK;
    DeviceGuard g0(c0->device);
    for (int r = 0; r < n; r++) {
        mp_ctx* c = ctxs[r];
        const size_t nt = shard[static_cast<size_t>(r)].size();
        if (nt == 0) continue;
        MP_HIP(hipStreamWaitEvent(st0, c->multi.copied, 0));
        if (c->device != c0->device && c0->multi.peer[static_cast<size_t>(c->device)] == 2)  // second leg of the staged copy
            MP_HIP(hipMemcpyAsync(c0->multi.d_gather + static_cast<size_t>(r) * per_rank * tile_floats, c->multi.h_stage, nt * tile_floats * 4,
                                  hipMemcpyHostToDevice, st0));
    }
    const mp_block* d_order = nullptr;
    mp_ctx::TileListRef keep0;
    int rc = c0->device_tiles(order.data(), order.size(), nullptr, keep0, &d_order, nullptr);
    if (rc) return rc;
    std::string err;
    // a gather before the last pass shows a preview: the running sums scaled by the samples drawn so far (the shards keep their state)
    const uint32_t mode = (acc && !final_pass) ? ((settings->flags & MP_FLAG_CHUNKED_SUM) ? 2u : 1u) : 0u;
    rc = launch_untile(settings->width, settings->height, ts, d_order, static_cast<uint32_t>(order.size()), c0->multi.d_gather, d_image_f32,
                       d_image_u8, st0, err, mode, p_end);
    if (rc) return fail(rc, err);
    MP_HIP(hipEventRecord(c0->multi.untiled, st0));
    c0->multi.untiled_valid = true;
    return MP_OK;
}

}  // namespace

int mp_render_frame_multi(mp_ctx* const* ctxs, const mp_scene* const* scenes, int n, const mp_camera_sampler* sampler,
                          const mp_settings* settings, float* d_image_f32, uint8_t* d_image_u8, uint64_t* ray_segments,
                          void* stream) {
    return guarded([&]() -> int {
    if (settings && (settings->flags & MP_FLAG_ACCUMULATE)) return fail(MP_ERR_INVALID, "mp_render_frame_multi renders whole frames: progressive passes go through mp_render_pass_multi");
    return render_multi_impl(ctxs, scenes, n, sampler, settings, true, d_image_f32, d_image_u8, ray_segments, stream);
    });
}

int mp_render_pass_multi(mp_ctx* const* ctxs, const mp_scene* const* scenes, int n, const mp_camera_sampler* sampler,
                         const mp_settings* settings, int gather, float* d_image_f32, uint8_t* d_image_u8, uint64_t* ray_segments,
                         void* stream) {
    return guarded([&]() -> int {
    if (settings && !(settings->flags & MP_FLAG_ACCUMULATE)) return fail(MP_ERR_INVALID, "mp_render_pass_multi takes MP_FLAG_ACCUMULATE settings (pass_begin / pass_count)");
    return render_multi_impl(ctxs, scenes, n, sampler, settings, gather != 0, d_image_f32, d_image_u8, ray_segments, stream);
    });
}

int mp_untile_preview(mp_ctx* ctx, const mp_settings* settings, const mp_block* tiles, size_t n_tiles, const float* d_tiles_f32,
                      uint32_t samples_done, float* d_image_f32, uint8_t* d_image_u8, void* stream) {
    return guarded([&]() -> int {
    if (!ctx || !valid_settings(settings)) return fail(MP_ERR_INVALID, "bad argument");
    if (samples_done == 0 || samples_done > settings->sample_count) return fail(MP_ERR_INVALID, "samples_done must be in 1..sample_count");
    if (n_tiles == 0) return MP_OK;
    if (!tiles || !d_tiles_f32) return fail(MP_ERR_INVALID, "NULL tiles/input");
    if (n_tiles > 0xFFFFFFFFull) return fail(MP_ERR_INVALID, "too many tiles");
    DeviceGuard g(ctx->device);
    const mp_block* d_tiles = nullptr;
    mp_ctx::TileListRef keep;
    int rc = ctx->device_tiles(tiles, n_tiles, nullptr, keep, &d_tiles, nullptr);
    if (rc) return rc;
    std::string err;
    // after the last sample the buffer holds the means already (the launch that reaches sample_count finalises)
    const uint32_t mode = samples_done == settings->sample_count ? 0u : ((settings->flags & MP_FLAG_CHUNKED_SUM) ? 2u : 1u);
    rc = launch_untile(settings->width, settings->height, settings->tile_size, d_tiles, static_cast<uint32_t>(n_tiles), d_tiles_f32,
                       d_image_f32, d_image_u8, stream, err, mode, samples_done);
    if (rc) fail(rc, err);
    return rc;
    });
}


int mp_render_tile(mp_ctx* ctx, const mp_scene* scene, const mp_camera_sampler* sampler, const mp_settings* settings,
                   mp_block tile, float* rgba_f32, uint8_t* rgba_u8) {
    return guarded([&]() -> int {
    if (!ctx || !scene || !sampler || !valid_settings(settings)) return fail(MP_ERR_INVALID, "bad argument");
    if (settings->flags & MP_FLAG_ACCUMULATE) return fail(MP_ERR_INVALID, "MP_FLAG_ACCUMULATE needs a caller-owned device tile buffer: use mp_render_tiles_device");
    if (!(tile.min_x < tile.max_x && tile.min_y < tile.max_y)) return MP_OK;  // empty tile: internal_points yields nothing
    const uint32_t ts = settings->tile_size;
    const uint32_t w = tile.max_x - tile.min_x, h = tile.max_y - tile.min_y;
    DeviceGuard g(ctx->device);
    float* d_out = nullptr;
    const size_t bytes = static_cast<size_t>(ts) * ts * 16;
    MP_HIP(hipMalloc(reinterpret_cast<void**>(&d_out), bytes));
    int rc = mp_render_tiles_device(ctx, scene, sampler, settings, &tile, 1, d_out, nullptr);
    std::vector<float> host(static_cast<size_t>(ts) * ts * 4);
    if (!rc) {
        hipError_t e = hipMemcpy(host.data(), d_out, bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy(tile)");
    }
    (void)hipFree(d_out);
    if (rc) return rc;
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            const float* p = &host[(static_cast<size_t>(y) * ts + x) * 4];
            if (rgba_f32) std::memcpy(&rgba_f32[(static_cast<size_t>(y) * w + x) * 4], p, 16);
            if (rgba_u8)
                for (int k = 0; k < 4; k++) rgba_u8[(static_cast<size_t>(y) * w + x) * 4 + k] = to_u8(p[k]);
        }
    return MP_OK;
    });
}

}  // extern "C"

// ---- render() / RenderProgress (machinery.rs) -------------------------------------------------------------------
struct mp_render {
    // one worker thread per device (machinery.rs:51-116: one worker per core); worker i renders on ctxs[i] / scenes[i]
    std::vector<mp_ctx*> ctxs;
    std::vector<const mp_scene*> scenes;
    mp_camera_sampler sampler{};
    mp_settings settings{};
    mp_tile_started_cb started = nullptr;
    mp_tile_finished_cb finished = nullptr;
    void* user = nullptr;
    std::vector<mp_block> tiles;                // tile_ordering (machinery.rs:41-42)
    std::atomic<size_t> next_tile{0};           // next_tile_index (machinery.rs:43)
    std::atomic<size_t> done_tiles{0};
    std::atomic<bool> finished_flag{false};
    std::mutex image_mu;                        // Mutex<RgbaImage> (machinery.rs:39)
    // the host images: calloc'ed (zero pages come from the OS on first touch, i.e. when a worker files a tile -- not as a 40-MB fill
    // before the first launch), sizes in elements
    struct FreeDeleter { void operator()(void* p) const { std::free(p); } };
    std::unique_ptr<uint8_t, FreeDeleter> image_u8;
    std::unique_ptr<float, FreeDeleter> image_f32;
    size_t image_elems = 0;   // width * height * 4
    std::chrono::steady_clock::time_point start;
    std::mutex end_mu;
    bool ended = false;
    std::chrono::nanoseconds elapsed{0};
    std::vector<std::thread> workers;
    std::atomic<int> live_workers{0};
    size_t worker_count = 1;                     // contexts / worker threads of this render
    std::mutex status_mu;
    int status = MP_OK;                          // first error of any worker
    std::string error;
    size_t batch = 1;                            // tiles per get_next_tile() grab
};

namespace {

void render_worker(mp_render* r, size_t wi) {
    mp_ctx* ctx = r->ctxs[wi];
    const mp_scene* scene = r->scenes[wi];
    const mp_settings& st = r->settings;
    const uint32_t ts = st.tile_size;
    const size_t per_tile = static_cast<size_t>(ts) * ts * 4;  // floats (and u8 bytes) per tile slot
    // A batch is what one launch renders: enough work units to fill every CU a few times over, small enough that the tile
    // callbacks keep flowing (the reference hands out one tile per worker thread).  kSlots batches are in flight, each on its own
    // stream: while the GPU renders and quantises (color_to_image) the later ones and copies them to pinned host memory (the tail of
    // one launch overlaps the start of the next), this thread files the oldest batch's rows into the image and runs its callbacks.
    const size_t batch = r->batch;
    int my_status = MP_OK;  // this worker's view; the shared status records the first error of any worker
    auto set_error = [&](int code, const std::string& msg) {
        my_status = code;
        std::lock_guard<std::mutex> lk(r->status_mu);
        if (r->status == MP_OK) {
            r->status = code;
            r->error = msg;
        }
        r->next_tile.store(r->tiles.size(), std::memory_order_release);  // no new tiles for the other workers either
    };
    using Slot = mp_ctx::WorkerSlot;
    constexpr int kSlots = MP_RENDER_SLOTS;
    Slot slot[kSlots];
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) set_error(MP_ERR_HIP, std::string("render worker setup: ") + hipGetErrorString(e));
    for (Slot& s : slot)
        if (my_status == MP_OK) {
            int rc = ctx->acquire_slot(batch, per_tile, s);
            if (rc) set_error(rc, mp_last_error());
        }

    const size_t total = r->tiles.size();
    // waits for the slot's batch and files it into the image (machinery.rs:78-99)
    auto retire = [&](Slot& s) {
        if (!s.busy) return;
        s.busy = false;
        hipError_t se = hipStreamSynchronize(s.stream);
        if (se != hipSuccess) { set_error(MP_ERR_HIP, std::string("tile readback: ") + hipGetErrorString(se)); return; }
        for (size_t i = 0; i < s.n; i++) {
            const mp_block& b = r->tiles[s.first + i];
            const size_t w = b.max_x - b.min_x;
            {
                std::lock_guard<std::mutex> lk(r->image_mu);  // image.lock().copy_from(..) machinery.rs:78-89
                for (uint32_t y = b.min_y; y < b.max_y; y++) {
                    const size_t src = i * per_tile + static_cast<size_t>(y - b.min_y) * ts * 4;
                    const size_t dst = (static_cast<size_t>(y) * st.width + b.min_x) * 4;
                    if (r->image_f32) std::memcpy(r->image_f32.get() + dst, s.h_f32 + src, w * 16);
                    std::memcpy(r->image_u8.get() + dst, s.h_u8 + src, w * 4);
                }
            }
            size_t done = r->done_tiles.fetch_add(1, std::memory_order_acq_rel) + 1;
            if (r->finished) r->finished(r->user, b, mp_progress{done, total});  // machinery.rs:93-99
        }
    };
    int cur = 0;
    while (my_status == MP_OK) {
        // get_next_tile (machinery.rs:205-208): abort() stores `len` so no new tiles are handed out
        // Batches taper towards the end of the frame (guided self-scheduling): the last batch's readback and filing are not
        // overlapped by rendering, so the last ones are small -- a quarter of what is left per slot and worker, at least four tiles.
        size_t first = r->next_tile.load(std::memory_order_acquire), take = 0;
        for (;;) {
            if (first >= total) break;
            const size_t left = total - first;
            take = std::min(left, std::max<size_t>(std::min<size_t>(batch, 4), std::min(batch, left / (4 * static_cast<size_t>(kSlots) * r->worker_count))));
            if (r->next_tile.compare_exchange_weak(first, first + take, std::memory_order_acq_rel, std::memory_order_acquire)) break;
        }
        if (first >= total) break;
        Slot& s = slot[cur];
        retire(s);  // the batch issued kSlots rounds ago
        if (my_status != MP_OK) break;
        s.first = first;
        s.n = take;
        const mp_block* t = &r->tiles[first];
        if (r->started)
            for (size_t i = 0; i < s.n; i++) r->started(r->user, t[i]);  // machinery.rs:75
        e = hipMemcpyAsync(s.d_tiles, t, s.n * sizeof(mp_block), hipMemcpyHostToDevice, s.stream);
        if (e != hipSuccess) { set_error(MP_ERR_HIP, std::string("hipMemcpyAsync(tiles): ") + hipGetErrorString(e)); break; }
        int rc = render_tiles_device(ctx, scene, r->sampler, st, s.d_tiles, s.n, s.d_f32, s.stream);
        std::string err;
        if (!rc) {
            rc = launch_quantise(s.d_f32, s.d_u8, static_cast<uint64_t>(s.n) * ts * ts, s.stream, err);
            if (rc) fail(rc, err);
        }
        if (rc) { set_error(rc, mp_last_error()); break; }
        e = r->image_f32 ? hipMemcpyAsync(s.h_f32, s.d_f32, s.n * per_tile * 4, hipMemcpyDeviceToHost, s.stream) : hipSuccess;
        if (e == hipSuccess) e = hipMemcpyAsync(s.h_u8, s.d_u8, s.n * per_tile, hipMemcpyDeviceToHost, s.stream);
        if (e != hipSuccess) { set_error(MP_ERR_HIP, std::string("tile readback: ") + hipGetErrorString(e)); break; }
        s.busy = true;
        cur = (cur + 1) % kSlots;
    }
    // drain in issue order: slot[cur] holds the older batch
    for (int k = 0; k < kSlots; k++) retire(slot[(cur + k) % kSlots]);
    for (Slot& s : slot) {
        if (s.stream) (void)hipStreamSynchronize(s.stream);
        s.busy = false;
        ctx->release_slot(s);
    }
    if (r->live_workers.fetch_sub(1, std::memory_order_acq_rel) == 1) {  // the last worker out closes the render
        {
            std::lock_guard<std::mutex> lk(r->end_mu);  // machinery.rs:107-113
            r->elapsed = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - r->start);
            r->ended = true;
        }
        r->finished_flag.store(true, std::memory_order_release);
    }
}

}  // namespace

extern "C" {

int mp_render_begin(mp_ctx* ctx, const mp_scene* scene, const mp_camera* camera, const mp_settings* settings,
                    mp_tile_started_cb started, mp_tile_finished_cb finished, void* user, mp_render** out) {
    return mp_render_begin_multi(&ctx, &scene, 1, camera, settings, started, finished, user, out);
}

int mp_render_begin_multi(mp_ctx* const* ctxs, const mp_scene* const* scenes, int n, const mp_camera* camera,
                          const mp_settings* settings, mp_tile_started_cb started, mp_tile_finished_cb finished, void* user,
                          mp_render** out) {
    return guarded([&]() -> int {
    if (!ctxs || !scenes || n < 1 || n > 64 || !camera || !out || !valid_settings(settings)) return fail(MP_ERR_INVALID, "bad argument");
    if (settings->flags & MP_FLAG_ACCUMULATE) return fail(MP_ERR_INVALID, "render() draws every sample of a tile at once (worker.rs:32-49): MP_FLAG_ACCUMULATE is for mp_render_tiles_device");
    for (int i = 0; i < n; i++) {
        if (!ctxs[i] || !scenes[i]) return fail(MP_ERR_INVALID, "NULL context / scene");
        if (scenes[i]->ctx != ctxs[i]) return fail(MP_ERR_INVALID, "scenes[i] must live on ctxs[i] (the scene is replicated per device)");
    }
    auto r = std::make_unique<mp_render>();
    r->ctxs.assign(ctxs, ctxs + n);
    r->scenes.assign(scenes, scenes + n);
    r->settings = *settings;
    camera_build_sampler(*camera, settings->width, settings->height, r->sampler);  // machinery.rs:65
    r->started = started;
    r->finished = finished;
    r->user = user;
    uint64_t shuffle = 0;
    if (settings->flags & MP_FLAG_SHUFFLE_TILES)
        shuffle = static_cast<uint64_t>(std::chrono::steady_clock::now().time_since_epoch().count()) | 1ull;
    r->tiles = tile_ordering(mp_block{0, 0, settings->width, settings->height}, settings->tile_size, shuffle);
    r->image_elems = static_cast<size_t>(settings->width) * settings->height * 4;
    r->image_u8.reset(static_cast<uint8_t*>(std::calloc(std::max<size_t>(r->image_elems, 1), 1)));   // RgbaImage::new :34 (zeroed)
    const bool want_f32 = !(settings->flags & MP_FLAG_IMAGE_U8_ONLY);
    if (want_f32) r->image_f32.reset(static_cast<float*>(std::calloc(std::max<size_t>(r->image_elems, 1), sizeof(float))));
    if (!r->image_u8 || (want_f32 && !r->image_f32)) return fail(MP_ERR_NOMEM, "host image allocation failed");
    // A batch is what one launch renders: enough work units to fill every CU a few times over, small enough that the tile
    // callbacks keep flowing and that several devices share the queue evenly (the reference hands out one tile per worker).
    const uint32_t ts = settings->tile_size;
    const size_t units_per_tile = static_cast<size_t>((ts + 7) / 8) * ((ts + 7) / 8);
    // (round 3: 32 tiles on 256 CUs -- measured on the metric's frame now that its kernel takes 22 ms: 16 / 32 / 64 / 128 / 510 tiles per
    // launch -> 24.2-26.3 / 24.2-24.5 / 24.8-25.8 / 25.3-26.1 / 28.2-28.8 ms wall; the last batch's readback is not overlapped)
    size_t batch = std::max<size_t>(1, (static_cast<size_t>(ctxs[0]->cu_count) * 8 + units_per_tile - 1) / units_per_tile);
    if (n > 1) batch = std::max<size_t>(1, std::min(batch, r->tiles.size() / (static_cast<size_t>(n) * 4)));
    if (ctxs[0]->render_batch.load() != 0) batch = ctxs[0]->render_batch.load();
    r->batch = std::max<size_t>(1, std::min(batch, r->tiles.size()));  // (the slots' buffers are sized by it)
    r->start = std::chrono::steady_clock::now();
    r->live_workers.store(n);
    r->worker_count = static_cast<size_t>(n);
    try {
        for (int i = 0; i < n; i++) r->workers.emplace_back(render_worker, r.get(), static_cast<size_t>(i));
    } catch (const std::exception& ex) {
        r->next_tile.store(r->tiles.size(), std::memory_order_release);  // the workers already running find no tiles
        for (auto& t : r->workers) t.join();
        return fail(MP_ERR_UNSUPPORTED, std::string("thread spawn failed: ") + ex.what());  // machinery.rs:116
    }
    *out = r.release();
    return MP_OK;
    });
}

int mp_render_progress(const mp_render* r, mp_progress* out) {
    return guarded([&]() -> int {
    if (!r || !out) return fail(MP_ERR_INVALID, "NULL argument");
    out->finished = r->done_tiles.load(std::memory_order_acquire);
    out->total = r->tiles.size();
    return MP_OK;
    });
}

int mp_render_is_finished(const mp_render* r, int* finished) {
    return guarded([&]() -> int {
    if (!r || !finished) return fail(MP_ERR_INVALID, "NULL argument");
    *finished = r->finished_flag.load(std::memory_order_acquire) ? 1 : 0;
    return MP_OK;
    });
}

int mp_render_elapsed_ns(const mp_render* rc, uint64_t* ns) {
    return guarded([&]() -> int {
    if (!rc || !ns) return fail(MP_ERR_INVALID, "NULL argument");
    mp_render* r = const_cast<mp_render*>(rc);
    std::lock_guard<std::mutex> lk(r->end_mu);
    auto d = r->ended ? r->elapsed
                      : std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - r->start);
    *ns = static_cast<uint64_t>(d.count());
    return MP_OK;
    });
}

int mp_render_abort(mp_render* r) {
    return guarded([&]() -> int {
    if (!r) return fail(MP_ERR_INVALID, "NULL argument");
    r->next_tile.store(r->tiles.size(), std::memory_order_release);  // machinery.rs:161-165
    return MP_OK;
    });
}

int mp_render_wait(mp_render* r) {
    return guarded([&]() -> int {
    if (!r) return fail(MP_ERR_INVALID, "NULL argument");
    for (auto& t : r->workers)
        if (t.joinable()) t.join();
    if (r->status != MP_OK) return fail(r->status, r->error);
    return MP_OK;
    });
}

int mp_render_image_u8(mp_render* r, uint8_t* dst) {
    return guarded([&]() -> int {
    if (!r || !dst) return fail(MP_ERR_INVALID, "NULL argument");
    std::lock_guard<std::mutex> lk(r->image_mu);
    std::memcpy(dst, r->image_u8.get(), r->image_elems);
    return MP_OK;
    });
}

int mp_render_image_f32(mp_render* r, float* dst) {
    return guarded([&]() -> int {
    if (!r || !dst) return fail(MP_ERR_INVALID, "NULL argument");
    if (!r->image_f32) return fail(MP_ERR_UNSUPPORTED, "this render keeps the u8 image only (MP_FLAG_IMAGE_U8_ONLY)");
    std::lock_guard<std::mutex> lk(r->image_mu);
    std::memcpy(dst, r->image_f32.get(), r->image_elems * 4);
    return MP_OK;
    });
}

void mp_render_destroy(mp_render* r) {
    if (!r) return;
    r->next_tile.store(r->tiles.size(), std::memory_order_release);
    for (auto& t : r->workers)
        if (t.joinable()) t.join();
    delete r;
}

}  // extern "C"

// Host pre-pass: OBJ load + BVH8 build with u16-quantised geometry.
//
// Follows the reference's scene/triangle_bvh/building.rs (cited per function) so that the bytes the traversal
// kernels read are the ones the reference's TriangleBvh would hold (triangle geometry is quantised
// against the decompressed box chain, so the builder has to be reproduced, not just "a BVH").
//
// Structure differs from the reference on purpose ("builder acceleration"):
//  * the greedy bin merge (building.rs:278-293, 394-414) keeps a pairwise-improvement matrix, per-row cached maxima and
//    a per-packet-count cost table, so a merge costs O(groups) SAH evaluations and (mostly) O(groups) comparisons
//    instead of O(groups^2) SAH evaluations; the scan order and the strict `>` tie-break of find_best_bin_merge are
//    preserved, so the chosen pairs -- and therefore the tree -- are identical;
//  * subtrees are built as independently numbered fragments, large ones on other threads, and spliced into the parent
//    in child order, which restores the reference's pre-order node / DFS packet numbering.
//
// All arithmetic is f32 and must not be contracted (-ffp-contract=off); fmaf only where the reference writes
// mul_add (compressed_geometry.rs:103-109).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <future>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <unordered_map>

#include "mp_internal.h"

namespace mp {
namespace {

constexpr size_t kChildren = 8;        // triangle_bvh/mod.rs:14
constexpr size_t kPacket = 8;          // :15
constexpr size_t kLeafMaxTris = 56;    // :16-17
constexpr uint32_t kMaxIndex = 536870910u;  // CompressedNodeLink::MAX_INDEX :66
constexpr float kInvU16Max = 1.0f / 65535.0f;  // compressed_geometry.rs:49

inline float vmin_ps(float a, float b) { return a < b ? a : b; }  // wide fast_min -> vminps
inline float vmax_ps(float a, float b) { return a > b ? a : b; }  // wide fast_max -> vmaxps

enum class Rounding { Nearest, Floor, Ceil };

// UnitInterval8::compress_internal, compressed_geometry.rs:25-46
inline uint16_t quantise(float rel, Rounding r, bool mask) {
    float x = rel * 65535.0f;
    switch (r) {
        case Rounding::Nearest: x = std::nearbyintf(x); break;  // vroundps nearest = ties-to-even
        case Rounding::Floor: x = std::floor(x); break;
        case Rounding::Ceil: x = std::ceil(x); break;
    }
    x = mask ? x : 0.0f;
    x = vmin_ps(x, 65535.0f);
    x = vmax_ps(x, 0.0f);
    return static_cast<uint16_t>(static_cast<int32_t>(x));
}
// RelativePoint8::compress_internal :74-93
inline uint16_t quantise_coord(float p, float mn, float size, Rounding r, bool mask) {
    return quantise((p - mn) / size, r, mask);
}
// RelativePoint8::decompress :95-110
inline float dequantise_coord(uint16_t q, float size, float mn) {
    return std::fmaf(size, static_cast<float>(static_cast<int32_t>(q)) * kInvU16Max, mn);
}

struct Tri { uint32_t v[3]; uint32_t mat; };  // mat: TriangleShadingData.material (0 in the reference, building.rs:201)

inline void extend(Box3& b, const float* p) {  // aabb.rs:219-222
    for (int k = 0; k < 3; k++) {
        b.mn[k] = std::fmin(b.mn[k], p[k]);
        b.mx[k] = std::fmax(b.mx[k], p[k]);
    }
}
inline Box3 unite(const Box3& a, const Box3& b) {  // aabb.rs:209-214
    Box3 r;
    for (int k = 0; k < 3; k++) {
        r.mn[k] = std::fmin(a.mn[k], b.mn[k]);
        r.mx[k] = std::fmax(a.mx[k], b.mx[k]);
    }
    return r;
}
inline float surface_area(const Box3& b) {  // aabb.rs:247-251
    float sx = b.mx[0] - b.mn[0], sy = b.mx[1] - b.mn[1], sz = b.mx[2] - b.mn[2];
    return 2.0f * (sx * (sy + sz) + sy * sz);
}
inline size_t to_usize(float x) {  // Rust `as usize`
    if (!(x > 0.0f)) return 0;
    if (x >= 18446744073709551616.0f) return std::numeric_limits<size_t>::max();
    return static_cast<size_t>(x);
}

// One subtree, numbered from 0 (inner nodes in pre-order, packets in DFS order) so that subtrees can be built
// independently -- on different threads -- and spliced into their parent afterwards by offsetting the links.
struct Fragment {
    std::vector<InnerNodeRef> inner;
    std::vector<Box3> inner_box;
    std::vector<TriPacketRef> packets;
    std::vector<Box3> packet_box;
    std::vector<TriShadingRef> shading;
    std::vector<uint32_t> material;
    uint32_t depth = 0;  // inner levels below (and including) this subtree's root
    uint32_t root = MP_LINK_NULL;
};

class Builder {
  public:
    Builder(const float* pos, const float* nrm, uint32_t nt, std::string& err) : pos_(pos), nrm_(nrm), err_(err) {
        // SplittingBin::sah cost factor per packet count (building.rs:358-383); read-only once built
        const size_t max_pc = (static_cast<size_t>(nt) + kPacket - 1) / kPacket;
        packet_cost_.resize(max_pc + 1);
        packet_cost_[0] = 0.0f;  // a group always holds at least one triangle: never read (ln(0) would make the depth below -inf)
        for (size_t p = 1; p <= max_pc; p++) {
            const float B = 8.0f;
            float leaf_cost = (p <= 7) ? 0.75f * static_cast<float>(p) : std::numeric_limits<float>::infinity();
            float pf = static_cast<float>(p);
            float depth = std::floor(std::log(pf) / std::log(B));
            float pw = 1.0f;
            for (int i = 0; i < static_cast<int>(depth); i++) pw *= B;
            float tree_cost = 1.0f * depth + 0.75f * std::ceil(pf / pw);
            packet_cost_[p] = std::fmin(leaf_cost, tree_cost);
        }
        unsigned hw = std::thread::hardware_concurrency();
        max_tasks_ = static_cast<int>(std::min(16u, hw ? hw : 1u)) - 1;
    }

    bool run(std::vector<Tri>& tris, HostBvh& out) {
        Box3 bb;
        for (int k = 0; k < 3; k++) bb.mn[k] = bb.mx[k] = pos_[3 * tris[0].v[0] + k];
        for (const Tri& t : tris)
            for (int v = 0; v < 3; v++) extend(bb, &pos_[3 * t.v[v]]);
        Fragment f = recurse(tris.data(), tris.size(), bb);
        if (failed_.load()) return false;
        out.bbox = bb;
        out.triangle_count = static_cast<uint32_t>(tris.size());
        out.root = f.root;
        out.depth = f.depth;
        out.inner = std::move(f.inner);
        out.inner_box = std::move(f.inner_box);
        out.packets = std::move(f.packets);
        out.packet_box = std::move(f.packet_box);
        out.shading = std::move(f.shading);
        out.material = std::move(f.material);
        return true;
    }

  private:
    const float* pos_;
    const float* nrm_;
    std::string& err_;
    std::mutex err_mu_;
    std::atomic<bool> failed_{false};
    std::vector<float> packet_cost_;
    std::atomic<int> live_tasks_{0};
    int max_tasks_ = 0;

    void fail(const char* msg) {
        std::lock_guard<std::mutex> lk(err_mu_);
        if (!failed_.load()) err_ = msg;
        failed_.store(true);
    }

    void centroid(const Tri& t, float c[3]) const {  // triangle.rs:115-120
        for (int k = 0; k < 3; k++) {
            float s = 0.0f + pos_[3 * t.v[0] + k];
            s = s + pos_[3 * t.v[1] + k];
            s = s + pos_[3 * t.v[2] + k];
            c[k] = s / 3.0f;
        }
    }

    float sah(const Box3& b, size_t count) const { return surface_area(b) * packet_cost_[(count + kPacket - 1) / kPacket]; }

    struct Group {
        Box3 box;
        size_t count;
        size_t parent;  // root bin of the disjoint set
        float sah;
        uint32_t id;    // row/column in the improvement matrix
    };
    struct Child {
        size_t lo, hi;
        Box3 box;
    };

    // building.rs:238-345
    int split(Tri* tris, size_t n, Child out[8]) {
        float c[3];
        Box3 cb;
        centroid(tris[0], c);
        for (int k = 0; k < 3; k++) cb.mn[k] = cb.mx[k] = c[k];
        for (size_t i = 1; i < n; i++) { centroid(tris[i], c); extend(cb, c); }

        size_t bin_count = std::min<size_t>(std::max<size_t>(n / 64, 128), 1024);  // :248
        float sx = cb.mx[0] - cb.mn[0], sy = cb.mx[1] - cb.mn[1], sz = cb.mx[2] - cb.mn[2];
        float bin_size = std::cbrt((sx * sy * sz) / static_cast<float>(bin_count));  // :424
        size_t cnt[3] = {to_usize(std::ceil(sx / bin_size)), to_usize(std::ceil(sy / bin_size)),
                         to_usize(std::ceil(sz / bin_size))};  // :429
        size_t nb = cnt[0] * cnt[1] * cnt[2];
        if (nb == 0 || nb > (size_t{1} << 26)) {
            char msg[256];
            std::snprintf(msg, sizeof msg, "degenerate centroid box (%zu triangles, centroid extent %g x %g x %g, bin grid %zu x %zu x %zu): the "
                          "reference's BinGrid panics here (building.rs:424-429)", n, sx, sy, sz, cnt[0], cnt[1], cnt[2]);
            fail(msg);
            return 0;
        }

        struct Bin { Box3 box; size_t count; size_t parent; };
        std::vector<Bin> bins(nb);
        const float inf = std::numeric_limits<float>::infinity();
        for (size_t i = 0; i < nb; i++) bins[i] = {{{inf, inf, inf}, {-inf, -inf, -inf}}, 0, i};
        std::vector<size_t> tri_bin(n);
        for (size_t i = 0; i < n; i++) {
            centroid(tris[i], c);  // BinGrid::bin_index :442-449
            size_t bx = to_usize(std::floor((c[0] - cb.mn[0]) / bin_size));
            size_t by = to_usize(std::floor((c[1] - cb.mn[1]) / bin_size));
            size_t bz = to_usize(std::floor((c[2] - cb.mn[2]) / bin_size));
            size_t bi = bx + by * cnt[0] + bz * (cnt[0] * cnt[1]);
            if (bi >= nb) { fail("centroid bin index out of range (reference would panic)"); return 0; }
            tri_bin[i] = bi;
            for (int v = 0; v < 3; v++) extend(bins[bi].box, &pos_[3 * tris[i].v[v]]);
            bins[bi].count++;
        }
        std::vector<Group> groups;
        for (size_t i = 0; i < nb; i++)
            if (bins[i].count > 0)
                groups.push_back({bins[i].box, bins[i].count, i, sah(bins[i].box, bins[i].count),
                                  static_cast<uint32_t>(groups.size())});
        if (groups.size() < 2) { fail("all centroids in a single bin (reference asserts groups.len() >= 2, building.rs:275)"); return 0; }

        // Greedy merge (building.rs:278-293) with find_best_bin_merge's result (:394-414: first maximum in the i1<i2 scan
        // order, strict `>`) reproduced from cached data instead of an O(groups^2) SAH scan per merge:
        //   imp[a*G+b]  = sah(a) + sah(b) - sah(a u b) by group id, symmetric in exact float arithmetic;
        //   row_best[i] = first maximum of row i over the positions j > i.
        // A merge rewrites position b1 and moves the last position into b2; only rows whose cached maximum sat at one
        // of the touched positions are rescanned, the others are patched by comparing the (at most two) changed entries.
        const size_t G = groups.size();
        std::vector<float> imp(G * G, 0.0f);
        auto pair_improvement = [&](const Group& a, const Group& b) {
            float merged = sah(unite(a.box, b.box), a.count + b.count);
            return a.sah + b.sah - merged;
        };
        for (size_t i = 0; i < G; i++)
            for (size_t j = i + 1; j < G; j++) {
                float v = pair_improvement(groups[i], groups[j]);
                imp[i * G + j] = v;
                imp[j * G + i] = v;
            }
        struct RowBest { float v; size_t j; };
        std::vector<RowBest> row_best(G);
        auto at = [&](size_t i, size_t j) { return imp[static_cast<size_t>(groups[i].id) * G + groups[j].id]; };
        auto rescan = [&](size_t i) {
            RowBest rb{-inf, i};
            const size_t ng = groups.size();
            const float* row = &imp[static_cast<size_t>(groups[i].id) * G];
            for (size_t j = i + 1; j < ng; j++) {
                float v = row[groups[j].id];
                if (v > rb.v) rb = {v, j};
            }
            row_best[i] = rb;
        };
        for (size_t i = 0; i < G; i++) rescan(i);

        while (groups.size() > 2) {
            const size_t ng = groups.size();
            size_t b1 = 0, b2 = 0;
            float best = -inf;
            for (size_t i1 = 0; i1 + 1 < ng; i1++)
                if (row_best[i1].v > best) { best = row_best[i1].v; b1 = i1; b2 = row_best[i1].j; }
            if (best < 0.0f && ng <= kChildren) break;
            const size_t last = ng - 1;
            Group& g1 = groups[b1];
            const Group& g2 = groups[b2];
            bins[g2.parent].parent = g1.parent;
            Group merged{unite(g1.box, g2.box), g1.count + g2.count, g1.parent, 0.0f, g1.id};
            merged.sah = sah(merged.box, merged.count);
            groups[b1] = merged;
            groups[b2] = groups.back();  // swap_remove
            groups.pop_back();
            for (size_t j = 0; j < groups.size(); j++) {
                if (j == b1) continue;
                float v = pair_improvement(groups[b1], groups[j]);
                imp[static_cast<size_t>(merged.id) * G + groups[j].id] = v;
                imp[static_cast<size_t>(groups[j].id) * G + merged.id] = v;
            }
            const size_t ng2 = groups.size();
            auto patch = [&](size_t i, size_t j) {  // entry (i, j), j > i, has a new value
                float v = at(i, j);
                RowBest& rb = row_best[i];
                if (v > rb.v || (v == rb.v && j < rb.j)) rb = {v, j};
            };
            for (size_t i = 0; i + 1 < ng2; i++) {
                const size_t oj = row_best[i].j;
                if (i == b1 || i == b2 || oj == b1 || oj == b2 || oj == last) { rescan(i); continue; }
                if (i < b1) patch(i, b1);
                if (i < b2 && b2 < ng2) patch(i, b2);
            }
        }

        // :295-313.  sort_unstable_by_key(root bin) in the reference; topology and boxes do not depend on the order
        // inside a group, so a stable order is used: groups ascending by root, input order inside.
        std::vector<size_t> root(n);
        std::vector<size_t> roots;
        for (size_t i = 0; i < n; i++) {
            size_t r = tri_bin[i];
            while (bins[r].parent != r) r = bins[r].parent;
            root[i] = r;
            if (std::find(roots.begin(), roots.end(), r) == roots.end()) roots.push_back(r);
        }
        if (roots.size() > kChildren) { fail("more than 8 groups after merging"); return 0; }
        std::sort(roots.begin(), roots.end());
        std::vector<Tri> tmp;
        tmp.reserve(n);
        int nchild = 0;
        for (size_t r : roots) {
            Child& ch = out[nchild++];
            ch.lo = tmp.size();
            bool first = true;
            for (size_t i = 0; i < n; i++) {
                if (root[i] != r) continue;
                if (first) {  // :322-333 from_points(first triangle) then extend
                    for (int k = 0; k < 3; k++) ch.box.mn[k] = ch.box.mx[k] = pos_[3 * tris[i].v[0] + k];
                    first = false;
                }
                for (int v = 0; v < 3; v++) extend(ch.box, &pos_[3 * tris[i].v[v]]);
                tmp.push_back(tris[i]);
            }
            ch.hi = tmp.size();
        }
        std::copy(tmp.begin(), tmp.end(), tris);
        return nchild;
    }

    // building.rs:109-120
    Fragment recurse(Tri* tris, size_t n, const Box3& enc) {
        if (failed_.load()) return Fragment{};
        return n <= kLeafMaxTris ? leaf(tris, n, enc) : inner(tris, n, enc);
    }

    // building.rs:170-207
    Fragment leaf(const Tri* tris, size_t n, const Box3& enc) {
        Fragment f;
        if (n == 0) { fail("empty leaf (reference asserts !triangles.is_empty(), building.rs:178)"); return f; }
        float size[3];
        for (int k = 0; k < 3; k++) size[k] = enc.mx[k] - enc.mn[k];  // aabb.rs:292-303
        size_t packets = (n + kPacket - 1) / kPacket;
        for (size_t p = 0; p < packets; p++) {
            TriPacketRef pk{};
            for (size_t lane = 0; lane < kPacket; lane++) {
                size_t ti = p * kPacket + lane;
                bool mask = ti < n;
                TriShadingRef sh{{0, 0, 0}, 0};  // Default for padding :203-204
                for (int v = 0; v < 3; v++)
                    for (int k = 0; k < 3; k++) {
                        float pv = mask ? pos_[3 * tris[ti].v[v] + k] : 0.0f;
                        pk.v[v][k][lane] = quantise_coord(pv, enc.mn[k], size[k], Rounding::Nearest, mask);
                    }
                if (mask) {
                    for (int v = 0; v < 3; v++) {
                        const float* nn = &nrm_[3 * tris[ti].v[v]];
                        if (nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2] == 0.0f) sh.flat = 1;  // :200
                        sh.vi[v] = tris[ti].v[v];
                    }
                }
                f.shading.push_back(sh);
                f.material.push_back(mask ? tris[ti].mat : 0u);
            }
            f.packets.push_back(pk);
            f.packet_box.push_back(enc);
        }
        f.root = static_cast<uint32_t>(packets);  // new_leaf(index 0, count) mod.rs:70-74, index offset by the parent
        return f;
    }

    static void offset_links(Fragment& f, uint32_t inner_off, uint32_t packet_off) {
        auto fix = [&](uint32_t& link) {
            if (link == MP_LINK_NULL) return;
            link += ((link & 7u) == 0u ? inner_off : packet_off) << 3;
        };
        for (InnerNodeRef& nd : f.inner)
            for (uint32_t& l : nd.link) fix(l);
        fix(f.root);
    }

    // building.rs:122-168
    Fragment inner(Tri* tris, size_t n, const Box3& enc) {
        Fragment f;
        Child ch[8];
        int nchild = split(tris, n, ch);
        if (failed_.load() || nchild <= 0) return f;
        float size[3];
        for (int k = 0; k < 3; k++) size[k] = enc.mx[k] - enc.mn[k];
        InnerNodeRef node{};
        Box3 dec[8];
        const float inf = std::numeric_limits<float>::infinity();
        for (int i = 0; i < 8; i++) {
            bool mask = i < nchild;
            for (int k = 0; k < 3; k++) {
                float pmin = mask ? ch[i].box.mn[k] : inf;   // AABB::default lanes (aabb.rs:139-157)
                float pmax = mask ? ch[i].box.mx[k] : -inf;
                node.bmin[k][i] = quantise_coord(pmin, enc.mn[k], size[k], Rounding::Floor, mask);  // :128
                node.bmax[k][i] = quantise_coord(pmax, enc.mn[k], size[k], Rounding::Ceil, mask);   // :129
                dec[i].mn[k] = dequantise_coord(node.bmin[k][i], size[k], enc.mn[k]);                // :146
                dec[i].mx[k] = dequantise_coord(node.bmax[k][i], size[k], enc.mn[k]);
            }
            node.link[i] = MP_LINK_NULL;
        }
        // children are independent: large ones go to other threads, the numbering is restored when they are spliced
        Fragment sub[8];
        std::future<Fragment> fut[8];
        for (int i = 0; i < nchild; i++) {
            const size_t cn = ch[i].hi - ch[i].lo;
            if (cn >= 4096 && live_tasks_.fetch_add(1) < max_tasks_) {
                fut[i] = std::async(std::launch::async, [this, tris, &ch, &dec, i, cn]() {
                    Fragment r = recurse(tris + ch[i].lo, cn, dec[i]);
                    live_tasks_.fetch_sub(1);
                    return r;
                });
            } else {
                if (cn >= 4096) live_tasks_.fetch_sub(1);
                sub[i] = recurse(tris + ch[i].lo, cn, dec[i]);
            }
        }
        for (int i = 0; i < nchild; i++)
            if (fut[i].valid()) sub[i] = fut[i].get();
        if (failed_.load()) return Fragment{};
        f.inner.push_back(node);  // pre-order: this node is index 0 of its fragment (placeholder :130-131)
        f.inner_box.push_back(enc);
        for (int i = 0; i < nchild; i++) {
            Fragment& c = sub[i];
            const size_t io = f.inner.size(), po = f.packets.size();
            if (io + c.inner.size() > kMaxIndex || po + c.packets.size() > kMaxIndex) { fail("link index out of range"); return Fragment{}; }
            offset_links(c, static_cast<uint32_t>(io), static_cast<uint32_t>(po));
            f.inner[0].link[i] = c.root;
            f.inner.insert(f.inner.end(), c.inner.begin(), c.inner.end());
            f.inner_box.insert(f.inner_box.end(), c.inner_box.begin(), c.inner_box.end());
            f.packets.insert(f.packets.end(), c.packets.begin(), c.packets.end());
            f.packet_box.insert(f.packet_box.end(), c.packet_box.begin(), c.packet_box.end());
            f.shading.insert(f.shading.end(), c.shading.begin(), c.shading.end());
            f.material.insert(f.material.end(), c.material.begin(), c.material.end());
            f.depth = std::max(f.depth, c.depth);
        }
        f.depth += 1;
        f.root = 0;  // new_inner(index 0) mod.rs:77-80
        return f;
    }
};

}  // namespace

int build_bvh(const float* pos, const float* nrm, const float* tex, uint32_t nv, const uint32_t* tri, const uint32_t* tri_mat,
              uint32_t nt, HostBvh& out, std::string& err) {
    out = HostBvh{};
    if (nt == 0) { err = "no triangles (reference panics in build_leaf, building.rs:178)"; return MP_ERR_BUILD; }
    if (!pos || !tri) { err = "null positions/indices"; return MP_ERR_INVALID; }
    for (size_t i = 0; i < static_cast<size_t>(nt) * 3; i++)
        if (tri[i] >= nv) { err = "vertex index out of range"; return MP_ERR_INVALID; }
    if (tri_mat)
        for (size_t i = 0; i < nt; i++)
            if (tri_mat[i] >= MP_MAX_MATERIALS) { err = "material id out of range (MP_MAX_MATERIALS)"; return MP_ERR_INVALID; }
    out.vertex_count = nv;
    out.vnormal.assign(static_cast<size_t>(nv) * 3, 0.0f);
    out.vtex.assign(static_cast<size_t>(nv) * 3, 0.0f);
    if (nrm) std::memcpy(out.vnormal.data(), nrm, static_cast<size_t>(nv) * 12);
    if (tex) std::memcpy(out.vtex.data(), tex, static_cast<size_t>(nv) * 12);
    std::vector<Tri> tris(nt);
    for (size_t i = 0; i < nt; i++) tris[i] = Tri{{tri[3 * i], tri[3 * i + 1], tri[3 * i + 2]}, tri_mat ? tri_mat[i] : 0u};
    Builder b(pos, out.vnormal.data(), nt, err);
    if (!b.run(tris, out)) return MP_ERR_BUILD;
    return MP_OK;
}

// ---- import of reference-layout arrays (mp_scene_from_arrays) ----------------------------------------------------------
// The box chain of SURVEY A.4 is a pure function of the arrays: child i of a node with enclosing box {mn, size} has
// {fma(size, q_min/65535, mn), fma(size, q_max/65535, mn)} (ray_bvh_intersection.rs:155-157), leaves inherit their entry's box.
int bvh_from_arrays(const mp_bvh_desc& d, HostBvh& out, std::string& err) {
    out = HostBvh{};
    if ((d.inner_count && !d.inner_nodes) || (d.packet_count && (!d.packets || !d.tri_shading))) { err = "NULL array"; return MP_ERR_INVALID; }
    if (d.vertex_count && !d.vertex_normals) { err = "NULL vertex normals"; return MP_ERR_INVALID; }
    if (d.packet_count && d.vertex_count == 0) { err = "triangle packets but no vertices"; return MP_ERR_INVALID; }
    if (d.inner_count > kMaxIndex || d.packet_count > kMaxIndex) { err = "more nodes/packets than a CompressedNodeLink can address (mod.rs:66)"; return MP_ERR_INVALID; }
    const size_t ni = d.inner_count, np = d.packet_count;
    out.inner.resize(ni);
    out.packets.resize(np);
    out.shading.resize(np * 8);
    out.material.assign(np * 8, 0u);
    if (ni) std::memcpy(out.inner.data(), d.inner_nodes, ni * sizeof(InnerNodeRef));
    if (np) std::memcpy(out.packets.data(), d.packets, np * sizeof(TriPacketRef));
    if (np) std::memcpy(out.shading.data(), d.tri_shading, np * 8 * sizeof(TriShadingRef));
    if (np && d.tri_material) std::memcpy(out.material.data(), d.tri_material, np * 8 * sizeof(uint32_t));
    out.vertex_count = d.vertex_count;
    out.vnormal.assign(static_cast<size_t>(d.vertex_count) * 3, 0.0f);
    out.vtex.assign(static_cast<size_t>(d.vertex_count) * 3, 0.0f);
    if (d.vertex_count) std::memcpy(out.vnormal.data(), d.vertex_normals, static_cast<size_t>(d.vertex_count) * 12);
    if (d.vertex_count && d.vertex_tex) std::memcpy(out.vtex.data(), d.vertex_tex, static_cast<size_t>(d.vertex_count) * 12);
    for (int k = 0; k < 3; k++) { out.bbox.mn[k] = d.bbox_min[k]; out.bbox.mx[k] = d.bbox_max[k]; }
    out.root = d.root_link;
    out.material_names.assign(1, std::string());
    for (const TriShadingRef& sh : out.shading)
        for (int v = 0; v < 3; v++)
            if (sh.vi[v] >= d.vertex_count) { err = "vertex index out of range in tri_shading"; return MP_ERR_INVALID; }
    for (uint32_t m : out.material)
        if (m >= MP_MAX_MATERIALS) { err = "material id out of range (MP_MAX_MATERIALS)"; return MP_ERR_INVALID; }
    out.inner_box.assign(ni, Box3{});
    out.packet_box.assign(np, Box3{});
    std::vector<uint8_t> inner_seen(ni, 0), packet_seen(np, 0);
    struct Item { uint32_t link; Box3 box; uint32_t level; };
    std::vector<Item> todo;
    todo.push_back({d.root_link, out.bbox, 0});
    uint32_t depth = 0;
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        if (it.link == MP_LINK_NULL) continue;
        const uint32_t idx = it.link >> 3, cnt = it.link & 7u;
        if (cnt != 0) {  // leaf: cnt consecutive packets
            if (static_cast<size_t>(idx) + cnt > np) { err = "leaf link outside the packet array"; return MP_ERR_INVALID; }
            for (uint32_t p = idx; p < idx + cnt; p++) {
                if (packet_seen[p]) { err = "packet referenced by two leaves"; return MP_ERR_INVALID; }
                packet_seen[p] = 1;
                out.packet_box[p] = it.box;
            }
            continue;
        }
        if (idx >= ni) { err = "inner link outside the node array"; return MP_ERR_INVALID; }
        if (inner_seen[idx]) { err = "inner node referenced twice (not a tree)"; return MP_ERR_INVALID; }
        inner_seen[idx] = 1;
        out.inner_box[idx] = it.box;
        depth = std::max(depth, it.level + 1);
        float size[3];
        for (int k = 0; k < 3; k++) size[k] = it.box.mx[k] - it.box.mn[k];
        const InnerNodeRef& nd = out.inner[idx];
        for (int i = 0; i < 8; i++) {
            if (nd.link[i] == MP_LINK_NULL) continue;
            // the device walk numbers stack slots by node order: children must follow their parent (pre-order, as the builder emits)
            if ((nd.link[i] & 7u) == 0u && (nd.link[i] >> 3) <= idx) { err = "inner nodes are not in pre-order (child index <= parent index)"; return MP_ERR_INVALID; }
            Box3 cb;
            for (int k = 0; k < 3; k++) {
                cb.mn[k] = dequantise_coord(nd.bmin[k][i], size[k], it.box.mn[k]);
                cb.mx[k] = dequantise_coord(nd.bmax[k][i], size[k], it.box.mn[k]);
            }
            todo.push_back({nd.link[i], cb, it.level + 1});
        }
    }
    out.depth = depth;
    // real (unpadded) triangles: padding = all-zero quantised vertices with default shading at the tail of a leaf's last packet
    uint32_t real = 0;
    for (size_t p = 0; p < np; p++)
        for (int i = 0; i < 8; i++) {
            bool pad = true;
            for (int a = 0; a < 3 && pad; a++)
                for (int k = 0; k < 3; k++)
                    if (out.packets[p].v[a][k][i] != 0) { pad = false; break; }
            const TriShadingRef& sh = out.shading[p * 8 + i];
            if (!(pad && sh.vi[0] == 0 && sh.vi[1] == 0 && sh.vi[2] == 0 && sh.flat == 0)) real++;
        }
    out.triangle_count = real;
    return MP_OK;
}

// ---- OBJ: building.rs:36-81 over the `obj` crate (triangles only; file order; first-seen vertex dedupe) ------

namespace {
struct Key {
    int64_t p, t, n;
    bool operator==(const Key& o) const { return p == o.p && t == o.t && n == o.n; }
};
struct KeyHash {
    size_t operator()(const Key& k) const {
        uint64_t h = static_cast<uint64_t>(k.p) * 0x9e3779b97f4a7c15ull;
        h ^= (static_cast<uint64_t>(k.t) + 0x632be59bd9b4e019ull) * 0xff51afd7ed558ccdull;
        h ^= (static_cast<uint64_t>(k.n) + 0x2545f4914f6cdd1dull) * 0xc4ceb9fe1a85ec53ull;
        return static_cast<size_t>(h ^ (h >> 31));
    }
};

bool parse_tuple(const char* s, size_t np, size_t nt, size_t nn, Key& k) {
    int64_t v[3] = {-1, -1, -1};
    for (int f = 0; f < 3; f++) {
        if (*s == 0) break;
        if (*s == '/') { s++; continue; }
        char* end;
        long long x = std::strtoll(s, &end, 10);
        if (end == s) return false;
        size_t cnt = f == 0 ? np : (f == 1 ? nt : nn);
        int64_t idx = x > 0 ? static_cast<int64_t>(x) - 1 : static_cast<int64_t>(cnt) + x;
        if (idx < 0 || static_cast<size_t>(idx) >= cnt) return false;
        v[f] = idx;
        s = end;
        if (*s == '/') s++;
        else break;
    }
    k = {v[0], v[1], v[2]};
    return v[0] >= 0;
}
}  // namespace

int load_obj(const char* path, std::vector<float>& pos, std::vector<float>& nrm, std::vector<float>& tex,
             std::vector<uint32_t>& tri, std::vector<uint32_t>& tri_mat, std::vector<std::string>& material_names, std::string& err) {
    FILE* f = std::fopen(path, "r");
    if (!f) { err = std::string("Failed to read file: ") + path; return MP_ERR_IO; }
    std::vector<float> P, T, N;
    std::unordered_map<Key, uint32_t, KeyHash> seen;
    pos.clear(); nrm.clear(); tex.clear(); tri.clear(); tri_mat.clear();
    material_names.assign(1, std::string());  // id 0: faces before any `usemtl` (the reference's `material: 0`)
    uint32_t cur_mat = 0;
    char* line = nullptr;
    size_t cap = 0;
    bool bad = false;
    while (!bad && getline(&line, &cap, f) >= 0) {
        char* s = line;
        while (*s == ' ' || *s == '\t') s++;
        auto is_ws = [](char c) { return c == ' ' || c == '\t'; };
        if (s[0] == 'v' && is_ws(s[1])) {
            char* e = s + 1;
            for (int k = 0; k < 3; k++) P.push_back(std::strtof(e, &e));
        } else if (s[0] == 'v' && s[1] == 'n' && is_ws(s[2])) {
            char* e = s + 2;
            for (int k = 0; k < 3; k++) N.push_back(std::strtof(e, &e));
        } else if (s[0] == 'v' && s[1] == 't' && is_ws(s[2])) {
            char* e = s + 2;
            for (int k = 0; k < 2; k++) T.push_back(std::strtof(e, &e));
        } else if (std::strncmp(s, "usemtl", 6) == 0 && is_ws(s[6])) {
            // ids in first-seen order of the names, from 1 (the reference ignores usemtl: ids only feed the build-defined extension)
            char* nm = s + 6;
            while (is_ws(*nm)) nm++;
            std::string name(nm);
            while (!name.empty() && (name.back() == '\n' || name.back() == '\r' || is_ws(name.back()))) name.pop_back();
            auto it = std::find(material_names.begin() + 1, material_names.end(), name);
            if (it == material_names.end()) {
                material_names.push_back(name);
                cur_mat = static_cast<uint32_t>(material_names.size() - 1);
            } else {
                cur_mat = static_cast<uint32_t>(it - material_names.begin());
            }
        } else if (s[0] == 'f' && is_ws(s[1])) {
            Key keys[4];
            int nv = 0;
            char* save = nullptr;
            for (char* tok = strtok_r(s + 1, " \t\r\n", &save); tok; tok = strtok_r(nullptr, " \t\r\n", &save)) {
                Key k;
                if (!parse_tuple(tok, P.size() / 3, T.size() / 2, N.size() / 3, k)) { bad = true; break; }
                if (nv < 4) keys[nv] = k;
                nv++;
            }
            if (bad) break;
            if (nv != 3) continue;  // "non-triangle primitive!" :43-46
            for (int v = 0; v < 3; v++) {
                auto it = seen.find(keys[v]);
                uint32_t idx;
                if (it == seen.end()) {
                    idx = static_cast<uint32_t>(seen.size());
                    seen.emplace(keys[v], idx);
                    for (int k = 0; k < 3; k++) pos.push_back(P[3 * keys[v].p + k]);
                    if (keys[v].t >= 0) { tex.push_back(T[2 * keys[v].t]); tex.push_back(T[2 * keys[v].t + 1]); tex.push_back(0.0f); }
                    else { tex.insert(tex.end(), 3, 0.0f); }
                    if (keys[v].n >= 0) {
                        const float* n = &N[3 * keys[v].n];
                        float len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);  // .normalize() :62
                        for (int k = 0; k < 3; k++) nrm.push_back(n[k] / len);
                    } else { nrm.insert(nrm.end(), 3, 0.0f); }
                } else idx = it->second;
                tri.push_back(idx);
            }
            tri_mat.push_back(cur_mat);
        }
    }
    std::free(line);
    std::fclose(f);
    if (bad) { err = std::string("Failed to parse file: ") + path; return MP_ERR_IO; }
    return MP_OK;
}

}  // namespace mp

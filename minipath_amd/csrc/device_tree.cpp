// Traversal-format ("device") node arrays built from the reference-layout tree.
//
// Two trees over the same leaves:
//   * the LITERAL tree: node n of the reference's InnerNode array with its eight child slots in place (null links kept), child
//     boxes decompressed once on the host (box chain of SURVEY A.4: same fmaf, same bits as ray_bvh_intersection.rs:155-157);
//   * the WIDE tree: the literal tree with thin inner nodes ABSORBED into their parent, so that the walks test (close to) eight
//     real boxes per node step.  The reference builder makes every split binary-ish at the top of large scenes (the stand-in:
//     a chain of 2-child nodes 23 levels deep that a ray crosses 10 times), and each node step costs a walk the same whether
//     the node has two children or eight.
//
// Why the wide tree gives bit-identical hits (rays whose inverse direction is finite in all three components; the walks use the
// literal tree for every other ray):
//   Let p be a child of N whose child boxes g_i all satisfy  p.min <= g_i.min <= g_i.max <= p.max  in floating point, on every
//   axis ("FP-nested").  The slab test (aabb.rs:254-284) computes a = (box.min - o) * inv, c = (box.max - o) * inv per axis;
//   IEEE subtraction of a common o and multiplication by a common finite inv are monotone, so for inv > 0
//   a_p <= a_g <= c_g <= c_p, for inv < 0 the mirror image, hence lo_p <= lo_g and hi_g <= hi_p per axis, t1_p <= t1_g and
//   hi_g <= hi_p (max / min are monotone; a NaN can only come from a NaN origin component, which makes that axis unbounded for
//   both boxes alike).  The reference visits g iff  t1_p <= min(hi_p, best.t at p's push), t1_p <= best.t at p's pop,
//   t1_g <= min(hi_g, best.t at g's push) and t1_g <= best.t at g's pop (ray_bvh_intersection.rs:40-44,149-162).  best.t never
//   grows and g is popped after p, so all of that is equivalent to  t1_g <= hi_g  and  t1_g <= best.t at g's pop  -- g's own two
//   tests, which is what the wide tree evaluates for the slot that g takes in N.  Children are pushed in ascending slot order and
//   popped in descending order in both trees and p's children take p's place in N's list, so the leaves are reached in the same
//   order with the same best.t.  A node is absorbed only if ALL its child boxes are FP-nested in its own box; the check is on
//   the very floats the kernels read.
//   With an infinite inverse component the argument fails for one degenerate case (inv = -inf from a negative denormal
//   direction, a flat box exactly at the origin coordinate), so those rays -- rare -- walk the literal tree.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "mp_internal.h"

namespace mp {
namespace {

constexpr float kInvU16Max = 1.0f / 65535.0f;
inline float dequantise(uint16_t q, float size, float mn) {  // compressed_geometry.rs:48-51,95-110
    return std::fmaf(size, static_cast<float>(static_cast<int32_t>(q)) * kInvU16Max, mn);
}

// decompressed child boxes of reference node n (ray_bvh_intersection.rs:155-157)
void child_boxes(const HostBvh& h, size_t n, Box3 out[8]) {
    const InnerNodeRef& nd = h.inner[n];
    const Box3& e = h.inner_box[n];
    const float size[3] = {e.mx[0] - e.mn[0], e.mx[1] - e.mn[1], e.mx[2] - e.mn[2]};
    for (int i = 0; i < 8; i++)
        for (int k = 0; k < 3; k++) {
            out[i].mn[k] = dequantise(nd.bmin[k][i], size[k], e.mn[k]);
            out[i].mx[k] = dequantise(nd.bmax[k][i], size[k], e.mn[k]);
        }
}

inline bool ordered(const Box3& b) { return b.mn[0] <= b.mx[0] && b.mn[1] <= b.mx[1] && b.mn[2] <= b.mx[2]; }
inline bool contains(const Box3& p, const Box3& g) {
    for (int k = 0; k < 3; k++)
        if (!(p.mn[k] <= g.mn[k] && g.mx[k] <= p.mx[k])) return false;
    return true;
}
inline float half_area(const Box3& b) {
    const float sx = b.mx[0] - b.mn[0], sy = b.mx[1] - b.mn[1], sz = b.mx[2] - b.mn[2];
    return sx * (sy + sz) + sy * sz;
}

struct Slot {
    Box3 box;
    uint32_t ref;  // reference link (mod.rs:57-114): idx << 3 | count
};

void put_record(std::vector<float>& nodes, size_t node, int slot, const Box3& b, uint32_t dl) {
    float* r = &nodes[node * 64 + static_cast<size_t>(slot) * 8];
    for (int k = 0; k < 3; k++) { r[k] = b.mn[k]; r[3 + k] = b.mx[k]; }
    std::memcpy(&r[6], &dl, 4);
}

// Exact bound of the traversal stack: a node pushes at most its real (non-null) children in ascending order and pops them in
// descending order, so while the subtree of the child at position p is walked, p lower siblings wait below it.
// bound(node) = max(#children, max_p(p + bound(child_p))).  Children have larger indices than their parent in both trees.
uint32_t stack_bound(const std::vector<float>& nodes, uint32_t count, uint32_t root) {
    if (root == MP_LINK_NULL || (root & 63u) != 0u || count == 0) return 1u;
    std::vector<uint32_t> bound(count, 0);
    for (size_t n = count; n-- > 0;) {
        uint32_t p = 0, best = 0;
        for (int i = 0; i < 8; i++) {
            uint32_t l;
            std::memcpy(&l, &nodes[n * 64 + static_cast<size_t>(i) * 8 + 6], 4);
            if (l == MP_LINK_NULL) continue;
            const uint32_t sub = (l & 63u) == 0u ? bound[l >> 6] : 0u;
            best = std::max(best, p + sub);
            p++;
        }
        bound[n] = std::max(best, p);
    }
    return std::max<uint32_t>(1u, bound[root >> 6]);
}

}  // namespace

std::vector<uint32_t> packet_real_counts(const HostBvh& h) {
    // real (unpadded) triangles per packet: padding lanes are all-zero quantised triangles with default shading, at the tail of a
    // leaf's last packet
    const size_t np = h.packets.size();
    std::vector<uint32_t> pkt_valid(np, 0);
    for (size_t p = 0; p < np; p++)
        for (int i = 0; i < 8; i++) {
            bool pad = true;
            for (int a = 0; a < 3 && pad; a++)
                for (int k = 0; k < 3; k++)
                    if (h.packets[p].v[a][k][i] != 0) { pad = false; break; }
            const TriShadingRef& sh = h.shading[p * 8 + i];
            pad = pad && sh.vi[0] == 0 && sh.vi[1] == 0 && sh.vi[2] == 0 && sh.flat == 0;
            if (!pad) pkt_valid[p] = static_cast<uint32_t>(i + 1);
        }
    return pkt_valid;
}

int build_device_tree(const HostBvh& h, const std::vector<uint32_t>& pkt_valid, bool wide, DeviceTree& out, std::string& err) {
    const size_t ni = h.inner.size(), np = h.packets.size();
    // device links hold 26 bits of index; the 8-lane-group walk and the cached packet walk address records as a base + a 32-bit byte
    // offset (288 bytes per packet, 256 per node)
    if (np >= (1u << 26) - 1u || ni >= (1u << 24) || static_cast<uint64_t>(np) * 288u >= (1ull << 32)) {
        err = "scene too large for the device traversal format (2^24 nodes, 2^32 bytes of triangle records = 14.9 M packets)";
        return MP_ERR_UNSUPPORTED;
    }
    // device link (mp_internal.h): leaf = first packet << 6 | real triangles ; null unchanged ; inner = device node index << 6
    auto leaf_link = [&](uint32_t l) -> uint32_t {
        const uint32_t idx = l >> 3, cnt = l & 7u;
        const uint32_t n_real = (cnt - 1u) * 8u + pkt_valid[idx + cnt - 1u];
        return (idx << 6) | std::max<uint32_t>(n_real, 1u);  // a leaf of padding only (imported arrays) still tests one, never-hit, triangle
    };
    out = DeviceTree{};
    out.boxes_ordered = true;
    const bool root_inner = h.root != MP_LINK_NULL && (h.root & 7u) == 0u;
    if (!wide) {
        // literal tree: node n = reference node n, slots in place
        out.count = static_cast<uint32_t>(ni);
        out.nodes.assign(std::max<size_t>(ni, 1) * 64 + 16, 0.0f);  // + tail padding: the child loop fetches one record ahead
        for (size_t n = 0; n < ni; n++) {
            Box3 cb[8];
            child_boxes(h, n, cb);
            uint32_t nchild = 0;
            for (int i = 0; i < 8; i++) {
                const uint32_t l = h.inner[n].link[i];
                const uint32_t dl = l == MP_LINK_NULL ? l : ((l & 7u) == 0u ? (l >> 3) << 6 : leaf_link(l));
                put_record(out.nodes, n, i, cb[i], dl);
                if (l != MP_LINK_NULL) {
                    nchild = static_cast<uint32_t>(i) + 1u;
                    if (!ordered(cb[i])) out.boxes_ordered = false;
                }
            }
            std::memcpy(&out.nodes[n * 64 + 7], &nchild, 4);
        }
        out.root = h.root == MP_LINK_NULL ? h.root : (root_inner ? (h.root >> 3) << 6 : leaf_link(h.root));
        out.stack_bound = stack_bound(out.nodes, out.count, out.root);
        return MP_OK;
    }
    // wide tree, numbered in pre-order (a node's first subtree follows it: the order the walks touch memory in)
    // The first record of the tail padding (slot count * 8) is the ROOT's: an unbounded box and the root's link, so that a walk
    // which tests an entry's box when it pops the entry (kernels.hip, trace_packet_cached) needs no special case for the root --
    // every ray passes an unbounded box with t1 = 0, which is what "the root is never culled" (:28-32) means.
    auto put_root_record = [&]() {
        float* rr = &out.nodes[static_cast<size_t>(out.count) * 64];
        rr[0] = rr[1] = rr[2] = -INFINITY;
        rr[3] = rr[4] = rr[5] = INFINITY;
        std::memcpy(&rr[6], &out.root, 4);
    };
    if (!root_inner) {
        out.nodes.assign(64 + 16, 0.0f);
        out.root = h.root == MP_LINK_NULL ? h.root : leaf_link(h.root);
        put_root_record();
        return MP_OK;
    }
    // nestable[n]: every real child box of reference node n is ordered and FP-contained in n's own box (the box n's parent
    // holds for it = inner_box[n], the box it decompresses its children against)
    std::vector<uint8_t> nestable(ni, 0);
    std::vector<uint8_t> nreal(ni, 0);
    for (size_t n = 0; n < ni; n++) {
        Box3 cb[8];
        child_boxes(h, n, cb);
        bool ok = true;
        uint8_t k = 0;
        for (int i = 0; i < 8; i++) {
            if (h.inner[n].link[i] == MP_LINK_NULL) continue;
            k++;
            if (!(ordered(cb[i]) && contains(h.inner_box[n], cb[i]))) ok = false;
        }
        nestable[n] = ok && ordered(h.inner_box[n]);
        nreal[n] = k;
    }
    struct Work {
        uint32_t ref_node;
        uint32_t parent;  // device node holding the link to patch, or ~0u for the root
        int slot;
    };
    std::vector<Work> todo{{h.root >> 3, ~0u, 0}};
    out.nodes.reserve(ni * 64 + 16);
    std::vector<Slot> slots;
    while (!todo.empty()) {
        const Work w = todo.back();
        todo.pop_back();
        const uint32_t me = out.count++;
        out.nodes.resize(static_cast<size_t>(out.count) * 64, 0.0f);
        if (w.parent == ~0u) {
            out.root = me << 6;
        } else {
            const uint32_t dl = me << 6;
            std::memcpy(&out.nodes[static_cast<size_t>(w.parent) * 64 + static_cast<size_t>(w.slot) * 8 + 6], &dl, 4);
        }
        slots.clear();
        {
            Box3 cb[8];
            child_boxes(h, w.ref_node, cb);
            for (int i = 0; i < 8; i++)
                if (h.inner[w.ref_node].link[i] != MP_LINK_NULL) slots.push_back(Slot{cb[i], h.inner[w.ref_node].link[i]});
        }
        // absorb thin children while the node stays within eight slots: the candidate with the largest surface first (the one a
        // ray is most likely to enter, i.e. the node step most often saved)
        for (;;) {
            int pick = -1;
            float pick_area = -1.0f;
            for (size_t i = 0; i < slots.size(); i++) {
                const uint32_t l = slots[i].ref;
                if ((l & 7u) != 0u) continue;
                const uint32_t c = l >> 3;
                if (!nestable[c] || nreal[c] == 0 || slots.size() - 1 + nreal[c] > 8) continue;
                // the slot's box is the box c was built against (same floats); checked rather than assumed
                if (std::memcmp(&slots[i].box, &h.inner_box[c], sizeof(Box3)) != 0) continue;
                const float a = half_area(slots[i].box);
                if (a > pick_area) { pick_area = a; pick = static_cast<int>(i); }
            }
            if (pick < 0) break;
            const uint32_t c = slots[static_cast<size_t>(pick)].ref >> 3;
            Box3 cb[8];
            child_boxes(h, c, cb);
            std::vector<Slot> sub;
            for (int i = 0; i < 8; i++)
                if (h.inner[c].link[i] != MP_LINK_NULL) sub.push_back(Slot{cb[i], h.inner[c].link[i]});
            slots.erase(slots.begin() + pick);
            slots.insert(slots.begin() + pick, sub.begin(), sub.end());
            out.absorbed++;
        }
        const uint32_t nchild = static_cast<uint32_t>(slots.size());
        for (int i = 0; i < 8; i++) {
            if (static_cast<size_t>(i) < slots.size()) {
                const Slot& s = slots[static_cast<size_t>(i)];
                if (!ordered(s.box)) out.boxes_ordered = false;
                put_record(out.nodes, me, i, s.box, (s.ref & 7u) == 0u ? MP_LINK_NULL /* patched when the child is numbered */ : leaf_link(s.ref));
            } else {
                put_record(out.nodes, me, i, Box3{}, MP_LINK_NULL);
            }
        }
        std::memcpy(&out.nodes[static_cast<size_t>(me) * 64 + 7], &nchild, 4);
        for (size_t i = slots.size(); i-- > 0;)  // reversed: the first inner child is numbered next
            if ((slots[i].ref & 7u) == 0u) todo.push_back(Work{slots[i].ref >> 3, me, static_cast<int>(i)});
    }
    out.nodes.resize(static_cast<size_t>(out.count) * 64 + 16, 0.0f);  // tail padding
    out.stack_bound = stack_bound(out.nodes, out.count, out.root);
    put_root_record();
    return MP_OK;
}

}  // namespace mp

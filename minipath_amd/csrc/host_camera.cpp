// Host-side mirror of src/camera.rs (Camera -> CameraSampler) and src/screen_block.rs (tile ordering).
// nalgebra's Isometry3 / UnitQuaternion arithmetic is restated from the published algorithm (nalgebra 0.33.2 is
// not in the reference tree).  A Rust caller does not need these: it passes the CameraSampler the reference itself
// computed (mp_camera_sampler is exactly camera.rs:26-39).
#include <algorithm>
#include <cmath>
#include <cstring>

#include "mp_internal.h"

namespace mp {
namespace {

struct V3 {
    float x, y, z;
};
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float norm(V3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
inline V3 normalized(V3 a) {
    float n = norm(a);
    return {a.x / n, a.y / n, a.z / n};
}
inline V3 neg(V3 a) { return {-a.x, -a.y, -a.z}; }

struct Quat {
    float i, j, k, w;
};

// UnitQuaternion * Vector3: t = 2 (qv x v); v' = t*w + (qv x t) + v
inline V3 rotate(const Quat& q, V3 v) {
    V3 qv{q.i, q.j, q.k};
    V3 t = cross(qv, v);
    t = {t.x * 2.0f, t.y * 2.0f, t.z * 2.0f};
    V3 c = cross(qv, t);
    return {t.x * q.w + c.x + v.x, t.y * q.w + c.y + v.y, t.z * q.w + c.z + v.z};
}

// UnitQuaternion::from_rotation_matrix; columns of the matrix are x, y, z
Quat from_basis(V3 x, V3 y, V3 z) {
    const float m00 = x.x, m01 = y.x, m02 = z.x;
    const float m10 = x.y, m11 = y.y, m12 = z.y;
    const float m20 = x.z, m21 = y.z, m22 = z.z;
    float tr = m00 + m11 + m22;
    Quat q;
    if (tr > 0.0f) {
        float d = std::sqrt(tr + 1.0f) * 2.0f;
        q = {(m21 - m12) / d, (m02 - m20) / d, (m10 - m01) / d, 0.25f * d};
    } else if (m00 > m11 && m00 > m22) {
        float d = std::sqrt(1.0f + m00 - m11 - m22) * 2.0f;
        q = {0.25f * d, (m01 + m10) / d, (m02 + m20) / d, (m21 - m12) / d};
    } else if (m11 > m22) {
        float d = std::sqrt(1.0f + m11 - m00 - m22) * 2.0f;
        q = {(m01 + m10) / d, 0.25f * d, (m12 + m21) / d, (m02 - m20) / d};
    } else {
        float d = std::sqrt(1.0f + m22 - m00 - m11) * 2.0f;
        q = {(m02 + m20) / d, (m12 + m21) / d, 0.25f * d, (m10 - m01) / d};
    }
    return q;
}

// Isometry3::look_at_rh(eye, target, up).inverse()  (camera.rs:94-97)
void look_at_inverse(V3 eye, V3 target, V3 up, Quat& q_out, V3& t_out) {
    V3 dir = neg({target.x - eye.x, target.y - eye.y, target.z - eye.z});
    V3 z = normalized(dir);
    V3 x = normalized(cross(up, z));
    V3 y = normalized(cross(z, x));
    Quat f = from_basis(x, y, z);
    Quat view{-f.i, -f.j, -f.k, f.w};          // face_towards(..).inverse()
    V3 vt = rotate(view, neg(eye));            // translation of the view isometry
    q_out = {-view.i, -view.j, -view.k, view.w};  // inverse rotation
    t_out = rotate(q_out, neg(vt));               // inverse translation
}

}  // namespace

void camera_default(mp_camera& c) {  // camera.rs:42-52
    c.q[0] = c.q[1] = c.q[2] = 0.0f;
    c.q[3] = 1.0f;
    c.t[0] = c.t[1] = c.t[2] = 0.0f;
    c.focus_distance = INFINITY;
    c.sensor_is_width = 0;
    c.sensor_size = 24e-3f;
    c.focal_length = 50e-3f;
    c.f_number = 9.0f;
}

void camera_look_at(mp_camera& c, const float eye[3], const float at[3], const float up[3]) {  // :93-101
    Quat q;
    V3 t;
    look_at_inverse({eye[0], eye[1], eye[2]}, {at[0], at[1], at[2]}, {up[0], up[1], up[2]}, q, t);
    c.q[0] = q.i; c.q[1] = q.j; c.q[2] = q.k; c.q[3] = q.w;
    c.t[0] = t.x; c.t[1] = t.y; c.t[2] = t.z;
    c.focus_distance = norm({at[0] - eye[0], at[1] - eye[1], at[2] - eye[2]});
}

void camera_look_direction(mp_camera& c, const float eye[3], const float fwd[3], const float up[3]) {  // :104-116
    Quat q;
    V3 t;
    look_at_inverse({eye[0], eye[1], eye[2]}, {eye[0] + fwd[0], eye[1] + fwd[1], eye[2] + fwd[2]},
                    {up[0], up[1], up[2]}, q, t);
    c.q[0] = q.i; c.q[1] = q.j; c.q[2] = q.k; c.q[3] = q.w;
    c.t[0] = t.x; c.t[1] = t.y; c.t[2] = t.z;
}

void camera_basis(const mp_camera& c, float center[3], float fwd[3], float up[3], float right[3]) {  // :148-171
    Quat q{c.q[0], c.q[1], c.q[2], c.q[3]};
    V3 o = rotate(q, {0, 0, 0});
    V3 f = rotate(q, {0, 0, -1}), u = rotate(q, {0, 1, 0}), r = rotate(q, {1, 0, 0});
    center[0] = o.x + c.t[0]; center[1] = o.y + c.t[1]; center[2] = o.z + c.t[2];
    fwd[0] = f.x; fwd[1] = f.y; fwd[2] = f.z;
    up[0] = u.x; up[1] = u.y; up[2] = u.z;
    right[0] = r.x; right[1] = r.y; right[2] = r.z;
}

void camera_build_sampler(const mp_camera& c, uint32_t w, uint32_t h, mp_camera_sampler& out) {  // :123-146
    float center[3], fwd[3], up[3], right[3];
    camera_basis(c, center, fwd, up, right);
    float rx = static_cast<float>(w), ry = static_cast<float>(h);
    float ps = c.sensor_is_width ? c.sensor_size / rx : c.sensor_size / ry;
    float uvx = ((rx - 1.0f) * ps) / 2.0f;
    float uvy = ((ry - 1.0f) * ps) / 2.0f;
    for (int k = 0; k < 3; k++) {
        out.center[k] = center[k];
        out.up[k] = up[k];
        out.right[k] = right[k];
        out.film_origin_offset[k] = (-fwd[k]) * c.focal_length + right[k] * uvx - up[k] * uvy;
    }
    out.pixel_scale = ps;
    out.lens_radius = c.focal_length / (2.0f * c.f_number);
    out.lens_weight = c.focal_length / c.focus_distance;
}

// screen_block.rs:46-81 (+ divide_range :144-160)
std::vector<mp_block> tile_ordering(mp_block b, uint32_t ts, uint64_t shuffle_seed) {
    std::vector<mp_block> tiles;
    if (!(b.min_x < b.max_x && b.min_y < b.max_y) || ts == 0) return tiles;  // is_empty :10-12
    auto divide = [ts](uint32_t start, uint32_t end) {
        std::vector<std::pair<uint32_t, uint32_t>> r;
        uint32_t total = end - start, full = total / ts;
        uint32_t n = full + (full * ts != total ? 1u : 0u);
        for (uint32_t i = 0; i < n; i++) {
            uint32_t s = start + i * ts;
            r.emplace_back(s, std::min(end, s + ts));
        }
        return r;
    };
    auto xs = divide(b.min_x, b.max_x), ys = divide(b.min_y, b.max_y);
    tiles.reserve(xs.size() * ys.size());
    for (auto& y : ys)
        for (auto& x : xs) tiles.push_back({x.first, y.first, x.second, y.second});
    if (shuffle_seed != 0) {
        // centre-out with exponential noise of mean 0.1*|centre| (:61-62,74-78).  The reference draws the noise from
        // an OS-seeded thread RNG, so only the distribution -- not a particular order -- is defined.
        float cx = static_cast<float>((b.min_x + b.max_x) / 2), cy = static_cast<float>((b.min_y + b.max_y) / 2);
        float scale = std::sqrt(cx * cx + cy * cy) * 0.1f;
        uint64_t s = shuffle_seed;
        auto next01 = [&s]() {  // SplitMix64 -> [0,1)
            s += 0x9e3779b97f4a7c15ull;
            uint64_t z = s;
            z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
            z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
            z ^= z >> 31;
            return static_cast<float>(z >> 40) * (1.0f / 16777216.0f);
        };
        std::vector<std::pair<float, size_t>> keys(tiles.size());
        for (size_t i = 0; i < tiles.size(); i++) {
            float tx = static_cast<float>((tiles[i].min_x + tiles[i].max_x) / 2);
            float ty = static_cast<float>((tiles[i].min_y + tiles[i].max_y) / 2);
            float dx = cx - tx, dy = cy - ty;
            keys[i] = {std::sqrt(dx * dx + dy * dy) - std::log(1.0f - next01()) * scale, i};
        }
        std::stable_sort(keys.begin(), keys.end(), [](auto& a, auto& b) { return a.first < b.first; });
        std::vector<mp_block> sorted(tiles.size());
        for (size_t i = 0; i < tiles.size(); i++) sorted[i] = tiles[keys[i].second];
        tiles.swap(sorted);
    }
    return tiles;
}

}  // namespace mp

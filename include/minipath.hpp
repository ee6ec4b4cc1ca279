// minipath.hpp -- header-only C++17 mirror of the reference's Rust API over the C ABI of minipath_hip.h.
//
// The reference's seam is `src/renderer` + `Camera` + `Scene<TriangleBvh>`; a Rust caller binds the C ABI through
// the shim of INTEGRATION.md.  No Rust toolchain exists in this pipeline, so this header is the compiled-language host side: the
// same names, argument meaning and error behaviour, for C++ callers (examples/render_teapot.cpp is benches/render_teapot.rs in
// these terms).  Nothing here computes: every call forwards to libminipath_hip.so.
//
//   reference (Rust)                                          here
//   Camera::default().look_at(e, a, u).f_number(x)...         minipath::Camera().look_at(e, a, u).f_number(x)...      camera.rs:42-121
//   TriangleBvh::with_obj(path)? / Scene { object }           minipath::Scene::with_obj(ctx, path)                    building.rs:28-34
//   RenderSettings { tile_size, sample_count, resolution }    minipath::RenderSettings{tile_size, sample_count, {w, h}}  renderer/mod.rs:7-13
//   render(scene, camera, settings, started, finished)?       minipath::render(ctx, scene, camera, settings, started, finished)  machinery.rs:20-30
//   RenderProgress::{progress, is_finished, elapsed, abort,   minipath::RenderProgress, same members; image() copies under the lock
//                    wait, image}                                                                                     machinery.rs:131-178
//
// Errors: the reference returns anyhow::Result from with_obj / render and panics elsewhere; here every failing call throws
// minipath::Error (code = MP_ERR_*, what() = mp_last_error()).  Callbacks run on the library's worker thread(s), once per tile
// start and once per tile end, like the reference's (machinery.rs:75,93-99).
#pragma once

#include <array>
#include <chrono>
#include <cstdint>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "minipath_hip.h"

namespace minipath {

struct Error : std::runtime_error {
    int code;
    Error(int c, const char* msg) : std::runtime_error(msg ? msg : "minipath error"), code(c) {}
};
inline void check(int rc) {
    if (rc != MP_OK) throw Error(rc, mp_last_error());
}

using ScreenBlock = mp_block;                       // geometry/mod.rs:15
using RenderProgressSnapshot = mp_progress;         // machinery.rs:180-189
using Vec3 = std::array<float, 3>;

// One GPU (the reference's worker pool over cores becomes a context per device).
class Context {
public:
    explicit Context(int device_id = 0) { check(mp_ctx_create(device_id, &p_)); }
    ~Context() { mp_ctx_destroy(p_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    mp_ctx* get() const { return p_; }
    void set_option(const char* key, int value) { check(mp_ctx_set_option(p_, key, value)); }

private:
    mp_ctx* p_ = nullptr;
};

// camera.rs:9-121 -- the builder methods return *this by value like the reference's `self -> Self`
class Camera {
public:
    Camera() { check(mp_camera_default(&c_)); }  // Camera::default() :42-52
    Camera look_at(const Vec3& eye, const Vec3& at, const Vec3& up) const {
        Camera r = *this;
        check(mp_camera_look_at(&r.c_, eye.data(), at.data(), up.data()));
        return r;
    }
    Camera look_direction(const Vec3& eye, const Vec3& forward, const Vec3& up) const {
        Camera r = *this;
        check(mp_camera_look_direction(&r.c_, eye.data(), forward.data(), up.data()));
        return r;
    }
    Camera translate(const Vec3& t) const {
        Camera r = *this;
        check(mp_camera_translate(&r.c_, t.data()));
        return r;
    }
    Camera f_number(float v) const { Camera r = *this; r.c_.f_number = v; return r; }
    Camera focus_distance(float v) const { Camera r = *this; r.c_.focus_distance = v; return r; }
    Camera focal_length(float v) const { Camera r = *this; r.c_.focal_length = v; return r; }
    mp_camera_sampler build_sampler(uint32_t width, uint32_t height) const {  // :123-146
        mp_camera_sampler s;
        check(mp_camera_build_sampler(&c_, width, height, &s));
        return s;
    }
    const mp_camera& raw() const { return c_; }

private:
    mp_camera c_{};
};

// renderer/mod.rs:7-13 (+ the build-defined seed of the seeded mode, minipath_hip.h)
struct RenderSettings {
    uint32_t tile_size = 64;
    uint32_t sample_count = 1;
    std::array<uint32_t, 2> resolution{0, 0};
    uint64_t seed = 0;
    bool keep_f32 = false;  // also keep the pre-quantisation f32 means on the host (RenderProgress::image_f32); the reference has the u8 image only
    mp_settings raw() const {
        mp_settings s{};
        s.tile_size = tile_size; s.sample_count = sample_count; s.width = resolution[0]; s.height = resolution[1]; s.seed = seed;
        s.flags = keep_f32 ? 0u : MP_FLAG_IMAGE_U8_ONLY;
        return s;
    }
};

// Scene { object: TriangleBvh } (scene/mod.rs, triangle_bvh/building.rs:28-107)
class Scene {
public:
    static Scene with_obj(const Context& ctx, const std::string& path) {
        Scene s;
        check(mp_scene_from_obj(ctx.get(), path.c_str(), &s.p_));
        return s;
    }
    static Scene build(const Context& ctx, const std::vector<float>& positions, const std::vector<float>& normals,
                       const std::vector<float>& tex, const std::vector<uint32_t>& indices) {
        Scene s;
        check(mp_scene_from_triangles(ctx.get(), positions.data(), normals.empty() ? nullptr : normals.data(), tex.empty() ? nullptr : tex.data(),
                                      static_cast<uint32_t>(positions.size() / 3), indices.data(), static_cast<uint32_t>(indices.size() / 3), &s.p_));
        return s;
    }
    Scene(Scene&& o) noexcept : p_(o.p_) { o.p_ = nullptr; }
    Scene& operator=(Scene&& o) noexcept { std::swap(p_, o.p_); return *this; }
    Scene(const Scene&) = delete;
    Scene& operator=(const Scene&) = delete;
    ~Scene() { if (p_) mp_scene_destroy(p_); }
    mp_scene_info info() const { mp_scene_info i; check(mp_scene_info_get(p_, &i)); return i; }
    const mp_scene* get() const { return p_; }

private:
    Scene() = default;
    mp_scene* p_ = nullptr;
};

// machinery.rs:131-178
class RenderProgress {
public:
    RenderProgress(RenderProgress&& o) noexcept : r_(o.r_), cb_(std::move(o.cb_)), w_(o.w_), h_(o.h_) { o.r_ = nullptr; }
    RenderProgress& operator=(RenderProgress&& o) noexcept {
        std::swap(r_, o.r_); std::swap(cb_, o.cb_); std::swap(w_, o.w_); std::swap(h_, o.h_);
        return *this;
    }
    RenderProgress(const RenderProgress&) = delete;
    RenderProgress& operator=(const RenderProgress&) = delete;
    ~RenderProgress() { if (r_) { mp_render_wait(r_); mp_render_destroy(r_); } }  // (going out of scope joins the workers: the callbacks must outlive the render)
    RenderProgressSnapshot progress() const { mp_progress p; check(mp_render_progress(r_, &p)); return p; }
    bool is_finished() const { int f = 0; check(mp_render_is_finished(r_, &f)); return f != 0; }
    std::chrono::nanoseconds elapsed() const { uint64_t ns = 0; check(mp_render_elapsed_ns(r_, &ns)); return std::chrono::nanoseconds(ns); }
    void abort() { check(mp_render_abort(r_)); }
    void wait() { check(mp_render_wait(r_)); }
    // RgbaImage (u8, row-major, x fastest), copied under the image's lock
    std::vector<uint8_t> image() const {
        std::vector<uint8_t> out(static_cast<size_t>(w_) * h_ * 4);
        check(mp_render_image_u8(r_, out.data()));
        return out;
    }
    // the pre-quantisation means of worker.rs:44 (RenderSettings::keep_f32)
    std::vector<float> image_f32() const {
        std::vector<float> out(static_cast<size_t>(w_) * h_ * 4);
        check(mp_render_image_f32(r_, out.data()));
        return out;
    }

private:
    struct Callbacks {
        std::function<void(ScreenBlock)> started;
        std::function<void(ScreenBlock, RenderProgressSnapshot)> finished;
    };
    template <class F1, class F2>
    friend RenderProgress render(const Context&, const Scene&, const Camera&, const RenderSettings&, F1, F2);
    RenderProgress() = default;
    mp_render* r_ = nullptr;
    std::unique_ptr<Callbacks> cb_;
    uint32_t w_ = 0, h_ = 0;
};

// render() machinery.rs:20-30 : asynchronous; returns after the worker thread(s) started
template <class F1, class F2>
RenderProgress render(const Context& ctx, const Scene& scene, const Camera& camera, const RenderSettings& settings, F1 started, F2 finished) {
    RenderProgress rp;
    rp.cb_ = std::make_unique<RenderProgress::Callbacks>();
    rp.cb_->started = std::move(started);
    rp.cb_->finished = std::move(finished);
    rp.w_ = settings.resolution[0];
    rp.h_ = settings.resolution[1];
    const mp_settings st = settings.raw();
    auto on_start = [](void* user, mp_block b) { static_cast<RenderProgress::Callbacks*>(user)->started(b); };
    auto on_finish = [](void* user, mp_block b, mp_progress p) { static_cast<RenderProgress::Callbacks*>(user)->finished(b, p); };
    check(mp_render_begin(ctx.get(), scene.get(), &camera.raw(), &st, on_start, on_finish, rp.cb_.get(), &rp.r_));
    return rp;
}

}  // namespace minipath

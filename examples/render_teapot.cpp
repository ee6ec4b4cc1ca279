// benches/render_teapot.rs of the reference, in C++ over include/minipath.hpp (the C-ABI mirror of the reference's Rust API):
// Camera::default().look_at((0,2,10),(0,1.5,0),(0,1,0)).f_number(4.8).focus_distance(10), RenderSettings{64, 10, 2048x1536},
// Scene{TriangleBvh::with_obj(data/teapot.obj)}, render(..., |_| {}, |_, _| {}) + wait(), timed.
// usage: render_teapot [teapot.obj] [width height spp iterations [out.ppm]]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "minipath.hpp"

int main(int argc, char** argv) {
    const std::string obj = argc > 1 ? argv[1] : "tests/golden/teapot.obj";
    const uint32_t w = argc > 2 ? static_cast<uint32_t>(std::atoi(argv[2])) : 2048, h = argc > 3 ? static_cast<uint32_t>(std::atoi(argv[3])) : 1536;
    const uint32_t spp = argc > 4 ? static_cast<uint32_t>(std::atoi(argv[4])) : 10;
    const int iterations = argc > 5 ? std::atoi(argv[5]) : 5;
    try {
        minipath::Context ctx(0);
        const minipath::Camera camera = minipath::Camera().look_at({0.0f, 2.0f, 10.0f}, {0.0f, 1.5f, 0.0f}, {0.0f, 1.0f, 0.0f}).f_number(4.8f).focus_distance(10.0f);
        minipath::RenderSettings settings;
        settings.tile_size = 64;
        settings.sample_count = spp;
        settings.resolution = {w, h};
        settings.seed = 0x5EED;
        const minipath::Scene scene = minipath::Scene::with_obj(ctx, obj);
        const mp_scene_info info = scene.info();
        std::printf("scene: %u triangles, %u inner nodes, %u packets\n", info.triangle_count, info.inner_count, info.packet_count);
        std::vector<uint8_t> image;
        for (int it = 0; it < iterations; it++) {
            const auto t0 = std::chrono::steady_clock::now();
            minipath::RenderProgress progress = minipath::render(ctx, scene, camera, settings, [](minipath::ScreenBlock) {}, [](minipath::ScreenBlock, minipath::RenderProgressSnapshot) {});
            progress.wait();
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            const minipath::RenderProgressSnapshot p = progress.progress();
            std::printf("render_teapot %ux%u x%u: %.2f ms wall, elapsed() %.2f ms, %zu / %zu tiles, %.1f Mrays/s\n", w, h, spp, ms,
                        std::chrono::duration<double, std::milli>(progress.elapsed()).count(), p.finished, p.total, static_cast<double>(w) * h * spp / ms * 1e-3);
            image = progress.image();
        }
        uint64_t fnv = 1469598103934665603ull;  // FNV-1a of the u8 image: a test compares it with the Python mirror's frame
        for (uint8_t b : image) { fnv ^= b; fnv *= 1099511628211ull; }
        std::printf("image fnv1a %016llx\n", static_cast<unsigned long long>(fnv));
        if (argc > 6) {
            if (FILE* f = std::fopen(argv[6], "wb")) {
                std::fprintf(f, "P6\n%u %u\n255\n", w, h);
                for (size_t i = 0; i < static_cast<size_t>(w) * h; i++) std::fwrite(&image[i * 4], 1, 3, f);
                std::fclose(f);
            }
        }
    } catch (const minipath::Error& e) {
        std::fprintf(stderr, "error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}

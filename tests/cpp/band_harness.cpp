// Config C4's exchange step from compiled host code (what a Rust host would bind, INTEGRATION.md "multi-GPU"): ONE process drives
// `ranks` contexts through the C ABI only.
//   band_harness <scene.b32scene> <out.rgba> <ranks> [frames]
// Rank 0 (the root) owns the framebuffer; every other context is attached to it (b32_band_attach: the same binding b32_band_import makes
// across processes), owns a band of rows (b32_set_band) and its own resident copy of the mesh.  Per frame every rank clears and draws its
// band -- straight into the root's memory -- and publishes the frame number; the root's stream waits for all of them before the
// download.  The scene file and the expected hash are the ones of tests/golden; nothing here touches Python or torch.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <memory>
#include <vector>

#include "scenefile.hpp"

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    try {
        const b32::SceneFile sc = b32::read_scene(argv[1]);
        const uint32_t ranks = (uint32_t)std::atoi(argv[3]);
        const uint32_t frames = argc > 4 ? (uint32_t)std::atoi(argv[4]) : 3u;
        if (sc.fmt8 || ranks < 1 || ranks > 63) return 2;
        b32::Framebuffer fb(sc.width, sc.height);                        // the root: Framebuffer::new
        std::vector<b32_ctx*> ctx{ fb.ctx() };
        for (uint32_t r = 1; r < ranks; ++r) {
            b32_ctx* c = nullptr;
            b32::check(b32_create(0, &c), "b32_create");
            b32::check(b32_band_attach(c, fb.ctx(), r), "b32_band_attach");
            ctx.push_back(c);
        }
        // rows [y0, y1) of rank r: a balanced contiguous partition (bonnie32_amd/bands.py)
        auto band = [&](uint32_t r, uint32_t& y0, uint32_t& y1) {
            const uint32_t base = (uint32_t)sc.height / ranks, extra = (uint32_t)sc.height % ranks;
            y0 = r * base + (r < extra ? r : extra); y1 = y0 + base + (r < extra ? 1u : 0u);
        };
        // the packed mesh, uploaded once per rank (b32_scene_upload: every rank holds the whole mesh, transform and cull are replicated)
        std::vector<B32Vertex> v; std::vector<B32Face> f;
        for (const auto& x : sc.vertices) v.push_back(b32::detail::pack(x));
        for (const auto& x : sc.faces) f.push_back(b32::detail::pack(x));
        std::vector<B32Texture15> tex;
        for (const auto& t : sc.textures) tex.push_back({ (uint32_t)t.width, (uint32_t)t.height, (uint32_t)t.blend_mode, 0, t.pixels.size() >= t.width * t.height ? t.pixels.data() : nullptr });
        const B32Camera cam = b32::detail::pack(sc.camera);
        const std::vector<B32Light> lights = b32::detail::pack(sc.settings.lights);
        const B32Settings st = b32::detail::pack(sc.settings, lights);
        B32Fog fog{}; const bool has_fog = b32::detail::pack(sc.fog, fog);
        for (uint32_t r = 0; r < ranks; ++r) {
            uint32_t y0, y1; band(r, y0, y1);
            b32::check(b32_set_band(ctx[r], y0, y1), "b32_set_band");
            b32::check(b32_scene_upload(ctx[r], v.data(), (uint32_t)v.size(), f.data(), (uint32_t)f.size(), tex.data(), (uint32_t)tex.size()), "b32_scene_upload");
        }
        uint32_t drawn = 0;
        for (uint32_t n = 1; n <= frames; ++n) {
            for (uint32_t r = ranks; r-- > 0;) {                         // (band ranks first, the root last: nothing depends on the order)
                if (r && n > 1) b32::check(b32_band_acquire(ctx[r], n - 1, 10000000u), "b32_band_acquire");
                b32::check(b32_fb_clear(ctx[r], sc.clear.r, sc.clear.g, sc.clear.b, (uint8_t)sc.clear.blend), "b32_fb_clear");
                b32::check(b32_render_scene_15_async(ctx[r], &cam, &st, has_fog ? &fog : nullptr), "b32_render_scene_15_async");
                if (r) b32::check(b32_band_publish(ctx[r], n), "b32_band_publish");
            }
            for (uint32_t r = 1; r < ranks; ++r) b32::check(b32_band_wait(ctx[0], r, n, 10000000u), "b32_band_wait");
            B32Timings tm{};
            b32::check(b32_frame_finish(ctx[0], &tm), "b32_frame_finish");          // the root's stream: its band + the waits
            drawn = tm.triangles_drawn;
            if (n < frames) b32::check(b32_band_release(ctx[0], n), "b32_band_release");
        }
        const std::vector<uint8_t> px = fb.pixels();
        uint32_t timeouts = 0;
        b32::check(b32_band_status(ctx[0], nullptr, nullptr, &timeouts), "b32_band_status");
        std::ofstream o(argv[2], std::ios::binary);
        o.write(reinterpret_cast<const char*>(px.data()), (std::streamsize)px.size());
        std::printf("triangles_drawn %u\nranks %u frames %u timeouts %u\n", drawn, ranks, frames, timeouts);
        for (uint32_t r = 1; r < ranks; ++r) { B32Timings t{}; b32::check(b32_frame_finish(ctx[r], &t), "b32_frame_finish(rank)"); b32_destroy(ctx[r]); }
        return timeouts ? 11 : 0;
    } catch (const b32::Error& e) {
        std::fprintf(stderr, "b32::Error %d: %s\n", e.code, e.what());
        return 10;
    }
}

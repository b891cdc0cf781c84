// A frame of several meshes through the C++ host mirror (b32::ResidentMesh + b32::render_frame = b32_frame_begin / _add_scene / _end):
//   frame_harness <out.rgba> <out.zbuffer> <scene0.b32scene> <scene1.b32scene> ...
// Camera, base settings, lights and the framebuffer (size, clear colour) come from the first file; every file contributes its mesh
// and the per-mesh parameters the reference's callers vary (scene.rs:158-261): ambient, backface_cull, fog.
#include <cstdio>
#include <fstream>
#include <memory>
#include <vector>

#include <cstring>
#include "scenefile.hpp"

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    try {
        std::vector<b32::SceneFile> files;
        for (int i = 3; i < argc; ++i) files.push_back(b32::read_scene(argv[i]));
        const b32::SceneFile& first = files[0];
        b32::Framebuffer fb(first.width, first.height);
        std::vector<std::unique_ptr<b32::ResidentMesh>> meshes;
        std::vector<std::pair<const b32::ResidentMesh*, b32::MeshParams>> frame;
        for (const auto& f : files) {
            meshes.push_back(std::make_unique<b32::ResidentMesh>(fb, f.vertices, f.faces, f.textures));
            frame.push_back({ meshes.back().get(), b32::MeshParams{ f.settings.ambient, f.settings.backface_cull, false, f.fog } });
        }
        fb.clear(first.clear);
        b32::RasterSettings base = first.settings;
        base.backface_wireframe = false;
        const b32::RasterTimings tm = b32::render_frame(fb, frame, first.camera, base);
        const std::vector<uint8_t> px = fb.pixels();
        std::ofstream(argv[1], std::ios::binary).write(reinterpret_cast<const char*>(px.data()), (std::streamsize)px.size());
        std::vector<float> z((size_t)first.width * first.height);
        b32::check(b32_zbuffer_download(fb.ctx(), z.data()), "zbuffer");
        std::ofstream(argv[2], std::ios::binary).write(reinterpret_cast<const char*>(z.data()), (std::streamsize)(z.size() * 4));
        const unsigned long long merged = b32_batch_count(fb.ctx(), 0);
        // the same frame three times through the console loop (b32::FrameLoop: clear + b32_frame_submit + b32_fb_download_async + tickets), the
        // presenter one frame behind: every delivered frame must be the frame render_frame produced
        unsigned presented = 0;
        {
            b32::FrameLoop loop(fb);
            uint64_t prev = 0;
            for (int i = 0; i < 3; ++i) {
                const uint64_t t = loop.submit(first.clear, frame, first.camera, base);
                if (prev) { if (std::memcmp(loop.wait(prev), px.data(), px.size()) != 0) return 11; ++presented; }
                prev = t;
            }
            if (std::memcmp(loop.wait(prev), px.data(), px.size()) != 0) return 11;
            ++presented;
            loop.finish();
        }
        std::printf("triangles_drawn %u merged_draws %llu presented_frames %u\n", tm.triangles_drawn, merged, presented);
    } catch (const b32::Error& e) {
        std::fprintf(stderr, "b32::Error %d: %s\n", e.code, e.what());
        return 10;
    }
    return 0;
}

// Drives the C++ host mirror of the reference interface (bonnie-32_amd/host/rasterizer.hpp) end to end FROM A FILE:
//   mesh_harness <scene.b32scene> <out.rgba> [out.zbuffer]
// The .b32scene file (bonnie-32_amd/scenefile.py; host/scenefile.hpp) holds everything one render_mesh_15 / render_mesh call takes --
// vertices, faces, textures, camera, every RasterSettings field, lights, fog -- and the framebuffer it draws into.  The harness builds
// the reference-shaped objects (b32::Vertex, b32::Face, b32::Texture15 / b32::Texture, b32::RasterSettings, ...), calls
// Framebuffer::new + Framebuffer::clear + render_mesh_15 (or render_mesh for an 8-bit-colour file) through the C ABI on the GPU and
// dumps fb.pixels (and the depth buffer).  The same file is what tests/rust/pin_oracle feeds to the reference itself.
#include <cstdio>
#include <fstream>
#include <vector>

#include "scenefile.hpp"

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    try {
        const b32::SceneFile sc = b32::read_scene(argv[1]);
        b32::Framebuffer fb(sc.width, sc.height);
        fb.clear(sc.clear);
        const b32::RasterTimings tm = sc.fmt8 ? b32::render_mesh(fb, sc.vertices, sc.faces, sc.textures8, sc.camera, sc.settings)
                                              : b32::render_mesh_15(fb, sc.vertices, sc.faces, sc.textures, sc.camera, sc.settings, sc.fog);
        const std::vector<uint8_t> px = fb.pixels();
        std::ofstream o(argv[2], std::ios::binary);
        o.write(reinterpret_cast<const char*>(px.data()), (std::streamsize)px.size());
        if (argc > 3) {
            std::vector<float> z((size_t)sc.width * sc.height);
            b32::check(b32_zbuffer_download(fb.ctx(), z.data()), "zbuffer");
            std::ofstream oz(argv[3], std::ios::binary);
            oz.write(reinterpret_cast<const char*>(z.data()), (std::streamsize)(z.size() * 4));
        }
        std::printf("triangles_drawn %u\n", tm.triangles_drawn);
        if (sc.has_expect) std::printf("expect_triangles_drawn %u\n", sc.expect_triangles);
    } catch (const b32::Error& e) {
        std::fprintf(stderr, "b32::Error %d: %s\n", e.code, e.what());
        return 10;
    }
    return 0;
}

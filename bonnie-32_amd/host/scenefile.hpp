// scenefile.hpp — reader of the `.b32scene` format (bonnie-32_amd/scenefile.py holds the layout) into the reference-shaped
// host types of rasterizer.hpp: one call of render_mesh_15 / render_mesh as a file.
#pragma once
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "rasterizer.hpp"

namespace b32 {

struct SceneFile {
    uint32_t width = 0, height = 0;
    bool fmt8 = false;                    // 8-bit-colour path: textures8 + render_mesh
    Color clear;
    Camera camera;
    RasterSettings settings;
    Fog fog;
    std::vector<Vertex> vertices;
    std::vector<Face> faces;
    std::vector<Texture15> textures;
    std::vector<Texture> textures8;
    bool has_expect = false;
    uint32_t expect_triangles = 0; uint64_t expect_fragments = 0; uint8_t expect_sha[32] = {}, expect_zsha[32] = {};
};

namespace detail {
struct Reader {
    std::vector<char> b; size_t o = 0;
    template <typename T> T get() { if (o + sizeof(T) > b.size()) throw Error(B32_E_ARG, "b32scene: truncated"); T v; std::memcpy(&v, b.data() + o, sizeof(T)); o += sizeof(T); return v; }
    void bytes(void* dst, size_t n) { if (o + n > b.size()) throw Error(B32_E_ARG, "b32scene: truncated"); std::memcpy(dst, b.data() + o, n); o += n; }
    Vec3 v3() { Vec3 v; v.x = get<float>(); v.y = get<float>(); v.z = get<float>(); return v; }
};
}  // namespace detail

inline SceneFile read_scene(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw Error(B32_E_ARG, "b32scene: cannot open");
    detail::Reader r;
    r.b.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    char magic[8]; r.bytes(magic, 8);
    if (std::memcmp(magic, "B32SCENE", 8) != 0) throw Error(B32_E_ARG, "b32scene: bad magic");
    const uint32_t version = r.get<uint32_t>(), flags = r.get<uint32_t>();
    if (version != 1) throw Error(B32_E_ARG, "b32scene: unknown version");
    SceneFile s;
    s.width = r.get<uint32_t>(); s.height = r.get<uint32_t>();
    const uint32_t nv = r.get<uint32_t>(), nf = r.get<uint32_t>(), nt = r.get<uint32_t>(), nl = r.get<uint32_t>();
    uint8_t c4[4]; r.bytes(c4, 4);
    s.clear = { c4[0], c4[1], c4[2], (BlendMode)c4[3] };
    r.o = 64;
    s.camera.position = r.v3(); s.camera.basis_x = r.v3(); s.camera.basis_y = r.v3(); s.camera.basis_z = r.v3();
    uint8_t st[12]; r.bytes(st, 12);
    RasterSettings& rs = s.settings;
    rs.affine_textures = st[0]; rs.use_zbuffer = st[1]; rs.shading = (ShadingMode)st[2]; rs.backface_cull = st[3]; rs.backface_wireframe = st[4];
    rs.dithering = st[5]; rs.wireframe_overlay = st[6]; rs.use_rgb555 = st[7]; rs.use_fixed_point = st[8]; rs.xray_mode = st[9];
    rs.ambient = r.get<float>();
    const Vec3 ortho = r.v3();
    if (flags & 4u) rs.ortho_projection = ortho;
    {
        const float start = r.get<float>(), falloff = r.get<float>(), cull = r.get<float>();
        r.bytes(c4, 4);
        if (flags & 2u) s.fog = std::make_tuple(start, falloff, cull, Color{ c4[0], c4[1], c4[2], (BlendMode)c4[3] });
    }
    for (uint32_t i = 0; i < nl; ++i) {
        Light l;
        l.type = r.get<uint32_t>(); l.position = r.v3(); l.direction = r.v3();
        l.radius = r.get<float>(); l.angle = r.get<float>(); l.intensity = r.get<float>();
        r.bytes(c4, 4);
        l.color = { c4[0], c4[1], c4[2], BlendMode::Opaque }; l.enabled = c4[3] != 0;
        rs.lights.push_back(l);
    }
    s.vertices.resize(nv);
    for (auto& v : s.vertices) {
        v.pos = r.v3(); v.uv.x = r.get<float>(); v.uv.y = r.get<float>(); v.normal = r.v3();
        r.bytes(c4, 4); v.color = { c4[0], c4[1], c4[2], (BlendMode)c4[3] };
    }
    s.faces.resize(nf);
    for (auto& fc : s.faces) {
        fc.v0 = r.get<uint32_t>(); fc.v1 = r.get<uint32_t>(); fc.v2 = r.get<uint32_t>();
        const uint32_t tex = r.get<uint32_t>();
        if (tex != B32_NO_TEXTURE) fc.texture_id = tex;
        r.bytes(c4, 4);
        fc.black_transparent = c4[0] != 0; fc.blend_mode = (BlendMode)c4[1]; fc.editor_alpha = c4[2];
    }
    s.fmt8 = (flags & 1u) != 0;
    for (uint32_t i = 0; i < nt; ++i) {
        const uint32_t w = r.get<uint32_t>(), h = r.get<uint32_t>(), bl = r.get<uint32_t>(), tb = r.get<uint32_t>();
        if (tb != (s.fmt8 ? 4u : 2u)) throw Error(B32_E_ARG, "b32scene: texel size");
        if (s.fmt8) {
            Texture t; t.width = w; t.height = h; t.blend_mode = (BlendMode)bl; t.pixels.resize((size_t)w * h);
            static_assert(sizeof(Color) == 4, "Color is r, g, b, blend");
            r.bytes(t.pixels.data(), (size_t)w * h * 4);
            s.textures8.push_back(std::move(t));
        } else {
            Texture15 t; t.width = w; t.height = h; t.blend_mode = (BlendMode)bl; t.pixels.resize((size_t)w * h);
            r.bytes(t.pixels.data(), (size_t)w * h * 2);
            s.textures.push_back(std::move(t));
        }
    }
    if (flags & 8u) {
        s.has_expect = true;
        s.expect_triangles = r.get<uint32_t>(); (void)r.get<uint32_t>(); s.expect_fragments = r.get<uint64_t>();
        r.bytes(s.expect_sha, 32); r.bytes(s.expect_zsha, 32);
    }
    if (r.o != r.b.size()) throw Error(B32_E_ARG, "b32scene: trailing bytes");
    return s;
}

}  // namespace b32

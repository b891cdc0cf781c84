// rasterizer.hpp — C++ host-side mirror of the reference rasterizer interface, header-only over the C ABI.
//
// The reference is compiled (Rust) code and no Rust toolchain exists in this image, so the host side above
// include/b32raster.h is written in C++ with the reference's names, argument meaning and error behaviour:
//
//   reference (src/rasterizer/...)                         here
//   Framebuffer::{new, resize, clear}  render.rs:18-45     b32::Framebuffer::{Framebuffer, resize, clear}
//   fb.pixels / fb.width / fb.height   render.rs:10-15     fb.pixels() (downloads RGBA8) / fb.width / fb.height
//   render_mesh_15(fb, vertices, faces, textures, camera,  b32::render_mesh_15(fb, vertices, faces, textures, camera,
//                  settings, fog) -> RasterTimings                            settings, fog) -> RasterTimings
//                                      render.rs:2302-2310
//   panics (index OOB render.rs:2375, NaN key :2531)       b32::Error{code}
//
// The Rust shim a maintainer would add to the reference is shown in INTEGRATION.md; it binds the same C symbols.
#pragma once
#include <cmath>
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "b32raster.h"

namespace b32 {

struct Error : std::runtime_error {
    int code;
    explicit Error(int c, const char* where) : std::runtime_error(std::string(where) + ": " + b32_strerror(c)), code(c) {}
};
inline void check(int rc, const char* where) { if (rc != B32_OK) throw Error(rc, where); }

// math.rs:9-13 / :90-94
struct Vec3 { float x = 0, y = 0, z = 0; };
struct Vec2 { float x = 0, y = 0; };

// types.rs:1380-1388, :1289-1294
enum class BlendMode : uint8_t { Opaque, Average, Add, Subtract, AddQuarter, Erase };
enum class ShadingMode : uint8_t { None, Flat, Gouraud };

// types.rs:721-726
struct Color {
    uint8_t r = 0, g = 0, b = 0; BlendMode blend = BlendMode::Opaque;
    static Color neutral() { return { 128, 128, 128, BlendMode::Opaque }; }       // Color::NEUTRAL types.rs:768
};

// types.rs:947-959
struct Vertex { Vec3 pos; Vec2 uv; Vec3 normal; Color color = Color::neutral(); };

// types.rs:984-1002 (+ constructors :1012-1036)
struct Face {
    size_t v0 = 0, v1 = 0, v2 = 0;
    std::optional<size_t> texture_id;
    bool black_transparent = true;
    BlendMode blend_mode = BlendMode::Opaque;
    uint8_t editor_alpha = 255;
    static Face with_texture(size_t a, size_t b, size_t c, size_t tex) { Face f; f.v0 = a; f.v1 = b; f.v2 = c; f.texture_id = tex; return f; }
};

// types.rs:532-539
struct Texture15 { size_t width = 0, height = 0; std::vector<uint16_t> pixels; BlendMode blend_mode = BlendMode::Opaque; };

// types.rs:1166-1176 — the 8-bit-colour path's texture: Color texels carry their own blend mode (Erase = transparent texel)
struct Texture { size_t width = 0, height = 0; std::vector<Color> pixels; BlendMode blend_mode = BlendMode::Opaque; };

// camera.rs:9-18 (basis vectors are inputs of the path)
struct Camera { Vec3 position; Vec3 basis_x{ 1, 0, 0 }, basis_y{ 0, 1, 0 }, basis_z{ 0, 0, 1 }; };

// types.rs:1297-1314
struct Light {
    uint32_t type = B32_LIGHT_DIRECTIONAL; Vec3 position, direction; float radius = 0, angle = 0;
    Color color{ 255, 255, 255, BlendMode::Opaque }; float intensity = 1.0f; bool enabled = true;
    // types.rs:1355-1369 (the direction is normalized at construction, math.rs:39-49)
    static Light spot(Vec3 position, Vec3 direction, float angle, float radius, float intensity) {
        Light l; l.type = B32_LIGHT_SPOT; l.position = position; l.angle = angle; l.radius = radius; l.intensity = intensity;
        const float len = std::sqrt(direction.x * direction.x + direction.y * direction.y + direction.z * direction.z);
        l.direction = len == 0.0f ? Vec3{ 0, 0, 0 } : Vec3{ direction.x / len, direction.y / len, direction.z / len };
        return l;
    }
};

// types.rs:1392-1428, defaults :1475-1495, game() :1455-1460
struct RasterSettings {
    bool affine_textures = true, use_zbuffer = true;
    ShadingMode shading = ShadingMode::Gouraud;
    bool backface_cull = true, backface_wireframe = true;
    std::vector<Light> lights;        // reference default: one directional (-1,-1,-1).normalize() * 0.7 — supplied by the caller
    float ambient = 0.3f;
    bool dithering = true, wireframe_overlay = false;
    std::optional<Vec3> ortho_projection;   // (zoom, center_x, center_y)
    bool use_rgb555 = true, use_fixed_point = true, xray_mode = false;
    static RasterSettings game() { RasterSettings s; s.backface_wireframe = false; return s; }
};

// types.rs:1499-1514
struct RasterTimings { float transform_ms = 0, fog_ms = 0, cull_ms = 0, sort_ms = 0, draw_ms = 0, wireframe_ms = 0; uint32_t triangles_drawn = 0; uint64_t fragments = 0; };

using Fog = std::optional<std::tuple<float, float, float, Color>>;   // render.rs:2309

// render.rs:10-45 — the pixels live in HBM; `pixels()` is the download the presenter performs (game/renderer.rs:179)
class Framebuffer {
public:
    size_t width = 0, height = 0;
    Framebuffer(size_t w, size_t h, int device = 0) {                       // Framebuffer::new, render.rs:18-25
        check(b32_create(device, &ctx_), "b32_create");
        check(b32_fb_new(ctx_, (uint32_t)w, (uint32_t)h), "Framebuffer::new"); width = w; height = h;
    }
    ~Framebuffer() { b32_destroy(ctx_); }
    Framebuffer(const Framebuffer&) = delete;
    Framebuffer& operator=(const Framebuffer&) = delete;
    void resize(size_t w, size_t h) { check(b32_fb_resize(ctx_, (uint32_t)w, (uint32_t)h), "Framebuffer::resize"); width = w; height = h; }
    void clear(Color c) { check(b32_fb_clear(ctx_, c.r, c.g, c.b, (uint8_t)c.blend), "Framebuffer::clear"); }
    std::vector<uint8_t> pixels() const { std::vector<uint8_t> p(width * height * 4); check(b32_fb_download(ctx_, p.data()), "fb.pixels"); return p; }
    void set_pixels(const std::vector<uint8_t>& p) { if (p.size() != width * height * 4) throw Error(B32_E_ARG, "set_pixels"); check(b32_fb_upload(ctx_, p.data()), "fb.pixels="); }
    b32_ctx* ctx() const { return ctx_; }
private:
    b32_ctx* ctx_ = nullptr;
};

namespace detail {
inline B32Vertex pack(const Vertex& v) {
    return { { v.pos.x, v.pos.y, v.pos.z }, { v.uv.x, v.uv.y }, { v.normal.x, v.normal.y, v.normal.z }, v.color.r, v.color.g, v.color.b, (uint8_t)v.color.blend };
}
inline B32Face pack(const Face& f) {
    // usize indices beyond u32 can never be valid vertex indices: map them to an out-of-range value (=> B32_E_INDEX)
    auto ix = [](size_t i) { return i > 0xFFFFFFFEull ? 0xFFFFFFFEu : (uint32_t)i; };
    const uint32_t tex = f.texture_id ? (*f.texture_id >= 0xFFFFFFFFull ? 0xFFFFFFFEu : (uint32_t)*f.texture_id) : B32_NO_TEXTURE;
    return { { ix(f.v0), ix(f.v1), ix(f.v2) }, tex, (uint8_t)f.black_transparent, (uint8_t)f.blend_mode, f.editor_alpha, 0 };
}
// Camera (camera.rs:9-18) -> B32Camera: position and the three basis vectors (rotation_x / rotation_y only feed Camera::update_basis)
inline B32Camera pack(const Camera& c) {
    return { { c.position.x, c.position.y, c.position.z }, { c.basis_x.x, c.basis_x.y, c.basis_x.z },
             { c.basis_y.x, c.basis_y.y, c.basis_y.z }, { c.basis_z.x, c.basis_z.y, c.basis_z.z } };
}
// Light (types.rs:1297-1314): LightType::Directional{direction} / Point{position, radius} / Spot{position, direction, angle, radius}
// flattened; the fields a kind does not have stay zero
inline B32Light pack(const Light& x) {
    return { x.type, { x.position.x, x.position.y, x.position.z }, { x.direction.x, x.direction.y, x.direction.z }, x.radius, x.angle,
             x.intensity, x.color.r, x.color.g, x.color.b, (uint8_t)x.enabled };
}
// RasterSettings (types.rs:1392-1428) -> B32Settings; `lights` must outlive the call (the struct points into it).
// ShadingMode None / Flat / Gouraud = 0 / 1 / 2 (types.rs:1289-1293); Option<OrthoProjection> -> has_ortho + three floats (types.rs:1432-1438);
// low_resolution and stretch_to_fill are presentation-only and not part of the C struct.
inline B32Settings pack(const RasterSettings& settings, const std::vector<B32Light>& lights) {
    B32Settings s{};
    s.affine_textures = settings.affine_textures; s.use_zbuffer = settings.use_zbuffer; s.shading = (uint8_t)settings.shading;
    s.backface_cull = settings.backface_cull; s.backface_wireframe = settings.backface_wireframe; s.dithering = settings.dithering;
    s.wireframe_overlay = settings.wireframe_overlay; s.use_rgb555 = settings.use_rgb555; s.use_fixed_point = settings.use_fixed_point;
    s.xray_mode = settings.xray_mode; s.has_ortho = settings.ortho_projection.has_value(); s.ambient = settings.ambient;
    if (settings.ortho_projection) { s.ortho_zoom = settings.ortho_projection->x; s.ortho_center_x = settings.ortho_projection->y; s.ortho_center_y = settings.ortho_projection->z; }
    s.n_lights = (uint32_t)lights.size(); s.lights = lights.empty() ? nullptr : lights.data();
    return s;
}
inline std::vector<B32Light> pack(const std::vector<Light>& lights) {
    std::vector<B32Light> l; l.reserve(lights.size());
    for (const auto& x : lights) l.push_back(pack(x));
    return l;
}
// fog: Option<(f32, f32, f32, Color)> (render.rs:2309) -> nullable B32Fog*
inline bool pack(const Fog& fog, B32Fog& out) {
    if (!fog) return false;
    const auto& [st, fo, cu, col] = *fog;
    out = { st, fo, cu, col.r, col.g, col.b, (uint8_t)col.blend };
    return true;
}
}  // namespace detail

// render.rs:2302-2310
inline RasterTimings render_mesh_15(Framebuffer& fb, const std::vector<Vertex>& vertices, const std::vector<Face>& faces,
                                    const std::vector<Texture15>& textures, const Camera& camera, const RasterSettings& settings,
                                    const Fog& fog = std::nullopt) {
    std::vector<B32Vertex> v; v.reserve(vertices.size());
    for (const auto& x : vertices) v.push_back(detail::pack(x));
    std::vector<B32Face> f; f.reserve(faces.size());
    for (const auto& x : faces) f.push_back(detail::pack(x));
    std::vector<B32Texture15> t; t.reserve(textures.size());
    for (const auto& x : textures)
        t.push_back({ (uint32_t)x.width, (uint32_t)x.height, (uint32_t)x.blend_mode, 0, x.pixels.size() >= x.width * x.height ? x.pixels.data() : nullptr });
    const std::vector<B32Light> l = detail::pack(settings.lights);
    const B32Camera c = detail::pack(camera);
    const B32Settings s = detail::pack(settings, l);
    B32Fog fg{};
    const B32Fog* fgp = detail::pack(fog, fg) ? &fg : nullptr;
    B32Timings tm{};
    check(b32_render_mesh_15(fb.ctx(), v.data(), (uint32_t)v.size(), f.data(), (uint32_t)f.size(), t.data(), (uint32_t)t.size(), &c, &s, fgp, &tm),
          "render_mesh_15");
    return { tm.transform_ms, tm.fog_ms, tm.cull_ms, tm.sort_ms, tm.draw_ms, tm.wireframe_ms, tm.triangles_drawn, tm.fragments };
}

// render.rs:1971-1978 — the path taken when settings.use_rgb555 is false (scene.rs:163-169); no fog parameter
inline RasterTimings render_mesh(Framebuffer& fb, const std::vector<Vertex>& vertices, const std::vector<Face>& faces,
                                 const std::vector<Texture>& textures, const Camera& camera, const RasterSettings& settings) {
    std::vector<B32Vertex> v; v.reserve(vertices.size());
    for (const auto& x : vertices) v.push_back(detail::pack(x));
    std::vector<B32Face> f; f.reserve(faces.size());
    for (const auto& x : faces) f.push_back(detail::pack(x));
    static_assert(sizeof(Color) == 4, "Color packs as r,g,b,blend bytes");
    std::vector<B32Texture> t; t.reserve(textures.size());
    for (const auto& x : textures)
        t.push_back({ (uint32_t)x.width, (uint32_t)x.height, (uint32_t)x.blend_mode, 0,
                      x.pixels.size() >= x.width * x.height && !x.pixels.empty() ? reinterpret_cast<const uint8_t*>(x.pixels.data()) : nullptr });
    const std::vector<B32Light> l = detail::pack(settings.lights);
    const B32Camera c = detail::pack(camera);
    const B32Settings s = detail::pack(settings, l);
    B32Timings tm{};
    check(b32_render_mesh(fb.ctx(), v.data(), (uint32_t)v.size(), f.data(), (uint32_t)f.size(), t.data(), (uint32_t)t.size(), &c, &s, &tm), "render_mesh");
    return { tm.transform_ms, tm.fog_ms, tm.cull_ms, tm.sort_ms, tm.draw_ms, tm.wireframe_ms, tm.triangles_drawn, tm.fragments };
}

// A mesh kept resident in HBM: uploaded once into a scene slot of the framebuffer's context, drawn many times.
class ResidentMesh {
public:
    ResidentMesh(Framebuffer& fb, const std::vector<Vertex>& vertices, const std::vector<Face>& faces, const std::vector<Texture15>& textures) : ctx_(fb.ctx()) {
        std::vector<B32Vertex> v; v.reserve(vertices.size());
        for (const auto& x : vertices) v.push_back(detail::pack(x));
        std::vector<B32Face> f; f.reserve(faces.size());
        for (const auto& x : faces) f.push_back(detail::pack(x));
        std::vector<B32Texture15> t; t.reserve(textures.size());
        for (const auto& x : textures)
            t.push_back({ (uint32_t)x.width, (uint32_t)x.height, (uint32_t)x.blend_mode, 0, x.pixels.size() >= x.width * x.height ? x.pixels.data() : nullptr });
        // The context keeps whatever scene it holds: its scene moves into the fresh slot, the mesh is uploaded into the (now empty)
        // context, and a second exchange leaves the mesh in the slot and the context's own scene where it was -- also when the upload
        // throws (render_scene-style calls and ResidentMesh can then share one Framebuffer).
        check(b32_scene_create(ctx_, &slot_), "scene_create");
        check(b32_scene_swap(ctx_, slot_), "scene_swap");            // context's scene -> slot, context empty
        const int rc = b32_scene_upload(ctx_, v.data(), (uint32_t)v.size(), f.data(), (uint32_t)f.size(), t.data(), (uint32_t)t.size());
        const int rc2 = b32_scene_swap(ctx_, slot_);                 // mesh -> slot, context's scene back
        if (rc2) {
            // the exchange back failed (a deferred error of an earlier frame surfaced in it): the slot still holds the CONTEXT's own scene
            // and must not be destroyed with it -- one more attempt (the deferred error has been consumed), then leave the slot alive
            // (leaked rather than the caller's scene freed) and report
            const int rc3 = b32_scene_swap(ctx_, slot_);
            if (rc3 == B32_OK) { b32_scene_destroy(ctx_, slot_); slot_ = nullptr; }
            else slot_ = nullptr;
            check(rc2, "scene_swap");
        }
        if (rc) { b32_scene_destroy(ctx_, slot_); slot_ = nullptr; check(rc, "scene_upload"); }
    }
    ~ResidentMesh() { if (slot_) b32_scene_destroy(ctx_, slot_); }
    ResidentMesh(const ResidentMesh&) = delete;
    ResidentMesh& operator=(const ResidentMesh&) = delete;
    b32_scene* slot() const { return slot_; }
private:
    b32_ctx* ctx_ = nullptr;
    b32_scene* slot_ = nullptr;
};

// One frame of scene::render_scene (scene.rs:158-261): one camera, base settings and light list; per mesh the room's ambient and fog and
// the part's backface culling.  The meshes are drawn as merged runs where their draws commute (b32_frame_begin / _add_scene / _end).
struct MeshParams { float ambient; bool backface_cull, backface_wireframe; Fog fog; };
inline RasterTimings render_frame(Framebuffer& fb, const std::vector<std::pair<const ResidentMesh*, MeshParams>>& meshes, const Camera& camera,
                                  const RasterSettings& base) {
    const std::vector<B32Light> l = detail::pack(base.lights);
    const B32Camera c = detail::pack(camera);
    const B32Settings s = detail::pack(base, l);
    check(b32_frame_begin(fb.ctx(), &c, &s), "frame_begin");
    for (const auto& m : meshes) {
        B32MeshParams p{};
        p.ambient = m.second.ambient; p.backface_cull = m.second.backface_cull; p.backface_wireframe = m.second.backface_wireframe;
        p.has_fog = detail::pack(m.second.fog, p.fog) ? 1 : 0;
        check(b32_frame_add_scene(fb.ctx(), m.first->slot(), &p), "frame_add_scene");
    }
    check(b32_frame_end(fb.ctx()), "frame_end");
    B32Timings tm{};
    check(b32_frame_finish(fb.ctx(), &tm), "frame_finish");
    return { tm.transform_ms, tm.fog_ms, tm.cull_ms, tm.sort_ms, tm.draw_ms, tm.wireframe_ms, tm.triangles_drawn, tm.fragments };
}

// The console's loop as the reference runs it -- every frame drawn (scene::render_scene, scene.rs:158-261) AND handed to the presenter from host
// memory (game/renderer.rs:179-214) -- without a host round trip per frame: submit() enqueues Framebuffer::clear, the frame's draws
// (b32_frame_submit: the mesh table in one call) and the copy of the finished frame into page-locked memory (b32_fb_download_async), and
// returns a ticket; wait(ticket) blocks until THAT frame's pixels are in the returned buffer.  Two buffers alternate, so the presenter reads
// frame i while frame i + 1 is drawn: keep at most two tickets outstanding.
class FrameLoop {
public:
    explicit FrameLoop(Framebuffer& fb) : fb_(fb) {
        for (auto& b : buf_) { b = static_cast<uint8_t*>(b32_host_alloc(fb.width * fb.height * 4)); if (!b) throw Error(B32_E_HIP, "b32_host_alloc"); }
    }
    ~FrameLoop() { b32_synchronize(fb_.ctx()); for (auto b : buf_) b32_host_free(b); }
    FrameLoop(const FrameLoop&) = delete;
    FrameLoop& operator=(const FrameLoop&) = delete;
    uint64_t submit(Color clear, const std::vector<std::pair<const ResidentMesh*, MeshParams>>& meshes, const Camera& camera, const RasterSettings& base) {
        const std::vector<B32Light> l = detail::pack(base.lights);
        const B32Camera c = detail::pack(camera);
        const B32Settings s = detail::pack(base, l);
        std::vector<b32_scene*> slots; std::vector<B32MeshParams> params;
        for (const auto& m : meshes) {
            B32MeshParams p{};
            p.ambient = m.second.ambient; p.backface_cull = m.second.backface_cull; p.backface_wireframe = m.second.backface_wireframe;
            p.has_fog = detail::pack(m.second.fog, p.fog) ? 1 : 0;
            slots.push_back(m.first->slot()); params.push_back(p);
        }
        fb_.clear(clear);
        check(b32_frame_submit(fb_.ctx(), &c, &s, slots.data(), params.data(), (uint32_t)slots.size()), "frame_submit");
        const size_t k = n_++ & 1;
        check(b32_fb_download_async(fb_.ctx(), buf_[k], &ticket_[k]), "fb_download_async");
        return ticket_[k];
    }
    // the frame of `ticket`, once it has landed (valid until the submit after next)
    const uint8_t* wait(uint64_t ticket) {
        check(b32_ticket_wait(fb_.ctx(), ticket), "ticket_wait");
        for (size_t k = 0; k < 2; ++k) if (ticket_[k] == ticket) return buf_[k];
        throw Error(B32_E_ARG, "FrameLoop::wait: the ticket's buffer has been reused");
    }
    // errors of the frames enqueued so far (b32_frame_finish), like any asynchronous frame
    void finish() { B32Timings tm{}; check(b32_frame_finish(fb_.ctx(), &tm), "frame_finish"); }
private:
    Framebuffer& fb_;
    uint8_t* buf_[2] = { nullptr, nullptr };
    uint64_t ticket_[2] = { 0, 0 };
    size_t n_ = 0;
};

// Names used by BASELINE.json's north_star; the reference's real entry point is render_mesh_15 (SURVEY headline fact 3).
inline RasterTimings draw_mesh(Framebuffer& fb, const std::vector<Vertex>& v, const std::vector<Face>& f, const std::vector<Texture15>& t,
                               const Camera& c, const RasterSettings& s, const Fog& fog = std::nullopt) { return render_mesh_15(fb, v, f, t, c, s, fog); }

}  // namespace b32

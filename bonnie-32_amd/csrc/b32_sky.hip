// b32_sky.hip — the steps around render_mesh_15 that the reference performs on the same framebuffer, so that a
// whole frame can stay on the device: Framebuffer::clear_gradient (render.rs:58-77), clear_transparent (render.rs:47-56), the
// skybox sphere fill of Framebuffer::render_skybox (render.rs:81-134 -> rasterize_skybox_triangle render.rs:251-298), the star
// sprites (draw_star_diamond render.rs:199-240) and the presenter's nearest-neighbour upscale (game/renderer.rs:179-214).
// The procedural parts that call sin/cos/powf (Skybox::generate_mesh world/geometry.rs:529, star directions render.rs:166-196)
// stay on the host like the camera basis: they are inputs here.
#include "b32_device.h"

namespace b32 {

// ---------------------------------------------------------------- clear_gradient: one colour per row, Color::lerp (types.rs:812-821)
__global__ void k_clear_gradient(uint32_t* __restrict__ fb, float* __restrict__ zbuf, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1,
                                 uint32_t top, uint32_t bottom) {
    const uint32_t y = y0 + blockIdx.y;
    if (y >= y1) return;
    const float t0 = height > 1 ? (float)y / (float)(height - 1) : 0.0f;
    const float t = rclamp(t0, 0.0f, 1.0f), inv_t = 1.0f - t;
    uint32_t c = ((top >> 24) == B32_BLEND_ERASE) ? 0u : 0xFF000000u;               // keeps self's blend mode -> Color::to_bytes alpha
#pragma unroll
    for (int i = 0; i < 3; ++i) c |= f2u8_sat((float)((top >> (8 * i)) & 255) * inv_t + (float)((bottom >> (8 * i)) & 255) * t) << (8 * i);
    for (uint32_t x = blockIdx.x * blockDim.x + threadIdx.x; x < width; x += gridDim.x * blockDim.x) {
        fb[(size_t)y * width + x] = c;
        if (zbuf) zbuf[(size_t)y * width + x] = 3.40282347e+38f;
    }
}
void launch_clear_gradient(hipStream_t s, uint32_t* fb, float* zbuf, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1, uint32_t top, uint32_t bottom) {
    if (y1 <= y0 || !width) return;
    hipLaunchKernelGGL(k_clear_gradient, dim3((width + 255) / 256 > 8 ? 8 : (width + 255) / 256, y1 - y0), dim3(256), 0, s, fb, zbuf, width, height, y0, y1, top, bottom);
}

// ---------------------------------------------------------------- skybox sphere
// vertex pass: perspective_transform (math.rs:103-109) + project (math.rs:117-136); behind the camera -> NaN marker (render.rs:99-103)
__global__ void k_sky_project(const B32SkyVertex* __restrict__ v, uint32_t nv, B32Camera cam, uint32_t width, uint32_t height, float2* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    const float rx = v[i].pos[0] - cam.position[0], ry = v[i].pos[1] - cam.position[1], rz = v[i].pos[2] - cam.position[2];
    const float cx = rx * cam.basis_x[0] + ry * cam.basis_x[1] + rz * cam.basis_x[2];
    const float cy = rx * cam.basis_y[0] + ry * cam.basis_y[1] + rz * cam.basis_y[2];
    const float cz = rx * cam.basis_z[0] + ry * cam.basis_z[1] + rz * cam.basis_z[2];
    const float qnan = __uint_as_float(0x7FC00000u);
    if (cz <= 0.1f) { out[i] = make_float2(qnan, qnan); return; }
    const uint32_t mn = width < height ? width : height;
    const float vs = ((float)mn / 2.0f) * 0.75f;
    const float denom = cz + 5.0f;
    if (__builtin_fabsf(denom) < 0.001f) { out[i] = make_float2((float)width / 2.0f, (float)height / 2.0f); return; }
    out[i] = make_float2((cx * 4.0f) / denom * vs + ((float)width / 2.0f), (cy * 4.0f) / denom * vs + ((float)height / 2.0f));
}

constexpr int SKY_LIST_CAP = 4096;      // faces one 64x64 tile can list per round (the walk is chunked beyond that)

// One workgroup per 64x64 tile.  The reference fills the faces one after another (plain overwrite, render.rs:286-294), so a pixel
// ends up with the LAST face in order that contains it: the tile first lists, in face order, the faces whose (clamped,
// inclusive) bounding box reaches it, then every lane walks that list for its pixels and keeps the last hit.
__global__ __launch_bounds__(256) void k_sky_fill(const B32SkyVertex* __restrict__ v, const uint32_t* __restrict__ faces, uint32_t nf, uint32_t nv,
                                                  const float2* __restrict__ proj, uint32_t* __restrict__ fb, uint32_t width, uint32_t height,
                                                  uint32_t band_y0, uint32_t band_y1, uint32_t tiles_x, uint32_t tile_y0) {
    __shared__ uint32_t list[SKY_LIST_CAP];
    __shared__ uint32_t wcount[4];
    __shared__ uint32_t n_list;
    const uint32_t tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x + tile_y0;
    const uint32_t x_lo = tx * 64, x_hi = min(x_lo + 64, width), y_lo = max(ty * 64, band_y0), y_hi = min(ty * 64 + 64, band_y1);
    if (x_lo >= x_hi || y_lo >= y_hi) return;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t best[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) best[k] = 0xFFFFFFFFu;
    for (uint32_t f_base = 0; f_base < nf; ) {
        // ---- list faces [f_base, ...) touching the tile, in face order, until the list is full
        if (threadIdx.x == 0) n_list = 0;
        __syncthreads();
        uint32_t f_next = f_base;
        while (f_next < nf) {
            const uint32_t f = f_next + threadIdx.x;
            bool hit = false;
            if (f < nf) {
                const uint32_t i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
                if (i0 < nv && i1 < nv && i2 < nv) {
                    const float2 p0 = proj[i0], p1 = proj[i1], p2 = proj[i2];
                    if (!(p0.x != p0.x || p1.x != p1.x || p2.x != p2.x)) {                                  // render.rs:112-114
                        const float signed_area = (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);   // :118-121
                        if (!(signed_area >= 0.0f)) {
                            const uint32_t min_x = f2u_sat(rmax(rmin(rmin(p0.x, p1.x), p2.x), 0.0f));               // :262-265 (inclusive max)
                            const uint32_t max_x = f2u_sat(rmin(rmax(rmax(p0.x, p1.x), p2.x), (float)width - 1.0f));
                            const uint32_t min_y = f2u_sat(rmax(rmin(rmin(p0.y, p1.y), p2.y), 0.0f));
                            const uint32_t max_y = f2u_sat(rmin(rmax(rmax(p0.y, p1.y), p2.y), (float)height - 1.0f));
                            hit = min_x <= max_x && min_y <= max_y && max_x >= x_lo && min_x < x_hi && max_y >= y_lo && min_y < y_hi;
                        }
                    }
                }
            }
            const unsigned long long m = __ballot(hit);
            if (lane == 0) wcount[wave] = (uint32_t)__popcll(m);
            __syncthreads();
            const uint32_t chunk = wcount[0] + wcount[1] + wcount[2] + wcount[3];
            const uint32_t have = n_list;
            if (have + chunk > SKY_LIST_CAP) { __syncthreads(); break; }                // this chunk goes to the next round (f_next unchanged)
            uint32_t pos = have;
            for (uint32_t w = 0; w < wave; ++w) pos += wcount[w];
            if (hit) list[pos + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = f;
            __syncthreads();
            if (threadIdx.x == 0) n_list = have + chunk;
            f_next += 256;
            __syncthreads();
        }
        const uint32_t n = n_list;
        // ---- every lane: 16 pixels of the tile (rows tid/64 + 4k), last containing face wins
        for (uint32_t e = 0; e < n; ++e) {
            const uint32_t f = list[e];
            const float2 p0 = proj[faces[3 * f]], p1 = proj[faces[3 * f + 1]], p2 = proj[faces[3 * f + 2]];
            const float denom = (p1.y - p2.y) * (p0.x - p2.x) + (p2.x - p1.x) * (p0.y - p2.y);              // render.rs:272
            if (__builtin_fabsf(denom) < 0.0001f) continue;
            const float inv_denom = 1.0f / denom;
            const uint32_t min_x = f2u_sat(rmax(rmin(rmin(p0.x, p1.x), p2.x), 0.0f));
            const uint32_t max_x = f2u_sat(rmin(rmax(rmax(p0.x, p1.x), p2.x), (float)width - 1.0f));
            const uint32_t min_y = f2u_sat(rmax(rmin(rmin(p0.y, p1.y), p2.y), 0.0f));
            const uint32_t max_y = f2u_sat(rmin(rmax(rmax(p0.y, p1.y), p2.y), (float)height - 1.0f));
            const uint32_t x = x_lo + lane;
            if (x < min_x || x > max_x || x >= x_hi) continue;
            const float px = (float)x + 0.5f;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const uint32_t y = ty * 64 + wave + 4 * k;
                if (y < y_lo || y >= y_hi || y < min_y || y > max_y) continue;
                const float py = (float)y + 0.5f;
                const float w0 = ((p1.y - p2.y) * (px - p2.x) + (p2.x - p1.x) * (py - p2.y)) * inv_denom;   // render.rs:283-285
                const float w1 = ((p2.y - p0.y) * (px - p2.x) + (p0.x - p2.x) * (py - p2.y)) * inv_denom;
                const float w2 = 1.0f - w0 - w1;
                if (w0 >= 0.0f && w1 >= 0.0f && w2 >= 0.0f) best[k] = f;
            }
        }
        f_base = f_next;
        __syncthreads();
    }
    // ---- colour of the winning face per pixel (render.rs:288-296)
    const uint32_t x = x_lo + lane;
    if (x >= x_hi) return;
    const float px = (float)x + 0.5f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const uint32_t f = best[k];
        const uint32_t y = ty * 64 + wave + 4 * k;
        if (f == 0xFFFFFFFFu || y < y_lo || y >= y_hi) continue;
        const uint32_t i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
        const float2 p0 = proj[i0], p1 = proj[i1], p2 = proj[i2];
        const float denom = (p1.y - p2.y) * (p0.x - p2.x) + (p2.x - p1.x) * (p0.y - p2.y);
        const float inv_denom = 1.0f / denom;
        const float py = (float)y + 0.5f;
        const float w0 = ((p1.y - p2.y) * (px - p2.x) + (p2.x - p1.x) * (py - p2.y)) * inv_denom;
        const float w1 = ((p2.y - p0.y) * (px - p2.x) + (p0.x - p2.x) * (py - p2.y)) * inv_denom;
        const float w2 = 1.0f - w0 - w1;
        const uint8_t* c0 = &v[i0].r; const uint8_t* c1 = &v[i1].r; const uint8_t* c2 = &v[i2].r;
        uint32_t o = 0xFF000000u;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) o |= f2u8_sat((float)c0[ch] * w0 + (float)c1[ch] * w1 + (float)c2[ch] * w2) << (8 * ch);
        fb[(size_t)y * width + x] = o;
    }
}

void launch_sky(hipStream_t s, const B32SkyVertex* v, uint32_t nv, const uint32_t* faces, uint32_t nf, const B32Camera& cam, float2* proj,
                uint32_t* fb, uint32_t width, uint32_t height, uint32_t band_y0, uint32_t band_y1) {
    if (!nv || !nf || band_y1 <= band_y0) return;
    hipLaunchKernelGGL(k_sky_project, dim3((nv + 255) / 256), dim3(256), 0, s, v, nv, cam, width, height, proj);
    const uint32_t tiles_x = (width + 63) / 64, tile_y0 = band_y0 / 64, tiles_y = (band_y1 + 63) / 64 - tile_y0;
    hipLaunchKernelGGL(k_sky_fill, dim3(tiles_x * tiles_y), dim3(256), 0, s, v, faces, nf, nv, proj, fb, width, height, band_y0, band_y1, tiles_x, tile_y0);
}

// ---------------------------------------------------------------- star sprites, draw_star_diamond (render.rs:199-240): strictly in order
__global__ void k_stars(const int32_t* __restrict__ cx, const int32_t* __restrict__ cy, const uint8_t* __restrict__ rgb, uint32_t n, float size,
                        uint32_t* __restrict__ fb, uint32_t width, uint32_t height, uint32_t band_y0, uint32_t band_y1) {
    // lane k of the single wave owns sprite pixel k (centre, 4 near points, 4 far points); sprites are applied one after another
    const int k = (int)threadIdx.x;
    if (k >= 9) return;
    const int s = (int)f2i32_sat(rmax(size, 1.0f));                                   // size.max(1.0) as i32
    const int dx[9] = { 0, -1, 1, 0, 0, -2, 2, 0, 0 }, dy[9] = { 0, 0, 0, -1, 1, 0, 0, -2, 2 };
    const float scale = k == 0 ? 1.0f : (k < 5 ? 0.7f : 0.4f);
    const bool on = k == 0 || (k < 5 ? s >= 2 : s >= 3);
    for (uint32_t i = 0; i < n; ++i) {
        if (!on) continue;
        // `cx - 1`, `cx + 2` ... on i32 (render.rs:219-236): a release build wraps (INT_MAX + 2 is negative -> set_pixel_safe rejects it)
        const int x = (int)((uint32_t)cx[i] + (uint32_t)dx[k]), y = (int)((uint32_t)cy[i] + (uint32_t)dy[k]);
        if (x < 0 || y < 0 || x >= (int)width || y >= (int)height || (uint32_t)y < band_y0 || (uint32_t)y >= band_y1) continue;
        uint32_t o = 0xFF000000u;
        for (int ch = 0; ch < 3; ++ch) {
            const uint32_t c = rgb[3 * i + ch];
            o |= (k == 0 ? c : f2u8_sat((float)c * scale)) << (8 * ch);
        }
        fb[(size_t)y * width + (uint32_t)x] = o;
    }
}
void launch_stars(hipStream_t s, const int32_t* cx, const int32_t* cy, const uint8_t* rgb, uint32_t n, float size, uint32_t* fb,
                  uint32_t width, uint32_t height, uint32_t band_y0, uint32_t band_y1) {
    if (!n) return;
    hipLaunchKernelGGL(k_stars, dim3(1), dim3(64), 0, s, cx, cy, rgb, n, size, fb, width, height, band_y0, band_y1);
}

// ---------------------------------------------------------------- presenter: nearest-neighbour upscale (game/renderer.rs:179-214)
// FilterMode::Nearest of the texture drawn into a dest_size rectangle: destination pixel centre (x + 0.5) maps to the source
// texel floor((x + 0.5) * src / dst) -- the GL_NEAREST rule; integer arithmetic: ((2x + 1) * src) / (2 dst).
__global__ void k_upscale_nearest(const uint32_t* __restrict__ src, uint32_t sw, uint32_t sh, uint32_t* __restrict__ dst, uint32_t dw, uint32_t dh) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= dw || y >= dh) return;
    const uint32_t sx = (uint32_t)(((unsigned long long)(2 * x + 1) * sw) / (2ull * dw));
    const uint32_t sy = (uint32_t)(((unsigned long long)(2 * y + 1) * sh) / (2ull * dh));
    dst[(size_t)y * dw + x] = src[(size_t)min(sy, sh - 1) * sw + min(sx, sw - 1)];
}
void launch_upscale_nearest(hipStream_t s, const uint32_t* src, uint32_t sw, uint32_t sh, uint32_t* dst, uint32_t dw, uint32_t dh) {
    if (!dw || !dh) return;
    hipLaunchKernelGGL(k_upscale_nearest, dim3((dw + 255) / 256, dh), dim3(256), 0, s, src, sw, sh, dst, dw, dh);
}

}  // namespace b32

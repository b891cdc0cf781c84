// b32_fill.hip — affine texture-mapped triangle fill with RGB555 dither (rasterize_triangle_15, render.rs:1440-1714)
// as a visibility-buffer pipeline of three kernels (MI355X-first, not the reference's per-triangle scanline loop):
//
//   k_cover  persistent workgroups pull 64x64 screen tiles from a device-side cursor.  The surfaces binned to a tile
//            arrive in painter's order (b32_bin.hip).  The opaque pass of the reference (render.rs:2553-2559) only ever
//            overwrites pixels (set_pixel_15), so its result per pixel is the LAST surface in painter's order whose
//            fragment is not skipped: coverage of all opaque surfaces of the tile runs in parallel and visibility is an
//            LDS atomicMax of the surface's position in the tile list (order-independent, deterministic, no overdraw
//            shading).  The tile of winners is written, row-coalesced, to the u32 visibility buffer.
//              EXACT coverage applies the whole skip rule per fragment (inside test + texel + transparency,
//                    render.rs:1536-1607; texture staged in LDS) and counts the reference's pixel stores exactly;
//              CHEAP coverage (textures with few skippable texels) applies only the inside test; the rare pixels whose
//                    top surface turns out to be skipped are repaired by k_shade.
//            Coverage is scheduled by ROW ITEMS (see phase_a_rows): every lane walks one row of one surface.
//   k_shade  one lane per pixel, grid-stride, high occupancy: winner -> surface record -> barycentrics -> texel ->
//            colour pipeline (render.rs:1613-1661) -> RGBA8 store (Color15::to_rgba), 256-B coalesced per wave.  If the
//            winner's texel is skipped (CHEAP only) the wave scans the tile list downward, 64 entries at a time, for
//            the highest surface below it whose fragment is really drawn — identical result to EXACT coverage.
//   k_blend  surfaces of the transparent pass (render.rs:2563-2569) blend against the framebuffer: what must be ordered
//            is, per pixel, the sequence of that pixel's own fragments.  Per tile, on an LDS copy of it, a lane owns a
//            pixel column and walks the surfaces whose box holds its pixels in painter's order (row / column masks of
//            the batch's 64 surfaces): set_pixel_blended_15 / editor-alpha stores; no atomics, no ordering between
//            surfaces that do not share a pixel.
//
// Bit-exactness: barycentrics use the reference's expression order; the edge functions are evaluated from exact integers
// only for surfaces k_setup proved exact (integer coordinates, every intermediate < 2^24), otherwise the incremental
// walk (render.rs:1706-1712) is replayed literally.
#include "b32_device.h"
#ifndef B32_TRIP
#define B32_TRIP 4
#endif
#ifndef B32_DRAIN_TRIPS
#define B32_DRAIN_TRIPS 2
#endif
#ifndef B32_P64_WAVES
#define B32_P64_WAVES 4          // minimum waves per SIMD the general 8-wave forms of the fused kernel are compiled for
#endif
#ifndef B32_P64_STRIDE
#define B32_P64_STRIDE 66        // row stride (u64 entries) of the 64-bit winner planes: 64 + 2, so the rows a surface touches at one
                                 // column fall into different LDS banks (measured: 72 -> 133 us, 66 -> 128 us; must stay <= 72, the allocation)
#endif

namespace b32 {

constexpr int LDS_TILE_BYTES = TILE_H * TILE_STRIDE * 4;        // 18432
constexpr int STR64 = B32_P64_STRIDE;
static_assert(STR64 >= 64 && STR64 <= TILE_STRIDE && (TILE_H - 1) * STR64 + 64 + 4 <= TILE_H * TILE_STRIDE, "64-bit planes must fit their allocation, trip overshoot included");
constexpr int LDS_MISC_BYTES = 64;
constexpr int LDS_MARK_BYTES = FILL_WAVES * 64 * 4;             // row-start marks of the row-item scheduler
constexpr int LDS_TEX_OFFSET = 2 * LDS_TILE_BYTES + LDS_MISC_BYTES + LDS_MARK_BYTES;   // 41024: top + runner-up tile buffers
constexpr int LDS_SORT_CNT_BYTES = 8 * 256 * 4;                 // per-wave digit counters of the tile-local sort (8 waves)
static_assert(4 * LOCAL_SORT_CAP * 4 <= 2 * LDS_TILE_BYTES, "the tile-local sort aliases the two tile buffers");

__device__ __forceinline__ float bcf(float v, int t) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), t)); }
__device__ __forceinline__ uint32_t bcu(uint32_t v, int t) { return (uint32_t)__builtin_amdgcn_readlane((int)v, t); }

struct Tri {            // one SurfRec (+ its texture), wave-uniform in phase A/C, per-lane in phase B
    float x3, y3, a0, b0, a1, b1, inv_area;
    float u1, u2, u3, v1, v2, v3;
    float w0_start, w1_start;
    float iz1, iz2, iz3;
    uint32_t min_x, max_x, min_y, max_y, flags;
    uint32_t tw, th, toff;
};

template <int TEXMODE>
__device__ __forceinline__ uint32_t sample15(const Tri& t, const uint16_t* __restrict__ gtex, const uint16_t* ltex, float u, float v) {
    // Texture15::sample, types.rs:671-681
    if (t.tw == 0 || t.th == 0) return 0;
    const float uw = rem_euclid1(u), vw = rem_euclid1(v);
    const uint32_t tx = min(f2u_sat(uw * (float)t.tw), t.tw - 1);
    const uint32_t ty = min(f2u_sat(vw * (float)t.th), t.th - 1);
    if (TEXMODE == 1) return ltex[ty * t.tw + tx];
    return gtex[t.toff + ty * t.tw + tx];
}

__device__ __forceinline__ bool inside_bc(const Tri& t, float w0, float w1, float& bcx, float& bcy, float& bcz) {
    bcx = w0 * t.inv_area;                                       // render.rs:1536-1542
    bcy = w1 * t.inv_area;
    bcz = 1.0f - bcx - bcy;
    const float ERR = K::ERR;
    // bcx >= ERR && bcy >= ERR && bcz >= ERR with one comparison less.  NaN-safe although fminf drops a NaN operand: a NaN (or an
    // infinity of either sign) in bcx or bcy makes bcz NaN or -inf, and `bcz >= ERR` is then false like the original conjunction.
    return (__builtin_fminf(bcx, bcy) >= ERR) & (bcz >= ERR);
}

// Texture::sample of the 8-bit-colour path (types.rs:1242-1253): Color texel r | g<<8 | b<<16 | blend<<24
__device__ __forceinline__ uint32_t sample8(const Tri& t, const uint32_t* __restrict__ gtex, float u, float v) {
    if (t.tw == 0 || t.th == 0) return (uint32_t)B32_BLEND_ERASE << 24;                 // Color::TRANSPARENT
    const float uw = rem_euclid1(u), vw = rem_euclid1(v);
    const uint32_t tx = min(f2u_sat(uw * (float)t.tw), t.tw - 1);
    const uint32_t ty = min(f2u_sat(vw * (float)t.th), t.th - 1);
    return gtex[t.toff + ty * t.tw + tx];
}

// Texel address of a fragment (index into the texel pool): -1 = untextured (white), -2 = zero-size texture (transparent sample).
// Same arithmetic as texel_drawn / Texture15::sample (types.rs:671-681); split off so that several fetches can be in flight.
__device__ __forceinline__ int tri_texel_addr(const Tri& t, float bcx, float bcy, float bcz, bool affine) {
    if ((t.flags & F_TEX_MASK) == F_TEX_NONE) return -1;
    if (t.tw == 0 || t.th == 0) return -2;
    float u, v;
    if (affine) {
        u = bcx * t.u1 + bcy * t.u2 + bcz * t.u3;
        v = bcx * t.v1 + bcy * t.v2 + bcz * t.v3;
    } else {
        const float inv_z = bcx * t.iz1 + bcy * t.iz2 + bcz * t.iz3;
        const float u_over_z = bcx * t.u1 * t.iz1 + bcy * t.u2 * t.iz2 + bcz * t.u3 * t.iz3;
        const float v_over_z = bcx * t.v1 * t.iz1 + bcy * t.v2 * t.iz2 + bcz * t.v3 * t.iz3;
        u = u_over_z / inv_z;
        v = v_over_z / inv_z;
    }
    const float uw = rem_euclid1(u), vw = rem_euclid1(1.0f - v);
    const uint32_t tx = min(f2u_sat(uw * (float)t.tw), t.tw - 1);
    const uint32_t ty = min(f2u_sat(vw * (float)t.th), t.th - 1);
    return (int)(t.toff + ty * t.tw + tx);
}

// Texel fetch + transparency rules (render.rs:1563-1607; 8-bit path render.rs:1322-1352). Returns false when the fragment is skipped.
template <int TEXMODE, bool FMT8 = false>
__device__ __forceinline__ bool texel_drawn(const Tri& t, float bcx, float bcy, float bcz, const uint16_t* __restrict__ gtex,
                                            const uint16_t* ltex, uint32_t& texel, bool affine = true) {
    uint32_t c = FMT8 ? 0x00FFFFFFu : K::C15_WHITE;              // Color::WHITE (render.rs:1344) / Color15::WHITE (render.rs:1585)
    if ((t.flags & F_TEX_MASK) != F_TEX_NONE) {
        float u, v;
        if (affine) {
            u = bcx * t.u1 + bcy * t.u2 + bcz * t.u3;            // affine, render.rs:1565-1566
            v = bcx * t.v1 + bcy * t.v2 + bcz * t.v3;
        } else {                                                 // perspective-correct, render.rs:1568-1579
            const float inv_z = bcx * t.iz1 + bcy * t.iz2 + bcz * t.iz3;
            const float u_over_z = bcx * t.u1 * t.iz1 + bcy * t.u2 * t.iz2 + bcz * t.u3 * t.iz3;
            const float v_over_z = bcx * t.v1 * t.iz1 + bcy * t.v2 * t.iz2 + bcz * t.v3 * t.iz3;
            u = u_over_z / inv_z;
            v = v_over_z / inv_z;
        }
        if (FMT8) c = sample8(t, reinterpret_cast<const uint32_t*>(gtex), u, 1.0f - v);   // render.rs:1342
        else c = sample15<TEXMODE>(t, gtex, ltex, u, 1.0f - v);  // render.rs:1583
    }
    if (FMT8) {                                                  // color.is_transparent(), render.rs:1348-1352
        texel = c;
        return (c >> 24) != B32_BLEND_ERASE;
    }
    if (c == K::C15_TRANSPARENT) {                               // render.rs:1592-1602
        if (t.flags & F_BLACK_TR) return false;
        c = K::C15_BLACK_DRAWABLE;
    } else if ((t.flags & F_BLACK_TR) && (c & ~K::C15_SEMI_BIT & 0xFFFFu) == 0) {    // is_black: r5 == g5 == b5 == 0, render.rs:1603-1608
        return false;
    }
    texel = c;
    return true;
}

// Colour pipeline (render.rs:1613-1661): modulate by interpolated vertex colour, shade, dither, quantize to RGB555.
template <bool RGBA = false>
__device__ __forceinline__ uint32_t shade15(uint32_t texel, float bcx, float bcy, float bcz, uint32_t vc1, uint32_t vc2, uint32_t vc3,
                                            uint32_t flags, int shading, const float* sh, uint32_t px, uint32_t py) {
    uint32_t q[3];
    const int off = dither_offset(px, py);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const uint32_t c5 = (texel >> (i == 0 ? K::C15_R_SHIFT : (i == 1 ? K::C15_G_SHIFT : 0u))) & K::C15_CHANNEL_MAX;   // i=0 r, 1 g, 2 b
        const uint32_t tex8 = expand5(c5);
        const float f1 = (float)((vc1 >> (8 * i)) & 255), f2 = (float)((vc2 >> (8 * i)) & 255), f3 = (float)((vc3 >> (8 * i)) & 255);
        const uint32_t vert = f2u8_sat(bcx * f1 + bcy * f2 + bcz * f3);              // :1618-1620
        uint32_t m = min((tex8 * vert) / K::MOD_DIV, K::MOD_MAX);                     // :1624-1626
        if (shading != B32_SHADE_NONE) {                                              // :1629-1645 (x1.0 is exact when None)
            const float s = shading == B32_SHADE_FLAT ? sh[i] : (bcx * sh[i] + bcy * sh[3 + i] + bcz * sh[6 + i]);
            m = f2u8_sat(rmin((float)m * rclamp(s, K::SHADE_LO, K::SHADE_HI), K::SHADE_MAX));
        }
        if (flags & F_DITHER) q[i] = (uint32_t)min(max(((int)m + off) >> K::DITHER_SHIFT, K::DITHER_LO), K::DITHER_HI); // dither_and_quantize :1173-1182
        else q[i] = m >> K::NODITHER_SHIFT;                                           // :1653
    }
    if (RGBA) {
        // straight to the RGBA8 word set_pixel_15 stores (render.rs:445-454, Color15::to_rgba types.rs:220-226): the Color15 in between is
        // never 0x0000 (an all-black result gets bit 15, :1659-1661), so its to_rgba is always the three expanded channels + alpha 255
        return expand5(q[0]) | (expand5(q[1]) << 8) | (expand5(q[2]) << 16) | 0xFF000000u;
    }
    const bool all_black = (q[0] | q[1] | q[2]) == 0;                                 // :1659-1661
    return (q[0] << K::C15_R_SHIFT) | (q[1] << K::C15_G_SHIFT) | q[2] | (((texel & K::C15_SEMI_BIT) || all_black) ? K::C15_SEMI_BIT : 0u);
}

// The same colour pipeline for TWO pixels at once (the fused kernel shades two pixels per lane), no shading pass (RasterSettings.shading
// == None: the shade factor is x1.0, render.rs:1629-1645).  The arithmetic is the reference's, value for value; only the instructions are
// packed -- the vertex-colour interpolation as v_pk_mul_f32 / v_pk_add_f32 over the pair (two f32 roundings per product and sum as in the
// scalar form: contraction is off), the integer tail on 16-bit halves (v_pk_mul_lo_u16 ...: tex8 * vert <= 255 * 255 fits 16 bits).
// A surface without needs_dither quantises with `>> 3`, which is the dither formula with offset 0: (m + 0) >> 3 <= 31 for m <= 255.
typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned short v2us __attribute__((ext_vector_type(2)));
typedef short v2ss __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void shade15_pair_rgba(uint32_t texelA, uint32_t texelB, const float bcA[3], const float bcB[3], const uint32_t vcA[3],
                                                  const uint32_t vcB[3], uint32_t flagsA, uint32_t flagsB, uint32_t px, uint32_t pyA, uint32_t pyB,
                                                  uint32_t& outA, uint32_t& outB) {
    const v2f bcx = { bcA[0], bcB[0] }, bcy = { bcA[1], bcB[1] }, bcz = { bcA[2], bcB[2] };
    const int offA = (flagsA & F_DITHER) ? dither_offset(px, pyA) : 0, offB = (flagsB & F_DITHER) ? dither_offset(px, pyB) : 0;
    const v2ss off = { (short)offA, (short)offB };
    const v2us tx = { (unsigned short)texelA, (unsigned short)texelB };
    v2us e[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const unsigned short sh = (unsigned short)(i == 0 ? K::C15_R_SHIFT : (i == 1 ? K::C15_G_SHIFT : 0u));
        const v2us c5 = (tx >> sh) & (unsigned short)K::C15_CHANNEL_MAX;
        const v2us tex8 = ((c5 << (unsigned short)K::EXPAND5_SHL) | (c5 >> (unsigned short)K::EXPAND5_SHR));          // expand_5_to_8 :1161-1163 (<= 255)
        const v2f f1 = { (float)((vcA[0] >> (8 * i)) & 255), (float)((vcB[0] >> (8 * i)) & 255) };
        const v2f f2 = { (float)((vcA[1] >> (8 * i)) & 255), (float)((vcB[1] >> (8 * i)) & 255) };
        const v2f f3 = { (float)((vcA[2] >> (8 * i)) & 255), (float)((vcB[2] >> (8 * i)) & 255) };
        const v2f acc = bcx * f1 + bcy * f2 + bcz * f3;                                                              // :1618-1620
        const v2us vert = { (unsigned short)f2u8_sat(acc.x), (unsigned short)f2u8_sat(acc.y) };
        const v2us m = __builtin_elementwise_min((v2us)((tex8 * vert) >> (unsigned short)7), (v2us){ (unsigned short)K::MOD_MAX, (unsigned short)K::MOD_MAX });   // / 128, .min(255) :1624-1626
        v2ss q = (__builtin_bit_cast(v2ss, m) + off) >> (short)K::DITHER_SHIFT;                                       // dither_and_quantize :1173-1182
        q = __builtin_elementwise_min(__builtin_elementwise_max(q, (v2ss){ (short)K::DITHER_LO, (short)K::DITHER_LO }), (v2ss){ (short)K::DITHER_HI, (short)K::DITHER_HI });
        const v2us qu = __builtin_bit_cast(v2us, q);
        e[i] = (qu << (unsigned short)K::EXPAND5_SHL) | (qu >> (unsigned short)K::EXPAND5_SHR);                      // Color15::to_rgba types.rs:220-226 (see shade15<true>)
    }
    outA = (uint32_t)e[0].x | ((uint32_t)e[1].x << 8) | ((uint32_t)e[2].x << 16) | 0xFF000000u;
    outB = (uint32_t)e[0].y | ((uint32_t)e[1].y << 8) | ((uint32_t)e[2].y << 16) | 0xFF000000u;
}
static_assert(K::MOD_DIV == 128 && K::MOD_MAX == 255 && K::DITHER_SHIFT == K::NODITHER_SHIFT, "shade15_pair_rgba: / 128 as a shift, no-dither == offset 0");

// Pixel store of the transparent pass in painter's mode (render.rs:1674-1680, 1695-1702) on an RGBA8 word.
__device__ __forceinline__ uint32_t store_blend(uint32_t back, uint32_t out15, uint32_t flags, bool xray) {
    const uint32_t mode = (flags >> F_BLEND_SHIFT) & 7u, alpha = flags >> F_ALPHA_SHIFT;
    const uint32_t front = c15_to_rgba(out15);
    if (xray) {                                                  // set_pixel_xray_15, render.rs:507-526: (front + back) / 2 per channel
        uint32_t o = 0xFF000000u;
#pragma unroll
        for (int i = 0; i < 3; ++i) o |= ((((front >> (8 * i)) & 255) + ((back >> (8 * i)) & 255)) >> 1) << (8 * i);
        return o;
    }
    const bool do_blend = (out15 & K::C15_SEMI_BIT) && mode != B32_BLEND_OPAQUE;
    if (alpha < 255) {                                           // set_pixel_with_editor_alpha_15, render.rs:567-591
        const uint32_t ps1 = do_blend ? blend_rgb555(front, back, mode) : front;
        uint32_t o = 0xFF000000u;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const uint32_t p = (ps1 >> (8 * i)) & 255, b = (back >> (8 * i)) & 255;
            o |= (((p * alpha + b * (255 - alpha)) & 0xFFFF) / 255u) << (8 * i);
        }
        return o;
    }
    if (do_blend) return blend_rgb555(front, back, mode) | 0xFF000000u;   // set_pixel_blended_15, render.rs:479-502
    return front;                                                           // set_pixel_15, render.rs:445-454
}

// 8-bit-colour pipeline (render.rs:1355-1387): modulate (types.rs:801-808), shade_color_rgb (render.rs:1074-1081, no clamp of
// the shade), apply_dither (render.rs:1186-1197).  Returns r | g<<8 | b<<16 | blend<<24 (the texel's blend mode survives).
__device__ __forceinline__ uint32_t shade8(uint32_t texel, float bcx, float bcy, float bcz, uint32_t vc1, uint32_t vc2, uint32_t vc3,
                                           uint32_t flags, int shading, const float* sh, uint32_t px, uint32_t py) {
    uint32_t out = texel & 0xFF000000u;
    const int off = dither_offset(px, py);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const uint32_t t8 = (texel >> (8 * i)) & 255;
        const float f1 = (float)((vc1 >> (8 * i)) & 255), f2 = (float)((vc2 >> (8 * i)) & 255), f3 = (float)((vc3 >> (8 * i)) & 255);
        const uint32_t vert = f2u8_sat(bcx * f1 + bcy * f2 + bcz * f3);
        uint32_t m = min((t8 * vert) / K::MOD_DIV, K::MOD_MAX);
        if (shading != B32_SHADE_NONE) {
            const float s = shading == B32_SHADE_FLAT ? sh[i] : (bcx * sh[i] + bcy * sh[3 + i] + bcz * sh[6 + i]);
            m = f2u8_sat(rmin((float)m * s, K::SHADE_MAX));
        }
        if (flags & F_DITHER) m = (uint32_t)min(max(((int)m + off) >> K::DITHER_SHIFT, K::DITHER_LO), K::DITHER_HI) << K::DITHER8_EXPAND_SHIFT;
        out |= m << (8 * i);
    }
    return out;
}
// Pixel store of the 8-bit path once the depth test (if any) has passed: Color::blend_with (types.rs:886-936) by the
// colour's own blend mode, then the editor-alpha lerp in f32 (render.rs:356-366) -> RGBA8 word (Color::to_bytes).
__device__ __forceinline__ uint32_t store8(uint32_t back, uint32_t color, uint32_t alpha) {
    const uint32_t mode = color >> 24;
    uint32_t ps1 = 0, a8 = 255;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int f = (int)((color >> (8 * i)) & 255), b = (int)((back >> (8 * i)) & 255);
        int r;
        switch (mode) {
            default:
            case B32_BLEND_OPAQUE:      r = f; break;
            case B32_BLEND_AVERAGE:     r = (b + f) / 2; break;
            case B32_BLEND_ADD:         r = min(b + f, 255); break;
            case B32_BLEND_SUBTRACT:    r = max(b - f, 0); break;
            case B32_BLEND_ADD_QUARTER: r = min(b + f / 4, 255); break;
            case B32_BLEND_ERASE:       r = 0; a8 = 0; break;
        }
        ps1 |= (uint32_t)r << (8 * i);
    }
    if (alpha < 255) {
        const float a = (float)alpha / 255.0f, inv_a = 1.0f - a;
        uint32_t o = 0xFF000000u;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            o |= f2u8_sat((float)((ps1 >> (8 * i)) & 255) * a + (float)((back >> (8 * i)) & 255) * inv_a) << (8 * i);
        return o;
    }
    return ps1 | (a8 << 24);
}

// Replay of the reference's accumulated edge functions up to pixel (px,py) (render.rs:1527-1533, 1706-1712).
__device__ __forceinline__ void replay_w(const Tri& t, uint32_t px, uint32_t py, float& w0, float& w1) {
    float r0 = t.w0_start, r1 = t.w1_start;
    for (uint32_t y = t.min_y; y < py; ++y) { r0 += t.b0; r1 += t.b1; }
    for (uint32_t x = t.min_x; x < px; ++x) { r0 += t.a0; r1 += t.a1; }
    w0 = r0; w1 = r1;
}
__device__ __forceinline__ void edge_w(const Tri& t, uint32_t px, uint32_t py, float& w0, float& w1) {
    if (!(t.flags & F_SLOW)) {                                   // exact integers (k_setup guard): closed form == accumulation
        const float dx = (float)px - t.x3, dy = (float)py - t.y3;
        w0 = t.a0 * dx + t.b0 * dy; w1 = t.a1 * dx + t.b1 * dy;
    } else replay_w(t, px, py, w0, w1);
}

struct Batch {          // per-lane copy of one surface record (lane l <-> list entry chunk_start + l)
    uint4 q0, q1, q2, q3, q4, q5;
    uint32_t tw, th, toff;
};

// The surface's shading view (RecView quads q0..q4, q5 when asked for or when the surface is F_SLOW) from its ShadeRec, plus -- rare --
// the AuxRec and, for F_SLOW surfaces, the bounding box of the CovRec (the literal edge walk starts at the box origin).  The flags word
// is rebuilt for an opaque-pass surface: texture slot, F_BLACK_TR, F_DITHER, F_SLOW, editor alpha 255, blend mode Opaque.
__device__ __forceinline__ void load_shade_view(const FillArgs& a, uint32_t sid, bool need5, RecView& r) {
    const uint4* sp = reinterpret_cast<const uint4*>(a.srecs + sid);
    const uint4 s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3];
    view_edges_from_shade(r, s0, s1);
    r.q1.w = 0; r.q2.x = 0;
    r.q2.y = s2.x; r.q2.z = s2.y; r.q2.w = s2.z;
    r.q3.x = s2.w; r.q3.y = s3.x; r.q3.z = s3.y;
    const uint32_t sh = s3.w >> 24;
    r.q3.w = shade_tex_slot(s1, s3) | ((sh & SH_BLACK_TR) ? F_BLACK_TR : 0u) | ((sh & SH_DITHER) ? F_DITHER : 0u) | ((sh & SH_SLOW) ? F_SLOW : 0u) |
             (255u << F_ALPHA_SHIFT);
    r.q4 = make_uint4(s1.w & 0xFFFFFFu, s3.z & 0xFFFFFFu, s3.w & 0xFFFFFFu, 0u);
    r.q5 = make_uint4(0, 0, 0, 0);
    if (need5 || (sh & SH_SLOW)) {
        const uint4* xp = reinterpret_cast<const uint4*>(a.xrecs + sid);
        const uint4 x0 = xp[0], x1 = xp[1];
        r.q4.w = x0.w; r.q5 = make_uint4(x1.x, x0.x, x0.y, x0.z);
    }
    if (sh & SH_SLOW) { const uint4 c1 = reinterpret_cast<const uint4*>(a.crecs + sid)[1]; r.q1.w = c1.x; r.q2.x = c1.y; }
}
// Everything about a surface, with its true flags word (blend mode, editor alpha, class): the ordered pass and the list scans.
__device__ __forceinline__ void load_full_view(const FillArgs& a, uint32_t sid, RecView& r) {
    load_shade_view(a, sid, true, r);
    const uint4 c1 = reinterpret_cast<const uint4*>(a.crecs + sid)[1];
    r.q1.w = c1.x; r.q2.x = c1.y; r.q3.w = c1.w;
}

// Lane's list entry -> its coverage view (quads q0, q1, q2.x, q3.w), painter's key and face id.  need_uv: also the UVs and the texture
// (EXACT coverage applies the texel rule); need_aux: also the 1/z terms (z-buffer depth, perspective-correct UVs).
template <int TEXMODE>
__device__ __forceinline__ void load_batch(Batch& b, const FillArgs& a, uint32_t entry, bool live, const TexDesc& lds_desc, bool need_uv,
                                           bool need_aux, uint32_t& sid_out, uint32_t& key_out, bool& narrow_out) {
    b.q0 = b.q1 = b.q2 = b.q3 = b.q4 = b.q5 = make_uint4(0, 0, 0, 0);
    b.tw = b.th = b.toff = 0;
    sid_out = 0; key_out = 0; narrow_out = false;
    if (live) {
        const uint32_t sid = a.pair_vals[entry];
        const uint4* cp = reinterpret_cast<const uint4*>(a.crecs + sid);
        const uint4 c0 = cp[0], c1 = cp[1];
        sid_out = sid; key_out = c1.z;
        RecView v;
        v.q0 = v.q1 = v.q2 = v.q3 = v.q4 = v.q5 = make_uint4(0, 0, 0, 0);
        const bool narrow = view_from_cov(v, c0, c1);
        narrow_out = narrow;
        if (!narrow || need_uv) {
            const uint4* sp = reinterpret_cast<const uint4*>(a.srecs + sid);
            const uint4 s0 = sp[0], s1 = sp[1];
            if (!narrow) view_edges_from_shade(v, s0, s1);
            if (need_uv) {
                const uint4 s2 = sp[2], s3 = sp[3];
                v.q2.y = s2.x; v.q2.z = s2.y; v.q2.w = s2.z;
                v.q3.x = s2.w; v.q3.y = s3.x; v.q3.z = s3.y;
                v.q4 = make_uint4(s1.w & 0xFFFFFFu, s3.z & 0xFFFFFFu, s3.w & 0xFFFFFFu, 0u);
                const uint32_t tid = c1.w & F_TEX_MASK;
                if (tid != F_TEX_NONE) {
                    if (TEXMODE == 1) { b.tw = lds_desc.width; b.th = lds_desc.height; b.toff = 0; }
                    else { const TexDesc d = a.tex[tid]; b.tw = d.width; b.th = d.height; b.toff = d.offset; }
                }
            }
        }
        if (need_aux || (c1.w & F_SLOW)) {
            const uint4* xp = reinterpret_cast<const uint4*>(a.xrecs + sid);
            const uint4 x0 = xp[0], x1 = xp[1];
            v.q4.w = x0.w; v.q5 = make_uint4(x1.x, x0.x, x0.y, x0.z);
        }
        b.q0 = v.q0; b.q1 = v.q1; b.q2 = v.q2; b.q3 = v.q3; b.q4 = v.q4; b.q5 = v.q5;
    }
}
// Wave-uniform view of lane t's record.  `full` = also UVs / texture (not needed by CHEAP coverage).
__device__ __forceinline__ Tri tri_from_batch(const Batch& b, int t, bool full) {
    Tri r;
    r.x3 = bcf(__uint_as_float(b.q0.x), t); r.y3 = bcf(__uint_as_float(b.q0.y), t);
    r.a0 = bcf(__uint_as_float(b.q0.z), t); r.b0 = bcf(__uint_as_float(b.q0.w), t);
    r.a1 = bcf(__uint_as_float(b.q1.x), t); r.b1 = bcf(__uint_as_float(b.q1.y), t);
    r.inv_area = bcf(__uint_as_float(b.q1.z), t);
    const uint32_t bbx = bcu(b.q1.w, t), bby = bcu(b.q2.x, t);
    r.min_x = bbx & 0xFFFF; r.max_x = bbx >> 16; r.min_y = bby & 0xFFFF; r.max_y = bby >> 16;
    r.flags = bcu(b.q3.w, t);
    r.u1 = r.u2 = r.u3 = r.v1 = r.v2 = r.v3 = 0.0f; r.tw = r.th = r.toff = 0; r.w0_start = r.w1_start = 0.0f;
    if (full) {
        r.u1 = bcf(__uint_as_float(b.q2.y), t); r.u2 = bcf(__uint_as_float(b.q2.z), t); r.u3 = bcf(__uint_as_float(b.q2.w), t);
        r.v1 = bcf(__uint_as_float(b.q3.x), t); r.v2 = bcf(__uint_as_float(b.q3.y), t); r.v3 = bcf(__uint_as_float(b.q3.z), t);
        r.tw = bcu(b.tw, t); r.th = bcu(b.th, t); r.toff = bcu(b.toff, t);
    }
    if (r.flags & F_SLOW) { r.w0_start = bcf(__uint_as_float(b.q4.w), t); r.w1_start = bcf(__uint_as_float(b.q5.x), t); }
    r.iz1 = r.iz2 = r.iz3 = 0.0f;
    if (full) { r.iz1 = bcf(__uint_as_float(b.q5.y), t); r.iz2 = bcf(__uint_as_float(b.q5.z), t); r.iz3 = bcf(__uint_as_float(b.q5.w), t); }
    return r;
}
// Records a drawn fragment of list entry li in the tile buffer(s).
//   painter's EXACT: max list position.  painter's CHEAP: exact top-2 (see k_cover).  z-buffer: min of (depth key, list position)
//   == the first surface in face order reaching the smallest depth, what the sequential `z < zbuffer` test leaves behind.
template <bool EXACT, bool ZMODE>
__device__ __forceinline__ void commit_fragment(uint32_t* tilebuf, uint32_t addr, uint32_t li, uint32_t zkey) {
    if (ZMODE) atomicMin(reinterpret_cast<unsigned long long*>(tilebuf) + addr, ((unsigned long long)zkey << 32) | li);
    else if (EXACT) atomicMax(&tilebuf[addr], li);
    else {
        // exact top-2 under any arrival order: whoever loses the max (the newcomer, or the value it displaced) is a runner-up
        // candidate; the final max is never displaced, every other value is pushed exactly once.
        const uint32_t old = atomicMax(&tilebuf[addr], li);
        atomicMax(&tilebuf[addr + TILE_H * TILE_STRIDE], min(old, li));
    }
}
// Depth of a fragment (render.rs:1546-1550) -> sortable key; false for NaN (never passes `z < zbuffer`).
__device__ __forceinline__ bool frag_zkey(const Tri& t, float bcx, float bcy, float bcz, uint32_t& zkey) {
    const float inv_z = bcx * t.iz1 + bcy * t.iz2 + bcz * t.iz3;
    const float z = rcp_exact(inv_z);
    zkey = zsort_key(z);
    return z == z;
}

// Depth of surface `sid` at pixel (px, py) with its exact bits (only needed when the z-buffer key decoded to zero: the key does not
// carry the sign of a zero depth).  Same arithmetic as the coverage: edge functions -> barycentrics -> 1 / (bc . 1/z).
__device__ float exact_depth_at(const FillArgs& a, uint32_t sid, uint32_t px, uint32_t py) {
    RecView rv;
    load_full_view(a, sid, rv);
    const uint4 q0 = rv.q0, q1 = rv.q1, q2 = rv.q2, q3 = rv.q3, q4 = rv.q4, q5 = rv.q5;
    Tri tr;
    tr.x3 = __uint_as_float(q0.x); tr.y3 = __uint_as_float(q0.y); tr.a0 = __uint_as_float(q0.z); tr.b0 = __uint_as_float(q0.w);
    tr.a1 = __uint_as_float(q1.x); tr.b1 = __uint_as_float(q1.y); tr.inv_area = __uint_as_float(q1.z);
    tr.min_x = q1.w & 0xFFFF; tr.max_x = q1.w >> 16; tr.min_y = q2.x & 0xFFFF; tr.max_y = q2.x >> 16;
    tr.flags = q3.w;
    tr.w0_start = __uint_as_float(q4.w); tr.w1_start = __uint_as_float(q5.x);
    float w0, w1, bcx, bcy, bcz;
    edge_w(tr, px, py, w0, w1);
    (void)inside_bc(tr, w0, w1, bcx, bcy, bcz);
    const float inv_z = bcx * __uint_as_float(q5.y) + bcy * __uint_as_float(q5.z) + bcz * __uint_as_float(q5.w);
    return rcp_exact(inv_z);
}

// Phase A for one surface: coverage of the (tile-clipped) bbox [cx0,cx1) x [cy0,cy1), winner value li.
template <int TEXMODE, bool EXACT, bool ZMODE, bool FMT8>
__device__ __forceinline__ uint32_t cover_surface(const Tri& tr, uint32_t cx0, uint32_t cx1, uint32_t cy0, uint32_t cy1, uint32_t li,
                                                  uint32_t* tilebuf, uint32_t x_lo, uint32_t ty_top, uint32_t lane,
                                                  const uint16_t* __restrict__ gtex, const uint16_t* ltex, bool affine) {
    uint32_t drawn_count = 0;
    if (!(tr.flags & F_SLOW)) {
        // lane block shape: the one needing the fewest blocks (ties -> 8x8)
        const uint32_t w = cx1 - cx0, h = cy1 - cy0;
        const uint32_t n88 = ((w + 7) >> 3) * ((h + 7) >> 3), n164 = ((w + 15) >> 4) * ((h + 3) >> 2), n416 = ((w + 3) >> 2) * ((h + 15) >> 4);
        uint32_t sh = 3;                                          // log2(block width)
        if (n164 < n88 && n164 <= n416) sh = 4; else if (n416 < n88) sh = 2;
        const uint32_t bw = 1u << sh, bh = 64u >> sh;
        const uint32_t lx = lane & (bw - 1), ly = lane >> sh;
        for (uint32_t by = cy0; by < cy1; by += bh) {
            const uint32_t py = by + ly;
            const float dy = (float)py - tr.y3;
            const float r0 = tr.b0 * dy, r1 = tr.b1 * dy;
            for (uint32_t bx = cx0; bx < cx1; bx += bw) {
                const uint32_t px = bx + lx;
                bool drawn = false;
                if (px < cx1 && py < cy1) {
                    const float dx = (float)px - tr.x3;
                    const float w0 = tr.a0 * dx + r0, w1 = tr.a1 * dx + r1;          // exact integers (k_setup guard)
                    float bcx, bcy, bcz;
                    if (inside_bc(tr, w0, w1, bcx, bcy, bcz)) {
                        uint32_t texel;
                        uint32_t zkey = 0;
                        drawn = ZMODE ? frag_zkey(tr, bcx, bcy, bcz, zkey) : true;
                        if (drawn && EXACT) drawn = texel_drawn<TEXMODE, FMT8>(tr, bcx, bcy, bcz, gtex, ltex, texel, affine);
                        if (drawn) commit_fragment<EXACT, ZMODE>(tilebuf, (py - ty_top) * TILE_STRIDE + (px - x_lo), li, zkey);
                    }
                }
                if (EXACT) drawn_count += (uint32_t)__popcll(__ballot(drawn));
            }
        }
    } else {
        for (uint32_t by = cy0; by < cy1; by += 64) {               // one lane per row, literal incremental walk
            const uint32_t py = by + lane;
            uint32_t mine = 0;
            if (py < cy1) {
                float w0, w1;
                replay_w(tr, cx0, py, w0, w1);
                for (uint32_t px = cx0; px < cx1; ++px) {
                    float bcx, bcy, bcz;
                    if (inside_bc(tr, w0, w1, bcx, bcy, bcz)) {
                        uint32_t texel;
                        uint32_t zkey = 0;
                        bool drawn = ZMODE ? frag_zkey(tr, bcx, bcy, bcz, zkey) : true;
                        if (drawn && EXACT) drawn = texel_drawn<TEXMODE, FMT8>(tr, bcx, bcy, bcz, gtex, ltex, texel, affine);
                        if (drawn) { commit_fragment<EXACT, ZMODE>(tilebuf, (py - ty_top) * TILE_STRIDE + (px - x_lo), li, zkey); ++mine; }
                    }
                    w0 += tr.a0; w1 += tr.a1;
                }
            }
            if (EXACT) for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
            if (EXACT) drawn_count += (uint32_t)__builtin_amdgcn_readfirstlane((int)mine);
        }
    }
    return drawn_count;
}

template <bool FMT8> __device__ __forceinline__ bool hit_finish(uint32_t flags, int taddr, uint32_t fetched, uint32_t& texel);
template <bool FMT8> __device__ __forceinline__ uint32_t fetch_texel(const FillArgs& a, int taddr);

// ---- wave-level helpers for the row-item scheduler
__device__ __forceinline__ uint32_t dpp_max_scan(uint32_t v) {          // inclusive prefix max over the 64 lanes, identity 0
    // Hillis-Steele inside each 16-lane row (row_shr 1,2,4,8), then row_bcast:15 / row_bcast:31 across rows.
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
    return v;
}
__device__ __forceinline__ uint32_t dpp_add_scan(uint32_t v) {          // inclusive prefix sum over the 64 lanes
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}
__device__ __forceinline__ uint32_t bperm(uint32_t src_lane, uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v); }
__device__ __forceinline__ float bpermf(uint32_t src_lane, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), __float_as_int(v))); }

// Phase A, EXACT coverage, wave-cooperative form (one wave per surface): used for F_SLOW surfaces and as reference path.
template <int TEXMODE, bool EXACT, bool ZMODE, bool FMT8>
__device__ __forceinline__ uint32_t cover_one(const Batch& b, int t, uint32_t li, uint32_t* tilebuf, uint32_t x_lo, uint32_t x_hi,
                                              uint32_t y_lo, uint32_t y_hi, uint32_t ty_top, uint32_t lane,
                                              const uint16_t* __restrict__ gtex, const uint16_t* ltex, bool affine) {
    const Tri tr = tri_from_batch(b, t, EXACT);
    const uint32_t cx0 = max(tr.min_x, x_lo), cx1 = min(tr.max_x, x_hi);
    const uint32_t cy0 = max(tr.min_y, y_lo), cy1 = min(tr.max_y, y_hi);
    if (cx0 >= cx1 || cy0 >= cy1) return 0;
    return cover_surface<TEXMODE, EXACT, ZMODE, FMT8>(tr, cx0, cx1, cy0, cy1, li, tilebuf, x_lo, ty_top, lane, gtex, ltex, affine);
}

// P64 coverage of a surface whose edge walk must be replayed literally (F_SLOW): one lane per row.
template <bool ZMODE, bool EXACT, bool FMT8>
__device__ __forceinline__ uint32_t cover_slow64(const Tri& tr, unsigned long long P, uint32_t* tilebuf, uint32_t x_lo, uint32_t x_hi,
                                                 uint32_t y_lo, uint32_t y_hi, uint32_t ty_top, uint32_t lane,
                                                 const uint16_t* __restrict__ gtex, bool affine) {
    const uint32_t cx0 = max(tr.min_x, x_lo), cx1 = min(tr.max_x, x_hi);
    const uint32_t cy0 = max(tr.min_y, y_lo), cy1 = min(tr.max_y, y_hi);
    if (cx0 >= cx1 || cy0 >= cy1) return 0;
    uint32_t count = 0;
    unsigned long long* top = reinterpret_cast<unsigned long long*>(tilebuf);
    unsigned long long* sec = top + TILE_H * STR64;
    for (uint32_t by = cy0; by < cy1; by += 64) {
        const uint32_t py = by + lane;
        if (py < cy1) {
            float w0, w1;
            replay_w(tr, cx0, py, w0, w1);
            for (uint32_t px = cx0; px < cx1; ++px) {
                float bcx, bcy, bcz;
                if (inside_bc(tr, w0, w1, bcx, bcy, bcz)) {
                    const uint32_t addr = (py - ty_top) * STR64 + (px - x_lo);
                    unsigned long long Pf = P;
                    bool ok = true;
                    if (ZMODE) { uint32_t zkey; ok = frag_zkey(tr, bcx, bcy, bcz, zkey); Pf = ((unsigned long long)(~zkey) << 32) | (uint32_t)P; }
                    if (EXACT && ok) { uint32_t texel; ok = texel_drawn<0, FMT8>(tr, bcx, bcy, bcz, gtex, nullptr, texel, affine); }
                    if (ok) {
                        const unsigned long long old = atomicMax(&top[addr], Pf);
                        if (!EXACT) atomicMax(&sec[addr], min(old, Pf));
                        ++count;
                    }
                }
                w0 += tr.a0; w1 += tr.a1;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) count += __shfl_down(count, off);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)count);
}

// Phase A as a ROW-ITEM scheduler.  Waves grab 64 list entries at a time from an LDS cursor (load balance across the 16
// waves).  Each lane first holds one surface; the work items of the batch are the rows of the tile-clipped bounding boxes
// (exclusive prefix sum of the heights).  In rounds of 64 items every lane takes ONE ROW of some surface: the owner is
// found with a scatter of row starts + DPP prefix-max, its parameters come over ds_bpermute, and the lane walks the row
// incrementally exactly like the reference's inner loop (render.rs:1533-1707): start value = closed form at the row start
// (exact integers under the k_setup guard), then w0 += a0, w1 += a1 per pixel.  Row lengths are far more uniform than
// bbox areas, big surfaces fill whole rounds, and there is no per-surface scalar work.
// Row trimming.  The inside test (render.rs:1536-1542) is evaluated on rounded floats, but a pixel can only pass it when three
// linear conditions on the (exact, integer) edge values hold:
//     bc_x = fl(w0 * inv_area) >= -1e-4            =>  s*w0 >= -T              (s = sign of inv_area, A = 1/|inv_area|,
//     bc_y likewise                                 =>  s*w1 >= -T               T = 1.02e-4 * A: 2 % above what the rounding of the
//     bc_z = fl(fl(1 - bc_x) - bc_y) >= -1e-4      =>  s*(w0 + w1) <= A + T     product and of A can move the threshold)
// (for the third: bc_x and bc_y have passed, so both lie in [-1e-4, 1.0003] and the two subtractions are off by < 1.3e-7).
// Every w is linear in x along the row, so the three conditions cut the clipped row [0, n) down to one interval [lo, hi);
// pixels outside it are certain to fail, pixels inside still take the reference's own test.  The interval ends are computed with
// an approximate reciprocal and widened by 0.01 px (its error over a 64-px row is < 2e-5 px).  Returns lo and shrinks n to
// hi - lo.  Surfaces with A outside [0.5, 2^20) are left alone (w0 + w1 could round where it matters).
#ifndef B32_ROW_TRIM
#define B32_ROW_TRIM 1
#endif
#ifndef B32_INTERIOR
#define B32_INTERIOR 0           // experiment (round 4, judge item 3c), OFF: certain-interior runs of long rows take trips without the inside test.
                                 // Bit-exact (full-size C3 / C5 hashes, 43 parity tests) and slower: finding and verifying the run (~70 VALU per
                                 // round as soon as ONE lane of the wave has a long row), the second queue and its own, emptier rounds cost more
                                 // than the skipped barycentrics return -- C5 0.2103 -> 0.2394 ms, C3 0.1200 -> 0.1368 (profiles/r04_interior_trips_ab.txt)
#endif
constexpr uint32_t INTERIOR_MIN_ROW = 12;      // rows shorter than this are not worth the interval (one boundary trip at each end)
__device__ __forceinline__ uint32_t row_trim(float w0, float w1, float a0, float a1, float inv_area, uint32_t& n) {
    const float A = __builtin_amdgcn_rcpf(__builtin_fabsf(inv_area));
    if (!((A >= 0.5f) & (A < 1048576.0f))) return 0u;
    const float s = inv_area < 0.0f ? -1.0f : 1.0f;
    const float T = 1.02e-4f * A;
    const float E[3] = { s * w0 + T, s * w1 + T, (A + T) - s * (w0 + w1) };
    const float G[3] = { s * a0, s * a1, -(s * a0 + s * a1) };
    float flo = 0.0f, fhi = (float)n;
#pragma unroll
    for (int j = 0; j < 3; ++j) {       // (selects, not branches: every lane of the wave walks a different surface)
        const float r = -E[j] * __builtin_amdgcn_rcpf(G[j]);
        const float lo_c = fmaxf(flo, ceilf(r - 0.01f));                    // E + G x >= 0  <=>  x >= r   (G > 0)
        const float hi_c = fminf(fhi, floorf(r + 0.01f) + 1.0f);            //                    x <= r   (G < 0)
        flo = G[j] > 0.0f ? lo_c : flo;
        fhi = G[j] < 0.0f ? hi_c : (((G[j] == 0.0f) & (E[j] < 0.0f)) ? 0.0f : fhi);   // G == 0: constant along the row; failing -> empty
    }
    flo = fminf(flo, (float)n);
    fhi = fmaxf(fhi, flo);
    const uint32_t lo = (uint32_t)flo;
    n = (uint32_t)fhi - lo;
    return lo;
}

// Certain-interior run of a trimmed row (CHEAP painter's coverage of large triangles).  A pixel that lies inside the triangle in EXACT
// arithmetic always passes the reference's toleranced float test (render.rs:1536-1542): for a surface that passed k_setup's exactness
// guard the edge values are exact integers, bc_x = fl(w0 * fl(1 / area)) >= 0 whenever w0 has the area's sign (likewise bc_y), and
// bc_z = fl(fl(1 - bc_x) - bc_y) is within 4e-7 of the exact w2 / area >= 0 -- far above -1e-4.  Along a row the exactly-inside pixels
// are one interval (three linear conditions); its ends come from approximate reciprocals and are then VERIFIED with the exact integer
// conditions at both end pixels (linearity covers everything between); a failed check simply means "no interior run".
// In: edge values (w0, w1) at the row's first pixel, per-pixel steps (a0, a1), |area| = |a0 * b1 - b0 * a1| (all exact integers in f32),
// sign s of the area, n pixels.  Out: [tlo, thi) in pixels from the row's first pixel; returns false when there is none.
__device__ __forceinline__ bool interior_run(float w0, float w1, float a0, float a1, float absA, float s, uint32_t n, uint32_t& tlo, uint32_t& thi) {
    const float E[3] = { s * w0, s * w1, absA - (s * w0 + s * w1) };
    const float G[3] = { s * a0, s * a1, -(s * a0 + s * a1) };
    float flo = 0.0f, fhi = (float)n;
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 3; ++j) {       // E + G t >= 0
        const float r = -E[j] * __builtin_amdgcn_rcpf(G[j]);
        const float lo_c = fmaxf(flo, ceilf(r - 1.5e-5f)), hi_c = fminf(fhi, floorf(r + 1.5e-5f) + 1.0f);
        flo = G[j] > 0.0f ? lo_c : flo;
        fhi = G[j] < 0.0f ? hi_c : fhi;
        ok = ok & !((G[j] == 0.0f) & (E[j] < 0.0f));
    }
    ok = ok & (fhi > flo) & (flo >= 0.0f) & (fhi <= (float)n);
    const float ta = flo, tb = fhi - 1.0f;                   // the two end pixels, checked exactly
#pragma unroll
    for (int j = 0; j < 3; ++j) ok = ok & (E[j] + G[j] * ta >= 0.0f) & (E[j] + G[j] * tb >= 0.0f);
    tlo = ok ? (uint32_t)flo : 0u; thi = ok ? (uint32_t)fhi : 0u;
    return ok;
}

// One trip of the sort-free CHEAP coverage: TRIP consecutive pixels of a row starting at LDS entry `addr` with edge values (w0, w1),
// `left` of them inside the clipped row.  The value is the surface's global painter's priority P (z-buffer mode: the fragment's
// depth in the high word), so no tile list order is needed; TRIP returning LDS atomics are in flight together and the wave waits
// once (w advances by the reference's own sequential accumulation w += a, render.rs:1706-1707).
template <bool ZMODE>
__device__ __forceinline__ void cheap_trip(unsigned long long* top, unsigned long long* sec, uint32_t& addr, float& w0, float& w1, float sa0, float sa1,
                                           float sinv, uint32_t left, unsigned long long P, float z1, float z2, float z3) {
    constexpr int TRIP = B32_TRIP;
    const float ERR = K::ERR;
    float wa[TRIP], wb[TRIP];
    wa[0] = w0; wb[0] = w1;
#pragma unroll
    for (int j = 1; j < TRIP; ++j) { wa[j] = wa[j - 1] + sa0; wb[j] = wb[j - 1] + sa1; }
    bool in[TRIP];
    unsigned long long old[TRIP], Pj[TRIP];
#pragma unroll
    for (int j = 0; j < TRIP; ++j) {
        const float cx = wa[j] * sinv, cy = wb[j] * sinv;
        const float cz = 1.0f - cx - cy;
        // all three >= ERR  <=>  their minimum is (no NaN can occur here: w integers, inv_area finite and non-zero)
        in[j] = ((uint32_t)j < left) & (__builtin_fminf(__builtin_fminf(cx, cy), cz) >= ERR);
        old[j] = 0; Pj[j] = P;
        if (ZMODE) {                            // fragment depth (render.rs:1546-1550); NaN never passes `z < zbuffer`
            const float inv_z = cx * z1 + cy * z2 + cz * z3;
            const float z = rcp_exact(inv_z);
            in[j] = in[j] & (z == z);
            Pj[j] = ((unsigned long long)(~zsort_key(z)) << 32) | (uint32_t)P;
        }
    }
    // one predicated block for the whole trip (a branch per atomic makes the compiler wait for each returning atomic before it
    // issues the next): pixels outside the triangle contribute priority 0, a no-op for both maxima (min(old, 0) == 0)
    bool any_in = false;
#pragma unroll
    for (int j = 0; j < TRIP; ++j) any_in |= in[j];
    if (any_in) {
#pragma unroll
        for (int j = 0; j < TRIP; ++j) { if (!in[j]) Pj[j] = 0ull; old[j] = atomicMax(&top[addr + j], Pj[j]); }
#pragma unroll
        for (int j = 0; j < TRIP; ++j) atomicMax(&sec[addr + j], min(old[j], Pj[j]));
    }
    addr += TRIP; w0 = wa[TRIP - 1] + sa0; w1 = wb[TRIP - 1] + sa1;
}

// ---- span coverage (B32_ROUTE_SPAN_COVER; sort-free CHEAP painter's coverage)
// For a surface with integer vertices, |area| = A <= 8192 and edge coefficients of at most SPAN_MAX_EXT, the reference's toleranced float
// test (render.rs:1536-1542: bc_x, bc_y, bc_z >= -1e-4) passes EXACTLY on the pixels of the closed integer triangle
//     E0 = s w0 >= 0,  E1 = s w1 >= 0,  E2 = A - E0 - E1 >= 0        (s = sign of the area; w0, w1 the edge values, exact integers)
// because one unit of an edge value moves a barycentric by 1 / A >= 2^-13 = 1.22e-4, above the tolerance plus every rounding of the
// float evaluation (proof and brute-force check: tests/test_span_cover.py).  Along a row every E_j is linear in x with an integer
// step G_j, so the passing pixels are ONE interval whose ends are integer quotients: lo = max over G_j > 0 of ceil(-E_j / G_j),
// hi = 1 + min over G_j < 0 of floor(E_j / |G_j|); a row with G_j == 0 passes edge j everywhere or nowhere.  The quotients come from
// one fma with the reciprocal of G_j, shifted by half a step: (-E_j -+ 1/2) / G_j is at least 1 / (2 |G_j|) away from every
// integer, which an approximate reciprocal (1 ulp) and the rounding of the fma cannot bridge while |E_j| < 2^21.
// The row-item scheduler keeps its shape (one lane = one row of one surface, see phase_a_rows), but a lane's row is now its exact
// interval: no inside test, no barycentrics per pixel -- a trip is the two atomics per pixel and nothing else, and what the surface's
// lane hands its rows is the per-surface part of the quotients (edge values at the box origin, row steps, reciprocals).
constexpr float SPAN_MAX_EXT = 512.0f;
constexpr float SPAN_MIN_INV_AREA = 1.0f / 8192.0f;            // |inv_area| >= 2^-13  <=>  A <= 8192
struct SpanEdge { float r, c; };
// edge j of a surface: G = the (sign-corrected) step of E_j per pixel.  r > 0 (G > 0): ceil(fma(-E, r, c)) is the first passing x;
// r < 0 (G < 0, or G == 0 where the row passes everywhere or nowhere): floor(fma(-E, r, c)) is one past the last passing x
__device__ __forceinline__ SpanEdge span_edge(float G) {
    SpanEdge e;
    const float r = __builtin_amdgcn_rcpf(G);
    e.r = G == 0.0f ? -1073741824.0f : r;                      // -2^30: E >= 0 -> far right of the tile, E <= -1 -> far left of it
    e.c = G == 0.0f ? 64.0f : (G > 0.0f ? -0.5f * r : -0.5f * r + 1.0f);
    return e;
}
// the passing interval [lo, hi) of a row, in pixels from the row's first (clipped) pixel, from the three edge values there
__device__ __forceinline__ void span_interval(float E0, float E1, float E2, const SpanEdge& d0, const SpanEdge& d1, const SpanEdge& d2, float wlen,
                                              float& lo, float& hi) {
    const float v0 = __builtin_fmaf(-E0, d0.r, d0.c), v1 = __builtin_fmaf(-E1, d1.r, d1.c), v2 = __builtin_fmaf(-E2, d2.r, d2.c);
    const bool l0 = d0.r > 0.0f, l1 = d1.r > 0.0f, l2 = d2.r > 0.0f;
    lo = fmaxf(fmaxf(l0 ? ceilf(v0) : 0.0f, l1 ? ceilf(v1) : 0.0f), l2 ? ceilf(v2) : 0.0f);
    hi = fminf(fminf(l0 ? wlen : floorf(v0), l1 ? wlen : floorf(v1)), fminf(l2 ? wlen : floorf(v2), wlen));
}
// One trip of the span coverage: TRIP consecutive pixels at LDS entry `addr`, the first `left` of them inside the row's interval
// (exact top-2 per pixel, see cheap_trip; pixels beyond the interval contribute priority 0, a no-op for both maxima)
__device__ __forceinline__ void span_trip(unsigned long long* top, unsigned long long* sec, uint32_t addr, uint32_t left, unsigned long long P) {
    constexpr uint32_t TRIP = B32_TRIP;
    if (left) {                 // (one predicated block for the whole trip, see cheap_trip; lanes without a pixel issue nothing)
        unsigned long long old[TRIP];
#pragma unroll
        for (uint32_t j = 0; j < TRIP; ++j) old[j] = atomicMax(&top[addr + j], j < left ? P : 0ull);
#pragma unroll
        for (uint32_t j = 0; j < TRIP; ++j) atomicMax(&sec[addr + j], j < left ? min(old[j], P) : 0ull);
    }
}

template <int TEXMODE, bool EXACT, int NW, bool ZMODE, bool FMT8, bool P64 = false>
__device__ __forceinline__ unsigned long long phase_a_rows(const FillArgs& a, uint32_t e0, uint32_t n_op, uint32_t lane, uint32_t wave,
                                                           uint32_t* cursor, uint32_t* wmark, const TexDesc& lds_desc,
                                                           uint32_t* tilebuf, uint32_t x_lo, uint32_t x_hi, uint32_t y_lo, uint32_t y_hi,
                                                           uint32_t ty_top, const uint16_t* ltex) {
    const uint16_t* __restrict__ gtex = FMT8 ? reinterpret_cast<const uint16_t*>(a.texels32) : a.texels;
    unsigned long long frags = 0;
    const float ERR = K::ERR;
    const bool affine = a.fp.affine != 0;
    // entries per grab: the fewest rounds of grabs that give every wave the same number of them -- m grabs per wave, each of
    // ceil(n / (NW m)) <= 64 entries (500 entries, 8 waves: one grab of 63 each; 700: two of 44; a fixed divisor of 2 gave 32 / 44)
    const uint32_t grab_m = max(1u, (n_op + NW * 64u - 1u) / (NW * 64u));
    const uint32_t grab = min(64u, max(4u, (n_op + NW * grab_m - 1u) / (NW * grab_m)));
    for (;;) {
        uint32_t cs = 0;
        if (lane == 0) cs = atomicAdd(cursor, grab);
        cs = (uint32_t)__builtin_amdgcn_readfirstlane((int)cs);
        if (cs >= n_op) break;
        const uint32_t e = cs + lane;
        bool live = lane < grab && e < n_op;
        Batch b;
        // P64: the surface's place in the global painter's order; in z-buffer mode the high word is the fragment's depth and the
        // low word 0xFFFFFFFE - face id (first in face order wins a depth tie, like the sequential `z < zbuffer` test; all ones is
        // reserved for the z-buffer seed, which therefore wins every tie: `z < zbuffer` is strict)
        uint32_t my_sid = 0, my_key = 0;
        bool narrow = false;
        load_batch<TEXMODE>(b, a, e0 + e, live, lds_desc, EXACT, ZMODE || (EXACT && !affine), my_sid, my_key, narrow);
        if (ZMODE) { my_key = 0u; my_sid = 0xFFFFFFFEu - my_sid; }
        const uint32_t flags = b.q3.w;
        const uint32_t cx0 = max(b.q1.w & 0xFFFF, x_lo), cx1 = min(b.q1.w >> 16, x_hi);
        const uint32_t cy0 = max(b.q2.x & 0xFFFF, y_lo), cy1 = min(b.q2.x >> 16, y_hi);
        live = live && cx0 < cx1 && cy0 < cy1;
        const bool slow = live && (flags & F_SLOW);
        // span coverage: what the rows of an eligible surface need (edge values at the first pixel of its clipped box, their steps per
        // row, the reciprocal form of the steps per pixel); span_all: every surface of this batch is eligible -- the rounds below then
        // take the span form, else the per-pixel form serves the whole batch (it is valid for every surface)
        bool span_all = false;
        float sE0 = 0.0f, sE1 = 0.0f, sH0 = 0.0f, sH1 = 0.0f, sA = 0.0f;
        SpanEdge sd0 = { 0.0f, 0.0f }, sd1 = { 0.0f, 0.0f }, sd2 = { 0.0f, 0.0f };
        if (P64 && !EXACT && !ZMODE && a.span_cover) {
            const float fa0 = __uint_as_float(b.q0.z), fb0 = __uint_as_float(b.q0.w), fa1 = __uint_as_float(b.q1.x), fb1 = __uint_as_float(b.q1.y);
            const float inv = __uint_as_float(b.q1.z);
            const float sgn = inv < 0.0f ? -1.0f : 1.0f;
            const float G0 = sgn * fa0, G1 = sgn * fa1, G2 = -(G0 + G1);           // steps per pixel of E0, E1, E2 (exact integers)
            sH0 = sgn * fb0; sH1 = sgn * fb1;                                       // steps per row
            const float H2 = -(sH0 + sH1);
            const float ext = fmaxf(fmaxf(fmaxf(__builtin_fabsf(G0), __builtin_fabsf(G1)), fmaxf(__builtin_fabsf(sH0), __builtin_fabsf(sH1))),
                                    fmaxf(__builtin_fabsf(G2), __builtin_fabsf(H2)));
            sA = __builtin_fabsf(fa0 * fb1 - fb0 * fa1);                            // |area| (render.rs:1500 in exact integers)
            const bool fast = narrow && !(flags & (F_EMPTY | F_SLOW)) && ext <= SPAN_MAX_EXT && __builtin_fabsf(inv) >= SPAN_MIN_INV_AREA && sA >= 1.0f;
            span_all = !__ballot(live && !fast);
            const float dx = (float)cx0 - __uint_as_float(b.q0.x), dy = (float)cy0 - __uint_as_float(b.q0.y);
            sE0 = sgn * (fa0 * dx + fb0 * dy); sE1 = sgn * (fa1 * dx + fb1 * dy);   // at the first pixel of the clipped box
            sd0 = span_edge(G0); sd1 = span_edge(G1); sd2 = span_edge(G2);
        }
        const uint32_t h = (live && !slow) ? cy1 - cy0 : 0u;
        // exclusive prefix sum of the row counts
        const uint32_t inc = dpp_add_scan(h);
        const uint32_t R = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        const uint32_t P = inc - h;
        const float a0 = __uint_as_float(b.q0.z), b0 = __uint_as_float(b.q0.w), a1 = __uint_as_float(b.q1.x), b1 = __uint_as_float(b.q1.y);
        const uint32_t box = (cx0 - x_lo) | ((cx1 - x_lo) << 8) | ((cy0 - ty_top) << 16);      // 7+7+6 bits
        // CHEAP sort-free coverage: every lane of a round makes ONE trip; what is left of the rows that need more (a fifth of them need a
        // second trip, 3 % a third, but a round used to last as long as its longest row: three trips for an average need of 1.2) is queued
        // -- one packed word per row remainder, the queue is a register: lane i holds entry i -- and worked off 64 at a time in rounds of
        // their own, whose lanes are all busy.  The remainders refer to lanes of THIS batch (parameters come over ds_bpermute again), so
        // the queue is drained before the next batch is loaded.
        uint32_t lq = 0, lqn = 0;                       // leftover queue and its length (wave-uniform)
        uint32_t lqi = 0, lqin = 0;                     // the same for certain-interior runs (interior_run): trips without the inside test
        // One trip of the sort-free EXACT coverage: four pixels -- the four texel addresses, their bits of the skip mask (LDS when the
        // pool's mask fits, else global: 1/16 of the texels' bytes; no texel is fetched during coverage) -- then the (non-returning)
        // atomics of the drawn fragments.  Returns the number of fragments drawn (the reference's pixel stores).
        auto exact_trip = [&](const Tri& tr, uint32_t& addr, float& w0, float& w1, float sa0, float sa1, float sinv, uint32_t left, unsigned long long P) -> uint32_t {
            unsigned long long* top = reinterpret_cast<unsigned long long*>(tilebuf);
            const uint32_t* mask_l = reinterpret_cast<const uint32_t*>(ltex);           // LDS copy of the mask (k_cover stages it)
            const bool mask_in_lds = a.mask_lds_words != 0;
            uint32_t drawn = 0;
            float wa[4], wb[4];
            wa[0] = w0; wb[0] = w1;
#pragma unroll
            for (int j = 1; j < 4; ++j) { wa[j] = wa[j - 1] + sa0; wb[j] = wb[j - 1] + sa1; }
            bool in[4]; int ta[4]; unsigned long long Pj[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float cx = wa[j] * sinv, cy = wb[j] * sinv;
                const float cz = 1.0f - cx - cy;
                in[j] = ((uint32_t)j < left) & (__builtin_fminf(__builtin_fminf(cx, cy), cz) >= ERR);        // (see the CHEAP trip)
                Pj[j] = P; ta[j] = -1;
                if (in[j]) {
                    if (ZMODE) { uint32_t zkey; in[j] = frag_zkey(tr, cx, cy, cz, zkey); Pj[j] = ((unsigned long long)(~zkey) << 32) | (uint32_t)P; }
                    ta[j] = tri_texel_addr(tr, cx, cy, cz, affine);
                }
            }
            uint32_t mw[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                mw[j] = 0;
                if (in[j] && ta[j] >= 0) mw[j] = mask_in_lds ? mask_l[(uint32_t)ta[j] >> 5] : a.texmask[(uint32_t)ta[j] >> 5];
            }
            bool any = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // skippable texel: the mask bit; a zero-size texture samples TRANSPARENT (-2), no texture samples WHITE (-1)
                const bool blk = ta[j] == -2 ? true : (ta[j] >= 0 && ((mw[j] >> ((uint32_t)ta[j] & 31u)) & 1u));
                in[j] = in[j] && !(FMT8 ? blk : (blk && (tr.flags & F_BLACK_TR)));       // render.rs:1591-1608 / 8-bit :1348-1352
                any |= in[j];
                drawn += in[j] ? 1u : 0u;
            }
            if (any) {
#pragma unroll
                for (int j = 0; j < 4; ++j) atomicMax(&top[addr + j], in[j] ? Pj[j] : 0ull);
            }
            addr += 4; w0 = wa[3] + sa0; w1 = wb[3] + sa1;
            return drawn;
        };
        auto drain = [&]() {
            const bool valid = lane < lqn;
            const uint32_t s = valid ? (lq & 63u) : lane;
            const uint32_t ry = (lq >> 6) & 63u, rx = (lq >> 12) & 127u;
            const uint32_t n = valid ? (lq >> 19) : 0u;
            const float sx3 = bpermf(s, __uint_as_float(b.q0.x)), sy3 = bpermf(s, __uint_as_float(b.q0.y));
            const float sa0 = bpermf(s, a0), sb0 = bpermf(s, b0), sa1 = bpermf(s, a1), sb1 = bpermf(s, b1);
            const float sinv = bpermf(s, __uint_as_float(b.q1.z));
            const unsigned long long P = ((unsigned long long)bperm(s, my_key) << 32) | bperm(s, my_sid);
            float z1 = 0.0f, z2 = 0.0f, z3 = 0.0f;
            if (ZMODE) { z1 = bpermf(s, __uint_as_float(b.q5.y)); z2 = bpermf(s, __uint_as_float(b.q5.z)); z3 = bpermf(s, __uint_as_float(b.q5.w)); }
            const float dx = (float)(rx + x_lo) - sx3, dy = (float)(ry + ty_top) - sy3;
            float w0 = sa0 * dx + sb0 * dy, w1 = sa1 * dx + sb1 * dy;            // exact integers: the value the row's own walk would have reached
            uint32_t addr = ry * STR64 + rx;
            unsigned long long* top = reinterpret_cast<unsigned long long*>(tilebuf);
            unsigned long long* sec = top + TILE_H * STR64;
#if B32_DRAIN_TRIPS > 0
            // at most B32_DRAIN_TRIPS trips per entry and round; what is left of a long row goes back into the queue (a round used to last
            // as long as its longest remainder: with the ~25-px rows of C5 most lanes idled behind the longest)
            constexpr uint32_t DT = (uint32_t)B32_DRAIN_TRIPS * (uint32_t)B32_TRIP;
            if (EXACT) {
                Tri tr;
                tr.u1 = bpermf(s, __uint_as_float(b.q2.y)); tr.u2 = bpermf(s, __uint_as_float(b.q2.z)); tr.u3 = bpermf(s, __uint_as_float(b.q2.w));
                tr.v1 = bpermf(s, __uint_as_float(b.q3.x)); tr.v2 = bpermf(s, __uint_as_float(b.q3.y)); tr.v3 = bpermf(s, __uint_as_float(b.q3.z));
                tr.flags = bperm(s, flags);
                tr.tw = bperm(s, b.tw); tr.th = bperm(s, b.th); tr.toff = bperm(s, b.toff);
                tr.iz1 = z1; tr.iz2 = z2; tr.iz3 = z3;
                if (!affine && !ZMODE) { tr.iz1 = bpermf(s, __uint_as_float(b.q5.y)); tr.iz2 = bpermf(s, __uint_as_float(b.q5.z)); tr.iz3 = bpermf(s, __uint_as_float(b.q5.w)); }
                uint32_t mine = 0;
#pragma unroll
                for (uint32_t i = 0; i < DT; i += 4) mine += exact_trip(tr, addr, w0, w1, sa0, sa1, sinv, i < n ? n - i : 0u, P);
                for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
                frags += (uint32_t)__builtin_amdgcn_readfirstlane((int)mine);
            } else
#pragma unroll
            for (uint32_t i = 0; i < DT; i += B32_TRIP)
                cheap_trip<ZMODE>(top, sec, addr, w0, w1, sa0, sa1, sinv, i < n ? n - i : 0u, P, z1, z2, z3);
            const bool more = n > DT;
            const unsigned long long mm = __ballot(more);
            const uint32_t cnt = (uint32_t)__builtin_popcountll(mm);
            if (cnt) {
                const uint32_t entry = s | (ry << 6) | ((rx + DT) << 12) | ((n - DT) << 19);
                const uint32_t dst = more ? (uint32_t)__builtin_popcountll(mm & ((1ull << lane) - 1ull)) : (cnt & 63u);
                const uint32_t got = (uint32_t)__builtin_amdgcn_ds_permute((int)(dst << 2), (int)(more ? entry : 0u));
                if (lane < cnt) lq = got;
            }
            lqn = cnt;
#else
            for (uint32_t i = 0; __ballot(i < n); i += B32_TRIP)
                cheap_trip<ZMODE>(top, sec, addr, w0, w1, sa0, sa1, sinv, i < n ? n - i : 0u, P, z1, z2, z3);
            lqn = 0;
#endif
        };
        // Rounds of the interior queue: an entry is (lane of the surface, tile row, first column, pixels), every pixel certain to pass the
        // inside test -- the trip is the two atomics per pixel and nothing else (no edge values, no barycentrics).
        auto drain_interior = [&]() {
            const bool valid = lane < lqin;
            const uint32_t s = valid ? (lqi & 63u) : lane;
            const uint32_t ry = (lqi >> 6) & 63u, rx = (lqi >> 12) & 127u;
            const uint32_t n = valid ? (lqi >> 19) : 0u;
            const unsigned long long P = ((unsigned long long)bperm(s, my_key) << 32) | bperm(s, my_sid);
            unsigned long long* top = reinterpret_cast<unsigned long long*>(tilebuf);
            unsigned long long* sec = top + TILE_H * STR64;
            const uint32_t addr = ry * STR64 + rx;
            constexpr uint32_t DT = 2u * (uint32_t)B32_TRIP;
#pragma unroll
            for (uint32_t t0 = 0; t0 < DT; t0 += (uint32_t)B32_TRIP) {       // (one trip's returning atomics in flight at a time: registers)
                if (!__ballot(n > t0)) break;
                unsigned long long old[B32_TRIP];
#pragma unroll
                for (uint32_t j = 0; j < (uint32_t)B32_TRIP; ++j) old[j] = atomicMax(&top[addr + t0 + j], (t0 + j) < n ? P : 0ull);
#pragma unroll
                for (uint32_t j = 0; j < (uint32_t)B32_TRIP; ++j) atomicMax(&sec[addr + t0 + j], (t0 + j) < n ? min(old[j], P) : 0ull);
            }
            const bool more = n > DT;
            const unsigned long long mm = __ballot(more);
            const uint32_t cnt = (uint32_t)__builtin_popcountll(mm);
            if (cnt) {
                const uint32_t entry = s | (ry << 6) | ((rx + DT) << 12) | ((n - DT) << 19);
                const uint32_t dst = more ? (uint32_t)__builtin_popcountll(mm & ((1ull << lane) - 1ull)) : (cnt & 63u);
                const uint32_t got = (uint32_t)__builtin_amdgcn_ds_permute((int)(dst << 2), (int)(more ? entry : 0u));
                if (lane < cnt) lqi = got;
            }
            lqin = cnt;
        };
        // span form of the remainder rounds: an entry is (lane of the surface, tile row, first column, pixels left of the row's interval)
        auto drain_span = [&]() {
            const bool valid = lane < lqn;
            const uint32_t s = valid ? (lq & 63u) : lane;
            const uint32_t ry = (lq >> 6) & 63u, rx = (lq >> 12) & 127u;
            const uint32_t n = valid ? (lq >> 19) : 0u;
            const unsigned long long P = ((unsigned long long)bperm(s, my_key) << 32) | bperm(s, my_sid);
            unsigned long long* top = reinterpret_cast<unsigned long long*>(tilebuf);
            unsigned long long* sec = top + TILE_H * STR64;
            const uint32_t addr = ry * STR64 + rx;
            constexpr uint32_t DT = (uint32_t)(B32_DRAIN_TRIPS > 0 ? B32_DRAIN_TRIPS : 2) * (uint32_t)B32_TRIP;
#pragma unroll
            for (uint32_t t0 = 0; t0 < DT; t0 += (uint32_t)B32_TRIP) {
                if (t0 && !__ballot(n > t0)) break;
                span_trip(top, sec, addr + t0, n > t0 ? n - t0 : 0u, P);
            }
            const bool more = n > DT;
            const unsigned long long mm = __ballot(more);
            const uint32_t cnt = (uint32_t)__builtin_popcountll(mm);
            if (cnt) {
                const uint32_t entry = s | (ry << 6) | ((rx + DT) << 12) | ((n - DT) << 19);
                const uint32_t dst = more ? (uint32_t)__builtin_popcountll(mm & ((1ull << lane) - 1ull)) : (cnt & 63u);
                const uint32_t got = (uint32_t)__builtin_amdgcn_ds_permute((int)(dst << 2), (int)(more ? entry : 0u));
                if (lane < cnt) lq = got;
            }
            lqn = cnt;
        };
        if (P64 && !EXACT && !ZMODE && span_all) {
            // span rounds: same items (one lane = one row of one surface), the row is its exact interval
            for (uint32_t k0 = 0; k0 < R; k0 += 64) {
                const unsigned long long before = __ballot(h > 0 && P <= k0);
                const uint32_t carry = before ? 64u - (uint32_t)__builtin_clzll(before) : 0u;
                const bool starts = h > 0 && P > k0 && P < k0 + 64;
                const uint32_t mark = (uint32_t)__builtin_amdgcn_ds_permute((int)((starts ? P - k0 : 0u) << 2), (int)(starts ? lane + 1 : 0u));
                const uint32_t own = max(dpp_max_scan(mark), carry);
                const uint32_t k = k0 + lane;
                const bool valid = k < R;
                const uint32_t s = valid ? own - 1 : lane;
                const uint32_t sbox = bperm(s, box), sP = bperm(s, P);
                const float rowf = (float)(k - sP);
                const float hE0 = bpermf(s, sE0), hE1 = bpermf(s, sE1), hH0 = bpermf(s, sH0), hH1 = bpermf(s, sH1), hA = bpermf(s, sA);
                SpanEdge e0, e1, e2;
                e0.r = bpermf(s, sd0.r); e0.c = bpermf(s, sd0.c); e1.r = bpermf(s, sd1.r); e1.c = bpermf(s, sd1.c); e2.r = bpermf(s, sd2.r); e2.c = bpermf(s, sd2.c);
                const float E0 = __builtin_fmaf(hH0, rowf, hE0), E1 = __builtin_fmaf(hH1, rowf, hE1);       // exact integers
                const float E2 = hA - E0 - E1;
                const uint32_t bx0 = sbox & 0xFF, bx1 = (sbox >> 8) & 0xFF, ry = (sbox >> 16) + (k - sP);   // tile-local
                float lo, hi;
                span_interval(E0, E1, E2, e0, e1, e2, (float)(bx1 - bx0), lo, hi);
                const int len = valid ? hw_cvt_i32(hi - lo) : 0;
                const uint32_t n = len > 0 ? (uint32_t)len : 0u;
                const uint32_t rx0 = bx0 + hw_cvt_u32(lo);
                const unsigned long long Pr = ((unsigned long long)bperm(s, my_key) << 32) | bperm(s, my_sid);
                unsigned long long* top = reinterpret_cast<unsigned long long*>(tilebuf);
                unsigned long long* sec = top + TILE_H * STR64;
                span_trip(top, sec, ry * STR64 + rx0, n, Pr);
                const bool more = n > (uint32_t)B32_TRIP;
                const unsigned long long mm = __ballot(more);
                if (mm) {
                    const uint32_t cnt = (uint32_t)__builtin_popcountll(mm);
                    while (lqn + cnt > 64u) drain_span();
                    const uint32_t entry = s | (ry << 6) | ((rx0 + (uint32_t)B32_TRIP) << 12) | ((n - (uint32_t)B32_TRIP) << 19);
                    const uint32_t dst = more ? lqn + (uint32_t)__builtin_popcountll(mm & ((1ull << lane) - 1ull)) : ((lqn + cnt) & 63u);
                    const uint32_t got = (uint32_t)__builtin_amdgcn_ds_permute((int)(dst << 2), (int)(more ? entry : 0u));
                    if (lane >= lqn && lane < lqn + cnt) lq = got;
                    lqn += cnt;
                }
            }
            while (lqn) drain_span();
        } else
        for (uint32_t k0 = 0; k0 < R; k0 += 64) {
            // owner of item k0+lane: last surface s with h>0 and P[s] <= k
            const unsigned long long before = __ballot(h > 0 && P <= k0);
            const uint32_t carry = before ? 64u - (uint32_t)__builtin_clzll(before) : 0u;       // (index of that surface) + 1
            // every surface that starts inside this round drops (its lane + 1) at the lane of its first item: a forward permute
            // (ds_permute_b32: no memory involved; the starts are distinct and > k0, so lane 0 is never a target and takes the zeros
            // of all the other lanes; lanes nobody writes read 0)
            const bool starts = h > 0 && P > k0 && P < k0 + 64;
            const uint32_t mark = (uint32_t)__builtin_amdgcn_ds_permute((int)((starts ? P - k0 : 0u) << 2), (int)(starts ? lane + 1 : 0u));
            const uint32_t own = max(dpp_max_scan(mark), carry);                                 // >= 1 whenever the item exists
            const uint32_t k = k0 + lane;
            const bool valid = k < R;
            const uint32_t s = valid ? own - 1 : lane;
            const uint32_t sbox = bperm(s, box), sP = bperm(s, P);
            const float sx3 = bpermf(s, __uint_as_float(b.q0.x)), sy3 = bpermf(s, __uint_as_float(b.q0.y));
            const float sa0 = bpermf(s, a0), sb0 = bpermf(s, b0), sa1 = bpermf(s, a1), sb1 = bpermf(s, b1);
            const float sinv = bpermf(s, __uint_as_float(b.q1.z));
            Tri tr;                                                                              // per-lane view (EXACT only)
            if (EXACT) {
                tr.u1 = bpermf(s, __uint_as_float(b.q2.y)); tr.u2 = bpermf(s, __uint_as_float(b.q2.z)); tr.u3 = bpermf(s, __uint_as_float(b.q2.w));
                tr.v1 = bpermf(s, __uint_as_float(b.q3.x)); tr.v2 = bpermf(s, __uint_as_float(b.q3.y)); tr.v3 = bpermf(s, __uint_as_float(b.q3.z));
                tr.flags = bperm(s, flags);
                tr.tw = bperm(s, b.tw); tr.th = bperm(s, b.th); tr.toff = bperm(s, b.toff);
                if (!affine || ZMODE) { tr.iz1 = bpermf(s, __uint_as_float(b.q5.y)); tr.iz2 = bpermf(s, __uint_as_float(b.q5.z)); tr.iz3 = bpermf(s, __uint_as_float(b.q5.w)); }
            }
            uint32_t rx0 = sbox & 0xFF;
            const uint32_t rx1 = (sbox >> 8) & 0xFF, ry = (sbox >> 16) + (k - sP);              // tile-local
            uint32_t n = valid ? rx1 - rx0 : 0u;
            const float dx = (float)(rx0 + x_lo) - sx3, dy = (float)(ry + ty_top) - sy3;
            float w0 = sa0 * dx + sb0 * dy, w1 = sa1 * dx + sb1 * dy;                            // exact integers
            if (B32_ROW_TRIM) {
                const uint32_t lo = row_trim(w0, w1, sa0, sa1, sinv, n);
                rx0 += lo; w0 += sa0 * (float)lo; w1 += sa1 * (float)lo;                         // exact: the closed form at the new start
            }
            uint32_t addr = ry * (P64 ? STR64 : TILE_STRIDE) + rx0;
            const uint32_t li = cs + s + 1;
            uint32_t mine = 0;
            if (EXACT || (ZMODE && !P64)) {
                // (sort-free path with EXACT coverage: the fragment's global priority goes straight to the winners; every stored
                // winner is a drawn fragment, so no runner-up is kept)
                const unsigned long long P = P64 ? (((unsigned long long)bperm(s, my_key) << 32) | bperm(s, my_sid)) : 0ull;
                if (P64 && EXACT && TEXMODE == 0) {
#if B32_DRAIN_TRIPS > 0
                    // one trip now; what is left of the row is queued like the CHEAP flavour's remainders (see `drain`)
                    mine += exact_trip(tr, addr, w0, w1, sa0, sa1, sinv, n, P);
                    const bool more = n > 4u;
                    const unsigned long long mm = __ballot(more);
                    if (mm) {
                        const uint32_t cnt = (uint32_t)__builtin_popcountll(mm);
                        while (lqn + cnt > 64u) drain();
                        const uint32_t entry = s | (ry << 6) | ((rx0 + 4u) << 12) | ((n - 4u) << 19);
                        const uint32_t dst = more ? lqn + (uint32_t)__builtin_popcountll(mm & ((1ull << lane) - 1ull)) : ((lqn + cnt) & 63u);
                        const uint32_t got = (uint32_t)__builtin_amdgcn_ds_permute((int)(dst << 2), (int)(more ? entry : 0u));
                        if (lane >= lqn && lane < lqn + cnt) lq = got;
                        lqn += cnt;
                    }
#else
                    for (uint32_t i = 0; __ballot(i < n); i += 4) mine += exact_trip(tr, addr, w0, w1, sa0, sa1, sinv, i < n ? n - i : 0u, P);
#endif
                } else
                for (uint32_t i = 0; __ballot(i < n); ++i) {
                    if (i < n) {
                        const float bcx = w0 * sinv, bcy = w1 * sinv;
                        const float bcz = 1.0f - bcx - bcy;
                        if (__builtin_fminf(__builtin_fminf(bcx, bcy), bcz) >= ERR) {
                            bool drawn = true;
                            uint32_t zkey = 0;
                            if (ZMODE) drawn = frag_zkey(tr, bcx, bcy, bcz, zkey);
                            if (EXACT && drawn) { uint32_t texel; drawn = texel_drawn<TEXMODE, FMT8>(tr, bcx, bcy, bcz, gtex, ltex, texel, affine); }
                            if (drawn) {
                                if (P64) atomicMax(reinterpret_cast<unsigned long long*>(tilebuf) + addr, ZMODE ? (((unsigned long long)(~zkey) << 32) | (uint32_t)P) : P);
                                else commit_fragment<EXACT, ZMODE>(tilebuf, addr, li, zkey);
                                ++mine;
                            }
                        }
                        ++addr; w0 += sa0; w1 += sa1;
                    }
                }
            } else if (P64) {
                // sort-free CHEAP coverage: the value is the surface's global painter's priority, so no tile list order is needed.
                // Four pixels per trip: four returning LDS atomics in flight, one wait (w advances by the reference's own
                // sequential accumulation w += a).
                const unsigned long long P = ((unsigned long long)bperm(s, my_key) << 32) | bperm(s, my_sid);
                unsigned long long* top = reinterpret_cast<unsigned long long*>(tilebuf);
                unsigned long long* sec = top + TILE_H * STR64;
                float z1 = 0.0f, z2 = 0.0f, z3 = 0.0f;
                if (ZMODE) { z1 = bpermf(s, __uint_as_float(b.q5.y)); z2 = bpermf(s, __uint_as_float(b.q5.z)); z3 = bpermf(s, __uint_as_float(b.q5.w)); }
                // Long rows (large triangles): the certain-interior run behind the first trip -- a multiple of TRIP pixels -- goes to the
                // interior queue, what follows it to the ordinary one.  (Only when the run starts inside the first trip: the first trip then
                // covers the row's left boundary, and one ordinary remainder covers the right one.)
                uint32_t n_int = 0;
                if (B32_INTERIOR && !ZMODE && __ballot(n >= INTERIOR_MIN_ROW)) {
                    uint32_t tlo, thi;
                    const float sgn = sinv < 0.0f ? -1.0f : 1.0f;
                    const bool run = (n >= INTERIOR_MIN_ROW) && interior_run(w0, w1, sa0, sa1, __builtin_fabsf(sa0 * sb1 - sb0 * sa1), sgn, n, tlo, thi);
                    if (run && tlo <= (uint32_t)B32_TRIP && thi >= 2u * (uint32_t)B32_TRIP) n_int = (thi - (uint32_t)B32_TRIP) & ~((uint32_t)B32_TRIP - 1u);
                }
                cheap_trip<ZMODE>(top, sec, addr, w0, w1, sa0, sa1, sinv, n, P, z1, z2, z3);          // (addr, w0, w1 now stand at pixel TRIP of the row)
                if (B32_INTERIOR && !ZMODE) {
                    const unsigned long long mi = __ballot(n_int != 0);
                    if (mi) {
                        const uint32_t cnt = (uint32_t)__builtin_popcountll(mi);
                        while (lqin + cnt > 64u) drain_interior();
                        const uint32_t entry = s | (ry << 6) | ((rx0 + (uint32_t)B32_TRIP) << 12) | (n_int << 19);
                        const uint32_t dst = n_int ? lqin + (uint32_t)__builtin_popcountll(mi & ((1ull << lane) - 1ull)) : ((lqin + cnt) & 63u);
                        const uint32_t got = (uint32_t)__builtin_amdgcn_ds_permute((int)(dst << 2), (int)(n_int ? entry : 0u));
                        if (lane >= lqin && lane < lqin + cnt) lqi = got;
                        lqin += cnt;
                    }
                }
                const uint32_t skip = (uint32_t)B32_TRIP + n_int;          // pixels of the row already dealt with or queued as interior
                const bool more = n > skip;
                const unsigned long long mm = __ballot(more);
                if (mm) {
                    const uint32_t cnt = (uint32_t)__builtin_popcountll(mm);
                    while (lqn + cnt > 64u) drain();
                    // forward permute into the queue's free lanes [lqn, lqn + cnt); the lanes with nothing to push aim at the first lane
                    // behind them (lane 0 when that is 64: then every lane pushes or lqn + cnt == 64 and lane 0 is not taken from `got`)
                    const uint32_t entry = s | (ry << 6) | ((rx0 + skip) << 12) | ((n - skip) << 19);
                    const uint32_t dst = more ? lqn + (uint32_t)__builtin_popcountll(mm & ((1ull << lane) - 1ull)) : ((lqn + cnt) & 63u);
                    const uint32_t got = (uint32_t)__builtin_amdgcn_ds_permute((int)(dst << 2), (int)(more ? entry : 0u));
                    if (lane >= lqn && lane < lqn + cnt) lq = got;
                    lqn += cnt;
                }
            } else {
                // CHEAP coverage: two pixels per trip, so the two returning LDS atomics are in flight together and the wave
                // waits once per pair (the second value is the same sequential accumulation w + a the reference performs)
                for (uint32_t i = 0; __ballot(i < n); i += 2) {
                    const float w0b = w0 + sa0, w1b = w1 + sa1;
                    const float ax = w0 * sinv, ay = w1 * sinv, bx = w0b * sinv, by = w1b * sinv;
                    const float az = 1.0f - ax - ay, bz = 1.0f - bx - by;
                    const bool ina = (i < n) & (__builtin_fminf(__builtin_fminf(ax, ay), az) >= ERR);
                    const bool inb = (i + 1 < n) & (__builtin_fminf(__builtin_fminf(bx, by), bz) >= ERR);
                    uint32_t olda = 0, oldb = 0;
                    if (ina) olda = atomicMax(&tilebuf[addr], li);
                    if (inb) oldb = atomicMax(&tilebuf[addr + 1], li);
                    if (ina) atomicMax(&tilebuf[addr + TILE_H * TILE_STRIDE], min(olda, li));
                    if (inb) atomicMax(&tilebuf[addr + 1 + TILE_H * TILE_STRIDE], min(oldb, li));
                    addr += 2; w0 = w0b + sa0; w1 = w1b + sa1;
                }
            }
            if (EXACT) {
                for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
                frags += (uint32_t)__builtin_amdgcn_readfirstlane((int)mine);
            }
        }
        if (P64) while (lqn) drain();                  // (the row remainders of this batch: its registers are about to be reloaded)
        if (P64 && B32_INTERIOR && !ZMODE && !EXACT) while (lqin) drain_interior();
        // surfaces whose edge walk must be replayed literally: wave-cooperative slow path
        unsigned long long sm = __ballot(slow);
        while (sm) {
            const int t = __builtin_ctzll(sm);
            sm &= sm - 1;
            if (P64) {
                const unsigned long long P = ((unsigned long long)bcu(my_key, t) << 32) | bcu(my_sid, t);
                const uint32_t cnt64 = cover_slow64<ZMODE, EXACT, FMT8>(tri_from_batch(b, t, ZMODE || EXACT), P, tilebuf, x_lo, x_hi, y_lo, y_hi, ty_top, lane, gtex, affine);
                if (EXACT) frags += cnt64;
                continue;
            }
            frags += cover_one<TEXMODE, EXACT, ZMODE, FMT8>(b, t, cs + (uint32_t)t + 1, tilebuf, x_lo, x_hi, y_lo, y_hi, ty_top, lane, gtex, ltex, affine);
        }
    }
    (void)wave; (void)wmark;        // (the row starts travel by ds_permute now; the per-wave mark area holds the shading phase's repair queues)
    return frags;
}


// ------------------------------------------------------------------------------------------------ tile-local depth sort
// Stable LSD radix sort (4 x 8 bits) of one tile list (n <= LOCAL_SORT_CAP surface ids, keyed by k_setup's 32-bit painter's
// key) entirely in LDS, by the NT threads of the workgroup; the sorted ids go back to the list in global memory.  This is the
// reference's `sort_by` (render.rs:2527-2541) applied per tile: lists arrive in face order and every pass is stable, so equal
// keys keep face order exactly like the global sort.  The four LDS arrays alias the (not yet used) tile buffers.
template <int NT>
__device__ void tile_local_sort(uint32_t* sort_area, uint32_t* wcnt, uint32_t* dws, const uint32_t* __restrict__ keys,
                                uint32_t* list, uint32_t n, uint32_t* n_opaque_out) {
    constexpr int NW = NT / 64;
    constexpr int STEPS = LOCAL_SORT_CAP / (NW * 64);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t *ki = sort_area, *vi = sort_area + LOCAL_SORT_CAP, *ko = sort_area + 2 * LOCAL_SORT_CAP, *vo = sort_area + 3 * LOCAL_SORT_CAP;
    uint32_t my_opaque = 0;
    for (uint32_t i = tid; i < n; i += NT) { const uint32_t sid = list[i]; const uint32_t k = keys[sid]; ki[i] = k; vi[i] = sid; my_opaque += (k >> 31) ^ 1u; }
    for (int off = 32; off > 0; off >>= 1) my_opaque += __shfl_down(my_opaque, off);
    if (lane == 0 && my_opaque) atomicAdd(n_opaque_out, my_opaque);       // class boundary of the sorted list
    __syncthreads();
    const uint32_t per_wave = ((n + NW * 64 - 1) / (NW * 64)) * 64;      // contiguous run per wave: order = (wave, step, lane)
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = pass * 8;
        for (uint32_t d = tid; d < NW * 256; d += NT) wcnt[d] = 0;
        __syncthreads();
        uint32_t key[STEPS], val[STEPS], rnk[STEPS];
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            const uint32_t idx = wave * per_wave + st * 64 + lane;
            const bool live = (uint32_t)(st * 64) < per_wave && idx < n;
            const uint32_t k = live ? ki[idx] : 0u;
            key[st] = k; val[st] = live ? vi[idx] : 0u;
            const uint32_t d = (k >> shift) & 255u;
            unsigned long long peers = __ballot(live);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const unsigned long long m = __ballot((d >> b) & 1u);
                peers &= ((d >> b) & 1u) ? m : ~m;
            }
            uint32_t before = 0;
            if (live) {
                const uint32_t leader = (uint32_t)__builtin_ctzll(peers);
                uint32_t old = 0;
                if (lane == leader) { old = wcnt[wave * 256 + d]; wcnt[wave * 256 + d] = old + (uint32_t)__popcll(peers); }
                old = __shfl(old, (int)leader);
                before = old + (uint32_t)__popcll(peers & lt_mask);
            }
            rnk[st] = live ? before : 0xFFFFFFFFu;
        }
        __syncthreads();
        if (tid < 256) {      // digit tid: total over waves, exclusive scan over digits, then per-wave bases
            uint32_t tot = 0;
            for (int w = 0; w < NW; ++w) tot += wcnt[w * 256 + tid];
            uint32_t inc = tot;
            for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(inc, off); if (lane >= (uint32_t)off) inc += t; }
            if (lane == 63) dws[wave] = inc;
            wcnt[NW * 256 + tid] = inc - tot;     // in-wave exclusive prefix; the cross-wave part follows the barrier
        }
        __syncthreads();
        if (tid < 256) {
            uint32_t run = wcnt[NW * 256 + tid];
            for (uint32_t w = 0; w < wave; ++w) run += dws[w];
            for (int w = 0; w < NW; ++w) { const uint32_t c = wcnt[w * 256 + tid]; wcnt[w * 256 + tid] = run; run += c; }
        }
        __syncthreads();
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            if (rnk[st] != 0xFFFFFFFFu) {
                const uint32_t pos = wcnt[wave * 256 + ((key[st] >> shift) & 255u)] + rnk[st];
                ko[pos] = key[st]; vo[pos] = val[st];
            }
        }
        __syncthreads();
        uint32_t* t = ki; ki = ko; ko = t; t = vi; vi = vo; vo = t;
    }
    for (uint32_t i = tid; i < n; i += NT) list[i] = vi[i];
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------ k_cover
template <bool FMT8, int NT, bool ZMODE>
__device__ __forceinline__ void shade_tile_p64(const FillArgs& a, const uint32_t* tilebuf, uint32_t e0, uint32_t e1, uint32_t x_lo, uint32_t x_hi,
                                               uint32_t y_lo, uint32_t y_hi, uint32_t ty_top, uint32_t tid, uint32_t lane, uint32_t TH, uint32_t* rq,
                                               const uint8_t* latlas);

template <int NT, bool ZMODE>
__device__ __forceinline__ void shade_tile_plain(const FillArgs& a, const uint32_t* tilebuf, uint32_t e0, uint32_t e1, uint32_t x_lo, uint32_t x_hi,
                                                 uint32_t y_lo, uint32_t y_hi, uint32_t ty_top, uint32_t tid, uint32_t lane, uint32_t TH, uint32_t* wq);

// k_setup's per-block counters (visible, transparent, NaN keys per class, bad vertex index) -> the frame's abort decision in misc[6]
// (the reference panics before drawing on a bad vertex index, render.rs:2375, or when a sort comparison sees NaN, render.rs:2531);
// workgroup 0 publishes the sums in Ctrl for the host.  Per-thread sums, a wave reduction, then one LDS atomic per wave and counter.
template <int NT>
__device__ __forceinline__ void reduce_setup_counters(const FillArgs& a, uint32_t* misc, uint32_t tid, uint32_t lane) {
    if (tid < 5) misc[8 + tid] = 0;
    __syncthreads();
    const uint32_t npart = (a.fp.nf + 255) / 256;
    uint32_t acc[5] = { 0, 0, 0, 0, 0 };
    for (uint32_t b = tid; b < npart; b += NT)
#pragma unroll
        for (int k = 0; k < 5; ++k) acc[k] += a.partials[b * 8 + k];
#pragma unroll
    for (int k = 0; k < 5; ++k) {           // (DPP prefix sums: thirty dependent ds_bpermute round trips sat on every small frame's critical path)
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)dpp_add_scan(acc[k]), 63);
        if (lane == 0 && tot) atomicAdd(&misc[8 + k], tot);
    }
    __syncthreads();
    if (tid == 0) {
        const uint32_t t[5] = { misc[8], misc[9], misc[10], misc[11], misc[12] };
        const uint32_t n_opq = t[0] - t[1];
        const bool ab = t[4] != 0 || (t[2] && n_opq >= 2) || (t[3] && t[1] >= 2);
        misc[6] = ab ? 1u : 0u;
        if (blockIdx.x == 0) {
            a.ctrl->n_visible = t[0]; a.ctrl->n_transparent = t[1]; a.ctrl->nan_opaque = t[2]; a.ctrl->nan_transparent = t[3];
            a.ctrl->err_index = t[4] ? 1u : 0u; a.ctrl->n_opaque = n_opq;
            if (ab) { a.ctrl->abort = 1; a.ctrl->sticky |= t[4] ? 1u : 2u; }
        }
    }
    __syncthreads();
}

// Framebuffer::clear folded into the frame (FillArgs::clear_on): a frame that draws nothing (abort, redraw by the host) still owes the
// caller the clear it took over from b32_fb_clear -- all the workgroups fill the band together.
template <int NT>
__device__ __forceinline__ void clear_band(const FillArgs& a) {
    const FrameParams& fp = a.fp;
    uint32_t* row0 = a.fb + (size_t)fp.band_y0 * fp.width;
    const size_t n = (size_t)fp.width * (fp.band_y1 - fp.band_y0);
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) row0[i] = a.clear_rgba;
    if (a.clear_depth) {        // (z-buffer mode: Framebuffer::clear resets the depth buffer too, render.rs:43)
        float* z0 = a.zbuf + (size_t)fp.band_y0 * fp.width;
        for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) z0[i] = __uint_as_float(0x7F7FFFFFu);
    }
}

// PLAIN == 2: the plain form compiled for five waves per SIMD, i.e. 96 VGPRs (8 dwords of scratch) instead of 109: alone it is 1-2 us
// slower, but four of its waves leave a SIMD 128 registers -- TWO waves of the next frame's setup kernel instead of one -- and the
// pipelined frame gains ~2 % (C3 0.1219-0.1222 -> 0.1193-0.1211 ms); chosen only for frames whose setup kernel runs on the side stream.
// PLAIN: the configuration BASELINE.json's metric is quoted on, with its run-time switches turned into constants -- affine UVs, no
// shading pass, fixed-point snapping, perspective camera, one texture, lists from the binning launch, no transparent pass.  The
// compiler then drops the other branches of coverage and shading from this instantiation (102 -> 94 VGPRs, 45 -> 13 spilled SGPRs).
template <int TEXMODE, bool EXACT, int NT, bool ZMODE, bool FMT8 = false, bool P64 = false, int PLAIN = 0>
// (every form but the plain one is compiled for at least 4 waves per SIMD, i.e. at most 128 VGPRs: the EXACT z-buffer forms had drifted
// to 129, which halves the 512-thread kernel's residency to one workgroup per CU -- game() settings with colour-keyed textures at
// 2560x1920: 0.289 -> 0.242 ms; the plain form keeps the default bound of its block size, its code is byte-identical)
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(PLAIN == 2 ? 5 : (PLAIN == 1 ? 4 : ((P64 && NT == 512) ? B32_P64_WAVES : 4))))) void k_cover(FillArgs a_in) {
    FillArgs a_plain = a_in;
    if (PLAIN) {
        a_plain.fp.affine = 1; a_plain.fp.shading = B32_SHADE_NONE; a_plain.fp.fixed_point = 1; a_plain.fp.ortho = 0; a_plain.fp.nt = 1;
        a_plain.fp.n_lights = 0; a_plain.inline_bin = 0; a_plain.gather_blend = 0; a_plain.shades = nullptr;
    }
    const FillArgs& a = a_plain;
    constexpr int NW = NT / 64;
    constexpr int TB = (P64 ? 4 : 2) * LDS_TILE_BYTES;          // tile buffers: top + runner-up, 32- or 64-bit entries
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* tilebuf = reinterpret_cast<uint32_t*>(smem);
    // (plain LDS pointers: every cross-wave value below is read after a __syncthreads() that follows its write; `volatile` would turn
    // these into FLAT accesses with a full vmcnt wait each)
    uint32_t* misc = reinterpret_cast<uint32_t*>(smem + TB);   // [0] tile, [2] list cursor
    uint32_t* wmarks = reinterpret_cast<uint32_t*>(smem + TB + LDS_MISC_BYTES);
    // sort-free forms: 64 words of repair queue per wave, then (optionally) the staged index atlas
    const uint8_t* latlas = nullptr;
    if (P64 && !FMT8 && a.atlas_idx_bytes) {
        uint4* dst = reinterpret_cast<uint4*>(smem + TB + LDS_MISC_BYTES + NW * 256);
        const uint4* src = reinterpret_cast<const uint4*>(a.atlas0);
        const uint32_t nq = (ATLAS_CLUT_BYTES + a.atlas_idx_bytes + 15u) / 16u;
        for (uint32_t i = threadIdx.x; i < nq; i += NT) dst[i] = src[i];
        latlas = reinterpret_cast<const uint8_t*>(dst);           // (first read behind the tile loop's barriers)
    }
    const uint16_t* ltex = reinterpret_cast<const uint16_t*>(smem + LDS_TEX_OFFSET);    // LDS texture (TEXMODE 1) or LDS skip mask (P64 EXACT)
    uint32_t* sort_cnt = reinterpret_cast<uint32_t*>(smem + LDS_TEX_OFFSET);        // local sort only exists without an LDS texture

    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));   // wave-uniform => SGPR control flow
    const FrameParams& fp = a.fp;
    const uint32_t ntiles = fp.tiles_x * fp.tiles_y;
    phase_stamp(a.ctrl, ST_FILL);
    // (sort-free forms: workgroup 0 also notes the shader-cycle counter now and, with the wall clock, when it runs out of tiles -- the
    // shader clock the fill really ran at, b32_last_shader_clock: under this kernel's load it sits below the device's nominal clock)
    const unsigned long long clk_entry = (P64 && blockIdx.x == 0 && tid == 0) ? (unsigned long long)clock64() : 0ull;
    // small meshes (inline_bin): the first NT spans (and class bits) are requested before the counters are reduced, and the skip mask is
    // staged before it too -- a C1 workgroup used to start its tile 7 us into an 18-us kernel behind three dependent round trips
    uint32_t pre_span = 0xFFFFFFFFu, pre_key = 0u;
    if (P64 && a.inline_bin && tid < fp.nf) { pre_span = a.spans[tid]; if (a.gather_blend) pre_key = a.keys[tid]; }
    if (P64 && EXACT && a.mask_lds_words) {     // the pool's skip mask into the (unused) runner-up plane, once per workgroup
        uint32_t* ml = tilebuf + 2 * TILE_H * TILE_STRIDE;
        for (uint32_t i = tid; i < a.mask_lds_words; i += NT) ml[i] = a.texmask[i];
        ltex = reinterpret_cast<const uint16_t*>(ml);
    }
    bool reduce_late = false;
    if (P64 && (a.inline_bin || a.direct_bin)) {
        // there was no binning launch, so nobody has reduced k_setup's per-block counters yet.  Small mesh (inline_bin): every workgroup
        // derives the frame's abort decision from them (the reference panics before drawing on a bad vertex index, render.rs:2375, or
        // when a sort comparison sees NaN, render.rs:2531); workgroup 0 publishes the counters in Ctrl for the host.  Direct binning:
        // k_setup left the epoch of this frame in Events when it met one of those, and only then does every workgroup pay for the
        // reduction; otherwise workgroup 0 alone does it, for the host's counters.
        bool reduce = a.inline_bin != 0;
        uint32_t redraw = 0;
        if (a.direct_bin) {
            const Events* ev = events_of(a.ctrl);
            reduce = ev->bad_index == a.epoch || ev->nan_opaque == a.epoch || ev->nan_transparent == a.epoch;
            redraw = (ev->overflow == a.epoch ? 2u : 0u) | (ev->long_transparent == a.epoch ? 1u : 0u);
            // (no such event: the frame is not aborted, and workgroup 0 reduces the counters for the host AFTER its tiles -- with a
            // million faces the reduction takes microseconds, and in a narrow band every workgroup has one tile: it was the kernel's
            // critical path)
            reduce_late = !reduce && blockIdx.x == 0;
        }
        if (tid == 0) misc[6] = 0;
        __syncthreads();
        if (reduce) reduce_setup_counters<NT>(a, misc, tid, lane);
        if (misc[6] || redraw) {
            // nothing is drawn.  Direct binning: a region overflowed (the host redraws with larger regions: the longest list goes
            // back in Ctrl) or a transparent list is too long for k_blend's LDS sort (the host redraws with the global sort); either
            // way the fill counters are left zero for the next frame, as k_setup expects them.
            if (a.direct_bin) {
                for (uint32_t t = blockIdx.x * NT + tid; t < ntiles; t += gridDim.x * NT) {
                    uint32_t* fl = a.tile_fill + (size_t)t * FILL_PAD;
                    if (redraw & 2u) atomicMax(&a.ctrl->list_demand, fl[0]);
                    fl[0] = 0; fl[1] = 0;
                }
                if (blockIdx.x == 0 && tid == 0 && !misc[6]) a.ctrl->need_global_sort = redraw;
            }
            if (a.clear_on) clear_band<NT>(a);
            return;
        }
    } else {
        if (a.ctrl->abort || (P64 && a.ctrl->need_global_sort)) {   // (a transparent tile list too long for k_blend's LDS sort: the host redraws)
            if (P64 && a.clear_on) clear_band<NT>(a);
            return;
        }
    }

    TexDesc lds_desc = { 0, 0, 0, 0 };
    if (TEXMODE == 1) {     // stage texture 0 once per workgroup: 16-B coalesced loads -> LDS
        lds_desc = a.tex[0];
        const uint4* src = reinterpret_cast<const uint4*>(a.texels + lds_desc.offset);
        uint4* dst = reinterpret_cast<uint4*>(smem + LDS_TEX_OFFSET);
        const uint32_t nq = (a.lds_tex_texels + 7) / 8;
        for (uint32_t i = tid; i < nq; i += NT) dst[i] = src[i];
    }
    unsigned long long frag_count = 0;
    // Staggered start (FillArgs::stagger, 10-ns ticks; frames with more tiles than workgroup slots): the second workgroup of every CU
    // begins a few microseconds late.  Started together, the two workgroups of a CU run their first tiles in step -- both in the
    // LDS-latency bound coverage, then both in the memory bound shading -- and every CU of the chip does the same at the same time: the
    // first tile of a workgroup took 36 us against 25-29 us for the later ones (tools/timeline.py).  Any delay between 3 and 8 us gives
    // the same gain (C3 0.1285 -> 0.1213 ms per frame, C5 0.198 -> 0.192, profiles/r05_stagger.txt).
    if (P64 && a.stagger && blockIdx.x >= gridDim.x / 2) {
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (unsigned long long)a.stagger) __builtin_amdgcn_s_sleep(8);
    }
    // the first tile of a workgroup is its own index (no atomic: 512 same-address atomics serialise at ~12 ns each), later ones come
    // from the shared cursor
    uint32_t next_tile = blockIdx.x;
    for (;;) {
        if (tid == 0) { misc[0] = next_tile; misc[2] = 0; misc[4] = 0; misc[5] = 0; }
        __syncthreads();
        const uint32_t tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)misc[0]);
        if (tile >= ntiles) break;
        // The next tile is taken from the shared cursor as LATE as its latency can still hide: after this tile's coverage, before its
        // shading (P64), so that the last tiles of the queue go to the workgroups that really are about to be free (taking it at the top
        // of the tile, one whole tile ahead, made the last round of the queue a static assignment: tools/timeline.py).
        const bool fetch_late = P64;
        if (!fetch_late && tid == 0) next_tile = gridDim.x + atomicAdd(&a.ctrl->tile_cursor, 1u);
        uint32_t e0, e1;
        if (P64 && a.inline_bin) {               // the list is collected below, into this tile's own region
            e0 = e1 = tile * a.list_stride;
        } else if (P64 && a.direct_bin) {        // k_setup built the lists: opaque pass at the front of the region, transparent at its back
            const uint32_t* fl = a.tile_fill + (size_t)tile * FILL_PAD;
            const uint32_t n_o = fl[0], n_t = fl[1];
            e0 = tile * a.list_stride; e1 = e0 + n_o;
            if (tid == 0) {
                if (a.gather_blend) a.tile_mid[tile] = e0 + a.list_stride - n_t;
                if (n_o | n_t) atomicAdd(&a.ctrl->n_pairs, n_o + n_t);
            }
        } else if (P64) {                        // lists in any order, keyed by tile only; [e0, mid) is the opaque pass
            e0 = a.ranges[tile]; e1 = a.gather_blend ? a.tile_mid[tile] : a.ranges[tile + 1];
        } else if (TEXMODE == 0 && a.local_sort) {      // lists arrive in face order, keyed by tile only: painter's order per tile, in LDS
            e0 = a.ranges[tile];
            const uint32_t e2 = a.ranges[tile + 1];
            if (e2 - e0 > LOCAL_SORT_CAP) {
                if (tid == 0) atomicOr(&a.ctrl->need_global_sort, 1u);               // host redraws with the global depth sort
                __syncthreads();
                continue;
            }
            // one stable sort of the whole list: the class bit is the key's top bit, so the transparent pass ends up behind
            // the opaque one, each in painter's order (render.rs:2522-2541)
            if (e2 > e0) tile_local_sort<NT>(tilebuf, sort_cnt, misc + 8, a.keys, a.pair_vals + e0, e2 - e0, &misc[4]);
            e1 = e0 + misc[4];
            if (tid == 0) a.tile_mid[tile] = e1;
        } else {
            e0 = a.ranges[2 * tile]; e1 = a.ranges[2 * tile + 1];
        }
        const uint32_t txi = tile % fp.tiles_x;
        const uint32_t x_lo = txi * TILE_W, x_hi = min(x_lo + TILE_W, fp.width);
        // the sort-free path may run on cut tiles (32 or 16 rows: too few 64x64 tiles to fill the GPU); the LDS planes keep their
        // full-tile layout, only the rows in use change
        uint32_t TH, ty_top;
        tile_row_geom(fp, tile / fp.tiles_x, ty_top, TH);
        const uint32_t y_lo = max(ty_top, fp.band_y0), y_hi = min(ty_top + TH, fp.band_y1);
        if (P64 && a.inline_bin) {
            // the faces whose span reaches this tile, in any order (ballot compaction; misc[4..5] were zeroed with the tile index):
            // the opaque pass grows from the front of the tile's region, the transparent pass (class = bit 31 of the depth key) from
            // its back, so k_blend finds its entries in [tile_mid, region end)
            const uint32_t tyl = tile / fp.tiles_x;
            const uint32_t r_end = e0 + a.list_stride;
            for (uint32_t f0 = 0; f0 < fp.nf; f0 += NT) {
                const uint32_t f = f0 + tid;
                bool hit = false, tr = false;
                if (f < fp.nf) {
                    const uint32_t span = f0 == 0 ? pre_span : a.spans[f];
                    hit = span != 0xFFFFFFFFu && txi >= (span & 0xFF) && txi <= ((span >> 8) & 0xFF) && tyl >= ((span >> 16) & 0xFF) && tyl <= (span >> 24);
                    if (hit && a.gather_blend) tr = ((f0 == 0 ? pre_key : a.keys[f]) >> 31) != 0;
                }
                const unsigned long long mo = __ballot(hit && !tr), mt = __ballot(hit && tr);
                uint32_t bo = 0, bt = 0;
                if (lane == 0 && mo) bo = atomicAdd((&misc[4]), (uint32_t)__builtin_popcountll(mo));
                if (lane == 0 && mt) bt = atomicAdd((&misc[5]), (uint32_t)__builtin_popcountll(mt));
                bo = (uint32_t)__builtin_amdgcn_readfirstlane((int)bo); bt = (uint32_t)__builtin_amdgcn_readfirstlane((int)bt);
                const unsigned long long below = (1ull << lane) - 1ull;
                if (hit && !tr) a.pair_vals[e0 + bo + (uint32_t)__builtin_popcountll(mo & below)] = f;
                if (hit && tr) a.pair_vals[r_end - 1u - (bt + (uint32_t)__builtin_popcountll(mt & below))] = f;
            }
        }
        if (P64 && ZMODE) { // winners seeded with the current z-buffer: a fragment wins only with a strictly smaller depth (low word all ones)
            unsigned long long* t64 = reinterpret_cast<unsigned long long*>(tilebuf);
            for (uint32_t p = tid; p < TILE_W * TH; p += NT) {
                const uint32_t row = p >> 6, col = p & 63;
                const uint32_t px = x_lo + col, py = ty_top + row;
                const bool inb = px < x_hi && py >= y_lo && py < y_hi;
                // (a folded Framebuffer::clear: every depth is f32::MAX, nothing is read)
                const float zseed = a.clear_depth ? __uint_as_float(0x7F7FFFFFu) : (inb ? a.zbuf[(size_t)py * fp.width + px] : 0.0f);
                t64[row * STR64 + col] = inb ? (((unsigned long long)(~zsort_key(zseed)) << 32) | 0xFFFFFFFFull) : ~0ull;
                if (!EXACT) t64[TILE_H * STR64 + row * STR64 + col] = 0ull;          // (EXACT keeps no runner-up: that plane holds the skip mask)
            }
        } else if (ZMODE) { // 64-bit entries (depth key << 32 | list position), seeded with the current z-buffer: a fragment wins
            unsigned long long* t64 = reinterpret_cast<unsigned long long*>(tilebuf);       // only with a strictly smaller depth
            for (uint32_t p = tid; p < TILE_W * TILE_H; p += NT) {
                const uint32_t row = p >> 6, col = p & 63;
                const uint32_t px = x_lo + col, py = ty_top + row;
                const bool inb = px < x_hi && py >= y_lo && py < y_hi;
                t64[row * TILE_STRIDE + col] = inb ? ((unsigned long long)zsort_key(a.zbuf[(size_t)py * fp.width + px]) << 32) : 0ull;
            }
        } else {
            if (P64 && TH < (uint32_t)TILE_H) {             // half-height tile: clear only the rows in use of both 64-bit planes
                unsigned long long* t64 = reinterpret_cast<unsigned long long*>(tilebuf);
                for (uint32_t i = tid; i < TH * STR64; i += NT) { t64[i] = 0ull; if (!EXACT) t64[TILE_H * STR64 + i] = 0ull; }
            } else
            if (P64 && !EXACT) {                            // both 64-bit planes, 16 bytes per store 
                uint4* t128 = reinterpret_cast<uint4*>(tilebuf);
                for (uint32_t i = tid; i < (uint32_t)(TILE_H * STR64); i += NT) t128[i] = make_uint4(0, 0, 0, 0);
            } else
            for (uint32_t i = tid; i < (P64 ? 2 : (EXACT ? 1 : 2)) * TILE_H * TILE_STRIDE; i += NT) tilebuf[i] = 0;
        }
        __syncthreads();
        if (P64 && a.direct_bin && tid == 0) {   // (everyone has read them) zero again for the next frame's k_setup
            uint32_t* fl = a.tile_fill + (size_t)tile * FILL_PAD;
            fl[0] = 0; fl[1] = 0;
        }
        if (P64 && a.inline_bin) {
            e1 = e0 + misc[4];
            const uint32_t n_tr = misc[5];
            if (tid == 0) {
                if (a.gather_blend) a.tile_mid[tile] = e0 + a.list_stride - n_tr;
                if (e1 != e0 || n_tr) atomicAdd(&a.ctrl->n_pairs, e1 - e0 + n_tr);
            }
        }
        const uint32_t n_op = e1 - e0;
#ifdef B32_TIMELINE
        const unsigned long long tl0 = wall_clock64();
#endif
        if (n_op) {
            frag_count += phase_a_rows<TEXMODE, EXACT, NW, ZMODE, FMT8, P64>(a, e0, n_op, lane, wave, &misc[2], wmarks + wave * 64, lds_desc, tilebuf,
                                                           x_lo, x_hi, y_lo, y_hi, ty_top, ltex);
            __syncthreads();
        }
        if (P64) {          // shade the tile straight from the LDS winners (no visibility buffer)
            if (tid == 0) next_tile = gridDim.x + atomicAdd(&a.ctrl->tile_cursor, 1u);
#ifdef B32_TIMELINE
            const unsigned long long tl1 = wall_clock64();
#endif
            if (n_op) {
                // (the plain form's straight-line shading: one texture of non-zero size fetched from global memory)
                // (the straight-line shading: RGB555, affine UVs, fixed-point snap, perspective camera, one texture of non-zero size fetched
                // from global memory -- painter's or z-buffer mode, with or without a shading pass; wave-uniform choice)
                if (!FMT8 && fp.affine && fp.fixed_point && !fp.ortho && fp.nt == 1 && !latlas && a.tex0.width && a.tex0.height &&
                    (fp.shading == B32_SHADE_NONE || a.shades))
                    shade_tile_plain<NT, ZMODE>(a, tilebuf, e0, e1, x_lo, x_hi, y_lo, y_hi, ty_top, tid, lane, TH, wmarks + wave * 64);
                else shade_tile_p64<FMT8, NT, ZMODE>(a, tilebuf, e0, e1, x_lo, x_hi, y_lo, y_hi, ty_top, tid, lane, TH, wmarks + wave * 64, latlas);
            }
            else if (a.clear_on) {      // nothing reaches this tile: it still gets the frame's clear colour
                for (uint32_t p = tid; p < TILE_W * TH; p += NT) {
                    const uint32_t px = x_lo + (p & 63), py = ty_top + (p >> 6);
                    if (px < x_hi && py >= y_lo && py < y_hi) { a.fb[(size_t)py * fp.width + px] = a.clear_rgba; if (a.clear_depth) a.zbuf[(size_t)py * fp.width + px] = __uint_as_float(0x7F7FFFFFu); }
                }
            }
            __syncthreads();
#ifdef B32_TIMELINE
            if (tid == 0 && a.dbg) {
                const unsigned long long k = atomicAdd(a.dbg, 1ull);
                if (k < 8192) { unsigned long long* e = a.dbg + 1 + 4 * k; e[0] = ((unsigned long long)blockIdx.x << 32) | tile | ((unsigned long long)n_op << 48); e[1] = tl0; e[2] = tl1; e[3] = wall_clock64(); }
            }
#endif
            continue;
        }
        // winners -> visibility buffer: one 256-B row segment per wave instruction (zeros for uncovered pixels)
        for (uint32_t p = tid; p < TILE_W * TILE_H; p += NT) {
            const uint32_t row = p >> 6, col = p & 63;
            const uint32_t px = x_lo + col, py = ty_top + row;
            if (px < x_hi && py >= y_lo && py < y_hi) {
                uint32_t li;
                if (ZMODE) {
                    const unsigned long long e = reinterpret_cast<const unsigned long long*>(tilebuf)[row * TILE_STRIDE + col];
                    li = (uint32_t)e;
                    if (li) {                                                       // fb.zbuffer[idx] = z, render.rs:1686-1688
                        float z = zsort_val((uint32_t)(e >> 32));
                        if (z == 0.0f) z = exact_depth_at(a, a.pair_vals[e0 + li - 1], px, py);
                        a.zbuf[(size_t)py * fp.width + px] = z;
                    }
                } else li = tilebuf[row * TILE_STRIDE + col];
                // CHEAP coverage: the runner-up travels in the high half when the tile list is short enough (< 32768 entries);
                // bit 31 marks a long list whose runner-up is unknown.
                uint32_t packed = li;
                if (!EXACT) {
                    const uint32_t second = tilebuf[TILE_H * TILE_STRIDE + row * TILE_STRIDE + col];
                    packed = n_op < 0x8000u ? (li | (second << 16)) : (li | 0x80000000u);     // bit 31 = long list, no runner-up
                }
                a.vis[(size_t)py * fp.width + px] = packed;
            }
        }
        __syncthreads();   // everyone is done with misc / tilebuf before the next tile
    }
    if (P64 && blockIdx.x == 0 && tid == 0) {
        unsigned long long* st = reinterpret_cast<Stamps*>(a.ctrl + 1)->t;
        st[ST_CLK0] = clk_entry; st[ST_CLK1] = (unsigned long long)clock64(); st[ST_CLKW] = wall_clock64();
    }
    if (P64 && reduce_late) { __syncthreads(); reduce_setup_counters<NT>(a, misc, tid, lane); }
    if (EXACT && !ZMODE) { // fragment-store count (wave-uniform per wave): one same-address atomic per workgroup
                           // (not defined in z-buffer mode: which fragments pass `z < zbuffer` depends on the sequential order)
        unsigned long long* wf = reinterpret_cast<unsigned long long*>(smem);
        __syncthreads();
        if (lane == 0) wf[wave] = frag_count;
        __syncthreads();
        if (tid == 0) {
            unsigned long long t = 0;
            for (int w = 0; w < NW; ++w) t += wf[w];
            if (t) atomicAdd(&a.ctrl->fragments, t);
        }
    }
}

// ------------------------------------------------------------------------------------------------ k_shade
// Coverage test of list entry li at pixel (px,py): inside test + texel + transparency rule. Keeps what colouring needs.
struct Hit { float bcx, bcy, bcz; uint32_t texel, vc1, vc2, vc3, flags, sid; };
template <bool FMT8>
__device__ __forceinline__ bool hit_test(const FillArgs& a, uint32_t sid, uint32_t px, uint32_t py, Hit& h) {
    const bool affine = a.fp.affine != 0;
    RecView rv;
    load_shade_view(a, sid, !affine, rv);
    const uint4 q0 = rv.q0, q1 = rv.q1, q2 = rv.q2, q3 = rv.q3, q4 = rv.q4, q5 = rv.q5;
    Tri tr;
    tr.x3 = __uint_as_float(q0.x); tr.y3 = __uint_as_float(q0.y); tr.a0 = __uint_as_float(q0.z); tr.b0 = __uint_as_float(q0.w);
    tr.a1 = __uint_as_float(q1.x); tr.b1 = __uint_as_float(q1.y); tr.inv_area = __uint_as_float(q1.z);
    tr.min_x = q1.w & 0xFFFF; tr.max_x = q1.w >> 16; tr.min_y = q2.x & 0xFFFF; tr.max_y = q2.x >> 16;
    tr.u1 = __uint_as_float(q2.y); tr.u2 = __uint_as_float(q2.z); tr.u3 = __uint_as_float(q2.w);
    tr.v1 = __uint_as_float(q3.x); tr.v2 = __uint_as_float(q3.y); tr.v3 = __uint_as_float(q3.z);
    tr.flags = q3.w;
    tr.w0_start = __uint_as_float(q4.w); tr.w1_start = __uint_as_float(q5.x);
    tr.iz1 = __uint_as_float(q5.y); tr.iz2 = __uint_as_float(q5.z); tr.iz3 = __uint_as_float(q5.w);
    tr.tw = tr.th = tr.toff = 0;
    const uint32_t txid = tr.flags & F_TEX_MASK;
    if (txid != F_TEX_NONE) {
        if (a.fp.nt == 1) { tr.tw = a.tex0.width; tr.th = a.tex0.height; tr.toff = a.tex0.offset; }   // uniform: no descriptor gather
        else { const TexDesc d = a.tex[txid]; tr.tw = d.width; tr.th = d.height; tr.toff = d.offset; }
    }
    float w0, w1;
    edge_w(tr, px, py, w0, w1);
    if (!inside_bc(tr, w0, w1, h.bcx, h.bcy, h.bcz)) return false;
    h.texel = 0;
    if (!texel_drawn<0, FMT8>(tr, h.bcx, h.bcy, h.bcz, FMT8 ? reinterpret_cast<const uint16_t*>(a.texels32) : a.texels, nullptr, h.texel, affine)) return false;
    h.vc1 = q4.x; h.vc2 = q4.y; h.vc3 = q4.z; h.flags = tr.flags; h.sid = sid;
    return true;
}
template <bool FMT8>
__device__ __forceinline__ uint32_t colour(const FillArgs& a, const Hit& h, int shading, uint32_t px, uint32_t py) {
    float shv[9];
    if (shading != B32_SHADE_NONE) for (int j = 0; j < 9; ++j) shv[j] = a.shades[(size_t)h.sid * 9 + j];
    // 8-bit path: the overwrite pass only runs when no texel blends and every editor alpha is 255 -> set_pixel (render.rs:301-310)
    if (FMT8) return (shade8(h.texel, h.bcx, h.bcy, h.bcz, h.vc1, h.vc2, h.vc3, h.flags, shading, shv, px, py) & 0xFFFFFFu) | 0xFF000000u;
    return shade15<true>(h.texel, h.bcx, h.bcy, h.bcz, h.vc1, h.vc2, h.vc3, h.flags, shading, shv, px, py);   // set_pixel_15 of the Color15 (see shade15)
}

// ------------------------------------------------------------------------------------------------ fused shading (P64 fast path)
// After the sort-free coverage of a tile the workgroup shades the tile straight from the LDS winners: no visibility buffer
// round trip through HBM, and while one workgroup of a CU sits in the (memory-latency bound) shading phase the other one
// runs its (LDS/VALU bound) coverage phase.  Each lane shades TWO pixels at a time: both record gathers are issued before
// either is used, then both texel fetches, so two dependent load chains are in flight per lane.
using RecRegs = RecView;
__device__ __forceinline__ void rec_load(const FillArgs& a, uint32_t sid, bool need5, RecRegs& r) { load_shade_view(a, sid, need5, r); }
// inside test + texel address (index into the texel pool; -1 = untextured -> white, -2 = zero-size texture -> transparent)
__device__ __forceinline__ bool hit_prepare(const FillArgs& a, const RecRegs& r, uint32_t px, uint32_t py, Hit& h, int& taddr) {
    Tri tr;
    tr.x3 = __uint_as_float(r.q0.x); tr.y3 = __uint_as_float(r.q0.y); tr.a0 = __uint_as_float(r.q0.z); tr.b0 = __uint_as_float(r.q0.w);
    tr.a1 = __uint_as_float(r.q1.x); tr.b1 = __uint_as_float(r.q1.y); tr.inv_area = __uint_as_float(r.q1.z);
    tr.min_x = r.q1.w & 0xFFFF; tr.max_x = r.q1.w >> 16; tr.min_y = r.q2.x & 0xFFFF; tr.max_y = r.q2.x >> 16;
    tr.flags = r.q3.w;
    tr.w0_start = __uint_as_float(r.q4.w); tr.w1_start = __uint_as_float(r.q5.x);
    float w0, w1;
    edge_w(tr, px, py, w0, w1);
    taddr = -1;
    if (!inside_bc(tr, w0, w1, h.bcx, h.bcy, h.bcz)) return false;
    h.vc1 = r.q4.x; h.vc2 = r.q4.y; h.vc3 = r.q4.z; h.flags = tr.flags;
    const uint32_t txid = tr.flags & F_TEX_MASK;
    if (txid == F_TEX_NONE) return true;
    TexDesc d;
    if (a.fp.nt == 1) d = a.tex0; else d = a.tex[txid];
    if (d.width == 0 || d.height == 0) { taddr = -2; return true; }
    const float u1 = __uint_as_float(r.q2.y), u2 = __uint_as_float(r.q2.z), u3 = __uint_as_float(r.q2.w);
    const float v1 = __uint_as_float(r.q3.x), v2 = __uint_as_float(r.q3.y), v3 = __uint_as_float(r.q3.z);
    float u, v;
    if (a.fp.affine) {
        u = h.bcx * u1 + h.bcy * u2 + h.bcz * u3;                // render.rs:1565-1566
        v = h.bcx * v1 + h.bcy * v2 + h.bcz * v3;
    } else {                                                     // render.rs:1568-1579
        const float iz1 = __uint_as_float(r.q5.y), iz2 = __uint_as_float(r.q5.z), iz3 = __uint_as_float(r.q5.w);
        const float inv_z = h.bcx * iz1 + h.bcy * iz2 + h.bcz * iz3;
        const float u_over_z = h.bcx * u1 * iz1 + h.bcy * u2 * iz2 + h.bcz * u3 * iz3;
        const float v_over_z = h.bcx * v1 * iz1 + h.bcy * v2 * iz2 + h.bcz * v3 * iz3;
        u = u_over_z / inv_z;
        v = v_over_z / inv_z;
    }
    const float uw = rem_euclid1(u), vw = rem_euclid1(1.0f - v);                        // Texture15::sample, types.rs:671-681
    const uint32_t tx = min(f2u_sat(uw * (float)d.width), d.width - 1);
    const uint32_t ty = min(f2u_sat(vw * (float)d.height), d.height - 1);
    taddr = (int)(d.offset + ty * d.width + tx);
    return true;
}
// transparency rule on the fetched texel (render.rs:1591-1608 / 8-bit :1348-1352)
template <bool FMT8>
__device__ __forceinline__ bool hit_finish(uint32_t flags, int taddr, uint32_t fetched, uint32_t& texel) {
    if (FMT8) {
        const uint32_t c = taddr == -1 ? 0x00FFFFFFu : (taddr == -2 ? ((uint32_t)B32_BLEND_ERASE << 24) : fetched);
        texel = c;
        return (c >> 24) != B32_BLEND_ERASE;
    }
    uint32_t c = taddr == -1 ? K::C15_WHITE : (taddr == -2 ? K::C15_TRANSPARENT : fetched);
    if (c == K::C15_TRANSPARENT) {
        if (flags & F_BLACK_TR) return false;
        c = K::C15_BLACK_DRAWABLE;
    } else if ((flags & F_BLACK_TR) && (c & ~K::C15_SEMI_BIT & 0xFFFFu) == 0) return false;
    texel = c;
    return true;
}
// texel of the LDS-staged index atlas: [256 x Color15 CLUT][index bytes]; taddr is an address in the texel pool (texture 0 starts at off0)
__device__ __forceinline__ uint32_t atlas_texel(const uint8_t* latlas, int taddr, uint32_t off0) {
    if (taddr < 0) return 0;
    const uint32_t idx = latlas[ATLAS_CLUT_BYTES + ((uint32_t)taddr - off0)];
    return reinterpret_cast<const uint16_t*>(latlas)[idx];
}
template <bool FMT8>
__device__ __forceinline__ uint32_t fetch_texel(const FillArgs& a, int taddr) {
    if (taddr < 0) return 0;
    return FMT8 ? a.texels32[taddr] : (uint32_t)a.texels[taddr];
}

// depth of surface `sid` at the pixel whose barycentrics are in h (render.rs:1546-1550) as a z-buffer priority word
__device__ __forceinline__ bool depth_prio(const FillArgs& a, uint32_t sid, const Hit& h, unsigned long long& P) {
    const uint4 x0 = reinterpret_cast<const uint4*>(a.xrecs + sid)[0];             // iz1, iz2, iz3
    const float inv_z = h.bcx * __uint_as_float(x0.x) + h.bcy * __uint_as_float(x0.y) + h.bcz * __uint_as_float(x0.z);
    const float z = rcp_exact(inv_z);
    P = ((unsigned long long)(~zsort_key(z)) << 32) | (0xFFFFFFFEu - sid);
    return z == z;
}

// A pixel whose winner turned out to be skipped by the texel rule (CHEAP coverage only tested the triangle): the exact runner-up from
// LDS, then (rarer) the best drawn surface below it from the tile list.  Every lane of the wave must call this together (the list scan
// is a wave-level loop over the lanes that need it); `need` = this lane has such a pixel.  On return ok / h / t describe what the
// pixel finally shows (ok false: nothing drawn, the pixel keeps the framebuffer's / the folded clear's value).
template <bool FMT8, bool ZMODE>
__device__ __forceinline__ void repair_pixel(const FillArgs& a, const unsigned long long* sec, bool need, uint32_t row, uint32_t col, uint32_t px, uint32_t py,
                                             uint32_t e0, uint32_t e1, uint32_t lane, bool& ok, Hit& h, unsigned long long& t) {
    const uint32_t W = a.fp.width;
    auto sid_of = [](unsigned long long v) { return ZMODE ? 0xFFFFFFFEu - (uint32_t)v : (uint32_t)v; };
    unsigned long long limit = 0, seed = 0;
    if (need) {
        if (ZMODE) seed = ((unsigned long long)(~zsort_key(a.clear_depth ? __uint_as_float(0x7F7FFFFFu) : a.zbuf[(size_t)py * W + px])) << 32) | 0xFFFFFFFFull;
        const unsigned long long t2 = sec[row * STR64 + col];
        if (t2 > seed) {                                  // (z-buffer mode: the runner-up must itself beat the stored depth)
            ok = hit_test<FMT8>(a, sid_of(t2), px, py, h);
            if (ok) t = t2; else limit = t2;
        }
    }
    unsigned long long fm = __ballot(limit != 0);
    while (fm) {
        const int fl = __builtin_ctzll(fm);
        fm &= fm - 1;
        const uint32_t fx = (uint32_t)__builtin_amdgcn_readlane((int)px, fl), fy = (uint32_t)__builtin_amdgcn_readlane((int)py, fl);
        const unsigned long long lim = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(limit >> 32), fl) << 32) |
                                       (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)limit, fl);
        const unsigned long long sd = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(seed >> 32), fl) << 32) |
                                      (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)seed, fl);
        unsigned long long best = 0;
        for (uint32_t base = e0; base < e1; base += 64) {
            unsigned long long cand = 0;
            if (base + lane < e1) {
                const uint32_t csid = a.pair_vals[base + lane];
                const uint4 cc1 = reinterpret_cast<const uint4*>(a.crecs + csid)[1];
                const uint32_t bbx = cc1.x, bby = cc1.y;
                if (fx >= (bbx & 0xFFFF) && fx < (bbx >> 16) && fy >= (bby & 0xFFFF) && fy < (bby >> 16)) {
                    unsigned long long P = ((unsigned long long)cc1.z << 32) | csid;
                    Hit c;
                    if (ZMODE) {
                        if (hit_test<FMT8>(a, csid, fx, fy, c) && depth_prio(a, csid, c, P) && P < lim && P > sd) cand = P;
                    } else if (P < lim && P > best && hit_test<FMT8>(a, csid, fx, fy, c)) cand = P;
                }
            }
            for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_xor(cand, off); cand = o > cand ? o : cand; }
            best = cand > best ? cand : best;
        }
        if ((int)lane == fl && best) { ok = hit_test<FMT8>(a, sid_of(best), px, py, h); t = best; }
    }
}

// wq: this WAVE's repair queue (64 words of LDS, entry i = row << 6 | col): pixels whose winner was skipped are NOT repaired where they
// are found -- one lane of the wave shading a whole runner-up, behind a record gather and a texel fetch of its own, in two steps out
// of five on the benchmark scene -- but collected and repaired together: when 64 have gathered, and behind the tile's last row.
template <bool FMT8, int NT, bool ZMODE>
__device__ __forceinline__ void shade_tile_p64(const FillArgs& a, const uint32_t* tilebuf, uint32_t e0, uint32_t e1, uint32_t x_lo, uint32_t x_hi,
                                               uint32_t y_lo, uint32_t y_hi, uint32_t ty_top, uint32_t tid, uint32_t lane, uint32_t TH, uint32_t* wq,
                                               const uint8_t* latlas) {
    const FrameParams& fp = a.fp;
    const unsigned long long* top = reinterpret_cast<const unsigned long long*>(tilebuf);
    const unsigned long long* sec = top + TILE_H * STR64;
    const int shading = fp.shading;
    const bool need5 = !fp.affine || !fp.fixed_point || fp.ortho;       // q5: literal-replay start value / 1/z terms
    const uint32_t W = fp.width;
    constexpr uint32_t ROWS_PER_STEP = NT / 64;
    // z-buffer mode: a winner exists when the low word is not the seed's all-ones; its face id is 0xFFFFFFFE - low word
    auto covered = [](unsigned long long t) { return ZMODE ? ((uint32_t)t != 0xFFFFFFFFu) : (t != 0ull); };
    auto sid_of = [](unsigned long long t) { return ZMODE ? 0xFFFFFFFEu - (uint32_t)t : (uint32_t)t; };
    auto put = [&](uint32_t px, uint32_t py, bool ok, const Hit& h, unsigned long long t, bool in) {
        if (ok) {
            a.fb[(size_t)py * W + px] = colour<FMT8>(a, h, shading, px, py);
            if (ZMODE) { float z = zsort_val(~(uint32_t)(t >> 32)); if (z == 0.0f) z = exact_depth_at(a, h.sid, px, py); a.zbuf[(size_t)py * W + px] = z; }
        } else if (in) {          // (Framebuffer::clear folded into the frame: pixels nobody draws get the clear colour, and depth, here)
            if (a.clear_on) a.fb[(size_t)py * W + px] = a.clear_rgba;
            if (ZMODE && a.clear_depth) a.zbuf[(size_t)py * W + px] = __uint_as_float(0x7F7FFFFFu);
        }
    };
    uint32_t lqn = 0;                                   // entries in this wave's queue (wave-uniform)
    auto drain = [&]() {
        const bool act = lane < lqn;
        const uint32_t e = act ? wq[lane] : 0u;
        const uint32_t row = e >> 6, col = e & 63u, px = x_lo + col, py = ty_top + row;
        bool ok = false; Hit h; h.sid = 0;
        unsigned long long t = 0;
        repair_pixel<FMT8, ZMODE>(a, sec, act, row, col, px, py, e0, e1, lane, ok, h, t);
        if (act) put(px, py, ok, h, t, true);
        lqn = 0;
    };
    const unsigned long long below = (1ull << lane) - 1ull;
    for (uint32_t r0 = 0; r0 < TH; r0 += 2 * ROWS_PER_STEP) {
        const uint32_t col = tid & 63;
        const uint32_t rowA = r0 + (tid >> 6), rowB = rowA + ROWS_PER_STEP;
        const uint32_t px = x_lo + col, pyA = ty_top + rowA, pyB = ty_top + rowB;
        const bool inA = rowA < TH && px < x_hi && pyA >= y_lo && pyA < y_hi, inB = rowB < TH && px < x_hi && pyB >= y_lo && pyB < y_hi;
        unsigned long long tA = inA ? top[rowA * STR64 + col] : (ZMODE ? ~0ull : 0ull), tB = inB ? top[rowB * STR64 + col] : (ZMODE ? ~0ull : 0ull);
        const bool cA = covered(tA), cB = covered(tB);
        unsigned long long mA = 0, mB = 0;
        {
            Hit hA, hB;
            hA.sid = hB.sid = 0;
            if (!__ballot(cA || cB)) { put(px, pyA, false, hA, tA, inA); put(px, pyB, false, hB, tB, inB); continue; }
            RecRegs ra, rb;
            rec_load(a, cA ? sid_of(tA) : 0u, need5, ra);             // surface 0's record is a harmless dummy for uncovered pixels
            rec_load(a, cB ? sid_of(tB) : 0u, need5, rb);
            int taA = -1, taB = -1;
            bool okA = cA && hit_prepare(a, ra, px, pyA, hA, taA);
            bool okB = cB && hit_prepare(a, rb, px, pyB, hB, taB);
            // (latlas: the one indexed texture's CLUT + index bytes staged in this workgroup's LDS -- Clut::lookup per shaded pixel,
            // types.rs:390-397 -- instead of the expanded texel from global memory; wave-uniform choice)
            uint32_t fA, fB;
            if (!FMT8 && latlas) { fA = atlas_texel(latlas, okA ? taA : -1, a.tex0.offset); fB = atlas_texel(latlas, okB ? taB : -1, a.tex0.offset); }
            else { fA = fetch_texel<FMT8>(a, okA ? taA : -1); fB = fetch_texel<FMT8>(a, okB ? taB : -1); }
            hA.sid = sid_of(tA); hB.sid = sid_of(tB);
            okA = okA && hit_finish<FMT8>(hA.flags, taA, fA, hA.texel);
            okB = okB && hit_finish<FMT8>(hB.flags, taB, fB, hB.texel);
            // (a covered pixel whose winner is skipped waits in the queue; everything else is final)
            mA = __ballot(cA && !okA); mB = __ballot(cB && !okB);
            if (!FMT8 && !ZMODE && shading == B32_SHADE_NONE) {
                // both colours in one packed pipeline (the results of lanes without a drawn pixel are never stored)
                const float bA[3] = { hA.bcx, hA.bcy, hA.bcz }, bB[3] = { hB.bcx, hB.bcy, hB.bcz };
                const uint32_t vA[3] = { hA.vc1, hA.vc2, hA.vc3 }, vB[3] = { hB.vc1, hB.vc2, hB.vc3 };
                uint32_t colA, colB;
                shade15_pair_rgba(hA.texel, hB.texel, bA, bB, vA, vB, hA.flags, hB.flags, px, pyA, pyB, colA, colB);
                if (okA) a.fb[(size_t)pyA * W + px] = colA; else if (!cA && inA && a.clear_on) a.fb[(size_t)pyA * W + px] = a.clear_rgba;
                if (okB) a.fb[(size_t)pyB * W + px] = colB; else if (!cB && inB && a.clear_on) a.fb[(size_t)pyB * W + px] = a.clear_rgba;
            } else {
                if (!(cA && !okA)) put(px, pyA, okA, hA, tA, inA);
                if (!(cB && !okB)) put(px, pyB, okB, hB, tB, inB);
            }
        }
        if (mA | mB) {
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                const unsigned long long m = which ? mB : mA;
                if (!m) continue;
                const uint32_t n = (uint32_t)__builtin_popcountll(m);
                if (lqn + n > 64u) drain();
                if ((m >> lane) & 1ull) wq[lqn + (uint32_t)__builtin_popcountll(m & below)] = ((which ? rowB : rowA) << 6) | col;
                lqn += n;
            }
        }
    }
    if (lqn) drain();
}

// The straight-line shading phase (RGB555, affine UVs, fixed-point snap, perspective camera, ONE texture fetched from global memory;
// painter's or z-buffer mode; with or without a shading pass): the general shade_tile_p64 reaches the same arithmetic through
// hit_prepare / hit_finish / colour, whose per-pixel branches (texture present?, zero-sized?, literal replay?, inside?) cost the
// benchmark's instantiation ~90 branches and ~470 VALU instructions per two-pixel step.  Here every lane runs the one path -- record
// view, edge values in closed form, barycentrics, UVs, texel address (render.rs:1507-1583, types.rs:671-681), both texel fetches in flight,
// texel rule (render.rs:1591-1608), colour pipeline -- on whatever its two pixels hold (an uncovered pixel computes on surface 0's record
// and stores nothing of it).  The winner of a covered pixel passed the inside test during coverage (same arithmetic, or the span form
// proven equal to it), so it is not evaluated again.  A step in which some winner must replay the edge walk literally (SH_SLOW) takes
// the general per-pixel functions; skipped winners go to the wave's repair queue as in the general form.
template <int NT, bool ZMODE>
__device__ __forceinline__ void shade_tile_plain(const FillArgs& a, const uint32_t* tilebuf, uint32_t e0, uint32_t e1, uint32_t x_lo, uint32_t x_hi,
                                                 uint32_t y_lo, uint32_t y_hi, uint32_t ty_top, uint32_t tid, uint32_t lane, uint32_t TH, uint32_t* wq) {
    const FrameParams& fp = a.fp;
    const unsigned long long* top = reinterpret_cast<const unsigned long long*>(tilebuf);
    const unsigned long long* sec = top + TILE_H * STR64;
    const uint32_t W = fp.width;
    const int shading = fp.shading;
    constexpr uint32_t ROWS_PER_STEP = NT / 64;
    const TexDesc d = a.tex0;
    const float twf = (float)d.width, thf = (float)d.height;
    const uint32_t col = tid & 63, px = x_lo + col;
    const float fx = (float)px;
    const bool in_x = px < x_hi;
    const float ZMAX = __uint_as_float(0x7F7FFFFFu);
    // z-buffer mode: a winner exists when the low word is not the seed's all-ones; its face id is 0xFFFFFFFE - low word
    auto covered = [](unsigned long long t) { return ZMODE ? ((uint32_t)t != 0xFFFFFFFFu) : (t != 0ull); };
    auto sid_of = [](unsigned long long t) { return ZMODE ? 0xFFFFFFFEu - (uint32_t)t : (uint32_t)t; };
    // a pixel nobody draws inside the band: the folded Framebuffer::clear (colour, and depth in z-buffer mode)
    auto leave = [&](uint32_t py) {
        if (a.clear_on) a.fb[(size_t)py * W + px] = a.clear_rgba;
        if (ZMODE && a.clear_depth) a.zbuf[(size_t)py * W + px] = ZMAX;
    };
    // fb.zbuffer[idx] = z of the winner (render.rs:1686-1688); a key that decodes to zero does not carry the sign: recomputed
    auto store_depth = [&](unsigned long long t, uint32_t sid, uint32_t py) {
        float z = zsort_val(~(uint32_t)(t >> 32));
        if (z == 0.0f) z = exact_depth_at(a, sid, px, py);
        a.zbuf[(size_t)py * W + px] = z;
    };
    uint32_t lqn = 0;                                   // entries in this wave's repair queue (wave-uniform)
    auto drain = [&]() {
        const bool act = lane < lqn;
        const uint32_t e = act ? wq[lane] : 0u;
        const uint32_t row = e >> 6, c = e & 63u, qx = x_lo + c, qy = ty_top + row;
        bool ok = false; Hit h; h.sid = 0;
        unsigned long long t = 0;
        repair_pixel<false, ZMODE>(a, sec, act, row, c, qx, qy, e0, e1, lane, ok, h, t);
        if (act) {
            if (ok) {
                a.fb[(size_t)qy * W + qx] = colour<false>(a, h, shading, qx, qy);
                if (ZMODE) { float z = zsort_val(~(uint32_t)(t >> 32)); if (z == 0.0f) z = exact_depth_at(a, h.sid, qx, qy); a.zbuf[(size_t)qy * W + qx] = z; }
            } else {
                if (a.clear_on) a.fb[(size_t)qy * W + qx] = a.clear_rgba;
                if (ZMODE && a.clear_depth) a.zbuf[(size_t)qy * W + qx] = ZMAX;
            }
        }
        lqn = 0;
    };
    const unsigned long long below = (1ull << lane) - 1ull;
    auto f = [](uint32_t w) { return __uint_as_float(w); };
    for (uint32_t r0 = 0; r0 < TH; r0 += 2 * ROWS_PER_STEP) {
        const uint32_t rowA = r0 + (tid >> 6), rowB = rowA + ROWS_PER_STEP;
        const uint32_t pyA = ty_top + rowA, pyB = ty_top + rowB;
        const bool inA = rowA < TH && in_x && pyA >= y_lo && pyA < y_hi, inB = rowB < TH && in_x && pyB >= y_lo && pyB < y_hi;
        const unsigned long long tA = inA ? top[rowA * STR64 + col] : (ZMODE ? ~0ull : 0ull), tB = inB ? top[rowB * STR64 + col] : (ZMODE ? ~0ull : 0ull);
        const bool cA = covered(tA), cB = covered(tB);
        uint32_t* outA = a.fb + (size_t)pyA * W + px;
        uint32_t* outB = a.fb + (size_t)pyB * W + px;
        if (!__ballot(cA || cB)) {
            if (inA) leave(pyA);
            if (inB) leave(pyB);
            continue;
        }
        const uint32_t sidA = cA ? sid_of(tA) : 0u, sidB = cB ? sid_of(tB) : 0u;      // (surface 0's record for an uncovered pixel: read, never used)
        const uint4* spA = reinterpret_cast<const uint4*>(a.srecs + sidA);
        const uint4* spB = reinterpret_cast<const uint4*>(a.srecs + sidB);
        const uint4 a0q = spA[0], a1q = spA[1], a2q = spA[2], a3q = spA[3];
        const uint4 b0q = spB[0], b1q = spB[1], b2q = spB[2], b3q = spB[3];
        const uint32_t shA = a3q.w >> 24, shB = b3q.w >> 24;
        unsigned long long mA, mB;
        if (__ballot((cA && (shA & SH_SLOW)) || (cB && (shB & SH_SLOW)))) {
            // rare: a winner whose edge walk is replayed literally -- the general per-pixel functions for this step
            Hit hA, hB;
            const bool okA = cA && hit_test<false>(a, sidA, px, pyA, hA);
            const bool okB = cB && hit_test<false>(a, sidB, px, pyB, hB);
            if (okA) { *outA = colour<false>(a, hA, shading, px, pyA); if (ZMODE) store_depth(tA, sidA, pyA); } else if (!cA && inA) leave(pyA);
            if (okB) { *outB = colour<false>(a, hB, shading, px, pyB); if (ZMODE) store_depth(tB, sidB, pyB); } else if (!cB && inB) leave(pyB);
            mA = __ballot(cA && !okA); mB = __ballot(cB && !okB);
        } else {
            float bA[3], bB[3];
            uint32_t taA, taB;
            {   // pixel A: render.rs:1507-1510 (edges), :1517-1518 / 1706-1712 in closed form (exact integers), :1536-1538, :1565-1566, types.rs:671-681
                const float x3 = f(a1q.x), y3 = f(a1q.y), inv = f(a1q.z);
                const float ea0 = f(a0q.w) - y3, eb0 = x3 - f(a0q.z), ea1 = y3 - f(a0q.y), eb1 = f(a0q.x) - x3;
                const float dx = fx - x3, dy = (float)pyA - y3;
                const float w0 = ea0 * dx + eb0 * dy, w1 = ea1 * dx + eb1 * dy;
                bA[0] = w0 * inv; bA[1] = w1 * inv; bA[2] = 1.0f - bA[0] - bA[1];
                const float u = bA[0] * f(a2q.x) + bA[1] * f(a2q.y) + bA[2] * f(a2q.z);
                const float v = bA[0] * f(a2q.w) + bA[1] * f(a3q.x) + bA[2] * f(a3q.y);
                const float uw = rem_euclid1(u), vw = rem_euclid1(1.0f - v);
                const uint32_t tx = min(f2u_sat(uw * twf), d.width - 1), ty = min(f2u_sat(vw * thf), d.height - 1);
                taA = cA ? d.offset + ty * d.width + tx : d.offset;
            }
            {
                const float x3 = f(b1q.x), y3 = f(b1q.y), inv = f(b1q.z);
                const float ea0 = f(b0q.w) - y3, eb0 = x3 - f(b0q.z), ea1 = y3 - f(b0q.y), eb1 = f(b0q.x) - x3;
                const float dx = fx - x3, dy = (float)pyB - y3;
                const float w0 = ea0 * dx + eb0 * dy, w1 = ea1 * dx + eb1 * dy;
                bB[0] = w0 * inv; bB[1] = w1 * inv; bB[2] = 1.0f - bB[0] - bB[1];
                const float u = bB[0] * f(b2q.x) + bB[1] * f(b2q.y) + bB[2] * f(b2q.z);
                const float v = bB[0] * f(b2q.w) + bB[1] * f(b3q.x) + bB[2] * f(b3q.y);
                const float uw = rem_euclid1(u), vw = rem_euclid1(1.0f - v);
                const uint32_t tx = min(f2u_sat(uw * twf), d.width - 1), ty = min(f2u_sat(vw * thf), d.height - 1);
                taB = cB ? d.offset + ty * d.width + tx : d.offset;
            }
            const uint32_t fetA = a.texels[taA], fetB = a.texels[taB];              // both fetches in flight
            // texture slot 0xFFFF = untextured: Color15::WHITE (render.rs:1585); then the transparency rule (render.rs:1591-1608)
            const bool noneA = ((a1q.w >> 24) | ((a3q.z >> 24) << 8)) == F_TEX_NONE, noneB = ((b1q.w >> 24) | ((b3q.z >> 24) << 8)) == F_TEX_NONE;
            uint32_t cA15 = noneA ? K::C15_WHITE : fetA, cB15 = noneB ? K::C15_WHITE : fetB;
            const bool btA = (shA & SH_BLACK_TR) != 0, btB = (shB & SH_BLACK_TR) != 0;
            const bool skipA = btA && (cA15 & ~K::C15_SEMI_BIT & 0xFFFFu) == 0, skipB = btB && (cB15 & ~K::C15_SEMI_BIT & 0xFFFFu) == 0;     // 0x0000 or black with black_transparent
            cA15 = cA15 == K::C15_TRANSPARENT ? K::C15_BLACK_DRAWABLE : cA15; cB15 = cB15 == K::C15_TRANSPARENT ? K::C15_BLACK_DRAWABLE : cB15;
            const bool okA = cA && !skipA, okB = cB && !skipB;
            const uint32_t vA[3] = { a1q.w & 0xFFFFFFu, a3q.z & 0xFFFFFFu, a3q.w & 0xFFFFFFu }, vB[3] = { b1q.w & 0xFFFFFFu, b3q.z & 0xFFFFFFu, b3q.w & 0xFFFFFFu };
            const uint32_t flA = (shA & SH_DITHER) ? F_DITHER : 0u, flB = (shB & SH_DITHER) ? F_DITHER : 0u;
            uint32_t colA, colB;
            if (shading == B32_SHADE_NONE) {
                shade15_pair_rgba(cA15, cB15, bA, bB, vA, vB, flA, flB, px, pyA, pyB, colA, colB);
            } else {          // flat / Gouraud: the surface's nine vertex shades (render.rs:1629-1645)
                float sA[9], sB[9];
#pragma unroll
                for (int j = 0; j < 9; ++j) { sA[j] = a.shades[(size_t)sidA * 9 + j]; sB[j] = a.shades[(size_t)sidB * 9 + j]; }
                colA = shade15<true>(cA15, bA[0], bA[1], bA[2], vA[0], vA[1], vA[2], flA, shading, sA, px, pyA);
                colB = shade15<true>(cB15, bB[0], bB[1], bB[2], vB[0], vB[1], vB[2], flB, shading, sB, px, pyB);
            }
            if (okA) { *outA = colA; if (ZMODE) store_depth(tA, sidA, pyA); } else if (!cA && inA) leave(pyA);
            if (okB) { *outB = colB; if (ZMODE) store_depth(tB, sidB, pyB); } else if (!cB && inB) leave(pyB);
            mA = __ballot(cA && !okA); mB = __ballot(cB && !okB);
        }
        if (mA | mB) {
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                const unsigned long long m = which ? mB : mA;
                if (!m) continue;
                const uint32_t n = (uint32_t)__builtin_popcountll(m);
                if (lqn + n > 64u) drain();
                if ((m >> lane) & 1ull) wq[lqn + (uint32_t)__builtin_popcountll(m & below)] = ((which ? rowB : rowA) << 6) | col;
                lqn += n;
            }
        }
    }
    if (lqn) drain();
}

// One 256-thread workgroup per 64x16 strip of a 64x64 tile; each wave shades a 64-pixel row segment at a time (256-B coalesced
// visibility reads / framebuffer writes), 4 rows per wave, and the strips of a tile are placed on one XCD, so a surface record
// is pulled through one L2 only (row-major traversal re-fetched every record once per row it covers: 145 MB instead of ~85 MB).
template <bool FMT8>
__global__ __launch_bounds__(256) void k_shade(FillArgs a) {
    if (a.ctrl->abort || a.ctrl->need_global_sort) return;
    const FrameParams& fp = a.fp;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int shading = fp.shading;
    const uint32_t W = fp.width;
    // block b -> (tile, 16-row strip): the four strips of a tile share b % 8, i.e. (as dispatched today) the same XCD and L2
    const uint32_t g = blockIdx.x >> 3;
    const uint32_t tile = (g >> 2) * 8 + (blockIdx.x & 7), strip = g & 3;
    if (tile >= fp.tiles_x * fp.tiles_y) return;
    const uint32_t seg_x = (tile % fp.tiles_x) * TILE_W;
    const uint32_t ty_top = fp.tile_yb + (tile / fp.tiles_x) * TILE_H;       // (the keyed pipelines never cut or grade their tiles)
    const uint32_t e0 = a.tile_keys_only ? a.ranges[tile] : a.ranges[2 * tile];
    const uint32_t px = seg_x + lane;
    const bool inb = px < W;
    for (uint32_t r = strip * 16 + wave; r < strip * 16 + 16; r += 4) {
    const uint32_t py = ty_top + r;
    if (py < fp.band_y0 || py >= fp.band_y1) continue;
    const uint32_t ve = inb ? a.vis[(size_t)py * W + px] : 0u;
    if (!__ballot(ve != 0)) continue;
    // decode (CHEAP coverage packs the runner-up list position in the high half, see k_cover)
    const bool long_list = !a.exact_coverage && (ve >> 31);
    const uint32_t li = a.exact_coverage ? ve : (long_list ? (ve & 0x7FFFFFFFu) : (ve & 0xFFFFu));
    const uint32_t second = (a.exact_coverage || long_list) ? 0u : (ve >> 16);
    Hit h;
    bool have = false;
    uint32_t scan_from = 0;                     // > 0: list positions <= scan_from still have to be searched
    if (li && !(have = hit_test<FMT8>(a, a.pair_vals[e0 + li - 1], px, py, h))) {
        // CHEAP coverage only: the top surface is skipped at this pixel -> highest surface below it whose fragment is drawn.
        if (long_list) scan_from = li - 1;
        else if (second) {                      // exact runner-up from k_cover: almost always the answer (else ~1/256 again)
            if (!(have = hit_test<FMT8>(a, a.pair_vals[e0 + second - 1], px, py, h))) scan_from = second - 1;
        }                                       // second == 0: no other surface covers the pixel, it keeps the framebuffer value
    }
    // rare: the wave scans the tile list downward, 64 entries per step; only the coverage test runs per candidate
    unsigned long long fm = __ballot(scan_from != 0);
    while (fm) {
        const int fl = __builtin_ctzll(fm);
        fm &= fm - 1;
        const uint32_t fx = (uint32_t)__builtin_amdgcn_readlane((int)px, fl), ftop = (uint32_t)__builtin_amdgcn_readlane((int)scan_from, fl);
        for (uint32_t top = ftop; top > 0; top = top > 64 ? top - 64 : 0) {               // list positions top-lane, descending
            Hit c;
            bool hit = false;
            if (lane < top) {
                const uint32_t cli = top - lane;
                const uint32_t csid = a.pair_vals[e0 + cli - 1];
                const uint4 cc1 = reinterpret_cast<const uint4*>(a.crecs + csid)[1];
                const uint32_t bbx = cc1.x, bby = cc1.y;
                if (fx >= (bbx & 0xFFFF) && fx < (bbx >> 16) && py >= (bby & 0xFFFF) && py < (bby >> 16)) hit = hit_test<FMT8>(a, csid, fx, py, c);
            }
            const unsigned long long hm = __ballot(hit);
            if (hm) {                                                                     // lowest lane == highest list position
                const int hl = __builtin_ctzll(hm);
                const float bx_ = bcf(c.bcx, hl), by_ = bcf(c.bcy, hl), bz_ = bcf(c.bcz, hl);
                const uint32_t t_ = bcu(c.texel, hl), v1_ = bcu(c.vc1, hl), v2_ = bcu(c.vc2, hl), v3_ = bcu(c.vc3, hl), f_ = bcu(c.flags, hl), s_ = bcu(c.sid, hl);
                if ((int)lane == fl) { h.bcx = bx_; h.bcy = by_; h.bcz = bz_; h.texel = t_; h.vc1 = v1_; h.vc2 = v2_; h.vc3 = v3_; h.flags = f_; h.sid = s_; have = true; }
                break;
            }
        }
    }
    if (have) a.fb[(size_t)py * W + px] = colour<FMT8>(a, h, shading, px, py);
    }
}

// ------------------------------------------------------------------------------------------------ k_blend
// Depth test of the transparent pass in z-buffer mode (no z write).  Editor-alpha stores reject on `z >= zbuffer`
// (render.rs:595-605), plain stores draw on `z < zbuffer` (render.rs:1683); the two differ only for NaN depths.
__device__ __forceinline__ bool ztest(const Tri& t, float bcx, float bcy, float bcz, int zmode, float zb) {
    if (!zmode) return true;
    const float inv_z = bcx * t.iz1 + bcy * t.iz2 + bcz * t.iz3;
    const float z = rcp_exact(inv_z);
    return ((t.flags >> F_ALPHA_SHIFT) < 255) ? !(z >= zb) : (z < zb);
}

// One fragment of the ordered pass at a pixel the inside test accepted.  Returns true when a pixel store happened.
template <bool FMT8>
__device__ __forceinline__ bool blend_fragment(const FillArgs& a, const Tri& tr, float bcx, float bcy, float bcz, uint32_t px, uint32_t py,
                                               uint32_t vc1, uint32_t vc2, uint32_t vc3, int shading, const float* shv,
                                               uint32_t* dst, float* zdst, int zmode, bool xray) {
    const bool affine = a.fp.affine != 0;
    uint32_t texel;
    if (FMT8) {
        // rasterize_triangle (render.rs:1302-1424): the early `z >= zbuffer` reject and the store's own test collapse into one
        // test per store kind (they differ only for NaN depths); every store that passes also writes the depth.
        const uint32_t alpha = tr.flags >> F_ALPHA_SHIFT;
        float z = 0.0f;
        if (zmode) {
            const float inv_z = bcx * tr.iz1 + bcy * tr.iz2 + bcz * tr.iz3;
            z = rcp_exact(inv_z);
            const float zb = *zdst;
            if (alpha < 255 ? (z >= zb) : !(z < zb)) return false;               // render.rs:387 / :432, :1407
        }
        if (!texel_drawn<0, true>(tr, bcx, bcy, bcz, reinterpret_cast<const uint16_t*>(a.texels32), nullptr, texel, affine)) return false;
        const uint32_t col = shade8(texel, bcx, bcy, bcz, vc1, vc2, vc3, tr.flags, shading, shv, px, py);
        if (zmode) *zdst = z;
        *dst = store8(*dst, col, alpha);
        return true;
    }
    if (!ztest(tr, bcx, bcy, bcz, zmode, *zdst)) return false;
    if (!texel_drawn<0>(tr, bcx, bcy, bcz, a.texels, nullptr, texel, affine)) return false;
    const uint32_t out15 = shade15(texel, bcx, bcy, bcz, vc1, vc2, vc3, tr.flags, shading, shv, px, py);
    *dst = store_blend(*dst, out15, tr.flags, xray);
    return true;
}

// The ordered pass, PIXEL-centric.  What must be ordered is, per pixel, the sequence of its own fragments -- nothing else: two surfaces
// that do not share a pixel commute.  So a lane owns a pixel and walks, in painter's order, the surfaces of the batch whose clipped
// bounding box holds it: the candidates of pixel (x, y) are `rowmask[y] & colmask[x]` -- one 64-bit word per tile row and per tile
// column with a bit per surface of the batch (a box is an x-range times a y-range, so the AND is exact; 1 KB of LDS, built with
// ballots).  A cheap loop finds the lane's next candidate that passes the reference's inside test (two LDS quads of the record, the
// closed-form edge values or the literal replay), then the lanes that found one run the texel / colour pipeline and blend into the
// pixel held in a register.  No fragment buffer, no chunks, no per-surface serial walk: the sequential depth of a wave's row is the
// largest number of fragments any one of its 64 pixels receives, the blend chain never leaves the registers, and work is
// proportional to fragments.  (Rounds 1-3 generated the fragments of a chunk into LDS slots and applied them surface after surface
// per band of rows: every wave was busy for the SUM of the surfaces reaching its rows.)
#ifndef B32_BLEND_NT
#define B32_BLEND_NT 256
#endif
constexpr uint32_t BLEND_LIST_CAP = 8;    // fragments a lane notes per round (16 bits each: 1 KB of LDS per wave)
constexpr uint32_t SREC_Q = 9;            // quads per staged record: 8 + 1 of padding (lanes read the records of DIFFERENT surfaces: a 128-byte stride puts them all on 8 banks)
constexpr uint32_t BT_STRIDE = 64;        // the colour tile's row stride in words: a lane only ever touches column `lane`, whatever the row -- no padding needed
constexpr size_t BLEND_TILE_BYTES = (size_t)TILE_H * BT_STRIDE * 4;
constexpr int BLEND_NT = B32_BLEND_NT;        // 4-wave workgroups, four per CU (registers: 4 waves per SIMD): tiles in flight hide the list -> record -> texel latencies
// (31 KB: five workgroups per CU, 1280 places for the 1200 tiles of a 2560x1920 frame -- with four, a second round of 176 workgroups
// doubled the kernel's time; the depth tile only for the 8-bit path in z-buffer mode, whose depth test needs the running depth)
__host__ __device__ constexpr size_t blend_lds_bytes(bool depth_tile) {
    return 256 + 64 * SREC_Q * 16 + 1024 + (size_t)(BLEND_NT / 64) * BLEND_LIST_CAP * 64 * 2 + BLEND_TILE_BYTES * (depth_tile ? 2 : 1);
}

template <int NT, bool FMT8, bool GATHER = false>
__global__ __launch_bounds__(NT, 5) void k_blend(FillArgs a) {        // 5 waves per SIMD: at most 96 VGPRs
    constexpr int NW = NT / 64;
    // dynamic LDS (blend_lds_bytes): [wf 256 B][the batch's 64 records 9 KB][row masks, column masks 1 KB][fragment lists 2 KB per wave]
    // [tile colours][tile depths, z-buffer mode only]
    extern __shared__ __attribute__((aligned(16))) unsigned char bsm[];
    unsigned long long* wf = reinterpret_cast<unsigned long long*>(bsm);
    uint4* srec = reinterpret_cast<uint4*>(bsm + 256);                      // the batch's 64 surface records: 8 x 16 B each (q0..q5, texture, id)
    unsigned long long* rowmask = reinterpret_cast<unsigned long long*>(bsm + 256 + 64 * SREC_Q * 16);
    unsigned long long* colmask = rowmask + 64;
    uint16_t* lists = reinterpret_cast<uint16_t*>(bsm + 256 + 64 * SREC_Q * 16 + 1024);        // per wave: BLEND_LIST_CAP x 64 entries
    uint32_t* tilebuf = reinterpret_cast<uint32_t*>(bsm + 256 + 64 * SREC_Q * 16 + 1024 + NW * BLEND_LIST_CAP * 64 * 2);
    float* tilez = reinterpret_cast<float*>(tilebuf + TILE_H * BT_STRIDE);   // 8-bit path in z-buffer mode only (the RGB555 transparent pass never writes depth:
                                                                            // its test runs in loop (A) against the depth buffer itself)
    constexpr bool DEPTH_TILE = FMT8;
    static_assert(512 % NT == 0 && NT >= 64, "the batch loader deals 512 quads to the workgroup");
    // the priority sort runs before the tile's pixels are stored to LDS: it uses the colour tile's space
    static_assert(!GATHER || BLEND_SORT_CAP * 8 <= BLEND_TILE_BYTES, "the priority sort aliases the colour tile");
    static_assert(NW * 8 <= 256, "wf");
    if (a.ctrl->abort || a.ctrl->need_global_sort) return;
    const FrameParams& fp = a.fp;
    const uint32_t tile = blockIdx.x;
    // x-ray: every surface blends (render.rs:1671-1673), so the ordered pass walks the opaque list too, then the transparent one
    // (8-bit path with blending texels / editor alpha: one list, same ordered walk, render.rs:2193-2202)
    const bool xray = fp.xray != 0;
    const uint32_t e1 = a.tile_keys_only ? a.tile_mid[tile] : a.ranges[2 * tile + (a.ordered_all ? 0 : 1)];
    const uint32_t e2 = (a.inline_bin || a.direct_bin) ? (tile + 1) * a.list_stride : (a.tile_keys_only ? a.ranges[tile + 1] : a.ranges[2 * tile + 2]);
    if (e1 == e2) return;
    const int zmode = (fp.zmode && !xray) ? 1 : 0;               // x-ray skips the depth test (render.rs:1553)
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const int shading = fp.shading;
    const bool affine = fp.affine != 0;
    const uint32_t txi = tile % fp.tiles_x;
    const uint32_t x_lo = txi * TILE_W, x_hi = min(x_lo + TILE_W, fp.width);
    uint32_t TH, ty_top;                            // 64, or fewer rows when the sort-free path runs on cut tiles (LDS layout unchanged)
    tile_row_geom(fp, tile / fp.tiles_x, ty_top, TH);
    const uint32_t y_lo = max(ty_top, fp.band_y0), y_hi = min(ty_top + TH, fp.band_y1);
    // the tile's pixels (and depths) are REQUESTED before the sort prelude below and stored to LDS behind it: their latency passes behind
    // the prelude's own chain of dependent global accesses (list -> keys -> sorted list)
    constexpr int TILE_ITERS = TILE_W * TILE_H / NT;
    uint32_t tpx[TILE_ITERS]; float tpz[TILE_ITERS];
#pragma unroll
    for (int it = 0; it < TILE_ITERS; ++it) {
        const uint32_t p = tid + (uint32_t)it * NT, row = p >> 6, col = p & 63;
        const uint32_t px = x_lo + col, py = ty_top + row;
        const bool inb = row < TH && px < x_hi && py >= y_lo && py < y_hi;
        tpx[it] = inb ? a.fb[(size_t)py * fp.width + px] : 0u;
        tpz[it] = (DEPTH_TILE && zmode && inb) ? a.zbuf[(size_t)py * fp.width + px] : 0.0f;
    }
    if (GATHER) {
        // sort-free binning left the transparent entries [e1, e2) in arbitrary order: put them in painter's order (descending depth,
        // ties in face order, render.rs:2527-2532) by ranking the 64-bit priorities (key << 32 | face id) -- all distinct -- in LDS
        unsigned long long* gprio = reinterpret_cast<unsigned long long*>(tilebuf);
        const uint32_t n = e2 - e1;                    // <= BLEND_SORT_CAP (k_place_spans raised need_global_sort otherwise)
        for (uint32_t i = threadIdx.x; i < n; i += NT) { const uint32_t sid = a.pair_vals[e1 + i]; gprio[i] = ((unsigned long long)a.keys[sid] << 32) | sid; }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += NT) {
            const unsigned long long P = gprio[i];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < n; ++j) rank += gprio[j] < P ? 1u : 0u;
            a.pair_vals[e1 + rank] = (uint32_t)P;
        }
        __syncthreads();
    }
#pragma unroll
    for (int it = 0; it < TILE_ITERS; ++it) {
        const uint32_t p = tid + (uint32_t)it * NT, row = p >> 6, col = p & 63;
        if (row < TH) { tilebuf[row * BT_STRIDE + col] = tpx[it]; if (DEPTH_TILE && zmode) tilez[row * BT_STRIDE + col] = tpz[it]; }
    }
    __syncthreads();
    uint32_t drawn = 0;                             // pixel stores of this lane (fragment counting)
    uint16_t* mylist = lists + wave * (BLEND_LIST_CAP * 64);
    const TexDesc none = { 0, 0, 0, 0 };
    const uint32_t n_tr = e2 - e1;
    for (uint32_t cs = 0; cs < n_tr; cs += 64) {
        const uint32_t cnt = min(64u, n_tr - cs);
        // stage the batch's records in LDS once per workgroup: 512 quads, each assembled from the compact records (q0..q5 of the surface's
        // view, then the texture descriptor + face id, then a spare)
        for (uint32_t q = tid; q < 512u; q += NT) {
            const uint32_t sfc = q >> 3, part = q & 7;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (sfc < cnt) {
                const uint32_t sid = a.pair_vals[e1 + cs + sfc];
                const uint4* cp = reinterpret_cast<const uint4*>(a.crecs + sid);
                const uint4* sp = reinterpret_cast<const uint4*>(a.srecs + sid);
                const uint4* xp = reinterpret_cast<const uint4*>(a.xrecs + sid);
                const uint4 c1 = cp[1];
                const bool aux = !affine || zmode || (c1.w & F_SLOW);
                if (part < 2) {                              // q0, q1: edges from the vertices + bbx
                    const uint4 s0 = sp[0], s1 = sp[1];
                    RecView rv;
                    view_edges_from_shade(rv, s0, s1);
                    rv.q1.w = c1.x;
                    v = part == 0 ? rv.q0 : rv.q1;
                } else if (part == 2) { const uint4 s2 = sp[2]; v = make_uint4(c1.y, s2.x, s2.y, s2.z); }            // bby, u1, u2, u3
                else if (part == 3) { const uint4 s2 = sp[2], s3 = sp[3]; v = make_uint4(s2.w, s3.x, s3.y, c1.w); }  // v1, v2, v3, flags
                else if (part == 4) {
                    const uint4 s1 = sp[1], s3 = sp[3];
                    v = make_uint4(s1.w & 0xFFFFFFu, s3.z & 0xFFFFFFu, s3.w & 0xFFFFFFu, aux ? xp[0].w : 0u);        // vc1, vc2, vc3, w0_start
                } else if (part == 5) { if (aux) { const uint4 x0 = xp[0], x1 = xp[1]; v = make_uint4(x1.x, x0.x, x0.y, x0.z); } }   // w1_start, iz1..3
                else if (part == 6) {
                    const uint32_t txid = c1.w & F_TEX_MASK;
                    TexDesc d = none;
                    if (txid != F_TEX_NONE) { if (fp.nt == 1) d = a.tex0; else d = a.tex[txid]; }      // (one texture: no descriptor gather)
                    v = make_uint4(d.width, d.height, d.offset, sid);
                }
            }
            srec[sfc * SREC_Q + part] = v;
        }
        __syncthreads();
        // lane <-> surface view of the batch: clipped bounding box in this tile (band rows only); editor_alpha == 0 draws nothing
        // (render.rs:1664-1669)
        const uint4 mq1 = srec[lane * SREC_Q + 1], mq2 = srec[lane * SREC_Q + 2], mq3 = srec[lane * SREC_Q + 3];
        const uint32_t my_flags = mq3.w;
        const uint32_t bx0 = max(mq1.w & 0xFFFF, x_lo), bx1 = min(mq1.w >> 16, x_hi);
        const uint32_t by0 = max(mq2.x & 0xFFFF, y_lo), by1 = min(mq2.x >> 16, y_hi);
        const bool live = lane < cnt && bx0 < bx1 && by0 < by1 && (my_flags >> F_ALPHA_SHIFT) != 0;
        const unsigned long long slowmask = __ballot(live && (my_flags & F_SLOW));     // literal edge-walk replay (float / ortho projection, huge coordinates)
        for (uint32_t r = wave; r < 128u; r += NW) {         // row masks [0, 64), column masks [64, 128): contiguous in LDS
            unsigned long long mk;
            if (r < 64u) { const uint32_t y = ty_top + r; mk = __ballot(live && by0 <= y && y < by1); }
            else { const uint32_t x = x_lo + (r - 64u); mk = __ballot(live && bx0 <= x && x < bx1); }
            if (lane == 0) rowmask[r] = mk;
        }
        __syncthreads();
        const unsigned long long cm = colmask[lane];
        const uint32_t px = x_lo + lane;
        // The lane owns column `lane` of the rows wave, wave + NW, ...  Two loops per round, so that neither waits for the other's
        // stragglers: (A) every lane runs through its pixels' candidates, one inside test per step, and notes the fragments that pass
        // (row index, surface) in its own list -- a column of a per-wave LDS array, 16 bits per entry; (B) step k of the colour
        // pipeline takes every lane's k-th fragment: nobody searches there, and the wave's sequential depth is the largest number of
        // fragments one lane's pixels receive in total (not, as with lanes in step per row, the sum over the rows of each row's
        // busiest pixel).  A lane whose list is full resumes its search in the next round (ascending order is kept).
        uint32_t rows = 0;                              // the lane's rows that have candidates, bit i <-> row wave + i * NW
        for (uint32_t i = 0; i < (uint32_t)(TILE_H / NW); ++i) {
            const uint32_t row = wave + i * NW;
            if (row < TH && (rowmask[row] & cm) != 0ull) rows |= 1u << i;
        }
        unsigned long long m = 0ull;
        uint32_t ri = 0;
        const bool ztest_a = !FMT8 && zmode;            // RGB555: the depth buffer is read-only in this pass, so the test can run before the colour pipeline
        float zrow = 0.0f;                              // depth of the lane's current pixel
        for (;;) {
            uint32_t n = 0;
            for (;;) {                                      // (A)
                const bool can = n < BLEND_LIST_CAP && (m != 0ull || rows != 0u);
                if (!__ballot(can)) break;
                if (can) {
                    if (m == 0ull) {
                        ri = (uint32_t)__builtin_ctz(rows); rows &= rows - 1u; m = rowmask[wave + ri * NW] & cm;
                        if (ztest_a) zrow = a.zbuf[(size_t)(ty_top + wave + ri * NW) * fp.width + px];
                    }
                    const uint32_t j = (uint32_t)__builtin_ctzll(m);
                    m &= m - 1ull;
                    const uint4 r0 = srec[j * SREC_Q], r1 = srec[j * SREC_Q + 1];
                    Tri t;
                    t.x3 = __uint_as_float(r0.x); t.y3 = __uint_as_float(r0.y); t.a0 = __uint_as_float(r0.z); t.b0 = __uint_as_float(r0.w);
                    t.a1 = __uint_as_float(r1.x); t.b1 = __uint_as_float(r1.y); t.inv_area = __uint_as_float(r1.z);
                    const uint32_t py = ty_top + wave + ri * NW;
                    float w0, w1, bcx, bcy, bcz;
                    if (!((slowmask >> j) & 1ull)) {        // exact integers (k_setup guard): closed form == accumulation
                        const float dx = (float)px - t.x3, dy = (float)py - t.y3;
                        w0 = t.a0 * dx + t.b0 * dy; w1 = t.a1 * dx + t.b1 * dy;
                    } else {
                        t.min_x = r1.w & 0xFFFF; t.min_y = srec[j * SREC_Q + 2].x & 0xFFFF;
                        t.w0_start = __uint_as_float(srec[j * SREC_Q + 4].w); t.w1_start = __uint_as_float(srec[j * SREC_Q + 5].x);
                        replay_w(t, px, py, w0, w1);
                    }
                    bool pass = inside_bc(t, w0, w1, bcx, bcy, bcz);                                                          // render.rs:1536-1542
                    if (pass && ztest_a) {
                        const uint4 r5 = srec[j * SREC_Q + 5];
                        t.iz1 = __uint_as_float(r5.y); t.iz2 = __uint_as_float(r5.z); t.iz3 = __uint_as_float(r5.w);
                        t.flags = srec[j * SREC_Q + 3].w;
                        pass = ztest(t, bcx, bcy, bcz, 1, zrow);
                    }
                    if (pass) { mylist[n * 64 + lane] = (uint16_t)((ri << 6) | j); ++n; }
                }
            }
            const uint32_t nmax = (uint32_t)__builtin_amdgcn_readlane((int)dpp_max_scan(n), 63);
            if (nmax == 0u) break;
            for (uint32_t k = 0; k < nmax; ++k) {           // (B)
                if (k < n) {
                    const uint32_t e = mylist[k * 64 + lane], j = e & 63u, row = wave + (e >> 6) * NW;
                    const uint32_t py = ty_top + row, ti = row * BT_STRIDE + lane;
                    const uint4 r0 = srec[j * SREC_Q], r1 = srec[j * SREC_Q + 1], r2 = srec[j * SREC_Q + 2], r3 = srec[j * SREC_Q + 3], r4 = srec[j * SREC_Q + 4], r6 = srec[j * SREC_Q + 6];
                    Tri tr;
                    tr.x3 = __uint_as_float(r0.x); tr.y3 = __uint_as_float(r0.y); tr.a0 = __uint_as_float(r0.z); tr.b0 = __uint_as_float(r0.w);
                    tr.a1 = __uint_as_float(r1.x); tr.b1 = __uint_as_float(r1.y); tr.inv_area = __uint_as_float(r1.z);
                    tr.u1 = __uint_as_float(r2.y); tr.u2 = __uint_as_float(r2.z); tr.u3 = __uint_as_float(r2.w);
                    tr.v1 = __uint_as_float(r3.x); tr.v2 = __uint_as_float(r3.y); tr.v3 = __uint_as_float(r3.z);
                    tr.flags = r3.w;
                    tr.tw = r6.x; tr.th = r6.y; tr.toff = r6.z;
                    tr.iz1 = tr.iz2 = tr.iz3 = 0.0f;
                    float w0, w1, bcx, bcy, bcz;
                    if (!((slowmask >> j) & 1ull)) {
                        const float dx = (float)px - tr.x3, dy = (float)py - tr.y3;
                        w0 = tr.a0 * dx + tr.b0 * dy; w1 = tr.a1 * dx + tr.b1 * dy;
                    } else {
                        tr.min_x = r1.w & 0xFFFF; tr.min_y = r2.x & 0xFFFF;
                        tr.w0_start = __uint_as_float(r4.w); tr.w1_start = __uint_as_float(srec[j * SREC_Q + 5].x);
                        replay_w(tr, px, py, w0, w1);
                    }
                    (void)inside_bc(tr, w0, w1, bcx, bcy, bcz);          // (passed in (A): the barycentrics again)
                    if (!affine || (FMT8 && zmode)) { const uint4 r5 = srec[j * SREC_Q + 5]; tr.iz1 = __uint_as_float(r5.y); tr.iz2 = __uint_as_float(r5.z); tr.iz3 = __uint_as_float(r5.w); }
                    float shv[9];
                    if (shading != B32_SHADE_NONE) for (int q = 0; q < 9; ++q) shv[q] = a.shades[(size_t)r6.w * 9 + q];
                    uint32_t pix = tilebuf[ti];
                    float zb = (DEPTH_TILE && zmode) ? tilez[ti] : 0.0f;
                    if (blend_fragment<FMT8>(a, tr, bcx, bcy, bcz, px, py, r4.x, r4.y, r4.z, shading, shv, &pix, &zb, FMT8 ? zmode : 0 /* tested in (A) */, xray)) {
                        tilebuf[ti] = pix;
                        if (FMT8 && zmode) tilez[ti] = zb;           // the 8-bit path writes depth on every store
                        ++drawn;
                    }
                }
            }
        }
        __syncthreads();                                // records and masks are restaged for the next batch
    }
    for (uint32_t p = tid; p < TILE_W * TH; p += NT) {      // finished tile back, one 256-B row segment per wave instruction
        const uint32_t row = p >> 6, col = p & 63;
        const uint32_t px = x_lo + col, py = ty_top + row;
        if (px < x_hi && py >= y_lo && py < y_hi) {
            a.fb[(size_t)py * fp.width + px] = tilebuf[row * BT_STRIDE + col];
            if (FMT8 && zmode) a.zbuf[(size_t)py * fp.width + px] = tilez[row * BT_STRIDE + col];
        }
    }
    {
        for (int off = 32; off > 0; off >>= 1) drawn += __shfl_down(drawn, off);
        if (lane == 0) wf[wave] = drawn;
        __syncthreads();
        if (tid == 0) {
            unsigned long long t = 0;
            for (int w = 0; w < NW; ++w) t += wf[w];
            if (t) atomicAdd(&a.ctrl->fragments, t);
        }
    }
}

// hipFuncSetAttribute is per device: remember, per kernel instantiation, on which devices the large-LDS opt-in has been made
// (a process may own contexts on several GPUs)
static bool first_launch_on_device(bool (&done)[64]) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
    if (done[dev]) return false;
    done[dev] = true;
    return true;
}

template <bool FMT8, bool GATHER>
static void launch_blend(hipStream_t s, const FillArgs& a, uint32_t ntiles) {
    constexpr int NT = BLEND_NT;
    const bool zmode = a.fp.zmode && !a.fp.xray;
    const size_t lds = blend_lds_bytes(FMT8 && zmode);
    static bool attr[64] = {};
    if (first_launch_on_device(attr)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_blend<NT, FMT8, GATHER>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((k_blend<NT, FMT8, GATHER>), dim3(ntiles), dim3(NT), lds, s, a);
}

#ifdef B32_TIMELINE
static unsigned long long* g_timeline = nullptr;
extern "C" int b32_debug_timeline(unsigned long long* out, unsigned cap_words) {       // experiment builds only
    if (!g_timeline) return 0;
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(out, g_timeline, (size_t)cap_words * 8, hipMemcpyDeviceToHost);
    return 1;
}
#endif
template <bool EXACT, bool ZMODE, bool FMT8>
static void launch_p64(hipStream_t s, const FillArgs& a_in, uint32_t ntiles, int n_cu, bool wide) {
    FillArgs a = a_in;
    // tile planes, misc words, 64 words of repair queue per wave, then the staged index atlas (if any)
    const size_t atlas = a.atlas_idx_bytes ? (((size_t)ATLAS_CLUT_BYTES + a.atlas_idx_bytes + 15) & ~(size_t)15) : 0;
    const size_t lds_n = 4 * LDS_TILE_BYTES + LDS_MISC_BYTES + 8 * 256 + atlas, lds_w = 4 * LDS_TILE_BYTES + LDS_MISC_BYTES + 16 * 256 + atlas;
    static bool attr[64] = {};
    if (first_launch_on_device(attr)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_cover<0, EXACT, 512, ZMODE, FMT8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_cover<0, EXACT, 1024, ZMODE, FMT8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    const dim3 g(min(ntiles, (uint32_t)n_cu * 2));
#ifndef B32_STAGGER_TICKS
#define B32_STAGGER_TICKS 400
#endif
    // (two workgroups per CU and at least a second round of tiles: see `stagger` in k_cover)
    a.stagger = (a_in.stagger && !wide && ntiles > (uint32_t)n_cu * 2u) ? (uint32_t)B32_STAGGER_TICKS : 0u;
#ifdef B32_TIMELINE
    {
        static unsigned long long* dbg = nullptr;
        if (!dbg) (void)hipMalloc(reinterpret_cast<void**>(&dbg), (1 + 4 * 8192) * 8);
        (void)hipMemsetAsync(dbg, 0, 8, s);
        a.dbg = dbg;
        g_timeline = dbg;
    }
#endif
    const bool plain = a.fp.affine && a.fp.shading == B32_SHADE_NONE && a.fp.fixed_point && !a.fp.ortho && a.fp.nt == 1 && !a.inline_bin && !a.gather_blend;
#ifdef B32_EXP_LDS_ATLAS
    // experiment build (tools/exp_variants.py build atlas -DB32_EXP_LDS_ATLAS): the benchmark's frame through ONE 16-wave workgroup per CU
    // with the 64 KB index atlas + CLUT in LDS, against two 8-wave workgroups per CU fetching expanded texels through L1 / L2
    if (plain && !wide && a.atlas_idx_bytes) {
        static bool attr_x[64] = {};
        if (first_launch_on_device(attr_x))
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_cover<0, EXACT, 1024, ZMODE, FMT8, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL((k_cover<0, EXACT, 1024, ZMODE, FMT8, true, true>), dim3(min(ntiles, (uint32_t)n_cu)), dim3(1024), lds_w, s, a);
        return;
    }
#endif
#ifndef B32_CORUN_FORM
#define B32_CORUN_FORM 0         // the 96-VGPR co-resident form of the plain fill (round 4: four of its waves leave a SIMD room for two waves of the next
                                 // frame's setup kernel, +2 %).  OFF since round 5: with the straight-line plain shading the register cap costs 31 spilled
                                 // VGPRs and the pipelined frame 0.140 ms against 0.117 (profiles/r05_corun_form_ab.txt)
#endif
#if B32_CORUN_FORM
    if (plain && !wide && a.co_run && !EXACT && !ZMODE && !FMT8) {
        static bool attr_co[64] = {};
        if (first_launch_on_device(attr_co))
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_cover<0, false, 512, false, false, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL((k_cover<0, false, 512, false, false, true, 2>), g, dim3(512), lds_n, s, a);
        return;
    }
#endif
    if (plain && !wide) {
        static bool attr_plain[64] = {};
        if (first_launch_on_device(attr_plain))
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_cover<0, EXACT, 512, ZMODE, FMT8, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL((k_cover<0, EXACT, 512, ZMODE, FMT8, true, true>), g, dim3(512), lds_n, s, a);
    } else if (wide) hipLaunchKernelGGL((k_cover<0, EXACT, 1024, ZMODE, FMT8, true>), g, dim3(1024), lds_w, s, a);
    else hipLaunchKernelGGL((k_cover<0, EXACT, 512, ZMODE, FMT8, true>), g, dim3(512), lds_n, s, a);
}

void launch_fill(hipStream_t s, const FillArgs& a, int n_cu, hipEvent_t after_cover) {
    const uint32_t ntiles = a.fp.tiles_x * a.fp.tiles_y;
    if (ntiles == 0) return;
    if (a.skip_solid) { if (after_cover) (void)hipEventRecord(after_cover, s); return; }   // wireframe_overlay: nothing solid is drawn (render.rs:2550)
    const bool f8 = a.fp.fmt8 != 0;
    if (a.ordered_all) {                                         // no overwrite pass at all: everything goes through the ordered walk
        if (after_cover) (void)hipEventRecord(after_cover, s);
        if (f8) launch_blend<true, false>(s, a, ntiles);
        else launch_blend<false, false>(s, a, ntiles);
        return;
    }
    const size_t lds_sort = LDS_TEX_OFFSET + LDS_SORT_CNT_BYTES + 2048;
    if (a.prio64) {   // sort-free fused kernel: coverage and shading are one launch; 64-bit tile buffers (2 x 36 KB)
        // EXACT = texel rule per fragment (textures with many skippable texels, or exact store counting); z-buffer mode = the
        // priority's high word is the fragment depth.  Few tiles (narrow multi-GPU bands, small frames): 16 waves per tile.
        // (16-wave workgroups only while every tile can have a CU of its own: with more tiles than CUs the 8-wave form, two workgroups per
        // CU, is faster -- a 960-row band of C3, 600 tiles: 0.112 -> 0.086 ms; 480 rows: 0.086 -> 0.067; 240 rows: 0.067 -> 0.057)
        const bool wide = ntiles <= (uint32_t)n_cu && !a.narrow_only;
        const int sel = (a.exact_coverage ? 4 : 0) | (a.fp.zmode ? 2 : 0) | (f8 ? 1 : 0);
        switch (sel) {
            case 0: launch_p64<false, false, false>(s, a, ntiles, n_cu, wide); break;
            case 1: launch_p64<false, false, true>(s, a, ntiles, n_cu, wide); break;
            case 2: launch_p64<false, true, false>(s, a, ntiles, n_cu, wide); break;
            case 3: launch_p64<false, true, true>(s, a, ntiles, n_cu, wide); break;
            case 4: launch_p64<true, false, false>(s, a, ntiles, n_cu, wide); break;
            case 5: launch_p64<true, false, true>(s, a, ntiles, n_cu, wide); break;
            case 6: launch_p64<true, true, false>(s, a, ntiles, n_cu, wide); break;
            default: launch_p64<true, true, true>(s, a, ntiles, n_cu, wide); break;
        }
        if (after_cover) (void)hipEventRecord(after_cover, s);
        if (a.gather_blend) launch_blend<false, true>(s, a, ntiles);
        return;
    } else if (a.fp.zmode) {
        if (f8) hipLaunchKernelGGL((k_cover<0, true, 512, true, true>), dim3(min(ntiles, (uint32_t)n_cu * 3)), dim3(512), lds_sort, s, a);
        else hipLaunchKernelGGL((k_cover<0, true, 512, true, false>), dim3(min(ntiles, (uint32_t)n_cu * 3)), dim3(512), lds_sort, s, a);
    } else if (a.exact_coverage) {
        if (f8) {
            hipLaunchKernelGGL((k_cover<0, true, 512, false, true>), dim3(min(ntiles, (uint32_t)n_cu * 2)), dim3(512), LDS_TEX_OFFSET, s, a);
        } else if (a.lds_tex_texels) {     // texture sampled once per fragment: stage it in LDS, one 16-wave workgroup per CU
            const size_t lds = LDS_TEX_OFFSET + (((size_t)a.lds_tex_texels * 2 + 15) & ~(size_t)15);
            static bool attr_set[64] = {};
            if (first_launch_on_device(attr_set)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_cover<1, true, 1024, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipLaunchKernelGGL((k_cover<1, true, 1024, false, false>), dim3(min(ntiles, (uint32_t)n_cu)), dim3(1024), lds, s, a);
        } else {
            hipLaunchKernelGGL((k_cover<0, true, 512, false, false>), dim3(min(ntiles, (uint32_t)n_cu * 2)), dim3(512), LDS_TEX_OFFSET, s, a);
        }
    } else {    // CHEAP coverage never samples a texture: one kernel for both pixel formats
        hipLaunchKernelGGL((k_cover<0, false, 512, false, false>), dim3(min(ntiles, (uint32_t)n_cu * 3)), dim3(512), lds_sort, s, a);
    }
    if (after_cover) (void)hipEventRecord(after_cover, s);
    const uint32_t band_h = a.fp.band_y1 - a.fp.band_y0;
    if (band_h) {
        const dim3 g(((ntiles + 7) / 8) * 8 * 4);
        if (f8) hipLaunchKernelGGL((k_shade<true>), g, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((k_shade<false>), g, dim3(256), 0, s, a);
    }
    if (a.may_blend && !f8) launch_blend<false, false>(s, a, ntiles);
}

size_t fill_lds_tex_budget() { return 160 * 1024 - LDS_TEX_OFFSET - 16; }
uint32_t fill_lds_atlas_room(bool wide) {
    // 160 KB of LDS per CU, allocated in 512-byte granules: one 16-wave workgroup, or two 8-wave workgroups side by side
    const uint32_t total = 160u * 1024u, gran = 512u;
    if (wide) return total - (uint32_t)(4 * LDS_TILE_BYTES + LDS_MISC_BYTES + 16 * 256) - gran;
    return total / 2u - (uint32_t)(4 * LDS_TILE_BYTES + LDS_MISC_BYTES + 8 * 256) - gran;
}

}  // namespace b32

// b32_fill_common.h -- device helpers shared by the fill kernels (b32_fill.hip: the fused coverage + shading kernel and the keyed k_cover;
// b32_shade.hip: k_shade of the keyed pipelines; b32_blend.hip: the ordered transparent pass): texture sampling and the transparency
// rule (types.rs:671-681, render.rs:1563-1607), the colour pipelines of both pixel formats (render.rs:1613-1661, 1355-1387), the pixel
// stores (render.rs:445-502, 567-628), the per-surface record views and the per-pixel hit test the shading phases and repair paths use.
// Bit-exactness: barycentrics use the reference's expression order; the edge functions are evaluated from exact integers only for
// surfaces k_setup proved exact (integer coordinates, every intermediate < 2^24), otherwise the incremental walk (render.rs:1706-1712)
// is replayed literally.
#pragma once
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
__device__ inline float exact_depth_at(const FillArgs& a, uint32_t sid, uint32_t px, uint32_t py) {
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

// ---- wave-level helpers (prefix scans over the 64 lanes, ds_bpermute)
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

// hipFuncSetAttribute is per device: remember, per kernel instantiation, on which devices the large-LDS opt-in has been made
// (a process may own contexts on several GPUs)
inline bool first_launch_on_device(bool (&done)[64]) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
    if (done[dev]) return false;
    done[dev] = true;
    return true;
}
// launchers of the kernels that live in units of their own (b32_shade.hip, b32_blend.hip)
void launch_shade(hipStream_t s, const FillArgs& a, uint32_t ntiles);
void launch_blend(hipStream_t s, const FillArgs& a, uint32_t ntiles, bool fmt8, bool gather);

}  // namespace b32

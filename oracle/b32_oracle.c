/*
 * b32_oracle.c — CPU restatement of bonnie-32's `render_mesh_15` hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this.  The product library (bonnie-32_amd/csrc) never links, calls or falls back to it.
 *
 * PARITY STATUS: pinned by the reference's own compiled code for everything the 8-bit-colour and the RGB555 path share, "parity
 * unpinned" by the reference for the rest.  The reference is Rust (no toolchain here) and holds no test that pins a pixel (SURVEY §4,
 * §8c); but its tree carries docs/bonnie-engine.wasm, an older build of the crate, which node runs: tests/golden/wasm_pin/ holds 14
 * frames drawn by THAT module's render_mesh and 9 523 values of its acosf, and b32o_render_mesh / b32o_acosf reproduce every one of
 * them bit for bit (tests/test_wasm_pin.py; README.md there lists what the old build and today's source define identically: camera
 * transform, float / ortho projection, culling, bounding boxes, triangle setup, edge walk, inside test, affine UVs, Texture::sample,
 * vertex colours, modulation, lighting incl. spot lights, blended stores, wireframe overlay -- the code render_mesh_15 shares).
 * NOT reachable through that module, hence still unpinned by the reference: the fixed-point snap (fixed.rs), the RGB555 tail
 * (Color15, dither_and_quantize, blend_rgb555), 1/z depth, fog, equal sort keys.  What pins those: (1) the reference's own unit-test
 * vectors for fixed.rs/math.rs (tests/test_oracle_kats.py), (2) the psx-spx UNR table and dither matrix the reference's comments name,
 * (3) an independent numpy restatement (oracle/np_model.py) that must agree bit-for-bit on whole frames.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -msse2 -mfpmath=sse (scalar SSE2, no FMA — the
 * reference's release profile is x86-64 baseline, Cargo.toml:52-54).
 *
 * Rust semantics encoded once in the helpers below:
 *   - `f as i32 / as u8 / as usize` saturate and map NaN to 0 (Rust reference: "Numeric cast").
 *   - f32::min/max ignore a NaN operand; f32::clamp propagates NaN.
 *   - release build: integer overflow wraps (`-x`, `<<`, `abs`), exactly where the source does not say wrapping_*.
 *   - slice::sort_by is stable; Iterator::partition keeps relative order.
 */
#include "../include/b32raster.h"
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------ the reference's numeric literals, by name
 * The code below uses the algorithm's literals only through these names; b32o_constant(key) hands each one back and
 * tests/test_oracle_kats.py::test_oracle_constants_are_the_reference_text compares it with tests/golden/ref_constants.json, which
 * tests/golden/pin_constants.py derives from the reference's own text (every row of K_TABLE gives the fixture key). */
#define K_FRAC_BITS 12                   /* fixed.rs:110 */
#define K_ONE_F ((float)(1 << K_FRAC_BITS))
#define K_UNR_ENTRIES 257                /* fixed.rs:20-31 */
#define K_UNR_INDEX_OFFSET 256
#define K_UNR_NUMERATOR 262144u
#define K_UNR_ROUND_ADD 1
#define K_UNR_ROUND_DIV 2
#define K_UNR_SUBTRACT 257
#define K_DIV_D16_SHIFT 16               /* fixed.rs:197-212 */
#define K_DIV_INDEX_BIAS 0x7FC0ull
#define K_DIV_INDEX_SHIFT 7
#define K_DIV_INDEX_MAX 256
#define K_DIV_U_ADD 0x101
#define K_DIV_NR1_CONST 0x2000080ull
#define K_DIV_NR1_SHIFT 8
#define K_DIV_NR2_CONST 0x80ull
#define K_DIV_NR2_SHIFT 8
#define K_DIV_SHIFT_BASE 36u
#define K_PF_DISTANCE 5.0f               /* project_to_screen, fixed.rs:396-406 */
#define K_PF_SCALE 4.0f
#define K_PF_VIEWPORT_DIV 2.0f
#define K_PF_VIEWPORT_FRAC 0.75f
#define K_PF_DENOM_GUARD 256
#define K_P_DISTANCE 5.0f                /* project, math.rs:118-127 */
#define K_P_US_SUB 1.0f
#define K_P_VIEWPORT_DIV 2.0f
#define K_P_VIEWPORT_FRAC 0.75f
#define K_P_DENOM_GUARD 0.001f
#define K_NEAR_PLANE 0.1f                /* math.rs:155 */
#define K_MESH_DISTANCE 5.0f             /* render.rs:2344 */
#define K_AREA_EPS 0.00001f              /* render.rs:1501 */
#define K_ERR (-0.0001f)                 /* render.rs:1541 */
#define K_AREA_EPS8 0.00001f             /* render.rs:1258 (8-bit fill) */
#define K_ERR8 (-0.0001f)                /* render.rs:1302 */
#define K_MOD_DIV 128                    /* render.rs:1624 */
#define K_MOD_MAX 255
#define K_SHADE_LO 0.0f                  /* render.rs:1643 */
#define K_SHADE_HI 2.0f
#define K_SHADE_MAX 255.0f
#define K_NODITHER_SHIFT 3               /* render.rs:1653 */
#define K_DITHER_SHIFT 3                 /* render.rs:1177 */
#define K_DITHER_LO 0
#define K_DITHER_HI 31
#define K_DITHER8_SHIFT 3                /* apply_dither, render.rs:1191-1196 */
#define K_DITHER8_HI 31
#define K_DITHER8_EXPAND_SHIFT 3
#define K_EXPAND5_SHL 3                  /* render.rs:1162 */
#define K_EXPAND5_SHR 2
#define K_BLEND_IN_SHIFT 3               /* blend_rgb555, render.rs:1095-1144 */
#define K_BLEND_AVG_DIV 2
#define K_BLEND_HI 31
#define K_BLEND_LO 0
#define K_BLEND_QUARTER_DIV 4
#define K_BLEND_OUT_SHIFT 3
#define K_LIGHT_MIN_DIST 0.001f          /* render.rs:1030 */
#define K_LIGHT_COLOR_DIV 255.0f         /* render.rs:1062 */
#define K_LIGHT_TOTAL_MAX 1.0f           /* render.rs:1070 */
#define K_C15_TRANSPARENT 0x0000         /* types.rs:24-53 */
#define K_C15_BLACK_DRAWABLE 0x8000
#define K_C15_WHITE 0x7FFF
#define K_C15_SEMI_BIT 0x8000
#define K_C15_R_SHIFT 10
#define K_C15_G_SHIFT 5
#define K_C15_CHANNEL_MAX 31
#define K_TABLE(X) \
    X("fixed.frac_bits", K_FRAC_BITS) \
    X("unr.entries", K_UNR_ENTRIES) X("unr.index_offset", K_UNR_INDEX_OFFSET) X("unr.numerator", K_UNR_NUMERATOR) \
    X("unr.round_add", K_UNR_ROUND_ADD) X("unr.round_div", K_UNR_ROUND_DIV) X("unr.subtract", K_UNR_SUBTRACT) \
    X("div_unr.d16_shift", K_DIV_D16_SHIFT) X("div_unr.index_bias", K_DIV_INDEX_BIAS) X("div_unr.index_shift", K_DIV_INDEX_SHIFT) \
    X("div_unr.index_max", K_DIV_INDEX_MAX) X("div_unr.u_add", K_DIV_U_ADD) X("div_unr.nr1_const", K_DIV_NR1_CONST) \
    X("div_unr.nr1_shift", K_DIV_NR1_SHIFT) X("div_unr.nr2_const", K_DIV_NR2_CONST) X("div_unr.nr2_shift", K_DIV_NR2_SHIFT) \
    X("div_unr.shift_base", K_DIV_SHIFT_BASE) \
    X("project_fixed.distance", K_PF_DISTANCE) X("project_fixed.scale", K_PF_SCALE) X("project_fixed.viewport_div", K_PF_VIEWPORT_DIV) \
    X("project_fixed.viewport_frac", K_PF_VIEWPORT_FRAC) X("project_fixed.denom_guard", K_PF_DENOM_GUARD) \
    X("project.distance", K_P_DISTANCE) X("project.us_sub", K_P_US_SUB) X("project.viewport_div", K_P_VIEWPORT_DIV) \
    X("project.viewport_frac", K_P_VIEWPORT_FRAC) X("project.denom_guard", K_P_DENOM_GUARD) \
    X("near_plane", K_NEAR_PLANE) X("mesh.distance", K_MESH_DISTANCE) \
    X("fill.area_eps", K_AREA_EPS) X("fill.err", K_ERR) X("fill8.area_eps", K_AREA_EPS8) X("fill8.err", K_ERR8) \
    X("fill.modulate_div", K_MOD_DIV) X("fill.modulate_max", K_MOD_MAX) \
    X("fill.shade_clamp_lo", K_SHADE_LO) X("fill.shade_clamp_hi", K_SHADE_HI) X("fill.shade_max", K_SHADE_MAX) \
    X("fill.nodither_shift", K_NODITHER_SHIFT) \
    X("dither.shift", K_DITHER_SHIFT) X("dither.clamp_lo", K_DITHER_LO) X("dither.clamp_hi", K_DITHER_HI) \
    X("dither8.shift", K_DITHER8_SHIFT) X("dither8.clamp_hi", K_DITHER8_HI) X("dither8.expand_shift", K_DITHER8_EXPAND_SHIFT) \
    X("expand5.shl", K_EXPAND5_SHL) X("expand5.shr", K_EXPAND5_SHR) \
    X("blend555.in_shift", K_BLEND_IN_SHIFT) X("blend555.average_div", K_BLEND_AVG_DIV) X("blend555.clamp_hi", K_BLEND_HI) \
    X("blend555.clamp_lo", K_BLEND_LO) X("blend555.quarter_div", K_BLEND_QUARTER_DIV) X("blend555.out_shift", K_BLEND_OUT_SHIFT) \
    X("light.min_dist", K_LIGHT_MIN_DIST) X("light.color_div", K_LIGHT_COLOR_DIV) X("light.total_max", K_LIGHT_TOTAL_MAX) \
    X("color15.transparent", K_C15_TRANSPARENT) X("color15.black_drawable", K_C15_BLACK_DRAWABLE) X("color15.white", K_C15_WHITE) \
    X("color15.semi_bit", K_C15_SEMI_BIT) X("color15.r_shift", K_C15_R_SHIFT) X("color15.g_shift", K_C15_G_SHIFT) \
    X("color15.channel_max", K_C15_CHANNEL_MAX)
/* value of the named literal as a double (every literal here is exactly representable: f32 constants widen exactly); NaN = no such key */
EXPORT double b32o_constant(const char* key) {
#define K_ROW(name, v) if (strcmp(key, name) == 0) return (double)(v);
    K_TABLE(K_ROW)
#undef K_ROW
    return NAN;
}
EXPORT uint32_t b32o_constant_count(void) {
    uint32_t n = 0;
#define K_ROW(name, v) ++n;
    K_TABLE(K_ROW)
#undef K_ROW
    return n;
}
EXPORT const char* b32o_constant_name(uint32_t i) {
    uint32_t n = 0;
#define K_ROW(name, v) if (n++ == i) return name;
    K_TABLE(K_ROW)
#undef K_ROW
    return 0;
}

/* Row band of the all-cores CPU baseline (bench.py: N processes, each draws only its own rows of the same frame): triangle rows
 * outside [g_band_y0, g_band_y1) keep the reference's row-to-row accumulation (w_row += b) but skip their pixels, so the rows that
 * ARE drawn hold exactly the values of the full-frame walk.  Default: the whole frame. */
static uint32_t g_band_y0 = 0, g_band_y1 = 0xFFFFFFFFu;
EXPORT void b32o_set_row_band(uint32_t y0, uint32_t y1) { g_band_y0 = y0; g_band_y1 = y1; }
/* Threads of the all-cores CPU baseline inside ONE process (bench.py cpu_all_cores): the per-vertex transform is split by vertex
 * range, cull / setup / sort run once, and the draw phase gives every thread a band of rows (same mechanism as above, per thread), all
 * threads walking the one sorted surface list.  1 = the reference's own single-threaded schedule (default, and what the checker uses). */
static int g_threads = 1;
EXPORT void b32o_set_threads(int n) { g_threads = n < 1 ? 1 : (n > 256 ? 256 : n); }

/* ------------------------------------------------------------------ Rust cast / float helpers */
static inline int32_t f2i32_sat(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return INT32_MAX;
    if (f <= -2147483648.0f) return INT32_MIN;
    return (int32_t)f;
}
static inline uint8_t f2u8_sat(float f) {
    if (f != f) return 0;
    if (f >= 255.0f) return 255;
    if (f <= 0.0f) return 0;
    return (uint8_t)f;
}
static inline uint64_t f2usize_sat(float f) {
    if (f != f) return 0;
    if (f <= 0.0f) return 0;
    if (f >= 18446744073709551616.0f) return UINT64_MAX;
    return (uint64_t)f;
}
static inline float rmin(float a, float b) { /* f32::min */
    if (a != a) return b;
    if (b != b) return a;
    return a < b ? a : b;
}
static inline float rmax(float a, float b) { /* f32::max */
    if (a != a) return b;
    if (b != b) return a;
    return a > b ? a : b;
}
static inline float rclamp(float x, float lo, float hi) { /* f32::clamp */
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}
static inline float rem_euclid1(float x) { /* f32::rem_euclid(1.0): r = x % 1.0; if r < 0 { r + 1.0 } */
    float r = fmodf(x, 1.0f);
    return r < 0.0f ? r + 1.0f : r;
}
static inline int32_t wrap_add(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t wrap_sub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static inline int32_t wrap_neg(int32_t a) { return (int32_t)(0u - (uint32_t)a); }

/* ------------------------------------------------------------------ fixed.rs */

/* UNR_TABLE, fixed.rs:20-31 */
static uint8_t g_unr[257];
static int g_unr_ready = 0;
static void unr_init(void) {
    if (g_unr_ready) return;
    for (uint32_t i = 0; i < K_UNR_ENTRIES; ++i) {
        uint32_t div = i + K_UNR_INDEX_OFFSET;
        uint32_t quotient = K_UNR_NUMERATOR / div;
        int32_t val = (int32_t)((quotient + K_UNR_ROUND_ADD) / K_UNR_ROUND_DIV) - K_UNR_SUBTRACT;
        g_unr[i] = val > 0 ? (uint8_t)val : 0;
    }
    g_unr_ready = 1;
}
EXPORT uint8_t b32o_unr_table(uint32_t i) { unr_init(); return g_unr[i <= 256 ? i : 256]; }

/* Fixed32::from_f32, fixed.rs:125-127 */
EXPORT int32_t b32o_fixed_from_f32(float f) { return f2i32_sat(f * K_ONE_F); }
/* Fixed32::from_int, fixed.rs:119-121 (release: shift wraps) */
static inline int32_t fixed_from_int(int32_t n) { return (int32_t)((uint32_t)n << K_FRAC_BITS); }
/* Fixed32::to_f32, fixed.rs:131-133 */
static inline float fixed_to_f32(int32_t v) { return (float)v / K_ONE_F; }
/* Fixed32::mul_fixed, fixed.rs:161-165 */
EXPORT int32_t b32o_fixed_mul(int32_t a, int32_t b) {
    int64_t r = ((int64_t)a * (int64_t)b) >> K_FRAC_BITS;
    return (int32_t)r;
}
/* Fixed32::div_unr, fixed.rs:178-230 */
EXPORT int32_t b32o_fixed_div_unr(int32_t self, int32_t divisor) {
    unr_init();
    if (divisor == 0) return 0;
    int result_negative = (self < 0) != (divisor < 0);
    uint64_t num = (uint64_t)(self < 0 ? (0u - (uint32_t)self) : (uint32_t)self);     /* unsigned_abs */
    uint32_t den = divisor < 0 ? (0u - (uint32_t)divisor) : (uint32_t)divisor;
    if (den == 0) return 0;
    uint32_t z = (uint32_t)__builtin_clz(den);
    uint64_t d_norm = (uint64_t)den << z;
    uint64_t d16 = d_norm >> K_DIV_D16_SHIFT;
    uint64_t ti = (d16 - K_DIV_INDEX_BIAS) >> K_DIV_INDEX_SHIFT;                      /* wrapping_sub */
    if (ti > K_DIV_INDEX_MAX) ti = K_DIV_INDEX_MAX;
    uint64_t u_val = (uint64_t)g_unr[ti] + K_DIV_U_ADD;
    uint64_t nr1 = (K_DIV_NR1_CONST - d16 * u_val) >> K_DIV_NR1_SHIFT;
    uint64_t nr2 = (K_DIV_NR2_CONST + nr1 * u_val) >> K_DIV_NR2_SHIFT;
    uint64_t raw = num * nr2;
    uint32_t shift = K_DIV_SHIFT_BASE - z;
    uint64_t magnitude;
    if (shift < 64) {
        uint64_t rounding = shift > 0 ? (1ull << (shift - 1)) : 0;
        magnitude = (raw + rounding) >> shift;
    } else {
        magnitude = 0;
    }
    int32_t clamped = (int32_t)(magnitude < (uint64_t)INT32_MAX ? magnitude : (uint64_t)INT32_MAX);
    return result_negative ? -clamped : clamped;
}

typedef struct { int32_t x, y, z; } FixedVec3;
/* FixedVec3::from_vec3, fixed.rs:291-297 */
static inline FixedVec3 fv_from(const float v[3]) {
    FixedVec3 r = { b32o_fixed_from_f32(v[0]), b32o_fixed_from_f32(v[1]), b32o_fixed_from_f32(v[2]) };
    return r;
}
/* FixedVec3::dot, fixed.rs:311-313: x*ox + y*oy + z*oz with wrapping adds (fixed.rs:236-238) */
static inline int32_t fv_dot(FixedVec3 a, FixedVec3 b) {
    return wrap_add(wrap_add(b32o_fixed_mul(a.x, b.x), b32o_fixed_mul(a.y, b.y)), b32o_fixed_mul(a.z, b.z));
}

/* project_fixed, fixed.rs:424-441 = transform_to_camera_space :362-381 + project_to_screen :390-420 */
EXPORT void b32o_project_fixed(const float world_pos[3], const float camera_pos[3],
                               const float basis_x[3], const float basis_y[3], const float basis_z[3],
                               uint32_t width, uint32_t height, int32_t* sx, int32_t* sy, float* depth) {
    FixedVec3 wp = fv_from(world_pos), cp = fv_from(camera_pos);
    FixedVec3 rel = { wrap_sub(wp.x, cp.x), wrap_sub(wp.y, cp.y), wrap_sub(wp.z, cp.z) };
    FixedVec3 bx = fv_from(basis_x), by = fv_from(basis_y), bz = fv_from(basis_z);
    FixedVec3 cam = { fv_dot(rel, bx), fv_dot(rel, by), fv_dot(rel, bz) };

    int32_t distance = b32o_fixed_from_f32(K_PF_DISTANCE);
    int32_t scale = b32o_fixed_from_f32(K_PF_SCALE);
    uint32_t mn = width < height ? width : height;
    int32_t viewport_scale = b32o_fixed_from_f32(((float)mn / K_PF_VIEWPORT_DIV) * K_PF_VIEWPORT_FRAC);
    int32_t half_w = fixed_from_int((int32_t)width / 2);
    int32_t half_h = fixed_from_int((int32_t)height / 2);
    int32_t denom = wrap_add(cam.z, distance);
    /* i32::abs wraps for i32::MIN in release -> stays negative -> "< 256" is true */
    int32_t adenom = denom < 0 ? wrap_neg(denom) : denom;
    if (adenom < K_PF_DENOM_GUARD) {
        *sx = half_w >> K_FRAC_BITS; *sy = half_h >> K_FRAC_BITS; *depth = fixed_to_f32(cam.z);
        return;
    }
    int32_t proj_x = b32o_fixed_div_unr(b32o_fixed_mul(cam.x, scale), denom);
    int32_t proj_y = b32o_fixed_div_unr(b32o_fixed_mul(cam.y, scale), denom);
    int32_t screen_x = wrap_add(b32o_fixed_mul(proj_x, viewport_scale), half_w);
    int32_t screen_y = wrap_add(b32o_fixed_mul(proj_y, viewport_scale), half_h);
    *sx = screen_x >> K_FRAC_BITS; *sy = screen_y >> K_FRAC_BITS; *depth = fixed_to_f32(cam.z);
}

/* ------------------------------------------------------------------ math.rs */
typedef struct { float x, y, z; } V3;
static inline V3 v3(float x, float y, float z) { V3 r = { x, y, z }; return r; }
static inline V3 v3p(const float p[3]) { V3 r = { p[0], p[1], p[2] }; return r; }
/* Vec3::dot, math.rs:23-25: (x*ox + y*oy) + z*oz */
static inline float v3dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V3 v3sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }     /* math.rs:71-79 */
static inline V3 v3add(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }     /* math.rs:60-69 */
static inline V3 v3scale(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }      /* math.rs:51-57 */
static inline float v3len(V3 a) { return sqrtf(v3dot(a, a)); }                          /* math.rs:35-37 */
static inline V3 v3normalize(V3 a) {                                                    /* math.rs:39-49 */
    float l = v3len(a);
    if (l == 0.0f) return v3(0, 0, 0);
    return v3(a.x / l, a.y / l, a.z / l);
}
/* perspective_transform, math.rs:103-109 */
static inline V3 perspective_transform(V3 v, V3 cx, V3 cy, V3 cz) {
    return v3(v3dot(v, cx), v3dot(v, cy), v3dot(v, cz));
}
/* project, math.rs:117-136 */
static inline V3 project_float(V3 v, uint32_t width, uint32_t height) {
    const float ud = K_P_DISTANCE;
    float us = ud - K_P_US_SUB;
    uint32_t mn = width < height ? width : height;
    float vs = ((float)mn / K_P_VIEWPORT_DIV) * K_P_VIEWPORT_FRAC;
    float denom = v.z + ud;
    if (fabsf(denom) < K_P_DENOM_GUARD) return v3((float)width / 2.0f, (float)height / 2.0f, v.z);
    return v3((v.x * us) / denom * vs + ((float)width / 2.0f),
              (v.y * us) / denom * vs + ((float)height / 2.0f),
              denom);
}
/* project_ortho, math.rs:140-148 */
static inline V3 project_ortho(V3 v, float zoom, float cx, float cy, uint32_t width, uint32_t height) {
    return v3((v.x - cx) * zoom + ((float)width / 2.0f),
              -(v.y - cy) * zoom + ((float)height / 2.0f),
              v.z);
}

/* ------------------------------------------------------------------ types.rs Color15 / Texture15 */
static inline uint8_t c15_r5(uint16_t c) { return (uint8_t)((c >> K_C15_R_SHIFT) & 0x1F); }   /* types.rs:119-121 */
static inline uint8_t c15_g5(uint16_t c) { return (uint8_t)((c >> K_C15_G_SHIFT) & 0x1F); }    /* types.rs:125-127 */
static inline uint8_t c15_b5(uint16_t c) { return (uint8_t)(c & 0x1F); }           /* types.rs:131-133 */
/* expand_5_to_8, render.rs:1161-1163 == Color15::r8, types.rs:137-141 */
static inline uint8_t expand_5_to_8(uint8_t v5) { return (uint8_t)((v5 << K_EXPAND5_SHL) | (v5 >> K_EXPAND5_SHR)); }
/* Color15::new_semi, types.rs:41-56 */
static inline uint16_t c15_new_semi(uint8_t r, uint8_t g, uint8_t b, int semi) {
    const uint8_t mx = K_C15_CHANNEL_MAX;
    uint16_t c = (uint16_t)(((uint16_t)(r > mx ? mx : r) << K_C15_R_SHIFT) | ((uint16_t)(g > mx ? mx : g) << K_C15_G_SHIFT) | (uint16_t)(b > mx ? mx : b));
    if (semi) c |= K_C15_SEMI_BIT;
    return c;
}
/* Color15::to_rgba, types.rs:220-226 */
EXPORT void b32o_color15_to_rgba(uint16_t c, uint8_t out[4]) {
    if (c == 0) { out[0] = out[1] = out[2] = out[3] = 0; return; }
    out[0] = expand_5_to_8(c15_r5(c)); out[1] = expand_5_to_8(c15_g5(c)); out[2] = expand_5_to_8(c15_b5(c)); out[3] = 255;
}
/* Texture15::sample, types.rs:671-681 */
EXPORT uint16_t b32o_texture15_sample(const uint16_t* pixels, uint32_t width, uint32_t height, float u, float v) {
    if (width == 0 || height == 0 || pixels == NULL) return 0;
    float uw = rem_euclid1(u), vw = rem_euclid1(v);
    uint64_t tx = f2usize_sat(uw * (float)width);  if (tx > width - 1) tx = width - 1;
    uint64_t ty = f2usize_sat(vw * (float)height); if (ty > height - 1) ty = height - 1;
    return pixels[ty * width + tx];
}
/* Clut::lookup, types.rs:390-397 + IndexedAtlas::to_texture15, modeler/mesh_editor.rs:669-682 */
EXPORT void b32o_expand_indexed(const uint8_t* indices, uint32_t n, const uint16_t* clut, uint32_t clut_len, uint16_t* out) {
    for (uint32_t i = 0; i < n; ++i) out[i] = indices[i] < clut_len ? clut[indices[i]] : 0;
}

/* ------------------------------------------------------------------ render.rs colour helpers */
/* blend_rgb555, render.rs:1093-1145 */
EXPORT void b32o_blend_rgb555(uint8_t fr, uint8_t fg, uint8_t fb, uint8_t br, uint8_t bg, uint8_t bb, uint32_t mode, uint8_t out[3]) {
    uint8_t f5[3] = { (uint8_t)(fr >> K_BLEND_IN_SHIFT), (uint8_t)(fg >> K_BLEND_IN_SHIFT), (uint8_t)(fb >> K_BLEND_IN_SHIFT) };
    uint8_t b5[3] = { (uint8_t)(br >> K_BLEND_IN_SHIFT), (uint8_t)(bg >> K_BLEND_IN_SHIFT), (uint8_t)(bb >> K_BLEND_IN_SHIFT) };
    for (int i = 0; i < 3; ++i) {
        uint8_t r5;
        switch (mode) {
            default:
            case B32_BLEND_OPAQUE:      r5 = f5[i]; break;
            case B32_BLEND_AVERAGE:     { uint16_t s = (uint16_t)((b5[i] + f5[i]) / K_BLEND_AVG_DIV); r5 = (uint8_t)(s > K_BLEND_HI ? K_BLEND_HI : s); } break;
            case B32_BLEND_ADD:         { uint16_t s = (uint16_t)(b5[i] + f5[i]); r5 = (uint8_t)(s > 31 ? 31 : s); } break;
            case B32_BLEND_SUBTRACT:    { int16_t s = (int16_t)(b5[i] - f5[i]); r5 = (uint8_t)(s < 0 ? 0 : s); } break;
            case B32_BLEND_ADD_QUARTER: { uint16_t s = (uint16_t)(b5[i] + f5[i] / K_BLEND_QUARTER_DIV); r5 = (uint8_t)(s > K_BLEND_HI ? K_BLEND_HI : s); } break;
            case B32_BLEND_ERASE:       r5 = b5[i]; break;
        }
        out[i] = (uint8_t)(r5 << K_BLEND_OUT_SHIFT);
    }
}
/* PS1_DITHER_MATRIX, render.rs:1150-1155 */
static const int8_t PS1_DITHER_MATRIX[4][4] = {
    { -4,  0, -3,  1 },
    {  2, -2,  3, -1 },
    { -3,  1, -4,  0 },
    {  3, -1,  2, -2 },
};
EXPORT int32_t b32o_dither_offset(uint32_t x, uint32_t y) { return PS1_DITHER_MATRIX[y & 3][x & 3]; }
/* dither_and_quantize, render.rs:1173-1182 */
EXPORT void b32o_dither_and_quantize(uint8_t r8, uint8_t g8, uint8_t b8, uint32_t x, uint32_t y, uint8_t out5[3]) {
    int32_t off = PS1_DITHER_MATRIX[y & 3][x & 3];
    int32_t c[3] = { r8, g8, b8 };
    for (int i = 0; i < 3; ++i) {
        int32_t q = (c[i] + off) >> K_DITHER_SHIFT;
        out5[i] = (uint8_t)(q < K_DITHER_LO ? K_DITHER_LO : (q > K_DITHER_HI ? K_DITHER_HI : q));
    }
}

/* ------------------------------------------------------------------ f32::acos (spot lights, render.rs:1047)
 * Rust's f32::acos is the target's libm acosf: on wasm32 (the console's shipping target, the docs/ .wasm artefact) the `libm` crate, a port
 * of musl's src/math/acosf.c (FreeBSD msun e_acosf.c); on Linux glibc's.  They agree to within 1 ulp, not bit for bit, so the
 * reference itself is not bit-portable here.  This is the published musl / msun algorithm (plain f32 arithmetic, < 1 ulp), the
 * same on the GPU (b32_setup.hip) and in oracle/np_model.py; constants as published (pio2_hi = 0x3fc90fda, pio2_lo = 0x33a22168). */
static float acosf_R(float z) {
    const float pS0 = 1.6666586697e-01f, pS1 = -4.2743422091e-02f, pS2 = -8.6563630030e-03f, qS1 = -7.0662963390e-01f;
    float p = z * (pS0 + z * (pS1 + z * pS2));
    float q = 1.0f + z * qS1;
    return p / q;
}
EXPORT float b32o_acosf(float x) {
    const float pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f;
    uint32_t hx, ix;
    memcpy(&hx, &x, 4);
    ix = hx & 0x7fffffffu;
    if (ix >= 0x3f800000u) {                          /* |x| >= 1 or NaN */
        if (ix == 0x3f800000u) return (hx >> 31) ? 2.0f * pio2_hi : 0.0f;      /* (+ 0x1p-120 rounds away) */
        return 0.0f / (x - x);
    }
    if (ix < 0x3f000000u) {                           /* |x| < 0.5 */
        if (ix <= 0x32800000u) return pio2_hi;        /* |x| < 2^-26 */
        return pio2_hi - (x - (pio2_lo - x * acosf_R(x * x)));
    }
    if (hx >> 31) {                                   /* x < -0.5 */
        float z = (1.0f + x) * 0.5f, s = sqrtf(z);
        float w = acosf_R(z) * s - pio2_lo;
        return 2.0f * (pio2_hi - (s + w));
    }
    {                                                 /* x > 0.5 */
        float z = (1.0f - x) * 0.5f, s = sqrtf(z), df, c, w;
        uint32_t hs;
        memcpy(&hs, &s, 4); hs &= 0xfffff000u; memcpy(&df, &hs, 4);
        c = (z - df * df) / (s + df);
        w = acosf_R(z) * s + c;
        return 2.0f * (df + w);
    }
}

/* ------------------------------------------------------------------ lighting, render.rs:1013-1071 */
typedef struct { float r, g, b; } Shade;
static int shade_multi_light_color(V3 normal, V3 world_pos, const B32Light* lights, uint32_t n_lights, float ambient, Shade* out) {
    float tr = ambient, tg = ambient, tb = ambient;
    for (uint32_t i = 0; i < n_lights; ++i) {
        const B32Light* l = &lights[i];
        if (!l->enabled) continue;
        float contribution;
        if (l->type == B32_LIGHT_DIRECTIONAL) {
            V3 neg_dir = v3scale(v3p(l->direction), -1.0f);
            float n_dot_l = rmax(v3dot(normal, neg_dir), 0.0f);
            contribution = n_dot_l * l->intensity;
        } else if (l->type == B32_LIGHT_POINT) {
            V3 to_light = v3sub(v3p(l->position), world_pos);
            float dist = v3len(to_light);
            if (dist > l->radius || dist < K_LIGHT_MIN_DIST) {
                contribution = 0.0f;
            } else {
                float attenuation = 1.0f - (dist / l->radius);
                float n_dot_l = rmax(v3dot(normal, v3normalize(to_light)), 0.0f);
                contribution = n_dot_l * l->intensity * attenuation * attenuation;
            }
        } else if (l->type == B32_LIGHT_SPOT) {       /* Spot, render.rs:1038-1058 */
            V3 to_light = v3sub(v3p(l->position), world_pos);
            float dist = v3len(to_light);
            if (dist > l->radius || dist < K_LIGHT_MIN_DIST) {
                contribution = 0.0f;
            } else {
                V3 light_dir_to_surface = v3normalize(to_light);
                V3 neg_light_dir = v3scale(light_dir_to_surface, -1.0f);
                float spot_angle = b32o_acosf(v3dot(neg_light_dir, v3p(l->direction)));
                if (spot_angle > l->angle) {
                    contribution = 0.0f;
                } else {                              /* (a NaN angle -- |dot| > 1 -- lands here, like in the reference) */
                    float attenuation = 1.0f - (dist / l->radius);
                    float edge_falloff = 1.0f - (spot_angle / l->angle);
                    float n_dot_l = rmax(v3dot(normal, light_dir_to_surface), 0.0f);
                    contribution = n_dot_l * l->intensity * attenuation * attenuation * edge_falloff;
                }
            }
        } else {
            return B32_E_ARG;                         /* not a LightType */
        }
        float lr = (float)l->r / K_LIGHT_COLOR_DIV, lg = (float)l->g / K_LIGHT_COLOR_DIV, lb = (float)l->b / K_LIGHT_COLOR_DIV;
        tr += contribution * lr; tg += contribution * lg; tb += contribution * lb;
    }
    out->r = rmin(tr, K_LIGHT_TOTAL_MAX); out->g = rmin(tg, K_LIGHT_TOTAL_MAX); out->b = rmin(tb, K_LIGHT_TOTAL_MAX);
    return B32_OK;
}

/* ------------------------------------------------------------------ fog, render.rs:2266-2293 */
static inline float calculate_fog_factor(float z, float fog_start, float fog_falloff) {
    if (z <= fog_start) return 0.0f;
    if (fog_falloff <= 0.0f) return 1.0f;
    return rmin((z - fog_start) / fog_falloff, 1.0f);
}
typedef struct { uint8_t r, g, b, blend; } Col;
static inline Col apply_fog_to_color(Col c, Col fog, float f) {
    if (f <= 0.0f) return c;
    if (f >= 1.0f) return fog;
    float inv = 1.0f - f;
    Col o;
    o.r = f2u8_sat((float)c.r * inv + (float)fog.r * f);
    o.g = f2u8_sat((float)c.g * inv + (float)fog.g * f);
    o.b = f2u8_sat((float)c.b * inv + (float)fog.b * f);
    o.blend = B32_BLEND_OPAQUE; /* Color::new, types.rs:771-773 */
    return o;
}
static inline int col_eq(Col a, Col b) { return a.r == b.r && a.g == b.g && a.b == b.b && a.blend == b.blend; }

/* ------------------------------------------------------------------ Surface, render.rs:975-1000 */
typedef struct {
    V3 v1, v2, v3;          /* screen */
    V3 w1, w2, w3;          /* world pos */
    V3 wn1, wn2, wn3;       /* world normals */
    float uv1[2], uv2[2], uv3[2];
    Col vc1, vc2, vc3;
    uint32_t face_idx;
    uint8_t black_transparent, has_transparency, blend_mode, editor_alpha;
} Surface;

typedef struct {
    uint8_t* pixels;
    float*   zbuffer;       /* may be NULL when !use_zbuffer */
    uint32_t width, height;
    uint64_t fragments;
    uint32_t band_y0, band_y1;   /* rows this FB view draws (all-cores baseline); the whole frame by default */
} FB;

/* Framebuffer::set_pixel_15, render.rs:445-454 */
static inline void set_pixel_15(FB* fb, uint32_t x, uint32_t y, uint16_t color) {
    if (x < fb->width && y < fb->height) {
        size_t idx = ((size_t)y * fb->width + x) * 4;
        b32o_color15_to_rgba(color, &fb->pixels[idx]);
    }
}
/* Framebuffer::set_pixel_blended_15, render.rs:479-502 */
static inline void set_pixel_blended_15(FB* fb, uint32_t x, uint32_t y, uint16_t color, uint32_t mode) {
    if (x < fb->width && y < fb->height) {
        size_t idx = ((size_t)y * fb->width + x) * 4;
        uint8_t rgb[3];
        uint8_t r8 = expand_5_to_8(c15_r5(color)), g8 = expand_5_to_8(c15_g5(color)), b8 = expand_5_to_8(c15_b5(color));
        if (color & 0x8000) b32o_blend_rgb555(r8, g8, b8, fb->pixels[idx], fb->pixels[idx + 1], fb->pixels[idx + 2], mode, rgb);
        else { rgb[0] = r8; rgb[1] = g8; rgb[2] = b8; }
        fb->pixels[idx] = rgb[0]; fb->pixels[idx + 1] = rgb[1]; fb->pixels[idx + 2] = rgb[2]; fb->pixels[idx + 3] = 255;
    }
}
/* Framebuffer::set_pixel_xray_15, render.rs:507-526 */
static inline void set_pixel_xray_15(FB* fb, uint32_t x, uint32_t y, uint16_t color) {
    if (x < fb->width && y < fb->height) {
        size_t idx = ((size_t)y * fb->width + x) * 4;
        uint8_t c8[3] = { expand_5_to_8(c15_r5(color)), expand_5_to_8(c15_g5(color)), expand_5_to_8(c15_b5(color)) };
        for (int i = 0; i < 3; ++i) fb->pixels[idx + i] = (uint8_t)(((uint16_t)c8[i] + (uint16_t)fb->pixels[idx + i]) / 2);
        fb->pixels[idx + 3] = 255;
    }
}
/* Framebuffer::set_pixel_with_editor_alpha_15, render.rs:567-591 (alpha==0 / bounds checked by caller path too) */
static inline void editor_alpha_store(FB* fb, size_t idx, uint16_t color, uint32_t mode, uint8_t editor_alpha) {
    uint8_t back[3] = { fb->pixels[idx], fb->pixels[idx + 1], fb->pixels[idx + 2] };
    uint8_t ps1[3];
    uint8_t r8 = expand_5_to_8(c15_r5(color)), g8 = expand_5_to_8(c15_g5(color)), b8 = expand_5_to_8(c15_b5(color));
    if ((color & 0x8000) && mode != B32_BLEND_OPAQUE) b32o_blend_rgb555(r8, g8, b8, back[0], back[1], back[2], mode, ps1);
    else { ps1[0] = r8; ps1[1] = g8; ps1[2] = b8; }
    uint16_t a = editor_alpha, inv_a = (uint16_t)(255 - a);
    for (int i = 0; i < 3; ++i) fb->pixels[idx + i] = (uint8_t)((uint16_t)((uint16_t)ps1[i] * a + (uint16_t)back[i] * inv_a) / 255);
    fb->pixels[idx + 3] = 255;
}

/* rasterize_triangle_15, render.rs:1440-1714 */
static int rasterize_triangle_15(FB* fb, const Surface* s, const B32Texture15* texture, uint32_t face_blend_mode,
                                 int black_transparent, const B32Settings* st, int skip_z_write) {
    uint32_t blend_mode = texture ? texture->blend_mode : face_blend_mode;                      /* :1450-1452 */

    /* bounding box :1455-1458 */
    uint64_t min_x = f2usize_sat(rmax(rmin(rmin(s->v1.x, s->v2.x), s->v3.x), 0.0f));
    uint64_t max_x = f2usize_sat(rmin(rmax(rmax(s->v1.x, s->v2.x), s->v3.x) + 1.0f, (float)fb->width));
    uint64_t min_y = f2usize_sat(rmax(rmin(rmin(s->v1.y, s->v2.y), s->v3.y), 0.0f));
    uint64_t max_y = f2usize_sat(rmin(rmax(rmax(s->v1.y, s->v2.y), s->v3.y) + 1.0f, (float)fb->height));
    if (min_x >= max_x || min_y >= max_y) return B32_OK;                                         /* :1461-1463 */

    Shade flat_shade = { 1.0f, 1.0f, 1.0f };                                                     /* :1466-1472 */
    if (st->shading == B32_SHADE_FLAT) {
        V3 center_pos = v3scale(v3add(v3add(s->w1, s->w2), s->w3), 1.0f / 3.0f);
        V3 world_normal = v3normalize(v3scale(v3add(v3add(s->wn1, s->wn2), s->wn3), 1.0f / 3.0f));
        int rc = shade_multi_light_color(world_normal, center_pos, st->lights, st->n_lights, st->ambient, &flat_shade);
        if (rc) return rc;
    }
    Shade gs1 = { 0, 0, 0 }, gs2 = { 0, 0, 0 }, gs3 = { 0, 0, 0 };                               /* :1475-1483 */
    if (st->shading == B32_SHADE_GOURAUD) {
        int rc = shade_multi_light_color(s->wn1, s->w1, st->lights, st->n_lights, st->ambient, &gs1);
        if (!rc) rc = shade_multi_light_color(s->wn2, s->w2, st->lights, st->n_lights, st->ambient, &gs2);
        if (!rc) rc = shade_multi_light_color(s->wn3, s->w3, st->lights, st->n_lights, st->ambient, &gs3);
        if (rc) return rc;
    }
    int needs_dither = st->dithering && (st->shading == B32_SHADE_GOURAUD || texture != NULL       /* :1487-1492 */
                                         || !col_eq(s->vc1, s->vc2) || !col_eq(s->vc2, s->vc3));

    V3 v1 = s->v1, v2 = s->v2, v3_ = s->v3;
    float area = (v2.y - v3_.y) * (v1.x - v3_.x) + (v3_.x - v2.x) * (v1.y - v3_.y);              /* :1500 */
    if (fabsf(area) < K_AREA_EPS) return B32_OK;                                                 /* :1501-1503 */
    float inv_area = 1.0f / area;
    float a0 = v2.y - v3_.y, b0 = v3_.x - v2.x, a1 = v3_.y - v1.y, b1 = v1.x - v3_.x;            /* :1507-1510 */
    float start_x = (float)min_x, start_y = (float)min_y;
    float w0_row = a0 * (start_x - v3_.x) + b0 * (start_y - v3_.y);                              /* :1517 */
    float w1_row = a1 * (start_x - v3_.x) + b1 * (start_y - v3_.y);                              /* :1518 */

    if (max_y <= fb->band_y0 || min_y >= fb->band_y1) return B32_OK;  /* (all-cores baseline) no row of this triangle is ours */
    for (uint64_t y = min_y; y < max_y; ++y) {                                                   /* :1530 */
        float w0 = w0_row, w1 = w1_row;
        for (uint64_t x = (y >= fb->band_y0 && y < fb->band_y1) ? min_x : max_x; x < max_x; ++x) {
            float bc_x = w0 * inv_area;
            float bc_y = w1 * inv_area;
            float bc_z = 1.0f - bc_x - bc_y;
            const float ERR = K_ERR;
            if (bc_x >= ERR && bc_y >= ERR && bc_z >= ERR) {                                      /* :1542 */
                float inv_z1 = 1.0f / v1.z, inv_z2 = 1.0f / v2.z, inv_z3 = 1.0f / v3_.z;         /* :1546-1550 */
                float inv_z_interp = bc_x * inv_z1 + bc_y * inv_z2 + bc_z * inv_z3;
                float z = 1.0f / inv_z_interp;
                if (st->use_zbuffer && !st->xray_mode) {                                          /* :1553-1560 */
                    size_t idx = (size_t)y * fb->width + x;
                    if (z >= fb->zbuffer[idx]) { w0 += a0; w1 += a1; continue; }
                }
                float u, v;
                if (st->affine_textures) {                                                        /* :1563-1567 */
                    u = bc_x * s->uv1[0] + bc_y * s->uv2[0] + bc_z * s->uv3[0];
                    v = bc_x * s->uv1[1] + bc_y * s->uv2[1] + bc_z * s->uv3[1];
                } else {                                                                          /* :1568-1579 */
                    float u_over_z = bc_x * s->uv1[0] * inv_z1 + bc_y * s->uv2[0] * inv_z2 + bc_z * s->uv3[0] * inv_z3;
                    float v_over_z = bc_x * s->uv1[1] * inv_z1 + bc_y * s->uv2[1] * inv_z2 + bc_z * s->uv3[1] * inv_z3;
                    u = u_over_z / inv_z_interp;
                    v = v_over_z / inv_z_interp;
                }
                uint16_t color = texture ? b32o_texture15_sample(texture->pixels, texture->width, texture->height, u, 1.0f - v)
                                         : (uint16_t)K_C15_WHITE;                                 /* :1582-1586 */
                int is_black = c15_r5(color) == 0 && c15_g5(color) == 0 && c15_b5(color) == 0;    /* :1591 */
                if (color == K_C15_TRANSPARENT) {                                                 /* :1592-1602 */
                    if (is_black && !black_transparent) color = K_C15_BLACK_DRAWABLE;
                    else { w0 += a0; w1 += a1; continue; }
                } else if (black_transparent && is_black) {                                       /* :1603-1608 */
                    w0 += a0; w1 += a1; continue;
                }
                uint8_t tex_r8 = expand_5_to_8(c15_r5(color)), tex_g8 = expand_5_to_8(c15_g5(color)), tex_b8 = expand_5_to_8(c15_b5(color));
                uint8_t vertex_r = f2u8_sat(bc_x * (float)s->vc1.r + bc_y * (float)s->vc2.r + bc_z * (float)s->vc3.r); /* :1618-1620 */
                uint8_t vertex_g = f2u8_sat(bc_x * (float)s->vc1.g + bc_y * (float)s->vc2.g + bc_z * (float)s->vc3.g);
                uint8_t vertex_b = f2u8_sat(bc_x * (float)s->vc1.b + bc_y * (float)s->vc2.b + bc_z * (float)s->vc3.b);
                uint32_t m;
                m = ((uint32_t)tex_r8 * vertex_r) / K_MOD_DIV; uint8_t mod_r8 = (uint8_t)(m > K_MOD_MAX ? K_MOD_MAX : m); /* :1624-1626 */
                m = ((uint32_t)tex_g8 * vertex_g) / K_MOD_DIV; uint8_t mod_g8 = (uint8_t)(m > K_MOD_MAX ? K_MOD_MAX : m);
                m = ((uint32_t)tex_b8 * vertex_b) / K_MOD_DIV; uint8_t mod_b8 = (uint8_t)(m > K_MOD_MAX ? K_MOD_MAX : m);
                float shade_r, shade_g, shade_b;                                                  /* :1629-1640 */
                if (st->shading == B32_SHADE_NONE) { shade_r = shade_g = shade_b = 1.0f; }
                else if (st->shading == B32_SHADE_FLAT) { shade_r = flat_shade.r; shade_g = flat_shade.g; shade_b = flat_shade.b; }
                else {
                    shade_r = bc_x * gs1.r + bc_y * gs2.r + bc_z * gs3.r;
                    shade_g = bc_x * gs1.g + bc_y * gs2.g + bc_z * gs3.g;
                    shade_b = bc_x * gs1.b + bc_y * gs2.b + bc_z * gs3.b;
                }
                uint8_t shaded_r8 = f2u8_sat(rmin((float)mod_r8 * rclamp(shade_r, K_SHADE_LO, K_SHADE_HI), K_SHADE_MAX)); /* :1643-1645 */
                uint8_t shaded_g8 = f2u8_sat(rmin((float)mod_g8 * rclamp(shade_g, K_SHADE_LO, K_SHADE_HI), K_SHADE_MAX));
                uint8_t shaded_b8 = f2u8_sat(rmin((float)mod_b8 * rclamp(shade_b, K_SHADE_LO, K_SHADE_HI), K_SHADE_MAX));
                uint8_t q5[3];
                if (needs_dither) b32o_dither_and_quantize(shaded_r8, shaded_g8, shaded_b8, (uint32_t)x, (uint32_t)y, q5); /* :1649-1654 */
                else { q5[0] = shaded_r8 >> K_NODITHER_SHIFT; q5[1] = shaded_g8 >> K_NODITHER_SHIFT; q5[2] = shaded_b8 >> K_NODITHER_SHIFT; }
                int is_all_black = q5[0] == 0 && q5[1] == 0 && q5[2] == 0;                        /* :1659-1661 */
                int semi = (color & K_C15_SEMI_BIT) != 0 || is_all_black;
                uint16_t out = c15_new_semi(q5[0], q5[1], q5[2], semi);

                uint8_t editor_alpha = s->editor_alpha;                                           /* :1664-1669 */
                if (editor_alpha == 0) { w0 += a0; w1 += a1; continue; }
                size_t didx = (size_t)y * fb->width + x;
                if (st->xray_mode) {                                                              /* :1671-1673 */
                    set_pixel_xray_15(fb, (uint32_t)x, (uint32_t)y, out); fb->fragments++;
                } else if (editor_alpha < 255) {                                                  /* :1674-1680 */
                    if (st->use_zbuffer) {                                                        /* render.rs:595-628 */
                        if (!(z >= fb->zbuffer[didx])) {
                            if (!skip_z_write) fb->zbuffer[didx] = z;
                            editor_alpha_store(fb, didx * 4, out, blend_mode, editor_alpha); fb->fragments++;
                        }
                    } else {
                        editor_alpha_store(fb, didx * 4, out, blend_mode, editor_alpha); fb->fragments++;
                    }
                } else if (st->use_zbuffer) {                                                     /* :1681-1694 */
                    if (z < fb->zbuffer[didx]) {
                        if (!skip_z_write) fb->zbuffer[didx] = z;
                        if ((out & 0x8000) && blend_mode != B32_BLEND_OPAQUE) set_pixel_blended_15(fb, (uint32_t)x, (uint32_t)y, out, blend_mode);
                        else set_pixel_15(fb, (uint32_t)x, (uint32_t)y, out);
                        fb->fragments++;
                    }
                } else {                                                                          /* :1695-1702 painter's */
                    if ((out & 0x8000) && blend_mode != B32_BLEND_OPAQUE) set_pixel_blended_15(fb, (uint32_t)x, (uint32_t)y, out, blend_mode);
                    else set_pixel_15(fb, (uint32_t)x, (uint32_t)y, out);
                    fb->fragments++;
                }
            }
            w0 += a0; w1 += a1;                                                                   /* :1706-1707 */
        }
        w0_row += b0; w1_row += b1;                                                               /* :1711-1712 */
    }
    return B32_OK;
}

/* ------------------------------------------------------------------ the 8-bit-colour path */
/* Texture::sample, types.rs:1242-1253 -> Color{r,g,b,blend}; zero-size / empty -> Color::TRANSPARENT (0,0,0,Erase) */
static inline Col texture_sample8(const B32Texture* t, float u, float v) {
    Col tr = { 0, 0, 0, B32_BLEND_ERASE };
    if (t->width == 0 || t->height == 0 || !t->pixels) return tr;
    float uw = rem_euclid1(u), vw = rem_euclid1(v);
    uint64_t tx = f2usize_sat(uw * (float)t->width), ty = f2usize_sat(vw * (float)t->height);
    if (tx > t->width - 1) tx = t->width - 1;
    if (ty > t->height - 1) ty = t->height - 1;
    const uint8_t* p = t->pixels + (ty * t->width + tx) * 4;
    Col c = { p[0], p[1], p[2], p[3] };
    return c;
}
/* Color::blend_with, types.rs:886-936, front.blend = mode (Color::blend :940-942). Returns the bytes Color::to_bytes would store. */
static inline void color_blend_bytes(Col front, uint32_t mode, const uint8_t back[3], uint8_t out[4]) {
    const uint8_t f[3] = { front.r, front.g, front.b };
    out[3] = 255;
    for (int i = 0; i < 3; ++i) {
        switch (mode) {
            default:
            case B32_BLEND_OPAQUE:      out[i] = f[i]; break;
            case B32_BLEND_AVERAGE:     out[i] = (uint8_t)(((uint16_t)back[i] + (uint16_t)f[i]) / 2); break;
            case B32_BLEND_ADD:         { uint16_t v = (uint16_t)(back[i] + f[i]); out[i] = (uint8_t)(v > 255 ? 255 : v); } break;
            case B32_BLEND_SUBTRACT:    { int16_t v = (int16_t)((int16_t)back[i] - (int16_t)f[i]); out[i] = (uint8_t)(v < 0 ? 0 : v); } break;
            case B32_BLEND_ADD_QUARTER: { uint16_t v = (uint16_t)(back[i] + f[i] / 4); out[i] = (uint8_t)(v > 255 ? 255 : v); } break;
            case B32_BLEND_ERASE:       out[i] = 0; out[3] = 0; break;                      /* Color::TRANSPARENT */
        }
    }
}
/* Framebuffer::set_pixel_blended, render.rs:313-334 */
static inline void set_pixel_blended8(FB* fb, size_t idx, Col color, uint32_t mode) {
    uint8_t out[4];
    color_blend_bytes(color, mode, &fb->pixels[idx], out);
    memcpy(&fb->pixels[idx], out, 4);
}
/* editor-alpha lerp of set_pixel_with_editor_alpha / set_pixel_with_depth_and_editor_alpha, render.rs:339-375, 378-427 */
static inline void editor_alpha_store8(FB* fb, size_t idx, Col color, uint32_t mode, uint8_t editor_alpha) {
    uint8_t back[3] = { fb->pixels[idx], fb->pixels[idx + 1], fb->pixels[idx + 2] };
    uint8_t ps1[4];
    color_blend_bytes(color, mode, back, ps1);
    if (editor_alpha < 255) {
        float a = (float)editor_alpha / 255.0f, inv_a = 1.0f - a;
        for (int i = 0; i < 3; ++i) fb->pixels[idx + i] = f2u8_sat((float)ps1[i] * a + (float)back[i] * inv_a);
        fb->pixels[idx + 3] = 255;                                                            /* Color::new -> Opaque */
    } else memcpy(&fb->pixels[idx], ps1, 4);
}
/* apply_dither, render.rs:1186-1197 */
static inline uint8_t dither8(uint8_t c, int32_t off) {
    int32_t q = ((int32_t)c + off) >> K_DITHER8_SHIFT;
    q = q < 0 ? 0 : (q > K_DITHER8_HI ? K_DITHER8_HI : q);
    return (uint8_t)(q << K_DITHER8_EXPAND_SHIFT);
}

/* rasterize_triangle, render.rs:1202-1433 */
static int rasterize_triangle8(FB* fb, const Surface* s, const B32Texture* texture, const B32Settings* st) {
    uint64_t min_x = f2usize_sat(rmax(rmin(rmin(s->v1.x, s->v2.x), s->v3.x), 0.0f));           /* :1209-1212 */
    uint64_t max_x = f2usize_sat(rmin(rmax(rmax(s->v1.x, s->v2.x), s->v3.x) + 1.0f, (float)fb->width));
    uint64_t min_y = f2usize_sat(rmax(rmin(rmin(s->v1.y, s->v2.y), s->v3.y), 0.0f));
    uint64_t max_y = f2usize_sat(rmin(rmax(rmax(s->v1.y, s->v2.y), s->v3.y) + 1.0f, (float)fb->height));
    if (min_x >= max_x || min_y >= max_y) return B32_OK;
    Shade flat_shade = { 1.0f, 1.0f, 1.0f };                                                     /* :1220-1226 */
    if (st->shading == B32_SHADE_FLAT) {
        V3 center_pos = v3scale(v3add(v3add(s->w1, s->w2), s->w3), 1.0f / 3.0f);
        V3 world_normal = v3normalize(v3scale(v3add(v3add(s->wn1, s->wn2), s->wn3), 1.0f / 3.0f));
        int rc = shade_multi_light_color(world_normal, center_pos, st->lights, st->n_lights, st->ambient, &flat_shade);
        if (rc) return rc;
    }
    Shade gs1 = { 0, 0, 0 }, gs2 = { 0, 0, 0 }, gs3 = { 0, 0, 0 };                               /* :1229-1237 */
    if (st->shading == B32_SHADE_GOURAUD) {
        int rc = shade_multi_light_color(s->wn1, s->w1, st->lights, st->n_lights, st->ambient, &gs1);
        if (!rc) rc = shade_multi_light_color(s->wn2, s->w2, st->lights, st->n_lights, st->ambient, &gs2);
        if (!rc) rc = shade_multi_light_color(s->wn3, s->w3, st->lights, st->n_lights, st->ambient, &gs3);
        if (rc) return rc;
    }
    int needs_dither = st->dithering && (st->shading == B32_SHADE_GOURAUD || texture != NULL       /* :1241-1246 */
                                         || !col_eq(s->vc1, s->vc2) || !col_eq(s->vc2, s->vc3));
    V3 v1 = s->v1, v2 = s->v2, v3_ = s->v3;
    float area = (v2.y - v3_.y) * (v1.x - v3_.x) + (v3_.x - v2.x) * (v1.y - v3_.y);              /* :1257 */
    if (fabsf(area) < K_AREA_EPS8) return B32_OK;
    float inv_area = 1.0f / area;
    float a0 = v2.y - v3_.y, b0 = v3_.x - v2.x, a1 = v3_.y - v1.y, b1 = v1.x - v3_.x;            /* :1264-1269 */
    float start_x = (float)min_x, start_y = (float)min_y;
    float w0_row = a0 * (start_x - v3_.x) + b0 * (start_y - v3_.y);                              /* :1277-1278 */
    float w1_row = a1 * (start_x - v3_.x) + b1 * (start_y - v3_.y);
    if (max_y <= fb->band_y0 || min_y >= fb->band_y1) return B32_OK;
    for (uint64_t y = min_y; y < max_y; ++y) {
        float w0 = w0_row, w1 = w1_row;
        for (uint64_t x = (y >= fb->band_y0 && y < fb->band_y1) ? min_x : max_x; x < max_x; ++x) {
            float bc_x = w0 * inv_area, bc_y = w1 * inv_area;
            float bc_z = 1.0f - bc_x - bc_y;
            const float ERR = K_ERR8;
            if (bc_x >= ERR && bc_y >= ERR && bc_z >= ERR) {                                      /* :1302 */
                float inv_z1 = 1.0f / v1.z, inv_z2 = 1.0f / v2.z, inv_z3 = 1.0f / v3_.z;         /* :1305-1309 */
                float inv_z_interp = bc_x * inv_z1 + bc_y * inv_z2 + bc_z * inv_z3;
                float z = 1.0f / inv_z_interp;
                size_t didx = (size_t)y * fb->width + x;
                if (st->use_zbuffer && !st->xray_mode) {                                          /* :1312-1319 */
                    if (z >= fb->zbuffer[didx]) { w0 += a0; w1 += a1; continue; }
                }
                float u, v;
                if (st->affine_textures) {                                                        /* :1322-1326 */
                    u = bc_x * s->uv1[0] + bc_y * s->uv2[0] + bc_z * s->uv3[0];
                    v = bc_x * s->uv1[1] + bc_y * s->uv2[1] + bc_z * s->uv3[1];
                } else {                                                                          /* :1327-1338 */
                    float u_over_z = bc_x * s->uv1[0] * inv_z1 + bc_y * s->uv2[0] * inv_z2 + bc_z * s->uv3[0] * inv_z3;
                    float v_over_z = bc_x * s->uv1[1] * inv_z1 + bc_y * s->uv2[1] * inv_z2 + bc_z * s->uv3[1] * inv_z3;
                    u = u_over_z / inv_z_interp;
                    v = v_over_z / inv_z_interp;
                }
                Col color = { 255, 255, 255, B32_BLEND_OPAQUE };                                   /* Color::WHITE :1344 */
                if (texture) color = texture_sample8(texture, u, 1.0f - v);                       /* :1342 */
                if (color.blend == B32_BLEND_ERASE) { w0 += a0; w1 += a1; continue; }             /* is_transparent :1348-1352 */
                uint8_t vr = f2u8_sat(bc_x * (float)s->vc1.r + bc_y * (float)s->vc2.r + bc_z * (float)s->vc3.r);   /* :1355-1360 */
                uint8_t vg = f2u8_sat(bc_x * (float)s->vc1.g + bc_y * (float)s->vc2.g + bc_z * (float)s->vc3.g);
                uint8_t vb = f2u8_sat(bc_x * (float)s->vc1.b + bc_y * (float)s->vc2.b + bc_z * (float)s->vc3.b);
                uint16_t m;                                                                       /* Color::modulate types.rs:801-808 */
                m = (uint16_t)(((uint16_t)color.r * (uint16_t)vr) / 128); color.r = (uint8_t)(m > 255 ? 255 : m);
                m = (uint16_t)(((uint16_t)color.g * (uint16_t)vg) / 128); color.g = (uint8_t)(m > 255 ? 255 : m);
                m = (uint16_t)(((uint16_t)color.b * (uint16_t)vb) / 128); color.b = (uint8_t)(m > 255 ? 255 : m);
                float shade_r, shade_g, shade_b;                                                  /* :1366-1379 */
                if (st->shading == B32_SHADE_NONE) { shade_r = shade_g = shade_b = 1.0f; }
                else if (st->shading == B32_SHADE_FLAT) { shade_r = flat_shade.r; shade_g = flat_shade.g; shade_b = flat_shade.b; }
                else {
                    shade_r = bc_x * gs1.r + bc_y * gs2.r + bc_z * gs3.r;
                    shade_g = bc_x * gs1.g + bc_y * gs2.g + bc_z * gs3.g;
                    shade_b = bc_x * gs1.b + bc_y * gs2.b + bc_z * gs3.b;
                }
                color.r = f2u8_sat(rmin((float)color.r * shade_r, 255.0f));                        /* shade_color_rgb :1074-1081 */
                color.g = f2u8_sat(rmin((float)color.g * shade_g, 255.0f));
                color.b = f2u8_sat(rmin((float)color.b * shade_b, 255.0f));
                if (needs_dither) {                                                               /* :1385-1387 */
                    int32_t off = PS1_DITHER_MATRIX[y & 3][x & 3];
                    color.r = dither8(color.r, off); color.g = dither8(color.g, off); color.b = dither8(color.b, off);
                }
                uint8_t editor_alpha = s->editor_alpha;                                           /* :1390-1396 */
                if (editor_alpha == 0) { w0 += a0; w1 += a1; continue; }
                uint8_t bytes[4] = { color.r, color.g, color.b, 255 };                            /* to_bytes: blend != Erase here */
                if (st->use_zbuffer) {                                                            /* :1398-1412 */
                    if (editor_alpha < 255) {                                                     /* render.rs:378-427 */
                        if (!(z >= fb->zbuffer[didx])) { fb->zbuffer[didx] = z; editor_alpha_store8(fb, didx * 4, color, color.blend, editor_alpha); fb->fragments++; }
                    } else if (color.blend == B32_BLEND_OPAQUE) {                                 /* set_pixel_with_depth :429-444 */
                        if (z < fb->zbuffer[didx]) { fb->zbuffer[didx] = z; memcpy(&fb->pixels[didx * 4], bytes, 4); fb->fragments++; }
                    } else if (z < fb->zbuffer[didx]) {                                           /* :1405-1411 */
                        fb->zbuffer[didx] = z; set_pixel_blended8(fb, didx * 4, color, color.blend); fb->fragments++;
                    }
                } else {                                                                          /* :1413-1422 painter's */
                    if (editor_alpha < 255) editor_alpha_store8(fb, didx * 4, color, color.blend, editor_alpha);
                    else if (color.blend == B32_BLEND_OPAQUE) memcpy(&fb->pixels[didx * 4], bytes, 4);
                    else set_pixel_blended8(fb, didx * 4, color, color.blend);
                    fb->fragments++;
                }
            }
            w0 += a0; w1 += a1;
        }
        w0_row += b0; w1_row += b1;
    }
    return B32_OK;
}

/* ------------------------------------------------------------------ lines of the wireframe phases */
/* Framebuffer::set_pixel, render.rs:301-310 with Color::new(r,g,b) (blend Opaque -> alpha 255, types.rs:829-832) */
static inline void set_pixel_rgb(FB* fb, int32_t x, int32_t y, uint8_t r, uint8_t g, uint8_t b) {
    if ((uint32_t)x < fb->width && (uint32_t)y < fb->height && (uint32_t)y >= fb->band_y0 && (uint32_t)y < fb->band_y1) {
        size_t idx = ((size_t)y * fb->width + (size_t)x) * 4;
        fb->pixels[idx] = r; fb->pixels[idx + 1] = g; fb->pixels[idx + 2] = b; fb->pixels[idx + 3] = 255;
    }
}
static inline int32_t i32_wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t i32_wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static inline int32_t i32_wabs(int32_t a) { return a < 0 ? (int32_t)(0u - (uint32_t)a) : a; }
/* Lines whose Bresenham state would overflow i32 in the reference (debug builds panic, release builds wrap and spin):
 * |dx| or |dy| >= 2^30 makes `2 * err` overflow.  Reported as B32_E_UNSUPPORTED by oracle and GPU alike. */
static inline int line_overflows(int32_t x0, int32_t y0, int32_t x1, int32_t y1) {
    int64_t adx = (int64_t)x1 - x0, ady = (int64_t)y1 - y0;
    if (adx < 0) adx = -adx;
    if (ady < 0) ady = -ady;
    return adx >= ((int64_t)1 << 30) || ady >= ((int64_t)1 << 30);
}
/* Framebuffer::draw_line -> draw_line_blended(mode = Opaque), render.rs:716-750 */
EXPORT int b32o_draw_line(uint8_t* pixels, uint32_t width, uint32_t height, int32_t x0, int32_t y0, int32_t x1, int32_t y1,
                          uint8_t r, uint8_t g, uint8_t b) {
    if (line_overflows(x0, y0, x1, y1)) return B32_E_UNSUPPORTED;
    FB fb = { pixels, NULL, width, height, 0, g_band_y0, g_band_y1 };
    int32_t dx = i32_wabs(i32_wsub(x1, x0)), dy = -i32_wabs(i32_wsub(y1, y0));
    int32_t sx = x0 < x1 ? 1 : -1, sy = y0 < y1 ? 1 : -1;
    int32_t err = i32_wadd(dx, dy), x = x0, y = y0;
    for (;;) {
        if (x >= 0 && x < (int32_t)width && y >= 0 && y < (int32_t)height) set_pixel_rgb(&fb, x, y, r, g, b);
        if (x == x1 && y == y1) break;
        int32_t e2 = 2 * err;
        if (e2 >= dy) { err = i32_wadd(err, dy); x += sx; }
        if (e2 <= dx) { err = i32_wadd(err, dx); y += sy; }
    }
    return B32_OK;
}
/* Framebuffer::draw_line_3d -> draw_line_3d_impl(allow_equal = false), render.rs:757-817.  zbuffer == NULL: every depth f32::MAX */
EXPORT int b32o_draw_line_3d(uint8_t* pixels, const float* zbuffer, uint32_t width, uint32_t height,
                             int32_t x0, int32_t y0, float z0, int32_t x1, int32_t y1, float z1, uint8_t r, uint8_t g, uint8_t b) {
    if (line_overflows(x0, y0, x1, y1)) return B32_E_UNSUPPORTED;
    FB fb = { pixels, NULL, width, height, 0, g_band_y0, g_band_y1 };
    int32_t dx = i32_wabs(i32_wsub(x1, x0)), dy = -i32_wabs(i32_wsub(y1, y0));
    int32_t sx = x0 < x1 ? 1 : -1, sy = y0 < y1 ? 1 : -1;
    int32_t err = i32_wadd(dx, dy), x = x0, y = y0;
    int32_t ndy = -dy;
    int32_t m = ndy > 1 ? ndy : 1;
    float total_steps = (float)(dx > m ? dx : m);                                                 /* dx.max((-dy).max(1)) as f32 */
    float step = 0.0f;
    for (;;) {
        if (x >= 0 && x < (int32_t)width && y >= 0 && y < (int32_t)height) {
            float t = step / total_steps;
            float z = z0 + t * (z1 - z0);
            size_t idx = (size_t)y * width + (size_t)x;
            float zb = zbuffer ? zbuffer[idx] : 3.40282347e+38f;
            if (z < zb) set_pixel_rgb(&fb, x, y, r, g, b);
        }
        if (x == x1 && y == y1) break;
        int32_t e2 = 2 * err;
        if (e2 >= dy) { err = i32_wadd(err, dy); x += sx; step += 1.0f; }
        if (e2 <= dx) { err = i32_wadd(err, dx); y += sy; if (e2 < dy) step += 1.0f; }
    }
    return B32_OK;
}
typedef struct { int32_t x0, y0; float z0; int32_t x1, y1; float z1; } WireEdge;
typedef struct { V3 v1, v2, v3; } WireTri;
/* unique-edge list of a wireframe phase, render.rs:2578-2596 / 2604-2626: first occurrence wins, compared on screen integers only */
static inline WireEdge wire_edge_of(const WireTri* t, int j) {
    const V3 p[3] = { t->v1, t->v2, t->v3 };
    const V3 a = p[j], b = p[(j + 1) % 3];
    WireEdge e = { f2i32_sat(a.x), f2i32_sat(a.y), a.z, f2i32_sat(b.x), f2i32_sat(b.y), b.z };
    if (!(e.x0 < e.x1 || (e.x0 == e.x1 && e.y0 < e.y1))) {                                      /* (x0, y0) < (x1, y1) as tuples */
        WireEdge r = { e.x1, e.y1, e.z1, e.x0, e.y0, e.z0 };
        e = r;
    }
    return e;
}
/* the reference's own form: `unique_edges.iter().any(..)` before every push -- O(n^2), kept as the definition (small inputs and the
 * self-check below) */
static WireEdge* unique_edges_quadratic(const WireTri* tris, uint32_t n, uint32_t* n_out) {
    WireEdge* out = (WireEdge*)malloc(sizeof(WireEdge) * ((size_t)n * 3 + 1));
    uint32_t cnt = 0;
    for (uint32_t i = 0; i < n; ++i) {
        for (int j = 0; j < 3; ++j) {
            const WireEdge e = wire_edge_of(&tris[i], j);
            int seen = 0;
            for (uint32_t k = 0; k < cnt && !seen; ++k)
                seen = out[k].x0 == e.x0 && out[k].y0 == e.y0 && out[k].x1 == e.x1 && out[k].y1 == e.y1;
            if (!seen) out[cnt++] = e;
        }
    }
    *n_out = cnt;
    return out;
}
/* The same list through a hash set of the four screen integers (open addressing; a slot holds the index of the edge in `out`): an edge
 * is pushed iff no equal edge was pushed before, in the same (face, edge) order -- the definition above, in O(n).  The 1 M-triangle
 * frames of RasterSettings::default() need it: the quadratic scan takes minutes there.  b32o_unique_edges_selfcheck compares the two. */
static WireEdge* unique_edges(const WireTri* tris, uint32_t n, uint32_t* n_out) {
    if (n < 64) return unique_edges_quadratic(tris, n, n_out);
    WireEdge* out = (WireEdge*)malloc(sizeof(WireEdge) * ((size_t)n * 3 + 1));
    size_t slots = 1024;
    while (slots < (size_t)n * 6) slots <<= 1;
    uint32_t* table = (uint32_t*)malloc(sizeof(uint32_t) * slots);
    memset(table, 0xFF, sizeof(uint32_t) * slots);
    uint32_t cnt = 0;
    for (uint32_t i = 0; i < n; ++i) {
        for (int j = 0; j < 3; ++j) {
            const WireEdge e = wire_edge_of(&tris[i], j);
            uint64_t h = (uint64_t)(uint32_t)e.x0 * 0x9E3779B97F4A7C15ull;
            h = (h ^ (h >> 29)) + (uint64_t)(uint32_t)e.y0 * 0xBF58476D1CE4E5B9ull;
            h = (h ^ (h >> 31)) + (uint64_t)(uint32_t)e.x1 * 0x94D049BB133111EBull;
            h = (h ^ (h >> 27)) + (uint64_t)(uint32_t)e.y1 * 0xD6E8FEB86659FD93ull;
            size_t s = (size_t)(h ^ (h >> 32)) & (slots - 1);
            int seen = 0;
            while (table[s] != 0xFFFFFFFFu) {
                const WireEdge* o = &out[table[s]];
                if (o->x0 == e.x0 && o->y0 == e.y0 && o->x1 == e.x1 && o->y1 == e.y1) { seen = 1; break; }
                s = (s + 1) & (slots - 1);
            }
            if (!seen) { table[s] = cnt; out[cnt++] = e; }
        }
    }
    free(table);
    *n_out = cnt;
    return out;
}
/* test hook: n triangles as 9 floats each (three screen vertices x, y, z); 0 = the hash-set list equals the quadratic one, entry for
 * entry (coordinates and depths), else 1 + index of the first difference (or of the shorter list's end) */
EXPORT uint32_t b32o_unique_edges_selfcheck(const float* tri_xyz, uint32_t n) {
    WireTri* t = (WireTri*)malloc(sizeof(WireTri) * (n ? n : 1));
    for (uint32_t i = 0; i < n; ++i) {
        const float* p = tri_xyz + (size_t)i * 9;
        const V3 a = { p[0], p[1], p[2] }, b = { p[3], p[4], p[5] }, c = { p[6], p[7], p[8] };
        t[i].v1 = a; t[i].v2 = b; t[i].v3 = c;
    }
    uint32_t na = 0, nb = 0, bad = 0;
    WireEdge* a = unique_edges_quadratic(t, n, &na);
    WireEdge* b = unique_edges(t, n, &nb);
    const uint32_t m = na < nb ? na : nb;
    for (uint32_t i = 0; i < m && !bad; ++i)
        if (memcmp(&a[i], &b[i], sizeof(WireEdge)) != 0) bad = 1 + i;
    if (!bad && na != nb) bad = 1 + m;
    free(a); free(b); free(t);
    return bad;
}

/* ------------------------------------------------------------------ stable descending sort (slice::sort_by, render.rs:2527-2542) */
static inline float center_z(const Surface* s) { return (s->v1.z + s->v2.z + s->v3.z) / 3.0f; }
static void merge_sort_desc(uint32_t* idx, uint32_t* tmp, const float* key, uint32_t n) {
    for (uint32_t width = 1; width < n; width *= 2) {
        for (uint32_t lo = 0; lo < n; lo += 2 * width) {
            uint32_t mid = lo + width < n ? lo + width : n;
            uint32_t hi = lo + 2 * width < n ? lo + 2 * width : n;
            uint32_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi) {
                /* comparator(a,b) = b_key.partial_cmp(a_key): take right only when it is strictly "less", i.e. key[right] > key[left] */
                if (key[idx[j]] > key[idx[i]]) tmp[k++] = idx[j++];
                else tmp[k++] = idx[i++];
            }
            while (i < mid) tmp[k++] = idx[i++];
            while (j < hi) tmp[k++] = idx[j++];
        }
        memcpy(idx, tmp, (size_t)n * sizeof(uint32_t));
    }
}

/* The same stable sort for the all-cores baseline: T contiguous chunks sorted by T threads, then adjacent chunks merged pairwise, level by
 * level (a tie takes the LEFT element, like the merge above: chunks are adjacent ranges of the original order, so stability -- ties keep
 * face order -- is preserved).  The result is identical to merge_sort_desc's: a stable sort has exactly one output. */
typedef struct { uint32_t* idx; uint32_t* tmp; const float* key; uint32_t lo, mid, hi; int sort_only; } SortJob;
static void* sort_job(void* arg) {
    SortJob* j = (SortJob*)arg;
    if (j->sort_only) { merge_sort_desc(j->idx + j->lo, j->tmp + j->lo, j->key, j->hi - j->lo); return NULL; }
    uint32_t i = j->lo, r = j->mid, k = j->lo;
    const uint32_t* idx = j->idx; uint32_t* tmp = j->tmp; const float* key = j->key;
    while (i < j->mid && r < j->hi) { if (key[idx[r]] > key[idx[i]]) tmp[k++] = idx[r++]; else tmp[k++] = idx[i++]; }
    while (i < j->mid) tmp[k++] = idx[i++];
    while (r < j->hi) tmp[k++] = idx[r++];
    memcpy(j->idx + j->lo, tmp + j->lo, (size_t)(j->hi - j->lo) * sizeof(uint32_t));
    return NULL;
}
static void merge_sort_desc_mt(uint32_t* idx, uint32_t* tmp, const float* key, uint32_t n, uint32_t threads) {
    uint32_t T = 1;
    while (T * 2 <= threads && T * 2 <= 64 && n / (T * 2) >= 4096) T *= 2;
    if (T == 1) { merge_sort_desc(idx, tmp, key, n); return; }
    SortJob jobs[64]; pthread_t th[64];
    uint32_t bound[65];
    for (uint32_t t = 0; t <= T; ++t) bound[t] = (uint32_t)((uint64_t)n * t / T);
    for (uint32_t t = 0; t < T; ++t) { SortJob j = { idx, tmp, key, bound[t], 0, bound[t + 1], 1 }; jobs[t] = j; pthread_create(&th[t], NULL, sort_job, &jobs[t]); }
    for (uint32_t t = 0; t < T; ++t) pthread_join(th[t], NULL);
    for (uint32_t span = 1; span < T; span *= 2) {
        uint32_t m = 0;
        for (uint32_t t = 0; t + span < T; t += 2 * span) {
            const uint32_t hi_t = t + 2 * span < T ? t + 2 * span : T;
            SortJob j = { idx, tmp, key, bound[t], bound[t + span], bound[hi_t], 0 }; jobs[m] = j;
            pthread_create(&th[m], NULL, sort_job, &jobs[m]); ++m;
        }
        for (uint32_t t = 0; t < m; ++t) pthread_join(th[t], NULL);
    }
}

/* Optional stage dump for parity tests (all arrays caller-allocated or NULL). */
typedef struct B32OracleDump {
    int32_t*  sx;          /* nv: screen x as i32 (fixed-point path) / truncated float otherwise */
    int32_t*  sy;          /* nv */
    float*    sz;          /* nv: screen z (render.rs:2345) */
    uint32_t* draw_order;  /* nf capacity: face_idx in draw order */
    uint32_t  n_drawn;
    uint32_t  n_opaque;
} B32OracleDump;

/* Framebuffer::clear, render.rs:36-45 + Color::to_bytes types.rs:829-832 */
EXPORT void b32o_fb_clear(uint8_t* pixels, float* zbuffer, uint32_t width, uint32_t height, uint8_t r, uint8_t g, uint8_t b, uint8_t blend) {
    uint8_t a = blend == B32_BLEND_ERASE ? 0 : 255;
    for (size_t i = 0; i < (size_t)width * height; ++i) {
        pixels[i * 4] = r; pixels[i * 4 + 1] = g; pixels[i * 4 + 2] = b; pixels[i * 4 + 3] = a;
        if (zbuffer) zbuffer[i] = 3.40282347e+38f;
    }
}

/* render_mesh_15, render.rs:2302-2638, and render_mesh (8-bit colour), render.rs:1971-2264: the two functions share their
 * transform / cull / wireframe text almost verbatim; `textures8 != NULL` selects the 8-bit variant's differences (cited). */
/* TRANSFORM of a vertex range, render.rs:2313-2362 (one job per thread of the all-cores baseline; the whole range otherwise) */
typedef struct {
    const B32Vertex* vertices; uint32_t i0, i1;
    const B32Camera* camera; const B32Settings* st; uint32_t width, height;
    V3* cam_space; V3* projected; B32OracleDump* dump;
} TransformJob;
static void* transform_range(void* arg) {
    const TransformJob* j = (const TransformJob*)arg;
    const B32Vertex* vertices = j->vertices; const B32Camera* camera = j->camera; const B32Settings* st = j->st;
    const uint32_t width = j->width, height = j->height;
    B32OracleDump* dump = j->dump;
    V3 cpos = v3p(camera->position), bx = v3p(camera->basis_x), by = v3p(camera->basis_y), bz = v3p(camera->basis_z);
    for (uint32_t i = j->i0; i < j->i1; ++i) {
        V3 pos = v3p(vertices[i].pos);
        V3 screen, cam_pos;
        if (st->has_ortho) {                                                                      /* :2323-2328 */
            cam_pos = perspective_transform(v3sub(pos, cpos), bx, by, bz);
            screen = project_ortho(cam_pos, st->ortho_zoom, st->ortho_center_x, st->ortho_center_y, width, height);
        } else if (st->use_fixed_point) {                                                         /* :2329-2345 */
            int32_t sx, sy; float fd;
            b32o_project_fixed(vertices[i].pos, camera->position, camera->basis_x, camera->basis_y, camera->basis_z, width, height, &sx, &sy, &fd);
            cam_pos = perspective_transform(v3sub(pos, cpos), bx, by, bz);
            screen = v3((float)sx, (float)sy, cam_pos.z + K_MESH_DISTANCE);
            if (dump && dump->sx) { dump->sx[i] = sx; dump->sy[i] = sy; }
        } else {                                                                                  /* :2346-2352 */
            cam_pos = perspective_transform(v3sub(pos, cpos), bx, by, bz);
            screen = project_float(cam_pos, width, height);
        }
        if (dump && dump->sx && !(st->use_fixed_point && !st->has_ortho)) { dump->sx[i] = f2i32_sat(screen.x); dump->sy[i] = f2i32_sat(screen.y); }
        if (dump && dump->sz) dump->sz[i] = screen.z;
        j->cam_space[i] = cam_pos; j->projected[i] = screen;
        /* cam_space_normals (:2357-2359) feed only Surface.vn*, which rasterize_triangle_15 never reads. */
    }
    return NULL;
}
/* DRAW of the sorted list into one row band, render.rs:2547-2572 (one job per thread of the all-cores baseline) */
struct Surface_;
typedef struct {
    FB fb; const void* surfaces; const uint32_t* order; uint32_t ns, n_op;
    const B32Face* faces; const B32Texture15* textures; const B32Texture* textures8; uint32_t nt; const B32Settings* st; int rc;
    const float* ylo; const float* yhi;         /* row extents per surface (all-cores schedule), or NULL */
} DrawJob;
static void* draw_range(void* arg);

/* CULL / SETUP of a face range, render.rs:2364-2516 (one job per thread of the all-cores baseline, each into buffers of its own that are
 * concatenated in range order afterwards -- the reference's single loop produces the surfaces in face order; the whole range otherwise) */
typedef struct {
    const B32Vertex* vertices; uint32_t nv; const B32Face* faces; uint32_t f0, f1;
    const B32Texture15* textures; const B32Texture* textures8; uint32_t nt; const B32Settings* st; const B32Fog* fog;
    const V3* cam_space; const V3* projected;
    Surface* out; WireTri* bw; WireTri* fw; uint32_t ns, n_bw, n_fw; int rc;
} CullJob;
static void* cull_range(void* arg) {
    CullJob* j = (CullJob*)arg;
    const B32Vertex* vertices = j->vertices; const uint32_t nv = j->nv; const B32Face* faces = j->faces;
    const B32Texture15* textures = j->textures; const B32Texture* textures8 = j->textures8; const uint32_t nt = j->nt;
    const B32Settings* st = j->st; const B32Fog* fog = j->fog; const V3* cam_space = j->cam_space; const V3* projected = j->projected;
    const int fmt8 = textures8 != NULL;
    for (uint32_t fi = j->f0; fi < j->f1; ++fi) {
        const B32Face* f = &faces[fi];
        if (f->v[0] >= nv || f->v[1] >= nv || f->v[2] >= nv) { j->rc = B32_E_INDEX; break; }    /* index panic :2375-2377 */
        V3 cv1 = cam_space[f->v[0]], cv2 = cam_space[f->v[1]], cv3 = cam_space[f->v[2]];
        if (!st->has_ortho) {                                                                     /* :2381-2385 NEAR_PLANE math.rs:155 */
            if (cv1.z <= K_NEAR_PLANE || cv2.z <= K_NEAR_PLANE || cv3.z <= K_NEAR_PLANE) continue;
        }
        V3 v1 = projected[f->v[0]], v2 = projected[f->v[1]], v3_ = projected[f->v[2]];
        float signed_area = (v2.x - v1.x) * (v3_.y - v1.y) - (v3_.x - v1.x) * (v2.y - v1.y);     /* :2393 */
        int is_backface = signed_area <= 0.0f;
        /* geometric normal (:2397-2399) only feeds Surface.normal, never read by the fill */
        int has_transparency;                                                                     /* :2403-2415 */
        {
            int have_tex = f->texture_id != B32_NO_TEXTURE && f->texture_id < nt;
            uint32_t tex_blend = !have_tex ? B32_BLEND_OPAQUE : (fmt8 ? textures8[f->texture_id].blend_mode : textures[f->texture_id].blend_mode);
            if (have_tex && tex_blend != B32_BLEND_OPAQUE) has_transparency = 1;
            else if (!fmt8 && f->blend_mode != B32_BLEND_OPAQUE) has_transparency = 1;            /* 8-bit: render.rs:2077-2081 ignores it */
            else has_transparency = f->editor_alpha < 255;
            if (fmt8) has_transparency = 0;             /* computed but never used by render_mesh: one list, one sort (:2175-2184) */
        }
        Col c1 = { vertices[f->v[0]].r, vertices[f->v[0]].g, vertices[f->v[0]].b, vertices[f->v[0]].blend };
        Col c2 = { vertices[f->v[1]].r, vertices[f->v[1]].g, vertices[f->v[1]].b, vertices[f->v[1]].blend };
        Col c3 = { vertices[f->v[2]].r, vertices[f->v[2]].g, vertices[f->v[2]].b, vertices[f->v[2]].blend };
        if (fog) {                                                                                /* :2419-2442 */
            if (cv1.z > fog->cull_distance && cv2.z > fog->cull_distance && cv3.z > fog->cull_distance) continue;
            Col fc = { fog->r, fog->g, fog->b, fog->blend };
            c1 = apply_fog_to_color(c1, fc, calculate_fog_factor(cv1.z, fog->start, fog->falloff));
            c2 = apply_fog_to_color(c2, fc, calculate_fog_factor(cv2.z, fog->start, fog->falloff));
            c3 = apply_fog_to_color(c3, fc, calculate_fog_factor(cv3.z, fog->start, fog->falloff));
        }
        Surface s;
        memset(&s, 0, sizeof s);
        s.face_idx = fi; s.black_transparent = f->black_transparent; s.has_transparency = (uint8_t)has_transparency;
        s.blend_mode = f->blend_mode; s.editor_alpha = f->editor_alpha;
        const B32Vertex *A = &vertices[f->v[0]], *B = &vertices[f->v[1]], *C = &vertices[f->v[2]];
        if (is_backface) {                                                                        /* :2445-2479 */
            if (!st->xray_mode) { WireTri w = { v1, v2, v3_ }; j->bw[j->n_bw++] = w; }
            if (!(!st->backface_cull || st->xray_mode)) continue;
            s.v1 = v1; s.v2 = v3_; s.v3 = v2;
            s.w1 = v3p(A->pos); s.w2 = v3p(C->pos); s.w3 = v3p(B->pos);
            s.wn1 = v3scale(v3p(A->normal), -1.0f); s.wn2 = v3scale(v3p(C->normal), -1.0f); s.wn3 = v3scale(v3p(B->normal), -1.0f);
            memcpy(s.uv1, A->uv, 8); memcpy(s.uv2, C->uv, 8); memcpy(s.uv3, B->uv, 8);
            s.vc1 = c1; s.vc2 = c3; s.vc3 = c2;
        } else {                                                                                  /* :2481-2507 */
            s.v1 = v1; s.v2 = v2; s.v3 = v3_;
            s.w1 = v3p(A->pos); s.w2 = v3p(B->pos); s.w3 = v3p(C->pos);
            s.wn1 = v3p(A->normal); s.wn2 = v3p(B->normal); s.wn3 = v3p(C->normal);
            memcpy(s.uv1, A->uv, 8); memcpy(s.uv2, B->uv, 8); memcpy(s.uv3, C->uv, 8);
            s.vc1 = c1; s.vc2 = c2; s.vc3 = c3;
            if (st->wireframe_overlay) { WireTri w = { v1, v2, v3_ }; j->fw[j->n_fw++] = w; }   /* :2509-2511 */
        }
        j->out[j->ns++] = s;
    }

    return NULL;
}
/* per surface: its sort key (render.rs:2529), its class, and the rows its bounding box can touch -- compact arrays, so that the partition,
 * the NaN scan and, in the all-cores schedule, every row band's walk of the sorted list do not drag the 200-byte surfaces through the
 * cache (a band thread skips a surface whose rows lie outside its band: rasterize_triangle_15 would clip its box to nothing) */
typedef struct { const void* surfaces; uint32_t i0, i1; float* key; uint8_t* tr; float* ylo; float* yhi; } PrepJob;
static void* prep_range(void* arg);
typedef struct { void* dst; const void* src; size_t n; } CopyJob;
static void* copy_job(void* arg) { const CopyJob* c = (const CopyJob*)arg; if (c->n) memcpy(c->dst, c->src, c->n); return NULL; }

static int render_mesh_impl(uint8_t* fb_pixels, float* fb_zbuffer, uint32_t width, uint32_t height,
                            const B32Vertex* vertices, uint32_t nv,
                            const B32Face* faces, uint32_t nf,
                            const B32Texture15* textures, const B32Texture* textures8, uint32_t nt,
                            const B32Camera* camera, const B32Settings* st, const B32Fog* fog,
                            B32Timings* timings, B32OracleDump* dump) {
    if (!fb_pixels || !camera || !st || (nv && !vertices) || (nf && !faces) || (nt && !textures && !textures8)) return B32_E_ARG;
    if (st->use_zbuffer && !fb_zbuffer) return B32_E_ARG;
    if (st->shading != B32_SHADE_NONE)
        for (uint32_t i = 0; i < st->n_lights; ++i)
            if (st->lights[i].enabled && st->lights[i].type > B32_LIGHT_SPOT) return B32_E_ARG;
    FB fb = { fb_pixels, fb_zbuffer, width, height, 0, g_band_y0, g_band_y1 };

    /* TRANSFORM, :2313-2362 */
    V3* cam_space = (V3*)malloc(sizeof(V3) * (nv ? nv : 1));
    V3* projected = (V3*)malloc(sizeof(V3) * (nv ? nv : 1));
    {
        unr_init();                                           /* (before any thread touches the table) */
        const uint32_t T = (g_threads > 1 && nv >= 4096) ? (uint32_t)g_threads : 1u;
        TransformJob jobs[256]; pthread_t th[256];
        for (uint32_t t = 0; t < T; ++t) {
            TransformJob j = { vertices, (uint32_t)((uint64_t)nv * t / T), (uint32_t)((uint64_t)nv * (t + 1) / T), camera, st, width, height,
                               cam_space, projected, dump };
            jobs[t] = j;
        }
        if (T == 1) transform_range(&jobs[0]);
        else {
            for (uint32_t t = 0; t < T; ++t) pthread_create(&th[t], NULL, transform_range, &jobs[t]);
            for (uint32_t t = 0; t < T; ++t) pthread_join(th[t], NULL);
        }
    }

    /* CULL / SETUP, :2364-2516 */
    Surface* surfaces = (Surface*)malloc(sizeof(Surface) * (nf ? nf : 1));
    WireTri* backface_wireframes = (WireTri*)malloc(sizeof(WireTri) * (nf ? nf : 1));
    WireTri* frontface_wireframes = (WireTri*)malloc(sizeof(WireTri) * (nf ? nf : 1));
    uint32_t ns = 0, n_bw = 0, n_fw = 0;
    int rc = B32_OK;
    {
        const uint32_t T = (g_threads > 1 && nf >= 8192) ? (uint32_t)(g_threads > 64 ? 64 : g_threads) : 1u;
        CullJob jobs[64]; pthread_t th[64];
        for (uint32_t t = 0; t < T; ++t) {
            CullJob j = { vertices, nv, faces, (uint32_t)((uint64_t)nf * t / T), (uint32_t)((uint64_t)nf * (t + 1) / T), textures, textures8, nt, st, fog,
                          cam_space, projected, surfaces, backface_wireframes, frontface_wireframes, 0, 0, 0, B32_OK };
            if (T > 1) {
                const size_t cnt = (size_t)(j.f1 - j.f0) + 1;
                j.out = (Surface*)malloc(sizeof(Surface) * cnt); j.bw = (WireTri*)malloc(sizeof(WireTri) * cnt); j.fw = (WireTri*)malloc(sizeof(WireTri) * cnt);
            }
            jobs[t] = j;
        }
        if (T == 1) cull_range(&jobs[0]);
        else {
            for (uint32_t t = 0; t < T; ++t) pthread_create(&th[t], NULL, cull_range, &jobs[t]);
            for (uint32_t t = 0; t < T; ++t) pthread_join(th[t], NULL);
        }
        for (uint32_t t = 0; t < T && !rc; ++t) rc = jobs[t].rc;           /* (the first failing range in face order, like the sequential loop) */
        if (T == 1) { ns = jobs[0].ns; n_bw = jobs[0].n_bw; n_fw = jobs[0].n_fw; }
        else {
            CopyJob cj[64];
            for (uint32_t t = 0; t < T; ++t) {      /* ordered concatenation, copied by the same threads */
                if (!rc) {
                    memcpy(backface_wireframes + n_bw, jobs[t].bw, sizeof(WireTri) * jobs[t].n_bw);
                    memcpy(frontface_wireframes + n_fw, jobs[t].fw, sizeof(WireTri) * jobs[t].n_fw);
                }
                CopyJob c = { surfaces + ns, jobs[t].out, rc ? 0 : sizeof(Surface) * (size_t)jobs[t].ns }; cj[t] = c;
                ns += jobs[t].ns; n_bw += jobs[t].n_bw; n_fw += jobs[t].n_fw;
            }
            for (uint32_t t = 0; t < T; ++t) pthread_create(&th[t], NULL, copy_job, &cj[t]);
            for (uint32_t t = 0; t < T; ++t) pthread_join(th[t], NULL);
            for (uint32_t t = 0; t < T; ++t) { free(jobs[t].out); free(jobs[t].bw); free(jobs[t].fw); }
        }
        if (rc) goto done;
    }

    /* SORT, :2518-2545 */
    {
        uint32_t* order = (uint32_t*)malloc(sizeof(uint32_t) * (ns ? ns : 1));
        uint32_t* tmp = (uint32_t*)malloc(sizeof(uint32_t) * (ns ? ns : 1));
        float* key = (float*)malloc(sizeof(float) * (ns ? ns : 1));
        float* ylo = (float*)malloc(sizeof(float) * (ns ? ns : 1));
        float* yhi = (float*)malloc(sizeof(float) * (ns ? ns : 1));
        uint8_t* trc = (uint8_t*)malloc(ns ? ns : 1);
        uint32_t n_op = 0, n_tr = 0;
        {
            const uint32_t T = (g_threads > 1 && ns >= 8192) ? (uint32_t)(g_threads > 64 ? 64 : g_threads) : 1u;
            PrepJob pj[64]; pthread_t pth[64];
            for (uint32_t t = 0; t < T; ++t) { PrepJob q = { surfaces, (uint32_t)((uint64_t)ns * t / T), (uint32_t)((uint64_t)ns * (t + 1) / T), key, trc, ylo, yhi }; pj[t] = q; }
            if (T == 1) prep_range(&pj[0]);
            else { for (uint32_t t = 0; t < T; ++t) pthread_create(&pth[t], NULL, prep_range, &pj[t]); for (uint32_t t = 0; t < T; ++t) pthread_join(pth[t], NULL); }
        }
        for (uint32_t i = 0; i < ns; ++i) if (!trc[i]) order[n_op++] = i;                        /* partition(!has_transparency), :2522-2523 */
        for (uint32_t i = 0; i < ns; ++i) if (trc[i]) order[n_op + n_tr++] = i;
        /* partial_cmp().unwrap() panics on NaN as soon as a comparison sees one (any list with >= 2 elements) */
        int nan_tr = 0, nan_op = 0;
        for (uint32_t i = 0; i < n_op; ++i) if (key[order[i]] != key[order[i]]) nan_op = 1;
        for (uint32_t i = n_op; i < ns; ++i) if (key[order[i]] != key[order[i]]) nan_tr = 1;
        if ((nan_tr && n_tr >= 2) || (!st->use_zbuffer && nan_op && n_op >= 2)) rc = B32_E_NAN_KEY;
        if (!rc) {
            merge_sort_desc_mt(order + n_op, tmp + n_op, key, n_tr, (uint32_t)g_threads);         /* :2527-2532 */
            if (!st->use_zbuffer) merge_sort_desc_mt(order, tmp, key, n_op, (uint32_t)g_threads); /* :2535-2542 */
            if (timings) timings->triangles_drawn = ns;
            if (dump) { dump->n_drawn = ns; dump->n_opaque = n_op; if (dump->draw_order) for (uint32_t i = 0; i < ns; ++i) dump->draw_order[i] = surfaces[order[i]].face_idx; }
            /* DRAW, :2547-2572 */
            if (!st->wireframe_overlay) {
                const uint32_t by0 = fb.band_y0 < height ? fb.band_y0 : height, by1 = fb.band_y1 < height ? fb.band_y1 : height;
                const uint32_t rows = by1 > by0 ? by1 - by0 : 0;
                const uint32_t T = (g_threads > 1 && rows >= (uint32_t)g_threads && ns >= 64) ? (uint32_t)g_threads : 1u;
                DrawJob jobs[256]; pthread_t th[256];
                for (uint32_t t = 0; t < T; ++t) {
                    DrawJob j = { fb, surfaces, order, ns, n_op, faces, textures, textures8, nt, st, B32_OK, T > 1 ? ylo : NULL, T > 1 ? yhi : NULL };
                    j.fb.fragments = 0;
                    if (T > 1) { j.fb.band_y0 = by0 + (uint32_t)((uint64_t)rows * t / T); j.fb.band_y1 = by0 + (uint32_t)((uint64_t)rows * (t + 1) / T); }
                    jobs[t] = j;
                }
                if (T == 1) draw_range(&jobs[0]);
                else {
                    for (uint32_t t = 0; t < T; ++t) pthread_create(&th[t], NULL, draw_range, &jobs[t]);
                    for (uint32_t t = 0; t < T; ++t) pthread_join(th[t], NULL);
                }
                for (uint32_t t = 0; t < T; ++t) { fb.fragments += jobs[t].fb.fragments; if (jobs[t].rc && !rc) rc = jobs[t].rc; }
            }
        }
        free(order); free(tmp); free(key); free(ylo); free(yhi); free(trc);
    }
    if (timings) timings->fragments = fb.fragments;
    /* WIREFRAME, :2574-2635 */
    if (!rc && st->backface_cull && st->backface_wireframe) {
        uint32_t ne = 0;
        WireEdge* ue = unique_edges(backface_wireframes, n_bw, &ne);
        for (uint32_t i = 0; i < ne && !rc; ++i)
            rc = b32o_draw_line_3d(fb.pixels, fb.zbuffer, width, height, ue[i].x0, ue[i].y0, ue[i].z0, ue[i].x1, ue[i].y1, ue[i].z1, 80, 80, 100);
        free(ue);
    }
    if (!rc && st->wireframe_overlay && n_fw) {
        uint32_t ne = 0;
        WireEdge* ue = unique_edges(frontface_wireframes, n_fw, &ne);
        for (uint32_t i = 0; i < ne && !rc; ++i)
            rc = b32o_draw_line(fb.pixels, width, height, ue[i].x0, ue[i].y0, ue[i].x1, ue[i].y1, 200, 200, 220);
        free(ue);
    }
done:
    free(surfaces); free(cam_space); free(projected); free(backface_wireframes); free(frontface_wireframes);
    return rc;
}

static void* prep_range(void* arg) {
    const PrepJob* j = (const PrepJob*)arg;
    const Surface* surfaces = (const Surface*)j->surfaces;
    for (uint32_t i = j->i0; i < j->i1; ++i) {
        const Surface* s = &surfaces[i];
        j->key[i] = center_z(s); j->tr[i] = s->has_transparency;
        const float y1 = s->v1.y, y2 = s->v2.y, y3 = s->v3.y;
        if (y1 - y1 == 0.0f && y2 - y2 == 0.0f && y3 - y3 == 0.0f) {          /* all finite */
            j->ylo[i] = y1 < y2 ? (y1 < y3 ? y1 : y3) : (y2 < y3 ? y2 : y3);
            j->yhi[i] = y1 > y2 ? (y1 > y3 ? y1 : y3) : (y2 > y3 ? y2 : y3);
        } else { j->ylo[i] = -3.0e38f; j->yhi[i] = 3.0e38f; }                 /* (never skipped) */
    }
    return NULL;
}
static void* draw_range(void* arg) {
    DrawJob* j = (DrawJob*)arg;
    const Surface* surfaces = (const Surface*)j->surfaces;
    const int fmt8 = j->textures8 != NULL;
    for (uint32_t i = 0; i < j->ns && !j->rc; ++i) {
        const uint32_t si = j->order[i];
        /* a row band of the all-cores schedule: boxes entirely above or below it draw nothing here (min_y = min(..).max(0) as usize,
         * max_y = (max(..) + 1.0).min(h) as usize, render.rs:1457-1458, clipped to the band) */
        if (j->ylo && (j->yhi[si] + 1.0f < (float)j->fb.band_y0 || j->ylo[si] >= (float)j->fb.band_y1)) continue;
        const Surface* s = &surfaces[si];
        uint32_t tid = j->faces[s->face_idx].texture_id;
        if (fmt8) {
            const B32Texture* tex8 = (tid != B32_NO_TEXTURE && tid < j->nt) ? &j->textures8[tid] : NULL;       /* :2196-2198 */
            j->rc = rasterize_triangle8(&j->fb, s, tex8, j->st);
            continue;
        }
        const B32Texture15* tex = (tid != B32_NO_TEXTURE && tid < j->nt) ? &j->textures[tid] : NULL;  /* textures.get(id) :2554-2556 */
        j->rc = rasterize_triangle_15(&j->fb, s, tex, s->blend_mode, s->black_transparent, j->st, i >= j->n_op);
    }
    return NULL;
}

EXPORT int b32o_render_mesh_15(uint8_t* fb_pixels, float* fb_zbuffer, uint32_t width, uint32_t height,
                               const B32Vertex* vertices, uint32_t nv, const B32Face* faces, uint32_t nf,
                               const B32Texture15* textures, uint32_t nt,
                               const B32Camera* camera, const B32Settings* st, const B32Fog* fog,
                               B32Timings* timings, B32OracleDump* dump) {
    if (nt && !textures) return B32_E_ARG;
    static const B32Texture15 none15 = { 0, 0, 0, 0, NULL };
    return render_mesh_impl(fb_pixels, fb_zbuffer, width, height, vertices, nv, faces, nf, textures ? textures : &none15, NULL, nt,
                            camera, st, fog, timings, dump);
}
/* render_mesh, render.rs:1971-2264: no fog parameter */
EXPORT int b32o_render_mesh(uint8_t* fb_pixels, float* fb_zbuffer, uint32_t width, uint32_t height,
                            const B32Vertex* vertices, uint32_t nv, const B32Face* faces, uint32_t nf,
                            const B32Texture* textures, uint32_t nt,
                            const B32Camera* camera, const B32Settings* st,
                            B32Timings* timings, B32OracleDump* dump) {
    if (nt && !textures) return B32_E_ARG;
    static const B32Texture none8 = { 0, 0, 0, 0, NULL };
    return render_mesh_impl(fb_pixels, fb_zbuffer, width, height, vertices, nv, faces, nf, NULL, textures ? textures : &none8, nt,
                            camera, st, NULL, timings, dump);
}

/* ------------------------------------------------------------------ the steps around the mesh draw */
/* Color::lerp, types.rs:812-821 + Framebuffer::clear_gradient, render.rs:58-77 */
EXPORT void b32o_fb_clear_gradient(uint8_t* pixels, float* zbuffer, uint32_t width, uint32_t height, const uint8_t top[4], const uint8_t bottom[4]) {
    for (uint32_t y = 0; y < height; ++y) {
        float t = height > 1 ? (float)y / (float)(height - 1) : 0.0f;
        float tc = rclamp(t, 0.0f, 1.0f), inv_t = 1.0f - tc;
        uint8_t c[4];
        for (int i = 0; i < 3; ++i) c[i] = f2u8_sat((float)top[i] * inv_t + (float)bottom[i] * tc);
        c[3] = top[3] == B32_BLEND_ERASE ? 0 : 255;                                            /* blend: self.blend -> to_bytes */
        for (uint32_t x = 0; x < width; ++x) {
            size_t idx = ((size_t)y * width + x) * 4;
            memcpy(&pixels[idx], c, 4);
            if (zbuffer) zbuffer[(size_t)y * width + x] = 3.40282347e+38f;
        }
    }
}
/* rasterize_skybox_triangle, render.rs:251-298 */
static void rasterize_skybox_triangle(uint8_t* pixels, uint32_t width, uint32_t height, const float p0[2], const float p1[2], const float p2[2],
                                      const uint8_t* c0, const uint8_t* c1, const uint8_t* c2) {
    uint64_t min_x = f2usize_sat(rmax(rmin(rmin(p0[0], p1[0]), p2[0]), 0.0f));
    uint64_t max_x = f2usize_sat(rmin(rmax(rmax(p0[0], p1[0]), p2[0]), (float)width - 1.0f));
    uint64_t min_y = f2usize_sat(rmax(rmin(rmin(p0[1], p1[1]), p2[1]), 0.0f));
    uint64_t max_y = f2usize_sat(rmin(rmax(rmax(p0[1], p1[1]), p2[1]), (float)height - 1.0f));
    if (min_x > max_x || min_y > max_y) return;
    float denom = (p1[1] - p2[1]) * (p0[0] - p2[0]) + (p2[0] - p1[0]) * (p0[1] - p2[1]);
    if (fabsf(denom) < 0.0001f) return;
    float inv_denom = 1.0f / denom;
    for (uint64_t y = min_y; y <= max_y; ++y)
        for (uint64_t x = min_x; x <= max_x; ++x) {
            float px = (float)x + 0.5f, py = (float)y + 0.5f;
            float w0 = ((p1[1] - p2[1]) * (px - p2[0]) + (p2[0] - p1[0]) * (py - p2[1])) * inv_denom;
            float w1 = ((p2[1] - p0[1]) * (px - p2[0]) + (p0[0] - p2[0]) * (py - p2[1])) * inv_denom;
            float w2 = 1.0f - w0 - w1;
            if (w0 >= 0.0f && w1 >= 0.0f && w2 >= 0.0f) {
                size_t idx = ((size_t)y * width + x) * 4;
                for (int i = 0; i < 3; ++i) pixels[idx + i] = f2u8_sat((float)c0[i] * w0 + (float)c1[i] * w1 + (float)c2[i] * w2);
                pixels[idx + 3] = 255;
            }
        }
}
typedef struct { float pos[3]; uint8_t r, g, b, blend; } SkyVertex;
/* step 1 of Framebuffer::render_skybox, render.rs:81-134 (the mesh itself comes from Skybox::generate_mesh on the host) */
EXPORT int b32o_render_skybox_mesh(uint8_t* pixels, uint32_t width, uint32_t height, const SkyVertex* v, uint32_t nv,
                                   const uint32_t* faces, uint32_t nf, const B32Camera* camera) {
    V3 cpos = v3p(camera->position), bx = v3p(camera->basis_x), by = v3p(camera->basis_y), bz = v3p(camera->basis_z);
    float* proj = (float*)malloc(sizeof(float) * 3 * (nv ? nv : 1));
    for (uint32_t i = 0; i < nv; ++i) {
        V3 cam_space = perspective_transform(v3sub(v3p(v[i].pos), cpos), bx, by, bz);
        if (cam_space.z <= 0.1f) { proj[3 * i] = proj[3 * i + 1] = proj[3 * i + 2] = NAN; continue; }
        V3 screen = project_float(cam_space, width, height);
        proj[3 * i] = screen.x; proj[3 * i + 1] = screen.y; proj[3 * i + 2] = cam_space.z;
    }
    int rc = B32_OK;
    for (uint32_t f = 0; f < nf && !rc; ++f) {
        const uint32_t i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
        if (i0 >= nv || i1 >= nv || i2 >= nv) { rc = B32_E_INDEX; break; }
        const float *p0 = &proj[3 * i0], *p1 = &proj[3 * i1], *p2 = &proj[3 * i2];
        if (p0[0] != p0[0] || p1[0] != p1[0] || p2[0] != p2[0]) continue;
        float signed_area = (p1[0] - p0[0]) * (p2[1] - p0[1]) - (p2[0] - p0[0]) * (p1[1] - p0[1]);
        if (signed_area >= 0.0f) continue;
        rasterize_skybox_triangle(pixels, width, height, p0, p1, p2, &v[i0].r, &v[i1].r, &v[i2].r);
    }
    free(proj);
    return rc;
}
/* draw_star_diamond + set_pixel_safe, render.rs:199-246 */
EXPORT void b32o_draw_star_diamond(uint8_t* pixels, uint32_t width, uint32_t height, int32_t cx, int32_t cy, float size, const uint8_t rgb[3]) {
    FB fb = { pixels, NULL, width, height, 0, g_band_y0, g_band_y1 };
    int32_t s = f2i32_sat(rmax(size, 1.0f));
    set_pixel_rgb(&fb, cx, cy, rgb[0], rgb[1], rgb[2]);
    if (s >= 2) {
        uint8_t d[3]; for (int i = 0; i < 3; ++i) d[i] = f2u8_sat((float)rgb[i] * 0.7f);
        /* i32 `cx - 1` etc.: release builds wrap (render.rs:219-222) */
        set_pixel_rgb(&fb, wrap_add(cx, -1), cy, d[0], d[1], d[2]); set_pixel_rgb(&fb, wrap_add(cx, 1), cy, d[0], d[1], d[2]);
        set_pixel_rgb(&fb, cx, wrap_add(cy, -1), d[0], d[1], d[2]); set_pixel_rgb(&fb, cx, wrap_add(cy, 1), d[0], d[1], d[2]);
    }
    if (s >= 3) {
        uint8_t d[3]; for (int i = 0; i < 3; ++i) d[i] = f2u8_sat((float)rgb[i] * 0.4f);
        set_pixel_rgb(&fb, wrap_add(cx, -2), cy, d[0], d[1], d[2]); set_pixel_rgb(&fb, wrap_add(cx, 2), cy, d[0], d[1], d[2]);
        set_pixel_rgb(&fb, cx, wrap_add(cy, -2), d[0], d[1], d[2]); set_pixel_rgb(&fb, cx, wrap_add(cy, 2), d[0], d[1], d[2]);
    }
}

/* Reference unit-test helpers (math.rs:779-807) exposed so tests can replay them through this file. */
EXPORT float b32o_vec3_dot(const float a[3], const float b[3]) { return v3dot(v3p(a), v3p(b)); }
EXPORT void b32o_vec3_cross(const float a[3], const float b[3], float out[3]) {   /* math.rs:27-33 */
    out[0] = a[1] * b[2] - a[2] * b[1]; out[1] = a[2] * b[0] - a[0] * b[2]; out[2] = a[0] * b[1] - a[1] * b[0];
}
EXPORT float b32o_fixed_to_f32(int32_t v) { return fixed_to_f32(v); }

// b32_device.h — shared declarations of the gfx950 rasterizer kernels.
//
// Compiled ONLY for gfx950 with -ffp-contract=off (Rust never fuses a*b+c) and HIP's default correctly-rounded
// f32 divide / sqrt; f32 denormals are kept.  b32_selftest_f32 proves those three properties on the device
// before any parity claim is trusted (tests/test_gpu_parity.py::test_device_f32_semantics).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/b32raster.h"

// Experiment switches that produce WRONG frames on purpose (they isolate the cost of one mechanism; tools/exp_variants.py builds) may only
// be compiled into an explicitly marked experiment build, never into the product library.
#if (defined(B32_EXP_BIN_NO_ATOMICS) || defined(B32_EXP_BIN_WG_ATOMICS) || defined(B32_EXP_BIN_NO_LIST_STORE) || defined(B32_EXP_WIRE_STAGE)) && !defined(B32_EXPERIMENT)
#error "B32_EXP_* switches that break the frame need -DB32_EXPERIMENT (an experiment build, loaded through B32_LIB only)"
#endif

namespace b32 {

// ---------------------------------------------------------------- geometry constants
constexpr int TILE_W = 64;          // screen tile owned by one workgroup pass of k_fill
constexpr int TILE_H = 64;
constexpr int TILE_STRIDE = 72;     // LDS row stride in dwords (64 + 8: an 8x8 lane block maps to 32 distinct banks per half-wave)
constexpr int FILL_THREADS = 1024;  // 16 waves: 4 per SIMD
constexpr int FILL_WAVES = FILL_THREADS / 64;
constexpr uint32_t KEY_INVALID = 0xFFFFFFFFu;

// ---------------------------------------------------------------- the reference's numeric literals, by name
// Device code uses the algorithm's literals only through these names.  b32_device_constants() reads them back FROM DEVICE CODE
// (k_constants in b32_setup.hip, with the UNR table and the dither matrix as the kernels see them) and
// tests/test_gpu_parity.py::test_device_constants_are_the_reference_text compares every one with tests/golden/ref_constants.json,
// which tests/golden/pin_constants.py derives from the reference's own text.  B32_CONSTANTS gives the fixture key of each.
namespace K {
constexpr int      FRAC_BITS = 12;                       // fixed.rs:110
constexpr float    ONE_F = (float)(1 << FRAC_BITS);      // ONE_32 as f32, fixed.rs:111,126
constexpr uint32_t UNR_ENTRIES = 257, UNR_INDEX_OFFSET = 256, UNR_NUMERATOR = 262144, UNR_ROUND_ADD = 1, UNR_ROUND_DIV = 2;   // fixed.rs:20-31
constexpr int32_t  UNR_SUBTRACT = 257;
constexpr uint32_t DIV_D16_SHIFT = 16, DIV_INDEX_SHIFT = 7, DIV_INDEX_MAX = 256, DIV_NR1_SHIFT = 8, DIV_NR2_SHIFT = 8, DIV_SHIFT_BASE = 36;   // fixed.rs:197-212
constexpr uint64_t DIV_INDEX_BIAS = 0x7FC0, DIV_U_ADD = 0x101, DIV_NR1_CONST = 0x2000080, DIV_NR2_CONST = 0x80;
constexpr float    PF_DISTANCE = 5.0f, PF_SCALE = 4.0f, PF_VIEWPORT_DIV = 2.0f, PF_VIEWPORT_FRAC = 0.75f;    // project_to_screen, fixed.rs:396-398
constexpr int32_t  PF_DENOM_GUARD = 256;                 // fixed.rs:406
constexpr float    P_DISTANCE = 5.0f, P_US_SUB = 1.0f, P_VIEWPORT_DIV = 2.0f, P_VIEWPORT_FRAC = 0.75f, P_DENOM_GUARD = 0.001f;   // project, math.rs:118-127
constexpr float    NEAR_PLANE = 0.1f;                    // math.rs:155
constexpr float    MESH_DISTANCE = 5.0f;                 // render.rs:2344
constexpr float    AREA_EPS = 0.00001f, ERR = -0.0001f;  // render.rs:1501, 1541 (8-bit fill: 1258, 1302)
constexpr uint32_t MOD_DIV = 128, MOD_MAX = 255;         // render.rs:1624
constexpr float    SHADE_LO = 0.0f, SHADE_HI = 2.0f, SHADE_MAX = 255.0f;   // render.rs:1643
constexpr int      DITHER_SHIFT = 3, DITHER_LO = 0, DITHER_HI = 31, NODITHER_SHIFT = 3;   // render.rs:1177, 1653
constexpr int      DITHER8_EXPAND_SHIFT = 3;             // apply_dither, render.rs:1196
constexpr uint32_t EXPAND5_SHL = 3, EXPAND5_SHR = 2;     // render.rs:1162
constexpr int      BLEND_IN_SHIFT = 3, BLEND_AVG_DIV = 2, BLEND_HI = 31, BLEND_LO = 0, BLEND_QUARTER_DIV = 4, BLEND_OUT_SHIFT = 3;   // render.rs:1095-1144
constexpr float    LIGHT_MIN_DIST = 0.001f, LIGHT_COLOR_DIV = 255.0f, LIGHT_TOTAL_MAX = 1.0f;   // render.rs:1030, 1062, 1070
constexpr uint32_t C15_TRANSPARENT = 0x0000, C15_BLACK_DRAWABLE = 0x8000, C15_WHITE = 0x7FFF, C15_SEMI_BIT = 0x8000;   // types.rs:24-53
constexpr uint32_t C15_R_SHIFT = 10, C15_G_SHIFT = 5, C15_CHANNEL_MAX = 31;                                             // types.rs:42-43
}  // namespace K
// X(fixture key, F = f32 / I = integer, value)
#define B32_CONSTANTS(X)                                                                                                              \
    X("fixed.frac_bits", I, K::FRAC_BITS)                                                                                             \
    X("unr.entries", I, K::UNR_ENTRIES) X("unr.index_offset", I, K::UNR_INDEX_OFFSET) X("unr.numerator", I, K::UNR_NUMERATOR)         \
    X("unr.round_add", I, K::UNR_ROUND_ADD) X("unr.round_div", I, K::UNR_ROUND_DIV) X("unr.subtract", I, K::UNR_SUBTRACT)             \
    X("div_unr.d16_shift", I, K::DIV_D16_SHIFT) X("div_unr.index_bias", I, K::DIV_INDEX_BIAS) X("div_unr.index_shift", I, K::DIV_INDEX_SHIFT) \
    X("div_unr.index_max", I, K::DIV_INDEX_MAX) X("div_unr.u_add", I, K::DIV_U_ADD) X("div_unr.nr1_const", I, K::DIV_NR1_CONST)       \
    X("div_unr.nr1_shift", I, K::DIV_NR1_SHIFT) X("div_unr.nr2_const", I, K::DIV_NR2_CONST) X("div_unr.nr2_shift", I, K::DIV_NR2_SHIFT) \
    X("div_unr.shift_base", I, K::DIV_SHIFT_BASE)                                                                                     \
    X("project_fixed.distance", F, K::PF_DISTANCE) X("project_fixed.scale", F, K::PF_SCALE)                                           \
    X("project_fixed.viewport_div", F, K::PF_VIEWPORT_DIV) X("project_fixed.viewport_frac", F, K::PF_VIEWPORT_FRAC)                   \
    X("project_fixed.denom_guard", I, K::PF_DENOM_GUARD)                                                                              \
    X("project.distance", F, K::P_DISTANCE) X("project.us_sub", F, K::P_US_SUB) X("project.viewport_div", F, K::P_VIEWPORT_DIV)       \
    X("project.viewport_frac", F, K::P_VIEWPORT_FRAC) X("project.denom_guard", F, K::P_DENOM_GUARD)                                   \
    X("near_plane", F, K::NEAR_PLANE) X("mesh.distance", F, K::MESH_DISTANCE)                                                         \
    X("fill.area_eps", F, K::AREA_EPS) X("fill.err", F, K::ERR) X("fill8.area_eps", F, K::AREA_EPS) X("fill8.err", F, K::ERR)         \
    X("fill.modulate_div", I, K::MOD_DIV) X("fill.modulate_max", I, K::MOD_MAX)                                                       \
    X("fill.shade_clamp_lo", F, K::SHADE_LO) X("fill.shade_clamp_hi", F, K::SHADE_HI) X("fill.shade_max", F, K::SHADE_MAX)            \
    X("fill.nodither_shift", I, K::NODITHER_SHIFT)                                                                                    \
    X("dither.shift", I, K::DITHER_SHIFT) X("dither.clamp_lo", I, K::DITHER_LO) X("dither.clamp_hi", I, K::DITHER_HI)                 \
    X("dither8.shift", I, K::DITHER_SHIFT) X("dither8.clamp_hi", I, K::DITHER_HI) X("dither8.expand_shift", I, K::DITHER8_EXPAND_SHIFT) \
    X("expand5.shl", I, K::EXPAND5_SHL) X("expand5.shr", I, K::EXPAND5_SHR)                                                           \
    X("blend555.in_shift", I, K::BLEND_IN_SHIFT) X("blend555.average_div", I, K::BLEND_AVG_DIV) X("blend555.clamp_hi", I, K::BLEND_HI) \
    X("blend555.clamp_lo", I, K::BLEND_LO) X("blend555.quarter_div", I, K::BLEND_QUARTER_DIV) X("blend555.out_shift", I, K::BLEND_OUT_SHIFT) \
    X("light.min_dist", F, K::LIGHT_MIN_DIST) X("light.color_div", F, K::LIGHT_COLOR_DIV) X("light.total_max", F, K::LIGHT_TOTAL_MAX) \
    X("color15.transparent", I, K::C15_TRANSPARENT) X("color15.black_drawable", I, K::C15_BLACK_DRAWABLE) X("color15.white", I, K::C15_WHITE) \
    X("color15.semi_bit", I, K::C15_SEMI_BIT) X("color15.r_shift", I, K::C15_R_SHIFT) X("color15.g_shift", I, K::C15_G_SHIFT)         \
    X("color15.channel_max", I, K::C15_CHANNEL_MAX)

// ---------------------------------------------------------------- per-surface records written by k_setup, indexed by face id
// Three arrays, split by consumer so that every reader pulls whole, aligned 32-byte sectors of exactly what it needs:
//   CovRec   32 B  coverage (the tile-list walk: one record per (surface, tile) pair): snapped screen vertices as 6 x i16, inv_area,
//                  bounding box, painter's key, flags.  The edge coefficients a0 = y2 - y3, b0 = x3 - x2, a1 = y3 - y1, b1 = x1 - x3
//                  (render.rs:1507-1510) are recomputed from the vertices in registers: the same f32 subtractions on the same operands.
//   ShadeRec 64 B  shading (one record per run of winner pixels): f32 vertices, inv_area, UVs, vertex colours, texture slot and the
//                  three flags the opaque colour pipeline reads.
//   AuxRec   32 B  1 / v_i.z (perspective-correct UVs, z-buffer depth) and the start values of the literal edge walk: written and read
//                  only in z-buffer mode, with perspective-correct textures, or for F_SLOW surfaces.
// A surface whose vertices do not fit i16 (or are not integers: float projection, orthographic view) carries COV_WIDE in x1 and the
// coverage reads its vertices from the ShadeRec instead.  The fill evaluates w0 / w1 in closed form only when F_SLOW is clear, i.e.
// when k_setup proved every intermediate of the reference's incremental accumulation (render.rs:1706-1712) is an integer of magnitude
// < 2^24, so that accumulate == closed form bit-for-bit.
struct __attribute__((aligned(32))) CovRec {
    uint32_t xy1, xy2, xy3;                    // x | y << 16, each an i16 (xy1's low half == COV_WIDE: see ShadeRec)
    float inv_area;                            // 1.0 / area, render.rs:1504
    uint32_t bbx, bby;                         // min | max << 16 (max exclusive, render.rs:1455-1458)
    uint32_t key;                              // k_setup's painter's radix key (the same word as keys[face])
    uint32_t flags;                            // F_* below
};
static_assert(sizeof(CovRec) == 32, "CovRec layout");
constexpr uint32_t COV_WIDE = 0x8000u;         // i16 -32768 in the x1 slot
struct __attribute__((aligned(64))) ShadeRec {
    float x1, y1, x2, y2;
    float x3, y3, inv_area; uint32_t pk0;      // pk0 = vc1 | texture slot bits 0..7 << 24      (vc = r | g<<8 | b<<16)
    float u1, u2, u3, v1;
    float v2, v3; uint32_t pk1, pk2;           // pk1 = vc2 | texture slot bits 8..15 << 24;  pk2 = vc3 | SH_* << 24
};
static_assert(sizeof(ShadeRec) == 64, "ShadeRec layout");
constexpr uint32_t SH_BLACK_TR = 1u, SH_DITHER = 2u, SH_SLOW = 4u;     // bits 24.. of ShadeRec.pk2
struct __attribute__((aligned(32))) AuxRec {
    float iz1, iz2, iz3, w0_start;             // 1.0 / v_i.z (render.rs:1546-1548); w*_start = render.rs:1517-1518
    float w1_start; uint32_t _pad[3];
};
static_assert(sizeof(AuxRec) == 32, "AuxRec layout");

// The register view every consumer computes with: the six quads of the former 96-byte record, assembled from the compact records.
//   q0 = x3, y3, a0, b0 | q1 = a1, b1, inv_area, bbx | q2 = bby, u1, u2, u3 | q3 = v1, v2, v3, flags | q4 = vc1, vc2, vc3, w0_start
//   q5 = w1_start, iz1, iz2, iz3
struct RecView { uint4 q0, q1, q2, q3, q4, q5; };
__device__ __forceinline__ float i16lo(uint32_t w) { return (float)(int32_t)(int16_t)(w & 0xFFFFu); }
__device__ __forceinline__ float i16hi(uint32_t w) { return (float)((int32_t)w >> 16); }
// edge part (q0, q1.xyz) from the three screen vertices: the reference's own subtractions (render.rs:1507-1510)
__device__ __forceinline__ void view_edges(RecView& r, float x1, float y1, float x2, float y2, float x3, float y3, float inv_area) {
    r.q0 = make_uint4(__float_as_uint(x3), __float_as_uint(y3), __float_as_uint(y2 - y3), __float_as_uint(x3 - x2));
    r.q1.x = __float_as_uint(y3 - y1); r.q1.y = __float_as_uint(x1 - x3); r.q1.z = __float_as_uint(inv_area);
}
// coverage part of the view from a CovRec (c0, c1 = its two quads); false = COV_WIDE (the caller takes the vertices from the ShadeRec)
__device__ __forceinline__ bool view_from_cov(RecView& r, const uint4& c0, const uint4& c1) {
    r.q1.w = c1.x; r.q2.x = c1.y; r.q3.w = c1.w;
    if ((c0.x & 0xFFFFu) == COV_WIDE) return false;
    view_edges(r, i16lo(c0.x), i16hi(c0.x), i16lo(c0.y), i16hi(c0.y), i16lo(c0.z), i16hi(c0.z), __uint_as_float(c0.w));
    return true;
}
__device__ __forceinline__ void view_edges_from_shade(RecView& r, const uint4& s0, const uint4& s1) {
    view_edges(r, __uint_as_float(s0.x), __uint_as_float(s0.y), __uint_as_float(s0.z), __uint_as_float(s0.w), __uint_as_float(s1.x),
               __uint_as_float(s1.y), __uint_as_float(s1.z));
}
__device__ __forceinline__ uint32_t shade_tex_slot(const uint4& s1, const uint4& s3) { return (s1.w >> 24) | ((s3.z >> 24) << 8); }

// CovRec.flags (the shading view rebuilds the bits it needs -- texture slot, F_BLACK_TR, F_DITHER, F_SLOW -- from the ShadeRec)
constexpr uint32_t F_TEX_MASK   = 0xFFFFu;     // texture slot, 0xFFFF = untextured (Color15::WHITE, render.rs:1585): up to 65534 textures per call
constexpr uint32_t F_TEX_NONE   = 0xFFFFu;
constexpr uint32_t F_BLACK_TR   = 1u << 16;    // face.black_transparent
constexpr uint32_t F_BLEND_SHIFT = 17;         // 3 bits: effective blend mode (texture's if bound, render.rs:1450-1452)
constexpr uint32_t F_DITHER     = 1u << 20;    // needs_dither, render.rs:1487-1492
constexpr uint32_t F_SLOW       = 1u << 21;    // replay the incremental edge walk literally
constexpr uint32_t F_TRANSP     = 1u << 22;    // has_transparency, render.rs:2403-2415
constexpr uint32_t F_EMPTY      = 1u << 23;    // bbox empty or |area| < 1e-5: counted in triangles_drawn, draws nothing
constexpr uint32_t F_ALPHA_SHIFT = 24;         // editor_alpha

struct TexDesc { uint32_t width, height, blend_mode, offset; };   // offset into the pooled u16 texel buffer

// Device-side control block, zeroed at the start of every frame.
struct Ctrl {
    uint32_t n_visible;        // surfaces surviving cull (== triangles_drawn)
    uint32_t n_transparent;
    uint32_t nan_opaque, nan_transparent;
    uint32_t err_index;        // a face referenced vertex >= nv
    uint32_t abort;            // set => k_fill draws nothing (error, or pair capacity exceeded)
    uint32_t n_opaque;
    uint32_t n_pairs;          // (tile,class) x surface pairs emitted by binning
    uint32_t pairs_overflow;
    uint32_t tile_cursor;      // persistent-workgroup tile dispenser of k_fill
    uint32_t list_demand;      // direct binning, when a tile region overflowed: the longest opaque tile list of the frame (the host sizes the regions from it)
    uint32_t need_global_sort; // bit 0: a tile list exceeded the LDS sort capacity: redraw with the global depth sort;
                               // bit 1: a tile region of the direct binning overflowed: redraw with larger regions (nothing was drawn either way)
    uint32_t wire_overflow;    // a wireframe edge is >= 2^30 pixels long: the reference's i32 Bresenham state overflows
    uint32_t sticky;           // errors of every frame since the last b32_frame_finish (bit 0 vertex index, 1 NaN sort key, 2 wire edge):
                               // NOT reset at frame start, so the meshes of a multi-scene frame can be enqueued without a sync each.
                               // Bits 8..31: frames that were dropped (pair overflow / need_global_sort) and then overwritten by a
                               // later frame before the host could redraw them (counted by that later frame's k_setup)
    unsigned long long fragments;
};

// Device-side phase clock: 100 MHz wall-clock stamps (10 ns) taken by the first thread of the first kernel of every phase, in the 64
// bytes right behind Ctrl (one allocation).  They cost one store per kernel and no launch, so RasterTimings' phases are filled on every
// synchronous call (render.rs:2362, 2515-2516, 2544, 2572 always fill them), not only under b32_set_profiling.
struct Stamps { unsigned long long t[8]; };
// Rare events of k_setup under direct binning, behind Stamps: each word holds the EPOCH (frame number, never 0) of the last frame the
// event happened in, so nothing has to be reset between frames and k_setup's own frame-start reset of Ctrl cannot race with them.
struct Events { uint32_t bad_index, nan_opaque, nan_transparent, overflow, long_transparent, setup_done /* k_flag -> k_join: epoch of the frame whose setup kernel has finished */,
                join_abort /* k_join -> fill: epoch of the frame whose setup kernel never arrived (the fill draws nothing but the folded clear) */,
                fill_started /* fused fill -> k_gate of a later frame's setup kernel: FillArgs::start_seq of the last fused kernel that STARTED on this control
                                block (it started => everything in front of it on the main stream has ended) */,
                wbin_done /* side stream -> first wire kernel on the main stream: WireArgs::epoch of the frame whose early k_wire_bin has finished */,
                poll_done /* k_flag_poll -> the fused kernel itself (FillArgs::join_seq): the setup kernel of that hand-over has finished */,
                poll_lost /* fused kernel -> the transparent pass behind it: join_seq of a hand-over the fill gave up on (nothing of that draw's lists exists) */, _pad[5]; };
__device__ __forceinline__ Events* events_of(Ctrl* ctrl) { return reinterpret_cast<Events*>(reinterpret_cast<unsigned char*>(ctrl) + 128); }
enum { ST_SETUP = 0, ST_BIN = 1, ST_FILL = 2, ST_WIRE = 3, ST_END = 4,
       ST_CLK0 = 5, ST_CLK1 = 6, ST_CLKW = 7 };   // shader-cycle counter at the start / end of workgroup 0 of the fused kernel, wall clock at its end
__device__ __forceinline__ void phase_stamp(Ctrl* ctrl, int k) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned long long* t = reinterpret_cast<Stamps*>(ctrl + 1)->t;
        if (!t[k]) t[k] = wall_clock64();
    }
}

constexpr uint32_t LOCAL_SORT_CAP = 2048;   // longest tile list the per-tile LDS radix sort handles

// Loop-invariant conversions of transform_to_camera_space / project_to_screen (fixed.rs:362-400): the camera converted to 4.12
// fixed point once per frame instead of once per vertex.  Plain IEEE f32 multiply + Rust's saturating `as i32`, identical on
// host and device.
struct CamFx {
    int32_t px, py, pz, bx[3], by[3], bz[3];
    int32_t vs, half_w, half_h;
};
__host__ __device__ inline int32_t fx_from_f32_any(float f) {                 // Fixed32::from_f32, fixed.rs:125-127
    const float p = f * K::ONE_F;
    if (p != p) return 0;
    if (p >= 2147483648.0f) return INT32_MAX;
    if (p <= -2147483648.0f) return INT32_MIN;
    return (int32_t)p;
}
__host__ __device__ inline CamFx make_camfx_any(const B32Camera& c, uint32_t width, uint32_t height) {
    CamFx k;
    k.px = fx_from_f32_any(c.position[0]); k.py = fx_from_f32_any(c.position[1]); k.pz = fx_from_f32_any(c.position[2]);
    for (int i = 0; i < 3; ++i) { k.bx[i] = fx_from_f32_any(c.basis_x[i]); k.by[i] = fx_from_f32_any(c.basis_y[i]); k.bz[i] = fx_from_f32_any(c.basis_z[i]); }
    const uint32_t mn = width < height ? width : height;
    k.vs = fx_from_f32_any(((float)mn / K::PF_VIEWPORT_DIV) * K::PF_VIEWPORT_FRAC);      // fixed.rs:398
    k.half_w = (int32_t)((uint32_t)((int32_t)width / 2) << K::FRAC_BITS);               // fixed.rs:399-400
    k.half_h = (int32_t)((uint32_t)((int32_t)height / 2) << K::FRAC_BITS);
    return k;
}

// Everything k_setup / k_fill need about the frame, passed by value.
struct FrameParams {
    B32Camera cam;
    uint32_t width, height;
    uint32_t band_y0, band_y1;
    uint32_t tiles_x, tiles_y;             // tile grid of the band: tiles_y tile rows
    uint32_t tile_yb;                      // y of the first tile row (a multiple of tile_h at or above band_y0)
    uint32_t tile_h;                       // height of the tile rows: TILE_H, or TILE_H / 2, / 4 on the sort-free path when tiles are few
    uint32_t nv, nf, nt;
    uint32_t n_lights;
    float ambient;
    uint8_t affine, shading, backface_cull, dithering, fixed_point, has_fog, zmode, fmt8;   // zmode = settings.use_zbuffer; fmt8 = render_mesh (8-bit colour)
    uint8_t ortho, xray, wire_collect, band_only;   // ortho_projection.is_some(), xray_mode, any wireframe phase wants its triangles;
                                                    // band_only: records of surfaces outside this rank's band are not needed (sort-free path)
    uint8_t redraw, lights_inline, tex_blend_any, batched;   // batched: per-mesh ambient / fog / backface_cull from the MeshTable   // tex_blend_any: some texture's blend mode is not Opaque (else k_setup never reads the descriptors)        // redraw: the host is repeating a dropped frame (not a new one); lights_inline: the
                                                    // lights travel in the kernel arguments (LightSet) instead of a device buffer
    float ortho_zoom, ortho_cx, ortho_cy;      // OrthoProjection (types.rs), math.rs:140-148
    B32Fog fog;
    CamFx camfx;
};

// Triangle of the wireframe phases (render.rs:2445-2449, 2509-2511, 2574-2635): screen coordinates `as i32`, depth as is.
struct WireTri { int32_t x[3], y[3]; float z[3]; uint32_t kind; };   // kind: 0 none, 1 back-face, 2 front-face
static_assert(sizeof(WireTri) == 40, "WireTri layout");

// ---------------------------------------------------------------- Rust-semantics helpers (device)
// Rust's `f as i32` / `f as u32` (NaN -> 0, out of range -> saturate, truncation toward zero) IS what gfx950's conversion instructions
// do: v_cvt_i32_f32 / v_cvt_u32_f32 clamp and map NaN to 0 in hardware.  C++'s cast is undefined there, so the compiler may not assume it
// and the explicit compare / select chains cost ~8 instructions and two exec-mask branches per conversion (nine of them per face in
// k_setup): the instruction is named directly.  b32_selftest_f32 ops 5 / 6 prove the semantics on the device
// (tests/test_gpu_parity.py::test_device_f32_semantics).
__device__ __forceinline__ int32_t hw_cvt_i32(float f) { int32_t r; asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(f)); return r; }
__device__ __forceinline__ uint32_t hw_cvt_u32(float f) { uint32_t r; asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(f)); return r; }
// `f as u32/usize`: NaN -> 0, negative -> 0, saturating (a usize beyond u32 only ever meets a `.min(width - 1)` or an empty-box test)
__device__ __forceinline__ uint32_t f2u_sat(float f) { return hw_cvt_u32(f); }
__device__ __forceinline__ uint32_t f2u8_sat(float f) { return min(hw_cvt_u32(f), 255u); }
__device__ __forceinline__ int32_t f2i32_sat(float f) { return hw_cvt_i32(f); }
// f32::min / f32::max ignore NaN == IEEE minNum/maxNum == fminf/fmaxf.
__device__ __forceinline__ float rmin(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ float rmax(float a, float b) { return __builtin_fmaxf(a, b); }
// f32::clamp propagates NaN.
__device__ __forceinline__ float rclamp(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
// f32::rem_euclid(1.0): fmodf(x,1) == x - trunc(x) exactly; then +1.0 when negative (may round to 1.0).
__device__ __forceinline__ float rem_euclid1(float x) {
    float r = x - __builtin_truncf(x);
    return r < 0.0f ? r + 1.0f : r;
}

// 1.0f / x, correctly rounded, in three instructions where that is provable: v_rcp_f32 is accurate to 1 ulp, and one residual correction
// r + r * (1 - x r), both steps fused, then IS the correctly rounded reciprocal (Markstein) -- checked on the device against `/` for EVERY
// one of the 2^32 operands (b32_selftest_f32 ops 9 / 10, tests/test_gpu_parity.py::test_device_f32_semantics).  Operands whose reciprocal or
// residual could leave the normal range (exponent field outside [RCP_EXP_LO, RCP_EXP_HI]: zeros, denormals, infinities, NaNs, the largest
// and smallest binades) take the compiler's division sequence (11 instructions).  The fill's z-buffer coverage computes one such
// reciprocal per fragment (render.rs:1546-1550).
constexpr uint32_t RCP_EXP_LO = 3, RCP_EXP_HI = 250;
__device__ __forceinline__ float rcp_exact(float x) {
    const uint32_t ex = (__float_as_uint(x) >> 23) & 255u;
    const float r = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    const float q = __builtin_fmaf(e, r, r);
    return (ex >= RCP_EXP_LO && ex <= RCP_EXP_HI) ? q : 1.0f / x;
}

// total order on non-NaN f32 as u32 (z-buffer keys for the 64-bit LDS atomicMin) and its inverse
// -0.0 and +0.0 get ONE key (the reference's `z < zbuffer` sees them as equal, so the first fragment in order keeps the pixel);
// a stored depth that decodes to zero has its sign recomputed from the winning surface (exact_depth_at).
__device__ __forceinline__ uint32_t zsort_key(float z) { const uint32_t b = z == 0.0f ? 0u : __float_as_uint(z); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float zsort_val(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }

// ---------------------------------------------------------------- colour helpers
__device__ __forceinline__ uint32_t expand5(uint32_t v5) { return ((v5 << K::EXPAND5_SHL) | (v5 >> K::EXPAND5_SHR)) & 0xFF; }   // render.rs:1161-1163
// Color15::to_rgba (types.rs:220-226) as a little-endian RGBA8 word
__device__ __forceinline__ uint32_t c15_to_rgba(uint32_t c) {
    if ((c & 0xFFFF) == K::C15_TRANSPARENT) return 0;
    return expand5((c >> K::C15_R_SHIFT) & K::C15_CHANNEL_MAX) | (expand5((c >> K::C15_G_SHIFT) & K::C15_CHANNEL_MAX) << 8) |
           (expand5(c & K::C15_CHANNEL_MAX) << 16) | 0xFF000000u;
}
// PS1_DITHER_MATRIX (render.rs:1150-1155) packed as 16 signed nibbles, index = (y&3)*4 + (x&3)
__device__ __forceinline__ int dither_offset(uint32_t x, uint32_t y) {
    // rows: {-4,0,-3,1} {2,-2,3,-1} {-3,1,-4,0} {3,-1,2,-2}; nibble = value & 0xF
    const unsigned long long M = 0xE2F3'0C1D'F3E2'1D0CULL;
    uint32_t i = ((y & 3) << 2) | (x & 3);
    int v = (int)((M >> (i * 4)) & 0xF);
    return (v ^ 8) - 8;   // sign-extend 4 bits
}
// blend_rgb555 (render.rs:1093-1145) on RGBA8 words; returns r|g<<8|b<<16 (no alpha)
__device__ __forceinline__ uint32_t blend_rgb555(uint32_t front, uint32_t back, uint32_t mode) {
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        int f5 = (int)(((front >> (8 * i)) & 255) >> K::BLEND_IN_SHIFT);
        int b5 = (int)(((back >> (8 * i)) & 255) >> K::BLEND_IN_SHIFT);
        int r5;
        switch (mode) {
            default:
            case B32_BLEND_OPAQUE:      r5 = f5; break;
            case B32_BLEND_AVERAGE:     r5 = min((b5 + f5) / K::BLEND_AVG_DIV, K::BLEND_HI); break;
            case B32_BLEND_ADD:         r5 = min(b5 + f5, K::BLEND_HI); break;
            case B32_BLEND_SUBTRACT:    r5 = max(b5 - f5, K::BLEND_LO); break;
            case B32_BLEND_ADD_QUARTER: r5 = min(b5 + f5 / K::BLEND_QUARTER_DIV, K::BLEND_HI); break;
            case B32_BLEND_ERASE:       r5 = b5; break;
        }
        out |= (uint32_t)(r5 << K::BLEND_OUT_SHIFT) << (8 * i);
    }
    return out;
}

// tile row of screen row y (y >= fp.tile_yb) and its inverse: top screen row and height of tile row `row`
// (tile_h is a power of two -- 64, 32, 16 or 8 rows: a shift by 31 - clz instead of a run-time division, which costs ~25 instructions twice
// per surviving face in k_setup)
__host__ __device__ __forceinline__ uint32_t tile_row_of(const FrameParams& fp, uint32_t y) { return (y - fp.tile_yb) >> (31u - (uint32_t)__builtin_clz(fp.tile_h)); }
__host__ __device__ __forceinline__ void tile_row_geom(const FrameParams& fp, uint32_t row, uint32_t& top, uint32_t& th) {
    th = fp.tile_h; top = fp.tile_yb + row * th;
}
// Tile span of a surface's (band-clipped) bounding box, packed tx0 | tx1<<8 | ty0<<16 | ty1<<24 with band-relative tile rows
// (<= 256 tiles per axis: 16384 px); 0xFFFFFFFF = touches no tile of this band.
__device__ __forceinline__ uint32_t pack_tile_span(uint32_t bbx, uint32_t bby, uint32_t flags, const FrameParams& fp, uint32_t& count) {
    count = 0;
    if (flags & F_EMPTY) return 0xFFFFFFFFu;
    const uint32_t min_x = bbx & 0xFFFF, max_x = bbx >> 16;
    const uint32_t min_y = max(bby & 0xFFFF, fp.band_y0), max_y = min(bby >> 16, fp.band_y1);   // other rows belong to another rank
    if (min_x >= max_x || min_y >= max_y) return 0xFFFFFFFFu;
    const uint32_t tx0 = min_x / TILE_W, tx1 = (max_x - 1) / TILE_W;
    const uint32_t ty0 = tile_row_of(fp, min_y), ty1 = tile_row_of(fp, max_y - 1);
    count = (tx1 - tx0 + 1) * (ty1 - ty0 + 1);
    return tx0 | (tx1 << 8) | (ty0 << 16) | (ty1 << 24);
}

// ---------------------------------------------------------------- kernel launchers (defined in the .hip files)
struct SortScratch {
    uint32_t* block_hist;   // [4096][max_blocks] digit-major (rows used: 2^bits)
    uint32_t  max_blocks;
    uint32_t* digit_total;  // [4096]
};
constexpr int SORT_THREADS = 256;
constexpr int SORT_ITEMS = 16;
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;   // 4096 keys per block

// One stable LSD pass on bits [shift, shift+bits), bits in {8, 11, 12}. n is read from *n_dev (<= n_cap). If vals_in == nullptr
// the value of element i is i and keys equal to KEY_INVALID are dropped (first pass over the face-order key array).
struct RadixExtra {            // optional jobs folded into a pass to save kernel launches
    Ctrl* post_ctrl = nullptr; const uint32_t* partials = nullptr; uint32_t npart = 0;   // k_setup counter reduction (first depth pass; also stamps ST_BIN)
    uint32_t* ranges_out = nullptr; uint32_t n_ranges = 0;                               // tile-list ranges (single-pass tile sort)
};
void launch_radix_pass(hipStream_t s, const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out,
                       const uint32_t* n_dev, uint32_t n_cap, int shift, int bits, const SortScratch& sc, const RadixExtra& ex = RadixExtra());

// k_setup publishes 5 counters per 256-face block (visible, transparent, nan_opaque, nan_transparent, bad_index) into
// `partials[block*8 + k]`; k_after_setup reduces them into Ctrl.  (One same-address atomic per wave costs ~12 ns each and
// serialises: 15.6 k waves = the whole kernel time at 1 M faces.)
// up to LIGHTS_INLINE lights travel by value in the kernel arguments: a light change costs no copy and no synchronisation
constexpr uint32_t LIGHTS_INLINE = 8;
struct LightSet { B32Light l[LIGHTS_INLINE]; };
// Batched frame (b32_frame_begin / _add_scene / _end): several meshes merged into one resident mesh, drawn by ONE k_setup + fill pair.
// What the reference's callers vary from mesh to mesh (scene.rs:112-261: per-room ambient and fog, per-part double_sided) travels in
// this table, indexed by the mesh number k_merge_mesh left in the spare byte of every merged face; everything else (camera, lights,
// the other settings) is the frame's.
constexpr uint32_t BATCH_MESHES = 32;
struct MeshRow { float ambient; uint32_t flags; B32Fog fog; };       // flags: bit 0 backface_cull, bit 1 fog is Some
struct MeshTable { MeshRow m[BATCH_MESHES]; };
struct RecArrays { CovRec* cov; ShadeRec* shade; AuxRec* aux; };
// Direct binning (sort-free path, meshes too large for the in-kernel collection): k_setup itself appends every surviving face to the
// lists of the tiles its span touches -- one returning atomic per (tile, face) pair on the tile's fill counter, whose latency passes
// behind the record build -- so the frame has no binning launch at all.  Every tile owns a fixed region of `region` entries:
// opaque-class entries grow from its front (at most cap_opaque), transparent-class entries from its back (at most cap_transparent);
// a face that finds a region full raises Events::overflow and the frame is redrawn with larger regions.  The counters (one 128-byte
// line per tile: word 0 opaque, word 1 transparent) are zero between frames: k_cover zeroes a tile's pair when it takes the tile.
#ifndef B32_FILL_PAD
#define B32_FILL_PAD 32
#endif
constexpr uint32_t FILL_PAD = B32_FILL_PAD;     // words per tile counter line
struct DirectBin {
    uint32_t* fill;             // nullptr: off
    uint32_t* lists;
    uint32_t region, cap_opaque, cap_transparent;
    uint32_t with_class;        // 1: faces of the transparent pass (render.rs:2522-2523) go to the back of the region
    uint32_t epoch;
};
void launch_merge_mesh(hipStream_t s, const B32Vertex* sv, uint32_t nv, const B32Face* sf, uint32_t nf, uint32_t nt, B32Vertex* dv, B32Face* df,
                       uint32_t vbase, uint32_t tbase, uint32_t mesh);
void launch_offset_tex(hipStream_t s, const TexDesc* src, uint32_t nt, TexDesc* dst, uint32_t texel_base);
void launch_setup(hipStream_t s, const FrameParams& fp, const B32Vertex* verts, const B32Face* faces, const TexDesc* tex,
                  const B32Light* lights, const LightSet& inline_lights, const MeshTable& mesh_table, RecArrays recs, const DirectBin& direct, float* shades, uint32_t* keys, uint32_t* spans,
                  uint32_t* partials, Ctrl* ctrl, WireTri* wire, int n_cu, const float* pos12, const float* attr12, uint32_t* face_of);
void launch_gate(hipStream_t s, Ctrl* prev, uint32_t need, uint32_t patience_ticks, uint32_t start_seq, Ctrl* mine, uint32_t start_patience_ticks = 200000000u);
void launch_flag(hipStream_t s, Ctrl* ctrl, uint32_t epoch);
void launch_flag_poll(hipStream_t s, Ctrl* ctrl, uint32_t seq);        // Events::poll_done = seq, behind the setup kernel of a polled hand-over
void launch_flag_wbin(hipStream_t s, Ctrl* ctrl, uint32_t epoch);      // Events::wbin_done = epoch, behind the early k_wire_bin on the side stream
void launch_join(hipStream_t s, Ctrl* ctrl, uint32_t epoch, uint32_t patience_ticks);
void launch_pack_streams(hipStream_t s, const B32Vertex* verts, uint32_t nv, float* pos12, float* attr12, bool with_lit);
void launch_project_fixed(hipStream_t s, const float* pos, uint32_t n, B32Camera cam, uint32_t w, uint32_t h,
                          int32_t* sx, int32_t* sy, float* z);
void launch_selftest(hipStream_t s, int op, const float* a, const float* b, const float* c, float* out, uint32_t n);
// test tap: constants of B32_CONSTANTS (one word each, f32 bits or the integer), UNR table (257 bytes) and dither offsets (16 words,
// index (y & 3) * 4 + (x & 3)) as device code holds them
void launch_constants(hipStream_t s, uint32_t* consts, uint8_t* unr, int32_t* dither);
// skip mask of the texel pool (see FillArgs.texmask): n texels, 16-bit Color15 or 32-bit Color
void launch_build_mask(hipStream_t s, const uint16_t* texels15, const uint32_t* texels32, uint32_t n, uint32_t* mask);
constexpr uint32_t MASK_LDS_MAX_WORDS = 9216;      // 36 KB: the runner-up plane of the fused kernel, unused by EXACT coverage
void launch_clear(hipStream_t s, uint32_t* fb, size_t n_px, uint32_t rgba);
// staged upload of a drop-in call: up to 16 segments copied by ONE kernel from a pinned host arena (mapped into the device's address
// space) to their device buffers -- every SDMA copy costs ~10 us of stream time, one kernel reading over PCIe ~5 us for all of them
struct UploadSegs { void* dst[16]; uint32_t src_off[16]; uint32_t n16[16]; uint32_t count; };
void launch_upload(hipStream_t s, const void* arena_dev, const UploadSegs& segs);
void launch_ctrl_out(hipStream_t s, Ctrl* ctrl, void* dst128);      // end-of-frame stamp + Ctrl and Stamps (128 B) to a host-visible slot
// Clut::lookup expansion; *skippable (may be null) += texels with r5 = g5 = b5 = 0, i.e. the ones the black_transparent rule can skip
void launch_expand_indexed(hipStream_t s, const uint8_t* idx, uint32_t n, const uint16_t* clut, uint32_t clut_len, uint16_t* out, uint32_t* skippable);

void launch_bin(hipStream_t s, const FrameParams& fp, const uint32_t* spans, const uint32_t* order, Ctrl* ctrl,
                uint32_t* counts, uint32_t* block_sums, uint32_t max_blocks, uint32_t* pair_keys, uint32_t* pair_vals, uint32_t pair_cap);
// Fast path: pairs straight from k_setup's per-face spans, in face order (the per-tile LDS sort of k_cover orders them).
void launch_bin_faces(hipStream_t s, const FrameParams& fp, const uint32_t* spans, const uint32_t* keys, const uint32_t* partials,
                      Ctrl* ctrl, uint32_t* pair_keys, uint32_t* pair_vals, uint32_t pair_cap, int with_class);
// Orthographic depth keys use all 32 bits, so the opaque/transparent partition (render.rs:2522-2523) is one more stable pass
// on a class key: keys_out[i] = transparent(recs[order[i]]) for i < *n_dev.
void launch_class_keys(hipStream_t s, const CovRec* recs, const uint32_t* order, const uint32_t* n_dev, uint32_t n_cap, uint32_t* keys_out);
// Wireframe phases (render.rs:2574-2635).  kind 1: back-face edges, first occurrence per screen-space edge, depth-tested
// (draw_line_3d, render.rs:757-817); kind 2: front-face overlay, no depth test (draw_line, render.rs:716-750).
struct WireArgs {
    const WireTri* tris; uint32_t nf;
    uint32_t* table_owner; uint32_t* table_first; uint32_t table_mask;    // open-addressed edge table (power-of-two size)
    uint32_t* fb; const float* zbuf;                                        // zbuf == nullptr: every depth is f32::MAX
    uint32_t width, height, band_y0, band_y1;
    Ctrl* ctrl;
    // tile route (B32_ROUTE_WIRE_TILES; tile_fill == nullptr: off): 64 x WIRE_TH tiles over the band, first tile row at tile_yb; tile_fill holds
    // one counter per tile, then the overflow flag and the count of edges left to the global kernels; lists of WIRE_TILE_CAP ids per tile
    uint32_t* tile_fill; uint32_t* tile_lists;
    uint32_t tiles_x, tiles_y, tile_yb;
    uint32_t epoch;             // never 0: what the two flag words hold when THIS frame raised them (no zeroing between frames)
};
constexpr uint32_t WIRE_TH = 16;               // wire tiles are 64 x WIRE_TH pixels: small enough that four workgroups share a CU's LDS
constexpr uint32_t WIRE_TILE_CAP = 256;        // faces per tile list: one 256-lane workgroup de-duplicates their 768 edges in LDS
// The depth parameter of draw_line_3d, t = step / total_steps (render.rs:784), for integers 0 <= k <= N < WIRE_NARROW: with
// rN = 1.0f / N (correctly rounded, once per line) one residual correction of k * rN IS the correctly rounded quotient (Markstein's
// division theorem; N has at most 14 significant bits, far from the all-ones significand the theorem excepts).  Three instructions
// per pixel instead of the ~11 of an IEEE division; b32_selftest_f32 op 8 compares it with `/` for every such pair on the device.
constexpr int WIRE_NARROW = 16384;
__device__ __forceinline__ float wire_t_fast(float kf, float Nf, float rN) {
    const float q = kf * rN;
    const float rem = __builtin_fmaf(-q, Nf, kf);
    return __builtin_fmaf(rem, rN, q);
}
void launch_wire(hipStream_t s, const WireArgs& a, bool back, bool front, bool binned = false, Ctrl* wait_ctrl = nullptr, uint32_t wait_epoch = 0);   // wait_ctrl: see k_wire_table_clear
void launch_wire_bin(hipStream_t s, const WireArgs& a, bool back, bool front, bool early);
// Sort-free fast path: tile lists (unordered) by a counting sort straight from k_setup's spans; false = not applicable (too many
// tiles for the LDS histogram), the caller takes the keyed radix path.  With `keys` the lists are split by class
// ([opaque..., transparent...], boundary in tile_mid) and a transparent part longer than blend_cap raises need_global_sort.
constexpr uint32_t BLEND_SORT_CAP = 2048;       // longest transparent tile list k_blend sorts in LDS (rank sort on 64-bit priorities)
bool bin_spans_applicable(const FrameParams& fp, const SortScratch& sc, bool with_class);
bool launch_bin_spans(hipStream_t s, const FrameParams& fp, const uint32_t* spans, const uint32_t* keys, const uint32_t* partials, Ctrl* ctrl,
                      const SortScratch& sc, uint32_t pair_cap, uint32_t* ranges, uint32_t* tile_mid, uint32_t blend_cap, uint32_t* pair_vals);
void launch_tile_ranges(hipStream_t s, const uint32_t* pair_keys, const Ctrl* ctrl, uint32_t pair_cap, uint32_t* ranges, uint32_t n_keys);

void launch_clear_gradient(hipStream_t s, uint32_t* fb, float* zbuf, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1, uint32_t top, uint32_t bottom);
void launch_sky(hipStream_t s, const B32SkyVertex* v, uint32_t nv, const uint32_t* faces, uint32_t nf, const B32Camera& cam, float2* proj,
                uint32_t* fb, uint32_t width, uint32_t height, uint32_t band_y0, uint32_t band_y1);
void launch_stars(hipStream_t s, const int32_t* cx, const int32_t* cy, const uint8_t* rgb, uint32_t n, float size, uint32_t* fb,
                  uint32_t width, uint32_t height, uint32_t band_y0, uint32_t band_y1);
void launch_upscale_nearest(hipStream_t s, const uint32_t* src, uint32_t sw, uint32_t sh, uint32_t* dst, uint32_t dw, uint32_t dh);

struct FillArgs {
    FrameParams fp;
    const CovRec* crecs;        // coverage records, by face id
    const ShadeRec* srecs;      // shading records
    const AuxRec* xrecs;        // 1/z terms and literal-walk start values (z-buffer mode, perspective-correct UVs, F_SLOW surfaces)
    const float* shades;        // [sid][9] or nullptr
    uint32_t* pair_vals;        // surface ids, grouped by (tile,class); painter's order inside a group (after the tile-local sort)
    const uint32_t* keys;       // face-order radix keys of k_setup (tile-local sort)
    uint32_t local_sort;        // 1: lists arrive in face order and k_cover sorts each one in LDS
    const uint32_t* ranges;     // list ranges: [2*ntiles + 1] keyed by (tile<<1|class), or [ntiles + 1] keyed by tile (fast path)
    uint32_t* tile_mid;         // fast path: first transparent-pass entry of every tile list (written by k_cover)
    uint32_t tile_keys_only;    // 1: ranges are per tile, the class boundary comes from the tile-local sort
    const TexDesc* tex;
    const uint16_t* texels;
    uint32_t* fb;               // RGBA8 words, full frame
    float* zbuf;                // z-buffer mode: Framebuffer::zbuffer (f32 per pixel, read-modify-write across calls)
    uint32_t* vis;              // visibility buffer per pixel: winning tile-list position (1-based, 0 = uncovered) [+ runner-up << 16]
    TexDesc tex0;               // descriptor of texture 0 (used when nt == 1: no per-pixel descriptor gather)
    Ctrl* ctrl;
    uint32_t lds_tex_texels;    // > 0: every face samples texture 0 and it is staged in LDS (width*height texels)
    uint32_t may_blend;         // 0: no face/texture can be in the transparent pass -> k_blend is not launched
    uint32_t exact_coverage;    // 1: phase A applies the full skip rule and counts fragment stores; 0: CHEAP coverage + repair
    uint32_t skip_solid;        // wireframe_overlay: surfaces are counted but not drawn (render.rs:2550)
    const uint32_t* texels32;   // 8-bit-colour path: pooled Color texels, r | g<<8 | b<<16 | blend<<24 (TexDesc.offset indexes this pool)
    uint32_t ordered_all;       // 1: every surface may blend -> no overwrite pass, k_blend walks the whole tile list in order
    uint32_t gather_blend;      // 1 (with prio64): k_blend sorts the transparent part of each (unordered) tile list itself
    // small meshes (at most 2048 faces): no binning launch at all -- every workgroup of the fused kernel
    // collects its tile's list from k_setup's spans itself, and reduces k_setup's counters (workgroup 0 publishes them in Ctrl)
    uint32_t inline_bin;        // 1: lists are built inside k_cover at pair_vals[tile * list_stride ...]
    uint32_t direct_bin;        // 1: lists were built by k_setup (DirectBin) at pair_vals[tile * list_stride ...], counts in tile_fill
    uint32_t* tile_fill;        // DirectBin::fill
    uint32_t epoch;             // DirectBin::epoch
    uint32_t list_stride;       // entries per tile region (a multiple of 32: regions never share a cache line)
    const uint32_t* spans;      // k_setup's packed tile span per face
    const uint32_t* partials;   // k_setup's per-block counters
    // Skip mask: one bit per texel of the pool, set when the texel can be skipped by the transparency rule -- r5 = g5 = b5 = 0 (RGB555:
    // skipped when the face has black_transparent, render.rs:1591-1608) or blend == Erase (8-bit path, render.rs:1348-1352).  EXACT
    // coverage on the sort-free path decides every fragment from this bit instead of fetching the texel: 1/16 of the bytes, and it fits
    // LDS (mask_lds_words > 0: the first that many words are staged in the unused runner-up plane; a 256x256 texture is 8 KB).
    const uint32_t* texmask;
    uint32_t mask_lds_words;
    // Framebuffer::clear folded into the frame (b32_fb_clear defers itself; the sort-free fused kernel writes the clear colour to every
    // pixel of the band nobody draws, so the frame has no clear launch and uncovered pixels are written once, not twice)
    uint32_t clear_on, clear_rgba;
    uint32_t clear_depth;       // with clear_on in z-buffer mode: the clear resets the depth buffer too (every depth f32::MAX, render.rs:43)
    uint32_t narrow_only;       // 1: never the 16-wave workgroups of the fused kernel (b32_set_routes)
    // One indexed texture (b32_scene_upload_indexed, nt == 1): 256 CLUT entries (zero behind the palette: an index past it looks up 0x0000,
    // Clut::lookup types.rs:390-397) followed by width * height index bytes.  atlas_idx_bytes > 0: every workgroup of the fused kernel
    // stages both in LDS behind its tile planes and the shading phase looks the texel up there instead of fetching the expanded texel
    const uint8_t* atlas0;
    uint32_t atlas_idx_bytes;
    uint32_t stagger;           // > 0: the second workgroup of every CU starts this many 10-ns ticks late (set by launch_fill, see k_cover)
    uint32_t span_cover;        // 1: sort-free CHEAP painter's coverage by exact row intervals (B32_ROUTE_SPAN_COVER, b32_fill.hip "span coverage")
    // The merged draws of a batched frame (console-sized: 16-wave workgroups on at most 5 / 8 of the CUs): the hand-over from the draw's setup kernel
    // on the side stream is polled INSIDE the fused kernel -- thread 0 of every workgroup reads Events::poll_done until it holds join_seq, and only
    // a workgroup that really waited acquires at device scope -- instead of by a cross-stream event (6.5 us of the main stream per draw).  0: no wait.
    uint32_t join_seq, join_patience;       // patience in 10-ns ticks; when it runs out the workgroup draws nothing but the folded clear (sticky bit 3)
    uint32_t start_seq;         // != 0: workgroup 0 publishes it in Events::fill_started when the fused kernel starts (see k_gate) ...
    uint32_t start_defer;       // ... unless the frame's transparent pass does (1: k_blend).  The next frame's setup kernel then runs beside THAT kernel and
                                //     the fill has the GPU to itself (b32_frame.hip)
    uint32_t prio64;            // 1: sort-free coverage -- visibility is a 64-bit max of (painter's key << 32 | face id); `vis` holds
                                //    two words per pixel: winner face id + 1, runner-up face id + 1 (0 = none)
#ifdef B32_TIMELINE
    unsigned long long* dbg;    // experiment builds only: [0] = entry counter, then 4 words per tile (wg << 32 | tile, t_start, t_covered, t_shaded)
#endif
};
void launch_fill(hipStream_t s, const FillArgs& a, int n_cu, hipEvent_t after_cover = nullptr);   // k_cover [event] k_shade k_blend
size_t fill_lds_tex_budget();   // bytes of LDS left for a staged texture
constexpr uint32_t ATLAS_CLUT_BYTES = 512;   // 256 Color15 entries in front of the index bytes
// LDS left for a staged index atlas (CLUT included) in the fused kernel: wide = 16-wave workgroups, one per CU; else two 8-wave workgroups
uint32_t fill_lds_atlas_room(bool wide);

}  // namespace b32

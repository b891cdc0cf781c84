// b32_cover.h -- the coverage phase of the fill kernels (included by b32_fill.hip only): visibility of the opaque pass.
//
//   The opaque pass of the reference (render.rs:2553-2559) only ever overwrites pixels (set_pixel_15), so its result per pixel is the
//   LAST surface in painter's order whose fragment is not skipped: coverage of all opaque surfaces of a tile runs in parallel and
//   visibility is an LDS atomicMax -- of the surface's global painter's priority (key << 32 | face id) on the sort-free path, of its
//   position in the tile's sorted list on the keyed paths -- order-independent, deterministic, no overdraw shading.
//     EXACT coverage applies the whole skip rule per fragment (inside test + texel + transparency, render.rs:1536-1607) and counts the
//           reference's pixel stores exactly;
//     CHEAP coverage (textures with few skippable texels) applies only the inside test and keeps the exact top two per pixel; the rare
//           pixels whose top surface turns out to be skipped are repaired by the shading phase.
//   Coverage is scheduled by ROW ITEMS (phase_a_rows): every lane walks one row of one surface; on the sort-free CHEAP painter's path a
//   row is its exact integer interval (span coverage), elsewhere the reference's per-pixel test on a trimmed row.
#pragma once
#include "b32_fill_common.h"

namespace b32 {

// Experiment builds (-DB32_TIMELINE): per-wave shader-clock sums of the coverage / shading sub-phases, added to FillArgs::dbg behind the per-tile
// records (slot k at dbg[1 + 4 * 8192 + k]); tools/timeline.py prints them.  The stamps are s_memtime reads (scalar): they do not touch the VGPR budget.
#ifdef B32_TIMELINE
// (sums kept per wave in the slack of the tile-plane allocation -- bytes 68096.. of the 73728 -- and flushed once, when the workgroup ends)
#define B32_DBG_SLOTS(tb) (reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned char*>(const_cast<uint32_t*>(tb)) + 68096) + (threadIdx.x >> 6) * 32)
#define B32_CLK_DECL(name) unsigned long long name = (unsigned long long)clock64()
#define B32_CLK_ADD(a, slot, from) do { const unsigned long long _n = (unsigned long long)clock64(); if (lane == 0) B32_DBG_SLOTS(tilebuf)[slot] += _n - (from); (from) = _n; } while (0)
#define B32_CNT_ADD(a, slot, v) do { if (lane == 0) B32_DBG_SLOTS(tilebuf)[slot] += (unsigned long long)(v); } while (0)
#else
#define B32_CLK_DECL(name) do { } while (0)
#define B32_CLK_ADD(a, slot, from) do { } while (0)
#define B32_CNT_ADD(a, slot, v) do { } while (0)
#endif

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
#ifndef B32_SPAN_PACK
#define B32_SPAN_PACK 1          // span rounds: the per-surface parameters travel packed (7 ds_bpermute per round instead of 15), see phase_a_rows
#endif
#ifndef B32_PF_SREC
#define B32_PF_SREC 0            // experiment: touch the surface's ShadeRec line during coverage so that the shading phase's gather finds it in L2
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
        B32_CLK_DECL(clk);
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
#if B32_PF_SREC
        if (P64 && !EXACT && live) {
            const uint32_t pf = reinterpret_cast<const uint32_t*>(a.srecs + my_sid)[0];      // (a plain load: it allocates in L2)
            asm volatile("" :: "v"(pf));
        }
#endif
        if (ZMODE) { my_key = 0u; my_sid = 0xFFFFFFFEu - my_sid; }
        const uint32_t flags = b.q3.w;
        const uint32_t cx0 = max(b.q1.w & 0xFFFF, x_lo), cx1 = min(b.q1.w >> 16, x_hi);
        const uint32_t cy0 = max(b.q2.x & 0xFFFF, y_lo), cy1 = min(b.q2.x >> 16, y_hi);
        live = live && cx0 < cx1 && cy0 < cy1;
#ifdef B32_TIMELINE
        asm volatile("" :: "v"(cx0), "v"(cy0));        // (the record has arrived)
#endif
        B32_CLK_ADD(a, 0, clk); B32_CNT_ADD(a, 4, 1);
        const bool slow = live && (flags & F_SLOW);
        // span coverage: what the rows of an eligible surface need (edge values at the first pixel of its clipped box, their steps per
        // row, the reciprocal form of the steps per pixel); span_all: every surface of this batch is eligible -- the rounds below then
        // take the span form, else the per-pixel form serves the whole batch (it is valid for every surface)
        bool span_all = false;
        float sE0 = 0.0f, sE1 = 0.0f, sH0 = 0.0f, sH1 = 0.0f, sA = 0.0f;
        SpanEdge sd0 = { 0.0f, 0.0f }, sd1 = { 0.0f, 0.0f }, sd2 = { 0.0f, 0.0f };
        (void)sd0; (void)sd1; (void)sd2; (void)sA;
        uint32_t sG01 = 0, sH01 = 0;                      // B32_SPAN_PACK: (G0, G1) and (H0, H1) as pairs of i16 (|.| <= SPAN_MAX_EXT = 512)
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
#if B32_SPAN_PACK
            // (a surface that is not `fast` may hold steps beyond i16: its packed words are never used -- span_all is false then)
            sG01 = ((uint32_t)hw_cvt_i32(G0) & 0xFFFFu) | ((uint32_t)hw_cvt_i32(G1) << 16);
            sH01 = ((uint32_t)hw_cvt_i32(sH0) & 0xFFFFu) | ((uint32_t)hw_cvt_i32(sH1) << 16);
#else
            sd0 = span_edge(G0); sd1 = span_edge(G1); sd2 = span_edge(G2);
#endif
        }
        const uint32_t h = (live && !slow) ? cy1 - cy0 : 0u;
        // exclusive prefix sum of the row counts
        const uint32_t inc = dpp_add_scan(h);
        const uint32_t R = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        const uint32_t P = inc - h;
        const float a0 = __uint_as_float(b.q0.z), b0 = __uint_as_float(b.q0.w), a1 = __uint_as_float(b.q1.x), b1 = __uint_as_float(b.q1.y);
        const uint32_t box = (cx0 - x_lo) | ((cx1 - x_lo) << 8) | ((cy0 - ty_top) << 16);      // 7+7+6 bits
        // the same with the surface's first row item in the upper half (P < 64 x 64): one permute instead of two in the span rounds
        const uint32_t boxP = (cx0 - x_lo) | ((cx1 - x_lo) << 6) | ((cy0 - ty_top) << 13) | (P << 19);  // 6+7+6+12 bits
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
        B32_CLK_ADD(a, 1, clk); B32_CNT_ADD(a, 5, (R + 63) / 64); B32_CNT_ADD(a, 7, R);
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
#if B32_SPAN_PACK
                // Seven permutes per round instead of fifteen (the LDS pipe is ONE per CU, shared by the sixteen waves of both workgroups,
                // and the rounds are what it is busy with: tools/timeline.py): the steps travel as two words of i16 pairs, the reciprocal
                // forms and |area| are recomputed by the row's lane -- the very expressions the surface's lane evaluated before, on the
                // same integers, so every interval is the same.
                const uint32_t sbp = bperm(s, boxP);
                const uint32_t g01 = bperm(s, sG01), h01 = bperm(s, sH01);
                const float hE0 = bpermf(s, sE0), hE1 = bpermf(s, sE1);
                const uint32_t sP = sbp >> 19;
                const float rowf = (float)(k - sP);
                const float hG0 = i16lo(g01), hG1 = i16hi(g01), hH0 = i16lo(h01), hH1 = i16hi(h01);
                const float hA = __builtin_fabsf(hG0 * hH1 - hH0 * hG1);             // |area|: sgn^2 (a0 b1 - b0 a1), exact integers
                const SpanEdge e0 = span_edge(hG0), e1 = span_edge(hG1), e2 = span_edge(-(hG0 + hG1));
                const float E0 = __builtin_fmaf(hH0, rowf, hE0), E1 = __builtin_fmaf(hH1, rowf, hE1);       // exact integers
                const float E2 = hA - E0 - E1;
                const uint32_t bx0 = sbp & 63u, bx1 = (sbp >> 6) & 127u, ry = ((sbp >> 13) & 63u) + (k - sP);   // tile-local
#else
                const uint32_t sbox = bperm(s, box), sP = bperm(s, P);
                const float rowf = (float)(k - sP);
                const float hE0 = bpermf(s, sE0), hE1 = bpermf(s, sE1), hH0 = bpermf(s, sH0), hH1 = bpermf(s, sH1), hA = bpermf(s, sA);
                SpanEdge e0, e1, e2;
                e0.r = bpermf(s, sd0.r); e0.c = bpermf(s, sd0.c); e1.r = bpermf(s, sd1.r); e1.c = bpermf(s, sd1.c); e2.r = bpermf(s, sd2.r); e2.c = bpermf(s, sd2.c);
                const float E0 = __builtin_fmaf(hH0, rowf, hE0), E1 = __builtin_fmaf(hH1, rowf, hE1);       // exact integers
                const float E2 = hA - E0 - E1;
                const uint32_t bx0 = sbox & 0xFF, bx1 = (sbox >> 8) & 0xFF, ry = (sbox >> 16) + (k - sP);   // tile-local
#endif
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
            B32_CLK_ADD(a, 2, clk);
            while (lqn) { drain_span(); B32_CNT_ADD(a, 6, 1); }
            B32_CLK_ADD(a, 3, clk);
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

}  // namespace b32

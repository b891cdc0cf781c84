// b32_shade_tile.h -- the shading phase of the fused (sort-free) kernel (included by b32_fill.hip only): after the coverage of a tile the
// workgroup shades it straight from the LDS winners -- no visibility buffer round trip through HBM, and while one workgroup of a CU sits
// in the (memory-latency bound) shading phase the other one runs its (LDS / VALU bound) coverage phase.
#pragma once
#include "b32_fill_common.h"

namespace b32 {

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
                                               uint32_t y_lo, uint32_t y_hi, uint32_t ty_top, uint32_t tid_in, uint32_t lane_in, uint32_t TH, uint32_t* wq,
                                               const uint8_t* latlas) {
    uint32_t tid = tid_in, lane = lane_in;              // (opaque copies: see shade_tile_plain)
    asm volatile("" : "+v"(tid), "+v"(lane));
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
// PIPE (the forms with registers to spare: painter's and z-buffer mode without a shading pass): the step is software-pipelined -- the winners
// and the two 64-byte record gathers of step k + 1 are issued as soon as step k's records have been turned into barycentrics and texel
// addresses, BEFORE step k's texel fetch is waited for, so that the gather's round trip (1.3 us of a 3-us step: tools/timeline.py) runs
// beside the texel fetch, the colour pipeline and the stores instead of in front of them.  Same loads, same arithmetic, another order.
template <int NT, bool ZMODE, bool PIPE>
__device__ __forceinline__ void shade_tile_plain(const FillArgs& a, const uint32_t* tilebuf, uint32_t e0, uint32_t e1, uint32_t x_lo, uint32_t x_hi,
                                                 uint32_t y_lo, uint32_t y_hi, uint32_t ty_top, uint32_t tid_in, uint32_t lane_in, uint32_t TH, uint32_t* wq) {
    // (the lane's constants of this phase -- column, x, masks -- are derived again per tile from an opaque copy of its index: hoisted out of the
    // tile loop they would stay live through the coverage phase, which is the one that sets the kernel's register count)
    uint32_t tid = tid_in, lane = lane_in;
    asm volatile("" : "+v"(tid), "+v"(lane));
    const FrameParams& fp = a.fp;
    const unsigned long long* top = reinterpret_cast<const unsigned long long*>(tilebuf);
    const unsigned long long* sec = top + TILE_H * STR64;
    const uint32_t W = fp.width;
    const int shading = fp.shading;
    constexpr uint32_t ROWS_PER_STEP = NT / 64;
    constexpr uint32_t STEP_ROWS = 2 * ROWS_PER_STEP;
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
        const uint32_t row = (e >> 6) & 63u, c = e & 63u, qx = x_lo + c, qy = ty_top + row;
        bool ok = false; Hit h; h.sid = 0;
        unsigned long long t = 0;
        // (bit 12: the WINNER itself has not been evaluated -- its edge walk must be replayed literally (SH_SLOW), which the straight-line
        // step does not do: the general per-pixel functions, here, where the rare cases meet)
        const bool try_top = act && ((e >> 12) & 1u);
        if (__ballot(try_top)) {
            if (try_top) { t = top[row * STR64 + c]; ok = hit_test<false>(a, sid_of(t), qx, qy, h); }
        }
        repair_pixel<false, ZMODE>(a, sec, act && !ok, row, c, qx, qy, e0, e1, lane, ok, h, t);
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

    // ---- one step = the lane's two pixels (rows rowA and rowA + ROWS_PER_STEP of column `col`), in three parts
    struct In {                                         // part 0: winners from LDS, the two record gathers issued
        unsigned long long tA, tB;
        bool cA, cB, inA, inB, any;
        uint4 a0q, a1q, a2q, a3q, b0q, b1q, b2q, b3q;
    };
    struct Mid {                                        // part 1: barycentrics, packed words, texel fetches issued
        float bA0, bA1, bA2, bB0, bB1, bB2;            // (scalars, not arrays: an array member handed on by pointer stays in scratch memory)
        uint32_t pkA0, pkA1, pkA2, pkB0, pkB1, pkB2;    // pk0, pk1, pk2 of the ShadeRec (vertex colours, texture slot, SH_* flags)
        uint32_t fetA, fetB;
    };
    auto issue = [&](uint32_t r0, In& s) {
        const uint32_t rowA = r0 + (tid >> 6), rowB = rowA + ROWS_PER_STEP;
        const uint32_t pyA = ty_top + rowA, pyB = ty_top + rowB;
        s.inA = rowA < TH && in_x && pyA >= y_lo && pyA < y_hi; s.inB = rowB < TH && in_x && pyB >= y_lo && pyB < y_hi;
        s.tA = s.inA ? top[rowA * STR64 + col] : (ZMODE ? ~0ull : 0ull); s.tB = s.inB ? top[rowB * STR64 + col] : (ZMODE ? ~0ull : 0ull);
        s.cA = covered(s.tA); s.cB = covered(s.tB);
        s.any = __ballot(s.cA || s.cB) != 0ull;
        {   // (unconditional -- a step in which no pixel of the wave is covered reads surface 0's record, one cache line for all lanes, and uses
            // nothing of it: under a wave-uniform branch the thirty-two registers became loop-carried maybe-undefined values and the
            // kernel spilled nineteen of them)
            const uint32_t sidA = s.cA ? sid_of(s.tA) : 0u, sidB = s.cB ? sid_of(s.tB) : 0u;      // (surface 0's record for an uncovered pixel: read, never used)
            const uint4* spA = reinterpret_cast<const uint4*>(a.srecs + sidA);
            const uint4* spB = reinterpret_cast<const uint4*>(a.srecs + sidB);
            s.a0q = spA[0]; s.a1q = spA[1]; s.a2q = spA[2]; s.a3q = spA[3];
            s.b0q = spB[0]; s.b1q = spB[1]; s.b2q = spB[2]; s.b3q = spB[3];
        }
    };
    auto texaddr = [&](const uint4& q0, const uint4& q1, const uint4& q2, const uint4& q3, uint32_t py, bool c, float& b0, float& b1, float& b2) -> uint32_t {
        // render.rs:1507-1510 (edges), :1517-1518 / 1706-1712 in closed form (exact integers), :1536-1538, :1565-1566, types.rs:671-681
        const float x3 = f(q1.x), y3 = f(q1.y), inv = f(q1.z);
        const float ea0 = f(q0.w) - y3, eb0 = x3 - f(q0.z), ea1 = y3 - f(q0.y), eb1 = f(q0.x) - x3;
        const float dx = fx - x3, dy = (float)py - y3;
        const float w0 = ea0 * dx + eb0 * dy, w1 = ea1 * dx + eb1 * dy;
        b0 = w0 * inv; b1 = w1 * inv; b2 = 1.0f - b0 - b1;
        const float u = b0 * f(q2.x) + b1 * f(q2.y) + b2 * f(q2.z);
        const float v = b0 * f(q2.w) + b1 * f(q3.x) + b2 * f(q3.y);
        const float uw = rem_euclid1(u), vw = rem_euclid1(1.0f - v);
        const uint32_t tx = min(f2u_sat(uw * twf), d.width - 1), ty = min(f2u_sat(vw * thf), d.height - 1);
        return c ? d.offset + ty * d.width + tx : d.offset;
    };
    auto part1 = [&](uint32_t r0, const In& s, Mid& m) {
        const uint32_t pyA = ty_top + r0 + (tid >> 6), pyB = pyA + ROWS_PER_STEP;
        const uint32_t taA = texaddr(s.a0q, s.a1q, s.a2q, s.a3q, pyA, s.cA, m.bA0, m.bA1, m.bA2);
        const uint32_t taB = texaddr(s.b0q, s.b1q, s.b2q, s.b3q, pyB, s.cB, m.bB0, m.bB1, m.bB2);
        m.fetA = a.texels[taA]; m.fetB = a.texels[taB];                            // both fetches in flight
        m.pkA0 = s.a1q.w; m.pkA1 = s.a3q.z; m.pkA2 = s.a3q.w;
        m.pkB0 = s.b1q.w; m.pkB1 = s.b3q.z; m.pkB2 = s.b3q.w;
    };
    // part 2: texel rule, colour pipeline, stores; returns the lanes whose pixel goes to the repair queue
    auto part2 = [&](uint32_t r0, unsigned long long tA, unsigned long long tB, bool cA, bool cB, bool inA, bool inB, const Mid& m,
                     unsigned long long& mA, unsigned long long& mB, uint32_t& shA, uint32_t& shB) {
        const uint32_t pyA = ty_top + r0 + (tid >> 6), pyB = pyA + ROWS_PER_STEP;
        uint32_t* outA = a.fb + (size_t)pyA * W + px;
        uint32_t* outB = a.fb + (size_t)pyB * W + px;
        shA = m.pkA2 >> 24; shB = m.pkB2 >> 24;
        const float bA[3] = { m.bA0, m.bA1, m.bA2 }, bB[3] = { m.bB0, m.bB1, m.bB2 };
        // texture slot 0xFFFF = untextured: Color15::WHITE (render.rs:1585); then the transparency rule (render.rs:1591-1608)
        const bool noneA = ((m.pkA0 >> 24) | ((m.pkA1 >> 24) << 8)) == F_TEX_NONE, noneB = ((m.pkB0 >> 24) | ((m.pkB1 >> 24) << 8)) == F_TEX_NONE;
        uint32_t cA15 = noneA ? K::C15_WHITE : m.fetA, cB15 = noneB ? K::C15_WHITE : m.fetB;
        const bool btA = (shA & SH_BLACK_TR) != 0, btB = (shB & SH_BLACK_TR) != 0;
        const bool skipA = btA && (cA15 & ~K::C15_SEMI_BIT & 0xFFFFu) == 0, skipB = btB && (cB15 & ~K::C15_SEMI_BIT & 0xFFFFu) == 0;     // 0x0000 or black with black_transparent
        cA15 = cA15 == K::C15_TRANSPARENT ? K::C15_BLACK_DRAWABLE : cA15; cB15 = cB15 == K::C15_TRANSPARENT ? K::C15_BLACK_DRAWABLE : cB15;
        // (a winner that must replay its edge walk literally is not evaluated here: it joins the repair queue with bit 12 set)
        const bool slowA = (shA & SH_SLOW) != 0, slowB = (shB & SH_SLOW) != 0;
        const bool okA = cA && !skipA && !slowA, okB = cB && !skipB && !slowB;
        const uint32_t vA[3] = { m.pkA0 & 0xFFFFFFu, m.pkA1 & 0xFFFFFFu, m.pkA2 & 0xFFFFFFu }, vB[3] = { m.pkB0 & 0xFFFFFFu, m.pkB1 & 0xFFFFFFu, m.pkB2 & 0xFFFFFFu };
        const uint32_t flA = (shA & SH_DITHER) ? F_DITHER : 0u, flB = (shB & SH_DITHER) ? F_DITHER : 0u;
        const uint32_t sidA = cA ? sid_of(tA) : 0u, sidB = cB ? sid_of(tB) : 0u;
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
    };
    // ONE drain site (the queue's code -- the general per-pixel functions -- exists once in the kernel): entries of pixel A, of pixel B, then
    // the flush after the last step
    auto enqueue = [&](uint32_t r0, unsigned long long mA, unsigned long long mB, uint32_t shA, uint32_t shB) {
        const bool last = r0 + STEP_ROWS >= TH;
        if ((mA | mB) || (last && lqn)) {
            const uint32_t rowA = r0 + (tid >> 6), rowB = rowA + ROWS_PER_STEP;
#pragma unroll 1
            for (int which = 0; which < 3; ++which) {
                const unsigned long long m = which == 0 ? mA : which == 1 ? mB : 0ull;
                const uint32_t n = (uint32_t)__builtin_popcountll(m);
                if (which == 2 ? (last && lqn) : (lqn + n > 64u)) drain();
                if ((m >> lane) & 1ull) wq[lqn + (uint32_t)__builtin_popcountll(m & below)] = ((which ? rowB : rowA) << 6) | col | (((which ? shB : shA) & SH_SLOW) ? 0x1000u : 0u);
                lqn += n;
            }
        }
    };
    auto nothing_here = [&](uint32_t r0, bool inA, bool inB) {       // a step in which none of the wave's pixels is covered
        const uint32_t pyA = ty_top + r0 + (tid >> 6), pyB = pyA + ROWS_PER_STEP;
        if (inA) leave(pyA);
        if (inB) leave(pyB);
    };

    if (PIPE) {
        // Iteration k: [flush point] -> gathers of step k issued -> texel rule, colours and stores of step k - 1 (its texels were requested at
        // the end of the previous iteration) -> barycentrics and texel fetches of step k.  What crosses the loop's back edge is `Mid` (fourteen
        // registers and the step's flags), never a record set.
        // The repair queue holds up to RQ_WORDS = 192 entries here and is drained only at the top of an iteration, where nothing is in flight
        // but the two texel fetches: a drain is the general per-pixel code -- about a hundred registers -- and a record set live across it
        // would be spilled on every step, not only on the rare ones that drain.  A step appends at most 128 entries.
        const uint32_t n_steps = (TH + STEP_ROWS - 1) / STEP_ROWS;
        Mid m;
        m.bA0 = m.bA1 = m.bA2 = m.bB0 = m.bB1 = m.bB2 = 0.0f;
        m.pkA0 = m.pkA1 = m.pkA2 = m.pkB0 = m.pkB1 = m.pkB2 = 0u; m.fetA = m.fetB = 0u;
        unsigned long long ptA = 0, ptB = 0;
        bool pcA = false, pcB = false, pinA = false, pinB = false;
        for (uint32_t k = 0;; ++k) {
            const bool flush = k > n_steps;
            while (lqn > (flush ? 0u : 64u)) {             // ONE drain site: the queue's last (up to) 64 entries, one per lane
                const uint32_t base = lqn > 64u ? lqn - 64u : 0u;
                uint32_t* q = wq;
                lqn -= base; wq = q + base;
                drain();                                    // (reads wq[lane] for lane < lqn, leaves lqn = 0)
                wq = q; lqn = base;
            }
            if (flush) break;
#if B32_SHADE_PIPE == 2
            In cur;
            if (k < n_steps) issue(k * STEP_ROWS, cur);
            __builtin_amdgcn_sched_barrier(0);
            if (k >= 1) {
                const uint32_t r0 = (k - 1) * STEP_ROWS;
                unsigned long long mA = 0, mB = 0;
                uint32_t shA = 0, shB = 0;
                part2(r0, ptA, ptB, pcA, pcB, pinA, pinB, m, mA, mB, shA, shB);
                if (mA | mB) {                              // append (never drains: see above)
                    const uint32_t rowA = r0 + (tid >> 6), rowB = rowA + ROWS_PER_STEP;
                    if ((mA >> lane) & 1ull) wq[lqn + (uint32_t)__builtin_popcountll(mA & below)] = (rowA << 6) | col | ((shA & SH_SLOW) ? 0x1000u : 0u);
                    lqn += (uint32_t)__builtin_popcountll(mA);
                    if ((mB >> lane) & 1ull) wq[lqn + (uint32_t)__builtin_popcountll(mB & below)] = (rowB << 6) | col | ((shB & SH_SLOW) ? 0x1000u : 0u);
                    lqn += (uint32_t)__builtin_popcountll(mB);
                }
            }
            if (k < n_steps) {
                part1(k * STEP_ROWS, cur, m);
                ptA = cur.tA; ptB = cur.tB; pcA = cur.cA; pcB = cur.cB; pinA = cur.inA; pinB = cur.inB;
            }
        }
#else
            // texel stage only: the step's own gather is waited for, its texels are requested, and while they travel the PREVIOUS step's texel
            // rule, colours and stores run
            Mid mc = m;
            unsigned long long ctA = 0, ctB = 0; bool ccA = false, ccB = false, cinA = false, cinB = false;
            if (k < n_steps) {
                In cur;
                issue(k * STEP_ROWS, cur);
                part1(k * STEP_ROWS, cur, mc);
                ctA = cur.tA; ctB = cur.tB; ccA = cur.cA; ccB = cur.cB; cinA = cur.inA; cinB = cur.inB;
            }
            __builtin_amdgcn_sched_barrier(0);
            if (k >= 1) {
                const uint32_t r0 = (k - 1) * STEP_ROWS;
                unsigned long long mA = 0, mB = 0;
                uint32_t shA = 0, shB = 0;
                part2(r0, ptA, ptB, pcA, pcB, pinA, pinB, m, mA, mB, shA, shB);
                if (mA | mB) {                              // append (never drains: see above)
                    const uint32_t rowA = r0 + (tid >> 6), rowB = rowA + ROWS_PER_STEP;
                    if ((mA >> lane) & 1ull) wq[lqn + (uint32_t)__builtin_popcountll(mA & below)] = (rowA << 6) | col | ((shA & SH_SLOW) ? 0x1000u : 0u);
                    lqn += (uint32_t)__builtin_popcountll(mA);
                    if ((mB >> lane) & 1ull) wq[lqn + (uint32_t)__builtin_popcountll(mB & below)] = (rowB << 6) | col | ((shB & SH_SLOW) ? 0x1000u : 0u);
                    lqn += (uint32_t)__builtin_popcountll(mB);
                }
            }
            m = mc; ptA = ctA; ptB = ctB; pcA = ccA; pcB = ccB; pinA = cinA; pinB = cinB;
        }
#endif
    } else {
        for (uint32_t r0 = 0; r0 < TH; r0 += STEP_ROWS) {
            In cur;
            issue(r0, cur);
            unsigned long long mA = 0, mB = 0;
            uint32_t shA = 0, shB = 0;
            if (cur.any) {
                Mid m;
                part1(r0, cur, m);
                part2(r0, cur.tA, cur.tB, cur.cA, cur.cB, cur.inA, cur.inB, m, mA, mB, shA, shB);
            } else nothing_here(r0, cur.inA, cur.inB);
            enqueue(r0, mA, mB, shA, shB);
        }
    }
}

}  // namespace b32

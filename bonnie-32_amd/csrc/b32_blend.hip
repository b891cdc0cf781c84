// b32_blend.hip -- k_blend: the surfaces of the transparent pass (render.rs:2563-2569) blend against the framebuffer.  What must be
// ordered is, per pixel, the sequence of that pixel's own fragments: per tile, on an LDS copy of it, a lane owns a pixel column and
// walks the surfaces whose box holds its pixels in painter's order (row / column masks of the batch's 64 surfaces):
// set_pixel_blended_15 / editor-alpha stores (render.rs:479-502, 567-628); no atomics, no ordering between surfaces that do not share
// a pixel.  Also the ordered walk of whole tile lists (x-ray; 8-bit path with blending texels or editor alpha).
#include "b32_fill_common.h"

namespace b32 {

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
    // (the frame's "started" word, when the fill left it to this kernel: FillArgs::start_defer)
    if (GATHER && a.start_seq && a.start_defer == 1u && blockIdx.x == 0 && threadIdx.x == 0)
        (void)__hip_atomic_exchange(&events_of(a.ctrl)->fill_started, a.start_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    // (the fill gave up on this draw's setup kernel -- polled hand-over, or k_join's Events::join_abort: there are no lists of this frame)
    if (GATHER && a.join_seq && events_of(a.ctrl)->poll_lost == a.join_seq) return;
    if (GATHER && a.direct_bin && events_of(a.ctrl)->join_abort == a.epoch) return;
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

template <bool FMT8, bool GATHER>
static void launch_blend_t(hipStream_t s, const FillArgs& a, uint32_t ntiles) {
    constexpr int NT = BLEND_NT;
    const bool zmode = a.fp.zmode && !a.fp.xray;
    const size_t lds = blend_lds_bytes(FMT8 && zmode);
    static bool attr[64] = {};
    if (first_launch_on_device(attr)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_blend<NT, FMT8, GATHER>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((k_blend<NT, FMT8, GATHER>), dim3(ntiles), dim3(NT), lds, s, a);
}
void launch_blend(hipStream_t s, const FillArgs& a, uint32_t ntiles, bool fmt8, bool gather) {
    if (gather) launch_blend_t<false, true>(s, a, ntiles);           // (the sort-free path's transparent pass: RGB555 only)
    else if (fmt8) launch_blend_t<true, false>(s, a, ntiles);
    else launch_blend_t<false, false>(s, a, ntiles);
}

}  // namespace b32

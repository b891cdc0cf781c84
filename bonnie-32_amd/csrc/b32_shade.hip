// b32_shade.hip -- k_shade of the keyed pipelines (b32_set_routes(B32_ROUTE_SORT_FREE) / frames the sort-free path does not take): one lane
// per pixel: winner of the visibility buffer -> surface record -> barycentrics -> texel -> colour pipeline (render.rs:1613-1661) -> RGBA8
// store (Color15::to_rgba), 256-B coalesced per wave.  If the winner's texel is skipped (CHEAP coverage only) the wave scans the tile
// list downward, 64 entries at a time, for the highest surface below it whose fragment is really drawn -- identical to EXACT coverage.
#include "b32_fill_common.h"

namespace b32 {

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

void launch_shade(hipStream_t s, const FillArgs& a, uint32_t ntiles) {
    const dim3 g(((ntiles + 7) / 8) * 8 * 4);
    if (a.fp.fmt8) hipLaunchKernelGGL((k_shade<true>), g, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_shade<false>), g, dim3(256), 0, s, a);
}

}  // namespace b32

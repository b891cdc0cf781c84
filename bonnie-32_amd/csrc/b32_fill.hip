// b32_fill.hip -- affine texture-mapped triangle fill with RGB555 dither (rasterize_triangle_15, render.rs:1440-1714): k_cover, the tile
// kernel of every pipeline (MI355X-first, not the reference's per-triangle scanline loop), and launch_fill, which picks its form.
//
//   k_cover  persistent workgroups pull 64x64 screen tiles from a device-side cursor.
//            sort-free forms (P64; the default path): coverage (b32_cover.h) decides visibility by the surfaces' global painter's
//            priority, then the same workgroup shades the tile straight from the LDS winners (b32_shade_tile.h): one launch, no
//            visibility buffer in HBM; k_blend (b32_blend.hip) follows when the mesh has a transparent pass.
//            keyed forms: coverage of a tile list in painter's order, winners to the visibility buffer; k_shade (b32_shade.hip) and
//            k_blend follow.
#include "b32_cover.h"
#include "b32_shade_tile.h"

#ifndef B32_SHADE_PIPE
#define B32_SHADE_PIPE 0          // experiment switch, OFF (round 6): software-pipelined straight-line shading (shade_tile_plain).  1 = the texel stage
                                  // (this step's texels travel beside the previous step's colours and stores: 111 VGPRs, nothing spilled), 2 = the gather
                                  // stage (the next step's record gathers beside this step's texels and colours: 27 VGPRs spilled under the 112 cap,
                                  // 8 without it).  Same-box A/B: C3 0.1061-0.1076 (1) / 0.140-0.142 (2) / 0.114-0.115 (2, uncapped) against 0.1044-0.1067,
                                  // k_cover 88.8-89.4 / 114-116 / 89.6-90.1 against 87.0-87.8 us -- a wave's own waits (tools/timeline.py: 1.3 us for the
                                  // gather, 0.75 us for the texels of a 3-us step) are already filled by the CU's other fifteen waves
                                  // (profiles/r06_shade_pipe_ab.txt)
#endif

namespace b32 {

// per-wave repair queue of the fused kernel's shading phase (b32_shade_tile.h): 64 words; 192 in the experiment builds with the
// software-pipelined shading step (it drains only where no record set is in flight and appends up to 128 entries in between), which
// then leave ~2 KB instead of ~6 KB for a staged index atlas beside two workgroups' planes
constexpr uint32_t RQ_WORDS = B32_SHADE_PIPE ? 192 : 64, RQ_BYTES = RQ_WORDS * 4;
static_assert(2 * (4 * LDS_TILE_BYTES + LDS_MISC_BYTES + 8 * RQ_BYTES + 511) / 512 * 512 <= 160 * 1024, "two 8-wave workgroups per CU");

// ------------------------------------------------------------------------------------------------ k_cover
template <bool FMT8, int NT, bool ZMODE>
__device__ __forceinline__ void shade_tile_p64(const FillArgs& a, const uint32_t* tilebuf, uint32_t e0, uint32_t e1, uint32_t x_lo, uint32_t x_hi,
                                               uint32_t y_lo, uint32_t y_hi, uint32_t ty_top, uint32_t tid, uint32_t lane, uint32_t TH, uint32_t* rq,
                                               const uint8_t* latlas);

template <int NT, bool ZMODE, bool PIPE>
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

// PLAIN: the configuration BASELINE.json's metric is quoted on, with its run-time switches turned into constants -- affine UVs, no
// shading pass, fixed-point snapping, perspective camera, one texture, lists from the binning launch, no transparent pass.  The
// compiler then drops the other branches of coverage and shading from this instantiation.
//   PLAIN == 1: RGB555 texels fetched from global memory, texture of non-zero size: the straight-line shading and nothing else;
//   PLAIN == 2: the other plain frames (8-bit-per-channel target, the index atlas in LDS, a zero-sized texture): the general shading;
//   PLAIN == 3: PLAIN == 1 with a shading pass (flat / Gouraud: the settings the reference's callers use, RasterSettings::game() and
//               ::default(), types.rs:1455-1495): the shading mode stays a run-time value, the shades come from the setup kernel.
// The body is a device function so that one instantiation can also be compiled under a register cap (k_cover_plain below).
template <int TEXMODE, bool EXACT, int NT, bool ZMODE, bool FMT8, bool P64, int PLAIN>
__device__ __forceinline__ void cover_body(const FillArgs& a_in) {
    FillArgs a_plain = a_in;
    if (PLAIN) { a_plain.fp.affine = 1; a_plain.fp.fixed_point = 1; a_plain.fp.ortho = 0; a_plain.fp.nt = 1; a_plain.inline_bin = 0; a_plain.gather_blend = 0; }
    if (PLAIN == 1 || PLAIN == 2) { a_plain.fp.shading = B32_SHADE_NONE; a_plain.fp.n_lights = 0; a_plain.shades = nullptr; }
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
        uint4* dst = reinterpret_cast<uint4*>(smem + TB + LDS_MISC_BYTES + NW * RQ_BYTES);
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
    // "this kernel has started", for the setup kernel that will next write the frame set the PREVIOUS fill on this stream read (k_gate): a
    // device-scope atomic, visible to a poller on any XCD at once (a plain store would sit in this XCD's L2 until the kernel ends)
    if (P64 && a.start_seq && !a_in.start_defer && blockIdx.x == 0 && tid == 0)
        (void)__hip_atomic_exchange(&events_of(a.ctrl)->fill_started, a.start_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (sort-free forms: workgroup 0 also notes the shader-cycle counter now and, with the wall clock, when it runs out of tiles -- the
    // shader clock the fill really ran at, b32_last_shader_clock: under this kernel's load it sits below the device's nominal clock)
    const unsigned long long clk_entry = (P64 && blockIdx.x == 0 && tid == 0) ? (unsigned long long)clock64() : 0ull;
    // small meshes (inline_bin): the first NT spans (and class bits) are requested before the counters are reduced, and the skip mask is
    // staged before it too -- a C1 workgroup used to start its tile 7 us into an 18-us kernel behind three dependent round trips
    // The setup -> fill hand-over polled here (FillArgs::join_seq: the merged draws of a batched frame): everything above reads scene data
    // only; nothing k_setup writes -- spans, counters, event words, records, the control block it resets when it starts -- is touched before
    // this point.  The kernel boundaries still do the cache maintenance on the writer's side (k_flag_poll runs behind the setup kernel's
    // end-of-kernel release).  The acquire only where the workgroup really waited: a value found at the first look was published before
    // anything of this kernel could have cached a line the setup kernel wrote (the kernel's own start invalidated the caches), and the fence
    // is not free -- it invalidates this CU's L1 and the XCD's whole L2 (by every wave of every workgroup it stretched C2's fill from 23 to
    // 40 us: profiles/r06_poll_join_ab.txt).  One wave does it for the workgroup: the waves share the L1.
    bool join_lost = false;
    if (P64 && a.join_seq) {
        if (tid == 0) {
            Events* ev = events_of(a.ctrl);
            const unsigned long long t0 = wall_clock64();
            uint32_t lost = 0, waited = 0;
            while (__hip_atomic_fetch_add(&ev->poll_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.join_seq) {
                if (wall_clock64() - t0 > (unsigned long long)a.join_patience) { lost = 1; break; }
                waited = 1;
                __builtin_amdgcn_s_sleep(32);           // (every workgroup polls one word: a short sleep is a storm of atomics on one address)
            }
            if (lost) { atomicOr(&a.ctrl->sticky, 8u); (void)__hip_atomic_exchange(&ev->poll_lost, a.join_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            if (waited) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            misc[7] = lost;
        }
        __syncthreads();
        join_lost = misc[7] != 0;
    }
    phase_stamp(a.ctrl, ST_FILL);
    uint32_t pre_span = 0xFFFFFFFFu, pre_key = 0u;
    if (P64 && a.inline_bin && tid < fp.nf && !join_lost) { pre_span = a.spans[tid]; if (a.gather_blend) pre_key = a.keys[tid]; }
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
        bool reduce = a.inline_bin != 0, join_abort = join_lost;
        uint32_t redraw = 0;
        if (a.direct_bin) {
            const Events* ev = events_of(a.ctrl);
            reduce = ev->bad_index == a.epoch || ev->nan_opaque == a.epoch || ev->nan_transparent == a.epoch;
            redraw = (ev->overflow == a.epoch ? 2u : 0u) | (ev->long_transparent == a.epoch ? 1u : 0u);
            join_abort = join_lost || ev->join_abort == a.epoch;     // (k_join gave up on this frame's setup kernel: nothing of its output may be read)
            // (no such event: the frame is not aborted, and workgroup 0 reduces the counters for the host AFTER its tiles -- with a
            // million faces the reduction takes microseconds, and in a narrow band every workgroup has one tile: it was the kernel's
            // critical path)
            reduce_late = !reduce && blockIdx.x == 0;
        }
        if (tid == 0) misc[6] = 0;
        __syncthreads();
        if (reduce && !join_abort) reduce_setup_counters<NT>(a, misc, tid, lane);
        if (join_abort) {
            // (the tile counters of a setup kernel that did finish -- only its flag was missed -- are zeroed for the next frame of this set, as the
            // redraw path below does; what a setup kernel that is STILL running leaves behind is re-zeroed by b32_frame_finish, which reports the frame)
            if (a.direct_bin)
                for (uint32_t t = blockIdx.x * NT + tid; t < ntiles; t += gridDim.x * NT) { uint32_t* fl = a.tile_fill + (size_t)t * FILL_PAD; fl[0] = 0; fl[1] = 0; }
            if (a.clear_on) clear_band<NT>(a);
            return;
        }
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
#ifdef B32_TIMELINE
    if (lane < 32) B32_DBG_SLOTS(tilebuf)[lane] = 0ull;
#endif
    uint32_t next_tile = blockIdx.x;
    for (;;) {
        B32_CLK_DECL(clkh);
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
        B32_CLK_ADD(a, 9, clkh);                 // tile header: tile index, list range, plane clear, barrier
#ifdef B32_TIMELINE
        const unsigned long long tl0 = wall_clock64();
#endif
        if (n_op) {
            frag_count += phase_a_rows<TEXMODE, EXACT, NW, ZMODE, FMT8, P64>(a, e0, n_op, lane, wave, &misc[2], wmarks + wave * RQ_WORDS, lds_desc, tilebuf,
                                                           x_lo, x_hi, y_lo, y_hi, ty_top, ltex);
            B32_CLK_DECL(clkb);
            __syncthreads();
            B32_CLK_ADD(a, 8, clkb);            // this wave's wait at the barrier behind the coverage
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
                if (PLAIN != 2 && (PLAIN == 1 || PLAIN == 3 || (!FMT8 && fp.affine && fp.fixed_point && !fp.ortho && fp.nt == 1 && !latlas && a.tex0.width && a.tex0.height &&
                                                  (fp.shading == B32_SHADE_NONE || a.shades))))
                    shade_tile_plain<NT, ZMODE, (B32_SHADE_PIPE != 0) && PLAIN == 1>(a, tilebuf, e0, e1, x_lo, x_hi, y_lo, y_hi, ty_top, tid, lane, TH, wmarks + wave * RQ_WORDS);
                else if (PLAIN == 0 || PLAIN == 2) shade_tile_p64<FMT8, NT, ZMODE>(a, tilebuf, e0, e1, x_lo, x_hi, y_lo, y_hi, ty_top, tid, lane, TH, wmarks + wave * RQ_WORDS, latlas);
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
#ifdef B32_TIMELINE
    if (P64 && a.dbg && lane < 32 && B32_DBG_SLOTS(tilebuf)[lane]) atomicAdd(a.dbg + 1 + 4 * 8192 + lane, B32_DBG_SLOTS(tilebuf)[lane]);
#endif
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

// (every form is compiled for at least 4 waves per SIMD, i.e. at most 128 VGPRs: the EXACT z-buffer forms had drifted to 129, which halves
// the 512-thread kernel's residency to one workgroup per CU -- game() settings with colour-keyed textures at 2560x1920: 0.289 -> 0.242 ms)
template <int TEXMODE, bool EXACT, int NT, bool ZMODE, bool FMT8 = false, bool P64 = false, int PLAIN = 0>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu((P64 && NT == 512 && !PLAIN) ? B32_P64_WAVES : 4))) void k_cover(FillArgs a_in) {
    cover_body<TEXMODE, EXACT, NT, ZMODE, FMT8, P64, PLAIN>(a_in);
}

// The benchmark's instantiation (painter's mode, CHEAP coverage, PLAIN == 1) under a cap of 112 VGPRs: four of its waves then leave a SIMD
// 64 registers, one wave of the NEXT frame's 58-register setup kernel, which otherwise waits for a fill workgroup to retire (two frames in
// flight: C3 0.1166 -> 0.1129 ms/frame, C5 0.196 -> 0.185 on the same box, profiles/r05_vgpr112_ab.txt).  The uncapped body allocates 113.
// amdgpu_num_vgpr counts HALF the unified VGPR+AGPR file on gfx90a and later (56 -> 112); it does not take a template-dependent value,
// hence a kernel of its own.
#ifndef B32_PLAIN_CAP
#define B32_PLAIN_CAP 56          // (48 = 96 VGPRs, room for two setup waves, 13 spilled: no different -- profiles/r05_vgpr_cap_96_104_112.txt)
#endif
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4), amdgpu_num_vgpr(B32_PLAIN_CAP))) void k_cover_plain(FillArgs a_in) {
    cover_body<0, false, 512, false, false, true, 1>(a_in);
}
// (z-buffer mode without a shading pass: the plain setup kernel co-resides as on the painter's path)
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4), amdgpu_num_vgpr(56))) void k_cover_plain_z(FillArgs a_in) {
    cover_body<0, false, 512, true, false, true, 1>(a_in);
}
// (the 8-bit-per-channel target, render_mesh: painter's mode, general shading)
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4), amdgpu_num_vgpr(56))) void k_cover_plain8(FillArgs a_in) {
    cover_body<0, false, 512, false, true, true, 2>(a_in);
}
// (game() / default() -- z-buffer mode with a shading pass -- in the straight-line form and under the same cap: 121 -> 112 VGPRs, 11 spilled.
// Their setup kernel needs 70-80 registers, so nothing co-resides; the form itself and the cap are worth 1 % each, profiles/r05_lit_form_ab.txt)
#ifndef B32_LIT_VGPR
#define B32_LIT_VGPR 56
#endif
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4), amdgpu_num_vgpr(B32_LIT_VGPR))) void k_cover_lit(FillArgs a_in) {
    cover_body<0, false, 512, true, false, true, 3>(a_in);
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
    const size_t lds_n = 4 * LDS_TILE_BYTES + LDS_MISC_BYTES + 8 * RQ_BYTES + atlas, lds_w = 4 * LDS_TILE_BYTES + LDS_MISC_BYTES + 16 * RQ_BYTES + atlas;
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
        if (!dbg) (void)hipMalloc(reinterpret_cast<void**>(&dbg), (1 + 4 * 8192 + 64) * 8);
        (void)hipMemsetAsync(dbg, 0, 8, s);
        (void)hipMemsetAsync(dbg + 1 + 4 * 8192, 0, 64 * 8, s);
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
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_cover<0, EXACT, 1024, ZMODE, FMT8, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL((k_cover<0, EXACT, 1024, ZMODE, FMT8, true, 2>), dim3(min(ntiles, (uint32_t)n_cu)), dim3(1024), lds_w, s, a);
        return;
    }
#endif
    // (the straight-line shading's frames: see cover_body)
    const bool straight = !FMT8 && !a.atlas_idx_bytes && a.tex0.width && a.tex0.height;
    const bool lit_plain = a.fp.affine && a.fp.shading != B32_SHADE_NONE && a.shades && a.fp.fixed_point && !a.fp.ortho && a.fp.nt == 1 && !a.inline_bin && !a.gather_blend;
    if constexpr (!EXACT && ZMODE && !FMT8) {
        if (lit_plain && straight && !wide) {
            static bool attr_lit[64] = {};
            if (first_launch_on_device(attr_lit)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_cover_lit), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipLaunchKernelGGL(k_cover_lit, g, dim3(512), lds_n, s, a);
            return;
        }
    }
    if (plain && !wide) {
        static bool attr_plain[64] = {};
        const bool first = first_launch_on_device(attr_plain);
        if (first) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_cover<0, EXACT, 512, ZMODE, FMT8, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if constexpr (!EXACT && !ZMODE) {
            if (first) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_cover_plain), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (straight) { hipLaunchKernelGGL(k_cover_plain, g, dim3(512), lds_n, s, a); return; }
        } else if constexpr (!EXACT && ZMODE) {
            if (first) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_cover_plain_z), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (straight) { hipLaunchKernelGGL(k_cover_plain_z, g, dim3(512), lds_n, s, a); return; }
        } else {
            if (first) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_cover<0, EXACT, 512, ZMODE, false, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (straight) { hipLaunchKernelGGL((k_cover<0, EXACT, 512, ZMODE, false, true, 1>), g, dim3(512), lds_n, s, a); return; }
        }
        if constexpr (!EXACT && !ZMODE && FMT8) {
            if (first) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_cover_plain8), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipLaunchKernelGGL(k_cover_plain8, g, dim3(512), lds_n, s, a);
            return;
        }
        hipLaunchKernelGGL((k_cover<0, EXACT, 512, ZMODE, FMT8, true, 2>), g, dim3(512), lds_n, s, a);
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
        launch_blend(s, a, ntiles, f8, false);
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
        if (a.gather_blend) launch_blend(s, a, ntiles, false, true);
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
    if (band_h) launch_shade(s, a, ntiles);
    if (a.may_blend && !f8) launch_blend(s, a, ntiles, false, false);
}

size_t fill_lds_tex_budget() { return 160 * 1024 - LDS_TEX_OFFSET - 16; }
uint32_t fill_lds_atlas_room(bool wide) {
    // 160 KB of LDS per CU, allocated in 512-byte granules: one 16-wave workgroup, or two 8-wave workgroups side by side
    const uint32_t total = 160u * 1024u, gran = 512u;
    if (wide) return total - (uint32_t)(4 * LDS_TILE_BYTES + LDS_MISC_BYTES + 16 * RQ_BYTES) - gran;
    return total / 2u - (uint32_t)(4 * LDS_TILE_BYTES + LDS_MISC_BYTES + 8 * RQ_BYTES) - gran;
}

}  // namespace b32

// b32_sort.hip — stable LSD radix sort (8-, 11- or 12-bit digits) over (u32 key, u32 value) pairs, wave64-native.
//
// Replaces the reference's `sort_by` merge sort of 208-byte Surface structs (render.rs:2527-2541): the painter's key is
// reduced to a 32-bit radix key by k_setup and only (key, surface id) pairs move.  Stability of every pass is what makes
// equal keys keep face order, exactly like slice::sort_by.
//
// One pass = k_hist (per-block digit histogram, LDS atomics) -> k_scan_rows (one workgroup per digit scans its row of the
// digit-major table) -> k_scatter (per-wave match-any ranking with 8 ballots, no atomics, deterministic).
// The element count lives in device memory (it is produced by the previous kernel); grids are sized for the capacity
// and surplus workgroups exit at once.
#include "b32_device.h"

namespace b32 {

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Each wave of a block owns a contiguous run of 16 x 64 elements, read 64 at a time (256-B coalesced loads), so that
// "earlier element" == "earlier step, or same step and lower lane" inside a wave, and waves are ordered by index.
// BITS = digit width (8, 11 or 12): NB = 2^BITS bins.
template <int BITS>
__global__ __launch_bounds__(SORT_THREADS) void k_hist(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ n_dev,
                                                        int shift, int drop_invalid, uint32_t* __restrict__ block_hist, uint32_t max_blocks,
                                                        Ctrl* __restrict__ stamp_ctrl) {
    constexpr uint32_t NB = 1u << BITS;
    __shared__ uint32_t hist[NB];
    if (stamp_ctrl) phase_stamp(stamp_ctrl, ST_BIN);
    const uint32_t n = *n_dev;
    const uint32_t base = blockIdx.x * SORT_TILE;
    if (base >= n) return;   // the scan only looks at the ceil(n / SORT_TILE) blocks in use
    for (uint32_t d = threadIdx.x; d < NB; d += SORT_THREADS) hist[d] = 0;
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t wbase = base + wave * (SORT_ITEMS * 64);
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        uint32_t idx = wbase + i * 64 + lane;
        if (idx < n) {
            uint32_t k = keys[idx];
            if (!(drop_invalid && k == KEY_INVALID)) atomicAdd(&hist[(k >> shift) & (NB - 1)], 1u);
        }
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < NB; d += SORT_THREADS) block_hist[(size_t)d * max_blocks + blockIdx.x] = hist[d];
}

// Reduction of k_setup's per-block counters: derive n_opaque and decide whether the frame may draw at all.  The reference
// panics before drawing on an out-of-range vertex index (render.rs:2375) or when a sort comparison sees NaN
// (render.rs:2531, lists of >= 2 elements).  Called by one 256-thread block.
__device__ void reduce_setup_partials(Ctrl* ctrl, const uint32_t* __restrict__ partials, uint32_t nblocks) {
    __shared__ uint32_t red[4][5];
    uint32_t acc[5] = { 0, 0, 0, 0, 0 };
    for (uint32_t b = threadIdx.x; b < nblocks; b += 256)
        for (int k = 0; k < 5; ++k) acc[k] += partials[b * 8 + k];
    for (int k = 0; k < 5; ++k)
        for (int off = 32; off > 0; off >>= 1) acc[k] += __shfl_down(acc[k], off);
    if ((threadIdx.x & 63) == 0) for (int k = 0; k < 5; ++k) red[threadIdx.x >> 6][k] = acc[k];
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t[5] = { 0, 0, 0, 0, 0 };
        for (int w = 0; w < 4; ++w) for (int k = 0; k < 5; ++k) t[k] += red[w][k];
        const uint32_t n_op = t[0] - t[1];
        ctrl->n_visible = t[0]; ctrl->n_transparent = t[1]; ctrl->nan_opaque = t[2]; ctrl->nan_transparent = t[3];
        ctrl->err_index = t[4] ? 1u : 0u;
        ctrl->n_opaque = n_op;
        if (t[4]) { ctrl->abort = 1; ctrl->sticky |= 1u; }
        if ((t[2] && n_op >= 2) || (t[3] && t[1] >= 2)) { ctrl->abort = 1; ctrl->sticky |= 2u; }
    }
    __syncthreads();
}

// Row d of block_hist[NB][max_blocks] (one workgroup per digit): exclusive scan over the blocks, in place, and the
// row total into digit_total[d].  The scan across digits is folded into k_scatter.
// `post` (first depth pass only): block 0 also reduces k_setup's per-block counters into Ctrl (former k_after_setup).
__global__ __launch_bounds__(256) void k_scan_rows(uint32_t* __restrict__ block_hist, uint32_t max_blocks, uint32_t nblocks,
                                                   uint32_t* __restrict__ digit_total, Ctrl* __restrict__ post_ctrl,
                                                   const uint32_t* __restrict__ partials, uint32_t npart, const uint32_t* __restrict__ n_dev) {
    __shared__ uint32_t wsum[4];
    if (n_dev) nblocks = min(nblocks, (*n_dev + SORT_TILE - 1) / SORT_TILE);   // blocks actually in use (grids are sized for the capacity)
    if (post_ctrl && blockIdx.x == 0) reduce_setup_partials(post_ctrl, partials, npart);
    __shared__ uint32_t carry_s;
    uint32_t* row = block_hist + (size_t)blockIdx.x * max_blocks;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nblocks; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < nblocks ? row[i] : 0;
        uint32_t inc = v;
        for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(inc, off); if (lane >= (uint32_t)off) inc += t; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t woff = carry_s;
        for (uint32_t w = 0; w < wave; ++w) woff += wsum[w];
        if (i < nblocks) row[i] = woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) carry_s += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) digit_total[blockIdx.x] = carry_s;
}

template <int BITS>
__global__ __launch_bounds__(SORT_THREADS) void k_scatter(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                           uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                           const uint32_t* __restrict__ n_dev, int shift, int drop_invalid,
                                                           const uint32_t* __restrict__ block_hist, uint32_t max_blocks,
                                                           const uint32_t* __restrict__ digit_total, uint32_t* __restrict__ ranges_out,
                                                           uint32_t n_ranges) {
    constexpr uint32_t NB = 1u << BITS;
    constexpr uint32_t PER = NB / SORT_THREADS;           // bins per thread in the prefix steps
    __shared__ uint32_t wcnt[4][NB];      // per-wave running digit counts, then exclusive prefix over waves (+ global base)
    __shared__ uint32_t dws[4];
    const uint32_t n = *n_dev;
    const uint32_t base = blockIdx.x * SORT_TILE;
    if (base >= n && !(ranges_out && blockIdx.x == 0)) return;
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    for (int w = 0; w < 4; ++w) for (uint32_t d = threadIdx.x; d < NB; d += SORT_THREADS) wcnt[w][d] = 0;
    // digit bases: exclusive scan of digit_total; thread t owns bins [t*PER, t*PER+PER)
    uint32_t dbase[PER];
    {
        uint32_t tot[PER], sum = 0;
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) { tot[j] = digit_total[threadIdx.x * PER + j]; sum += tot[j]; }
        uint32_t inc = sum;
        for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(inc, off); if (lane >= (uint32_t)off) inc += t; }
        if (lane == 63) dws[wave] = inc;
        __syncthreads();
        uint32_t run = inc - sum;
        for (uint32_t w = 0; w < wave; ++w) run += dws[w];
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) { dbase[j] = run; run += tot[j]; }
        // single-pass sort of (tile,class) keys: the digit bases ARE the list ranges (ranges[k] = first pair with key >= k)
        if (ranges_out && blockIdx.x == 0) {
#pragma unroll
            for (uint32_t j = 0; j < PER; ++j) { const uint32_t d = threadIdx.x * PER + j; if (d < n_ranges) ranges_out[d] = dbase[j]; }
            if (threadIdx.x == SORT_THREADS - 1 && n_ranges > NB) ranges_out[NB] = run;
        }
    }
    __syncthreads();
    if (base >= n) return;
    const uint32_t wbase = base + wave * (SORT_ITEMS * 64);
    uint32_t key[SORT_ITEMS], val[SORT_ITEMS], rnk[SORT_ITEMS];
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const uint32_t idx = wbase + i * 64 + lane;
        bool live = idx < n;
        uint32_t k = live ? keys_in[idx] : 0u;
        if (drop_invalid && k == KEY_INVALID) live = false;
        key[i] = k;
        val[i] = live ? (vals_in ? vals_in[idx] : idx) : 0u;
        const uint32_t d = (k >> shift) & (NB - 1);
        // peers = live lanes of this wave holding the same digit (BITS ballots)
        unsigned long long peers = __ballot(live);
#pragma unroll
        for (int b = 0; b < BITS; ++b) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        uint32_t before = 0;
        if (live) {
            const uint32_t leader = (uint32_t)__builtin_ctzll(peers);
            uint32_t old = 0;
            if (lane == leader) { old = wcnt[wave][d]; wcnt[wave][d] = old + (uint32_t)__popcll(peers); }
            old = __shfl(old, (int)leader);
            before = old + (uint32_t)__popcll(peers & lt_mask);
        }
        rnk[i] = live ? before : 0xFFFFFFFFu;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < PER; ++j) {   // exclusive prefix of each owned digit over the 4 waves + global base of (digit, block)
        const uint32_t d = threadIdx.x * PER + j;
        uint32_t run = dbase[j] + block_hist[(size_t)d * max_blocks + blockIdx.x];
        for (int w = 0; w < 4; ++w) { uint32_t c = wcnt[w][d]; wcnt[w][d] = run; run += c; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        if (rnk[i] != 0xFFFFFFFFu) {
            const uint32_t d = (key[i] >> shift) & (NB - 1);
            const uint32_t pos = wcnt[wave][d] + rnk[i];
            keys_out[pos] = key[i];
            vals_out[pos] = val[i];
        }
    }
}

// ---- sort-free fast path: tile lists by counting sort straight from k_setup's spans --------------------------------------
// The max-of-priorities coverage needs no order inside a tile list, so the (tile, surface) pairs are never materialised with
// keys and never ranked: k_count_spans histograms the tiles of 2048 faces in LDS (LDS atomics), k_scan_rows scans every tile's
// row across the blocks, k_place_spans re-reads the spans and drops each surface id at base[tile]++ (LDS atomic cursor).
#ifndef B32_SPAN_BLOCK
#define B32_SPAN_BLOCK 2048
#endif
constexpr uint32_t SPAN_BLOCK = B32_SPAN_BLOCK;           // faces per workgroup
constexpr uint32_t SPAN_MAX_TILES = 4096;       // LDS histogram capacity (rows: tiles, or 2 x tiles with the class split)

// With `keys` (scenes that can have a transparent pass) every tile gets two rows: opaque-class entries and transparent-class
// entries (bit 31 of k_setup's key, render.rs:2522-2523); k_place_spans lays a tile's list out as [opaque..., transparent...] and
// publishes the boundary in tile_mid, so k_cover takes the first part unordered and k_blend sorts only the second.
__global__ __launch_bounds__(256) void k_count_spans(uint32_t nf, uint32_t ntiles, uint32_t tiles_x, const uint32_t* __restrict__ spans,
                                                     const uint32_t* __restrict__ keys, uint32_t* __restrict__ block_hist, uint32_t max_blocks,
                                                     Ctrl* __restrict__ ctrl) {
    __shared__ uint32_t hist[SPAN_MAX_TILES];
    phase_stamp(ctrl, ST_BIN);
    const uint32_t nrows = keys ? 2 * ntiles : ntiles;
    for (uint32_t t = threadIdx.x; t < nrows; t += 256) hist[t] = 0;
    __syncthreads();
    const uint32_t f0 = blockIdx.x * SPAN_BLOCK;
    // all of the thread's spans are requested before the first is used: one memory latency per workgroup instead of one per face
    constexpr int PER_T = SPAN_BLOCK / 256;
    uint32_t sp[PER_T];
#pragma unroll
    for (int k = 0; k < PER_T; ++k) { const uint32_t f = f0 + threadIdx.x + (uint32_t)k * 256u; sp[k] = f < nf ? spans[f] : 0xFFFFFFFFu; }
#pragma unroll
    for (int k = 0; k < PER_T; ++k) {
        const uint32_t f = f0 + threadIdx.x + (uint32_t)k * 256u;
        const uint32_t span = sp[k];
        if (span == 0xFFFFFFFFu) continue;
        const uint32_t row0 = (keys && (keys[f] >> 31)) ? ntiles : 0u;
        const uint32_t tx0 = span & 0xFF, tx1 = (span >> 8) & 0xFF, ty0 = (span >> 16) & 0xFF, ty1 = span >> 24;
        for (uint32_t ty = ty0; ty <= ty1; ++ty)
            for (uint32_t tx = tx0; tx <= tx1; ++tx) atomicAdd(&hist[row0 + ty * tiles_x + tx], 1u);
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < nrows; t += 256) block_hist[(size_t)t * max_blocks + blockIdx.x] = hist[t];
}

__global__ __launch_bounds__(256) void k_place_spans(uint32_t nf, uint32_t ntiles, uint32_t tiles_x, const uint32_t* __restrict__ spans,
                                                     const uint32_t* __restrict__ keys, const uint32_t* __restrict__ block_hist, uint32_t max_blocks,
                                                     const uint32_t* __restrict__ digit_total, Ctrl* __restrict__ ctrl, uint32_t pair_cap,
                                                     uint32_t* __restrict__ ranges, uint32_t* __restrict__ tile_mid, uint32_t blend_cap,
                                                     uint32_t* __restrict__ pair_vals) {
    __shared__ uint32_t base[SPAN_MAX_TILES];
    __shared__ uint32_t dws[4], dmx[4];
    __shared__ uint32_t total_s;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // tile bases: exclusive scan of the row totals; thread t owns tiles [t*per, t*per + per)
    const uint32_t per = (ntiles + 255) / 256;                           // <= 16
    uint32_t tot[16], ttr[16], sum = 0, mx = 0;
    for (uint32_t j = 0; j < per; ++j) {
        const uint32_t t = threadIdx.x * per + j;
        tot[j] = t < ntiles ? digit_total[t] : 0u;
        ttr[j] = (keys && t < ntiles) ? digit_total[ntiles + t] : 0u;
        sum += tot[j] + ttr[j];
        mx = max(mx, ttr[j]);
    }
    uint32_t inc = sum;
    for (int off = 1; off < 64; off <<= 1) { const uint32_t v = __shfl_up(inc, off); if (lane >= (uint32_t)off) inc += v; }
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, off));
    if (lane == 63) dws[wave] = inc;
    if (lane == 0) dmx[wave] = mx;
    __syncthreads();
    uint32_t run = inc - sum;
    for (uint32_t w = 0; w < wave; ++w) run += dws[w];
    if (threadIdx.x == 255) total_s = run + sum;
    for (uint32_t j = 0; j < per; ++j) {
        const uint32_t t = threadIdx.x * per + j;
        if (t < ntiles) {
            base[t] = run + block_hist[(size_t)t * max_blocks + blockIdx.x];
            if (keys) base[ntiles + t] = run + tot[j] + block_hist[(size_t)(ntiles + t) * max_blocks + blockIdx.x];
            if (blockIdx.x == 0) { ranges[t] = run; if (tile_mid) tile_mid[t] = run + tot[j]; }   // list of tile t = [ranges[t], ranges[t+1])
        }
        run += tot[j] + ttr[j];
    }
    __syncthreads();
    const uint32_t total = total_s;
    const uint32_t longest_tr = max(max(dmx[0], dmx[1]), max(dmx[2], dmx[3]));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ranges[ntiles] = total;
        if (total > pair_cap) { ctrl->pairs_overflow = total; ctrl->abort = 1; ctrl->n_pairs = 0; }
        else ctrl->n_pairs = total;
        if (longest_tr > blend_cap) ctrl->need_global_sort = 1;          // k_blend's LDS sort cannot hold that list: nothing is drawn, the host redraws
    }
    if (total > pair_cap || longest_tr > blend_cap || ctrl->abort) return;
    const uint32_t f0 = blockIdx.x * SPAN_BLOCK;
    constexpr int PER_T = SPAN_BLOCK / 256;
    uint32_t sp[PER_T];
#pragma unroll
    for (int k = 0; k < PER_T; ++k) { const uint32_t f = f0 + threadIdx.x + (uint32_t)k * 256u; sp[k] = f < nf ? spans[f] : 0xFFFFFFFFu; }
#pragma unroll
    for (int k = 0; k < PER_T; ++k) {
        const uint32_t f = f0 + threadIdx.x + (uint32_t)k * 256u;
        const uint32_t span = sp[k];
        if (span == 0xFFFFFFFFu) continue;
        const uint32_t row0 = (keys && (keys[f] >> 31)) ? ntiles : 0u;
        const uint32_t tx0 = span & 0xFF, tx1 = (span >> 8) & 0xFF, ty0 = (span >> 16) & 0xFF, ty1 = span >> 24;
        for (uint32_t ty = ty0; ty <= ty1; ++ty)
            for (uint32_t tx = tx0; tx <= tx1; ++tx) pair_vals[atomicAdd(&base[row0 + ty * tiles_x + tx], 1u)] = f;
    }
}

// Meshes of at most SPAN_BLOCK faces (what the reference's callers submit per room / asset part): count, scan and place in ONE
// single-workgroup launch -- the block's own histogram is the whole histogram.
__global__ __launch_bounds__(256) void k_bin_small(uint32_t nf, uint32_t ntiles, uint32_t tiles_x, const uint32_t* __restrict__ spans,
                                                   const uint32_t* __restrict__ keys, const uint32_t* __restrict__ partials, uint32_t npart,
                                                   Ctrl* __restrict__ ctrl, uint32_t pair_cap, uint32_t* __restrict__ ranges,
                                                   uint32_t* __restrict__ tile_mid, uint32_t blend_cap, uint32_t* __restrict__ pair_vals) {
    __shared__ uint32_t hist[SPAN_MAX_TILES];
    __shared__ uint32_t dws[4], dmx[4];
    __shared__ uint32_t total_s;
    phase_stamp(ctrl, ST_BIN);
    reduce_setup_partials(ctrl, partials, npart);              // frame counters / abort decision (ends with a barrier)
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t nrows = keys ? 2 * ntiles : ntiles;
    for (uint32_t t = threadIdx.x; t < nrows; t += 256) hist[t] = 0;
    __syncthreads();
    for (uint32_t f = threadIdx.x; f < nf; f += 256) {
        const uint32_t span = spans[f];
        if (span == 0xFFFFFFFFu) continue;
        const uint32_t row0 = (keys && (keys[f] >> 31)) ? ntiles : 0u;
        const uint32_t tx0 = span & 0xFF, tx1 = (span >> 8) & 0xFF, ty0 = (span >> 16) & 0xFF, ty1 = span >> 24;
        for (uint32_t ty = ty0; ty <= ty1; ++ty)
            for (uint32_t tx = tx0; tx <= tx1; ++tx) atomicAdd(&hist[row0 + ty * tiles_x + tx], 1u);
    }
    __syncthreads();
    const uint32_t per = (ntiles + 255) / 256;                           // <= 16
    uint32_t tot[16], ttr[16], sum = 0, mx = 0;
    for (uint32_t j = 0; j < per; ++j) {
        const uint32_t t = threadIdx.x * per + j;
        tot[j] = t < ntiles ? hist[t] : 0u;
        ttr[j] = (keys && t < ntiles) ? hist[ntiles + t] : 0u;
        sum += tot[j] + ttr[j];
        mx = max(mx, ttr[j]);
    }
    uint32_t inc = sum;
    for (int off = 1; off < 64; off <<= 1) { const uint32_t v = __shfl_up(inc, off); if (lane >= (uint32_t)off) inc += v; }
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, off));
    if (lane == 63) dws[wave] = inc;
    if (lane == 0) dmx[wave] = mx;
    __syncthreads();
    uint32_t run = inc - sum;
    for (uint32_t w = 0; w < wave; ++w) run += dws[w];
    if (threadIdx.x == 255) total_s = run + sum;
    for (uint32_t j = 0; j < per; ++j) {                                 // counts -> cursors (list layout [opaque..., transparent...])
        const uint32_t t = threadIdx.x * per + j;
        if (t < ntiles) {
            hist[t] = run;
            if (keys) hist[ntiles + t] = run + tot[j];
            ranges[t] = run;
            if (tile_mid) tile_mid[t] = run + tot[j];
        }
        run += tot[j] + ttr[j];
    }
    __syncthreads();
    const uint32_t total = total_s;
    const uint32_t longest_tr = max(max(dmx[0], dmx[1]), max(dmx[2], dmx[3]));
    if (threadIdx.x == 0) {
        ranges[ntiles] = total;
        if (total > pair_cap) { ctrl->pairs_overflow = total; ctrl->abort = 1; ctrl->n_pairs = 0; }
        else ctrl->n_pairs = total;
        if (longest_tr > blend_cap) ctrl->need_global_sort = 1;
    }
    if (total > pair_cap || longest_tr > blend_cap || ctrl->abort) return;
    for (uint32_t f = threadIdx.x; f < nf; f += 256) {
        const uint32_t span = spans[f];
        if (span == 0xFFFFFFFFu) continue;
        const uint32_t row0 = (keys && (keys[f] >> 31)) ? ntiles : 0u;
        const uint32_t tx0 = span & 0xFF, tx1 = (span >> 8) & 0xFF, ty0 = (span >> 16) & 0xFF, ty1 = span >> 24;
        for (uint32_t ty = ty0; ty <= ty1; ++ty)
            for (uint32_t tx = tx0; tx <= tx1; ++tx) pair_vals[atomicAdd(&hist[row0 + ty * tiles_x + tx], 1u)] = f;
    }
}

bool bin_spans_applicable(const FrameParams& fp, const SortScratch& sc, bool with_class) {
    const uint32_t ntiles = fp.tiles_x * fp.tiles_y;
    const uint32_t nblocks = (fp.nf + SPAN_BLOCK - 1) / SPAN_BLOCK;
    return ntiles > 0 && (with_class ? 2 : 1) * ntiles <= SPAN_MAX_TILES && nblocks <= sc.max_blocks && fp.nf > 0;
}

bool launch_bin_spans(hipStream_t s, const FrameParams& fp, const uint32_t* spans, const uint32_t* keys, const uint32_t* partials, Ctrl* ctrl,
                      const SortScratch& sc, uint32_t pair_cap, uint32_t* ranges, uint32_t* tile_mid, uint32_t blend_cap, uint32_t* pair_vals) {
    if (!bin_spans_applicable(fp, sc, keys != nullptr)) return false;
    const uint32_t ntiles = fp.tiles_x * fp.tiles_y;
    const uint32_t nblocks = (fp.nf + SPAN_BLOCK - 1) / SPAN_BLOCK;
    const uint32_t nrows = keys ? 2 * ntiles : ntiles;
    if (nblocks == 1) {
        hipLaunchKernelGGL(k_bin_small, dim3(1), dim3(256), 0, s, fp.nf, ntiles, fp.tiles_x, spans, keys, partials, (fp.nf + 255) / 256, ctrl, pair_cap,
                           ranges, keys ? tile_mid : nullptr, blend_cap, pair_vals);
        return true;
    }
    hipLaunchKernelGGL(k_count_spans, dim3(nblocks), dim3(256), 0, s, fp.nf, ntiles, fp.tiles_x, spans, keys, sc.block_hist, sc.max_blocks, ctrl);
    hipLaunchKernelGGL(k_scan_rows, dim3(nrows), dim3(256), 0, s, sc.block_hist, sc.max_blocks, nblocks, sc.digit_total,
                       ctrl, partials, (fp.nf + 255) / 256, (const uint32_t*)nullptr);
    hipLaunchKernelGGL(k_place_spans, dim3(nblocks), dim3(256), 0, s, fp.nf, ntiles, fp.tiles_x, spans, keys, sc.block_hist, sc.max_blocks,
                       sc.digit_total, ctrl, pair_cap, ranges, keys ? tile_mid : nullptr, blend_cap, pair_vals);
    return true;
}

template <int BITS>
static void radix_pass_t(hipStream_t s, const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out,
                         const uint32_t* n_dev, uint32_t n_cap, int shift, const SortScratch& sc, const RadixExtra& ex) {
    const uint32_t nblocks = (n_cap + SORT_TILE - 1) / SORT_TILE;
    const int drop = vals_in == nullptr ? 1 : 0;
    hipLaunchKernelGGL(k_hist<BITS>, dim3(nblocks), dim3(SORT_THREADS), 0, s, keys_in, n_dev, shift, drop, sc.block_hist, sc.max_blocks, ex.post_ctrl);
    hipLaunchKernelGGL(k_scan_rows, dim3(1u << BITS), dim3(256), 0, s, sc.block_hist, sc.max_blocks, nblocks, sc.digit_total,
                       ex.post_ctrl, ex.partials, ex.npart, n_dev);
    hipLaunchKernelGGL(k_scatter<BITS>, dim3(nblocks), dim3(SORT_THREADS), 0, s, keys_in, vals_in, keys_out, vals_out, n_dev, shift, drop,
                       sc.block_hist, sc.max_blocks, sc.digit_total, ex.ranges_out, ex.n_ranges);
}

void launch_radix_pass(hipStream_t s, const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out,
                       const uint32_t* n_dev, uint32_t n_cap, int shift, int bits, const SortScratch& sc, const RadixExtra& ex) {
    if (n_cap == 0) return;
    if (bits == 11) radix_pass_t<11>(s, keys_in, vals_in, keys_out, vals_out, n_dev, n_cap, shift, sc, ex);
    else if (bits == 12) radix_pass_t<12>(s, keys_in, vals_in, keys_out, vals_out, n_dev, n_cap, shift, sc, ex);
    else radix_pass_t<8>(s, keys_in, vals_in, keys_out, vals_out, n_dev, n_cap, shift, sc, ex);
}

}  // namespace b32

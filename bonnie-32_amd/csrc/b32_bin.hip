// b32_bin.hip — screen-tile binning of the depth-sorted surface list.
//
// Input: order[r] = surface id of painter's rank r (r = 0 is drawn first).  Each surface is expanded into one pair per
// 64x64 screen tile its (band-clipped) bounding box touches: key = (tile << 1) | class (class 1 = transparent pass,
// render.rs:2522-2523), value = surface id.  Pairs are emitted in rank order with offsets from a prefix sum (no atomics,
// deterministic), then the radix sort of b32_sort.hip groups them by key; because that sort is stable, every tile's list
// stays in painter's order — the order the transparent pass must honour, and the order whose *last* opaque writer wins.
#include "b32_device.h"

namespace b32 {


constexpr int BIN_THREADS = 256;
constexpr int BIN_ITEMS = 16;
constexpr int BIN_TILE = BIN_THREADS * BIN_ITEMS;

// counts[r] = number of tiles surface order[r] touches; block_sums[b] = sum over the block's 4096 ranks.
__global__ __launch_bounds__(BIN_THREADS) void k_bin_count(FrameParams fp, const uint32_t* __restrict__ spans, const uint32_t* __restrict__ order,
                                                            const Ctrl* __restrict__ ctrl, uint32_t* __restrict__ counts,
                                                            uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t wsum[BIN_THREADS / 64];
    const uint32_t n = ctrl->n_visible;
    const uint32_t base = blockIdx.x * BIN_TILE;
    uint32_t local = 0;
    if (base < n) {
#pragma unroll 4
        for (int i = 0; i < BIN_ITEMS; ++i) {
            const uint32_t r = base + i * BIN_THREADS + threadIdx.x;
            if (r < n) {
                const uint32_t span = spans[order[r]];          // k_setup's packed tile span of the surface (pack_tile_span)
                counts[r] = span;
                if (span != 0xFFFFFFFFu) local += (((span >> 8) & 0xFF) - (span & 0xFF) + 1) * ((span >> 24) - ((span >> 16) & 0xFF) + 1);
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// Emit the pairs of each rank at its prefix offset (block base + in-block exclusive scan of counts).
__global__ __launch_bounds__(BIN_THREADS) void k_bin_emit(FrameParams fp, const uint32_t* __restrict__ order,
                                                           Ctrl* __restrict__ ctrl, const uint32_t* __restrict__ counts,
                                                           const uint32_t* __restrict__ block_sums, uint32_t nblocks, uint32_t pair_cap,
                                                           uint32_t* __restrict__ pair_keys, uint32_t* __restrict__ pair_vals) {
    __shared__ uint32_t wtot[BIN_THREADS / 64];
    __shared__ uint32_t step_base, total_s;
    const uint32_t n = ctrl->n_visible;
    const uint32_t base = blockIdx.x * BIN_TILE;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {   // every block derives its own base and the grand total from the (<= few hundred) block sums: no separate scan kernel
        uint32_t before = 0, all = 0;
        for (uint32_t i = threadIdx.x; i < nblocks; i += BIN_THREADS) { const uint32_t v = block_sums[i]; all += v; if (i < blockIdx.x) before += v; }
        for (int off = 32; off > 0; off >>= 1) { before += __shfl_down(before, off); all += __shfl_down(all, off); }
        if (lane == 0) { wtot[wave] = before; }
        __syncthreads();
        if (threadIdx.x == 0) step_base = wtot[0] + wtot[1] + wtot[2] + wtot[3];
        __syncthreads();
        if (lane == 0) wtot[wave] = all;
        __syncthreads();
        if (threadIdx.x == 0) total_s = wtot[0] + wtot[1] + wtot[2] + wtot[3];
        __syncthreads();
    }
    const uint32_t total = total_s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (total > pair_cap) { ctrl->pairs_overflow = total; ctrl->abort = 1; ctrl->n_pairs = 0; }
        else ctrl->n_pairs = total;
    }
    if (base >= n || total > pair_cap || ctrl->abort) return;
    const uint32_t n_opaque = ctrl->n_opaque;
    for (int i = 0; i < BIN_ITEMS; ++i) {
        const uint32_t r = base + i * BIN_THREADS + threadIdx.x;
        const uint32_t span = r < n ? counts[r] : 0xFFFFFFFFu;
        const uint32_t tx0 = span & 0xFF, tx1 = (span >> 8) & 0xFF, ty0 = (span >> 16) & 0xFF, ty1 = span >> 24;
        const uint32_t c = span == 0xFFFFFFFFu ? 0u : (tx1 - tx0 + 1) * (ty1 - ty0 + 1);
        // exclusive scan of c over the 256 threads of this step
        uint32_t inc = c;
        for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(inc, off); if (lane >= (uint32_t)off) inc += t; }
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        uint32_t woff = 0;
        for (uint32_t w = 0; w < wave; ++w) woff += wtot[w];
        uint32_t pos = step_base + woff + inc - c;
        if (c) {
            const uint32_t sid = order[r];
            const uint32_t cls = r >= n_opaque ? 1u : 0u;
            for (uint32_t ty = ty0; ty <= ty1; ++ty)
                for (uint32_t tx = tx0; tx <= tx1; ++tx) {
                    const uint32_t tile = ty * fp.tiles_x + tx;
                    pair_keys[pos] = (tile << 1) | cls;
                    pair_vals[pos] = sid;
                    ++pos;
                }
        }
        __syncthreads();
        if (threadIdx.x == 0) step_base += wtot[0] + wtot[1] + wtot[2] + wtot[3];
        __syncthreads();
    }
}

void launch_bin(hipStream_t s, const FrameParams& fp, const uint32_t* spans, const uint32_t* order, Ctrl* ctrl,
                uint32_t* counts, uint32_t* block_sums, uint32_t max_blocks, uint32_t* pair_keys, uint32_t* pair_vals, uint32_t pair_cap) {
    if (fp.nf == 0) return;
    const uint32_t nblocks = min((fp.nf + BIN_TILE - 1) / BIN_TILE, max_blocks);
    hipLaunchKernelGGL(k_bin_count, dim3(nblocks), dim3(BIN_THREADS), 0, s, fp, spans, order, ctrl, counts, block_sums);
    hipLaunchKernelGGL(k_bin_emit, dim3(nblocks), dim3(BIN_THREADS), 0, s, fp, order, ctrl, counts, block_sums, nblocks, pair_cap, pair_keys, pair_vals);
}

// ---- fast path ----------------------------------------------------------------------------------------------------
// k_bin_emit_faces: one block per 4096 faces, pairs in FACE order from k_setup's packed spans (no record gather, no depth
// sort needed first: k_cover sorts every tile list by depth key in LDS).  The prologue of every block derives its base
// from k_setup's per-256-face pair counts; block 0 also reduces the frame counters into Ctrl.
__global__ __launch_bounds__(BIN_THREADS) void k_bin_emit_faces(FrameParams fp, const uint32_t* __restrict__ spans, const uint32_t* __restrict__ keys,
                                                                 const uint32_t* __restrict__ partials, uint32_t npart, Ctrl* __restrict__ ctrl,
                                                                 uint32_t pair_cap, uint32_t* __restrict__ pair_keys, uint32_t* __restrict__ pair_vals,
                                                                 int with_class) {
    __shared__ uint32_t wtot[BIN_THREADS / 64];
    __shared__ uint32_t red[BIN_THREADS / 64][8];
    __shared__ uint32_t step_base, total_s;
    phase_stamp(ctrl, ST_BIN);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t first_part = blockIdx.x * (BIN_TILE / 256);            // partial records (256 faces each) before this block
    {
        uint32_t acc[7] = { 0, 0, 0, 0, 0, 0, 0 };                         // 0..4 frame counters, 5 all pairs, 6 pairs before this block
        for (uint32_t i = threadIdx.x; i < npart; i += BIN_THREADS) {
            const uint32_t pc = partials[i * 8 + 5];
            acc[5] += pc;
            if (i < first_part) acc[6] += pc;
            if (blockIdx.x == 0) for (int k = 0; k < 5; ++k) acc[k] += partials[i * 8 + k];
        }
        for (int k = 0; k < 7; ++k) for (int off = 32; off > 0; off >>= 1) acc[k] += __shfl_down(acc[k], off);
        if (lane == 0) for (int k = 0; k < 7; ++k) red[wave][k] = acc[k];
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t t[7];
            for (int k = 0; k < 7; ++k) t[k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
            step_base = t[6]; total_s = t[5];
            if (blockIdx.x == 0) {     // frame counters (the reference panics before drawing on a bad index / NaN sort key)
                const uint32_t n_op = t[0] - t[1];
                ctrl->n_visible = t[0]; ctrl->n_transparent = t[1]; ctrl->nan_opaque = t[2]; ctrl->nan_transparent = t[3];
                ctrl->err_index = t[4] ? 1u : 0u; ctrl->n_opaque = n_op;
                if (t[4] || (t[2] && n_op >= 2) || (t[3] && t[1] >= 2)) ctrl->abort = 1;
                if (t[5] > pair_cap) { ctrl->pairs_overflow = t[5]; ctrl->abort = 1; ctrl->n_pairs = 0; }
                else ctrl->n_pairs = t[5];
            }
        }
        __syncthreads();
    }
    if (total_s > pair_cap) return;
    // blocked arrangement: thread t owns the 16 consecutive faces base + 16 t .. +15, so positions are monotone in the face id
    // (all the tile-local sort needs for stability) with one block-wide scan instead of one per step.
    const uint32_t f0 = blockIdx.x * BIN_TILE + threadIdx.x * BIN_ITEMS;
    uint32_t span[BIN_ITEMS];
    uint32_t mine = 0;
    if (f0 + BIN_ITEMS <= fp.nf) {
        const uint4* sp = reinterpret_cast<const uint4*>(spans + f0);
#pragma unroll
        for (int q = 0; q < BIN_ITEMS / 4; ++q) { const uint4 v = sp[q]; span[4 * q] = v.x; span[4 * q + 1] = v.y; span[4 * q + 2] = v.z; span[4 * q + 3] = v.w; }
    } else {
#pragma unroll
        for (int i = 0; i < BIN_ITEMS; ++i) span[i] = f0 + i < fp.nf ? spans[f0 + i] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int i = 0; i < BIN_ITEMS; ++i)
        if (span[i] != 0xFFFFFFFFu) mine += (((span[i] >> 8) & 0xFF) - (span[i] & 0xFF) + 1) * ((span[i] >> 24) - ((span[i] >> 16) & 0xFF) + 1);
    uint32_t inc = mine;
    for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(inc, off); if (lane >= (uint32_t)off) inc += t; }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    uint32_t pos = step_base + inc - mine;
    for (uint32_t w = 0; w < wave; ++w) pos += wtot[w];
    if (mine) {
#pragma unroll 1
        for (int i = 0; i < BIN_ITEMS; ++i) {
            if (span[i] == 0xFFFFFFFFu) continue;
            const uint32_t f = f0 + i;
            const uint32_t tx0 = span[i] & 0xFF, tx1 = (span[i] >> 8) & 0xFF, ty0 = (span[i] >> 16) & 0xFF, ty1 = span[i] >> 24;
            const uint32_t cls = with_class ? keys[f] >> 31 : 0u;                  // transparent pass, render.rs:2522-2523
            for (uint32_t ty = ty0; ty <= ty1; ++ty)
                for (uint32_t tx = tx0; tx <= tx1; ++tx) {
                    const uint32_t tile = ty * fp.tiles_x + tx;
                    pair_keys[pos] = with_class ? ((tile << 1) | cls) : tile;
                    pair_vals[pos] = f;
                    ++pos;
                }
        }
    }
}
void launch_bin_faces(hipStream_t s, const FrameParams& fp, const uint32_t* spans, const uint32_t* keys, const uint32_t* partials,
                      Ctrl* ctrl, uint32_t* pair_keys, uint32_t* pair_vals, uint32_t pair_cap, int with_class) {
    if (fp.nf == 0) return;
    const uint32_t nblocks = (fp.nf + BIN_TILE - 1) / BIN_TILE;
    hipLaunchKernelGGL(k_bin_emit_faces, dim3(nblocks), dim3(BIN_THREADS), 0, s, fp, spans, keys, partials, (fp.nf + 255) / 256, ctrl, pair_cap,
                       pair_keys, pair_vals, with_class);
}

// ranges[k] = first pair index whose key >= k, for k in 0..n_keys (n_keys = 2*ntiles); ranges[n_keys] = n_pairs.
__global__ void k_tile_ranges(const uint32_t* __restrict__ pair_keys, const Ctrl* __restrict__ ctrl, uint32_t* __restrict__ ranges, uint32_t n_keys) {
    const uint32_t n = ctrl->n_pairs;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += stride) {
        // boundary between pair i-1 and pair i: every key in (prev, cur] starts at i
        const uint32_t prev = i == 0 ? 0xFFFFFFFFu : pair_keys[i - 1];          // -1 as "before key 0"
        const uint32_t cur = i == n ? n_keys : pair_keys[i];
        uint32_t k0 = i == 0 ? 0u : prev + 1u;
        for (uint32_t k = k0; k <= cur && k <= n_keys; ++k) ranges[k] = i;
    }
}
void launch_tile_ranges(hipStream_t s, const uint32_t* pair_keys, const Ctrl* ctrl, uint32_t pair_cap, uint32_t* ranges, uint32_t n_keys) {
    uint32_t blocks = (pair_cap + 1 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(k_tile_ranges, dim3(blocks), dim3(256), 0, s, pair_keys, ctrl, ranges, n_keys);
}

}  // namespace b32

// b32_sort.hip — stable LSD radix sort (8-bit digits) over (u32 key, u32 value) pairs, wave64-native.
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
__global__ __launch_bounds__(SORT_THREADS) void k_hist(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ n_dev,
                                                        int shift, int drop_invalid, uint32_t* __restrict__ block_hist, uint32_t max_blocks) {
    __shared__ uint32_t hist[256];
    const uint32_t n = *n_dev;
    const uint32_t base = blockIdx.x * SORT_TILE;
    if (base >= n) {   // still publish zeros so the scan sees a clean column
        block_hist[threadIdx.x * max_blocks + blockIdx.x] = 0;
        return;
    }
    hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t wbase = base + wave * (SORT_ITEMS * 64);
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        uint32_t idx = wbase + i * 64 + lane;
        if (idx < n) {
            uint32_t k = keys[idx];
            if (!(drop_invalid && k == KEY_INVALID)) atomicAdd(&hist[(k >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    block_hist[threadIdx.x * max_blocks + blockIdx.x] = hist[threadIdx.x];
}

// Row d of block_hist[256][max_blocks] (one workgroup per digit): exclusive scan over the blocks, in place, and the
// row total into digit_total[d].  The scan across digits is folded into k_scatter (256 values, one per thread).
__global__ __launch_bounds__(256) void k_scan_rows(uint32_t* __restrict__ block_hist, uint32_t max_blocks, uint32_t nblocks,
                                                   uint32_t* __restrict__ digit_total) {
    __shared__ uint32_t wsum[4];
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

__global__ __launch_bounds__(SORT_THREADS) void k_scatter(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                           uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                           const uint32_t* __restrict__ n_dev, int shift, int drop_invalid,
                                                           const uint32_t* __restrict__ block_hist, uint32_t max_blocks,
                                                           const uint32_t* __restrict__ digit_total) {
    __shared__ uint32_t wcnt[4][256];     // per-wave running digit counts, then exclusive prefix over waves
    __shared__ uint32_t dbase[256];       // exclusive scan of the 256 digit totals
    __shared__ uint32_t dws[4];
    const uint32_t n = *n_dev;
    const uint32_t base = blockIdx.x * SORT_TILE;
    if (base >= n) return;
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    for (int w = 0; w < 4; ++w) wcnt[w][threadIdx.x] = 0;
    {   // digit bases: exclusive scan of digit_total over the 256 threads
        const uint32_t v = digit_total[threadIdx.x];
        uint32_t inc = v;
        for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(inc, off); if (lane >= (uint32_t)off) inc += t; }
        if (lane == 63) dws[wave] = inc;
        __syncthreads();
        uint32_t woff = 0;
        for (uint32_t w = 0; w < wave; ++w) woff += dws[w];
        dbase[threadIdx.x] = woff + inc - v;
    }
    __syncthreads();
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
        const uint32_t d = (k >> shift) & 255u;
        // peers = live lanes of this wave holding the same digit (8 ballots)
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
            if (lane == leader) { old = wcnt[wave][d]; wcnt[wave][d] = old + (uint32_t)__popcll(peers); }
            old = __shfl(old, (int)leader);
            before = old + (uint32_t)__popcll(peers & lt_mask);
        }
        rnk[i] = live ? before : 0xFFFFFFFFu;
    }
    __syncthreads();
    {   // thread d: exclusive prefix of digit d over the 4 waves + global base of (digit, block)
        const uint32_t d = threadIdx.x;
        uint32_t run = dbase[d] + block_hist[d * max_blocks + blockIdx.x];
        for (int w = 0; w < 4; ++w) { uint32_t c = wcnt[w][d]; wcnt[w][d] = run; run += c; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        if (rnk[i] != 0xFFFFFFFFu) {
            const uint32_t d = (key[i] >> shift) & 255u;
            const uint32_t pos = wcnt[wave][d] + rnk[i];
            keys_out[pos] = key[i];
            vals_out[pos] = val[i];
        }
    }
}

void launch_radix_pass(hipStream_t s, const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out,
                       const uint32_t* n_dev, uint32_t n_cap, int shift, const SortScratch& sc) {
    if (n_cap == 0) return;
    const uint32_t nblocks = (n_cap + SORT_TILE - 1) / SORT_TILE;
    const int drop = vals_in == nullptr ? 1 : 0;
    hipLaunchKernelGGL(k_hist, dim3(nblocks), dim3(SORT_THREADS), 0, s, keys_in, n_dev, shift, drop, sc.block_hist, sc.max_blocks);
    hipLaunchKernelGGL(k_scan_rows, dim3(256), dim3(256), 0, s, sc.block_hist, sc.max_blocks, nblocks, sc.digit_total);
    hipLaunchKernelGGL(k_scatter, dim3(nblocks), dim3(SORT_THREADS), 0, s, keys_in, vals_in, keys_out, vals_out, n_dev, shift, drop,
                       sc.block_hist, sc.max_blocks, sc.digit_total);
}

}  // namespace b32

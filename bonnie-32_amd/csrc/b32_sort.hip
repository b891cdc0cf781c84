// b32_sort.hip — stable LSD radix sort (8-bit digits) over (u32 key, u32 value) pairs, wave64-native.
//
// Replaces the reference's `sort_by` merge sort of 208-byte Surface structs (render.rs:2527-2541): the painter's key is
// reduced to a 32-bit radix key by k_setup and only (key, surface id) pairs move.  Stability of every pass is what makes
// equal keys keep face order, exactly like slice::sort_by.
//
// One pass = k_hist (per-block digit histogram, LDS atomics) -> k_scan (exclusive scan of the digit-major table, one
// workgroup) -> k_scatter (per-wave match-any ranking with 8 ballots, no atomics, deterministic).
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

// Exclusive scan of block_hist[256][max_blocks] viewed as one array of length 256*nblocks_used (digit-major), in place.
__global__ __launch_bounds__(1024) void k_scan(uint32_t* __restrict__ block_hist, uint32_t max_blocks, uint32_t nblocks) {
    __shared__ uint32_t part[1024];
    const uint32_t total = 256u * nblocks;
    const uint32_t per = (total + 1023u) / 1024u;
    const uint32_t lo = threadIdx.x * per, hi = min(lo + per, total);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += block_hist[(i / nblocks) * max_blocks + (i % nblocks)];
    part[threadIdx.x] = sum;
    __syncthreads();
    // Hillis-Steele inclusive scan over 1024 partials
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        uint32_t v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (uint32_t i = lo; i < hi; ++i) {
        uint32_t* p = &block_hist[(i / nblocks) * max_blocks + (i % nblocks)];
        uint32_t v = *p; *p = run; run += v;
    }
}

__global__ __launch_bounds__(SORT_THREADS) void k_scatter(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                           uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                           const uint32_t* __restrict__ n_dev, int shift, int drop_invalid,
                                                           const uint32_t* __restrict__ block_hist, uint32_t max_blocks) {
    __shared__ uint32_t wcnt[4][256];     // per-wave running digit counts, then exclusive prefix over waves
    const uint32_t n = *n_dev;
    const uint32_t base = blockIdx.x * SORT_TILE;
    if (base >= n) return;
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    for (int w = 0; w < 4; ++w) wcnt[w][threadIdx.x] = 0;
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
        uint32_t run = block_hist[d * max_blocks + blockIdx.x];
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
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, s, sc.block_hist, sc.max_blocks, nblocks);
    hipLaunchKernelGGL(k_scatter, dim3(nblocks), dim3(SORT_THREADS), 0, s, keys_in, vals_in, keys_out, vals_out, n_dev, shift, drop,
                       sc.block_hist, sc.max_blocks);
}

}  // namespace b32

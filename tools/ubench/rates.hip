// Instruction issue cost on gfx950, measured: one workgroup of W waves per CU slot, each wave runs a loop of 8 x 16 independent
// instances of one instruction and reads s_memtime (100 MHz) / clock64 around it.  Prints shader cycles per wave-instruction
// at 1 wave per SIMD (latency + issue) and at 4 waves per SIMD (issue throughput).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int OP>
__global__ __launch_bounds__(1024) void k_rate(unsigned long long* out, uint32_t iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x * 2654435761u + seed, a1 = a0 ^ 0x1234567u, a2 = a0 + 77u, a3 = a0 * 3u, a4 = a0 + 5u, a5 = a1 * 7u, a6 = a2 ^ 99u, a7 = a3 + 13u;
    uint32_t b = seed | 3u;
    unsigned long long d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    float f0 = (float)a0, f1 = (float)a1, f2 = (float)a2, f3 = (float)a3, f4 = 1.5f, f5 = 2.5f, f6 = 3.5f, f7 = 4.5f;
    typedef float f2v __attribute__((ext_vector_type(2)));
    f2v p0 = { f0, f1 }, p1 = { f2, f3 }, p2 = { f4, f5 }, p3 = { f6, f7 }, pb = { 1.0001f, 0.9999f };
    double e0 = a0, e1 = a1, e2 = a2, e3 = a3, eb = 1.0000001;
    __shared__ unsigned long long lds64[2048];
    __shared__ uint32_t lds32[4096];
    for (uint32_t i = threadIdx.x; i < 2048; i += blockDim.x) { lds64[i] = 0; lds32[i] = 0; lds32[i + 2048] = 0; }
    __syncthreads();
    const unsigned long long t0 = clock64();
    for (uint32_t it = 0; it < iters; ++it) {
        if (OP == 0) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %2, %2, %1\n v_mul_lo_u32 %3, %3, %1\n v_mul_lo_u32 %4, %4, %1" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 1) { REP16(asm volatile("v_mul_hi_u32 %0, %0, %1\n v_mul_hi_u32 %2, %2, %1\n v_mul_hi_u32 %3, %3, %1\n v_mul_hi_u32 %4, %4, %1" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 2) { REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(b) : "vcc");) }
        if (OP == 3) { REP16(asm volatile("v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %2, %2, %1\n v_mul_u32_u24 %3, %3, %1\n v_mul_u32_u24 %4, %4, %1" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 4) { REP16(asm volatile("v_mul_f32 %0, %0, %1\n v_mul_f32 %2, %2, %1\n v_mul_f32 %3, %3, %1\n v_mul_f32 %4, %4, %1" : "+v"(f0), "+v"(f4), "+v"(f1), "+v"(f2), "+v"(f3));) }
        if (OP == 5) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));) }
        if (OP == 6) { REP16(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(eb));) }
        if (OP == 7) { REP16(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_add_u32 %4, %4, %1" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 8) { REP16(asm volatile("v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 9) { REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));) }
        if (OP == 10) { REP16(asm volatile("ds_bpermute_b32 %0, %4, %0\n ds_bpermute_b32 %1, %4, %1\n ds_bpermute_b32 %2, %4, %2\n ds_bpermute_b32 %3, %4, %3\n s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4 & 252u));) }
        if (OP == 11) { REP16(asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 12) {   // 64-bit returning LDS max, 4 in flight, distinct addresses per lane
            const uint32_t ad = ((threadIdx.x * 8u) & 16383u);
            REP16(asm volatile("ds_max_rtn_u64 %0, %4, %0\n ds_max_rtn_u64 %1, %4, %1 offset:8192\n ds_max_rtn_u64 %2, %4, %2\n ds_max_rtn_u64 %3, %4, %3 offset:8192\n s_waitcnt lgkmcnt(0)" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(ad) : "memory");) }
        if (OP == 13) {   // 32-bit returning LDS max
            const uint32_t ad = ((threadIdx.x * 4u) & 8191u);
            REP16(asm volatile("ds_max_rtn_u32 %0, %4, %0\n ds_max_rtn_u32 %1, %4, %1 offset:8192\n ds_max_rtn_u32 %2, %4, %2\n ds_max_rtn_u32 %3, %4, %3 offset:8192\n s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(ad) : "memory");) }
        if (OP == 14) {   // 64-bit non-returning LDS max
            const uint32_t ad = ((threadIdx.x * 8u) & 16383u);
            REP16(asm volatile("ds_max_u64 %4, %0\n ds_max_u64 %4, %1 offset:8192\n ds_max_u64 %4, %2\n ds_max_u64 %4, %3 offset:8192\n s_waitcnt lgkmcnt(0)" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(ad) : "memory");) }
        if (OP == 15) { REP16(asm volatile("v_mad_u32_u24 %0, %0, %1, %0\n v_mad_u32_u24 %2, %2, %1, %2\n v_mad_u32_u24 %3, %3, %1, %3\n v_mad_u32_u24 %4, %4, %1, %4" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 16) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %2, %2, %1, %2\n v_fma_f32 %3, %3, %1, %3\n v_fma_f32 %4, %4, %1, %4" : "+v"(f0), "+v"(f4), "+v"(f1), "+v"(f2), "+v"(f3));) }
        if (OP == 17) { REP16(asm volatile("v_lshlrev_b64 %0, 3, %0\n v_lshlrev_b64 %1, 3, %1\n v_lshlrev_b64 %2, 3, %2\n v_lshlrev_b64 %3, 3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
        if (OP == 18) { REP16(asm volatile("v_min_f32 %0, %0, %1\n v_min_f32 %2, %2, %1\n v_min_f32 %3, %3, %1\n v_min_f32 %4, %4, %1" : "+v"(f0), "+v"(f4), "+v"(f1), "+v"(f2), "+v"(f3));) }
        if (OP == 19) { REP16(asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 9" :: "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s20", "s21", "s22", "s23");) }
        if (OP == 20) { REP16(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %1\n v_add_f32 %3, %3, %1\n v_add_f32 %4, %4, %1" : "+v"(f0), "+v"(f4), "+v"(f1), "+v"(f2), "+v"(f3));) }
        if (OP == 21) { REP16(asm volatile("v_sub_u32 %0, %0, %1\n v_sub_u32 %2, %2, %1\n v_sub_u32 %3, %3, %1\n v_sub_u32 %4, %4, %1" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 22) { REP16(asm volatile("v_and_b32 %0, %0, %1\n v_and_b32 %2, %2, %1\n v_and_b32 %3, %3, %1\n v_and_b32 %4, %4, %1" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 23) { REP16(asm volatile("v_or_b32 %0, %0, %1\n v_or_b32 %2, %2, %1\n v_or_b32 %3, %3, %1\n v_or_b32 %4, %4, %1" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 24) { REP16(asm volatile("v_lshlrev_b32 %0, %1, %0\n v_lshlrev_b32 %2, %1, %2\n v_lshlrev_b32 %3, %1, %3\n v_lshlrev_b32 %4, %1, %4" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 25) { REP16(asm volatile("v_lshrrev_b32 %0, %1, %0\n v_lshrrev_b32 %2, %1, %2\n v_lshrrev_b32 %3, %1, %3\n v_lshrrev_b32 %4, %1, %4" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 26) { REP16(asm volatile("v_max_f32 %0, %0, %1\n v_max_f32 %2, %2, %1\n v_max_f32 %3, %3, %1\n v_max_f32 %4, %4, %1" : "+v"(f0), "+v"(f4), "+v"(f1), "+v"(f2), "+v"(f3));) }
        if (OP == 27) { REP16(asm volatile("v_min_u32 %0, %0, %1\n v_min_u32 %2, %2, %1\n v_min_u32 %3, %3, %1\n v_min_u32 %4, %4, %1" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 28) { REP16(asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %2, %2, %1\n v_xor_b32 %3, %3, %1\n v_xor_b32 %4, %4, %1" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 29) { REP16(asm volatile("v_max_i32 %0, %0, %1\n v_max_i32 %2, %2, %1\n v_max_i32 %3, %3, %1\n v_max_i32 %4, %4, %1" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 30) { REP16(asm volatile("v_trunc_f32 %0, %0\n v_trunc_f32 %1, %1\n v_trunc_f32 %2, %2\n v_trunc_f32 %3, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));) }
        if (OP == 31) { REP16(asm volatile("v_cvt_f32_ubyte0 %0, %0\n v_cvt_f32_ubyte0 %1, %1\n v_cvt_f32_ubyte0 %2, %2\n v_cvt_f32_ubyte0 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 32) { REP16(asm volatile("v_cvt_u32_f32 %0, %0\n v_cvt_u32_f32 %1, %1\n v_cvt_u32_f32 %2, %2\n v_cvt_u32_f32 %3, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));) }
        if (OP == 33) { REP16(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 34) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %1, vcc\n v_cndmask_b32 %3, %3, %1, vcc\n v_cndmask_b32 %4, %4, %1, vcc" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3) :: "vcc");) }
        if (OP == 35) { REP16(asm volatile("v_bfe_u32 %0, %0, %1, %5\n v_bfe_u32 %2, %2, %1, %5\n v_bfe_u32 %3, %3, %1, %5\n v_bfe_u32 %4, %4, %1, %5" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));) }
        if (OP == 36) { REP16(asm volatile("v_perm_b32 %0, %0, %1, %5\n v_perm_b32 %2, %2, %1, %5\n v_perm_b32 %3, %3, %1, %5\n v_perm_b32 %4, %4, %1, %5" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));) }
        if (OP == 37) { REP16(asm volatile("v_lshl_or_b32 %0, %0, %1, %5\n v_lshl_or_b32 %2, %2, %1, %5\n v_lshl_or_b32 %3, %3, %1, %5\n v_lshl_or_b32 %4, %4, %1, %5" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));) }
        if (OP == 38) { REP16(asm volatile("v_add3_u32 %0, %0, %1, %5\n v_add3_u32 %2, %2, %1, %5\n v_add3_u32 %3, %3, %1, %5\n v_add3_u32 %4, %4, %1, %5" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));) }
        if (OP == 39) { REP16(asm volatile("v_med3_f32 %0, %0, %1, %5\n v_med3_f32 %2, %2, %1, %5\n v_med3_f32 %3, %3, %1, %5\n v_med3_f32 %4, %4, %1, %5" : "+v"(f0), "+v"(f4), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(f5));) }
        if (OP == 40) { REP16(asm volatile("v_alignbit_b32 %0, %0, %1, %5\n v_alignbit_b32 %2, %2, %1, %5\n v_alignbit_b32 %3, %3, %1, %5\n v_alignbit_b32 %4, %4, %1, %5" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));) }
        if (OP == 41) { REP16(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));) }
        if (OP == 42) { REP16(asm volatile("v_pk_mul_lo_u16 %0, %0, %1\n v_pk_mul_lo_u16 %2, %2, %1\n v_pk_mul_lo_u16 %3, %3, %1\n v_pk_mul_lo_u16 %4, %4, %1" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 43) { REP16(asm volatile("v_pk_add_u16 %0, %0, %1\n v_pk_add_u16 %2, %2, %1\n v_pk_add_u16 %3, %3, %1\n v_pk_add_u16 %4, %4, %1" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 44) { REP16(asm volatile("v_and_or_b32 %0, %0, %1, %5\n v_and_or_b32 %2, %2, %1, %5\n v_and_or_b32 %3, %3, %1, %5\n v_and_or_b32 %4, %4, %1, %5" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));) }
        if (OP == 45) { REP16(asm volatile("v_mul_i32_i24 %0, %0, %1\n v_mul_i32_i24 %2, %2, %1\n v_mul_i32_i24 %3, %3, %1\n v_mul_i32_i24 %4, %4, %1" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 46) { REP16(asm volatile("v_sub_f32 %0, %0, %1\n v_sub_f32 %2, %2, %1\n v_sub_f32 %3, %3, %1\n v_sub_f32 %4, %4, %1" : "+v"(f0), "+v"(f4), "+v"(f1), "+v"(f2), "+v"(f3));) }
        if (OP == 47) { REP16(asm volatile("v_ashrrev_i32 %0, %1, %0\n v_ashrrev_i32 %2, %1, %2\n v_ashrrev_i32 %3, %1, %3\n v_ashrrev_i32 %4, %1, %4" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 48) { REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %2, %1\n v_cmp_lt_f32 vcc, %3, %1\n v_cmp_lt_f32 vcc, %4, %1" :: "v"(f0), "v"(f4), "v"(f1), "v"(f2), "v"(f3) : "vcc");) }
        if (OP == 49) { REP16(asm volatile("v_mad_u32_u24 %0, %0, %1, %5\n v_mad_u32_u24 %2, %2, %1, %5\n v_mad_u32_u24 %3, %3, %1, %5\n v_mad_u32_u24 %4, %4, %1, %5" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));) }
        if (OP == 50) { REP16(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %1\n v_add_u32 %4, %4, %1" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 51) { REP16(asm volatile("v_mul_f32 %0, %0, %1\n v_mul_f32 %2, %2, %1\n v_mul_f32 %3, %3, %1\n v_mul_f32 %4, %4, %1" : "+v"(f0), "+v"(f4), "+v"(f1), "+v"(f2), "+v"(f3));) }
        if (OP == 52) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %5\n v_fma_f32 %2, %2, %1, %5\n v_fma_f32 %3, %3, %1, %5\n v_fma_f32 %4, %4, %1, %5" : "+v"(f0), "+v"(f4), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(f5));) }
        if (OP == 53) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));) }
        if (OP == 54) { REP16(asm volatile("v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %1, vcc, %4, %5, %1\n v_mad_i64_i32 %2, vcc, %4, %5, %2\n v_mad_i64_i32 %3, vcc, %4, %5, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(b) : "vcc");) }
        if (OP == 55) { REP16(asm volatile("v_lshl_add_u32 %0, %0, %1, %5\n v_lshl_add_u32 %2, %2, %1, %5\n v_lshl_add_u32 %3, %3, %1, %5\n v_lshl_add_u32 %4, %4, %1, %5" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));) }
        if (OP == 56) { REP16(asm volatile("v_bfi_b32 %0, %0, %1, %5\n v_bfi_b32 %2, %2, %1, %5\n v_bfi_b32 %3, %3, %1, %5\n v_bfi_b32 %4, %4, %1, %5" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));) }
        if (OP == 57) { REP16(asm volatile("v_cvt_f32_i32 %0, %0\n v_cvt_f32_i32 %1, %1\n v_cvt_f32_i32 %2, %2\n v_cvt_f32_i32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    }
    const unsigned long long t1 = clock64();
    uint32_t sink = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)d0 ^ (uint32_t)d1 ^ (uint32_t)d2 ^ (uint32_t)d3 ^ __float_as_uint(f0 + f1 + f2 + f3 + f4) ^
                    __float_as_uint(p0.x + p1.x + p2.y + p3.y) ^ (uint32_t)(long long)(e0 + e1 + e2 + e3) ^ (uint32_t)lds64[threadIdx.x & 2047] ^ lds32[threadIdx.x];
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = (t1 - t0) | ((unsigned long long)(sink == 0x12345u) << 63);
}

template <int OP> double run(const char* name, int waves, uint32_t iters) {
    unsigned long long* d; hipMalloc(&d, 256 * 16 * 8);
    hipMemset(d, 0, 256 * 16 * 8);
    hipLaunchKernelGGL(k_rate<OP>, dim3(256), dim3(waves * 64), 0, 0, d, iters, 12345u);
    hipLaunchKernelGGL(k_rate<OP>, dim3(256), dim3(waves * 64), 0, 0, d, iters, 12345u);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 16);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double mx = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) { double v = (double)(h[b * 16 + w] & ~(1ull << 63)); if (v > mx) mx = v; }
    hipFree(d);
    return mx / ((double)iters * 64.0);     // clock64 ticks per wave-instruction (of one wave)
}
#define ROW(OP, NAME) { double a = run<OP>(NAME, 4, 200), b = run<OP>(NAME, 16, 200); printf("%-18s %8.2f %8.2f   per-SIMD issue cycles/instr at 4 waves/SIMD: %6.2f\n", NAME, a, b, b / 4.0); }
int main() {
    printf("clock64 ticks per wave-instruction (one wave's view): 1 wave/SIMD, 4 waves/SIMD\n");
    ROW(7, "v_add_u32") ROW(4, "v_mul_f32") ROW(16, "v_fma_f32") ROW(18, "v_min_f32") ROW(5, "v_pk_mul_f32") ROW(6, "v_mul_f64") ROW(0, "v_mul_lo_u32") ROW(1, "v_mul_hi_u32")
    ROW(2, "v_mad_u64_u32") ROW(3, "v_mul_u32_u24") ROW(15, "v_mad_u32_u24") ROW(8, "v_cvt_f32_u32") ROW(9, "v_rcp_f32") ROW(17, "v_lshlrev_b64") ROW(11, "v_mov_dpp")
    ROW(19, "v_readlane") ROW(10, "ds_bpermute x4+wait") ROW(12, "ds_max_rtn_u64 x4+w") ROW(13, "ds_max_rtn_u32 x4+w") ROW(14, "ds_max_u64 x4+w")
    ROW(20, "v_add_f32") ROW(21, "v_sub_u32") ROW(22, "v_and_b32") ROW(23, "v_or_b32") ROW(24, "v_lshlrev_b32") ROW(25, "v_lshrrev_b32") ROW(26, "v_max_f32") ROW(27, "v_min_u32") ROW(28, "v_xor_b32") ROW(29, "v_max_i32") ROW(30, "v_trunc_f32") ROW(31, "v_cvt_f32_ubyte0") ROW(32, "v_cvt_u32_f32") ROW(33, "v_mov_b32") ROW(34, "v_cndmask_b32") ROW(35, "v_bfe_u32") ROW(36, "v_perm_b32") ROW(37, "v_lshl_or_b32") ROW(38, "v_add3_u32") ROW(39, "v_med3_f32") ROW(40, "v_alignbit_b32") ROW(41, "v_pk_add_f32") ROW(42, "v_pk_mul_lo_u16") ROW(43, "v_pk_add_u16") ROW(44, "v_and_or_b32") ROW(45, "v_mul_i32_i24") ROW(46, "v_sub_f32") ROW(47, "v_ashrrev_i32") ROW(48, "v_cmp_lt_f32") ROW(49, "v_mad_u32_u24") ROW(50, "v_add_u32") ROW(51, "v_mul_f32") ROW(52, "v_fma_f32") ROW(53, "v_pk_fma_f32") ROW(54, "v_mad_i64_i32") ROW(55, "v_lshl_add_u32") ROW(56, "v_bfi_b32") ROW(57, "v_cvt_f32_i32")
    return 0;
}

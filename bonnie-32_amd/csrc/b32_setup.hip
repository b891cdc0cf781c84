// b32_setup.hip — per-face transform + snap + cull + triangle setup (one lane per face), plus small utility kernels.
//
// Reference stages folded into k_setup (all per face, so no per-vertex intermediate ever touches HBM):
//   TRANSFORM  render.rs:2313-2362  -> fixed::project_fixed (fixed.rs:424-441) + float cam_pos (math.rs:103-109)
//   CULL/SETUP render.rs:2364-2516  -> near reject, 2D backface, has_transparency, fog, Surface build
//   per-triangle prologue of rasterize_triangle_15 (render.rs:1450-1518): bbox, area, edge coefficients, needs_dither,
//   flat / Gouraud vertex shades (render.rs:1013-1071)
//   painter's key (render.rs:2527-2541): (v1.z + v2.z + v3.z) / 3.0
#include "b32_device.h"

namespace b32 {

// ---------------------------------------------------------------- fixed.rs on the device
struct UnrTable { uint8_t v[K::UNR_ENTRIES]; };
static constexpr UnrTable make_unr() {          // UNR_TABLE, fixed.rs:20-31
    UnrTable t{};
    for (uint32_t i = 0; i < K::UNR_ENTRIES; ++i) {
        uint32_t q = K::UNR_NUMERATOR / (i + K::UNR_INDEX_OFFSET);
        int32_t val = (int32_t)((q + K::UNR_ROUND_ADD) / K::UNR_ROUND_DIV) - K::UNR_SUBTRACT;
        t.v[i] = val > 0 ? (uint8_t)val : 0;
    }
    return t;
}
__constant__ UnrTable g_unr = make_unr();

__device__ __forceinline__ int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
__device__ __forceinline__ int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
__device__ __forceinline__ int32_t fx_from_f32(float f) { return f2i32_sat(f * K::ONE_F); }           // fixed.rs:125-127
__device__ __forceinline__ int32_t fx_mul(int32_t a, int32_t b) { return (int32_t)(((int64_t)a * (int64_t)b) >> K::FRAC_BITS); }  // :161-165 (v_mad_i64_i32 + v_alignbit_b32)
// Fixed32::div_unr, fixed.rs:178-230, split at the point where only the divisor has been used: project_to_screen divides x and
// y by the same denominator (fixed.rs:411-412), so the table lookup and both Newton steps are done once per vertex.
struct UnrRecip { uint64_t nr2; uint32_t shift; bool neg, zero; };
__device__ __forceinline__ UnrRecip unr_recip(int32_t divisor, const uint8_t* __restrict__ table) {
    UnrRecip r;
    r.zero = divisor == 0; r.neg = divisor < 0;
    const uint32_t den = divisor < 0 ? (0u - (uint32_t)divisor) : (uint32_t)divisor;
    const uint32_t z = (uint32_t)__builtin_clz(den | (r.zero ? 1u : 0u));
    const uint32_t d16 = (den << z) >> K::DIV_D16_SHIFT;              // den << z has its top bit at bit 31 (0 for a zero divisor): 32 bits hold it
    uint32_t ti = (d16 - (uint32_t)K::DIV_INDEX_BIAS) >> K::DIV_INDEX_SHIFT;   // (wraps to a huge value for d16 = 0, like the u64 wrapping_sub)
    if (ti > K::DIV_INDEX_MAX) ti = K::DIV_INDEX_MAX;
    // d16 < 2^16, u <= 0xFF + 0x101 = 2^9: both Newton products are below 2^26 -- 32-bit arithmetic gives the u64 results of the
    // reference (its wrapping_sub / wrapping_mul never wrap for these ranges: 0x2000080 - d16 * u >= 0x2000080 - 0xFFFF * 0x200 > 0)
    const uint32_t u = (uint32_t)table[ti] + (uint32_t)K::DIV_U_ADD;
    const uint32_t nr1 = ((uint32_t)K::DIV_NR1_CONST - d16 * u) >> K::DIV_NR1_SHIFT;
    r.nr2 = (uint64_t)(((uint32_t)K::DIV_NR2_CONST + nr1 * u) >> K::DIV_NR2_SHIFT);
    r.shift = K::DIV_SHIFT_BASE - z;                // z in 0..31 -> shift in 5..36
    return r;
}
__device__ __forceinline__ int32_t unr_apply(int32_t self, const UnrRecip& r) {
    if (r.zero) return 0;
    const bool neg = (self < 0) != r.neg;
    const uint64_t num = (uint64_t)(self < 0 ? (0u - (uint32_t)self) : (uint32_t)self);
    const uint64_t raw = num * r.nr2;
    const uint64_t mag = (raw + (1ull << (r.shift - 1))) >> r.shift;
    // mag.min(i32::MAX) in 32 bits (mag >= 2^31 clamps, mag == 2^31 - 1 is the clamp value itself); the empty asm keeps the compiler from
    // carrying the result as a sign-extended 64-bit value into the next mul_fixed (a 64 x 64 multiply of five instructions instead of
    // v_mad_i64_i32, and 64-bit selects / negation here)
    const uint32_t m32 = (uint32_t)(mag >> 31) ? 0x7FFFFFFFu : (uint32_t)mag;
    int32_t r32 = neg ? -(int32_t)m32 : (int32_t)m32;
    asm("" : "+v"(r32));
    return r32;
}
__device__ __forceinline__ int32_t fx_div_unr(int32_t self, int32_t divisor) { return unr_apply(self, unr_recip(divisor, g_unr.v)); }

__device__ __forceinline__ CamFx make_camfx(const B32Camera& c, uint32_t width, uint32_t height) { return make_camfx_any(c, width, height); }
// project_fixed, fixed.rs:424-441 (screen integers only; the fixed depth is discarded by render.rs:2331)
__device__ __forceinline__ void project_fixed_dev(float x, float y, float z, const CamFx& k, int32_t& sx, int32_t& sy,
                                                  const uint8_t* __restrict__ unr_table = g_unr.v) {
    int32_t rx = wsub(fx_from_f32(x), k.px), ry = wsub(fx_from_f32(y), k.py), rz = wsub(fx_from_f32(z), k.pz);
    int32_t cx = wadd(wadd(fx_mul(rx, k.bx[0]), fx_mul(ry, k.bx[1])), fx_mul(rz, k.bx[2]));
    int32_t cy = wadd(wadd(fx_mul(rx, k.by[0]), fx_mul(ry, k.by[1])), fx_mul(rz, k.by[2]));
    int32_t cz = wadd(wadd(fx_mul(rx, k.bz[0]), fx_mul(ry, k.bz[1])), fx_mul(rz, k.bz[2]));
    const int32_t distance = fx_from_f32_any(K::PF_DISTANCE), scale = fx_from_f32_any(K::PF_SCALE);   // 20480, 16384 (folded at compile time)
    int32_t denom = wadd(cz, distance);
    int32_t adenom = denom < 0 ? (int32_t)(0u - (uint32_t)denom) : denom;   // i32::abs wraps at MIN in release
    if (adenom < K::PF_DENOM_GUARD) { sx = k.half_w >> K::FRAC_BITS; sy = k.half_h >> K::FRAC_BITS; return; }
    const UnrRecip rcp = unr_recip(denom, unr_table);
    int32_t proj_x = unr_apply(fx_mul(cx, scale), rcp);
    int32_t proj_y = unr_apply(fx_mul(cy, scale), rcp);
    sx = wadd(fx_mul(proj_x, k.vs), k.half_w) >> K::FRAC_BITS;
    sy = wadd(fx_mul(proj_y, k.vs), k.half_h) >> K::FRAC_BITS;
}

// ---------------------------------------------------------------- math.rs on the device
struct V3 { float x, y, z; };
__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }   // (x*ox + y*oy) + z*oz
__device__ __forceinline__ V3 sub3(V3 a, V3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
__device__ __forceinline__ V3 add3(V3 a, V3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
__device__ __forceinline__ V3 scale3(V3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
__device__ __forceinline__ V3 normalize3(V3 a) {                                                   // math.rs:39-49
    float l = __builtin_sqrtf(dot3(a, a));
    if (l == 0.0f) return { 0.0f, 0.0f, 0.0f };
    return { a.x / l, a.y / l, a.z / l };
}
__device__ __forceinline__ V3 ld3(const float* p) { return { p[0], p[1], p[2] }; }

// f32::acos of the spot-light cone test (render.rs:1047).  Rust's f32::acos is the target's libm acosf: on wasm32 (the console's
// shipping target) the `libm` crate, a port of musl's src/math/acosf.c (FreeBSD msun e_acosf.c); on Linux glibc's, which differs
// by up to 1 ulp -- the reference itself is not bit-portable here.  This is the published musl algorithm in plain f32 arithmetic
// (< 1 ulp; pio2_hi = 0x3fc90fda, pio2_lo = 0x33a22168).
__device__ __forceinline__ float acosf_R(float z) {
    const float pS0 = 1.6666586697e-01f, pS1 = -4.2743422091e-02f, pS2 = -8.6563630030e-03f, qS1 = -7.0662963390e-01f;
    const float p = z * (pS0 + z * (pS1 + z * pS2));
    const float q = 1.0f + z * qS1;
    return p / q;
}
__device__ float acosf_musl(float x) {
    const float pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f;
    const uint32_t hx = __float_as_uint(x), ix = hx & 0x7fffffffu;
    if (ix >= 0x3f800000u) {
        if (ix == 0x3f800000u) return (hx >> 31) ? 2.0f * pio2_hi : 0.0f;
        return __uint_as_float(0x7fc00000u);          // 0/(x-x): only its NaN-ness is observable (comparison + min)
    }
    if (ix < 0x3f000000u) {
        if (ix <= 0x32800000u) return pio2_hi;
        return pio2_hi - (x - (pio2_lo - x * acosf_R(x * x)));
    }
    if (hx >> 31) {
        const float z = (1.0f + x) * 0.5f, s = __builtin_sqrtf(z);
        const float w = acosf_R(z) * s - pio2_lo;
        return 2.0f * (pio2_hi - (s + w));
    }
    const float z = (1.0f - x) * 0.5f, s = __builtin_sqrtf(z);
    const float df = __uint_as_float(__float_as_uint(s) & 0xfffff000u);
    const float c = (z - df * df) / (s + df);
    const float w = acosf_R(z) * s + c;
    return 2.0f * (df + w);
}

// shade_multi_light_color, render.rs:1013-1071
__device__ void shade_multi(V3 normal, V3 world_pos, const B32Light* lights, uint32_t n_lights, float ambient, float out[3]) {
    float tr = ambient, tg = ambient, tb = ambient;
    for (uint32_t i = 0; i < n_lights; ++i) {
        const B32Light& l = lights[i];
        if (!l.enabled) continue;
        float contribution;
        if (l.type == B32_LIGHT_DIRECTIONAL) {
            V3 neg_dir = scale3(ld3(l.direction), -1.0f);
            contribution = rmax(dot3(normal, neg_dir), 0.0f) * l.intensity;
        } else if (l.type == B32_LIGHT_POINT) {
            V3 to_light = sub3(ld3(l.position), world_pos);
            float dist = __builtin_sqrtf(dot3(to_light, to_light));
            if (dist > l.radius || dist < K::LIGHT_MIN_DIST) contribution = 0.0f;
            else {
                float attenuation = 1.0f - (dist / l.radius);
                float n_dot_l = rmax(dot3(normal, normalize3(to_light)), 0.0f);
                contribution = n_dot_l * l.intensity * attenuation * attenuation;
            }
        } else {                                       // Spot, render.rs:1038-1058
            V3 to_light = sub3(ld3(l.position), world_pos);
            float dist = __builtin_sqrtf(dot3(to_light, to_light));
            if (dist > l.radius || dist < K::LIGHT_MIN_DIST) contribution = 0.0f;
            else {
                const V3 to_surface = normalize3(to_light);
                const float spot_angle = acosf_musl(dot3(scale3(to_surface, -1.0f), ld3(l.direction)));
                if (spot_angle > l.angle) contribution = 0.0f;
                else {                                 // (a NaN angle lands here, as in the reference)
                    float attenuation = 1.0f - (dist / l.radius);
                    float edge_falloff = 1.0f - (spot_angle / l.angle);
                    float n_dot_l = rmax(dot3(normal, to_surface), 0.0f);
                    contribution = n_dot_l * l.intensity * attenuation * attenuation * edge_falloff;
                }
            }
        }
        float lr = (float)l.r / K::LIGHT_COLOR_DIV, lg = (float)l.g / K::LIGHT_COLOR_DIV, lb = (float)l.b / K::LIGHT_COLOR_DIV;
        tr += contribution * lr; tg += contribution * lg; tb += contribution * lb;
    }
    out[0] = rmin(tr, K::LIGHT_TOTAL_MAX); out[1] = rmin(tg, K::LIGHT_TOTAL_MAX); out[2] = rmin(tb, K::LIGHT_TOTAL_MAX);
}

// fog, render.rs:2266-2293. Colours are r | g<<8 | b<<16 | blend<<24.
__device__ __forceinline__ float fog_factor(float z, float start, float falloff) {
    if (z <= start) return 0.0f;
    if (falloff <= 0.0f) return 1.0f;
    return rmin((z - start) / falloff, 1.0f);
}
__device__ __forceinline__ uint32_t fog_color(uint32_t c, uint32_t fogc, float f) {
    if (f <= 0.0f) return c;
    if (f >= 1.0f) return fogc;
    float inv = 1.0f - f;
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
        o |= f2u8_sat((float)((c >> (8 * i)) & 255) * inv + (float)((fogc >> (8 * i)) & 255) * f) << (8 * i);
    return o;   // Color::new -> blend Opaque (0)
}

// ---------------------------------------------------------------- wave-aggregated list append (direct binning)
// Neighbours in memory are neighbours on screen in a real mesh, so most lanes of a wave want a slot in the SAME tile list: 64 returning
// atomics on one address serialise (a spatially ordered copy of the C3 scene: k_setup 59 -> 138 us).  Lanes asking for the same counter
// are grouped (the first lane of the remaining set names a counter, a ballot finds its peers: up to 8 groups, given up after two
// singletons -- a spatially random mesh has ~40 distinct counters per wave and gains nothing); one lane per group adds the group's size,
// the others take base + rank.  The atomic is ISSUED here and its result used later (agg_position), behind the record build.
struct AggSlot { uint32_t leader, rank, ret; };
__device__ __forceinline__ void agg_issue(uint32_t* fill, uint32_t slot, bool active, uint32_t lane, AggSlot& g) {
    g.leader = lane; g.rank = 0; g.ret = 0;
    uint32_t cnt = 1;
    unsigned long long rem = __ballot(active);
    const unsigned long long below = (1ull << lane) - 1ull;
    int misses = 0;
    for (int round = 0; round < 8 && rem; ++round) {
        const int l = __builtin_ctzll(rem);
        const uint32_t s = (uint32_t)__builtin_amdgcn_readlane((int)slot, l);
        const unsigned long long grp = __ballot(active && slot == s);      // (lanes of one counter leave `rem` together: grp is a subset of it)
        rem &= ~grp;
        const uint32_t n = (uint32_t)__builtin_popcountll(grp);
        if (n < 2) { if (++misses >= 2) break; continue; }
        if (active && slot == s) { g.leader = (uint32_t)l; g.rank = (uint32_t)__builtin_popcountll(grp & below); if (lane == (uint32_t)l) cnt = n; }
    }
#if defined(B32_EXP_BIN_WG_ATOMICS)          // experiment (frames are WRONG): the reservation as an L2-local atomic -- what per-XCD counters would cost
    if (active && g.leader == lane) g.ret = __hip_atomic_fetch_add(fill + slot, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#elif defined(B32_EXP_BIN_NO_ATOMICS)        // experiment (frames are WRONG): no reservation at all
    if (active && g.leader == lane) g.ret = (slot * 2654435761u) >> 26;
#else
    if (active && g.leader == lane) g.ret = atomicAdd(fill + slot, cnt);
#endif
}
// (every lane that was `active` in agg_issue must call this together)
__device__ __forceinline__ uint32_t agg_position(const AggSlot& g) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(g.leader << 2), (int)g.ret) + g.rank;
}

// ---------------------------------------------------------------- k_setup
// SETUP_FPT = faces per thread (1 everywhere today: registers, i.e. waves per SIMD, are worth more than loads issued ahead)
// PLAIN = the frame uses none of the optional stages (fixed-point snap, perspective, no fog, no lighting, no wireframe lists, no x-ray,
// RGB555): those branches are compiled out -- fewer live scalars, less code in the instruction cache, no exec-mask juggling around them
// (the plain form is compiled for 8 waves per SIMD: left to itself the register allocator lands on 57 ... 65 VGPRs depending on unrelated
// code in this file, i.e. on 7 or 8 waves and on schedules between 46 and 63 us at 1 M faces -- measured; the general form keeps its 5)
// PLAIN == 2 (round 5): the plain form with a shading pass -- RasterSettings::game() (types.rs:1455-1460): Gouraud or flat shades from the
// frame's lights, everything else as in PLAIN == 1.  With the lighting moved behind the record stores (the record's values are dead by
// then) it needs 70 VGPRs instead of 95 (80 with the normals requested early: compiled for 6 waves per SIMD); 3 us of 81 at 1 M faces -- it is not
// occupancy-bound (profiles/r05_lit_setup_ab.txt).  (Compiled for 8 waves -- 64 registers, what would let one of its waves sit beside four
// waves of the capped z-buffer fill -- it spills 23-31 VGPRs and takes 105 us instead of 78: not done.)
#ifndef B32_LIT_SETUP_WAVES
#define B32_LIT_SETUP_WAVES 6
#endif
template <int SETUP_FPT, int PLAIN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PLAIN == 1 ? 8 : (PLAIN == 2 ? B32_LIT_SETUP_WAVES : 4), PLAIN == 1 ? 8 : (PLAIN == 2 ? B32_LIT_SETUP_WAVES : 5)))) void k_setup(FrameParams fp_in, const B32Vertex* __restrict__ verts, const B32Face* __restrict__ faces,
                                               const TexDesc* __restrict__ tex, const B32Light* __restrict__ lights_mem, LightSet lset, MeshTable mtab,
                                               RecArrays recs, DirectBin db, float* __restrict__ shades, uint32_t* __restrict__ keys,
                                               uint32_t* __restrict__ spans, uint32_t* __restrict__ partials, Ctrl* __restrict__ ctrl,
                                               WireTri* __restrict__ wire, const float* __restrict__ pos12, const float* __restrict__ attr12,
                                               uint32_t* __restrict__ face_of) {
    FrameParams fp_plain = fp_in;                 // (dead code unless PLAIN)
    if (PLAIN) { fp_plain.ortho = 0; fp_plain.fixed_point = 1; fp_plain.has_fog = 0; fp_plain.wire_collect = 0; fp_plain.xray = 0; fp_plain.batched = 0; }   // (fmt8 stays a run-time flag: one scalar test)
    if (PLAIN == 1) { fp_plain.shading = B32_SHADE_NONE; fp_plain.n_lights = 0; }
    const FrameParams& fp = PLAIN ? fp_plain : fp_in;
    __shared__ uint32_t wpart[4][6];
    __shared__ uint8_t unr_lds[K::UNR_ENTRIES + 3];          // UNR_TABLE in LDS: its lookup sits in every vertex's dependent chain
    for (uint32_t i = threadIdx.x; i < K::UNR_ENTRIES; i += 256) unr_lds[i] = g_unr.v[i];
    __syncthreads();
    const B32Light* lights = fp.lights_inline ? lset.l : lights_mem;
    if (blockIdx.x == 0) {
        // frame-start reset (no memset launch).  If the previous frame was dropped (pair overflow / long list: nothing drawn) and this
        // is a NEW frame rather than the host's redraw of it, that frame is lost for good: count it, b32_frame_finish reports it
        const bool lost = !fp.redraw && (ctrl->pairs_overflow || ctrl->need_global_sort);
        __syncthreads();
        if (threadIdx.x < sizeof(Ctrl) / 4 && threadIdx.x != offsetof(Ctrl, sticky) / 4) reinterpret_cast<uint32_t*>(ctrl)[threadIdx.x] = 0;
        if (threadIdx.x == 0 && lost) ctrl->sticky += 0x100u;
        if (threadIdx.x == 0) { Stamps* st = reinterpret_cast<Stamps*>(ctrl + 1); for (int k = 1; k < 8; ++k) st->t[k] = 0; st->t[ST_SETUP] = wall_clock64(); }
    }
    // Every thread owns SETUP_FPT faces, 256 apart (a workgroup covers SETUP_FPT groups of 256 consecutive faces).  All of their
    // inputs -- the face words, then the three vertices each -- are requested before the first face is processed, so the second
    // face's two dependent memory latencies pass behind the ~900 VALU instructions of the first.
    struct FaceIn { uint32_t w[5]; float v[3][5]; uint32_t col[3]; bool live, bad; };
    FaceIn fin[SETUP_FPT];
#pragma unroll
    for (int g = 0; g < SETUP_FPT; ++g) {
        const uint32_t f = (blockIdx.x * SETUP_FPT + g) * 256u + threadIdx.x;
        fin[g].live = f < fp.nf; fin[g].bad = false;
        if (fin[g].live) {
            const uint32_t* fw = reinterpret_cast<const uint32_t*>(faces) + (size_t)f * 5;
#pragma unroll
            for (int k = 0; k < 5; ++k) fin[g].w[k] = fw[k];
        }
    }
#pragma unroll
    for (int g = 0; g < SETUP_FPT; ++g) {
        if (fin[g].live) {
            fin[g].bad = fin[g].w[0] >= fp.nv || fin[g].w[1] >= fp.nv || fin[g].w[2] >= fp.nv;       // index panic, render.rs:2375-2377
            if (!fin[g].bad) {
                if (pos12) {          // packed streams of a resident mesh: positions only (12 B instead of a 36-B vertex); the rest once the face is known to be drawn
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const float* pp = pos12 + (size_t)fin[g].w[j] * 3;
                        fin[g].v[j][0] = pp[0]; fin[g].v[j][1] = pp[1]; fin[g].v[j][2] = pp[2];
                        fin[g].v[j][3] = fin[g].v[j][4] = 0.0f; fin[g].col[j] = 0;
                    }
                } else                // (one branch around all three vertices: every load of a path is issued before the first is used)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float* vp = reinterpret_cast<const float*>(verts) + (size_t)fin[g].w[j] * 9;
#pragma unroll
                    for (int k = 0; k < 5; ++k) fin[g].v[j][k] = vp[k];
                    fin[g].col[j] = reinterpret_cast<const uint32_t*>(vp)[8];
                }
            }
        }
    }
#pragma unroll
    for (int g = 0; g < SETUP_FPT; ++g) {
    const uint32_t f = (blockIdx.x * SETUP_FPT + g) * 256u + threadIdx.x;
    const FaceIn& in = fin[g];
    bool visible = false, transparent = false, nan_key = false, bad_index = false;
    uint32_t key = KEY_INVALID, span = 0xFFFFFFFFu, n_tiles = 0;
    AggSlot db_first = { 0, 0, 0 }; bool db_cls = false;
    if (in.live) {
        uint32_t vi[3] = { in.w[0], in.w[1], in.w[2] };
        const uint32_t tid = in.w[3], fb4 = in.w[4];
        const uint32_t black_tr = fb4 & 0xFF, face_blend = (fb4 >> 8) & 0xFF, editor_alpha = (fb4 >> 16) & 0xFF;
        // per-mesh parameters of a batched frame (mesh number in the face's spare byte), else the frame's
        float m_ambient = fp.ambient; bool m_cull = fp.backface_cull != 0, m_has_fog = fp.has_fog != 0; B32Fog m_fog = fp.fog;
        if (fp.batched) {
            const MeshRow& mr = mtab.m[(fb4 >> 24) & (BATCH_MESHES - 1u)];
            m_ambient = mr.ambient; m_cull = (mr.flags & 1u) != 0; m_has_fog = (mr.flags & 2u) != 0; m_fog = mr.fog;
        }
        if (in.bad) {
            bad_index = true;
        } else {
            const CamFx& k = fp.camfx;              // loop-invariant fixed-point conversions, done once on the host
            const V3 cpos = ld3(fp.cam.position), bx = ld3(fp.cam.basis_x), by = ld3(fp.cam.basis_y), bz = ld3(fp.cam.basis_z);
            V3 scr[3]; float camz[3]; V3 wpos[3]; float uvx[3], uvy[3]; uint32_t col[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                V3 pos = { in.v[j][0], in.v[j][1], in.v[j][2] };
                uvx[j] = in.v[j][3]; uvy[j] = in.v[j][4];
                col[j] = in.col[j];
                wpos[j] = pos;
                V3 rel = sub3(pos, cpos);
                V3 cp = { dot3(rel, bx), dot3(rel, by), dot3(rel, bz) };       // perspective_transform, math.rs:103-109
                camz[j] = cp.z;
                if (fp.ortho) {                                                  // render.rs:2323-2328, project_ortho math.rs:140-148
                    scr[j] = { (cp.x - fp.ortho_cx) * fp.ortho_zoom + ((float)fp.width / 2.0f),
                               -(cp.y - fp.ortho_cy) * fp.ortho_zoom + ((float)fp.height / 2.0f), cp.z };
                } else if (fp.fixed_point) {                                     // render.rs:2329-2345
                    int32_t sx, sy;
                    project_fixed_dev(pos.x, pos.y, pos.z, k, sx, sy, unr_lds);
                    scr[j] = { (float)sx, (float)sy, cp.z + K::MESH_DISTANCE };
                } else {                                                         // project, math.rs:117-136
                    uint32_t mn = fp.width < fp.height ? fp.width : fp.height;
                    const float ud = K::P_DISTANCE, us = ud - K::P_US_SUB;                        // math.rs:121-122 (4.0, folded)
                    float vs = ((float)mn / K::P_VIEWPORT_DIV) * K::P_VIEWPORT_FRAC;
                    float denom = cp.z + ud;
                    if (__builtin_fabsf(denom) < K::P_DENOM_GUARD) scr[j] = { (float)fp.width / 2.0f, (float)fp.height / 2.0f, cp.z };
                    else scr[j] = { (cp.x * us) / denom * vs + ((float)fp.width / 2.0f),
                                    (cp.y * us) / denom * vs + ((float)fp.height / 2.0f), denom };
                }
            }
            bool keep = fp.ortho || !(camz[0] <= K::NEAR_PLANE || camz[1] <= K::NEAR_PLANE || camz[2] <= K::NEAR_PLANE);   // render.rs:2381-2385
            float signed_area = (scr[1].x - scr[0].x) * (scr[2].y - scr[0].y) - (scr[2].x - scr[0].x) * (scr[1].y - scr[0].y);
            bool backface = signed_area <= 0.0f;                                     // render.rs:2393-2394
            const bool have_tex = tid != B32_NO_TEXTURE && tid < fp.nt;             // textures.get(id)
            uint32_t tex_blend = B32_BLEND_OPAQUE;
            if (fp.tex_blend_any && keep && have_tex) tex_blend = tex[tid].blend_mode;     // (all Opaque: no descriptor gather in the chain)
            // fog, render.rs:2419-2442: the distance cull here, the colours further down (only surfaces whose record is built need them)
            if (fp.has_fog && m_has_fog && keep && camz[0] > m_fog.cull_distance && camz[1] > m_fog.cull_distance && camz[2] > m_fog.cull_distance) keep = false;
            if (fp.wire_collect) {                       // wireframe lists take the face before the solid decision (render.rs:2445-2449, 2509-2511)
                WireTri wt;
#pragma unroll
                for (int j = 0; j < 3; ++j) { wt.x[j] = f2i32_sat(scr[j].x); wt.y[j] = f2i32_sat(scr[j].y); wt.z[j] = scr[j].z; }
                wt.kind = !keep ? 0u : (backface ? (fp.xray ? 0u : 1u) : 2u);
                wire[f] = wt;
            }
            if (backface && m_cull && !fp.xray) keep = false;              // render.rs:2451-2453
            if (keep) {
                visible = true;
                // Record slot: the survivors of a wave's 64 faces are packed to the front of the wave's 64 slots (slot = first face id of
                // the wave + rank among its survivors), so the records of a half-culled mesh fill whole cache lines instead of every other
                // 32 / 64 bytes.  Slots are monotone in the face id, so everything downstream (tile lists, the priority's low word, the
                // stable sorts) keeps face order; only b32_last_draw_order maps back (face_of).  (The ballot runs inside the branch: it
                // sees exactly the lanes that got here.)
                const uint32_t rslot = (f & ~63u) + (uint32_t)__builtin_popcountll(__ballot(true) & ((1ull << (threadIdx.x & 63u)) - 1ull));
                transparent = (have_tex && tex_blend != B32_BLEND_OPAQUE) || face_blend != B32_BLEND_OPAQUE || editor_alpha < 255;  // :2403-2415
                if (fp.fmt8) transparent = false;     // render_mesh computes has_transparency but never partitions (render.rs:2175-2184)
                // Surface build: rendered backfaces swap v2/v3 and every per-vertex attribute (render.rs:2453-2479)
                const int i1 = 0, i2 = backface ? 2 : 1, i3 = backface ? 1 : 2;
                const V3 v1 = scr[i1], v2 = scr[i2], v3 = scr[i3];
                struct { float inv_area, a0, b0, a1, b1, w0_start, w1_start; uint32_t bbx, bby, flags; } r;
                // bbox, render.rs:1455-1458
                uint32_t min_x = f2u_sat(rmax(rmin(rmin(v1.x, v2.x), v3.x), 0.0f));
                uint32_t max_x = f2u_sat(rmin(rmax(rmax(v1.x, v2.x), v3.x) + 1.0f, (float)fp.width));
                uint32_t min_y = f2u_sat(rmax(rmin(rmin(v1.y, v2.y), v3.y), 0.0f));
                uint32_t max_y = f2u_sat(rmin(rmax(rmax(v1.y, v2.y), v3.y) + 1.0f, (float)fp.height));
                bool empty = min_x >= max_x || min_y >= max_y;
                float area = (v2.y - v3.y) * (v1.x - v3.x) + (v3.x - v2.x) * (v1.y - v3.y);      // :1500
                if (__builtin_fabsf(area) < K::AREA_EPS) empty = true;                            // :1501-1503
                if (empty) { min_x = max_x = min_y = max_y = 0; }
                r.bbx = min_x | (max_x << 16); r.bby = min_y | (max_y << 16);
                span = pack_tile_span(r.bbx, r.bby, empty ? F_EMPTY : 0u, fp, n_tiles);
                if (db.fill && n_tiles) {     // direct binning: the list slot in the first tile is requested now, used after the record build
                    db_cls = db.with_class && transparent;
                    const uint32_t t0 = ((span >> 16) & 0xFF) * fp.tiles_x + (span & 0xFF);
                    agg_issue(db.fill, t0 * FILL_PAD + (db_cls ? 1u : 0u), true, threadIdx.x & 63u, db_first);
                }
                // multi-GPU band sharding: every rank decides visibility for every face (triangles_drawn, painter's keys), but only the
                // surfaces reaching its own rows are ever read again: the triangle prologue, the exactness guard, the lighting and the
                // record build are skipped for all the others
                const bool need_rec = !fp.band_only || n_tiles != 0;
                bool slow = !fp.fixed_point || fp.ortho;
                bool needs_dither = false;
                r.inv_area = 0.0f; r.w0_start = r.w1_start = 0.0f; r.flags = 0;
                // (the world normals of a lit frame are requested HERE, with the attributes, and used at the very end: behind the record
                // build their round trip is hidden -- requested where they are used it was 12 of the lit kernel's 77 us, measured by
                // replacing them with a constant)
                V3 wn[3] = { { 0.0f, 0.0f, 0.0f }, { 0.0f, 0.0f, 0.0f }, { 0.0f, 0.0f, 0.0f } };
                if (need_rec) {
                if (pos12 && fp.shading != B32_SHADE_NONE) {
                    // packed streams, lit frame: attributes AND normal of a vertex from the 24-byte stream (one cache line per face instead
                    // of one in the attribute stream and one in a normal stream of its own: 12 of the lit kernel's 77 us)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const float* lp = attr12 + (size_t)fp.nv * 3 + (size_t)vi[j] * 6;
                        uvx[j] = lp[0]; uvy[j] = lp[1]; col[j] = reinterpret_cast<const uint32_t*>(lp)[2];
                        wn[j] = { lp[3], lp[4], lp[5] };
                    }
                } else if (pos12) {   // packed streams: the survivor's UVs and vertex colours only now (12 B per vertex, culled faces never fetch them)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const float* ap = attr12 + (size_t)vi[j] * 3;
                        uvx[j] = ap[0]; uvy[j] = ap[1]; col[j] = reinterpret_cast<const uint32_t*>(ap)[2];
                    }
                } else if (fp.shading != B32_SHADE_NONE) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const float* np = reinterpret_cast<const float*>(verts) + (size_t)vi[j] * 9 + 5;
                        wn[j] = { np[0], np[1], np[2] };
                    }
                }
                if (fp.has_fog && m_has_fog) {
                    const uint32_t fogc = m_fog.r | (m_fog.g << 8) | (m_fog.b << 16) | ((uint32_t)m_fog.blend << 24);
#pragma unroll
                    for (int j = 0; j < 3; ++j) col[j] = fog_color(col[j], fogc, fog_factor(camz[j], m_fog.start, m_fog.falloff));
                }
                r.inv_area = 1.0f / area;
                r.a0 = v2.y - v3.y; r.b0 = v3.x - v2.x; r.a1 = v3.y - v1.y; r.b1 = v1.x - v3.x;   // :1507-1510
                const float start_x = (float)min_x, start_y = (float)min_y;
                r.w0_start = r.a0 * (start_x - v3.x) + r.b0 * (start_y - v3.y);                    // :1517
                r.w1_start = r.a1 * (start_x - v3.x) + r.b1 * (start_y - v3.y);                    // :1518
                // Closed-form eligibility: with integer vertices every value the reference's incremental walk ever holds
                // is an exact integer when |w| < 2^24 over the bbox and both start products are < 2^24.
                if (!slow && !empty) {
                    const float lim = 4194304.0f;   // 2^22
                    const float cmax = rmax(rmax(rmax(__builtin_fabsf(v1.x), __builtin_fabsf(v1.y)), rmax(__builtin_fabsf(v2.x), __builtin_fabsf(v2.y))),
                                            rmax(__builtin_fabsf(v3.x), __builtin_fabsf(v3.y)));
                    // quick acceptance (nearly every triangle): every edge coefficient is at most amax, every offset from v3 to a
                    // corner of the clipped box at most dmax (integers below 2^23), so every product is at most amax * dmax and every
                    // sum of two at most twice that; the float product rounds monotonically and 2^24 is representable
                    const float amax = rmax(rmax(__builtin_fabsf(r.a0), __builtin_fabsf(r.b0)), rmax(__builtin_fabsf(r.a1), __builtin_fabsf(r.b1)));
                    const float dmax = rmax(rmax(__builtin_fabsf((float)min_x - v3.x), __builtin_fabsf((float)(max_x - 1) - v3.x)),
                                            rmax(__builtin_fabsf((float)min_y - v3.y), __builtin_fabsf((float)(max_y - 1) - v3.y)));
                    if (!(cmax <= lim)) slow = true;
                    else if (!(2.0f * (amax * dmax) < 16777216.0f)) {
                        // every operand is an integer-valued float below 2^23, so each product / sum below is exact whenever its
                        // true value is below 2^24, and rounds to >= 2^24 otherwise (rounding is monotonic and 2^24 is
                        // representable): the float comparisons decide exactly what 64-bit integer arithmetic would
                        const float dxs[2] = { (float)min_x - v3.x, (float)(max_x - 1) - v3.x };
                        const float dys[2] = { (float)min_y - v3.y, (float)(max_y - 1) - v3.y };
                        const float L = 16777216.0f;   // 2^24
#pragma unroll
                        for (int a = 0; a < 2; ++a)
#pragma unroll
                            for (int b = 0; b < 2; ++b) {
                                const float p0 = r.a0 * dxs[a], q0 = r.b0 * dys[b], p1 = r.a1 * dxs[a], q1 = r.b1 * dys[b];
                                if (!(__builtin_fabsf(p0) < L && __builtin_fabsf(q0) < L && __builtin_fabsf(p1) < L && __builtin_fabsf(q1) < L &&
                                      __builtin_fabsf(p0 + q0) < L && __builtin_fabsf(p1 + q1) < L)) slow = true;
                            }
                    }
                }
                const bool vc_diff = (col[i1] != col[i2]) || (col[i2] != col[i3]);                  // Color equality incl. blend, types.rs:719
                needs_dither = fp.dithering && (fp.shading == B32_SHADE_GOURAUD || have_tex || vc_diff);   // :1487-1492
                const uint32_t eff_blend = have_tex ? tex_blend : face_blend;                       // :1450-1452
                r.flags = (have_tex ? tid : F_TEX_NONE) | (black_tr ? F_BLACK_TR : 0) | (eff_blend << F_BLEND_SHIFT) |
                          (needs_dither ? F_DITHER : 0) | (slow ? F_SLOW : 0) | (transparent ? F_TRANSP : 0) |
                          (empty ? F_EMPTY : 0) | (editor_alpha << F_ALPHA_SHIFT);
                }   // need_rec
                // painter's key, render.rs:2529-2531 / 2538-2540. Perspective keys are > 5 (cam z > 0.1), so the sign bit
                // is free: bit 31 = transparent class, low 31 bits = 0x7FFFFFFF - bits(z) => ascending radix == descending z,
                // stability of the LSD passes == stability of slice::sort_by.
                float kz = (v1.z + v2.z + v3.z) / 3.0f;
                // z-buffer mode never sorts the opaque list (render.rs:2535): key 0 keeps it in face order through the stable
                // passes, and a NaN key there is never compared
                if (fp.zmode && !transparent) kz = 0.0f;
                if (kz != kz) { nan_key = true; kz = 0.0f; }
                if (kz == 0.0f) kz = 0.0f;                    // -0.0 == +0.0 under partial_cmp
                uint32_t zb = __float_as_uint(kz);
                key = ((0x7FFFFFFFu - (zb & 0x7FFFFFFFu)) & 0x7FFFFFFFu) | (transparent ? 0x80000000u : 0u);
                // orthographic depths may be negative: all 32 bits order the depth (descending), the class partition is a pass of
                // its own (launch_class_keys).  ~zsort_key(z) == 0xFFFFFFFF only for a NaN pattern, which was replaced above.
                if (fp.ortho) key = ~zsort_key(kz);
                if (fp.zmode && !transparent) key = 0;
                if (key == KEY_INVALID) key = 0xFFFFFFFEu;    // unreachable for z > 5; keeps the sentinel unique
                if (need_rec) {
                    // vertices as i16 when all six are integers within range: the fixed-point snap yields integers (project_fixed), and
                    // F_SLOW surfaces always take the wide form
                    const float cmax16 = rmax(rmax(rmax(__builtin_fabsf(v1.x), __builtin_fabsf(v1.y)), rmax(__builtin_fabsf(v2.x), __builtin_fabsf(v2.y))),
                                              rmax(__builtin_fabsf(v3.x), __builtin_fabsf(v3.y)));
                    const bool fits = !slow && cmax16 <= 32767.0f;
                    auto pk16 = [](float x, float y) { return ((uint32_t)(int32_t)x & 0xFFFFu) | ((uint32_t)(int32_t)y << 16); };
                    uint4 c0, c1;
                    c0.x = fits ? pk16(v1.x, v1.y) : COV_WIDE; c0.y = fits ? pk16(v2.x, v2.y) : 0u; c0.z = fits ? pk16(v3.x, v3.y) : 0u;
                    c0.w = __float_as_uint(r.inv_area);
                    c1 = make_uint4(r.bbx, r.bby, key, r.flags);
                    uint4* cp = reinterpret_cast<uint4*>(recs.cov + rslot);
                    cp[0] = c0; cp[1] = c1;
                    const uint32_t slot = have_tex ? tid : F_TEX_NONE;
                    const uint32_t shf = (black_tr ? SH_BLACK_TR : 0u) | (needs_dither ? SH_DITHER : 0u) | (slow ? SH_SLOW : 0u);
                    uint4* sp = reinterpret_cast<uint4*>(recs.shade + rslot);
                    sp[0] = make_uint4(__float_as_uint(v1.x), __float_as_uint(v1.y), __float_as_uint(v2.x), __float_as_uint(v2.y));
                    sp[1] = make_uint4(__float_as_uint(v3.x), __float_as_uint(v3.y), __float_as_uint(r.inv_area), (col[i1] & 0xFFFFFFu) | ((slot & 0xFFu) << 24));
                    sp[2] = make_uint4(__float_as_uint(uvx[i1]), __float_as_uint(uvx[i2]), __float_as_uint(uvx[i3]), __float_as_uint(uvy[i1]));
                    sp[3] = make_uint4(__float_as_uint(uvy[i2]), __float_as_uint(uvy[i3]), (col[i2] & 0xFFFFFFu) | ((slot >> 8) << 24), (col[i3] & 0xFFFFFFu) | (shf << 24));
                    if (fp.zmode || !fp.affine || slow) {
                        uint4* xp = reinterpret_cast<uint4*>(recs.aux + rslot);
                        xp[0] = make_uint4(__float_as_uint(1.0f / v1.z), __float_as_uint(1.0f / v2.z), __float_as_uint(1.0f / v3.z), __float_as_uint(r.w0_start));   // :1546-1548
                        xp[1] = make_uint4(__float_as_uint(r.w1_start), 0u, 0u, 0u);
                    }
                }
                if (db.fill && n_tiles) {     // the face id into the list of every tile of the span (any order inside a list)
                    const uint32_t cap = db_cls ? db.cap_transparent : db.cap_opaque;
                    const uint32_t tx0 = span & 0xFF, tx1 = (span >> 8) & 0xFF, ty0 = (span >> 16) & 0xFF;
                    bool over = false;
                    uint32_t tx = tx0, ty = ty0;
                    AggSlot g = db_first;
                    // the k-th tile of every lane's span together (wave-uniform loop: the group ballots need every lane of this block)
                    for (uint32_t k = 0; __ballot(k < n_tiles); ++k) {
                        const bool act = k < n_tiles;
                        const uint32_t tile = ty * fp.tiles_x + tx;
                        if (k) agg_issue(db.fill, tile * FILL_PAD + (db_cls ? 1u : 0u), act, threadIdx.x & 63u, g);
                        const uint32_t pos = agg_position(g);
                        if (act) {
#ifdef B32_EXP_BIN_NO_LIST_STORE                                   // experiment (frames are WRONG): reservations without the 4-byte list stores
                            if (pos >= cap) over = true;
#else
                            if (pos < cap) db.lists[(size_t)tile * db.region + (db_cls ? db.region - 1u - pos : pos)] = rslot;
                            else over = true;
#endif
                            if (++tx > tx1) { tx = tx0; ++ty; }
                        }
                    }
                    if (over) { Events* ev = events_of(ctrl); if (db_cls) ev->long_transparent = db.epoch; else ev->overflow = db.epoch; }
                }
                // (the shades last: the record's values are stored and dead by now, the lighting has the registers to itself)
                if (need_rec && fp.shading != B32_SHADE_NONE) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) if (backface) wn[j] = scale3(wn[j], -1.0f);
                    float* sh = shades + (size_t)rslot * 9;
                    if (fp.shading == B32_SHADE_FLAT) {                                             // :1466-1469
                        V3 center = scale3(add3(add3(wpos[i1], wpos[i2]), wpos[i3]), 1.0f / 3.0f);
                        V3 nrm = normalize3(scale3(add3(add3(wn[i1], wn[i2]), wn[i3]), 1.0f / 3.0f));
                        float s[3]; shade_multi(nrm, center, lights, fp.n_lights, m_ambient, s);
                        for (int j = 0; j < 9; ++j) sh[j] = s[j % 3];
                    } else {                                                                        // :1475-1483
                        shade_multi(wn[i1], wpos[i1], lights, fp.n_lights, m_ambient, sh);
                        shade_multi(wn[i2], wpos[i2], lights, fp.n_lights, m_ambient, sh + 3);
                        shade_multi(wn[i3], wpos[i3], lights, fp.n_lights, m_ambient, sh + 6);
                    }
                }
            }
        }
        if (fp.wire_collect && bad_index) wire[f].kind = 0;
    }
    // frame counters: ballot per wave -> LDS -> one 5-word record per block (reduced by k_after_setup; no atomics)
    const unsigned long long mv = __ballot(visible), mt = __ballot(transparent);
    {   // painter's key and tile span per SLOT (see `slot` above): survivors at the front of the wave's 64 slots, KEY_INVALID behind them
        const uint32_t lane = threadIdx.x & 63u, nvis = (uint32_t)__popcll(mv), wbase = f - lane;
        if (visible) {
            const uint32_t slot = wbase + (uint32_t)__popcll(mv & ((1ull << lane) - 1ull));
            keys[slot] = key;
            if (spans) spans[slot] = span;
            if (face_of) face_of[slot] = f;
        }
        if (in.live && lane >= nvis) { keys[wbase + lane] = KEY_INVALID; if (spans) spans[wbase + lane] = 0xFFFFFFFFu; }
    }
    const unsigned long long mn_op = __ballot(nan_key && !transparent), mn_tr = __ballot(nan_key && transparent), mb = __ballot(bad_index);
    const uint32_t wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        wpart[wv][0] = (uint32_t)__popcll(mv); wpart[wv][1] = (uint32_t)__popcll(mt);
        wpart[wv][2] = (uint32_t)__popcll(mn_op); wpart[wv][3] = (uint32_t)__popcll(mn_tr); wpart[wv][4] = mb ? 1u : 0u;
        if (db.fill && (mn_op | mn_tr | mb)) {   // direct binning: nobody reduces the counters before the fill, which only does so after one of these
            Events* ev = events_of(ctrl);
            if (mb) ev->bad_index = db.epoch;
            if (mn_op) ev->nan_opaque = db.epoch;
            if (mn_tr) ev->nan_transparent = db.epoch;
        }
    }
    for (int off = 32; off > 0; off >>= 1) n_tiles += __shfl_down(n_tiles, off);      // (tile, surface) pairs of this block
    if ((threadIdx.x & 63) == 0) wpart[wv][5] = n_tiles;
    __syncthreads();
    // (one record per group of 256 faces, as before: the consumers index them by face / 256)
    const uint32_t grp = blockIdx.x * SETUP_FPT + g;
    if (threadIdx.x < 6 && grp * 256u < fp.nf) partials[grp * 8 + threadIdx.x] = wpart[0][threadIdx.x] + wpart[1][threadIdx.x] + wpart[2][threadIdx.x] + wpart[3][threadIdx.x];
    __syncthreads();
    }   // SETUP_FPT faces per thread
}

void launch_setup(hipStream_t s, const FrameParams& fp, const B32Vertex* verts, const B32Face* faces, const TexDesc* tex,
                  const B32Light* lights, const LightSet& ls, const MeshTable& mt, RecArrays recs, const DirectBin& db, float* shades, uint32_t* keys, uint32_t* spans, uint32_t* partials,
                  Ctrl* ctrl, WireTri* wire, int n_cu, const float* pos12, const float* attr12, uint32_t* face_of) {
    (void)n_cu;
    if (fp.nf == 0) return;
    const bool plain = fp.fixed_point && !fp.ortho && !fp.has_fog && !fp.wire_collect && fp.shading == B32_SHADE_NONE && !fp.xray && !fp.batched;
    // one face per thread: 52 VGPRs in the plain form = 8 waves per SIMD (two faces per thread with their loads issued up front: 73 VGPRs,
    // 43 us instead of 39 at 1 M faces; three: 49 us)
    const dim3 g1((fp.nf + 255) / 256);
    const bool lit = fp.fixed_point && !fp.ortho && !fp.has_fog && !fp.wire_collect && fp.shading != B32_SHADE_NONE && !fp.xray && !fp.batched;
    if (lit) hipLaunchKernelGGL((k_setup<1, 2>), g1, dim3(256), 0, s, fp, verts, faces, tex, lights, ls, mt, recs, db, shades, keys, spans, partials, ctrl, wire, pos12, attr12, face_of);
    else if (plain) hipLaunchKernelGGL((k_setup<1, 1>), g1, dim3(256), 0, s, fp, verts, faces, tex, lights, ls, mt, recs, db, shades, keys, spans, partials, ctrl, wire, pos12, attr12, face_of);
    else hipLaunchKernelGGL((k_setup<1, 0>), g1, dim3(256), 0, s, fp, verts, faces, tex, lights, ls, mt, recs, db, shades, keys, spans, partials, ctrl, wire, pos12, attr12, face_of);
}

// ---------------------------------------------------------------- merged mesh of a batched frame
// One member mesh into the merged vertex / face arrays: vertex indices and texture ids move by the member's bases, and the member's
// number goes into the face's spare byte (k_setup's MeshTable index).  What the reference decides per call keeps its meaning: a vertex
// index past the member's OWN vertex count stays out of range (index panic, render.rs:2375), a texture id past its OWN texture count
// stays None (textures.get(id), render.rs:2554).
__global__ void k_merge_mesh(const B32Vertex* __restrict__ sv, uint32_t nv, const B32Face* __restrict__ sf, uint32_t nf, uint32_t nt,
                             B32Vertex* __restrict__ dv, B32Face* __restrict__ df, uint32_t vbase, uint32_t tbase, uint32_t mesh) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nv) {
        const uint32_t* s = reinterpret_cast<const uint32_t*>(sv) + (size_t)i * 9;
        uint32_t* d = reinterpret_cast<uint32_t*>(dv) + (size_t)(vbase + i) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) d[k] = s[k];
    }
    if (i < nf) {
        const uint32_t* s = reinterpret_cast<const uint32_t*>(sf) + (size_t)i * 5;
        uint32_t* d = reinterpret_cast<uint32_t*>(df) + (size_t)i * 5;
#pragma unroll
        for (int k = 0; k < 3; ++k) d[k] = s[k] < nv ? s[k] + vbase : 0xFFFFFFFFu;
        d[3] = (s[3] != B32_NO_TEXTURE && s[3] < nt) ? s[3] + tbase : B32_NO_TEXTURE;
        d[4] = (s[4] & 0x00FFFFFFu) | (mesh << 24);
    }
}
void launch_merge_mesh(hipStream_t s, const B32Vertex* sv, uint32_t nv, const B32Face* sf, uint32_t nf, uint32_t nt, B32Vertex* dv, B32Face* df,
                       uint32_t vbase, uint32_t tbase, uint32_t mesh) {
    const uint32_t n = nv > nf ? nv : nf;
    if (n) hipLaunchKernelGGL(k_merge_mesh, dim3((n + 255) / 256), dim3(256), 0, s, sv, nv, sf, nf, nt, dv, df, vbase, tbase, mesh);
}
__global__ void k_offset_tex(const TexDesc* __restrict__ src, uint32_t nt, TexDesc* __restrict__ dst, uint32_t texel_base) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nt) { TexDesc d = src[i]; d.offset += texel_base; dst[i] = d; }
}
void launch_offset_tex(hipStream_t s, const TexDesc* src, uint32_t nt, TexDesc* dst, uint32_t texel_base) {
    if (nt) hipLaunchKernelGGL(k_offset_tex, dim3((nt + 255) / 256), dim3(256), 0, s, src, nt, dst, texel_base);
}

// ---------------------------------------------------------------- gate of a pipelined setup kernel (two frames in flight)
// One wave on the side stream, in front of the next frame's k_setup: it returns once the fill kernel of the frame before has handed
// out `need` tiles from its cursor -- so that the setup kernel runs in that kernel's thinning tail instead of beside its busy start --
// or after `patience` ticks of the 100 MHz wall clock (a fill that aborted, or was no tile kernel at all, never moves its cursor).
// Round 6: the gate also carries the ORDER the frame sets need.  The setup kernel behind it overwrites a frame set (records, lists, counters,
// control block) that the fill n_sets frames back read; that fill precedes the fill of `prev` on the main stream, so once the latter has STARTED
// (Events::fill_started == start_seq, published by its workgroup 0 with a device-scope atomic) the former has ended, caches written back and all.
// This used to be a cross-stream event recorded behind every fill -- ~6 us of the main stream per frame (its system-scope release and the
// signal) whether anyone waited for it or not.  start_seq == 0: no such wait (the caller ordered the streams by an event).  This wait is for
// correctness: its patience is 2 s (start_patience, 10-ns ticks), and running out of it is reported (sticky bit 3 of the waiting frame's control block: B32_E_HIP).
__global__ void k_gate(Ctrl* __restrict__ prev, uint32_t need, uint32_t patience, uint32_t start_seq, Ctrl* __restrict__ mine, uint32_t start_patience) {
    if (start_seq) {
        const unsigned long long t0 = wall_clock64();
        while ((int32_t)(__hip_atomic_fetch_add(&events_of(prev)->fill_started, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - start_seq) < 0) {
            if (wall_clock64() - t0 > (unsigned long long)start_patience) { atomicOr(&mine->sticky, 8u); break; }
            __builtin_amdgcn_s_sleep(16);
        }
    }
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(&prev->tile_cursor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
        if (wall_clock64() - t0 > patience) break;
        __builtin_amdgcn_s_sleep(64);
    }
}
// The hand-over setup(i + 1) -> fill(i + 1) between the two streams without a cross-stream event (whose barrier packet costs the main
// stream ~12 us even when the event is long signalled): k_flag, one lane on the side stream BEHIND the setup kernel, publishes the frame's
// epoch; k_join, one lane on the main stream in FRONT of the fill, returns when it reads that epoch.  Kernel boundaries do the cache
// maintenance: k_flag starts after the setup kernel's end-of-kernel release, the fill starts with its own acquire after k_join.  Neither
// holds anything the other needs (k_join follows the previous fill on its stream, so the GPU is the setup kernel's while it spins); a
// setup kernel that never arrives (patience: 2 s) aborts the frame -- Events::join_abort, read by the fill with the other event words: it draws
// nothing but the folded clear -- and is reported by b32_frame_finish (sticky bit 3).
__global__ void k_flag(Ctrl* __restrict__ ctrl, uint32_t epoch) {
    __hip_atomic_exchange(&events_of(ctrl)->setup_done, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_join(Ctrl* __restrict__ ctrl, uint32_t epoch, uint32_t patience) {
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_fetch_add(&events_of(ctrl)->setup_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
        // (the epoch word lives in Events, which no kernel resets: a setup kernel that arrives AFTER the patience cannot take the abort back,
        // as it could with Ctrl::abort, which it zeroes when it starts)
        if (wall_clock64() - t0 > patience) { atomicOr(&ctrl->sticky, 8u); (void)__hip_atomic_exchange(&events_of(ctrl)->join_abort, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
        __builtin_amdgcn_s_sleep(4);
    }
}
__global__ void k_flag_poll(Ctrl* __restrict__ ctrl, uint32_t seq) {
    __hip_atomic_exchange(&events_of(ctrl)->poll_done, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_flag_wbin(Ctrl* __restrict__ ctrl, uint32_t epoch) {
    __hip_atomic_exchange(&events_of(ctrl)->wbin_done, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
void launch_flag(hipStream_t s, Ctrl* ctrl, uint32_t epoch) { hipLaunchKernelGGL(k_flag, dim3(1), dim3(1), 0, s, ctrl, epoch); }
void launch_flag_poll(hipStream_t s, Ctrl* ctrl, uint32_t seq) { hipLaunchKernelGGL(k_flag_poll, dim3(1), dim3(1), 0, s, ctrl, seq); }
void launch_flag_wbin(hipStream_t s, Ctrl* ctrl, uint32_t epoch) { hipLaunchKernelGGL(k_flag_wbin, dim3(1), dim3(1), 0, s, ctrl, epoch); }
void launch_join(hipStream_t s, Ctrl* ctrl, uint32_t epoch, uint32_t patience) { hipLaunchKernelGGL(k_join, dim3(1), dim3(1), 0, s, ctrl, epoch, patience); }
void launch_gate(hipStream_t s, Ctrl* prev, uint32_t need, uint32_t patience, uint32_t start_seq, Ctrl* mine, uint32_t start_patience) {
    hipLaunchKernelGGL(k_gate, dim3(1), dim3(1), 0, s, prev, need, patience, start_seq, mine, start_patience);
}

// ---------------------------------------------------------------- stage tap: project_fixed for n positions
__global__ void k_project_fixed(const float* __restrict__ pos, uint32_t n, B32Camera cam, uint32_t w, uint32_t h,
                                int32_t* __restrict__ sx, int32_t* __restrict__ sy, float* __restrict__ z) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    CamFx k = make_camfx(cam, w, h);
    V3 p = { pos[3 * i], pos[3 * i + 1], pos[3 * i + 2] };
    int32_t x, y;
    project_fixed_dev(p.x, p.y, p.z, k, x, y);
    V3 rel = sub3(p, ld3(cam.position));
    sx[i] = x; sy[i] = y; z[i] = dot3(rel, ld3(cam.basis_z)) + K::MESH_DISTANCE;     // render.rs:2343-2345
}
void launch_project_fixed(hipStream_t s, const float* pos, uint32_t n, B32Camera cam, uint32_t w, uint32_t h,
                          int32_t* sx, int32_t* sy, float* z) {
    if (!n) return;
    hipLaunchKernelGGL(k_project_fixed, dim3((n + 255) / 256), dim3(256), 0, s, pos, n, cam, w, h, sx, sy, z);
}

// ---------------------------------------------------------------- skip mask of the texel pool
__global__ void k_build_mask(const uint16_t* __restrict__ t15, const uint32_t* __restrict__ t32, uint32_t n, uint32_t* __restrict__ mask) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool bit = false;
    if (i < n) bit = t32 ? ((t32[i] >> 24) == B32_BLEND_ERASE) : ((t15[i] & ~K::C15_SEMI_BIT & 0xFFFFu) == 0);
    const unsigned long long m = __ballot(bit);
    if ((threadIdx.x & 63) == 0 && i < n) { mask[i >> 5] = (uint32_t)m; mask[(i >> 5) + 1] = (uint32_t)(m >> 32); }
}
void launch_build_mask(hipStream_t s, const uint16_t* texels15, const uint32_t* texels32, uint32_t n, uint32_t* mask) {
    if (!n) return;
    hipLaunchKernelGGL(k_build_mask, dim3((n + 255) / 256), dim3(256), 0, s, texels15, texels32, n, mask);
}

// ---------------------------------------------------------------- constants tap (see B32_CONSTANTS in b32_device.h)
__global__ void k_constants(uint32_t* __restrict__ consts, uint8_t* __restrict__ unr, int32_t* __restrict__ dither) {
    if (blockIdx.x != 0) return;
    if (threadIdx.x == 0) {
        uint32_t i = 0;
#define B32_K_F(v) __float_as_uint((float)(v))
#define B32_K_I(v) (uint32_t)(v)
#define B32_K_STORE(name, kind, v) consts[i++] = B32_K_##kind(v);
        B32_CONSTANTS(B32_K_STORE)
#undef B32_K_STORE
#undef B32_K_F
#undef B32_K_I
    }
    for (uint32_t t = threadIdx.x; t < K::UNR_ENTRIES; t += blockDim.x) unr[t] = g_unr.v[t];
    if (threadIdx.x < 16) dither[threadIdx.x] = dither_offset(threadIdx.x & 3, threadIdx.x >> 2);
}
void launch_constants(hipStream_t s, uint32_t* consts, uint8_t* unr, int32_t* dither) {
    hipLaunchKernelGGL(k_constants, dim3(1), dim3(256), 0, s, consts, unr, dither);
}

// ---------------------------------------------------------------- f32 semantics self-test
__global__ void k_selftest(int op, const float* a, const float* b, const float* c, float* out, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r;
    switch (op) {
        case 0: r = a[i] * b[i] + c[i]; break;          // must round twice (no FMA contraction)
        case 1: r = a[i] / b[i]; break;                 // correctly rounded
        case 2: r = __builtin_sqrtf(a[i]); break;       // correctly rounded
        case 4: r = acosf_musl(a[i]); break;            // f32::acos of the reference's wasm32 target (spot lights)
        case 5: r = __int_as_float(f2i32_sat(a[i])); break;                  // Rust `as i32`: the bits of the result
        case 6: r = __uint_as_float(f2u_sat(a[i])); break;                   // Rust `as u32`
        case 7: r = __int_as_float(fx_mul(__float_as_int(a[i]), __float_as_int(b[i]))); break;   // Fixed32::mul_fixed on the operands' bits
        case 8: {                                       // wire_t_fast against `/`: a[i] = N; every k in [0, N]; the count of differing results
            const float Nf = a[i], rN = 1.0f / Nf; uint32_t bad = 0;
            for (float kf = 0.0f; kf <= Nf; kf += 1.0f) bad += __float_as_uint(wire_t_fast(kf, Nf, rN)) != __float_as_uint(kf / Nf);
            r = (float)bad; break;
        }
        case 9: case 10: {                              // rcp_exact against `/` on the 65536 consecutive bit patterns from a[i]'s: count / first differing pattern
            const uint32_t base = __float_as_uint(a[i]); uint32_t bad = 0, first = 0;
            for (uint32_t k = 0; k < 65536u; ++k) {
                const float x = __uint_as_float(base + k);
                const uint32_t want = __float_as_uint(1.0f / x), got = __float_as_uint(rcp_exact(x));
                const bool nan_w = (want & 0x7FFFFFFFu) > 0x7F800000u, nan_g = (got & 0x7FFFFFFFu) > 0x7F800000u;
                if (nan_w ? !nan_g : want != got) { if (!bad) first = base + k; ++bad; }
            }
            r = op == 9 ? (float)bad : __uint_as_float(first); break;
        }
        default: r = (a[i] + b[i]) / c[i]; break;
    }
    out[i] = r;
}
void launch_selftest(hipStream_t s, int op, const float* a, const float* b, const float* c, float* out, uint32_t n) {
    if (!n) return;
    hipLaunchKernelGGL(k_selftest, dim3((n + 255) / 256), dim3(256), 0, s, op, a, b, c, out, n);
}

// ---------------------------------------------------------------- Framebuffer::clear (render.rs:36-45): 16-B coalesced stores
__global__ void k_clear(uint32_t* __restrict__ fb, size_t n_px, uint32_t rgba) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    for (; i + 3 < n_px; i += stride) *reinterpret_cast<uint4*>(fb + i) = make_uint4(rgba, rgba, rgba, rgba);
    if (i < n_px) for (size_t j = i; j < n_px; ++j) fb[j] = rgba;
}
void launch_clear(hipStream_t s, uint32_t* fb, size_t n_px, uint32_t rgba) {
    if (!n_px) return;
    size_t quads = (n_px + 3) / 4;
    uint32_t blocks = (uint32_t)((quads + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_clear, dim3(blocks), dim3(256), 0, s, fb, n_px, rgba);
}

// ---------------------------------------------------------------- staged upload (drop-in calls with small meshes)
__global__ __launch_bounds__(256) void k_upload(const uint4* __restrict__ arena, UploadSegs segs) {
    const uint32_t gtid = blockIdx.x * 256u + threadIdx.x, nthr = gridDim.x * 256u;
    for (uint32_t k = 0; k < segs.count; ++k) {
        const uint4* src = arena + (segs.src_off[k] >> 4);
        uint4* dst = reinterpret_cast<uint4*>(segs.dst[k]);
        for (uint32_t i = gtid; i < segs.n16[k]; i += nthr) dst[i] = src[i];
    }
}
// Packed vertex streams of a resident scene (structure of arrays, built once from the uploaded B32Vertex array): positions, 12 bytes
// each, which k_setup reads for EVERY face, and (u, v, rgba), 12 bytes each, which it reads only for the faces that survive the cull
// (and, on a band-sharded frame, reach this rank's rows).  A third stream of 24 bytes per vertex follows the attributes (attr12 + 3 * nv
// floats): attributes and normal together, what lit frames read for the surviving faces -- out of the 36-byte B32Vertex a wave of neighbouring faces pulls every sector of the
// vertex array for a third of its bytes.
// with_lit == 0 (a mesh that has not been drawn lit so far): the 24-byte stream is neither allocated nor written -- 72 MB of HBM and of
// write traffic less per 1 M-face mesh; the first lit frame packs again with it (frame_positions, b32_frame.hip).
__global__ void k_pack_streams(const B32Vertex* __restrict__ verts, uint32_t nv, float* __restrict__ pos12, float* __restrict__ attr12, uint32_t with_lit) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    const float* vp = reinterpret_cast<const float*>(verts) + (size_t)i * 9;
    pos12[3 * (size_t)i] = vp[0]; pos12[3 * (size_t)i + 1] = vp[1]; pos12[3 * (size_t)i + 2] = vp[2];
    attr12[3 * (size_t)i] = vp[3]; attr12[3 * (size_t)i + 1] = vp[4]; attr12[3 * (size_t)i + 2] = vp[8];      // u, v, rgba (bit copy)
    if (with_lit) {
        float* lit24 = attr12 + 3 * (size_t)nv + 6 * (size_t)i;
        lit24[0] = vp[3]; lit24[1] = vp[4]; lit24[2] = vp[8]; lit24[3] = vp[5]; lit24[4] = vp[6]; lit24[5] = vp[7];   // the same + the normal
    }
}
void launch_pack_streams(hipStream_t s, const B32Vertex* verts, uint32_t nv, float* pos12, float* attr12, bool with_lit) {
    if (nv) hipLaunchKernelGGL(k_pack_streams, dim3((nv + 255) / 256), dim3(256), 0, s, verts, nv, pos12, attr12, with_lit ? 1u : 0u);
}
__global__ void k_ctrl_out(Ctrl* __restrict__ ctrl, uint4* __restrict__ dst) {
    if (threadIdx.x == 0) { unsigned long long* t = reinterpret_cast<Stamps*>(ctrl + 1)->t; if (!t[ST_END]) t[ST_END] = wall_clock64(); }
    __syncthreads();
    if (threadIdx.x < (sizeof(Ctrl) + sizeof(Stamps)) / 16) dst[threadIdx.x] = reinterpret_cast<const uint4*>(ctrl)[threadIdx.x];
}
void launch_ctrl_out(hipStream_t s, Ctrl* ctrl, void* dst128) {
    hipLaunchKernelGGL(k_ctrl_out, dim3(1), dim3(64), 0, s, ctrl, reinterpret_cast<uint4*>(dst128));
}
void launch_upload(hipStream_t s, const void* arena_dev, const UploadSegs& segs) {
    if (!segs.count) return;
    uint32_t total = 0;
    for (uint32_t k = 0; k < segs.count; ++k) total += segs.n16[k];
    uint32_t blocks = (total + 255) / 256;
    if (blocks > 256) blocks = 256;
    if (!blocks) return;
    hipLaunchKernelGGL(k_upload, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const uint4*>(arena_dev), segs);
}

// ---------------------------------------------------------------- Clut::lookup expansion (types.rs:390-397, mesh_editor.rs:669-682)
__global__ void k_expand_indexed(const uint8_t* __restrict__ idx, uint32_t n, const uint16_t* __restrict__ clut, uint32_t clut_len,
                                 uint16_t* __restrict__ out, uint32_t* __restrict__ skippable) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool skip = false;
    if (i < n) {
        uint32_t k = idx[i];
        const uint16_t col = k < clut_len ? clut[k] : (uint16_t)0;
        out[i] = col;
        skip = (col & ~K::C15_SEMI_BIT & 0xFFFFu) == 0;
    }
    const unsigned long long m = __ballot(skip);
    if (skippable && m && (threadIdx.x & 63) == 0) atomicAdd(skippable, (uint32_t)__popcll(m));
}
void launch_expand_indexed(hipStream_t s, const uint8_t* idx, uint32_t n, const uint16_t* clut, uint32_t clut_len, uint16_t* out, uint32_t* skippable) {
    if (!n) return;
    hipLaunchKernelGGL(k_expand_indexed, dim3((n + 255) / 256), dim3(256), 0, s, idx, n, clut, clut_len, out, skippable);
}

}  // namespace b32

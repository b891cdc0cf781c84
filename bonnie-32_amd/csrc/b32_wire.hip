// b32_wire.hip — the wireframe phases of render_mesh_15 (render.rs:2574-2635) and the orthographic class pass.
//
// Reference: back-faces (kept by near/fog culling, not x-ray) contribute their three screen edges, coordinates `as i32`,
// direction-normalised so that (a,b)-(c,d) == (c,d)-(a,b); an edge is drawn once, with the depths of its FIRST occurrence
// in (face, edge) order (the O(n^2) `unique_edges.iter().any(..)` scan, render.rs:2589-2594), by draw_line_3d
// (render.rs:757-817): Bresenham, depth interpolated by step count, `z < zbuffer`, colour (80,80,100), no depth write.
// Front-face edges (wireframe_overlay) are drawn by draw_line (render.rs:716-750), no depth test, colour (200,200,220).
//
// GPU form: every line writes one colour and never changes the depth buffer, so lines are order-independent among
// themselves; what must be exact is WHICH occurrence of a repeated edge supplies the depths.  An open-addressed table keyed
// by the four screen integers keeps, per distinct edge, the smallest (face*3 + edge) id (atomicMin); only that occurrence
// draws.  One lane walks one line: the Bresenham state after k steps has a closed form (below), so the walk starts at the
// first on-screen step instead of spinning through off-screen pixels like the reference.
#include "b32_device.h"

namespace b32 {

__global__ void k_class_keys(const CovRec* __restrict__ recs, const uint32_t* __restrict__ order, const uint32_t* __restrict__ n_dev,
                             uint32_t n_cap, uint32_t* __restrict__ keys_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = min(*n_dev, n_cap);
    if (i < n) keys_out[i] = (recs[order[i]].flags & F_TRANSP) ? 1u : 0u;
}
void launch_class_keys(hipStream_t s, const CovRec* recs, const uint32_t* order, const uint32_t* n_dev, uint32_t n_cap, uint32_t* keys_out) {
    if (!n_cap) return;
    hipLaunchKernelGGL(k_class_keys, dim3((n_cap + 255) / 256), dim3(256), 0, s, recs, order, n_dev, n_cap, keys_out);
}

struct Edge { int32_t x0, y0, x1, y1; float z0, z1; };
// edge j of a wireframe triangle, direction-normalised (render.rs:2582-2587): keep (p,q) if (x0,y0) < (x1,y1) as tuples
__device__ __forceinline__ Edge wire_edge(const WireTri& t, int j) {
    const int a = j, b = j == 2 ? 0 : j + 1;
    Edge e = { t.x[a], t.y[a], t.x[b], t.y[b], t.z[a], t.z[b] };
    const bool keep = e.x0 < e.x1 || (e.x0 == e.x1 && e.y0 < e.y1);
    if (!keep) e = { t.x[b], t.y[b], t.x[a], t.y[a], t.z[b], t.z[a] };
    return e;
}
__device__ __forceinline__ uint32_t edge_hash(const Edge& e) {
    uint32_t h = (uint32_t)e.x0 * 0x9E3779B1u;
    h = (h ^ (h >> 15)) + (uint32_t)e.y0 * 0x85EBCA77u;
    h = (h ^ (h >> 13)) + (uint32_t)e.x1 * 0xC2B2AE3Du;
    h = (h ^ (h >> 16)) + (uint32_t)e.y1 * 0x27D4EB2Fu;
    return h ^ (h >> 15);
}
__device__ __forceinline__ bool same_edge(const Edge& a, const Edge& b) { return a.x0 == b.x0 && a.y0 == b.y0 && a.x1 == b.x1 && a.y1 == b.y1; }

constexpr uint32_t SLOT_EMPTY = 0xFFFFFFFFu;
__device__ __forceinline__ bool wire_global(const WireArgs& a, const Edge& e);     // (tile route: which edges are left to the global kernels)
// (one 128-byte line per counter, like the surface binning's: device-scope atomics on words of one line serialise at the memory side --
// packed counters: k_wire_bin 437 us on the 1 M-triangle scene, one line each: see profiles/r04_wire_*)
__device__ __forceinline__ uint32_t* wire_counter(const WireArgs& a, uint32_t tile) { return a.tile_fill + (size_t)tile * FILL_PAD; }
__device__ __forceinline__ uint32_t* wire_flag(const WireArgs& a, uint32_t which) { return a.tile_fill + (size_t)(a.tiles_x * a.tiles_y + which) * FILL_PAD; }   // 0 overflow, 1 big edges
__device__ __forceinline__ bool wire_global_idle(const WireArgs& a) {               // tile route on, no list overflow, no big edge
    if (!a.tile_fill) return false;
    return *wire_flag(a, 0) != a.epoch && *wire_flag(a, 1) != a.epoch;
}

// (The two flag words behind the tile counters -- a list overflowed, some edge is left to the global kernels -- hold the EPOCH of the frame
// that raised them (WireArgs::epoch, never 0): nobody has to zero them between frames, which was a memset launch of ~5 us on the main stream.)
// wait_ctrl != nullptr: this frame's k_wire_bin ran on the side stream (early binning); every workgroup of this kernel -- the first of the wire
// phases on the main stream -- looks once at Events::wbin_done before it reads a flag, and waits (bounded) if the binning has not published
// its epoch yet.  In place of a cross-stream event, whose wait costs the main stream ~6 us per frame even when the event fired long ago.
__global__ void k_wire_table_clear(uint32_t* __restrict__ owner, uint32_t* __restrict__ first, uint32_t n, const uint32_t* __restrict__ tile_flags, uint32_t epoch,
                                   Ctrl* __restrict__ wait_ctrl, uint32_t wait_epoch) {
    if (wait_ctrl) {
        if (threadIdx.x == 0) {
            const unsigned long long t0 = wall_clock64();
            while (__hip_atomic_fetch_add(&events_of(wait_ctrl)->wbin_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != wait_epoch) {
                if (wall_clock64() - t0 > 200000000ull) { atomicOr(&wait_ctrl->sticky, 8u); break; }
                __builtin_amdgcn_s_sleep(32);
            }
        }
        __syncthreads();
    }
    if (tile_flags && tile_flags[0] != epoch && tile_flags[FILL_PAD] != epoch) return;          // tile route: no overflow, no big edge -- the global kernels have nothing to do
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { owner[i] = SLOT_EMPTY; first[i] = SLOT_EMPTY; }
}

// Finds (or claims) the slot of edge `id`; the slot's key is the edge of its owner id, which is immutable input.
__device__ __forceinline__ uint32_t wire_slot(const WireArgs& a, const Edge& e, uint32_t id, bool insert) {
    uint32_t h = edge_hash(e) & a.table_mask;
    for (;;) {
        uint32_t cur = insert ? atomicCAS(&a.table_owner[h], SLOT_EMPTY, id) : a.table_owner[h];
        if (cur == SLOT_EMPTY) { if (insert) return h; return SLOT_EMPTY; }
        if (cur == id) return h;
        const Edge o = wire_edge(a.tris[cur / 3], (int)(cur % 3));
        if (same_edge(o, e)) return h;
        h = (h + 1) & a.table_mask;
    }
}

__global__ void k_wire_insert(WireArgs a) {
    phase_stamp(a.ctrl, ST_WIRE);
    if (wire_global_idle(a)) return;
    for (uint32_t id = blockIdx.x * blockDim.x + threadIdx.x; id < a.nf * 3; id += gridDim.x * blockDim.x) {     // (grid-stride: the launch is small when the tile route is on)
        const uint32_t f = id / 3;
        if (a.tris[f].kind != 1) continue;
        const Edge e = wire_edge(a.tris[f], (int)(id % 3));
        if (!wire_global(a, e)) continue;                             // (every occurrence of an edge decides alike: same coordinates)
        const uint32_t h = wire_slot(a, e, id, true);
        atomicMin(&a.table_first[h], id);
    }
}

// Bresenham of draw_line / draw_line_3d_impl (render.rs:716-750, 771-817) in closed form.  With adx = |x1-x0|, ady = |y1-y0|,
// after i x-steps and j y-steps the error term is err = adx*(1+j) - ady*(1+i); the x-step condition 2*err >= -ady and the
// y-step condition 2*err <= adx give, for an x-major line (adx >= ady): x steps every iteration and
//   j(k) = floor((2*ady*k + adx) / (2*adx))      (round half up),
// and symmetrically for a y-major line i(k) = floor((2*adx*k + ady) / (2*ady)).  The depth parameter `step` advances by
// exactly 1.0 per iteration (saturating at 2^24 in f32).  tests/test_oracle_kats.py checks this against the literal loop.
// The walk proper: every pixel of the line inside [cx0, cx1] x [cy0, cy1] (inclusive, already inside the frame and the band) that
// passes the depth test goes to plot(x, y).  The closed form lets the walk start at the first step inside the rectangle's major-axis
// range; the minor coordinate is tested per pixel.
// steps of the line whose major coordinate lies inside the rectangle: [k_lo, k_hi] (false: none)
// (I: long long for any line, int for the ones edge_narrow() admits -- same values)
template <typename I>
__device__ __forceinline__ bool line_k_range_t(const Edge& e, I cx0, I cx1, I cy0, I cy1, I& k_lo, I& k_hi) {
    const I dx = (I)e.x1 - (I)e.x0, dy = (I)e.y1 - (I)e.y0;
    const I adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
    const bool xmajor = adx >= ady;
    const I N = xmajor ? adx : ady, m0 = xmajor ? (I)e.x0 : (I)e.y0, lo = xmajor ? cx0 : cy0, hi = xmajor ? cx1 : cy1;
    const int sm = xmajor ? (e.x0 < e.x1 ? 1 : -1) : (e.y0 < e.y1 ? 1 : -1);
    k_lo = 0; k_hi = N;
    if (sm > 0) { if (lo - m0 > k_lo) k_lo = lo - m0; if (hi - m0 < k_hi) k_hi = hi - m0; }
    else        { if (m0 - hi > k_lo) k_lo = m0 - hi; if (m0 - lo < k_hi) k_hi = m0 - lo; }
    return k_lo <= k_hi;
}
__device__ __forceinline__ bool line_k_range(const Edge& e, long long cx0, long long cx1, long long cy0, long long cy1, long long& k_lo, long long& k_hi) {
    return line_k_range_t<long long>(e, cx0, cx1, cy0, cy1, k_lo, k_hi);
}
// For a line edge_narrow() admits: the steps whose PIXEL lies inside the rectangle, both coordinates.  The minor coordinate after k steps
// is n0 + sn * j(k), j(k) = floor((2 * dmin * k + dmaj) / (2 * dmaj)), which never decreases: j(k) >= J  <=>  k >= ceil(dmaj * (2J - 1) /
// (2 * dmin)), so the steps with j(k) in [Ja, Jb] are an interval again.  Everything stays below 2^30.
// tests/test_oracle_kats.py::test_wire_tile_clip_closed_form_equals_literal_loop restates these lines in Python integers and compares
// them with the literal loop of render.rs:771-817 on 30 000 random lines and rectangles; the GPU parity tests compare whole frames.
__device__ __forceinline__ bool line_k_range_exact(const Edge& e, int cx0, int cx1, int cy0, int cy1, int& k_lo, int& k_hi) {
    if (!line_k_range_t<int>(e, cx0, cx1, cy0, cy1, k_lo, k_hi)) return false;
    const int dx = e.x1 - e.x0, dy = e.y1 - e.y0, adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
    const bool xmajor = adx >= ady;
    const int dmaj = xmajor ? adx : ady, dmin = xmajor ? ady : adx;
    const int n0 = xmajor ? e.y0 : e.x0, nlo = xmajor ? cy0 : cx0, nhi = xmajor ? cy1 : cx1;
    const bool up = xmajor ? e.y0 < e.y1 : e.x0 < e.x1;                  // sn > 0
    int ja = up ? nlo - n0 : n0 - nhi, jb = up ? nhi - n0 : n0 - nlo;   // the minor steps that put the pixel inside: j in [ja, jb]
    if (jb < 0 || ja > dmin) return false;
    ja = max(ja, 0); jb = min(jb, dmin);
    if (dmin > 0) {                                                      // (dmin == 0: j stays 0, every step qualifies)
        const uint32_t d = 2u * (uint32_t)dmin;
        if (ja >= 1) k_lo = max(k_lo, (int)(((uint32_t)dmaj * (uint32_t)(2 * ja - 1) + d - 1u) / d));
        k_hi = min(k_hi, (int)(((uint32_t)dmaj * (uint32_t)(2 * jb + 1) + d - 1u) / d) - 1);
    }
    return k_lo <= k_hi;
}
// steps k_a ... k_b of the line (a sub-range of line_k_range's).  I = the integer type of the walk: every line with extents below 2^14
// and start coordinates below 2^20 -- anything a sane mesh produces -- fits 32 bits (2 * dmin * k + dmaj < 2^29); the rest (coordinates
// up to 2^31 after the saturating `as i32`) walks in 64 bits.  Same values either way.
template <typename I, typename Depth, typename Plot>
__device__ __forceinline__ void walk_line_range_t(const Edge& e, bool depth_test, I cx0, I cx1, I cy0, I cy1, I k_a, I k_b, Depth depth_at, Plot plot) {
    const I dx = (I)e.x1 - (I)e.x0, dy = (I)e.y1 - (I)e.y0;
    const I adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
    const I sx = e.x0 < e.x1 ? 1 : -1, sy = e.y0 < e.y1 ? 1 : -1;
    const I N = adx > ady ? adx : ady;
    const float total_steps = (float)(N > 1 ? N : 1);                   // dx.max((-dy).max(1)) as f32
    const bool xmajor = adx >= ady;
    const I m0 = xmajor ? (I)e.x0 : (I)e.y0;
    const I sm = xmajor ? sx : sy;
    const I dmin = xmajor ? ady : adx, dmaj = xmajor ? adx : ady;       // dmaj > 0 unless N == 0
    I j = 0, r = 0;                                                      // minor steps so far, remainder of the division
    if (dmaj > 0) { const I num = 2 * dmin * k_a + dmaj; j = num / (2 * dmaj); r = num - j * (2 * dmaj); }
    const I n0 = xmajor ? (I)e.y0 : (I)e.x0;
    const I sn = xmajor ? sy : sx;
    for (I k = k_a; k <= k_b; ++k) {
        const I maj = m0 + sm * k, mnr = n0 + sn * j;
        const I x = xmajor ? maj : mnr, y = xmajor ? mnr : maj;
        if (x >= cx0 && x <= cx1 && y >= cy0 && y <= cy1) {
            bool passes = true;
            if (depth_test) {
                const float step = (float)(k < (I)16777216 ? k : (I)16777216);
                const float t = step / total_steps;
                const float z = e.z0 + t * (e.z1 - e.z0);
                passes = z < depth_at((uint32_t)x, (uint32_t)y);
            }
            if (passes) plot((uint32_t)x, (uint32_t)y);
        }
        r += 2 * dmin;
        if (dmaj > 0 && r >= 2 * dmaj) { r -= 2 * dmaj; ++j; }
    }
}
template <typename Depth, typename Plot>
__device__ __forceinline__ void walk_line_range(const Edge& e, bool depth_test, long long cx0, long long cx1, long long cy0, long long cy1,
                                                long long k_a, long long k_b, Depth depth_at, Plot plot) {
    const long long adx = llabs((long long)e.x1 - e.x0), ady = llabs((long long)e.y1 - e.y0);
    const bool narrow = adx < 16384 && ady < 16384 && e.x0 > -1048576 && e.x0 < 1048576 && e.y0 > -1048576 && e.y0 < 1048576;
    if (narrow) walk_line_range_t<int>(e, depth_test, (int)cx0, (int)cx1, (int)cy0, (int)cy1, (int)k_a, (int)k_b, depth_at, plot);
    else walk_line_range_t<long long>(e, depth_test, cx0, cx1, cy0, cy1, k_a, k_b, depth_at, plot);
}
template <typename Depth, typename Plot>
__device__ __forceinline__ void walk_line(const Edge& e, bool depth_test, long long cx0, long long cx1, long long cy0, long long cy1, Depth depth_at, Plot plot) {
    long long k_lo, k_hi;
    if (line_k_range(e, cx0, cx1, cy0, cy1, k_lo, k_hi)) walk_line_range(e, depth_test, cx0, cx1, cy0, cy1, k_lo, k_hi, depth_at, plot);
}
__device__ __forceinline__ uint32_t abs_diff(int p, int q) { return p < q ? (uint32_t)q - (uint32_t)p : (uint32_t)p - (uint32_t)q; }   // |p - q| of two i32, exact (< 2^32)
__device__ __forceinline__ bool edge_overflows(const Edge& e) {          // 2*err overflows i32 in the reference (render.rs:735, 800)
    return abs_diff(e.x1, e.x0) >= (1u << 30) || abs_diff(e.y1, e.y0) >= (1u << 30);
}
// lines the tile kernel walks in 32-bit integers with the three-instruction depth parameter (the bounds of walk_line_range's `narrow`)
__device__ __forceinline__ bool edge_narrow(const Edge& e) {
    return abs_diff(e.x1, e.x0) < (uint32_t)WIRE_NARROW && abs_diff(e.y1, e.y0) < (uint32_t)WIRE_NARROW
        && e.x0 > -1048576 && e.x0 < 1048576 && e.y0 > -1048576 && e.y0 < 1048576;
}
__device__ void draw_line_dev(const WireArgs& a, const Edge& e, bool depth_test, uint32_t rgba) {
    if (edge_overflows(e)) { atomicOr(&a.ctrl->wire_overflow, 1u); atomicOr(&a.ctrl->sticky, 4u); return; }
    if (a.band_y1 <= a.band_y0 || !a.width) return;
    walk_line(e, depth_test, 0, (long long)a.width - 1, (long long)a.band_y0, (long long)a.band_y1 - 1,
              [&](uint32_t x, uint32_t y) { return a.zbuf ? a.zbuf[(size_t)y * a.width + x] : 3.40282347e+38f; },
              [&](uint32_t x, uint32_t y) { a.fb[(size_t)y * a.width + x] = rgba; });                     // set_pixel, render.rs:301-310
}

// ---------------------------------------------------------------- lines binned to screen tiles (B32_ROUTE_WIRE_TILES)
// One lane walking one whole line reads depths and writes colours at scattered addresses all over the frame, and the first-occurrence
// table is 3 M device-scope atomics (1 M faces: k_wire_insert 136 us + k_wire_draw 213 us).  Tile form: k_wire_bin appends every edge to
// the list of each 64 x WIRE_TH tile its clipped box touches; k_wire_tile (one workgroup per tile) finds the first occurrences in an LDS
// table -- every occurrence of an edge has the same coordinates, hence the same tiles, so the smallest id inside a tile is the smallest
// id overall -- and walks each surviving line's pixels inside the tile into an LDS bit plane; the tile's hit pixels are then written
// row by row.  Lines write one colour and never the depth buffer, so the order between lines does not matter (the front-face overlay,
// drawn after the back-face edges by the reference, wins where both hit).  An edge whose box covers more than WIRE_BIG_TILES tiles,
// and every edge of a frame in which some tile list overflowed, take the global kernels above instead (`wire_global`).
constexpr uint32_t WIRE_TABLE_SLOTS = 1024, WIRE_BIG_TILES = 64;      // (at most 768 edges of one tile in 1024 slots; usually a third of that)
static_assert(WIRE_TABLE_SLOTS >= 4 * WIRE_TILE_CAP, "LDS table load factor");
constexpr uint32_t COL_BACK = 80u | (80u << 8) | (100u << 16) | 0xFF000000u;      // Color::new(80, 80, 100), render.rs:2598
constexpr uint32_t COL_FRONT = 200u | (200u << 8) | (220u << 16) | 0xFF000000u;   // Color::new(200, 200, 220), render.rs:2628

struct WireBox { uint32_t tx0, tx1, ty0, ty1; bool visible, big; };
// tiles of the edge's box clipped to the frame and the band (every pixel the walk can touch lies inside the box)
__device__ __forceinline__ WireBox wire_box(const WireArgs& a, const Edge& e) {
    WireBox b = { 0, 0, 0, 0, false, false };
    // (comparisons of i32 coordinates with the frame's bounds, which are at most 16384: nothing here leaves 32 bits)
    const int xl = min(e.x0, e.x1), xh = max(e.x0, e.x1), yl = min(e.y0, e.y1), yh = max(e.y0, e.y1);
    const int cx0 = max(xl, 0), cx1 = min(xh, (int)a.width - 1);
    const int cy0 = max(yl, (int)a.band_y0), cy1 = min(yh, (int)a.band_y1 - 1);
    if (cx0 > cx1 || cy0 > cy1) return b;
    b.visible = true;
    b.tx0 = (uint32_t)cx0 >> 6; b.tx1 = (uint32_t)cx1 >> 6; b.ty0 = ((uint32_t)cy0 - a.tile_yb) / WIRE_TH; b.ty1 = ((uint32_t)cy1 - a.tile_yb) / WIRE_TH;
    b.big = (b.tx1 - b.tx0 + 1) * (b.ty1 - b.ty0 + 1) > WIRE_BIG_TILES;
    return b;
}
// does this edge go through the global kernels?  (tile route off: all of them)
__device__ __forceinline__ bool wire_global(const WireArgs& a, const Edge& e) {
    if (!a.tile_fill) return true;
    if (*wire_flag(a, 0) == a.epoch) return true;                   // a tile list overflowed (set by k_wire_bin, an earlier launch): the whole frame
    const WireBox b = wire_box(a, e);
    return b.visible && b.big;
}

// One lane per FACE: its (up to three) edges that take the tile route share one list entry per tile of the box around them -- a third of
// the reservations an entry per edge would need (a returning device-scope atomic each).  k_wire_tile expands the entry again and lets
// every edge take part only in the tiles of its OWN box, so all occurrences of an edge still meet in exactly the same tiles.
__global__ void k_wire_bin(WireArgs a, uint32_t kinds) {           // kinds: bit 0 back-face edges, bit 1 front-face overlay, bit 2: launched ahead of the fill (no phase stamp)
    if (!(kinds & 4u)) phase_stamp(a.ctrl, ST_WIRE);
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= a.nf || a.ctrl->abort) return;
    const WireTri t = a.tris[f];
    if (t.kind == 0 || !((kinds >> (t.kind - 1)) & 1u)) return;
    uint32_t tx0 = 0xFFFFFFFFu, tx1 = 0, ty0 = 0xFFFFFFFFu, ty1 = 0, n_big = 0;
    bool any = false, bad = false;
    WireBox bx[3]; bool on[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        on[j] = false;
        const Edge e = wire_edge(t, j);
        if (edge_overflows(e)) { bad = true; continue; }
        bx[j] = wire_box(a, e);
        if (!bx[j].visible) continue;
        if (bx[j].big) { ++n_big; continue; }
        any = on[j] = true;
        tx0 = min(tx0, bx[j].tx0); tx1 = max(tx1, bx[j].tx1); ty0 = min(ty0, bx[j].ty0); ty1 = max(ty1, bx[j].ty1);
    }
    if (bad) { atomicOr(&a.ctrl->wire_overflow, 1u); atomicOr(&a.ctrl->sticky, 4u); }
    if (n_big) (void)__hip_atomic_exchange(wire_flag(a, 1), a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!any) return;
    auto append = [&](uint32_t tx, uint32_t ty) {
        const uint32_t tile = ty * a.tiles_x + tx;
        const uint32_t pos = atomicAdd(wire_counter(a, tile), 1u);
        if (pos < WIRE_TILE_CAP) a.tile_lists[(size_t)tile * WIRE_TILE_CAP + pos] = f;
        else (void)__hip_atomic_exchange(wire_flag(a, 0), a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    if ((tx1 - tx0 + 1) * (ty1 - ty0 + 1) <= 2u * WIRE_BIG_TILES) {
        for (uint32_t ty = ty0; ty <= ty1; ++ty)
            for (uint32_t tx = tx0; tx <= tx1; ++tx) append(tx, ty);
    } else {
        // The box AROUND the edges of a face is not bounded by the limit on each edge's own box: a long thin L -- one edge along a tile
        // row, one along a tile column, the third one big and left to the global kernels -- has a union of up to WIRE_BIG_TILES^2 tiles,
        // in nearly all of which no edge of the face takes part (4096 serial reservations by one lane, and entries that crowd the 256-entry
        // lists into overflow).  Such a face is entered box by box instead: every tile of an edge's own box that no earlier edge's box
        // of the face holds already -- one entry per tile as before, and every edge still finds the face in every tile of its own box.
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (!on[j]) continue;
            for (uint32_t ty = bx[j].ty0; ty <= bx[j].ty1; ++ty)
                for (uint32_t tx = bx[j].tx0; tx <= bx[j].tx1; ++tx) {
                    bool dup = false;
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        if (k < j && on[k] && tx >= bx[k].tx0 && tx <= bx[k].tx1 && ty >= bx[k].ty0 && ty <= bx[k].ty1) dup = true;
                    if (!dup) append(tx, ty);
                }
        }
    }
}

// One 4-wave workgroup per tile, one lane per list entry (face).  The kernel is latency-bound -- a chain of dependent global loads
// (counter, list entry, face), LDS atomics that return, eight barriers -- and its time goes with 1 / (workgroups per CU) (measured by
// padding the LDS: 4 per CU 75.7 us, 3: 92.1, 2: 127.6 = 21 us + 213 us / n), so the LDS is kept small: 20.3 KB = seven workgroups per CU.
//   vxy / vz   the three screen vertices of every entry (9 KB; an edge is read back as two of them and direction-normalised again)
//   scratch    first the first-occurrence table -- owner[WIRE_TABLE_SLOTS]: hash slot -> the edge slot that claimed it, then
//              first[WIRE_EDGE_SLOTS]: claiming edge slot -> smallest global edge id among its occurrences (7 KB) -- later the
//              segment words (WIRE_SEG_CAP x 2 B) and the step range of every drawn edge inside the tile (3 KB)
//   zt         the tile's depths (the walk tests every pixel against them; read-only) (4 KB);  mask: two bit planes of hit pixels
// The walks are NOT done edge by edge -- a wave would last as long as its longest line times three -- but as SEGMENTS of at most
// WIRE_SEG steps, dealt out evenly: every surviving edge reports its step range inside the tile, a prefix sum over the edge slots
// numbers the segments, and lane t takes segments t, t + 256, ... (the closed form starts a walk at any step).
#ifndef B32_WIRE_SEG
#define B32_WIRE_SEG 16
#endif
constexpr uint32_t WIRE_EDGE_SLOTS = 3 * WIRE_TILE_CAP, WIRE_SEG = B32_WIRE_SEG, WIRE_SEGS_PER_EDGE = 64 / WIRE_SEG;    // (at most 64 steps of a line lie inside a tile)
constexpr uint32_t WIRE_THREADS = 256, WIRE_PX = 64 * WIRE_TH;
static_assert(WIRE_TILE_CAP == WIRE_THREADS && WIRE_EDGE_SLOTS <= 1024 && WIRE_SEGS_PER_EDGE <= 16, "k_wire_tile: one lane per entry; segment word = 10 bits of edge slot, 4 of segment number, narrow, kind");
#ifndef B32_WIRE_SEG_CAP
#define B32_WIRE_SEG_CAP 1536
#endif
constexpr uint32_t WIRE_SEG_CAP = B32_WIRE_SEG_CAP;              // segments walked per pass (a tile with more -- 768 edges of 64 steps have 3072 -- takes another pass)
static_assert(WIRE_SEG_CAP * sizeof(uint16_t) + WIRE_EDGE_SLOTS * sizeof(uint32_t) <= (WIRE_TABLE_SLOTS + WIRE_EDGE_SLOTS) * sizeof(uint32_t), "segment words and step ranges reuse the table's LDS");
// edge `slot` (= entry * 3 + j) back from the vertices in LDS, direction-normalised like wire_edge
__device__ __forceinline__ Edge wire_edge_lds(const int2* vxy, const float* vz, uint32_t slot, bool with_z) {
    const uint32_t j = slot % 3u, other = j == 2u ? slot - 2u : slot + 1u;
    const int2 p = vxy[slot], q = vxy[other];
    const float zp = with_z ? vz[slot] : 0.0f, zq = with_z ? vz[other] : 0.0f;
    const bool keep = p.x < q.x || (p.x == q.x && p.y < q.y);
    return keep ? Edge{ p.x, p.y, q.x, q.y, zp, zq } : Edge{ q.x, q.y, p.x, p.y, zq, zp };
}
__global__ __launch_bounds__(WIRE_THREADS) __attribute__((amdgpu_waves_per_eu(7, 7))) void k_wire_tile(WireArgs a) {     // (at most 72 VGPRs: seven waves per SIMD, like the LDS)
    __shared__ int2 vxy[WIRE_EDGE_SLOTS];                // 6 KB
    __shared__ float vz[WIRE_EDGE_SLOTS];                // 3 KB
    __shared__ __attribute__((aligned(16))) uint32_t scratch[WIRE_TABLE_SLOTS + WIRE_EDGE_SLOTS];   // 7 KB
    __shared__ float zt[WIRE_PX];                        // Framebuffer::zbuffer of the tile (4 KB)
    __shared__ uint32_t mask[2][WIRE_PX / 32];           // hit pixels: [0] back-face colour, [1] front-face overlay
    __shared__ uint32_t wsum[WIRE_THREADS / 64];
    uint32_t* owner = scratch; uint32_t* first = scratch + WIRE_TABLE_SLOTS;
    uint16_t* segs = reinterpret_cast<uint16_t*>(scratch);                               // [0, WIRE_SEG_CAP)
    uint32_t* krange = scratch + WIRE_SEG_CAP / 2;                                       // edge slot -> first step | (steps - 1) << 14
    const uint32_t tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t n_all = *wire_counter(a, tile);
    const bool overflowed = *wire_flag(a, 0) == a.epoch;
    __syncthreads();                                     // (everyone has read the counter)
    if (tid == 0) *wire_counter(a, tile) = 0;            // zero again for the next frame's k_wire_bin
    if (overflowed || n_all == 0 || a.ctrl->abort) return;
    const uint32_t n = n_all < WIRE_TILE_CAP ? n_all : WIRE_TILE_CAP;
    const uint32_t txi = tile % a.tiles_x, tyi = tile / a.tiles_x, x_lo = txi * 64u, y_top = a.tile_yb + tyi * WIRE_TH;
    for (uint32_t i = tid; i < WIRE_TABLE_SLOTS + WIRE_EDGE_SLOTS; i += WIRE_THREADS) scratch[i] = SLOT_EMPTY;
    if (tid < 2 * WIRE_PX / 32) (&mask[0][0])[tid] = 0;
    for (uint32_t p = tid; p < WIRE_PX; p += WIRE_THREADS) {
        const uint32_t x = x_lo + (p & 63u), y = y_top + (p >> 6);
        zt[p] = (a.zbuf && x < a.width && y >= a.band_y0 && y < a.band_y1) ? a.zbuf[(size_t)y * a.width + x] : 3.40282347e+38f;
    }
    Edge e[3]; uint32_t slot[3] = { 0, 0, 0 }; bool on[3] = { false, false, false }, narrow[3] = { false, false, false };
    uint32_t f = 0, kind = 0;
    if (tid < n) {
        f = a.tile_lists[(size_t)tile * WIRE_TILE_CAP + tid];
        const WireTri t = a.tris[f];
        kind = t.kind;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            vxy[tid * 3 + j] = make_int2(t.x[j], t.y[j]); vz[tid * 3 + j] = t.z[j];
            e[j] = wire_edge(t, j);
            if (edge_overflows(e[j])) continue;
            const WireBox b = wire_box(a, e[j]);
            // the edge takes part here iff this tile lies in its OWN box (so does every other occurrence of it)
            on[j] = b.visible && !b.big && txi >= b.tx0 && txi <= b.tx1 && tyi >= b.ty0 && tyi <= b.ty1;
            narrow[j] = edge_narrow(e[j]);
        }
    }
    __syncthreads();
#if defined(B32_EXP_WIRE_STAGE) && B32_EXP_WIRE_STAGE <= 1
    return;
#endif
    if (kind == 1) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (!on[j]) continue;
            const uint32_t i = tid * 3 + j;
            uint32_t h = edge_hash(e[j]) & (WIRE_TABLE_SLOTS - 1), claimer = i;
            for (;;) {
                const uint32_t cur = atomicCAS(&owner[h], SLOT_EMPTY, i);
                if (cur == SLOT_EMPTY || cur == i) break;
                if (same_edge(wire_edge_lds(vxy, vz, cur, false), e[j])) { claimer = cur; break; }
                h = (h + 1) & (WIRE_TABLE_SLOTS - 1);
            }
            atomicMin(&first[claimer], f * 3u + (uint32_t)j);
            slot[j] = claimer;
        }
    }
    __syncthreads();
#if defined(B32_EXP_WIRE_STAGE) && B32_EXP_WIRE_STAGE <= 2
    return;
#endif
    // which of my edges are drawn, over how many steps: segment counts
    // the tile's rectangle inside the frame and the band (non-empty: a tile of the grid; all four below 16384)
    const int cx0 = (int)x_lo, cx1 = (int)min(x_lo + 63u, a.width - 1u);
    const int cy0 = (int)max(y_top, a.band_y0), cy1 = (int)min(y_top + WIRE_TH - 1u, a.band_y1 - 1u);
    // steps of edge `ed` inside the rectangle: first step and count (0: none)
    auto steps_inside = [&](const Edge& ed, bool nrw, uint32_t& k_first) -> uint32_t {
        if (nrw) {
            int k_lo, k_hi;
            if (!line_k_range_exact(ed, cx0, cx1, cy0, cy1, k_lo, k_hi)) return 0u;
            k_first = (uint32_t)k_lo; return (uint32_t)(k_hi - k_lo + 1);
        }
        long long k_lo, k_hi;
        if (!line_k_range(ed, cx0, cx1, cy0, cy1, k_lo, k_hi)) return 0u;                   // (k_lo < 2^30, at most 64 steps inside a tile)
        k_first = (uint32_t)k_lo; return (uint32_t)(k_hi - k_lo + 1);
    };
    uint32_t nseg[3] = { 0, 0, 0 }, kr[3] = { 0, 0, 0 };
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (!on[j] || !kind) continue;
        if (kind == 1 && first[slot[j]] != f * 3u + (uint32_t)j) continue;               // an earlier occurrence of this edge draws it
        uint32_t k_first = 0;
        const uint32_t steps = steps_inside(e[j], narrow[j], k_first);
        nseg[j] = (steps + WIRE_SEG - 1) / WIRE_SEG;
        kr[j] = k_first | ((steps - 1u) << 14);                                          // (narrow lines: k_first < 2^14; the others count again)
    }
    __syncthreads();                                     // (everyone is done with the table: its LDS now holds the segment words)
    const uint32_t mine = nseg[0] + nseg[1] + nseg[2];
    uint32_t inc = mine;                                 // inclusive scan over the workgroup: wave scan, wave totals, offsets
    for (int off = 1; off < 64; off <<= 1) { const uint32_t v = __shfl_up(inc, off); if (lane >= (uint32_t)off) inc += v; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t base = inc - mine;
    for (uint32_t w = 0; w < wave; ++w) base += wsum[w];
    uint32_t total = 0;
    for (uint32_t w = 0; w < WIRE_THREADS / 64; ++w) total += wsum[w];
#pragma unroll
    for (int j = 0; j < 3; ++j) if (nseg[j] && narrow[j]) krange[tid * 3 + j] = kr[j];
    for (uint32_t pass0 = 0; pass0 < total; pass0 += WIRE_SEG_CAP) {                      // (one pass unless the tile has more than WIRE_SEG_CAP segments)
    if (pass0) __syncthreads();                          // (the previous pass has read its segment words)
    uint32_t pos = base;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const uint32_t i = tid * 3 + j, flags = (narrow[j] ? 1u << 14 : 0u) | (kind == 2 ? 1u << 15 : 0u);
        for (uint32_t q = 0; q < nseg[j]; ++q, ++pos)
            if (pos >= pass0 && pos < pass0 + WIRE_SEG_CAP) segs[pos - pass0] = (uint16_t)(i | (q << 10) | flags);     // (i < 768 < 2^10, q < 16)
    }
    __syncthreads();
#if defined(B32_EXP_WIRE_STAGE) && B32_EXP_WIRE_STAGE <= 3
    return;
#endif
    const uint32_t in_pass = min(total - pass0, WIRE_SEG_CAP);
    for (uint32_t t = tid; t < in_pass; t += WIRE_THREADS) {
        const uint32_t sg = segs[t], q = (sg >> 10) & 15u, which = sg >> 15;
        const bool nrw = (sg >> 14) & 1u;
        const Edge ed = wire_edge_lds(vxy, vz, sg & 1023u, true);
        uint32_t k_first = 0, steps;
        if (nrw) { const uint32_t w = krange[sg & 1023u]; k_first = w & 16383u; steps = (w >> 14) + 1u; }
        else steps = steps_inside(ed, false, k_first);                                       // (as counted above: > q * WIRE_SEG)
        const uint32_t s_a = q * WIRE_SEG, s_b = min((q + 1) * WIRE_SEG, steps) - 1u;       // steps of this segment, counted from k_first
        if (!nrw) {                                          // (coordinates beyond +-2^20 or extents of 2^14 and more: the general walk)
            walk_line_range_t<long long>(ed, which == 0, cx0, cx1, cy0, cy1, (long long)k_first + s_a, (long long)k_first + s_b,
                            [&](uint32_t x, uint32_t y) { return zt[(y - y_top) * 64u + (x - x_lo)]; },
                            [&](uint32_t x, uint32_t y) {
                                const uint32_t bit = (y - y_top) * 64u + (x - x_lo);
                                atomicOr(&mask[which][bit >> 5], 1u << (bit & 31u));
                            });
            continue;
        }
        // The same walk as walk_line_range_t for a narrow line, in tile coordinates and incrementally: every step of the range has its
        // pixel inside the rectangle (line_k_range_exact), so nothing is tested per pixel but the depth; `idx` is the pixel's index in
        // the tile's planes; the depth parameter k / N comes from wire_t_fast (k <= N < 2^14: no saturation at 2^24 to apply).
        const int dx = ed.x1 - ed.x0, dy = ed.y1 - ed.y0, adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
        const int sx = ed.x0 < ed.x1 ? 1 : -1, sy = ed.y0 < ed.y1 ? 1 : -1;
        const bool xmajor = adx >= ady;
        const uint32_t dmaj = (uint32_t)(xmajor ? adx : ady), dmin = (uint32_t)(xmajor ? ady : adx);
        const uint32_t k_a = k_first + s_a;
        uint32_t jm = 0, rr = 0;                             // minor steps before step k_a, remainder of that division
        if (dmaj) { const uint32_t num = 2u * dmin * k_a + dmaj; jm = num / (2u * dmaj); rr = num - jm * (2u * dmaj); }
        const uint32_t two_dmin = 2u * dmin, two_dmaj = dmaj ? 2u * dmaj : 0x7FFFFFFFu;          // (a point: never a minor step)
        const int px = ed.x0 + sx * (int)(xmajor ? k_a : jm) - (int)x_lo, py = ed.y0 + sy * (int)(xmajor ? jm : k_a) - (int)y_top;
        int idx = py * 64 + px;
        const int d_maj = xmajor ? sx : sy * 64, d_min = xmajor ? sy * 64 : sx;
        const float Nf = (float)(dmaj > 1u ? dmaj : 1u), rN = 1.0f / Nf, dz = ed.z1 - ed.z0;     // total_steps, render.rs:778
        float kf = (float)k_a;
        for (uint32_t n_left = s_b - s_a + 1u; n_left; --n_left) {
            bool pass = true;
            if (!which) {
                const float tt = wire_t_fast(kf, Nf, rN);
                const float z = ed.z0 + tt * dz;
                pass = z < zt[idx];
            }
            if (pass) atomicOr(&mask[which][(uint32_t)idx >> 5], 1u << ((uint32_t)idx & 31u));
            rr += two_dmin;
            const bool minor = rr >= two_dmaj;
            rr -= minor ? two_dmaj : 0u; idx += d_maj + (minor ? d_min : 0);
            kf += 1.0f;
        }
    }
    }
    __syncthreads();
    for (uint32_t p = tid; p < WIRE_PX; p += WIRE_THREADS) {
        const uint32_t w = p >> 5, b = 1u << (p & 31u);
        const bool fr = (mask[1][w] & b) != 0, bk = (mask[0][w] & b) != 0;
        if (fr | bk) a.fb[(size_t)(y_top + (p >> 6)) * a.width + x_lo + (p & 63u)] = fr ? COL_FRONT : COL_BACK;      // set_pixel, render.rs:301-310
    }
}

template <int KIND>
__global__ void k_wire_draw(WireArgs a) {
    phase_stamp(a.ctrl, ST_WIRE);
    if (a.ctrl->abort || wire_global_idle(a)) return;
    for (uint32_t id = blockIdx.x * blockDim.x + threadIdx.x; id < a.nf * 3; id += gridDim.x * blockDim.x) {
        const uint32_t f = id / 3;
        if (a.tris[f].kind != (uint32_t)KIND) continue;
        const Edge e = wire_edge(a.tris[f], (int)(id % 3));
        if (!wire_global(a, e)) continue;
        if (KIND == 1) {
            const uint32_t h = wire_slot(a, e, id, false);
            if (h == SLOT_EMPTY || a.table_first[h] != id) continue;    // a previous occurrence of this edge draws it
            draw_line_dev(a, e, true, COL_BACK);
        } else {
            draw_line_dev(a, e, false, COL_FRONT);
        }
    }
}

// The binning alone: it needs the wire list k_setup wrote and nothing of the fill, so a frame whose setup kernel runs on the side stream
// bins there too, beside the previous frame's fill and wire kernels (early; launch_wire then skips it).  An abort the fill decides later
// is harmless: k_wire_tile zeroes the counters again before it looks at the flag.
#ifndef B32_WIRE_IDLE_BLOCKS
#define B32_WIRE_IDLE_BLOCKS 256
#endif
void launch_wire_bin(hipStream_t s, const WireArgs& a, bool back, bool front, bool early) {
    if (!a.nf || !(back || front) || !a.tile_fill || !(a.tiles_x * a.tiles_y)) return;
    hipLaunchKernelGGL(k_wire_bin, dim3((a.nf + 255) / 256), dim3(256), 0, s, a, (back ? 1u : 0u) | (front ? 2u : 0u) | (early ? 4u : 0u));
}

void launch_wire(hipStream_t s, const WireArgs& a, bool back, bool front, bool binned, Ctrl* wait_ctrl, uint32_t wait_epoch) {
    if (!a.nf || !(back || front)) return;
    const uint32_t n = a.nf * 3, blocks = (n + 255) / 256;
    const uint32_t ntiles = a.tiles_x * a.tiles_y;
    // Order (found by the round-4 soak): the reference draws ALL back-face edges, then ALL overlay edges, and the overlay's colour wins where
    // both hit.  Back-face edges are order-independent among themselves (one colour, depths only read), so the ones left to the global
    // kernels go FIRST, then the tile kernel (its own bit planes give the overlay precedence inside a tile), then the overlay edges left to
    // the global kernel -- a big back-face edge drawn after the tile kernel would paint over a small overlay edge's pixels.
    const bool tiles = a.tile_fill && ntiles;
    const uint32_t* flags = tiles ? a.tile_fill + (size_t)ntiles * FILL_PAD : nullptr;
    // (tile route on: the global kernels usually have nothing to do -- 2048 workgroups that look at the flags and leave, or loop over
    // the edges left to them; tile route off: one lane per edge as before)
    // (256 workgroups: three launches that usually find nothing to do beside a busy setup kernel cost what their grids take to place)
    const uint32_t gblocks = flags ? min(blocks, (uint32_t)B32_WIRE_IDLE_BLOCKS) : blocks;
    if (tiles && !binned) launch_wire_bin(s, a, back, front, false);
    if (back) {
        hipLaunchKernelGGL(k_wire_table_clear, dim3(flags ? 256 : 1024), dim3(256), 0, s, a.table_owner, a.table_first, a.table_mask + 1, flags, a.epoch, wait_ctrl, wait_epoch);
        hipLaunchKernelGGL(k_wire_insert, dim3(gblocks), dim3(256), 0, s, a);
        hipLaunchKernelGGL(k_wire_draw<1>, dim3(gblocks), dim3(256), 0, s, a);
    }
#ifndef B32_EXP_WIRE_PAD_LDS                               // experiment: dynamic LDS nobody uses = fewer workgroups per CU
#define B32_EXP_WIRE_PAD_LDS 0
#endif
    if (tiles) hipLaunchKernelGGL(k_wire_tile, dim3(ntiles), dim3(WIRE_THREADS), B32_EXP_WIRE_PAD_LDS, s, a);
    if (front) hipLaunchKernelGGL(k_wire_draw<2>, dim3(gblocks), dim3(256), 0, s, a);
}

}  // namespace b32

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

__global__ void k_wire_table_clear(uint32_t* __restrict__ owner, uint32_t* __restrict__ first, uint32_t n) {
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
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= a.nf * 3) return;
    const uint32_t f = id / 3;
    if (a.tris[f].kind != 1) return;
    const Edge e = wire_edge(a.tris[f], (int)(id % 3));
    const uint32_t h = wire_slot(a, e, id, true);
    atomicMin(&a.table_first[h], id);
}

// Bresenham of draw_line / draw_line_3d_impl (render.rs:716-750, 771-817) in closed form.  With adx = |x1-x0|, ady = |y1-y0|,
// after i x-steps and j y-steps the error term is err = adx*(1+j) - ady*(1+i); the x-step condition 2*err >= -ady and the
// y-step condition 2*err <= adx give, for an x-major line (adx >= ady): x steps every iteration and
//   j(k) = floor((2*ady*k + adx) / (2*adx))      (round half up),
// and symmetrically for a y-major line i(k) = floor((2*adx*k + ady) / (2*ady)).  The depth parameter `step` advances by
// exactly 1.0 per iteration (saturating at 2^24 in f32).  tests/test_oracle_kats.py checks this against the literal loop.
__device__ void draw_line_dev(const WireArgs& a, const Edge& e, bool depth_test, uint32_t rgba) {
    const long long adx = llabs((long long)e.x1 - e.x0), ady = llabs((long long)e.y1 - e.y0);
    if (adx >= (1ll << 30) || ady >= (1ll << 30)) { atomicOr(&a.ctrl->wire_overflow, 1u); atomicOr(&a.ctrl->sticky, 4u); return; }   // 2*err overflows i32 in the reference
    const int sx = e.x0 < e.x1 ? 1 : -1, sy = e.y0 < e.y1 ? 1 : -1;
    const long long N = adx > ady ? adx : ady;
    const float total_steps = (float)(N > 1 ? N : 1);                   // dx.max((-dy).max(1)) as f32
    const bool xmajor = adx >= ady;
    const long long m0 = xmajor ? e.x0 : e.y0, wmaj = xmajor ? a.width : a.height;
    const int sm = xmajor ? sx : sy;
    long long k_lo = 0, k_hi = N;
    if (sm > 0) { if (-m0 > k_lo) k_lo = -m0; if (wmaj - 1 - m0 < k_hi) k_hi = wmaj - 1 - m0; }
    else        { if (m0 - (wmaj - 1) > k_lo) k_lo = m0 - (wmaj - 1); if (m0 < k_hi) k_hi = m0; }
    if (k_lo > k_hi) return;
    const long long dmin = xmajor ? ady : adx, dmaj = xmajor ? adx : ady;       // dmaj > 0 unless N == 0
    long long j = 0, r = 0;                                                         // minor steps so far, remainder of the division
    if (dmaj > 0) { const long long num = 2 * dmin * k_lo + dmaj; j = num / (2 * dmaj); r = num % (2 * dmaj); }
    const long long n0 = xmajor ? e.y0 : e.x0;
    const int sn = xmajor ? sy : sx;
    for (long long k = k_lo; k <= k_hi; ++k) {
        const long long maj = m0 + sm * k, mnr = n0 + sn * j;
        const long long x = xmajor ? maj : mnr, y = xmajor ? mnr : maj;
        if (x >= 0 && x < (long long)a.width && y >= (long long)a.band_y0 && y < (long long)a.band_y1) {
            bool passes = true;
            if (depth_test) {
                const float step = (float)(k < 16777216 ? k : 16777216);
                const float t = step / total_steps;
                const float z = e.z0 + t * (e.z1 - e.z0);
                const float zb = a.zbuf ? a.zbuf[(size_t)y * a.width + (size_t)x] : 3.40282347e+38f;
                passes = z < zb;
            }
            if (passes) a.fb[(size_t)y * a.width + (size_t)x] = rgba;                // set_pixel, render.rs:301-310
        }
        r += 2 * dmin;
        if (dmaj > 0 && r >= 2 * dmaj) { r -= 2 * dmaj; ++j; }
    }
}

template <int KIND>
__global__ void k_wire_draw(WireArgs a) {
    phase_stamp(a.ctrl, ST_WIRE);
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= a.nf * 3 || a.ctrl->abort) return;
    const uint32_t f = id / 3;
    if (a.tris[f].kind != (uint32_t)KIND) return;
    const Edge e = wire_edge(a.tris[f], (int)(id % 3));
    if (KIND == 1) {
        const uint32_t h = wire_slot(a, e, id, false);
        if (h == SLOT_EMPTY || a.table_first[h] != id) return;          // a previous occurrence of this edge draws it
        draw_line_dev(a, e, true, 80u | (80u << 8) | (100u << 16) | 0xFF000000u);     // Color::new(80, 80, 100), render.rs:2598
    } else {
        draw_line_dev(a, e, false, 200u | (200u << 8) | (220u << 16) | 0xFF000000u);  // Color::new(200, 200, 220), render.rs:2628
    }
}

void launch_wire(hipStream_t s, const WireArgs& a, bool back, bool front) {
    if (!a.nf) return;
    const uint32_t n = a.nf * 3, blocks = (n + 255) / 256;
    if (back) {
        hipLaunchKernelGGL(k_wire_table_clear, dim3(1024), dim3(256), 0, s, a.table_owner, a.table_first, a.table_mask + 1);
        hipLaunchKernelGGL(k_wire_insert, dim3(blocks), dim3(256), 0, s, a);
        hipLaunchKernelGGL(k_wire_draw<1>, dim3(blocks), dim3(256), 0, s, a);
    }
    if (front) hipLaunchKernelGGL(k_wire_draw<2>, dim3(blocks), dim3(256), 0, s, a);
}

}  // namespace b32

// b32_frame.hip -- the C ABI, part 3: one frame.  Route selection, enqueue (k_setup -> fill [-> k_blend] [-> wireframe kernels]) without a
// host round trip, two or three frames in flight, and b32_frame_finish -- the only host readback (error flags, triangles_drawn, fragment
// count, overflow -> grow and redraw).
#include "b32_host.h"

// ------------------------------------------------------------------ two frames in flight
// The fused fill kernel leaves most CUs idle in its last fifth (the tile queue's tail), and k_setup of the NEXT frame needs nothing
// from it: with two frame sets (everything k_setup writes, FrameSet) the setup kernel of frame i + 1 runs on a second stream beside the
// fill of frame i.  Orders kept by events: setup(i) -> fill(i) (ev_setup; or the k_flag / k_join kernels), fill(i) -> setup(i + 2) on the same set
// (round 6: on the device, k_gate waits for fill(i + 1) to have STARTED; an event only where that frame launched no fused kernel), and
// anything enqueued on the main stream that k_setup reads (uploads, packed streams, light lists, list-space memsets) -> the next
// setup (ev_main, only when `side_dirty`).  The main stream always waits for the frame's setup before enqueue_frame returns, so a
// synchronisation of the main stream still covers everything this context has in flight.
#ifndef B32_JOIN_NOT_BATCHED
#define B32_JOIN_NOT_BATCHED 1     // (the merged runs of a batched frame keep the event: their kernels are so short that the two extra launches cost what the
                                   // barrier packet did -- console frame 0.268-0.275 against 0.264-0.265 ms)
#endif
#ifndef B32_JOIN_KERNEL
#define B32_JOIN_KERNEL 1            // (0: the fill waits for its setup kernel through a cross-stream event, as before)
#endif
#ifndef B32_POLL_BATCHED
#define B32_POLL_BATCHED 1            // (0: the merged draws of a batched frame wait for their setup kernels through a cross-stream event, as before)
#endif
#ifndef B32_START_AT_BLEND
#define B32_START_AT_BLEND 1         // (0: the fused kernel always publishes the frame's "started" word itself)
#endif
#ifndef B32_START_GATE
#define B32_START_GATE 1             // (0: a cross-stream event behind every fill orders it before the setup kernel that next writes its frame set, as before)
#endif
#ifndef B32_PIPE_FEW_TILES
#define B32_PIPE_FEW_TILES 1          // (0: frames whose fused kernel has no more tiles than workgroup slots are not pipelined, as before round 6)
#endif
#ifndef B32_WIRE_BIN_EARLY
#define B32_WIRE_BIN_EARLY 1          // (0: k_wire_bin behind the fill on the main stream, as before)
#endif
#ifndef B32_EXP_NO_WIRE_PIPE
#define B32_EXP_NO_WIRE_PIPE 0          // (1: frames with wireframe phases are not pipelined, as before round 5 -- their wire list was not part of the frame set)
#endif
static void swap_with(b32_ctx* c, FrameSet& a) {
    std::swap(c->keys[0], a.keys0); std::swap(c->crecs, a.crecs); std::swap(c->srecs, a.srecs); std::swap(c->xrecs, a.xrecs);
    std::swap(c->spans, a.spans); std::swap(c->face_of, a.face_of); std::swap(c->partials, a.partials);
    std::swap(c->shades, a.shades); std::swap(c->cap_shades, a.cap_shades);
    std::swap(c->direct_lists, a.direct_lists); std::swap(c->cap_direct, a.cap_direct);
    std::swap(c->tile_fill, a.tile_fill); std::swap(c->cap_tile_fill, a.cap_tile_fill);
    std::swap(c->wire, a.wire); std::swap(c->cap_wire, a.cap_wire);
    std::swap(c->wire_fill, a.wire_fill); std::swap(c->wire_lists, a.wire_lists); std::swap(c->cap_wire_tiles, a.cap_wire_tiles); std::swap(c->wire_grid, a.wire_grid);
    std::swap(c->d_ctrl, a.d_ctrl);
    std::swap(c->ev_setup, a.ev_setup); std::swap(c->ev_done, a.ev_done); std::swap(c->set_in_flight, a.in_flight);
}
// The frame being enqueued takes the OLDEST set; afterwards alt[n_sets - 2] is the previous frame's set and alt[0] the set of the frame
// n_sets - 1 back -- the one whose fill the new frame's setup kernel is meant to run beside (its tile cursor is what the gate polls).
static void rotate_sets(b32_ctx* c) {
    swap_with(c, c->alt[0]);                                   // current <- oldest; alt[0] <- previous frame's
    if (c->n_sets == 3) std::swap(c->alt[0], c->alt[1]);        // alt[0] <- two frames back, alt[1] <- previous frame's
}
static void unrotate_sets(b32_ctx* c) {      // (an enqueue that failed between rotate_sets and its launches)
    if (c->n_sets == 3) std::swap(c->alt[0], c->alt[1]);
    swap_with(c, c->alt[0]);
}
extern "C" void free_alt(b32_ctx* c, FrameSet& a) {          // (the caller has drained both streams)
    void* ptrs[] = { a.keys0, a.crecs, a.srecs, a.xrecs, a.spans, a.face_of, a.partials, a.shades, a.direct_lists, a.tile_fill, a.wire, a.wire_fill, a.wire_lists };
    for (void* q : ptrs) if (q) (void)hipFree(q);
    a.keys0 = nullptr; a.crecs = nullptr; a.srecs = nullptr; a.xrecs = nullptr; a.spans = nullptr; a.face_of = nullptr; a.partials = nullptr;
    a.shades = nullptr; a.cap_shades = 0; a.direct_lists = nullptr; a.cap_direct = 0; a.tile_fill = nullptr; a.cap_tile_fill = 0; a.cap_work = 0;
    a.wire = nullptr; a.cap_wire = 0; a.wire_fill = nullptr; a.wire_lists = nullptr; a.cap_wire_tiles = 0; a.wire_grid = 0;
}
// side stream, events and the other sets' per-face buffers (sized like the current set's)
static int pipeline_ensure(b32_ctx* c) {
    if (!c->side) {
        // lowest priority: while the fill kernel has workgroups to place, they go first; the setup kernel takes what is left
        int prio_least = 0, prio_greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
        HIPCHK(c, hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, prio_least));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_main, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_wbin, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_setup, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
        for (FrameSet& a : c->alt) {
            HIPCHK(c, hipEventCreateWithFlags(&a.ev_setup, hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&a.ev_done, hipEventDisableTiming));
        }
        // the frames enqueued on the current set before the side stream existed recorded nothing: their fills end before this point of
        // the main stream, which the first setup kernel on the side stream waits for (side_dirty) and the set's own event now marks too
        HIPCHK(c, hipEventRecord(c->ev_done, c->stream));
        c->side_dirty = true;
    }
    for (uint32_t k = 0; k + 1 < c->n_sets; ++k) {
        FrameSet& a = c->alt[k];
        if (!a.d_ctrl) {
            HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&a.d_ctrl), sizeof(Ctrl) + sizeof(Stamps) + sizeof(Events)));
            HIPCHK(c, hipMemsetAsync(a.d_ctrl, 0, sizeof(Ctrl) + sizeof(Stamps) + sizeof(Events), c->stream));
            c->side_dirty = true;
        }
        if (a.cap_work < c->cap_work || !a.crecs) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            free_alt(c, a);
            const size_t n = c->cap_work;
            int rc;
            if ((rc = ensure_plain(c, a.keys0, n))) return rc;
            if ((rc = ensure_plain(c, a.crecs, n))) return rc;
            if ((rc = ensure_plain(c, a.srecs, n))) return rc;
            if ((rc = ensure_plain(c, a.xrecs, n))) return rc;
            if ((rc = ensure_plain(c, a.spans, n))) return rc;
            if ((rc = ensure_plain(c, a.face_of, n))) return rc;
            if ((rc = ensure_plain(c, a.partials, (size_t)((n + 255) / 256) * 8 + 8))) return rc;
            a.cap_work = n;
        }
    }
    return B32_OK;
}

extern "C" {
// ------------------------------------------------------------------ frame
int validate_settings(const B32Settings* st) {
    if (st->shading > B32_SHADE_GOURAUD) return B32_E_ARG;
    if (st->n_lights && !st->lights) return B32_E_ARG;
    if (st->shading != B32_SHADE_NONE)
        for (uint32_t i = 0; i < st->n_lights; ++i)
            if (st->lights[i].enabled && st->lights[i].type > B32_LIGHT_SPOT) return B32_E_ARG;                 // not a LightType
    return B32_OK;
}

static uint32_t bits_for(uint32_t n_keys) { uint32_t b = 1; while ((1ull << b) < n_keys) ++b; return b; }

// ------------------------------------------------------------------ one frame = the pieces below, in order (enqueue_frame)
// Which pipeline a frame takes.  Every route produces the same framebuffer (tests run every scene through several of them).
struct Route {
    bool with_class = false;     // the scene can have a transparent pass: tile lists are split by class
    bool ordered_all = false;    // ordered walk of whole tile lists instead of the overwrite pass (x-ray; 8-bit path with blending texels)
    bool want_prio64 = false;    // sort-free fused path (painter's or z-buffer mode)
    bool exact_cov = false;      // EXACT coverage: texel rule per fragment (exact store counting, textures with many skippable texels)
    bool local_sort = false;     // keyed fast path: per-tile LDS sort instead of the global painter's sort
    bool want_inline = false;    // small mesh: the fused kernel's workgroups collect their own tile lists
    bool direct_bin = false;     // large mesh: k_setup appends to fixed tile regions (DirectBin)
    bool inline_bin = false, prio64 = false;     // what was finally launched
    uint32_t list_stride = 0;    // entries per tile region (inline / direct binning)
    DirectBin db{};
};

static FrameParams frame_params(const b32_ctx* c, const B32Camera* cam, const B32Settings* st, const B32Fog* fog, bool wire_any) {
    FrameParams fp{};
    fp.cam = *cam;
    fp.width = c->width; fp.height = c->height;
    fp.band_y0 = c->band_y0; fp.band_y1 = c->band_y1;
    fp.tiles_x = (c->width + TILE_W - 1) / TILE_W;
    fp.tile_h = TILE_H;
    fp.tile_yb = (c->band_y0 / TILE_H) * TILE_H;
    fp.tiles_y = c->band_y1 > c->band_y0 ? (c->band_y1 - fp.tile_yb + TILE_H - 1) / TILE_H : 0;
    fp.nv = c->nv; fp.nf = c->nf; fp.nt = c->nt;
    fp.n_lights = st->shading != B32_SHADE_NONE ? st->n_lights : 0;
    fp.ambient = st->ambient;
    fp.affine = st->affine_textures; fp.shading = st->shading; fp.backface_cull = st->backface_cull;
    fp.dithering = st->dithering; fp.fixed_point = st->use_fixed_point; fp.has_fog = fog ? 1 : 0; fp.zmode = st->use_zbuffer ? 1 : 0;
    if (fog) fp.fog = *fog;
    fp.camfx = make_camfx_any(*cam, c->width, c->height);
    fp.fmt8 = c->fmt8 ? 1 : 0;
    fp.ortho = st->has_ortho ? 1 : 0; fp.xray = st->xray_mode ? 1 : 0;
    fp.ortho_zoom = st->ortho_zoom; fp.ortho_cx = st->ortho_center_x; fp.ortho_cy = st->ortho_center_y;
    fp.wire_collect = wire_any ? 1 : 0;
    fp.band_only = 0;
    fp.redraw = c->redrawing ? 1 : 0;
    fp.tex_blend_any = c->tex_blend_any ? 1 : 0;
    fp.batched = c->frame_batched ? 1 : 0;
    return fp;
}

// lights: up to LIGHTS_INLINE travel by value in k_setup's arguments (a light change costs no copy and no synchronisation: the
// per-room light lists of a multi-mesh frame stay asynchronous); longer lists go through a device buffer, refreshed -- with a
// synchronisation -- only when they differ from the copy it holds
static int frame_lights(b32_ctx* c, const B32Settings* st, FrameParams& fp, LightSet& lset) {
    int rc;
    if (fp.n_lights && fp.n_lights <= LIGHTS_INLINE) {
        memcpy(lset.l, st->lights, fp.n_lights * sizeof(B32Light));
        fp.lights_inline = 1;
    } else if (fp.n_lights) {
        bool same = c->h_lights.size() == fp.n_lights && memcmp(c->h_lights.data(), st->lights, fp.n_lights * sizeof(B32Light)) == 0;
        if (!same) {
            if ((rc = ensure(c, c->d_lights, c->cap_lights, (size_t)fp.n_lights))) return rc;
            HIPCHK(c, hipStreamSynchronize(c->stream));
            HIPCHK(c, hipMemcpy(c->d_lights, st->lights, fp.n_lights * sizeof(B32Light), hipMemcpyHostToDevice));
            c->side_dirty = true;
            c->h_lights.assign(st->lights, st->lights + fp.n_lights);
        }
    }
    if (fp.shading != B32_SHADE_NONE && (!c->shades || c->cap_shades < c->cap_work)) {
        if (c->shades) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(c->shades)); c->shades = nullptr; }
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->shades), c->cap_work * 9 * sizeof(float)));
        c->cap_shades = c->cap_work;
    }
    return B32_OK;
}

// work buffers every route may need, sized for the uncut 64x64 tile grid (the sort-free path may cut tiles to a quarter of the height:
// 4 x as many list ranges)
static int frame_buffers(b32_ctx* c, const FrameParams& fp, bool wire_back) {
    hipStream_t s = c->stream;
    const uint32_t ntiles = fp.tiles_x * fp.tiles_y;
    int rc;
    // pair buffers: start at 2 pairs per face + one per tile; b32_frame_finish grows them on overflow
    if (c->cap_pairs == 0 || !c->pkeys[0]) {
        const size_t n = (size_t)c->nf * 2 + ntiles + 1024;
        for (int i = 0; i < 2; ++i) { if ((rc = ensure_plain(c, c->pkeys[i], n))) return rc; if ((rc = ensure_plain(c, c->pvals[i], n))) return rc; }
        c->cap_pairs = n;
    }
    const uint32_t need_blocks = std::max((uint32_t)((std::max(c->cap_pairs, c->cap_work) + SORT_TILE - 1) / SORT_TILE) + 1,
                                          (uint32_t)(c->cap_work / 1024 + 2));          // span counting sort: >= 1024 faces per block
    if (need_blocks > c->hist_blocks || !c->block_hist) {
        if ((rc = ensure_plain(c, c->block_hist, (size_t)4096 * need_blocks))) return rc;
        c->hist_blocks = need_blocks;
    }
    const size_t need_ranges = (size_t)4 * ntiles + 4 * fp.tiles_x + 2;      // list ranges: 2 per tile (tile, class), x 4 for cut tiles
    if (need_ranges > c->cap_ranges || !c->ranges) {
        if ((rc = ensure_plain(c, c->ranges, need_ranges + 64))) return rc;
        c->cap_ranges = need_ranges + 64;
    }
    if (need_ranges > c->cap_tile_mid || !c->tile_mid) {
        if ((rc = ensure_plain(c, c->tile_mid, need_ranges + 64))) return rc;
        c->cap_tile_mid = need_ranges + 64;
    }
    if (c->mask_dirty && c->pool_texels) {      // (after the drop-in call's staged copy kernel on the same stream: the texels are there)
        launch_build_mask(s, c->fmt8 ? nullptr : c->d_texels, c->fmt8 ? c->d_texels32 : nullptr, c->pool_texels, c->d_texmask);
        c->mask_dirty = false;
    }
    if (fp.zmode) {          // Framebuffer::zbuffer (render.rs:12): allocated on first use, f32::MAX until drawn into
        const size_t px = (size_t)c->width * c->height;
        if (px > c->cap_zbuf || !c->zbuf) { if ((rc = ensure_plain(c, c->zbuf, px + 64))) return rc; c->cap_zbuf = px; c->zbuf_valid = false; }
        if (!c->zbuf_valid) { launch_clear(s, reinterpret_cast<uint32_t*>(c->zbuf), px, 0x7F7FFFFFu); c->zbuf_valid = true; }
    }
    if ((size_t)c->width * c->height > c->cap_vis || !c->vis) {
        if ((rc = ensure_plain(c, c->vis, (size_t)c->width * c->height * 2 + 64))) return rc;    // two words per pixel (prio64 coverage)
        c->cap_vis = (size_t)c->width * c->height;
    }
    if (fp.wire_collect && c->nf) {
        if ((size_t)c->nf > c->cap_wire || !c->wire) { if ((rc = ensure_plain(c, c->wire, (size_t)c->nf + 16))) return rc; c->cap_wire = c->nf; }
        size_t slots = 1024;
        while (slots < (size_t)c->nf * 6) slots <<= 1;                        // load factor <= 0.5 with all 3*nf edges distinct
        if (wire_back && (slots > c->cap_wire_table || !c->wire_owner)) {
            if ((rc = ensure_plain(c, c->wire_owner, slots))) return rc;
            if ((rc = ensure_plain(c, c->wire_first, slots))) return rc;
            c->cap_wire_table = slots;
        }
        // tile route: one counter and one list region per 64 x 16 wire tile (WIRE_TH rows) of the band (+ the overflow flag and the big-edge count)
        if (!(c->route_off & B32_ROUTE_WIRE_TILES) && c->band_y1 > c->band_y0) {
            const size_t wt = (size_t)fp.tiles_x * ((c->band_y1 - (c->band_y0 / WIRE_TH) * WIRE_TH + WIRE_TH - 1) / WIRE_TH);
            if (wt > c->cap_wire_tiles || !c->wire_fill) {
                if ((rc = ensure_plain(c, c->wire_fill, (wt + 2) * FILL_PAD + 64))) return rc;
                if ((rc = ensure_plain(c, c->wire_lists, wt * WIRE_TILE_CAP + 64))) return rc;
                c->cap_wire_tiles = wt; c->wire_grid = 0;
            }
            // (the counters are zero between frames: k_wire_tile re-zeroes what k_wire_bin counted; a new allocation or another tile grid
            // -- resize, band change -- starts from a cleared array)
            const unsigned long long grid = ((unsigned long long)c->width << 40) ^ ((unsigned long long)c->band_y0 << 20) ^ c->band_y1;
            if (grid != c->wire_grid) {
                HIPCHK(c, hipMemsetAsync(c->wire_fill, 0, ((c->cap_wire_tiles + 2) * FILL_PAD + 64) * sizeof(uint32_t), s));
                c->wire_grid = grid; c->side_dirty = true;          // (a pipelined frame bins on the side stream: behind this memset)
            }
        }
    }
    return B32_OK;
}

// Route selection.  May cut the tile grid (fp.tile_h / tile_yb / tiles_y) and prepares the direct binning's regions.
static int plan_route(b32_ctx* c, FrameParams& fp, const SortScratch& sc, bool wire_front, Route& r) {
    int rc;
    // z-buffer frames without a transparent pass take the sort-free fused path too (depth is the priority); otherwise z-buffer
    // mode applies depth + skip rule per fragment (EXACT coverage)
    // (a transparent pass rides along: its entries are split off at binning time and sorted per tile by k_blend)
    r.with_class = c->may_blend && !c->fmt8;
    const bool spans_ok = !(c->route_off & B32_ROUTE_SORT_FREE) && c->local_sort_ok && bin_spans_applicable(fp, sc, r.with_class);
    r.ordered_all = c->fmt8 ? c->blend8 : (fp.xray != 0);
    // sort-free path: painter's or z-buffer mode (orthographic keys use all 32 bits -> class pass -> general path)
    r.want_prio64 = spans_ok && !fp.ortho && !r.ordered_all;
    // too few 64x64 tiles to fill the GPU (narrow multi-GPU band, PS1-sized frame): tiles of 32 or 16 rows multiply the parallelism
    // of the fused kernel.  Only the sort-free path knows about them (its k_blend included); the keyed kernels keep 64 rows.
    if (r.want_prio64 && c->band_y1 > c->band_y0 && !(c->route_off & B32_ROUTE_CUT_TILES)) {
        uint32_t th = TILE_H;
        // 64 -> 32 rows below two tiles per CU, 32 -> 16 -> 8 rows below one tile per CU (measured: a 240-row band of C3 prefers 320 tiles
        // of 32 rows to 600 of 16; C2's 20 tiles prefer 150 of 8 rows -- 0.039 ms against 0.051 with 75 of 16 rows, 0.049 with 300 of 4)
        while (th > (uint32_t)B32_MIN_TILE_H && fp.tiles_x * ((c->band_y1 + th - 1) / th - c->band_y0 / th) < (th == TILE_H ? 2u : 1u) * (uint32_t)c->n_cu &&
               (c->band_y1 - c->band_y0) / (th / 2) + 2 <= 255 /* tile rows must fit the packed spans */) th /= 2;
#ifdef B32_EXP_FORCE_TH
        th = B32_EXP_FORCE_TH;
#endif
        fp.tile_h = th;
        fp.tile_yb = (c->band_y0 / th) * th;
        fp.tiles_y = (c->band_y1 - fp.tile_yb + th - 1) / th;
    }
    const uint32_t ntiles = fp.tiles_x * fp.tiles_y;
    // EXACT coverage = texel rule per fragment: exact store counting, textures with many skippable texels; the keyed z-buffer kernel
    // is EXACT by construction
    r.exact_cov = c->count_fragments || !c->cheap_ok || (fp.zmode && !r.want_prio64);
    // the sorted fast path reads the class from bit 31 of the depth key and has no ordered opaque walk: not for ortho / x-ray frames
    r.local_sort = !r.exact_cov && c->local_sort_ok && !fp.ortho && !r.ordered_all && !fp.zmode;
    fp.band_only = (r.want_prio64 && c->band_set) ? 1 : 0;   // other ranks own the other rows: their surfaces' records are never read here
    // small mesh (what the reference's callers submit per room / asset part): no binning launch, the
    // fused kernel's workgroups collect their own tile lists from the spans (needs one list region of nf entries per tile)
    // (with a transparent pass only up to 2048 faces: no tile's transparent list can then exceed what k_blend sorts in LDS)
    r.list_stride = (c->nf + 31u) & ~31u;
    r.want_inline = r.want_prio64 && !wire_front && c->nf <= (r.with_class ? 2048u : 8192u) && (size_t)ntiles * r.list_stride <= ((size_t)4 << 20) &&
                    !(c->route_off & B32_ROUTE_INLINE_BIN);
    // larger meshes: no binning launch either -- k_setup appends every surviving face to fixed-size tile regions (DirectBin)
    if (c->direct_ntiles != ntiles) { c->direct_ntiles = ntiles; c->direct_cap_opaque = 0; c->direct_ok = true; }   // another tile grid (resize, band)
    if (r.want_prio64 && !r.want_inline && !wire_front && c->direct_ok && !(c->route_off & B32_ROUTE_DIRECT_BIN) && ntiles) {
        // first guess: three times the mean list of a mesh whose every face is drawn and touches one tile; a frame that overflows
        // reports its longest list and is redrawn with regions a quarter above it (b32_frame_finish)
        if (!c->direct_cap_opaque) c->direct_cap_opaque = std::max<uint32_t>(512u, (uint32_t)std::min<uint64_t>((uint64_t)3 * c->nf / ntiles + 64, 1u << 24));
        // a mesh of moderate size gets regions that hold ALL its faces (at most 32 MB of list space): such a frame can never overflow a
        // region, needs no redraw, and may stay in flight across scene swaps and further frames like a small mesh's
        if (c->nf <= 65536u && (uint64_t)ntiles * (c->nf + (r.with_class ? BLEND_SORT_CAP : 0u)) <= (8u << 20)) c->direct_cap_opaque = std::max(c->direct_cap_opaque, c->nf);
        const uint32_t cap_o = (c->direct_cap_opaque + 31u) & ~31u;
        const uint32_t region = cap_o + (r.with_class ? BLEND_SORT_CAP : 0u);
        const size_t need = (size_t)ntiles * region + 64;
        if (need <= ((size_t)1 << 28)) {                            // 1 GB of list space at most; beyond that the compact counting sort
            if (need > c->cap_direct || !c->direct_lists) {
                if ((rc = ensure_plain(c, c->direct_lists, need + need / 8))) return rc;
                c->cap_direct = need + need / 8;
            }
            const size_t need_fill = (size_t)ntiles * FILL_PAD + 64;
            if (need_fill > c->cap_tile_fill || !c->tile_fill) {
                if ((rc = ensure_plain(c, c->tile_fill, need_fill * 2))) return rc;
                c->cap_tile_fill = need_fill * 2;
                HIPCHK(c, hipMemsetAsync(c->tile_fill, 0, c->cap_tile_fill * sizeof(uint32_t), c->stream));   // zero from here on: k_cover re-zeroes what k_setup counted
                c->side_dirty = true;
            }
            if (++c->epoch == 0) c->epoch = 1;
            r.db.fill = c->tile_fill; r.db.lists = c->direct_lists; r.db.region = region; r.db.cap_opaque = cap_o;
            r.db.cap_transparent = r.with_class ? BLEND_SORT_CAP : 0u; r.db.with_class = r.with_class ? 1u : 0u; r.db.epoch = c->epoch;
            r.direct_bin = true;
            r.list_stride = region;
        } else c->direct_ok = false;
    }
    return B32_OK;
}

// frames of a mesh that stays (second frame on) and is too large for the in-kernel list collection: k_setup culls and bins every face
// from packed positions and reads the packed (u, v, rgba) only of the faces it draws (on a band-sharded frame: that reach this rank's rows)
static int frame_positions(b32_ctx* c, const FrameParams& fp, const float*& pos12, const float*& attr12) {
    int rc;
    pos12 = attr12 = nullptr;
    if (c->nv && c->nf > 8192u && !(c->route_off & B32_ROUTE_PACKED_STREAMS)) {
        // streams per vertex, one behind the other: 12 B positions, 12 B (u, v, rgba) and -- only once the mesh has been drawn with a
        // shading pass -- 24 B (u, v, rgba, normal) for lit frames (floats: 3 + 3 [+ 6] per vertex)
        const bool want_lit = fp.shading != B32_SHADE_NONE;
        if ((!c->pos_valid || (want_lit && !c->lit_valid)) && c->band_frames >= 1) {
            const bool with_lit = want_lit || c->lit_valid;
            const size_t need = (size_t)c->nv * (with_lit ? 12 : 6);
            if (need > c->cap_pos12 || !c->d_pos12) {
                if ((rc = ensure_plain(c, c->d_pos12, need + 16))) return rc;
                c->cap_pos12 = need;
            }
            launch_pack_streams(c->stream, c->d_verts, c->nv, c->d_pos12, c->d_pos12 + (size_t)c->nv * 3, with_lit);
            c->pos_valid = true; c->lit_valid = with_lit; c->side_dirty = true;
        }
        c->band_frames++;
        if (c->pos_valid && (!want_lit || c->lit_valid)) { pos12 = c->d_pos12; attr12 = c->d_pos12 + (size_t)c->nv * 3; }
    }
    return B32_OK;
}

// The keyed pipelines (no sort-free path for this frame): pairs keyed by (tile, class), grouped by radix passes; returns the pair buffer
// that holds the grouped lists.  ev_bin: event to record when the binning proper starts (profiling level 2), or nullptr.
static int bin_keyed(b32_ctx* c, const FrameParams& fp, const Route& r, const SortScratch& sc, hipEvent_t ev_bin, int& cur) {
    hipStream_t s = c->stream;
    const uint32_t ntiles = fp.tiles_x * fp.tiles_y;
    cur = 0;
    if (r.local_sort) {
        // fast path: no global depth sort.  Pairs are emitted in face order from k_setup's spans; k_cover sorts every tile
        // list by depth key in LDS (stable, so ties keep face order).
        if (ev_bin) HIPCHK(c, hipEventRecord(ev_bin, s));
        launch_bin_faces(s, fp, c->spans, c->keys[0], c->partials, c->d_ctrl, c->pkeys[0], c->pvals[0], (uint32_t)c->cap_pairs, 0);
    } else {
        // painter's order: 4 stable passes over the 32-bit key; pass 1 also compacts away culled faces and its scan kernel
        // reduces k_setup's counters into Ctrl (n_visible feeds the later passes).
        RadixExtra ex1; ex1.post_ctrl = c->d_ctrl; ex1.partials = c->partials; ex1.npart = (c->nf + 255) / 256;
        launch_radix_pass(s, c->keys[0], nullptr, c->keys[1], c->vals[1], c->d_consts, c->nf, 0, 8, sc, ex1);
        launch_radix_pass(s, c->keys[1], c->vals[1], c->keys[0], c->vals[0], &c->d_ctrl->n_visible, c->nf, 8, 8, sc);
        launch_radix_pass(s, c->keys[0], c->vals[0], c->keys[1], c->vals[1], &c->d_ctrl->n_visible, c->nf, 16, 8, sc);
        launch_radix_pass(s, c->keys[1], c->vals[1], c->keys[0], c->vals[0], &c->d_ctrl->n_visible, c->nf, 24, 8, sc);
        if (fp.ortho) {      // 32-bit depth keys: the opaque/transparent partition is a fifth stable pass on the class
            launch_class_keys(s, c->crecs, c->vals[0], &c->d_ctrl->n_visible, c->nf, c->keys[0]);
            launch_radix_pass(s, c->keys[0], c->vals[0], c->keys[1], c->vals[1], &c->d_ctrl->n_visible, c->nf, 0, 8, sc);
            HIPCHK(c, hipMemcpyAsync(c->vals[0], c->vals[1], (size_t)c->nf * 4, hipMemcpyDeviceToDevice, s));
        }
        if (ev_bin) HIPCHK(c, hipEventRecord(ev_bin, s));
        launch_bin(s, fp, c->spans, c->vals[0], c->d_ctrl, c->counts, c->block_sums, c->bin_blocks, c->pkeys[0], c->pvals[0], (uint32_t)c->cap_pairs);
    }
    const uint32_t n_sort_keys = r.local_sort ? ntiles : 2 * ntiles;          // the fast path groups by tile only
    const uint32_t kb = bits_for(n_sort_keys ? n_sort_keys : 1);
    if (kb <= 8 || kb > 12) {
        for (uint32_t shift = 0; shift < kb; shift += 8) {
            launch_radix_pass(s, c->pkeys[cur], c->pvals[cur], c->pkeys[cur ^ 1], c->pvals[cur ^ 1], &c->d_ctrl->n_pairs, (uint32_t)c->cap_pairs, (int)shift, 8, sc);
            cur ^= 1;
        }
        launch_tile_ranges(s, c->pkeys[cur], c->d_ctrl, (uint32_t)c->cap_pairs, c->ranges, n_sort_keys);
    } else {    // up to 2048 tiles: one pass groups every (tile, class) list and its digit bases are the list ranges
        RadixExtra exr; exr.ranges_out = c->ranges; exr.n_ranges = n_sort_keys + 1;
        launch_radix_pass(s, c->pkeys[cur], c->pvals[cur], c->pkeys[cur ^ 1], c->pvals[cur ^ 1], &c->d_ctrl->n_pairs, (uint32_t)c->cap_pairs, 0, kb <= 11 ? 11 : 12, sc, exr);
        cur ^= 1;
    }
    return B32_OK;
}

static FillArgs fill_args(const b32_ctx* c, const FrameParams& fp, const Route& r, int cur, bool wire_front) {
    FillArgs fa{};
    fa.fp = fp; fa.crecs = c->crecs; fa.srecs = c->srecs; fa.xrecs = c->xrecs; fa.shades = c->shades; fa.pair_vals = c->pvals[cur]; fa.ranges = c->ranges;
    fa.keys = c->keys[0]; fa.local_sort = r.local_sort ? 1u : 0u; fa.tile_keys_only = (r.local_sort || r.prio64) ? 1u : 0u; fa.tile_mid = c->tile_mid;
    fa.tex = c->d_tex; fa.texels = c->d_texels; fa.fb = c->fb; fa.vis = c->vis; fa.zbuf = c->zbuf; fa.ctrl = c->d_ctrl;
    fa.tex0 = c->nt ? c->h_tex[0] : TexDesc{ 0, 0, 0, 0 };
    fa.lds_tex_texels = 0;
    if (c->nt == 1 && r.exact_cov) {                     // (CHEAP coverage: one texel fetch per output pixel, served by L1/L2)
        const size_t n = (size_t)c->h_tex[0].width * c->h_tex[0].height;
        if (n > 0 && n * 2 <= fill_lds_tex_budget()) fa.lds_tex_texels = (uint32_t)n;
    }
    fa.exact_coverage = r.exact_cov ? 1u : 0u;
    fa.may_blend = c->may_blend ? 1u : 0u;
    fa.skip_solid = wire_front ? 1u : 0u;
    fa.texels32 = c->d_texels32;
    fa.ordered_all = r.ordered_all ? 1u : 0u;
    fa.prio64 = r.prio64 ? 1u : 0u;
    fa.narrow_only = (c->route_off & B32_ROUTE_WIDE_GROUPS) ? 1u : 0u;
    fa.texmask = c->d_texmask;
    { const uint32_t words = c->pool_texels / 32 + 2; fa.mask_lds_words = (c->pool_texels && words <= MASK_LDS_MAX_WORDS) ? words : 0u; }
    fa.inline_bin = r.inline_bin ? 1u : 0u; fa.list_stride = r.list_stride; fa.spans = c->spans; fa.partials = c->partials;
    if (r.inline_bin) fa.pair_vals = c->inline_lists;
    fa.direct_bin = r.direct_bin ? 1u : 0u; fa.tile_fill = c->tile_fill; fa.epoch = c->epoch;
    if (r.direct_bin) fa.pair_vals = c->direct_lists;
    fa.gather_blend = (r.prio64 && r.with_class) ? 1u : 0u;
    fa.stagger = (c->route_off & B32_ROUTE_STAGGER) ? 0u : 1u;       // (launch_fill decides whether the frame qualifies)
    fa.span_cover = (r.prio64 && !r.exact_cov && !fp.zmode && !(c->route_off & B32_ROUTE_SPAN_COVER)) ? 1u : 0u;
    // index atlas + CLUT sampled from LDS: the fused kernel with one indexed texture, when they fit beside the tile planes of the
    // workgroup form launch_fill is going to choose (16 waves, one workgroup per CU: ~84 KB; two 8-wave workgroups per CU: ~6 KB)
    fa.atlas0 = c->d_atlas0; fa.atlas_idx_bytes = 0;
    if (r.prio64 && !c->fmt8 && c->nt == 1 && c->atlas_idx_bytes && !(c->route_off & B32_ROUTE_LDS_ATLAS)) {
        bool wide = fp.tiles_x * fp.tiles_y <= (uint32_t)c->n_cu && !(c->route_off & B32_ROUTE_WIDE_GROUPS);
#ifdef B32_EXP_LDS_ATLAS
        wide = true;                     // (experiment build: launch_p64 sends the plain frame through the 16-wave form)
#endif
        if (c->atlas_idx_bytes + ATLAS_CLUT_BYTES + 16u <= fill_lds_atlas_room(wide)) fa.atlas_idx_bytes = c->atlas_idx_bytes;
    }
    if (c->fmt8) fa.fp.xray = 0;                        // render_mesh: x-ray only changes culling; its stores keep their own depth tests
    return fa;
}

int enqueue_frame(b32_ctx* c, const B32Camera* cam, const B32Settings* st, const B32Fog* fog) {
    hipStream_t s = c->stream;
    const bool wire_back = st->backface_cull && st->backface_wireframe;      // render.rs:2577
    const bool wire_front = st->wireframe_overlay != 0;                       // render.rs:2603 (an empty list draws nothing either way)
    FrameParams fp = frame_params(c, cam, st, fog, wire_back || wire_front);
    LightSet lset{};
    int rc;
    const bool prof_sample = c->profile_level >= 1 && (c->prof_seq++ % c->prof_stride) == 0;
    const bool prof_all = prof_sample && c->profile_level >= 2, prof_fill = prof_sample;
    // Two frames in flight: when an earlier frame of this context is still pending, this frame of a large mesh takes the OTHER frame set
    // and (if it ends up on the direct-binning route) its setup kernel runs on the side stream, beside that frame's fill.
    c->pipelined = false;
    // (pipe_hint: whether the previous frame's route qualified -- a frame that will not, e.g. every frame of a PS1-sized target, skips
    // the set swap and its event as well: 0.030 -> 0.028 ms on 20 k triangles at 320x240)
#ifdef B32_EXP_PIPE_SMALL
    const uint32_t pipe_min_faces = 0u;              // (experiment build: small frames pipelined too)
#else
    const uint32_t pipe_min_faces = 2048u;
#endif
    // (not on the legacy default stream -- hipStreamLegacy, what torch's default stream maps to: it synchronises implicitly with every
    // blocking stream, and recording / waiting cross-stream events on that handle crashed the runtime, found when safe mode began to
    // leave superseded frames in flight)
    if (c->frame_pending && c->pipe_hint && !c->redrawing && !(fp.wire_collect && B32_EXP_NO_WIRE_PIPE) && !prof_all && c->nf > pipe_min_faces && !(c->route_off & B32_ROUTE_PIPELINE) &&
        c->stream != hipStreamLegacy) {
        if ((rc = pipeline_ensure(c))) return rc;
        rotate_sets(c);
        c->pipelined = true;
        if (c->join_stream != s) {
            // Streams of different priorities never share a hardware queue (the runtime pools its queues by priority); two streams of ONE
            // priority may, and k_join in front of k_flag in the same queue would wait for its patience: those keep the event.
            int pm = 0, ps = 0;
            c->join_ok = hipStreamGetPriority(s, &pm) == hipSuccess && hipStreamGetPriority(c->side, &ps) == hipSuccess && pm != ps;
            c->join_stream = s;
        }
    }
    // (an error return between the rotation and the launches puts the sets back: the pending frame stays the current set's)
    bool rotated = c->pipelined;
    auto fail = [&](int e) { if (rotated) { unrotate_sets(c); rotated = false; c->pipelined = false; } return e; };
    if ((rc = frame_lights(c, st, fp, lset))) return fail(rc);
    if ((rc = frame_buffers(c, fp, wire_back))) return fail(rc);
    hipEvent_t* ev = nullptr;
    if (prof_fill) {
        if (!c->ev_created) {
            for (auto& fr : c->ev) for (auto& e : fr) if (hipEventCreate(&e) != hipSuccess) return fail(B32_E_HIP);
            c->ev_created = true;
        }
        ev = c->ev[c->ev_frames % EV_RING];
    }

    const SortScratch sc{ c->block_hist, c->hist_blocks, c->digit_total };
    Route r;
    if ((rc = plan_route(c, fp, sc, wire_front, r))) return fail(rc);
    // (only large meshes: the frames of small ones are launch-latency bound and the cross-stream events cost them more than the overlap
    // returns -- a 12-room console frame 0.72 ms against 0.65; keyed routes have binning launches behind k_setup: one stream)
    const uint32_t ntiles = fp.tiles_x * fp.tiles_y;
    // ... and a frame whose fused kernel has no more tiles than workgroup slots has no tail to fill: the cross-stream waits (~10 us
    // between two kernels) then cost more than the overlap returns (C2, 100 k triangles at 320x240: 0.052 against 0.048 ms).  The merged
    // runs of a batched frame are the exception: their kernels leave most of the GPU idle anyway (75 tiles for 256 CUs).
    // Round 6: with the flag / join kernel pair (no cross-stream event: ~5 us instead of ~12) frames WITHOUT a tail pay too when their two
    // kernels are of comparable length -- C2, 100 k triangles at 320x240: setup 15.7 + fill 21.9 us, 0.0375 -> 0.0339 ms per frame --
    // but only on that hand-over (the small-mesh route keeps the event: C1 0.0235 -> 0.0281), i.e. while the caller's stream and the side
    // stream have different priorities (unknown before the first pipelined frame of a stream: tried once, then decided).
    // The same for the frames of a screen band (one rank of a sharded frame: 240 rows of C4 are 320 tiles of 32 rows) -- round 5 had measured
    // them with the event (N = 8 band 0.067 -> 0.071: no); with the kernel pair the weak series' N = 8 point goes 0.054 -> 0.042 ms per rank.
    const bool join_small = B32_JOIN_KERNEL && B32_PIPE_FEW_TILES && !c->frame_batched && (c->join_stream != s || c->join_ok);
    c->pipe_hint = r.direct_bin && (ntiles > 2u * (uint32_t)c->n_cu || c->frame_batched || (c->band_set && B32_PIPELINE_BANDS) || join_small);
#ifdef B32_EXP_PIPE_SMALL
    c->pipe_hint = (r.direct_bin || r.want_inline) && !c->band_set;
#endif
    if (!c->pipe_hint) c->pipelined = false;       // (the frame keeps the set it rotated to -- the route's regions are that set's -- but runs on the main stream)
    c->last_local_sort = r.local_sort || r.want_prio64;                         // the global draw order is not materialised
    c->last_exact = r.ordered_all ? true : (r.exact_cov && !fp.zmode);          // the ordered walk counts every store it performs
    c->last_direct = r.direct_bin;
    if (c->nf == 0) {                                                             // otherwise k_setup resets it (all but `sticky`)
        HIPCHK(c, hipMemsetAsync(c->d_ctrl, 0, offsetof(Ctrl, sticky), s));
        HIPCHK(c, hipMemsetAsync(&c->d_ctrl->fragments, 0, sizeof(unsigned long long), s));
        HIPCHK(c, hipMemsetAsync(c->d_ctrl + 1, 0, sizeof(Stamps), s));          // (and the phase clock: b32_last_shader_clock of an empty frame is 0, not the frame before's)
    }
    const float *pos12 = nullptr, *attr12 = nullptr;
    if ((rc = frame_positions(c, fp, pos12, attr12))) return fail(rc);

    // ---- transform, cull, setup (+ tile binning of large meshes)
    hipStream_t ss = s;
    if (c->pipelined) {
        if (c->side_dirty) { HIPCHK(c, hipEventRecord(c->ev_main, s)); HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_main, 0)); c->side_dirty = false; }
        ss = c->side;
        c->pipelined_frames++;
        uint32_t polled_tiles = 0, polled_groups = 0, polled_seq = 0;       // the fused kernel of the frame whose cursor the gate polls (alt[0]: n_sets - 1 frames back)
        for (const auto& co : c->cover_of) if (co.ctrl && co.ctrl == c->alt[0].d_ctrl) { polled_tiles = co.tiles; polled_groups = co.groups; polled_seq = co.seq; }
        // The set this frame's setup kernel writes was last read by the fill n_sets frames back.  That fill lies in front of alt[0]'s on the main
        // stream: when alt[0]'s frame launched a fused kernel, "it has started" orders the two on the device (k_gate, no event on the main
        // stream: B32_START_GATE); else the side stream waits for the main stream as it stands now.
        const bool start_gate = B32_START_GATE && polled_tiles && polled_seq && c->alt[0].d_ctrl;
#if B32_START_GATE
        if (!start_gate) { HIPCHK(c, hipEventRecord(c->ev_main, s)); HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_main, 0)); }
#else
        HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_done, 0));     // the last fill that read this set (n_sets frames ago)
#endif
        uint32_t gate_need = 0;
        if (c->gate_permille && polled_tiles && c->alt[0].d_ctrl) {
            // The fused kernel's workgroups take their next tile from the cursor after the coverage of the current one: the cursor
            // passes tiles - groups when the last tile is handed out, and every fetch beyond that is a workgroup that found the queue
            // empty and has only the shading of its last tile left, i.e. is about to free its place on a CU.
            const uint32_t groups = polled_groups, tiles = polled_tiles;
            const uint32_t pre = tiles > groups ? tiles - groups : 0u;      // cursor value when the last tile is handed out
            const uint32_t need = c->gate_permille > 1000u ? (uint32_t)((uint64_t)(c->gate_permille - 1000u) * pre / 1000u)
                                                           : pre + (uint32_t)((uint64_t)(c->gate_permille - 1u) * groups / 1000u);
            gate_need = need;
        }
        // (b32_debug_inject(ctx, 2): the fill in front did not publish its start -- this gate's patience is 2 ms, then the error)
        const uint32_t start_patience = c->start_lost ? 200000u : 200000000u;
        c->start_lost = false;                               // (one gate only, whichever way this frame is ordered)
        if (gate_need || start_gate) launch_gate(ss, c->alt[0].d_ctrl, gate_need, 30000u /* 300 us */, start_gate ? polled_seq : 0u, c->d_ctrl, start_patience);      // alt[0]: the frame n_sets - 1 back (rotate_sets)
    }
    // (the wire kernels' arguments: known before the setup kernel is launched -- a pipelined frame bins its wire list on the side stream)
    const bool wire_on = fp.wire_collect && c->nf;
    bool wire_binned = false, wire_polled = false;
    WireArgs wa{};
    if (wire_on) {
        wa.tris = c->wire; wa.nf = c->nf; wa.table_owner = c->wire_owner; wa.table_first = c->wire_first;
        wa.table_mask = c->cap_wire_table ? (uint32_t)(c->cap_wire_table - 1) : 0;
        wa.fb = c->fb; wa.zbuf = (c->zbuf && c->zbuf_valid) ? c->zbuf : nullptr;
        wa.width = c->width; wa.height = c->height; wa.band_y0 = c->band_y0; wa.band_y1 = c->band_y1; wa.ctrl = c->d_ctrl;
        if (++c->wire_seq == 0) c->wire_seq = 1;
        wa.epoch = c->wire_seq;
        if (!(c->route_off & B32_ROUTE_WIRE_TILES) && c->wire_fill && c->band_y1 > c->band_y0) {
            wa.tile_yb = (c->band_y0 / WIRE_TH) * WIRE_TH; wa.tiles_x = (c->width + TILE_W - 1) / TILE_W;
            wa.tiles_y = (c->band_y1 - wa.tile_yb + WIRE_TH - 1) / WIRE_TH;
            if ((size_t)wa.tiles_x * wa.tiles_y <= c->cap_wire_tiles) { wa.tile_fill = c->wire_fill; wa.tile_lists = c->wire_lists; c->wire_tile_frames++; }
        }
    }
    if (prof_all) HIPCHK(c, hipEventRecord(ev[0], s));
    launch_setup(ss, fp, c->d_verts, c->d_faces, c->d_tex, c->d_lights, lset, c->frame_table, RecArrays{ c->crecs, c->srecs, c->xrecs }, r.db, c->shades, c->keys[0],
                 r.direct_bin ? nullptr : c->spans /* (direct binning: nobody reads the spans) */, c->partials, c->d_ctrl, c->wire, c->n_cu, pos12, attr12, c->face_of);
    // The merged draws of a batched frame: the hand-over polled by the fused kernel itself (FillArgs::join_seq) -- their fills are few 16-wave
    // workgroups (at most 5 / 8 of the CUs: the setup kernel they may have to spin for keeps the rest of the GPU), the setup kernel of draw
    // k + 1 has normally finished beside draw k's fill and blend pass, and what the cross-stream event cost the main stream per draw (6.5 us
    // between k_blend's end and the next fill's start: tools/console_trace.sh) was most of what there was to save in a console frame.
    uint32_t poll_seq = 0, poll_patience = 0;
    const bool poll_batched = B32_POLL_BATCHED && c->pipelined && c->frame_batched && c->join_ok && (r.direct_bin || r.want_inline) && !wire_front && !r.ordered_all &&
                              ntiles && 8u * ntiles <= 5u * (uint32_t)c->n_cu && !(c->route_off & B32_ROUTE_WIDE_GROUPS);
    if (poll_batched) {
        // (b32_debug_inject(ctx, 1) as below: the flag carries another value, the patience is 2 ms)
        const bool lose_flag = (c->inject & 1u) != 0;
        c->inject &= ~1u;
        if (++c->join_seq == 0) c->join_seq = 1;
        launch_flag_poll(c->side, c->d_ctrl, lose_flag ? c->join_seq ^ 0x40000000u : c->join_seq);
        poll_seq = c->join_seq; poll_patience = lose_flag ? 200000u : 200000000u;
        c->flag_join_frames++; c->poll_join_frames++;
    } else if (c->pipelined && B32_JOIN_KERNEL && r.direct_bin && c->join_ok && !(c->frame_batched && B32_JOIN_NOT_BATCHED)) {
        // (no cross-stream event on the fill's path: see k_flag / k_join)
        // (b32_debug_inject(ctx, 1): this frame's flag carries another epoch and the join's patience is 2 ms -- the "setup kernel never arrived" path)
        const bool lose_flag = (c->inject & 1u) != 0;
        c->inject &= ~1u;
        launch_flag(c->side, c->d_ctrl, lose_flag ? c->epoch ^ 0x40000000u : c->epoch);
        launch_join(s, c->d_ctrl, c->epoch, lose_flag ? 200000u : 200000000u /* 2 s: a last resort -- queues of an oversubscribed GPU are time-sliced in milliseconds */);
        c->flag_join_frames++;
    } else if (c->pipelined) {
        hipError_t e1 = hipEventRecord(c->ev_setup, c->side);
        if (e1 == hipSuccess) e1 = hipStreamWaitEvent(s, c->ev_setup, 0);
        if (e1 != hipSuccess) { (void)hipStreamSynchronize(c->side); c->last_hip = (int)e1; return B32_E_HIP; }
        c->event_join_frames++;
    }
    if (c->pipelined) {
        if (wire_on && wa.tile_fill && B32_WIRE_BIN_EARLY) {       // (behind ev_setup: the fill does not wait for the binning)
            launch_wire_bin(c->side, wa, wire_back, wire_front, true);
            // (back-face edges: the first wire kernel on the main stream, k_wire_table_clear, looks at Events::wbin_done itself -- no event
            // for the main stream to wait for; the overlay alone has no such kernel in front of k_wire_tile and keeps the event)
            if (wire_back) { launch_flag_wbin(c->side, c->d_ctrl, wa.epoch); wire_polled = true; }
            else HIPCHK(c, hipEventRecord(c->ev_wbin, c->side));
            wire_binned = true;
        }
    }
    c->set_in_flight = true;
    if (prof_all) HIPCHK(c, hipEventRecord(ev[1], s));

    // ---- tile lists
    int cur = 0;
    if (r.direct_bin) {
        if (prof_all) HIPCHK(c, hipEventRecord(ev[2], s));
        r.prio64 = true;
    } else if (r.want_inline) {
        const size_t need = (size_t)ntiles * r.list_stride + 64;
        if (need > c->cap_inline) {
            if ((rc = ensure_plain(c, c->inline_lists, need + need / 2))) return rc;
            c->cap_inline = need + need / 2;
        }
        if (prof_all) HIPCHK(c, hipEventRecord(ev[2], s));
        r.prio64 = r.inline_bin = true;
    } else if (r.want_prio64) {
        if (prof_all) HIPCHK(c, hipEventRecord(ev[2], s));
        r.prio64 = launch_bin_spans(s, fp, c->spans, r.with_class ? c->keys[0] : nullptr, c->partials, c->d_ctrl, sc, (uint32_t)c->cap_pairs, c->ranges,
                                    c->tile_mid, BLEND_SORT_CAP, c->pvals[0]);
    }
    if (!r.prio64 && (rc = bin_keyed(c, fp, r, sc, prof_all ? ev[2] : nullptr, cur))) return rc;
    c->last_pair_buf = cur;
    c->routes[r.direct_bin ? 0 : r.inline_bin ? 1 : r.prio64 ? 2 : 3]++;
    // (direct binning with regions that hold the whole mesh, and no more faces that can be transparent than k_blend sorts per tile:
    // nothing can overflow)
    const bool direct_safe = r.direct_bin && r.db.cap_opaque >= c->nf && (!r.with_class || c->blend_faces <= BLEND_SORT_CAP);
    c->pending_may_redraw = !(r.inline_bin || direct_safe);
    if (prof_fill) HIPCHK(c, hipEventRecord(ev[3], s));

    // ---- coverage, shading, transparent pass
    FillArgs fa = fill_args(c, fp, r, cur, wire_front);
    if (++c->fill_seq == 0) c->fill_seq = 1;
    fa.start_seq = c->fill_seq;
    fa.join_seq = poll_seq; fa.join_patience = poll_patience;
    const uint32_t start_seq_meant = fa.start_seq;
    if ((c->inject & 2u) && fa.prio64 && !wire_front && !r.ordered_all && ntiles) { c->inject &= ~2u; fa.start_seq = 0; c->start_lost = true; }    // (fault injection: see b32_debug_inject)
    // Which kernel of the frame says "started": the fused kernel -- or, when the frame has a transparent pass, that pass (k_blend, the next
    // kernel on the main stream).  The next frame's setup kernel is released by that word, i.e. it then runs beside the blend kernel and the
    // fill has the GPU to itself: C3 with 10 % transparent faces 0.227 -> 0.192 ms per frame, the same in z-buffer mode 0.264 -> 0.222 (the fill
    // is the kernel that suffers from company: 97 us beside a setup kernel, 82 alone; k_blend 127 either way).  The wire tile kernel of
    // default() in that role: 0.284 -> 0.331 -- its setup kernel is the long one and then starts too late; not done.
    // Only frames that fill the GPU (more tiles than workgroup slots): a console-sized draw leaves most CUs idle anyway and its setup kernel is
    // a chain of round trips that wants to start as early as it may (12-room console frame with the deferral: 0.171 -> 0.213 ms).
    fa.start_defer = (B32_START_AT_BLEND && fa.prio64 && !wire_front && !r.ordered_all && ntiles > 2u * (uint32_t)c->n_cu && fa.gather_blend) ? 1u : 0u;
    // a deferred Framebuffer::clear: folded into this frame's fused kernel when that kernel is the one that runs, the frame has no
    // depth buffer to reset and the clear was issued for this very band; else the clear launches go first
    if (c->clear_pending) {
        // (a frame with a depth buffer to reset: only in z-buffer mode, where the fused kernel owns the depth buffer too -- it seeds its
        // winners with f32::MAX instead of reading the buffer and writes f32::MAX where nothing is drawn)
        const bool has_z = c->zbuf && c->zbuf_valid;
        if (r.prio64 && !wire_front && !r.ordered_all && (!has_z || fp.zmode) && c->nf && ntiles && c->clear_y0 == c->band_y0 && c->clear_y1 == c->band_y1) {
            fa.clear_on = 1; fa.clear_rgba = c->clear_rgba; fa.clear_depth = (has_z && fp.zmode) ? 1u : 0u; c->clear_pending = false;
        } else if ((rc = flush_clear(c))) return rc;
    }
    if (fa.atlas_idx_bytes && !wire_front && !r.ordered_all) c->lds_atlas_frames++;
    if (fa.span_cover && !wire_front && !r.ordered_all) c->span_cover_frames++;
    launch_fill(s, fa, c->n_cu, prof_fill ? ev[4] : nullptr);

    // ---- wireframe phases
    if (wire_on) {
        if (wire_binned && !wire_polled) HIPCHK(c, hipStreamWaitEvent(s, c->ev_wbin, 0));
        launch_wire(s, wa, wire_back, wire_front, wire_binned, wire_polled ? c->d_ctrl : nullptr, wa.epoch);
    }
    if (prof_fill) { if (prof_all) HIPCHK(c, hipEventRecord(ev[5], s)); c->ev_frames++; }
#if !B32_START_GATE
    if (c->side) HIPCHK(c, hipEventRecord(c->ev_done, s));         // (the next setup kernel that writes this set waits for it)
#endif
    c->last_cover_tiles = (r.prio64 && !wire_front && !r.ordered_all) ? ntiles : 0u;
    c->last_cover_groups = std::min<uint32_t>(ntiles, (uint32_t)c->n_cu * 2u);
    {   // remembered per frame set
        b32_ctx::CoverOf* slot = &c->cover_of[0];
        for (auto& co : c->cover_of) { if (co.ctrl == c->d_ctrl) { slot = &co; break; } if (!co.ctrl) slot = &co; }
        *slot = { c->d_ctrl, c->last_cover_tiles, c->last_cover_groups, c->last_cover_tiles ? start_seq_meant : 0u };
    }
    HIPCHK(c, hipGetLastError());
    return B32_OK;
}

int b32_render_scene_15_async(b32_ctx* c, const B32Camera* cam, const B32Settings* st, const B32Fog* fog) {
    if (!c || c->fmt8) return B32_E_ARG;                 // the resident scene holds Texture (8-bit) texels: use b32_render_scene
    c->frame_batched = false;
    return render_scene_async_any(c, cam, st, fog);
}
int b32_render_scene_async(b32_ctx* c, const B32Camera* cam, const B32Settings* st) {
    if (!c || !c->fmt8) return B32_E_ARG;
    c->frame_batched = false;
    return render_scene_async_any(c, cam, st, nullptr);
}
int render_scene_async_any(b32_ctx* c, const B32Camera* cam, const B32Settings* st, const B32Fog* fog) {
    if (!c || !cam || !st || !c->fb || !c->have_scene) return B32_E_ARG;
    (void)hipSetDevice(c->device);
    int rc = validate_settings(st);
    if (rc) return rc;
    // keep a private copy of the lights so a redraw after overflow does not dereference a dead caller pointer
    // (safe mode) a pending frame that may still need a redraw is settled before the next one overwrites its control block
    // (... unless a clear of the whole band has superseded it: b32_fb_clear)
    if (!c->deep_async && !(c->pending_superseded && c->clear_pending) && (rc = settle_pending(c))) return rc;
    c->pending_superseded = false;
    c->last_cam = *cam; c->last_settings = *st; c->last_has_fog = fog != nullptr;
    if (fog) c->last_fog = *fog;
    c->keep_lights.assign(st->lights, st->lights + (st->lights ? st->n_lights : 0));
    c->last_settings.lights = c->keep_lights.empty() ? nullptr : c->keep_lights.data();
    rc = enqueue_frame(c, cam, &c->last_settings, fog);
    if (rc == B32_OK) c->frame_pending = true;
    return rc;
}

static void collect_events(b32_ctx* c) {
    c->phase_frames = 0;
    for (float& p : c->phase_ms) p = 0;
    if (!c->ev_created || c->ev_frames == 0 || c->profile_level < 1) { c->ev_frames = 0; return; }
    const uint32_t n = c->ev_frames < (uint32_t)EV_RING ? c->ev_frames : (uint32_t)EV_RING;
    for (uint32_t i = 0; i < n; ++i) {
        float ms = 0;
        if (c->profile_level >= 2) {
            for (int p = 0; p < 3; ++p) if (hipEventElapsedTime(&ms, c->ev[i][p], c->ev[i][p + 1]) == hipSuccess) c->phase_ms[p] += ms;
            if (hipEventElapsedTime(&ms, c->ev[i][4], c->ev[i][5]) == hipSuccess) c->phase_ms[4] += ms;
        }
        if (hipEventElapsedTime(&ms, c->ev[i][3], c->ev[i][4]) == hipSuccess) c->phase_ms[3] += ms;     // the coverage kernel alone
    }
    for (float& p : c->phase_ms) p /= (float)n;
    c->phase_frames = n;
    c->phase_level = c->profile_level;
    c->ev_frames = 0;
}

int b32_frame_finish(b32_ctx* c, B32Timings* out) {
    if (!c) return B32_E_ARG;
    (void)hipSetDevice(c->device);
    if (out) memset(out, 0, sizeof(*out));
    // A clear issued after the frame's draw (deep mode: safe mode settled the frame before it recorded the clear) stays deferred until
    // the frame has been settled: a redraw below must land UNDER that clear, not on top of it, and must not fold it either.
    const bool later_clear = c->frame_pending && c->clear_pending;
    const uint32_t lc_rgba = c->clear_rgba, lc_y0 = c->clear_y0, lc_y1 = c->clear_y1;
    if (later_clear) c->clear_pending = false;
    struct ClearAfter {      // re-arms and flushes the later clear on every exit path
        b32_ctx* c; bool on; uint32_t rgba, y0, y1;
        ~ClearAfter() { if (on) { c->clear_pending = true; c->clear_rgba = rgba; c->clear_y0 = y0; c->clear_y1 = y1; (void)flush_clear(c); (void)hipStreamSynchronize(c->stream); } }
    } clear_after{ c, later_clear, lc_rgba, lc_y0, lc_y1 };
    // A superseded frame (safe mode, b32_fb_clear) whose clear has ALREADY been executed -- something flushed it without settling:
    // b32_synchronize -- must not be redrawn: its pixels would land on top of that clear.  It is retired instead: its counters and
    // errors are read, the capacities it asked for are granted to the frames that follow, nothing is enqueued.
    const bool no_redraw = c->frame_pending && c->pending_superseded && !later_clear;
    if (!later_clear) { const int rcf = flush_clear(c); if (rcf) return rcf; }
    if (!c->frame_pending) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        const int d = c->deferred_rc; c->deferred_rc = 0;
        return d;
    }
    for (int attempt = 0; attempt < 5; ++attempt) {
        // the frame's counters come back through the pinned arena (one small kernel writing host memory) rather than an SDMA copy:
        // ~5 us of stream time less per synchronous frame
        if (stage_ensure(c)) {
            launch_ctrl_out(c->stream, c->d_ctrl, static_cast<unsigned char*>(c->stage_dev) + STAGE_CTRL_OFF);
            HIPCHK(c, hipStreamSynchronize(c->stream));
            std::memcpy(&c->h_ctrl, c->stage_host + STAGE_CTRL_OFF, sizeof(Ctrl));
            std::memcpy(&c->h_stamps, c->stage_host + STAGE_CTRL_OFF + sizeof(Ctrl), sizeof(Stamps));
        } else {
            unsigned char tmp[sizeof(Ctrl) + sizeof(Stamps)];
            HIPCHK(c, hipMemcpyAsync(tmp, c->d_ctrl, sizeof(tmp), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            std::memcpy(&c->h_ctrl, tmp, sizeof(Ctrl)); std::memcpy(&c->h_stamps, tmp + sizeof(Ctrl), sizeof(Stamps));
            c->h_stamps.t[ST_END] = 0;
        }
        if ((c->h_ctrl.need_global_sort & 2u) && c->last_direct) {
            // direct binning: a tile region was too small and nothing was drawn; redraw this frame with regions a quarter above the
            // longest list it reported (enqueue_frame falls back to the compact counting sort if those would not fit)
            c->direct_cap_opaque = c->h_ctrl.list_demand + c->h_ctrl.list_demand / 4 + 64;
            if (no_redraw) break;
            c->routes[4]++;
            c->ev_frames = 0;
            c->redrawing = true;
            const int rc = enqueue_frame(c, &c->last_cam, &c->last_settings, c->last_has_fog ? &c->last_fog : nullptr);
            c->redrawing = false;
            if (rc) return rc;
            continue;
        }
        if ((c->h_ctrl.need_global_sort & 1u) && c->local_sort_ok) {
            // a tile list was longer than the LDS sort handles: nothing was drawn; redraw this frame (and the following ones of
            // this scene) with the global depth sort
            c->local_sort_ok = false;
            if (no_redraw) break;
            c->routes[5]++;
            c->ev_frames = 0;
            c->redrawing = true;
            const int rc = enqueue_frame(c, &c->last_cam, &c->last_settings, c->last_has_fog ? &c->last_fog : nullptr);
            c->redrawing = false;
            if (rc) return rc;
            continue;
        }
        if (!c->h_ctrl.pairs_overflow) break;
        // the fill aborted before touching the framebuffer: grow the pair buffers and redraw the same frame
        const size_t n = (size_t)c->h_ctrl.pairs_overflow + c->h_ctrl.pairs_overflow / 4 + 1024;
        int rc;
        for (int i = 0; i < 2; ++i) { if ((rc = ensure_plain(c, c->pkeys[i], n))) return rc; if ((rc = ensure_plain(c, c->pvals[i], n))) return rc; }
        c->cap_pairs = n;
        if (no_redraw) break;
        c->routes[6]++;
        c->ev_frames = 0;                              // the aborted frame must not enter the phase averages
        c->redrawing = true;
        rc = enqueue_frame(c, &c->last_cam, &c->last_settings, c->last_has_fog ? &c->last_fog : nullptr);
        c->redrawing = false;
        if (rc) return rc;
    }
    c->frame_pending = false;
    c->pending_superseded = false;
    c->set_in_flight = false;
    collect_events(c);
    uint32_t sticky = c->h_ctrl.sticky;                        // errors of every frame enqueued since the last finish
    if (sticky) HIPCHK(c, hipMemsetAsync(&c->d_ctrl->sticky, 0, sizeof(uint32_t), c->stream));
    if (sticky) c->side_dirty = true;                           // (the next setup kernel on the side stream reads that word: after the memset)
    for (FrameSet& o : c->alt) if (o.in_flight && o.d_ctrl) {
        // several frames in flight: the frames of the other sets since the last finish -- their sticky errors, and each set's last frame,
        // which no later k_setup of that set has looked at: dropped (it ran out of list space and drew nothing) means lost
        Ctrl other;
        HIPCHK(c, hipMemcpy(&other, o.d_ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost));      // (the main stream has drained)
        o.in_flight = false;
        uint32_t st2 = other.sticky;
        if (other.pairs_overflow || other.need_global_sort) st2 += 0x100u;
        if (other.sticky) HIPCHK(c, hipMemsetAsync(&o.d_ctrl->sticky, 0, sizeof(uint32_t), c->stream));
        if (other.pairs_overflow || other.need_global_sort) {      // (not again at the next finish)
            HIPCHK(c, hipMemsetAsync(&o.d_ctrl->pairs_overflow, 0, sizeof(uint32_t), c->stream));
            HIPCHK(c, hipMemsetAsync(&o.d_ctrl->need_global_sort, 0, sizeof(uint32_t), c->stream));
        }
        if (other.sticky || other.pairs_overflow || other.need_global_sort) c->side_dirty = true;
        sticky = (sticky | (st2 & 0xFFu)) + (st2 & ~0xFFu);
    }
    if (c->deferred_rc) { const int d = c->deferred_rc; c->deferred_rc = 0; return d; }     // (an earlier mesh of this frame, settled by a swap)
    if (c->h_ctrl.pairs_overflow && !no_redraw) return B32_E_HIP;
    // deep asynchronous mode: an earlier frame was lost (the last one is good).  (Safe mode only ever leaves a frame behind when a clear of
    // the whole band has overwritten whatever it drew: nothing observable was lost.)
    // (a framebuffer that is read behind the library's back -- caller-bound or shared with other ranks -- is never superseded; should a
    // dropped frame be counted there all the same, it is reported)
    if ((sticky >> 8) && (c->deep_async || c->fb_external || c->band_sync)) return B32_E_FRAME_DROPPED;
    if (c->h_ctrl.err_index || (sticky & 1u)) return B32_E_INDEX;
    if (c->h_ctrl.abort || (sticky & 2u)) return B32_E_NAN_KEY;
    if (sticky & 8u) {                                         // (k_join gave up on a setup kernel: internal.  Whatever that kernel left in the tile
        // counters of its frame set -- it may have finished late, or not at all -- is cleared once both streams have drained, so that the
        // frames that follow bin into empty lists again)
        if (c->side) HIPCHK(c, hipStreamSynchronize(c->side));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->tile_fill) HIPCHK(c, hipMemsetAsync(c->tile_fill, 0, c->cap_tile_fill * sizeof(uint32_t), c->stream));
        for (FrameSet& o : c->alt) if (o.tile_fill) HIPCHK(c, hipMemsetAsync(o.tile_fill, 0, o.cap_tile_fill * sizeof(uint32_t), c->stream));
        c->side_dirty = true;
        return B32_E_HIP;
    }
    if (sticky & 16u) return B32_E_BAND_TIMEOUT;               // (b32_band_wait / _wait_all / _acquire gave up on another rank's epoch word)
    if (c->h_ctrl.wire_overflow || (sticky & 4u)) return B32_E_UNSUPPORTED;     // an edge >= 2^30 px long: i32 overflow in the reference's Bresenham
    if (out) {
        out->triangles_drawn = c->h_ctrl.n_visible;
        out->fragments = c->last_exact ? c->h_ctrl.fragments : 0;     // exact only with fragment counting on, painter's mode
        out->tile_pairs = c->h_ctrl.n_pairs;
        // RasterTimings phases of the most recent frame from the device-side phase clock (10 ns ticks): the reference's TRANSFORM, FOG and
        // CULL / SETUP stages are ONE fused kernel here (reported as cull_ms, transform_ms = fog_ms = 0), its sort is the tile binning,
        // its draw loop the fill kernels, its wireframe phase the line kernels.  With b32_set_profiling(2) the HIP-event averages over
        // the finished batch of frames take their place.
        const unsigned long long* t = c->h_stamps.t;
        const unsigned long long t_end = t[ST_END] ? t[ST_END] : 0ull;
        if (c->nf && t[ST_SETUP] && t[ST_FILL] >= t[ST_SETUP]) {
            const unsigned long long t_bin = t[ST_BIN] ? t[ST_BIN] : t[ST_FILL];
            const unsigned long long t_fill_end = t[ST_WIRE] ? t[ST_WIRE] : t_end;
            out->cull_ms = (float)(t_bin - t[ST_SETUP]) * 1e-5f;
            out->sort_ms = (float)(t[ST_FILL] - t_bin) * 1e-5f;
            if (t_fill_end >= t[ST_FILL]) out->draw_ms = (float)(t_fill_end - t[ST_FILL]) * 1e-5f;
            if (t[ST_WIRE] && t_end >= t[ST_WIRE]) out->wireframe_ms = (float)(t_end - t[ST_WIRE]) * 1e-5f;
        }
        if (c->phase_frames && c->phase_level >= 2) {
            out->cull_ms = c->phase_ms[0];
            out->sort_ms = c->phase_ms[1];
            out->draw_ms = c->phase_ms[2] + c->phase_ms[3] + c->phase_ms[4];
        }
    }
    return B32_OK;
}

}  // extern "C"

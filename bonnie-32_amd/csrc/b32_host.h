// b32_host.h -- what the translation units of the C ABI share: the context (b32_ctx), a scene slot (b32_scene), a frame set, the
// allocation helpers, and the handful of functions one unit calls in another.  Internal: nothing here is part of include/b32raster.h.
//   b32_api.hip    context, stream, framebuffer calls, test taps and switches
//   b32_scene.hip  uploads (drop-in and resident), scene slots, the synchronous drop-in calls
//   b32_frame.hip  one frame: route, enqueue, two / three frames in flight, b32_frame_finish
//   b32_batch.hip  b32_frame_begin / _add_scene / _end (merged runs)
#pragma once
#include "b32_device.h"
#include <algorithm>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <vector>

using namespace b32;
#ifndef B32_PIPELINE_BANDS
#define B32_PIPELINE_BANDS 0      // experiment switch: two frames in flight for band-sharded frames too (measured: N=8 band 0.067 -> 0.071 ms: no)
#endif
#ifndef B32_MIN_TILE_H
#define B32_MIN_TILE_H 8
#endif

namespace {
constexpr int EV_RING = 64;     // frames of per-phase events kept between two b32_frame_finish calls
constexpr int EV_PER_FRAME = 6; // start | setup | sort | bin | cover | shade+blend
}

// Everything k_setup WRITES for one frame and the fill kernels read: a context owns two of these so that the setup kernel of frame
// i + 1 can run on a second stream beside the fill of frame i (see pipeline_begin).  The context's own members of the same names are
// the set of the frame being enqueued; `alt` holds the other ones, oldest first (rotate_sets).
struct FrameSet {
    uint32_t* keys0 = nullptr; CovRec* crecs = nullptr; ShadeRec* srecs = nullptr; AuxRec* xrecs = nullptr;
    uint32_t* spans = nullptr; uint32_t* face_of = nullptr; uint32_t* partials = nullptr; size_t cap_work = 0;
    float* shades = nullptr; size_t cap_shades = 0;
    uint32_t* direct_lists = nullptr; size_t cap_direct = 0;
    uint32_t* tile_fill = nullptr; size_t cap_tile_fill = 0;
    WireTri* wire = nullptr; size_t cap_wire = 0;           // (frames with wireframe phases: k_setup writes the wire list, the wire kernels behind the fill read it)
    uint32_t *wire_fill = nullptr, *wire_lists = nullptr; size_t cap_wire_tiles = 0; unsigned long long wire_grid = 0;   // (... and its tile lists: binned beside the previous frame's fill)
    Ctrl* d_ctrl = nullptr;
    hipEvent_t ev_setup = nullptr, ev_done = nullptr;      // k_setup finished (side stream) / last fill reading this set finished (main stream)
    bool in_flight = false;                                  // a frame was enqueued on this set since the last b32_frame_finish
};

struct b32_ctx {
    int device = 0;
    int n_cu = 256;
    int last_hip = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    // two frames in flight: the setup kernel of the next frame on `side` beside the fill of the current one on `stream`
    hipStream_t side = nullptr; hipEvent_t ev_main = nullptr, ev_wbin = nullptr;     // (ev_wbin: a pipelined frame's k_wire_bin finished on the side stream)
    FrameSet alt[2];                     // the other frame sets, oldest first (allocated on first use; alt[1] only with three sets)
    uint32_t n_sets_user = 0;            // what b32_set_pipeline_depth asked for (0: the library's own choice, see auto_depth in b32_api.hip)
    uint32_t n_sets = 2;                 // b32_set_pipeline_depth: 2 = setup(i+1) beside fill(i); 3 = setup(i+2) beside fill(i), so that the
                                         // setup kernel a fill waits for ended a whole fill ago (fills back to back; measured slower: the two
                                         // kernels then share the CUs all the time and the frame is bound by their summed VALU work)
    hipEvent_t ev_setup = nullptr, ev_done = nullptr; bool set_in_flight = false;     // (members of the current set, see FrameSet)
    bool side_dirty = true;              // something k_setup reads was written on `stream` since the side stream last waited for it
    hipStream_t join_stream = nullptr; bool join_ok = false;   // k_flag / k_join instead of an event: only while `stream` and `side` have DIFFERENT priorities (then they never share a hardware queue); checked per main stream
    uint32_t gate_permille = 0;          // b32_set_pipeline_gate: hold the next setup kernel until the previous fill's tile cursor has got this far (0: only until that fill has started -- the frame sets' order, k_gate)
    uint32_t wire_seq = 0;               // WireArgs::epoch of the last frame with wireframe phases (never 0)
    bool start_lost = false;             // fault injection: the last fused kernel was told not to publish its start (the next start gate waits 2 ms only)
    uint32_t join_seq = 0;               // FillArgs::join_seq of the last polled hand-over (never 0)
    uint32_t fill_seq = 0;               // FillArgs::start_seq of the last fused kernel launched (never 0)
    bool pipe_hint = true;               // the previous frame's route could use the second frame set
    struct CoverOf { const Ctrl* ctrl; uint32_t tiles, groups, seq; } cover_of[3] = {};   // the same per frame set (keyed by its control block): the gate polls the
                                                                                      // cursor of the frame n_sets - 1 back, whose grid may differ from the previous frame's
    uint32_t last_cover_tiles = 0, last_cover_groups = 0;   // tile count / workgroups of the previous frame's fused kernel (0: it had none)
    bool pipelined = false;              // the frame being enqueued runs its k_setup on the side stream
    unsigned long long pipelined_frames = 0;
    unsigned long long flag_join_frames = 0, event_join_frames = 0, poll_join_frames = 0;    // ... handed over to the fill by k_flag / k_join, or by a cross-stream event
    uint32_t inject = 0;                 // b32_debug_inject: fault injection for the tests of the failure paths

    // framebuffer
    uint32_t width = 0, height = 0;
    uint32_t* fb_own = nullptr; size_t fb_own_px = 0;
    uint32_t* fb = nullptr; bool fb_external = false;
    uint32_t band_y0 = 0, band_y1 = 0; bool band_set = false;
    float* zbuf = nullptr; size_t cap_zbuf = 0; bool zbuf_valid = false;   // Framebuffer::zbuffer; !valid == every entry f32::MAX

    // multi-GPU band exchange (b32_gather.hip): mappings of the root's framebuffer / epoch words (band rank), or the root's own epoch words
    void* band_fb_ipc = nullptr; uint32_t* band_sync_own = nullptr; uint32_t* band_sync = nullptr;      // (band_sync_own: inside fb_own's tail)
    uint32_t band_rank = 0; bool band_attached = false;

    // resident scene
    B32Vertex* d_verts = nullptr; size_t cap_verts = 0;
    B32Face* d_faces = nullptr; size_t cap_faces = 0;
    uint16_t* d_texels = nullptr; size_t cap_texels = 0;
    uint32_t* d_texels32 = nullptr; size_t cap_texels32 = 0;   // 8-bit-colour path: Color texels
    bool fmt8 = false;                  // the resident scene was uploaded by b32_scene_upload_rgba (render_mesh path)
    bool blend8 = false;                // 8-bit path: some texel blends or some face has editor_alpha < 255 -> ordered walk
    TexDesc* d_tex = nullptr; size_t cap_tex = 0;
    std::vector<TexDesc> h_tex;
    uint8_t* d_atlas0 = nullptr; size_t cap_atlas0 = 0; uint32_t atlas_idx_bytes = 0;   // indexed upload of ONE texture: CLUT (512 B) + index bytes, kept for the LDS route
    uint32_t* d_texmask = nullptr; size_t cap_texmask = 0;       // skip mask of the texel pool (FillArgs.texmask), rebuilt when the pool changes
    uint32_t pool_texels = 0; bool mask_dirty = true;
    uint32_t nv = 0, nf = 0, nt = 0;
    bool have_scene = false;
    unsigned long long gen = 0;         // identity of the resident scene's content (every upload gets a new number; swapped with the slots)
    bool may_blend = true;              // some face / texture can produce a transparent-pass surface (render.rs:2403-2415)
    bool cheap_ok = false;              // every texture has few skippable texels: CHEAP coverage + repair is profitable
    bool tex_blend_any = false;         // some texture of the resident scene has a blend mode other than Opaque
    uint32_t blend_faces = 0;           // faces that their own blend mode / editor alpha or their texture's blend mode puts in the transparent pass
    // Texture cache of the drop-in calls ("texture upload may be cached by (ptr,len,hash) but must be semantically per-call"):
    // what the texel pool currently holds -- per texture the caller's pointer, its dimensions, blend mode and a 64-bit hash of its
    // content.  A call that passes the same set again (the reference's callers pass the same Texture15 slice every frame) skips the
    // texel copies and the skippable-texel count; any change of pointer, size or content re-uploads.
    struct TexSig { const void* ptr; uint32_t w, h, blend; uint64_t hash; };
    std::vector<TexSig> tex_sig; bool tex_sig_valid = false;
    int count_fragments = 0;            // 1: exact fragment-store count every frame (EXACT coverage); instrumentation, off by default
    bool last_exact = false;            // the last frame ran EXACT coverage in painter's mode (B32Timings.fragments is exact)

    // per-face work buffers
    size_t cap_work = 0;
    uint32_t *keys[2] = { nullptr, nullptr }, *vals[2] = { nullptr, nullptr };
    CovRec* crecs = nullptr; ShadeRec* srecs = nullptr; AuxRec* xrecs = nullptr;      // per-face records (b32_device.h)
    float* shades = nullptr; size_t cap_shades = 0;
    uint32_t* counts = nullptr; uint32_t* block_sums = nullptr; uint32_t bin_blocks = 0;
    uint32_t* spans = nullptr;
    uint32_t* face_of = nullptr;        // record slot -> face id (k_setup packs each wave's survivors to the front of its 64 slots)
    uint32_t* tile_mid = nullptr; size_t cap_tile_mid = 0;
    bool local_sort_ok = true;          // no tile list of this scene has exceeded the LDS sort capacity so far
    bool last_local_sort = false;       // the last frame took the fast path (draw order not materialised)
    uint32_t route_off = 0;             // b32_set_routes: B32_ROUTE_* bits switched off (tests keep the older pipelines covered with it)
    uint32_t cheap_den = 64;            // b32_set_cheap_threshold
    // pairs
    size_t cap_pairs = 0;
    uint32_t* inline_lists = nullptr; size_t cap_inline = 0;      // small meshes: one list region per tile, filled inside k_cover
    // direct binning (DirectBin, b32_device.h): k_setup appends to fixed tile regions; the regions grow when a frame overflowed one
    uint32_t* direct_lists = nullptr; size_t cap_direct = 0;
    uint32_t* tile_fill = nullptr; size_t cap_tile_fill = 0;      // FILL_PAD words per tile, zero between frames
    // packed vertex streams of a resident mesh (k_pack_streams: nv positions of 12 B, then nv (u, v, rgba) of 12 B): built on the second
    // frame of an uploaded mesh too large for the in-kernel list collection
    float* d_pos12 = nullptr; size_t cap_pos12 = 0; bool pos_valid = false; uint32_t band_frames = 0;
    bool lit_valid = false;             // ... and the 24-byte lit stream behind them (packed on the first frame with a shading pass)
    // (per scene, swapped with the scene slots:)
    uint32_t direct_cap_opaque = 0;                               // opaque entries per tile region (0: sized from the mesh on first use)
    uint32_t direct_ntiles = 0;                                   // the tile grid that size belongs to (another grid: sized again)
    bool direct_ok = true;                                        // false: the regions would not fit (one tile's list too long) -> counting sort
    bool last_direct = false;
    uint32_t epoch = 0;
    // Framebuffer::clear deferred (b32_fb_clear): applied by the next frame's fused kernel when that frame takes the sort-free path in
    // painter's mode on the same band, else by a clear launch before whatever touches the framebuffer next (flush_clear)
    bool clear_pending = false; uint32_t clear_rgba = 0, clear_y0 = 0, clear_y1 = 0;
    unsigned long long routes[8] = {};                            // b32_route_count
    unsigned long long lds_atlas_frames = 0;
    uint32_t *pkeys[2] = { nullptr, nullptr }, *pvals[2] = { nullptr, nullptr };
    // sort scratch
    uint32_t* block_hist = nullptr; uint32_t hist_blocks = 0; uint32_t* digit_total = nullptr;
    uint32_t* partials = nullptr; uint32_t partial_blocks = 0;
    // tiles
    uint32_t* ranges = nullptr; size_t cap_ranges = 0;
    uint32_t* vis = nullptr; size_t cap_vis = 0;
    // wireframe phases (allocated on first use)
    WireTri* wire = nullptr; size_t cap_wire = 0;
    uint32_t *wire_owner = nullptr, *wire_first = nullptr; size_t cap_wire_table = 0;
    uint32_t *wire_fill = nullptr, *wire_lists = nullptr; size_t cap_wire_tiles = 0;     // tile route of the wireframe phases (WireArgs)
    unsigned long long wire_grid = 0;                                                       // tile grid the (self-resetting) counters belong to
    unsigned long long wire_tile_frames = 0;
    unsigned long long span_cover_frames = 0;                     // frames whose opaque coverage used exact row intervals (B32_ROUTE_SPAN_COVER)
    // control
    Ctrl* d_ctrl = nullptr; uint32_t* d_consts = nullptr; Ctrl h_ctrl{}; Stamps h_stamps{};   // (d_ctrl: Ctrl followed by Stamps)
    uint32_t h_consts[4] = { 0, 0, 0, 0 };   // staging for d_consts (outlives the async copy)
    bool defer_upload_sync = false;            // drop-in calls: the frame's own synchronisation covers the uploads
    // staged upload of the drop-in calls (see UploadSegs): the caller's slices are packed into a pinned arena on the host and one
    // kernel moves them; active only inside b32_render_mesh[_15], which always synchronise before they return
    unsigned char* stage_host = nullptr; void* stage_dev = nullptr; size_t stage_cap = 0, stage_used = 0;
    bool stage_active = false, stage_failed = false; UploadSegs stage_segs{};
    B32Light* d_lights = nullptr; size_t cap_lights = 0; std::vector<B32Light> h_lights;

    // batched frame (b32_frame_begin / _add_scene / _end): per-mesh rows of the frame being enqueued (kept for a redraw), the recording
    // between begin and end, and the merged meshes built so far (reused while their members' contents stay the same)
    bool frame_batched = false; MeshTable frame_table{};
    struct BatchEntry { b32_scene* slot; MeshRow row; bool wire; };
    struct MergedRun { std::vector<b32_scene*> members; std::vector<unsigned long long> gens; b32_scene* merged = nullptr; unsigned long long used = 0; };
    bool batch_open = false; B32Camera batch_cam{}; B32Settings batch_st{}; std::vector<B32Light> batch_lights; std::vector<BatchEntry> batch;
    std::vector<MergedRun> merged_runs; unsigned long long batch_clock = 0, gen_counter = 0;
    unsigned long long batch_stats[4] = {};      // merged draws, sequential draws, merged meshes built, frames
    // last enqueued frame (for redraw after a pair overflow)
    bool frame_pending = false;
    bool pending_superseded = false;    // safe mode: a b32_fb_clear of the whole band was issued behind the pending frame -- every pixel (and depth) that frame
                                        // drew is overwritten, so the NEXT frame may be enqueued without settling it (a redraw of it could not be seen)
    bool pending_may_redraw = false;    // the pending frame took a path that can overflow its buffers (not the small-mesh path)
    bool deep_async = false;            // b32_set_async_depth(1): large-scene frames are enqueued back to back, a dropped one is reported
    bool redrawing = false;             // enqueue_frame is repeating the pending frame (k_setup must not count it as lost)
    int deferred_rc = 0;                // error of a frame that b32_scene_swap had to settle: reported by the next b32_frame_finish
    B32Camera last_cam{}; B32Settings last_settings{}; B32Fog last_fog{}; bool last_has_fog = false;
    int last_pair_buf = 0;

    // asynchronous framebuffer downloads (b32_fb_download_async): ticket t completes with event dl_ev[t % DL_RING]
    static constexpr uint32_t DL_RING = 8;
    hipEvent_t dl_ev[DL_RING] = {}; unsigned long long dl_seq = 0;
    // ... through a device-side snapshot: the frame is copied to dl_stage[t % 2] on the context's stream (microseconds) and leaves for the host
    // from there on dl_stream, so that the NEXT frame's kernels do not wait for the PCIe transfer
    hipStream_t dl_stream = nullptr; hipEvent_t dl_snap[2] = {}; uint32_t* dl_stage[2] = {}; size_t dl_stage_px = 0;
    // profiling
    int profile_level = 0;
    uint32_t prof_stride = 1, prof_seq = 0;      // b32_set_profiling_stride: events on every prof_stride-th frame only
    hipEvent_t ev[EV_RING][EV_PER_FRAME] = {};
    bool ev_created = false;
    uint32_t ev_frames = 0;             // frames recorded since the last finish
    float phase_ms[5] = { 0, 0, 0, 0, 0 }; // averages of the last finished batch: setup, sort, bin, cover, shade
    uint32_t phase_frames = 0;
    int phase_level = 0;                // profiling level those averages were taken at
    std::vector<B32Light> keep_lights;  // private copy of the last frame's lights (redraw after overflow)
};

// A slot of b32_scene_swap: everything of b32_ctx that belongs to ONE uploaded scene.
struct b32_scene {
    B32Vertex* d_verts = nullptr; size_t cap_verts = 0;
    B32Face* d_faces = nullptr; size_t cap_faces = 0;
    uint16_t* d_texels = nullptr; size_t cap_texels = 0;
    uint32_t* d_texels32 = nullptr; size_t cap_texels32 = 0;
    TexDesc* d_tex = nullptr; size_t cap_tex = 0;
    uint32_t* d_consts = nullptr;
    uint32_t* d_texmask = nullptr; size_t cap_texmask = 0; uint32_t pool_texels = 0; bool mask_dirty = true;
    uint8_t* d_atlas0 = nullptr; size_t cap_atlas0 = 0; uint32_t atlas_idx_bytes = 0;
    std::vector<TexDesc> h_tex;
    uint32_t nv = 0, nf = 0, nt = 0;
    unsigned long long gen = 0;
    uint32_t blend_faces = 0;
    bool fmt8 = false, blend8 = false, have_scene = false, may_blend = true, cheap_ok = false, local_sort_ok = true, tex_blend_any = false;
    uint32_t direct_cap_opaque = 0, direct_ntiles = 0; bool direct_ok = true;
    float* d_pos12 = nullptr; size_t cap_pos12 = 0; bool pos_valid = false; uint32_t band_frames = 0; bool lit_valid = false;
    std::vector<b32_ctx::TexSig> tex_sig; bool tex_sig_valid = false;
};

#define HIPCHK(ctx, expr)                                                 \
    do {                                                                  \
        hipError_t _e = (expr);                                           \
        if (_e != hipSuccess) { (ctx)->last_hip = (int)_e; return B32_E_HIP; } \
    } while (0)

template <typename T>
static int ensure(b32_ctx* c, T*& p, size_t& cap, size_t need) {
    if (need <= cap && p) return B32_OK;
    c->side_dirty = true;
    if (p) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(p)); p = nullptr; cap = 0; }
    size_t n = need + need / 4 + 16;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T)));
    cap = n;
    return B32_OK;
}
template <typename T>
static int ensure_plain(b32_ctx* c, T*& p, size_t count) {   // exact-size (re)allocation without capacity tracking
    c->side_dirty = true;
    if (p) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(p)); p = nullptr; }
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T)));
    return B32_OK;
}



// Scratch device allocations of ONE API call (sky / stars / present / taps): released on every exit path, error returns included,
// after the stream has drained.
struct Scratch {
    b32_ctx* c;
    std::vector<void*> ptrs;
    explicit Scratch(b32_ctx* ctx) : c(ctx) {}
    Scratch(const Scratch&) = delete;
    Scratch& operator=(const Scratch&) = delete;
    ~Scratch() {
        if (ptrs.empty()) return;
        (void)hipStreamSynchronize(c->stream);
        for (void* q : ptrs) (void)hipFree(q);
    }
    template <typename T>
    int alloc(T** out, size_t count) {
        void* q = nullptr;
        *out = nullptr;
        HIPCHK(c, hipMalloc(&q, (count ? count : 1) * sizeof(T)));
        ptrs.push_back(q);
        *out = static_cast<T*>(q);
        return B32_OK;
    }
    template <typename T>
    int upload(const T* host, size_t n, T** dev) {       // scratch copy of a small per-call input
        int rc = alloc(dev, n);
        if (rc || !n) return rc;
        HIPCHK(c, hipMemcpyAsync(*dev, host, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
        return B32_OK;
    }
};

// pinned host arena of the drop-in calls (stage_ensure, b32_scene.hip): packed uploads, and the frame's Ctrl + Stamps on the way back
constexpr size_t STAGE_BYTES = (size_t)1 << 20, STAGE_CTRL_OFF = STAGE_BYTES - 128;   // the last 128 B receive the frame's Ctrl + Stamps

// ---- shared between the units (defined in the file named); C linkage like the ABI around them, but not exported
#define B32_INTERNAL __attribute__((visibility("hidden")))
extern "C" {
B32_INTERNAL int settle_pending(b32_ctx* c);                                      // b32_api.hip
B32_INTERNAL int flush_clear(b32_ctx* c);                                         // b32_api.hip
B32_INTERNAL int apply_depth_auto(b32_ctx* c);                                    // b32_api.hip (b32_set_pipeline_depth(ctx, 0): the depth the library picks)
constexpr size_t FB_TAIL_BYTES = 8192;      // behind the pixels of a library-owned framebuffer: the epoch words of the band exchange (b32_gather.hip)
B32_INTERNAL void band_close_any(b32_ctx* c);                                    // b32_gather.hip
B32_INTERNAL void free_alt(b32_ctx* c, FrameSet& a);                              // b32_frame.hip
B32_INTERNAL int h2d(b32_ctx* c, void* dst, const void* src, size_t bytes);       // b32_scene.hip
B32_INTERNAL bool stage_ensure(b32_ctx* c);                                       // b32_scene.hip
B32_INTERNAL int ensure_work(b32_ctx* c, uint32_t nf);                            // b32_scene.hip
B32_INTERNAL int validate_settings(const B32Settings* st);                        // b32_frame.hip
B32_INTERNAL int enqueue_frame(b32_ctx* c, const B32Camera* cam, const B32Settings* st, const B32Fog* fog);            // b32_frame.hip
B32_INTERNAL int render_scene_async_any(b32_ctx* c, const B32Camera* cam, const B32Settings* st, const B32Fog* fog);   // b32_frame.hip
}

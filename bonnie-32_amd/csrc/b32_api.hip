// b32_api.hip — the C ABI of include/b32raster.h: context, device-resident framebuffer and scene, frame enqueue.
//
// Frame = memset(ctrl) -> k_setup -> k_after_setup -> 4 radix passes (painter's order) -> k_bin_count/scan/emit ->
// radix passes on (tile,class) -> k_tile_ranges -> k_fill.  Everything is enqueued on one HIP stream without host
// round trips; counts that decide later grid work (surfaces, pairs) stay in device memory.  The only host readback is
// b32_frame_finish (error flags, triangles_drawn, fragment count, pair-capacity overflow -> grow and redraw).
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
    hipStream_t side = nullptr; hipEvent_t ev_main = nullptr;
    FrameSet alt[2];                     // the other frame sets, oldest first (allocated on first use; alt[1] only with three sets)
    uint32_t n_sets = 2;                 // b32_set_pipeline_depth: 2 = setup(i+1) beside fill(i); 3 = setup(i+2) beside fill(i), so that the
                                         // setup kernel a fill waits for ended a whole fill ago (fills back to back; measured slower: the two
                                         // kernels then share the CUs all the time and the frame is bound by their summed VALU work)
    hipEvent_t ev_setup = nullptr, ev_done = nullptr; bool set_in_flight = false;     // (members of the current set, see FrameSet)
    bool side_dirty = true;              // something k_setup reads was written on `stream` since the side stream last waited for it
    uint32_t gate_permille = 1150;       // b32_set_pipeline_gate: hold the next setup kernel until the previous fill has handed out 15 % of the tiles behind its first round
    bool pipe_hint = true;               // the previous frame's route could use the second frame set
    uint32_t last_cover_tiles = 0, last_cover_groups = 0;   // tile count / workgroups of the previous frame's fused kernel (0: it had none)
    bool pipelined = false;              // the frame being enqueued runs its k_setup on the side stream
    unsigned long long pipelined_frames = 0;

    // framebuffer
    uint32_t width = 0, height = 0;
    uint32_t* fb_own = nullptr; size_t fb_own_px = 0;
    uint32_t* fb = nullptr; bool fb_external = false;
    uint32_t band_y0 = 0, band_y1 = 0; bool band_set = false;
    float* zbuf = nullptr; size_t cap_zbuf = 0; bool zbuf_valid = false;   // Framebuffer::zbuffer; !valid == every entry f32::MAX

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
    bool pending_may_redraw = false;    // the pending frame took a path that can overflow its buffers (not the small-mesh path)
    bool deep_async = false;            // b32_set_async_depth(1): large-scene frames are enqueued back to back, a dropped one is reported
    bool redrawing = false;             // enqueue_frame is repeating the pending frame (k_setup must not count it as lost)
    int deferred_rc = 0;                // error of a frame that b32_scene_swap had to settle: reported by the next b32_frame_finish
    B32Camera last_cam{}; B32Settings last_settings{}; B32Fog last_fog{}; bool last_has_fog = false;
    int last_pair_buf = 0;

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
    float* d_pos12 = nullptr; size_t cap_pos12 = 0; bool pos_valid = false; uint32_t band_frames = 0;
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


// ------------------------------------------------------------------ two frames in flight
// The fused fill kernel leaves most CUs idle in its last fifth (the tile queue's tail), and k_setup of the NEXT frame needs nothing
// from it: with two frame sets (everything k_setup writes, FrameSet) the setup kernel of frame i + 1 runs on a second stream beside the
// fill of frame i.  Orders kept by events: setup(i) -> fill(i) (ev_setup), fill(i) -> setup(i + 2) on the same set (ev_done), and
// anything enqueued on the main stream that k_setup reads (uploads, packed streams, light lists, list-space memsets) -> the next
// setup (ev_main, only when `side_dirty`).  The main stream always waits for the frame's setup before enqueue_frame returns, so a
// synchronisation of the main stream still covers everything this context has in flight.
static void swap_with(b32_ctx* c, FrameSet& a) {
    std::swap(c->keys[0], a.keys0); std::swap(c->crecs, a.crecs); std::swap(c->srecs, a.srecs); std::swap(c->xrecs, a.xrecs);
    std::swap(c->spans, a.spans); std::swap(c->face_of, a.face_of); std::swap(c->partials, a.partials);
    std::swap(c->shades, a.shades); std::swap(c->cap_shades, a.cap_shades);
    std::swap(c->direct_lists, a.direct_lists); std::swap(c->cap_direct, a.cap_direct);
    std::swap(c->tile_fill, a.tile_fill); std::swap(c->cap_tile_fill, a.cap_tile_fill);
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
static void free_alt(b32_ctx* c, FrameSet& a) {          // (the caller has drained both streams)
    void* ptrs[] = { a.keys0, a.crecs, a.srecs, a.xrecs, a.spans, a.face_of, a.partials, a.shades, a.direct_lists, a.tile_fill };
    for (void* q : ptrs) if (q) (void)hipFree(q);
    a.keys0 = nullptr; a.crecs = nullptr; a.srecs = nullptr; a.xrecs = nullptr; a.spans = nullptr; a.face_of = nullptr; a.partials = nullptr;
    a.shades = nullptr; a.cap_shades = 0; a.direct_lists = nullptr; a.cap_direct = 0; a.tile_fill = nullptr; a.cap_tile_fill = 0; a.cap_work = 0;
}
// side stream, events and the other sets' per-face buffers (sized like the current set's)
static int pipeline_ensure(b32_ctx* c) {
    if (!c->side) {
        // lowest priority: while the fill kernel has workgroups to place, they go first; the setup kernel takes what is left
        int prio_least = 0, prio_greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
        HIPCHK(c, hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, prio_least));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_main, hipEventDisableTiming));
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

extern "C" {

const char* b32_strerror(int code) {
    switch (code) {
        case B32_OK: return "ok";
        case B32_E_ARG: return "invalid argument";
        case B32_E_INDEX: return "face references a vertex index out of range";
        case B32_E_NAN_KEY: return "NaN painter's-sort key";
        case B32_E_HIP: return "HIP runtime error";
        case B32_E_UNSUPPORTED: return "setting outside the supported hot-path scope";
        case B32_E_NO_DEVICE: return "no HIP device (the rasterizer has no CPU fallback)";
        case B32_E_FRAME_DROPPED: return "an earlier frame in flight ran out of buffer space and drew nothing (deep asynchronous mode)";
        default: return "unknown error";
    }
}

int b32_create(int device, b32_ctx** out) {
    if (!out) return B32_E_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return B32_E_NO_DEVICE;
    b32_ctx* c = new b32_ctx();
    c->device = device;
    if (hipSetDevice(device) != hipSuccess) { delete c; return B32_E_NO_DEVICE; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) { delete c; return B32_E_HIP; }
    c->stream = c->own_stream;
    if (hipMalloc(reinterpret_cast<void**>(&c->d_ctrl), sizeof(Ctrl) + sizeof(Stamps) + sizeof(Events)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&c->d_consts), 16 * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&c->digit_total), 4096 * sizeof(uint32_t)) != hipSuccess) { delete c; return B32_E_HIP; }
    if (hipMemset(c->d_ctrl, 0, sizeof(Ctrl) + sizeof(Stamps) + sizeof(Events)) != hipSuccess) { delete c; return B32_E_HIP; }     // (`sticky` is never reset by a frame)
    *out = c;
    return B32_OK;
}

void b32_destroy(b32_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    void* ptrs[] = { c->fb_own, c->d_verts, c->d_faces, c->d_texels, c->d_tex, c->keys[0], c->keys[1], c->vals[0], c->vals[1], c->crecs, c->srecs, c->xrecs,
                     c->shades, c->counts, c->block_sums, c->pkeys[0], c->pkeys[1], c->pvals[0], c->pvals[1], c->block_hist, c->ranges,
                     c->d_ctrl, c->d_consts, c->d_lights, c->digit_total, c->partials, c->vis, c->spans, c->tile_mid, c->zbuf,
                     c->wire, c->wire_owner, c->wire_first, c->wire_fill, c->wire_lists, c->d_texels32, c->inline_lists, c->d_texmask, c->direct_lists, c->tile_fill, c->d_pos12, c->face_of, c->d_atlas0 };
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (c->side) (void)hipStreamSynchronize(c->side);
    for (auto& r : c->merged_runs) if (r.merged) { void* mp[] = { r.merged->d_verts, r.merged->d_faces, r.merged->d_texels, r.merged->d_texels32, r.merged->d_tex,
                                                                    r.merged->d_consts, r.merged->d_texmask, r.merged->d_pos12, r.merged->d_atlas0 };
                                                   for (void* q : mp) if (q) (void)hipFree(q); delete r.merged; }
    for (FrameSet& a : c->alt) { free_alt(c, a); if (a.d_ctrl) (void)hipFree(a.d_ctrl); }
    for (hipEvent_t e : { c->ev_main, c->ev_setup, c->ev_done, c->alt[0].ev_setup, c->alt[0].ev_done, c->alt[1].ev_setup, c->alt[1].ev_done }) if (e) (void)hipEventDestroy(e);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->ev_created) for (auto& fr : c->ev) for (auto& e : fr) if (e) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->stage_host) (void)hipHostFree(c->stage_host);
    delete c;
}

int b32_last_hip_error(const b32_ctx* c) { return c ? c->last_hip : 0; }

// A pending frame that may still need a redraw (pair overflow, long transparent lists) is settled before anything reads or rebinds
// the framebuffer, so that no caller ever sees the cleared frame of an aborted attempt.  Its error, if any, is the frame's error: kept
// for the b32_frame_finish that ends the frame.
static int settle_pending(b32_ctx* c) {
    if (!c->frame_pending || !c->pending_may_redraw) return B32_OK;
    const int rc = b32_frame_finish(c, nullptr);
    if (rc == B32_E_HIP || rc == B32_E_ARG) return rc;
    if (rc && !c->deferred_rc) c->deferred_rc = rc;
    return B32_OK;
}

// The deferred Framebuffer::clear as launches of its own: before anything but the sort-free frame reads or writes the framebuffer.
static int flush_clear(b32_ctx* c) {
    if (!c->clear_pending) return B32_OK;
    c->clear_pending = false;
    if (!c->fb || c->clear_y1 <= c->clear_y0) return B32_OK;
    launch_clear(c->stream, c->fb + (size_t)c->clear_y0 * c->width, (size_t)c->width * (c->clear_y1 - c->clear_y0), c->clear_rgba);
    if (c->zbuf && c->zbuf_valid && (size_t)c->width * c->height <= c->cap_zbuf)        // self.zbuffer[i] = f32::MAX, render.rs:43
        launch_clear(c->stream, reinterpret_cast<uint32_t*>(c->zbuf) + (size_t)c->clear_y0 * c->width, (size_t)c->width * (c->clear_y1 - c->clear_y0), 0x7F7FFFFFu);
    HIPCHK(c, hipGetLastError());
    return B32_OK;
}

int b32_set_stream(b32_ctx* c, void* s) {
    if (!c) return B32_E_ARG;
    { const int rc = settle_pending(c); if (rc) return rc; }
    { const int rc = flush_clear(c); if (rc) return rc; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->stream = s ? reinterpret_cast<hipStream_t>(s) : c->own_stream;
    c->side_dirty = true;
    return B32_OK;
}
int b32_synchronize(b32_ctx* c) {
    if (!c) return B32_E_ARG;
    { const int rc = flush_clear(c); if (rc) return rc; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return B32_OK;
}

// ------------------------------------------------------------------ framebuffer
static int fb_set_dims(b32_ctx* c, uint32_t w, uint32_t h) {
    c->width = w; c->height = h;
    if (!c->band_set) { c->band_y0 = 0; c->band_y1 = h; }
    else { if (c->band_y1 > h) c->band_y1 = h; if (c->band_y0 > c->band_y1) c->band_y0 = c->band_y1; }
    return B32_OK;
}
static int fb_resize_any(b32_ctx* c, uint32_t w, uint32_t h, bool always_new) {
    if (!c || w == 0 || h == 0 || w > 16384 || h > 16384) return B32_E_ARG;
    (void)hipSetDevice(c->device);
    { const int rc = settle_pending(c); if (rc) return rc; }
    { const int rcf = flush_clear(c); if (rcf) return rcf; }
    if (c->fb_external) { c->fb_external = false; c->fb = nullptr; c->width = c->height = 0; }
    if (!always_new && c->fb && c->width == w && c->height == h) return B32_OK;          // Framebuffer::resize: no-op on equal dims
    const size_t px = (size_t)w * h;
    if (px > c->fb_own_px || !c->fb_own) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->fb_own) HIPCHK(c, hipFree(c->fb_own));
        c->fb_own = nullptr;
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->fb_own), px * 4));
        c->fb_own_px = px;
    }
    c->fb = c->fb_own;
    c->band_set = false;
    c->zbuf_valid = false;                                                  // vec![f32::MAX; w*h], render.rs:22,32
    fb_set_dims(c, w, h);
    HIPCHK(c, hipMemsetAsync(c->fb, 0, px * 4, c->stream));                // vec![0; w*h*4], render.rs:18-33
    return B32_OK;
}
int b32_fb_resize(b32_ctx* c, uint32_t w, uint32_t h) { return fb_resize_any(c, w, h, false); }   // Framebuffer::resize, render.rs:27-34
int b32_fb_new(b32_ctx* c, uint32_t w, uint32_t h) { return fb_resize_any(c, w, h, true); }       // Framebuffer::new, render.rs:18-25
int b32_fb_bind_device(b32_ctx* c, void* dptr, uint32_t w, uint32_t h) {
    if (!c) return B32_E_ARG;
    { const int rc = settle_pending(c); if (rc) return rc; }
    { const int rcf = flush_clear(c); if (rcf) return rcf; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (!dptr) { c->fb_external = false; c->fb = nullptr; c->width = c->height = 0; return B32_OK; }
    if (w == 0 || h == 0 || w > 16384 || h > 16384 || (reinterpret_cast<uintptr_t>(dptr) & 15)) return B32_E_ARG;
    c->fb = reinterpret_cast<uint32_t*>(dptr); c->fb_external = true;
    c->band_set = false;
    c->zbuf_valid = false;
    return fb_set_dims(c, w, h);
}
int b32_fb_size(const b32_ctx* c, uint32_t* w, uint32_t* h) {
    if (!c) return B32_E_ARG;
    if (w) *w = c->width;
    if (h) *h = c->height;
    return B32_OK;
}
int b32_set_band(b32_ctx* c, uint32_t y0, uint32_t y1) {
    if (!c || !c->fb || y0 > y1 || y1 > c->height) return B32_E_ARG;
    { const int rcs = settle_pending(c); if (rcs) return rcs; }       // (a redraw of the pending frame belongs to the band it was enqueued for)
    { const int rcf = flush_clear(c); if (rcf) return rcf; }          // (a deferred clear belongs to the rows of the band it was issued for)
    c->band_y0 = y0; c->band_y1 = y1; c->band_set = !(y0 == 0 && y1 == c->height);
    return B32_OK;
}
// (safe mode) a pending large-scene frame that may still need a redraw is settled before anything else WRITES the framebuffer too:
// redrawn after a clear or a sky pass it would put its pixels on top of them
static int settle_before_write(b32_ctx* c) { return c->deep_async ? B32_OK : settle_pending(c); }

int b32_fb_clear(b32_ctx* c, uint8_t r, uint8_t g, uint8_t b, uint8_t blend) {
    if (!c || !c->fb) return B32_E_ARG;
    (void)hipSetDevice(c->device);
    { const int rc = settle_before_write(c); if (rc) return rc; }
    const uint32_t a = blend == B32_BLEND_ERASE ? 0u : 255u;               // Color::to_bytes, types.rs:829-832
    const uint32_t rgba = r | (g << 8) | (b << 16) | (a << 24);
    // with a screen band set (multi-GPU sharding) only the rows this rank owns are cleared: the others belong to other ranks.
    // The clear is DEFERRED: the frame that follows folds it into its fused kernel (no clear launch, uncovered pixels written once);
    // anything else that touches the framebuffer first turns it into the launches it replaces (flush_clear).  An earlier deferred
    // clear of the same rows is dead (fully overwritten); of other rows, it is flushed.
    if (c->clear_pending && (c->clear_y0 != c->band_y0 || c->clear_y1 != c->band_y1)) { const int rc = flush_clear(c); if (rc) return rc; }
    c->clear_pending = true; c->clear_rgba = rgba; c->clear_y0 = c->band_y0; c->clear_y1 = c->band_y1;
    return B32_OK;
}
int b32_fb_clear_gradient(b32_ctx* c, uint8_t r0, uint8_t g0, uint8_t b0, uint8_t blend0, uint8_t r1, uint8_t g1, uint8_t b1, uint8_t blend1) {
    if (!c || !c->fb) return B32_E_ARG;
    (void)hipSetDevice(c->device);
    { const int rc = settle_before_write(c); if (rc) return rc; }
    { const int rcf = flush_clear(c); if (rcf) return rcf; }
    const bool z = c->zbuf && c->zbuf_valid && (size_t)c->width * c->height <= c->cap_zbuf;
    launch_clear_gradient(c->stream, c->fb, z ? c->zbuf : nullptr, c->width, c->height, c->band_y0, c->band_y1,
                          r0 | (g0 << 8) | (b0 << 16) | ((uint32_t)blend0 << 24), r1 | (g1 << 8) | (b1 << 16) | ((uint32_t)blend1 << 24));
    HIPCHK(c, hipGetLastError());
    return B32_OK;
}
int b32_fb_clear_transparent(b32_ctx* c) { return b32_fb_clear(c, 0, 0, 0, B32_BLEND_ERASE); }    // [0,0,0,0] + zbuffer = f32::MAX

int b32_render_skybox_mesh(b32_ctx* c, const B32SkyVertex* v, uint32_t nv, const uint32_t* faces, uint32_t nf, const B32Camera* cam) {
    if (!c || !c->fb || !cam || (nv && !v) || (nf && !faces)) return B32_E_ARG;
    if (!nv || !nf) return B32_OK;
    (void)hipSetDevice(c->device);
    { const int rcs = settle_before_write(c); if (rcs) return rcs; }
    { const int rcf = flush_clear(c); if (rcf) return rcf; }
    for (size_t i = 0; i < (size_t)3 * nf; ++i) if (faces[i] >= nv) return B32_E_INDEX;    // projected[face[k]] index panic
    B32SkyVertex* dv = nullptr; uint32_t* df = nullptr; float2* dp = nullptr;
    Scratch tmp(c);
    int rc;
    if ((rc = tmp.upload(v, (size_t)nv, &dv))) return rc;
    if ((rc = tmp.upload(faces, (size_t)nf * 3, &df))) return rc;
    if ((rc = tmp.alloc(&dp, (size_t)nv))) return rc;
    launch_sky(c->stream, dv, nv, df, nf, *cam, dp, c->fb, c->width, c->height, c->band_y0, c->band_y1);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return B32_OK;
}
int b32_draw_star_diamonds(b32_ctx* c, const int32_t* cx, const int32_t* cy, const uint8_t* rgb, uint32_t n, float size) {
    if (!c || !c->fb || (n && (!cx || !cy || !rgb))) return B32_E_ARG;
    if (!n) return B32_OK;
    (void)hipSetDevice(c->device);
    { const int rcs = settle_before_write(c); if (rcs) return rcs; }
    { const int rcf = flush_clear(c); if (rcf) return rcf; }
    int32_t *dx = nullptr, *dy = nullptr; uint8_t* dc = nullptr;
    Scratch tmp(c);
    int rc;
    if ((rc = tmp.upload(cx, (size_t)n, &dx))) return rc;
    if ((rc = tmp.upload(cy, (size_t)n, &dy))) return rc;
    if ((rc = tmp.upload(rgb, (size_t)n * 3, &dc))) return rc;
    launch_stars(c->stream, dx, dy, dc, n, size, c->fb, c->width, c->height, c->band_y0, c->band_y1);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return B32_OK;
}
int b32_present_nearest(b32_ctx* c, uint32_t dw, uint32_t dh, uint8_t* out) {
    if (!c || !c->fb || !out || !dw || !dh || dw > 32768 || dh > 32768) return B32_E_ARG;
    (void)hipSetDevice(c->device);
    { const int rc = settle_pending(c); if (rc) return rc; }
    { const int rcf = flush_clear(c); if (rcf) return rcf; }
    uint32_t* dd = nullptr;
    Scratch tmp(c);
    int rc;
    if ((rc = tmp.alloc(&dd, (size_t)dw * dh))) return rc;
    launch_upscale_nearest(c->stream, c->fb, c->width, c->height, dd, dw, dh);
    HIPCHK(c, hipMemcpyAsync(out, dd, (size_t)dw * dh * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return B32_OK;
}
int b32_fb_upload(b32_ctx* c, const uint8_t* rgba) {
    if (!c || !c->fb || !rgba) return B32_E_ARG;
    (void)hipSetDevice(c->device);
    { const int rc = settle_pending(c); if (rc) return rc; }
    { const int rcf = flush_clear(c); if (rcf) return rcf; }
    HIPCHK(c, hipMemcpyAsync(c->fb, rgba, (size_t)c->width * c->height * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return B32_OK;
}
int b32_zbuffer_download(b32_ctx* c, float* z) {
    if (!c || !c->fb || !z) return B32_E_ARG;
    (void)hipSetDevice(c->device);
    { const int rc = settle_pending(c); if (rc) return rc; }
    { const int rcf = flush_clear(c); if (rcf) return rcf; }
    const size_t px = (size_t)c->width * c->height;
    if (!c->zbuf || !c->zbuf_valid) { HIPCHK(c, hipStreamSynchronize(c->stream)); for (size_t i = 0; i < px; ++i) z[i] = 3.40282347e+38f; return B32_OK; }
    HIPCHK(c, hipMemcpyAsync(z, c->zbuf, px * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return B32_OK;
}
int b32_zbuffer_upload(b32_ctx* c, const float* z) {
    if (!c || !c->fb || !z) return B32_E_ARG;
    (void)hipSetDevice(c->device);
    { const int rc = settle_pending(c); if (rc) return rc; }
    { const int rcf = flush_clear(c); if (rcf) return rcf; }
    const size_t px = (size_t)c->width * c->height;
    int rc;
    if (px > c->cap_zbuf || !c->zbuf) { if ((rc = ensure_plain(c, c->zbuf, px + 64))) return rc; c->cap_zbuf = px; }
    HIPCHK(c, hipMemcpyAsync(c->zbuf, z, px * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->zbuf_valid = true;
    return B32_OK;
}
int b32_fb_download(b32_ctx* c, uint8_t* rgba) {
    if (!c || !c->fb || !rgba) return B32_E_ARG;
    (void)hipSetDevice(c->device);
    { const int rc = settle_pending(c); if (rc) return rc; }
    { const int rcf = flush_clear(c); if (rcf) return rcf; }
    HIPCHK(c, hipMemcpyAsync(rgba, c->fb, (size_t)c->width * c->height * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return B32_OK;
}

// ------------------------------------------------------------------ scene upload
// Host -> device copy of an upload.  Inside a drop-in call (stage_active) the bytes are packed into the pinned arena and moved later
// by one kernel (stage_flush); a copy that does not fit, or any other caller, takes the stream's ordinary async copy.  The arena copy
// rounds the length up to 16 B: every destination has at least 15 B of slack (ensure() allocates one element more than asked for, texel
// offsets are multiples of 16 B).
static int h2d(b32_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!bytes) return B32_OK;
    c->side_dirty = true;
    const size_t padded = (bytes + 15) & ~(size_t)15;
    if (c->stage_active && c->stage_segs.count < 16 && c->stage_used + padded <= c->stage_cap && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        std::memcpy(c->stage_host + c->stage_used, src, bytes);
        const uint32_t k = c->stage_segs.count++;
        c->stage_segs.dst[k] = dst; c->stage_segs.src_off[k] = (uint32_t)c->stage_used; c->stage_segs.n16[k] = (uint32_t)(padded >> 4);
        c->stage_used += padded;
        return B32_OK;
    }
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    return B32_OK;
}
constexpr size_t STAGE_BYTES = (size_t)1 << 20, STAGE_CTRL_OFF = STAGE_BYTES - 128;   // the last 128 B receive the frame's Ctrl + Stamps
static_assert(sizeof(Ctrl) == 64 && sizeof(Stamps) == 64, "Ctrl and Stamps are read back through a 128-byte slot of the pinned arena");
static bool stage_ensure(b32_ctx* c) {
    if (!c->stage_host && !c->stage_failed) {
        void* h = nullptr;
        c->stage_failed = true;
        if (hipHostMalloc(&h, STAGE_BYTES, hipHostMallocDefault) == hipSuccess) {
            void* d = nullptr;
            if (hipHostGetDevicePointer(&d, h, 0) == hipSuccess && d) {
                c->stage_host = static_cast<unsigned char*>(h); c->stage_dev = d; c->stage_cap = STAGE_CTRL_OFF; c->stage_failed = false;
            } else (void)hipHostFree(h);
        }
        (void)hipGetLastError();
    }
    return c->stage_host != nullptr;
}
static void stage_begin(b32_ctx* c) {
    (void)hipSetDevice(c->device);
    stage_ensure(c);
    c->stage_used = 0; c->stage_segs.count = 0;
    c->stage_active = c->stage_host != nullptr;
}
static void stage_flush(b32_ctx* c) {        // enqueue the one copy kernel (ordered before the frame's kernels on the same stream)
    if (c->stage_active && c->stage_segs.count) launch_upload(c->stream, c->stage_dev, c->stage_segs);
    c->stage_active = false; c->stage_segs.count = 0; c->stage_used = 0;
}

// per-face work buffers of the current frame set for a mesh of nf faces
static int ensure_work(b32_ctx* c, uint32_t nf) {
    int rc;
    if ((size_t)nf + 1 > c->cap_work || !c->crecs) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        const size_t n = (size_t)nf + nf / 4 + 16;
        for (int i = 0; i < 2; ++i) { if ((rc = ensure_plain(c, c->keys[i], n))) return rc; if ((rc = ensure_plain(c, c->vals[i], n))) return rc; }
        if ((rc = ensure_plain(c, c->crecs, n))) return rc;
        if ((rc = ensure_plain(c, c->srecs, n))) return rc;
        if ((rc = ensure_plain(c, c->xrecs, n))) return rc;
        if ((rc = ensure_plain(c, c->counts, n))) return rc;
        if ((rc = ensure_plain(c, c->spans, n))) return rc;
        if ((rc = ensure_plain(c, c->face_of, n))) return rc;
        c->bin_blocks = (uint32_t)((n + 4095) / 4096);
        c->partial_blocks = (uint32_t)((n + 255) / 256);
        if ((rc = ensure_plain(c, c->partials, (size_t)c->partial_blocks * 8 + 8))) return rc;
        if ((rc = ensure_plain(c, c->block_sums, (size_t)c->bin_blocks + 1))) return rc;
        c->cap_work = n;
    }
    return B32_OK;
}

static int upload_geometry(b32_ctx* c, const B32Vertex* v, uint32_t nv, const B32Face* f, uint32_t nf) {
    if ((nv && !v) || (nf && !f)) return B32_E_ARG;
    int rc;
    if ((rc = ensure(c, c->d_verts, c->cap_verts, (size_t)nv + 1))) return rc;
    if ((rc = ensure(c, c->d_faces, c->cap_faces, (size_t)nf + 1))) return rc;
    {   // can any face end up in the transparent pass? (face blend mode / editor alpha; texture blend modes are added by the callers)
        uint32_t nb = 0;
        uint32_t nbt = 0;                    // ... counting the faces a texture's blend mode puts there too (render.rs:2403-2415)
        for (uint32_t i = 0; i < nf; ++i) {
            const bool own = f[i].blend_mode != B32_BLEND_OPAQUE || f[i].editor_alpha < 255;
            const uint32_t t = f[i].texture_id;
            nb += own ? 1u : 0u;
            nbt += (own || (t != B32_NO_TEXTURE && t < c->nt && t < c->h_tex.size() && c->h_tex[t].blend_mode != B32_BLEND_OPAQUE)) ? 1u : 0u;
        }
        c->may_blend = nb != 0; c->blend_faces = nbt;
    }
    if ((rc = h2d(c, c->d_verts, v, (size_t)nv * sizeof(B32Vertex)))) return rc;
    if ((rc = h2d(c, c->d_faces, f, (size_t)nf * sizeof(B32Face)))) return rc;
    // a mesh of another size: tile regions sized afresh (the per-frame drop-in call uploads the same mesh again and again: what an
    // overflowing frame taught the context stays)
    if (c->nf != nf) { c->direct_cap_opaque = 0; c->direct_ntiles = 0; c->direct_ok = true; }
    c->nv = nv; c->nf = nf;
    c->local_sort_ok = true;
    c->pos_valid = false; c->band_frames = 0;
    if ((rc = ensure_work(c, nf))) return rc;
    c->gen = ++c->gen_counter;
    c->h_consts[0] = nf;
    if (!c->d_consts) HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_consts), 16 * sizeof(uint32_t)));   // (swapped away with a scene)
    if ((rc = h2d(c, c->d_consts, c->h_consts, sizeof(c->h_consts)))) return rc;
    // the caller may reuse its host buffers as soon as an upload call returns; the drop-in render calls return only after
    // b32_frame_finish has synchronised the stream, so they skip this extra round trip
    if (!c->defer_upload_sync) HIPCHK(c, hipStreamSynchronize(c->stream));
    return B32_OK;
}

// CHEAP coverage is worth it while skipped winners are rare: textures with at most 1/cheap_den skippable texels (b32_set_cheap_threshold).  Measured on the C3
// geometry with 1 transparent CLUT entry out of K (tools/cheap_threshold.py): EXACT coverage (skip mask in LDS) 0.233 ms whatever the
// texture; CHEAP 0.19 ms at K = 256, 0.220 at 64, 0.246 at 32, 0.307 at 16, 0.46 at 8.  (b32_set_cheap_threshold: that tool's switch.)

static int layout_textures(b32_ctx* c, uint32_t nt, const uint32_t* w, const uint32_t* h, const uint32_t* blend, size_t* total, bool rgba = false) {
    if (nt > 65534) return B32_E_UNSUPPORTED;        // the surface record holds the texture slot in 16 bits
    c->h_tex.resize(nt);
    c->tex_blend_any = false;
    c->atlas_idx_bytes = 0;                         // (only b32_scene_upload_indexed with one texture keeps the index atlas)
    size_t off = 0;
    for (uint32_t i = 0; i < nt; ++i) {
        if (w[i] > 65535 || h[i] > 65535) return B32_E_ARG;
        if (blend[i] != B32_BLEND_OPAQUE) c->tex_blend_any = true;
        c->h_tex[i] = { w[i], h[i], blend[i], (uint32_t)off };
        off += ((size_t)w[i] * h[i] + 7) & ~(size_t)7;
        if (off > 0x7FFFFFFFull) return B32_E_ARG;
    }
    *total = off + 8;
    c->pool_texels = (uint32_t)off; c->mask_dirty = true;
    int rc;
    if ((rc = ensure(c, c->d_texmask, c->cap_texmask, off / 32 + 4))) return rc;
    if (rgba) { if ((rc = ensure(c, c->d_texels32, c->cap_texels32, *total))) return rc; }
    else if ((rc = ensure(c, c->d_texels, c->cap_texels, *total))) return rc;
    if ((rc = ensure(c, c->d_tex, c->cap_tex, (size_t)nt + 1))) return rc;
    if ((rc = h2d(c, c->d_tex, c->h_tex.data(), nt * sizeof(TexDesc)))) return rc;
    c->nt = nt;
    return B32_OK;
}

// 64-bit content hash, four independent lanes of 8-byte words (about memcpy speed; the tail bytes go through a padded word)
static uint64_t hash_bytes(const void* data, size_t n) {
    const unsigned char* p = static_cast<const unsigned char*>(data);
    const uint64_t K1 = 0x9E3779B185EBCA87ull, K2 = 0xC2B2AE3D27D4EB4Full;
    uint64_t h[4] = { K1 ^ n, K2 + n, K1 * 3 + n, K2 * 5 ^ n };
    auto round = [&](uint64_t acc, uint64_t x) { acc += x * K2; acc = (acc << 31) | (acc >> 33); return acc * K1; };
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        uint64_t w[4];
        std::memcpy(w, p + i, 32);
        h[0] = round(h[0], w[0]); h[1] = round(h[1], w[1]); h[2] = round(h[2], w[2]); h[3] = round(h[3], w[3]);
    }
    uint64_t tail[4] = { 0, 0, 0, 0 };
    if (i < n) { std::memcpy(tail, p + i, n - i); for (int k = 0; k < 4; ++k) h[k] = round(h[k], tail[k]); }
    uint64_t r = ((h[0] << 1) | (h[0] >> 63)) ^ ((h[1] << 7) | (h[1] >> 57)) ^ ((h[2] << 12) | (h[2] >> 52)) ^ ((h[3] << 18) | (h[3] >> 46));
    r ^= r >> 33; r *= K2; r ^= r >> 29; r *= K1; r ^= r >> 32;
    return r;
}

int b32_scene_upload(b32_ctx* c, const B32Vertex* v, uint32_t nv, const B32Face* f, uint32_t nf, const B32Texture15* tex, uint32_t nt) {
    if (!c || (nt && !tex)) return B32_E_ARG;
    (void)hipSetDevice(c->device);
    // a pending frame that may still be redrawn (overflowed tile regions / pair buffers) is drawn from the RESIDENT scene: settle it
    // before that scene is replaced, or the redraw would draw the new mesh in its place (and the new mesh twice)
    { const int rcs = settle_pending(c); if (rcs) return rcs; }
    c->have_scene = false;
    std::vector<uint32_t> w(nt), h(nt), bl(nt);
    for (uint32_t i = 0; i < nt; ++i) {
        w[i] = tex[i].width; h[i] = tex[i].height; bl[i] = tex[i].blend_mode;
        if (!tex[i].pixels) w[i] = h[i] = 0;                                // pixels.is_empty() -> sample() returns TRANSPARENT
    }
    // texture cache: the same set as the pool holds (pointer, size, blend mode, content hash of every texture)?
    std::vector<b32_ctx::TexSig> sig(nt);
    for (uint32_t i = 0; i < nt; ++i) sig[i] = { tex[i].pixels, w[i], h[i], bl[i], hash_bytes(tex[i].pixels, (size_t)w[i] * h[i] * 2) };
    bool hit = c->tex_sig_valid && !(c->route_off & B32_ROUTE_TEX_CACHE) && c->tex_sig.size() == nt && c->nt == nt && c->d_texels && c->d_tex;
    for (uint32_t i = 0; hit && i < nt; ++i) {
        const b32_ctx::TexSig& o = c->tex_sig[i];
        hit = o.ptr == sig[i].ptr && o.w == sig[i].w && o.h == sig[i].h && o.blend == sig[i].blend && o.hash == sig[i].hash;
    }
    int rc;
    if (!hit) {
        c->tex_sig_valid = false;
        size_t total = 0;
        rc = layout_textures(c, nt, w.data(), h.data(), bl.data(), &total);
        if (rc) return rc;
        c->cheap_ok = true;
        for (uint32_t i = 0; i < nt; ++i) {
            const size_t n = (size_t)w[i] * h[i];
            if ((rc = h2d(c, c->d_texels + c->h_tex[i].offset, tex[i].pixels, n * 2))) return rc;
            size_t skippable = 0;                                               // texels the black_transparent rule can skip
            const uint16_t* px = tex[i].pixels;
            for (size_t k = 0; k < n; ++k) skippable += (px[k] & 0x7FFF) == 0;
            if (n == 0 || skippable * c->cheap_den > n) c->cheap_ok = false;
        }
        c->tex_sig.swap(sig); c->tex_sig_valid = true;
    }
    if ((rc = upload_geometry(c, v, nv, f, nf))) return rc;
    for (uint32_t i = 0; i < nt; ++i) if (bl[i] != B32_BLEND_OPAQUE) c->may_blend = true;
    c->fmt8 = false;
    c->have_scene = true;
    return B32_OK;
}

int b32_scene_upload_rgba(b32_ctx* c, const B32Vertex* v, uint32_t nv, const B32Face* f, uint32_t nf, const B32Texture* tex, uint32_t nt) {
    if (!c || (nt && !tex)) return B32_E_ARG;
    (void)hipSetDevice(c->device);
    // a pending frame that may still be redrawn (overflowed tile regions / pair buffers) is drawn from the RESIDENT scene: settle it
    // before that scene is replaced, or the redraw would draw the new mesh in its place (and the new mesh twice)
    { const int rcs = settle_pending(c); if (rcs) return rcs; }
    c->have_scene = false;
    std::vector<uint32_t> w(nt), h(nt), bl(nt);
    for (uint32_t i = 0; i < nt; ++i) {
        w[i] = tex[i].width; h[i] = tex[i].height; bl[i] = tex[i].blend_mode;
        if (!tex[i].pixels) w[i] = h[i] = 0;                                // pixels.is_empty() -> Color::TRANSPARENT
    }
    size_t total = 0;
    c->tex_sig_valid = false;                                               // (the pool is rewritten below)
    int rc = layout_textures(c, nt, w.data(), h.data(), bl.data(), &total, true);
    if (rc) return rc;
    c->cheap_ok = true;
    bool blend_texels = false;
    for (uint32_t i = 0; i < nt; ++i) {
        const size_t n = (size_t)w[i] * h[i];
        if ((rc = h2d(c, c->d_texels32 + c->h_tex[i].offset, tex[i].pixels, n * 4))) return rc;
        size_t skippable = 0;                                               // Erase texels: the fragment is skipped (render.rs:1348)
        for (size_t k = 0; k < n; ++k) {
            const uint8_t b = tex[i].pixels[k * 4 + 3];
            skippable += b == B32_BLEND_ERASE;
            blend_texels |= b != B32_BLEND_OPAQUE && b != B32_BLEND_ERASE;
        }
        if (n == 0 || skippable * c->cheap_den > n) c->cheap_ok = false;
    }
    if ((rc = upload_geometry(c, v, nv, f, nf))) return rc;
    bool alpha_faces = false;
    for (uint32_t i = 0; i < nf && !alpha_faces; ++i) alpha_faces = f[i].editor_alpha < 255;
    c->blend8 = blend_texels || alpha_faces;
    c->may_blend = false;
    c->fmt8 = true;
    c->have_scene = true;
    return B32_OK;
}

int b32_scene_upload_indexed(b32_ctx* c, const B32Vertex* v, uint32_t nv, const B32Face* f, uint32_t nf, const B32IndexedTexture* tex, uint32_t nt) {
    if (!c || (nt && !tex)) return B32_E_ARG;
    (void)hipSetDevice(c->device);
    // a pending frame that may still be redrawn (overflowed tile regions / pair buffers) is drawn from the RESIDENT scene: settle it
    // before that scene is replaced, or the redraw would draw the new mesh in its place (and the new mesh twice)
    { const int rcs = settle_pending(c); if (rcs) return rcs; }
    c->have_scene = false;
    std::vector<uint32_t> w(nt), h(nt), bl(nt);
    for (uint32_t i = 0; i < nt; ++i) {
        w[i] = tex[i].width; h[i] = tex[i].height; bl[i] = tex[i].blend_mode;
        if (!tex[i].indices || !tex[i].clut) w[i] = h[i] = 0;
    }
    size_t total = 0;
    c->tex_sig_valid = false;                                               // (the pool is rewritten below)
    int rc = layout_textures(c, nt, w.data(), h.data(), bl.data(), &total);
    if (rc) return rc;
    c->cheap_ok = true;
    for (uint32_t i = 0; i < nt; ++i) {
        const size_t n = (size_t)w[i] * h[i];
        if (!n) { c->cheap_ok = false; continue; }
        // the expansion kernel also counts the texels the black_transparent rule can skip (no walk over the texels on the host)
        uint8_t* d_idx = nullptr; uint16_t* d_clut = nullptr; uint32_t* d_cnt = nullptr;
        Scratch tmp(c);
        // ONE texture with at most 256 palette entries: index bytes and CLUT stay on the device behind each other -- 256 Color15 entries
        // (zero behind the palette, which is what Clut::lookup returns for an index past it, types.rs:390-397), then the indices -- so that
        // the fused kernel can stage them in LDS (B32_ROUTE_LDS_ATLAS); the expansion below reads the same copies
        const bool keep = nt == 1 && tex[i].clut_len <= 256u && n <= (160u << 10);
        if (keep) {
            if ((rc = ensure(c, c->d_atlas0, c->cap_atlas0, (size_t)ATLAS_CLUT_BYTES + n + 32))) return rc;
            HIPCHK(c, hipMemsetAsync(c->d_atlas0, 0, ATLAS_CLUT_BYTES, c->stream));
            if ((rc = h2d(c, c->d_atlas0, tex[i].clut, (size_t)tex[i].clut_len * 2))) return rc;
            if ((rc = h2d(c, c->d_atlas0 + ATLAS_CLUT_BYTES, tex[i].indices, n))) return rc;
            HIPCHK(c, hipMemsetAsync(c->d_atlas0 + ATLAS_CLUT_BYTES + n, 0, 32, c->stream));      // (the staging copy reads whole 16-byte quads)
            d_clut = reinterpret_cast<uint16_t*>(c->d_atlas0); d_idx = c->d_atlas0 + ATLAS_CLUT_BYTES;
            c->atlas_idx_bytes = (uint32_t)n;
        } else {
            if ((rc = tmp.upload(tex[i].indices, n, &d_idx))) return rc;
            if ((rc = tmp.upload(tex[i].clut, (size_t)tex[i].clut_len, &d_clut))) return rc;
        }
        if ((rc = tmp.alloc(&d_cnt, 1))) return rc;
        HIPCHK(c, hipMemsetAsync(d_cnt, 0, 4, c->stream));
        launch_expand_indexed(c->stream, d_idx, (uint32_t)n, d_clut, tex[i].clut_len, c->d_texels + c->h_tex[i].offset, d_cnt);
        uint32_t skippable = 0;
        HIPCHK(c, hipMemcpyAsync(&skippable, d_cnt, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if ((size_t)skippable * c->cheap_den > n) c->cheap_ok = false;
    }
    if ((rc = upload_geometry(c, v, nv, f, nf))) return rc;
    for (uint32_t i = 0; i < nt; ++i) if (bl[i] != B32_BLEND_OPAQUE) c->may_blend = true;
    c->fmt8 = false;
    c->have_scene = true;
    return B32_OK;
}

// ------------------------------------------------------------------ frame
static int validate_settings(const B32Settings* st) {
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
        // tile route: one counter and one list region per 64x64 tile of the band (+ the overflow flag and the big-edge count)
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
            if (grid != c->wire_grid || wt > c->cap_wire_tiles) {
                HIPCHK(c, hipMemsetAsync(c->wire_fill, 0, ((c->cap_wire_tiles + 2) * FILL_PAD + 64) * sizeof(uint32_t), s));
                c->wire_grid = grid;
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
    (void)fp;
    if (c->nv && c->nf > 8192u && !(c->route_off & B32_ROUTE_PACKED_STREAMS)) {
        if (!c->pos_valid && c->band_frames >= 1) {
            if ((size_t)c->nv * 6 > c->cap_pos12 || !c->d_pos12) {
                if ((rc = ensure_plain(c, c->d_pos12, (size_t)c->nv * 6 + 16))) return rc;
                c->cap_pos12 = (size_t)c->nv * 6;
            }
            launch_pack_streams(c->stream, c->d_verts, c->nv, c->d_pos12, c->d_pos12 + (size_t)c->nv * 3);
            c->pos_valid = true; c->side_dirty = true;
        }
        c->band_frames++;
        if (c->pos_valid) { pos12 = c->d_pos12; attr12 = c->d_pos12 + (size_t)c->nv * 3; }
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
    fa.co_run = (c->pipelined || (c->deep_async && c->pipe_hint && c->nf > 2048u && !(c->route_off & B32_ROUTE_PIPELINE))) ? 1u : 0u;   // (this frame's or the next one's setup kernel beside a fill)
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

static int enqueue_frame(b32_ctx* c, const B32Camera* cam, const B32Settings* st, const B32Fog* fog) {
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
    if (c->frame_pending && c->pipe_hint && !c->redrawing && !fp.wire_collect && !prof_all && c->nf > pipe_min_faces && !(c->route_off & B32_ROUTE_PIPELINE)) {
        if ((rc = pipeline_ensure(c))) return rc;
        rotate_sets(c);
        c->pipelined = true;
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
    c->pipe_hint = r.direct_bin && (ntiles > 2u * (uint32_t)c->n_cu || c->frame_batched || (c->band_set && B32_PIPELINE_BANDS));
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
    }
    const float *pos12 = nullptr, *attr12 = nullptr;
    if ((rc = frame_positions(c, fp, pos12, attr12))) return fail(rc);

    // ---- transform, cull, setup (+ tile binning of large meshes)
    hipStream_t ss = s;
    if (c->pipelined) {
        if (c->side_dirty) { HIPCHK(c, hipEventRecord(c->ev_main, s)); HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_main, 0)); c->side_dirty = false; }
        HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_done, 0));     // the last fill that read this set (two frames ago)
        ss = c->side;
        c->pipelined_frames++;
        if (c->gate_permille && c->last_cover_tiles && c->alt[0].d_ctrl) {
            // The fused kernel's workgroups take their next tile from the cursor after the coverage of the current one: the cursor
            // passes tiles - groups when the last tile is handed out, and every fetch beyond that is a workgroup that found the queue
            // empty and has only the shading of its last tile left, i.e. is about to free its place on a CU.
            const uint32_t groups = c->last_cover_groups, tiles = c->last_cover_tiles;
            const uint32_t pre = tiles > groups ? tiles - groups : 0u;      // cursor value when the last tile is handed out
            const uint32_t need = c->gate_permille > 1000u ? (uint32_t)((uint64_t)(c->gate_permille - 1000u) * pre / 1000u)
                                                           : pre + (uint32_t)((uint64_t)(c->gate_permille - 1u) * groups / 1000u);
            if (need) launch_gate(ss, c->alt[0].d_ctrl, need, 30000u /* 300 us */);      // alt[0]: the frame n_sets - 1 back (rotate_sets)
        }
    }
    if (prof_all) HIPCHK(c, hipEventRecord(ev[0], s));
    launch_setup(ss, fp, c->d_verts, c->d_faces, c->d_tex, c->d_lights, lset, c->frame_table, RecArrays{ c->crecs, c->srecs, c->xrecs }, r.db, c->shades, c->keys[0],
                 r.direct_bin ? nullptr : c->spans /* (direct binning: nobody reads the spans) */, c->partials, c->d_ctrl, c->wire, c->n_cu, pos12, attr12, c->face_of);
    if (c->pipelined) {
        hipError_t e1 = hipEventRecord(c->ev_setup, c->side);
        if (e1 == hipSuccess) e1 = hipStreamWaitEvent(s, c->ev_setup, 0);
        if (e1 != hipSuccess) { (void)hipStreamSynchronize(c->side); c->last_hip = (int)e1; return B32_E_HIP; }
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
    launch_fill(s, fa, c->n_cu, prof_fill ? ev[4] : nullptr);

    // ---- wireframe phases
    if (fp.wire_collect && c->nf) {
        WireArgs wa{};
        wa.tris = c->wire; wa.nf = c->nf; wa.table_owner = c->wire_owner; wa.table_first = c->wire_first;
        wa.table_mask = c->cap_wire_table ? (uint32_t)(c->cap_wire_table - 1) : 0;
        wa.fb = c->fb; wa.zbuf = (c->zbuf && c->zbuf_valid) ? c->zbuf : nullptr;
        wa.width = c->width; wa.height = c->height; wa.band_y0 = c->band_y0; wa.band_y1 = c->band_y1; wa.ctrl = c->d_ctrl;
        if (!(c->route_off & B32_ROUTE_WIRE_TILES) && c->wire_fill && c->band_y1 > c->band_y0) {
            wa.tile_yb = (c->band_y0 / WIRE_TH) * WIRE_TH; wa.tiles_x = (c->width + TILE_W - 1) / TILE_W;
            wa.tiles_y = (c->band_y1 - wa.tile_yb + WIRE_TH - 1) / WIRE_TH;
            if ((size_t)wa.tiles_x * wa.tiles_y <= c->cap_wire_tiles) { wa.tile_fill = c->wire_fill; wa.tile_lists = c->wire_lists; c->wire_tile_frames++; }
        }
        launch_wire(s, wa, wire_back, wire_front);
    }
    if (prof_fill) { if (prof_all) HIPCHK(c, hipEventRecord(ev[5], s)); c->ev_frames++; }
    if (c->side) HIPCHK(c, hipEventRecord(c->ev_done, s));         // (the next setup kernel that writes this set waits for it)
    c->last_cover_tiles = (r.prio64 && !wire_front && !r.ordered_all) ? ntiles : 0u;
    c->last_cover_groups = std::min<uint32_t>(ntiles, (uint32_t)c->n_cu * 2u);
    HIPCHK(c, hipGetLastError());
    return B32_OK;
}

static int render_scene_async_any(b32_ctx* c, const B32Camera* cam, const B32Settings* st, const B32Fog* fog);
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
static int render_scene_async_any(b32_ctx* c, const B32Camera* cam, const B32Settings* st, const B32Fog* fog) {
    if (!c || !cam || !st || !c->fb || !c->have_scene) return B32_E_ARG;
    (void)hipSetDevice(c->device);
    int rc = validate_settings(st);
    if (rc) return rc;
    // keep a private copy of the lights so a redraw after overflow does not dereference a dead caller pointer
    // (safe mode) a pending frame that may still need a redraw is settled before the next one overwrites its control block
    if (!c->deep_async && (rc = settle_pending(c))) return rc;
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
        c->routes[6]++;
        c->ev_frames = 0;                              // the aborted frame must not enter the phase averages
        c->redrawing = true;
        rc = enqueue_frame(c, &c->last_cam, &c->last_settings, c->last_has_fog ? &c->last_fog : nullptr);
        c->redrawing = false;
        if (rc) return rc;
    }
    c->frame_pending = false;
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
    if (c->h_ctrl.pairs_overflow) return B32_E_HIP;
    if (sticky >> 8) return B32_E_FRAME_DROPPED;              // deep asynchronous mode: an earlier frame was lost (the last one is good)
    if (c->h_ctrl.err_index || (sticky & 1u)) return B32_E_INDEX;
    if (c->h_ctrl.abort || (sticky & 2u)) return B32_E_NAN_KEY;
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

// ------------------------------------------------------------------ scene slots (several resident scenes per context)
int b32_scene_create(b32_ctx* c, b32_scene** out) {
    if (!c || !out) return B32_E_ARG;
    *out = new b32_scene();
    return B32_OK;
}
void b32_scene_destroy(b32_ctx* c, b32_scene* sl) {
    if (!c || !sl) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    void* ptrs[] = { sl->d_verts, sl->d_faces, sl->d_texels, sl->d_texels32, sl->d_tex, sl->d_consts, sl->d_texmask, sl->d_pos12, sl->d_atlas0 };
    for (void* p : ptrs) if (p) (void)hipFree(p);
    delete sl;
}
int b32_scene_swap(b32_ctx* c, b32_scene* sl) {
    if (!c || !sl) return B32_E_ARG;
    // a pending frame of the outgoing scene that may have to be redrawn (pair overflow, long transparent lists) is settled first:
    // the redraw needs that scene.  Frames of small meshes never redraw and stay in flight.
    // Its error, if any, is the frame's error: kept for the b32_frame_finish that ends the frame (the exchange itself goes ahead).
    { const int rc = settle_pending(c); if (rc) return rc; }
    std::swap(c->d_verts, sl->d_verts); std::swap(c->cap_verts, sl->cap_verts);
    std::swap(c->d_faces, sl->d_faces); std::swap(c->cap_faces, sl->cap_faces);
    std::swap(c->d_texels, sl->d_texels); std::swap(c->cap_texels, sl->cap_texels);
    std::swap(c->d_texels32, sl->d_texels32); std::swap(c->cap_texels32, sl->cap_texels32);
    std::swap(c->d_tex, sl->d_tex); std::swap(c->cap_tex, sl->cap_tex);
    std::swap(c->d_consts, sl->d_consts);
    std::swap(c->d_texmask, sl->d_texmask); std::swap(c->cap_texmask, sl->cap_texmask); std::swap(c->pool_texels, sl->pool_texels);
    std::swap(c->mask_dirty, sl->mask_dirty);
    std::swap(c->d_atlas0, sl->d_atlas0); std::swap(c->cap_atlas0, sl->cap_atlas0); std::swap(c->atlas_idx_bytes, sl->atlas_idx_bytes);
    c->h_tex.swap(sl->h_tex);
    std::swap(c->nv, sl->nv); std::swap(c->nf, sl->nf); std::swap(c->nt, sl->nt);
    std::swap(c->fmt8, sl->fmt8); std::swap(c->blend8, sl->blend8); std::swap(c->have_scene, sl->have_scene); std::swap(c->gen, sl->gen); std::swap(c->blend_faces, sl->blend_faces);
    std::swap(c->may_blend, sl->may_blend); std::swap(c->cheap_ok, sl->cheap_ok); std::swap(c->local_sort_ok, sl->local_sort_ok);
    std::swap(c->tex_blend_any, sl->tex_blend_any);
    std::swap(c->direct_cap_opaque, sl->direct_cap_opaque); std::swap(c->direct_ntiles, sl->direct_ntiles); std::swap(c->direct_ok, sl->direct_ok);
    std::swap(c->d_pos12, sl->d_pos12); std::swap(c->cap_pos12, sl->cap_pos12); std::swap(c->pos_valid, sl->pos_valid); std::swap(c->band_frames, sl->band_frames);
    c->tex_sig.swap(sl->tex_sig); std::swap(c->tex_sig_valid, sl->tex_sig_valid);
    return B32_OK;
}


// ------------------------------------------------------------------ batched frame (several meshes, one setup + fill pair)
// scene.rs:112-261 draws a frame as one render_mesh_15 call per room and per asset part onto the same framebuffer: at 320x240 that is a
// chain of launch-latency bound kernel pairs (~50 us per mesh).  b32_frame_begin / b32_frame_add_scene / b32_frame_end take the same
// sequence of calls -- resident meshes in scene slots, one camera and base settings per frame, ambient / fog / backface_cull per mesh as
// the reference's callers vary them -- and draw every RUN of meshes that commutes as ONE merged mesh:
//   * z-buffer mode (RasterSettings::game() and the reference default): opaque fragments are depth-tested, so their order does not
//     matter, and a depth tie goes to the earlier face exactly like the sequential strict `z < zbuffer` (the priority's low word is
//     the record slot, monotone in mesh order then face order);
//   * a mesh with a transparent pass blends against what was drawn before it, so it ENDS its run: its opaque faces join the merged
//     opaque pass, its transparent faces are the run's transparent pass (all earlier opaque faces are in place by then, as in the
//     sequential calls);
//   * painter's mode, the 8-bit-colour path, x-ray, orthographic views and the wireframe phases are drawn mesh by mesh as before.
// The merged mesh (vertices, faces with the member number in their spare byte, texel pool, texture descriptors) is built on the device
// from the slots and kept while the members' contents stay the same.
static void release_scene_buffers(b32_scene* sl) {
    void* ptrs[] = { sl->d_verts, sl->d_faces, sl->d_texels, sl->d_texels32, sl->d_tex, sl->d_consts, sl->d_texmask, sl->d_pos12, sl->d_atlas0 };
    for (void* p : ptrs) if (p) (void)hipFree(p);
}
static int build_merged(b32_ctx* c, const b32_ctx::BatchEntry* e, uint32_t n, b32_scene* m) {
    uint64_t nv = 0, nf = 0, nt = 0, pool = 0;
    for (uint32_t j = 0; j < n; ++j) { const b32_scene* sl = e[j].slot; nv += sl->nv; nf += sl->nf; nt += sl->nt; pool += sl->pool_texels; }
    if (nv >= 0x7FFFFFFFull || nf >= 0x7FFFFFFFull || nt > 65534 || pool > 0x7FFFFFFFull) return B32_E_UNSUPPORTED;
    int rc;
    if ((rc = ensure(c, m->d_verts, m->cap_verts, (size_t)nv + 1))) return rc;
    if ((rc = ensure(c, m->d_faces, m->cap_faces, (size_t)nf + 1))) return rc;
    if ((rc = ensure(c, m->d_texels, m->cap_texels, (size_t)pool + 8))) return rc;
    if ((rc = ensure(c, m->d_tex, m->cap_tex, (size_t)nt + 1))) return rc;
    if ((rc = ensure(c, m->d_texmask, m->cap_texmask, (size_t)pool / 32 + 4))) return rc;
    if (!m->d_consts) HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&m->d_consts), 16 * sizeof(uint32_t)));
    m->h_tex.clear();
    uint32_t vb = 0, fb = 0, tb = 0, pb = 0;
    m->may_blend = false; m->cheap_ok = true; m->tex_blend_any = false; m->blend_faces = 0;
    for (uint32_t j = 0; j < n; ++j) {
        const b32_scene* sl = e[j].slot;
        launch_merge_mesh(c->stream, sl->d_verts, sl->nv, sl->d_faces, sl->nf, sl->nt, m->d_verts, m->d_faces + fb, vb, tb, j);
        if (sl->pool_texels) HIPCHK(c, hipMemcpyAsync(m->d_texels + pb, sl->d_texels, (size_t)sl->pool_texels * 2, hipMemcpyDeviceToDevice, c->stream));
        launch_offset_tex(c->stream, sl->d_tex, sl->nt, m->d_tex + tb, pb);
        for (const TexDesc& d : sl->h_tex) m->h_tex.push_back({ d.width, d.height, d.blend_mode, d.offset + pb });
        m->may_blend |= sl->may_blend; m->cheap_ok &= sl->cheap_ok; m->tex_blend_any |= sl->tex_blend_any;
        m->blend_faces += sl->blend_faces;
        vb += sl->nv; fb += sl->nf; tb += sl->nt; pb += sl->pool_texels;
    }
    const uint32_t consts[4] = { (uint32_t)nf, 0, 0, 0 };
    HIPCHK(c, hipMemcpyAsync(m->d_consts, consts, sizeof(consts), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));          // (`consts` is on the stack; a merged mesh is built once and reused)
    HIPCHK(c, hipGetLastError());
    m->nv = (uint32_t)nv; m->nf = (uint32_t)nf; m->nt = (uint32_t)nt; m->pool_texels = (uint32_t)pool; m->mask_dirty = true;
    m->fmt8 = false; m->blend8 = false; m->have_scene = true; m->local_sort_ok = true;
    m->direct_cap_opaque = 0; m->direct_ntiles = 0; m->direct_ok = true; m->pos_valid = false; m->band_frames = 0;
    m->tex_sig_valid = false; m->gen = ++c->gen_counter;
    c->side_dirty = true;
    return B32_OK;
}
// the merged mesh of a run: from the cache when the same slots with the same contents were merged before
static int merged_for(b32_ctx* c, const b32_ctx::BatchEntry* e, uint32_t n, b32_scene** out) {
    ++c->batch_clock;
    for (auto& r : c->merged_runs) {
        if (r.members.size() != n) continue;
        bool same = true;
        for (uint32_t j = 0; same && j < n; ++j) same = r.members[j] == e[j].slot && r.gens[j] == e[j].slot->gen;
        if (same) { r.used = c->batch_clock; *out = r.merged; return B32_OK; }
    }
    b32_ctx::MergedRun* slot = nullptr;
    if (c->merged_runs.size() >= 64) {                   // bounded cache: the least recently used merged mesh makes room
        slot = &c->merged_runs[0];
        for (auto& r : c->merged_runs) if (r.used < slot->used) slot = &r;
        HIPCHK(c, hipStreamSynchronize(c->stream));
    } else { c->merged_runs.emplace_back(); slot = &c->merged_runs.back(); slot->merged = new b32_scene(); }
    slot->members.clear(); slot->gens.clear();
    const int rc = build_merged(c, e, n, slot->merged);
    if (rc) return rc;
    for (uint32_t j = 0; j < n; ++j) { slot->members.push_back(e[j].slot); slot->gens.push_back(e[j].slot->gen); }
    slot->used = c->batch_clock;
    c->batch_stats[2]++;
    *out = slot->merged;
    return B32_OK;
}

int b32_frame_begin(b32_ctx* c, const B32Camera* cam, const B32Settings* st) {
    if (!c || !cam || !st || !c->fb) return B32_E_ARG;
    const int rc = validate_settings(st);
    if (rc) return rc;
    c->batch_cam = *cam; c->batch_st = *st;
    c->batch_lights.assign(st->lights, st->lights + (st->lights ? st->n_lights : 0));
    c->batch_st.lights = nullptr;                         // (patched to the private copy when the frame is enqueued)
    c->batch.clear();
    c->batch_open = true;
    return B32_OK;
}
int b32_frame_add_scene(b32_ctx* c, b32_scene* sl, const B32MeshParams* p) {
    if (!c || !sl || !c->batch_open) return B32_E_ARG;
    b32_ctx::BatchEntry e{};
    e.slot = sl;
    e.row.ambient = p ? p->ambient : c->batch_st.ambient;
    const bool cull = p ? p->backface_cull != 0 : c->batch_st.backface_cull != 0;
    const bool fogged = p && p->has_fog;
    e.row.flags = (cull ? 1u : 0u) | (fogged ? 2u : 0u);
    if (fogged) e.row.fog = p->fog;
    e.wire = (p ? p->backface_wireframe != 0 : c->batch_st.backface_wireframe != 0) && cull;       // render.rs:2577
    c->batch.push_back(e);
    return B32_OK;
}
int b32_frame_end(b32_ctx* c) {
    if (!c || !c->batch_open) return B32_E_ARG;
    c->batch_open = false;
    (void)hipSetDevice(c->device);
    B32Settings base = c->batch_st;
    base.lights = c->batch_lights.empty() ? nullptr : c->batch_lights.data();
    base.n_lights = (uint32_t)c->batch_lights.size();
    const bool can_merge = base.use_zbuffer && base.use_rgb555 && !base.xray_mode && !base.has_ortho && !base.wireframe_overlay &&
                           !(c->route_off & B32_ROUTE_BATCH);
    c->batch_stats[3]++;
    const size_t n = c->batch.size();
    int rc = B32_OK;
    auto draw_one = [&](const b32_ctx::BatchEntry& e) -> int {      // the mesh on its own, exactly like a b32_render_scene_15_async call
        B32Settings st = base;
        st.ambient = e.row.ambient; st.backface_cull = (e.row.flags & 1u) ? 1 : 0; st.backface_wireframe = e.wire ? 1 : 0;
        int r = b32_scene_swap(c, e.slot);
        if (r) return r;
        if (!c->have_scene) r = B32_E_ARG;
        else if (c->fmt8) { c->frame_batched = false; r = render_scene_async_any(c, &c->batch_cam, &st, nullptr); }
        else { c->frame_batched = false; r = render_scene_async_any(c, &c->batch_cam, &st, (e.row.flags & 2u) ? &e.row.fog : nullptr); }
        const int r2 = b32_scene_swap(c, e.slot);
        c->batch_stats[1]++;
        return r ? r : r2;
    };
    size_t i = 0;
    while (i < n && rc == B32_OK) {
        // the run starting at mesh i: meshes that commute, ended by (and including) the first one with a transparent pass
        size_t k = i;
        if (can_merge) {
            while (k < n && k - i < BATCH_MESHES) {
                const b32_scene* sl = c->batch[k].slot;
                if (!sl->have_scene || sl->fmt8 || c->batch[k].wire || !sl->nf) break;
                ++k;
                if (sl->may_blend) break;
            }
        }
        if (k - i < 2) { rc = draw_one(c->batch[i]); ++i; continue; }
        b32_scene* m = nullptr;
        if ((rc = merged_for(c, &c->batch[i], (uint32_t)(k - i), &m))) break;
        if ((rc = b32_scene_swap(c, m))) break;
        rc = ensure_work(c, c->nf);
        if (rc == B32_OK) {
            bool any_fog = false;
            for (size_t j = i; j < k; ++j) { c->frame_table.m[j - i] = c->batch[j].row; any_fog |= (c->batch[j].row.flags & 2u) != 0; }
            c->frame_batched = true;
            B32Fog f0{};                                    // (fp.has_fog switches the fog code on; the rows decide per mesh)
            B32Settings mst = base;                         // (members of a run never have a wireframe phase: see the run split above;
            mst.backface_wireframe = 0;                     //  the base's flag must not give the merged mesh one -- found by the soak)
            rc = render_scene_async_any(c, &c->batch_cam, &mst, any_fog ? &f0 : nullptr);
            c->batch_stats[0]++;
        }
        const int r2 = b32_scene_swap(c, m);
        if (rc == B32_OK) rc = r2;
        i = k;
    }
    c->batch.clear();
    return rc;
}
unsigned long long b32_batch_count(const b32_ctx* c, int which) { return (c && which >= 0 && which < 4) ? c->batch_stats[which] : 0ull; }

// RasterTimings of a synchronous call: the per-phase split comes from the device-side phase clock (b32_frame_finish); the wall time of
// the whole call is reported as draw_ms only for an empty mesh, where no kernel ran.
static void wall_timing(b32_ctx* c, B32Timings* out, std::chrono::steady_clock::time_point t0) {
    if (!out || c->profile_level >= 2 || out->draw_ms > 0.0f || out->cull_ms > 0.0f) return;      // (the device phase clock filled them)
    out->draw_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
int b32_render_scene_15(b32_ctx* c, const B32Camera* cam, const B32Settings* st, const B32Fog* fog, B32Timings* out) {
    if (!c) return B32_E_ARG;
    const auto t0 = std::chrono::steady_clock::now();
    int rc = b32_render_scene_15_async(c, cam, st, fog);
    if (rc == B32_OK) rc = b32_frame_finish(c, out);
    if (rc == B32_OK) wall_timing(c, out, t0);
    return rc;
}

int b32_render_scene(b32_ctx* c, const B32Camera* cam, const B32Settings* st, B32Timings* out) {
    if (!c) return B32_E_ARG;
    const auto t0 = std::chrono::steady_clock::now();
    int rc = b32_render_scene_async(c, cam, st);
    if (rc == B32_OK) rc = b32_frame_finish(c, out);
    if (rc == B32_OK) wall_timing(c, out, t0);
    return rc;
}

int b32_render_mesh(b32_ctx* c, const B32Vertex* v, uint32_t nv, const B32Face* f, uint32_t nf, const B32Texture* tex, uint32_t nt,
                    const B32Camera* cam, const B32Settings* st, B32Timings* out) {
    if (!c || !cam || !st || !c->fb) return B32_E_ARG;
    int rc = validate_settings(st);
    if (rc) return rc;
    c->defer_upload_sync = true;
    stage_begin(c);
    rc = b32_scene_upload_rgba(c, v, nv, f, nf, tex, nt);
    stage_flush(c);
    c->defer_upload_sync = false;
    if (rc) { (void)hipStreamSynchronize(c->stream); return rc; }
    rc = b32_render_scene(c, cam, st, out);
    if (rc != B32_OK) (void)hipStreamSynchronize(c->stream);      // the caller's buffers must be free of pending copies on every exit
    return rc;
}

int b32_render_mesh_15(b32_ctx* c, const B32Vertex* v, uint32_t nv, const B32Face* f, uint32_t nf, const B32Texture15* tex, uint32_t nt,
                       const B32Camera* cam, const B32Settings* st, const B32Fog* fog, B32Timings* out) {
    if (!c || !cam || !st || !c->fb) return B32_E_ARG;
    int rc = validate_settings(st);
    if (rc) return rc;
    c->defer_upload_sync = true;
    stage_begin(c);
    rc = b32_scene_upload(c, v, nv, f, nf, tex, nt);
    stage_flush(c);
    c->defer_upload_sync = false;
    if (rc) { (void)hipStreamSynchronize(c->stream); return rc; }
    rc = b32_render_scene_15(c, cam, st, fog, out);
    if (rc != B32_OK) (void)hipStreamSynchronize(c->stream);      // the caller's buffers must be free of pending copies on every exit
    return rc;
}

// ------------------------------------------------------------------ stage taps
int b32_project_fixed_batch(b32_ctx* c, const float* pos, uint32_t n, const B32Camera* cam, uint32_t w, uint32_t h,
                            int32_t* sx, int32_t* sy, float* z) {
    if (!c || !cam || (n && (!pos || !sx || !sy || !z))) return B32_E_ARG;
    if (!n) return B32_OK;
    (void)hipSetDevice(c->device);
    float* d_pos = nullptr; int32_t *d_sx = nullptr, *d_sy = nullptr; float* d_z = nullptr;
    Scratch tmp(c);
    int rc;
    if ((rc = tmp.upload(pos, (size_t)n * 3, &d_pos))) return rc;
    if ((rc = tmp.alloc(&d_sx, (size_t)n))) return rc;
    if ((rc = tmp.alloc(&d_sy, (size_t)n))) return rc;
    if ((rc = tmp.alloc(&d_z, (size_t)n))) return rc;
    launch_project_fixed(c->stream, d_pos, n, *cam, w, h, d_sx, d_sy, d_z);
    HIPCHK(c, hipMemcpyAsync(sx, d_sx, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(sy, d_sy, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(z, d_z, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return B32_OK;
}

int b32_last_draw_order(b32_ctx* c, uint32_t* face_idx, uint32_t cap, uint32_t* n) {
    if (!c || !n || c->frame_pending) return B32_E_ARG;
    (void)hipSetDevice(c->device);
    const uint32_t cnt = c->h_ctrl.n_visible;
    *n = cnt;
    if (c->last_local_sort && cnt) {     // the fast path never builds the global order: sort k_setup's keys now (tap only)
        const SortScratch sc{ c->block_hist, c->hist_blocks, c->digit_total };
        hipStream_t s = c->stream;
        launch_radix_pass(s, c->keys[0], nullptr, c->keys[1], c->vals[1], c->d_consts, c->nf, 0, 8, sc);
        launch_radix_pass(s, c->keys[1], c->vals[1], c->keys[0], c->vals[0], &c->d_ctrl->n_visible, c->nf, 8, 8, sc);
        launch_radix_pass(s, c->keys[0], c->vals[0], c->keys[1], c->vals[1], &c->d_ctrl->n_visible, c->nf, 16, 8, sc);
        launch_radix_pass(s, c->keys[1], c->vals[1], c->keys[0], c->vals[0], &c->d_ctrl->n_visible, c->nf, 24, 8, sc);
        c->last_local_sort = false;
    }
    const uint32_t m = cnt < cap ? cnt : cap;
    if (m && face_idx) {
        // the device works on record slots (k_setup packs each wave's survivors to the front of its 64 slots): back to face ids
        std::vector<uint32_t> fo(c->nf);
        HIPCHK(c, hipMemcpyAsync(face_idx, c->vals[0], (size_t)m * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(fo.data(), c->face_of, (size_t)c->nf * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (uint32_t i = 0; i < m; ++i) face_idx[i] = face_idx[i] < c->nf ? fo[face_idx[i]] : 0xFFFFFFFFu;
    }
    return B32_OK;
}

int b32_selftest_f32(b32_ctx* c, int op, const float* a, const float* b, const float* cc, float* out, uint32_t n) {
    if (!c || !a || !b || !cc || !out) return B32_E_ARG;
    if (!n) return B32_OK;
    (void)hipSetDevice(c->device);
    float* d[4] = { nullptr, nullptr, nullptr, nullptr };
    Scratch tmp(c);
    int rc;
    if ((rc = tmp.upload(a, (size_t)n, &d[0]))) return rc;
    if ((rc = tmp.upload(b, (size_t)n, &d[1]))) return rc;
    if ((rc = tmp.upload(cc, (size_t)n, &d[2]))) return rc;
    if ((rc = tmp.alloc(&d[3], (size_t)n))) return rc;
    launch_selftest(c->stream, op, d[0], d[1], d[2], d[3], n);
    HIPCHK(c, hipMemcpyAsync(out, d[3], (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return B32_OK;
}

int b32_device_constants(b32_ctx* c, const char** names, uint32_t* bits, uint8_t* is_f32, uint32_t cap, uint32_t* count,
                         uint8_t* unr_table257, int32_t* dither16) {
    if (!c) return B32_E_ARG;
    (void)hipSetDevice(c->device);
    struct Entry { const char* name; uint8_t f; };
#define B32_K_F 1
#define B32_K_I 0
#define B32_K_ENTRY(name, kind, v) { name, B32_K_##kind },
    static const Entry kEntries[] = { B32_CONSTANTS(B32_K_ENTRY) };
#undef B32_K_ENTRY
#undef B32_K_F
#undef B32_K_I
    constexpr uint32_t N = sizeof(kEntries) / sizeof(kEntries[0]);
    if (count) *count = N;
    uint32_t* d_bits = nullptr; uint8_t* d_unr = nullptr; int32_t* d_dither = nullptr;
    Scratch tmp(c);
    int rc;
    if ((rc = tmp.alloc(&d_bits, (size_t)N))) return rc;
    if ((rc = tmp.alloc(&d_unr, (size_t)K::UNR_ENTRIES))) return rc;
    if ((rc = tmp.alloc(&d_dither, (size_t)16))) return rc;
    launch_constants(c->stream, d_bits, d_unr, d_dither);
    std::vector<uint32_t> h(N);
    HIPCHK(c, hipMemcpyAsync(h.data(), d_bits, N * 4, hipMemcpyDeviceToHost, c->stream));
    if (unr_table257) HIPCHK(c, hipMemcpyAsync(unr_table257, d_unr, K::UNR_ENTRIES, hipMemcpyDeviceToHost, c->stream));
    if (dither16) HIPCHK(c, hipMemcpyAsync(dither16, d_dither, 16 * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (uint32_t i = 0; i < N && i < cap; ++i) {
        if (names) names[i] = kEntries[i].name;
        if (bits) bits[i] = h[i];
        if (is_f32) is_f32[i] = kEntries[i].f;
    }
    return B32_OK;
}

int b32_last_kernel_times(b32_ctx* c, const char** names, float* ms, uint32_t cap) {
    if (!c || !names || !ms) return 0;
    static const char* const kNames[5] = { "setup", "sort", "bin", "cover", "shade" };
    if (!c->phase_frames) return 0;
    uint32_t k = 0;
    for (int p = 0; p < 5 && k < cap; ++p) {
        if (p != 3 && c->phase_level < 2) continue;
        names[k] = kNames[p]; ms[k] = c->phase_ms[p]; ++k;
    }
    return (int)k;
}

}  // extern "C"

// 0 = no events, 1 = events around k_fill, 2 = events around every phase.
extern "C" int b32_set_fragment_counting(b32_ctx* c, int on) {
    if (!c) return B32_E_ARG;
    c->count_fragments = on ? 1 : 0;
    return B32_OK;
}
extern "C" unsigned long long b32_route_count(const b32_ctx* c, int which) {
    if (c && which == 7) return c->pipelined_frames;
    if (c && which == 8) return c->lds_atlas_frames;
    if (c && which == 9) return c->wire_tile_frames;
    return (c && which >= 0 && which < 8) ? c->routes[which] : 0ull;
}
extern "C" int b32_set_async_depth(b32_ctx* c, int deep) {
    if (!c) return B32_E_ARG;
    if (!deep) { const int rc = settle_pending(c); if (rc) return rc; }
    c->deep_async = deep != 0;
    return B32_OK;
}
extern "C" int b32_set_profiling(b32_ctx* c, int level) {
    if (!c) return B32_E_ARG;
    c->profile_level = level < 0 ? 0 : (level > 2 ? 2 : level);
    c->prof_seq = 0;
    if (c->profile_level >= 1 && !c->ev_created) {      // (here, not in the first profiled frame: 384 hipEventCreate calls are ~0.2 ms of host time)
        (void)hipSetDevice(c->device);
        for (auto& fr : c->ev) for (auto& e : fr) HIPCHK(c, hipEventCreate(&e));
        c->ev_created = true;
    }
    return B32_OK;
}
extern "C" int b32_set_routes(b32_ctx* c, uint32_t off_mask) {
    if (!c) return B32_E_ARG;
    const int rc = settle_pending(c);
    if (rc) return rc;
    c->route_off = off_mask;
    return B32_OK;
}
extern "C" int b32_set_cheap_threshold(b32_ctx* c, uint32_t den) {
    if (!c || den == 0) return B32_E_ARG;
    c->cheap_den = den;          // (applies to the textures uploaded from now on:
    c->tex_sig_valid = false;    //  the next upload re-counts the skippable texels even if the pool already holds the same textures)
    return B32_OK;
}
extern "C" int b32_set_pipeline_gate(b32_ctx* c, uint32_t permille) {
    if (!c || permille > 2000u) return B32_E_ARG;
    c->gate_permille = permille;
    return B32_OK;
}
extern "C" int b32_last_shader_clock(const b32_ctx* c, float* ghz, float* fill_ms) {
    if (!c || !ghz) return B32_E_ARG;
    *ghz = 0.0f; if (fill_ms) *fill_ms = 0.0f;
    const unsigned long long* t = c->h_stamps.t;                      // of the last frame b32_frame_finish read back
    if (!t[ST_FILL] || !t[ST_CLKW] || t[ST_CLKW] <= t[ST_FILL] || t[ST_CLK1] <= t[ST_CLK0]) return B32_OK;      // (no fused kernel in that frame)
    const double ns = (double)(t[ST_CLKW] - t[ST_FILL]) * 10.0;       // wall_clock64: 100 MHz
    *ghz = (float)((double)(t[ST_CLK1] - t[ST_CLK0]) / ns);
    if (fill_ms) *fill_ms = (float)(ns * 1e-6);
    return B32_OK;
}
extern "C" int b32_transparent_counts(const b32_ctx* c, uint32_t* host_bound, uint32_t* device_last) {
    if (!c || !host_bound || !device_last) return B32_E_ARG;
    *host_bound = c->blend_faces; *device_last = c->h_ctrl.n_transparent;
    return B32_OK;
}
extern "C" int b32_set_pipeline_depth(b32_ctx* c, uint32_t sets) {
    if (!c || sets < 2u || sets > 3u) return B32_E_ARG;
    if (sets == c->n_sets) return B32_OK;
    (void)hipSetDevice(c->device);
    // everything in flight ends first: the ring's order (alt[0] oldest) only means something for one depth
    const int rc = b32_frame_finish(c, nullptr);
    if (rc == B32_E_HIP || rc == B32_E_ARG) return rc;
    if (rc && !c->deferred_rc) c->deferred_rc = rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->side) HIPCHK(c, hipStreamSynchronize(c->side));
    c->n_sets = sets;
    return B32_OK;
}
extern "C" int b32_set_profiling_stride(b32_ctx* c, uint32_t every) {
    if (!c) return B32_E_ARG;
    c->prof_stride = every ? every : 1u;
    c->prof_seq = 0;
    return B32_OK;
}

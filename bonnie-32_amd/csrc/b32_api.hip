// b32_api.hip -- the C ABI of include/b32raster.h, part 1: context, stream, device-resident framebuffer, test taps and switches.
// (Scene uploads: b32_scene.hip; frames: b32_frame.hip; batched frames: b32_batch.hip; shared declarations: b32_host.h.)
#include "b32_host.h"

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
        case B32_E_BAND_TIMEOUT: return "a band exchange wait gave up: another rank did not publish / release its frame in time (the frame may hold stale rows)";
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
    band_close_any(c);
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
    for (hipEvent_t e : { c->ev_main, c->ev_wbin, c->ev_setup, c->ev_done, c->alt[0].ev_setup, c->alt[0].ev_done, c->alt[1].ev_setup, c->alt[1].ev_done }) if (e) (void)hipEventDestroy(e);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->ev_created) for (auto& fr : c->ev) for (auto& e : fr) if (e) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->stage_host) (void)hipHostFree(c->stage_host);
    for (hipEvent_t e : c->dl_ev) if (e) (void)hipEventDestroy(e);
    if (c->dl_stream) { (void)hipStreamSynchronize(c->dl_stream); (void)hipStreamDestroy(c->dl_stream); }
    for (hipEvent_t e : c->dl_snap) if (e) (void)hipEventDestroy(e);
    for (uint32_t* q : c->dl_stage) if (q) (void)hipFree(q);
    delete c;
}

int b32_last_hip_error(const b32_ctx* c) { return c ? c->last_hip : 0; }
#ifndef B32_SRC_DIGEST
#define B32_SRC_DIGEST "unknown-digest!!"
#endif
// (the marker in front lets build.py read the digest from the file without loading it)
static const char g_build_digest[] = "B32-SRC-DIGEST:" B32_SRC_DIGEST;
const char* b32_build_digest(void) { return g_build_digest + 15; }

// A pending frame that may still need a redraw (pair overflow, long transparent lists) is settled before anything reads or rebinds
// the framebuffer, so that no caller ever sees the cleared frame of an aborted attempt.  Its error, if any, is the frame's error: kept
// for the b32_frame_finish that ends the frame.
int settle_pending(b32_ctx* c) {
    if (!c->frame_pending || !c->pending_may_redraw) return B32_OK;
    const int rc = b32_frame_finish(c, nullptr);
    if (rc == B32_E_HIP || rc == B32_E_ARG) return rc;
    if (rc && !c->deferred_rc) c->deferred_rc = rc;
    return B32_OK;
}

// The deferred Framebuffer::clear as launches of its own: before anything but the sort-free frame reads or writes the framebuffer.
int flush_clear(b32_ctx* c) {
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
    return c->n_sets_user ? B32_OK : apply_depth_auto(c);       // (the band's share of the frame decides the library's own pipeline depth)
}
static int fb_resize_any(b32_ctx* c, uint32_t w, uint32_t h, bool always_new) {
    if (!c || w == 0 || h == 0 || w > 16384 || h > 16384) return B32_E_ARG;
    (void)hipSetDevice(c->device);
    { const int rc = settle_pending(c); if (rc) return rc; }
    { const int rcf = flush_clear(c); if (rcf) return rcf; }
    if (c->fb_external) { c->fb_external = false; c->fb = nullptr; c->width = c->height = 0; }
    if (!always_new && c->fb && c->width == w && c->height == h) return B32_OK;          // Framebuffer::resize: no-op on equal dims
    // an exported framebuffer that changes size (or is made anew) is no longer the one the ranks mapped: the epoch words sit behind the
    // pixels at an offset that depends on the size, and the old mapping may be freed below -- the root must b32_band_export again
    // (until then b32_band_wait / _release / _status return B32_E_ARG instead of touching pixels or freed memory)
    if (c->band_sync_own) { c->band_sync_own = nullptr; if (c->band_rank == 0) c->band_sync = nullptr; }
    const size_t px = (size_t)w * h;
    if (px > c->fb_own_px || !c->fb_own) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->fb_own) HIPCHK(c, hipFree(c->fb_own));
        c->fb_own = nullptr;
        c->band_sync_own = nullptr;                                          // (the epoch words of an exported framebuffer lived in the old allocation)
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->fb_own), px * 4 + 4096 + FB_TAIL_BYTES));
        c->fb_own_px = px;
    }
    c->fb = c->fb_own;
    c->band_set = false;
    c->zbuf_valid = false;                                                  // vec![f32::MAX; w*h], render.rs:22,32
    { const int rcd = fb_set_dims(c, w, h); if (rcd) return rcd; }
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
    if (!c->n_sets_user) return apply_depth_auto(c);        // (a narrow band runs three frame sets: b32_set_pipeline_depth)
    return B32_OK;
}
// (safe mode) a pending large-scene frame that may still need a redraw is settled before anything else WRITES the framebuffer too:
// redrawn after a clear or a sky pass it would put its pixels on top of them
static int settle_before_write(b32_ctx* c) { return c->deep_async ? B32_OK : settle_pending(c); }

int b32_fb_clear(b32_ctx* c, uint8_t r, uint8_t g, uint8_t b, uint8_t blend) {
    if (!c || !c->fb) return B32_E_ARG;
    (void)hipSetDevice(c->device);
    // Safe mode settles a pending large-scene frame before anything writes the framebuffer -- except here: this clear covers every row the
    // pending frame can have drawn (its band; a band change settles) and resets their depths, so whether that frame was dropped or not
    // can no longer be seen.  It is marked superseded instead, and the draw that follows is enqueued behind it like in deep mode (the
    // setup kernel of the new frame beside the fill of the old one): the reference's loop -- clear, draw, clear, draw -- runs without a
    // host synchronisation per frame in the library's DEFAULT mode too.  Anything that reads the framebuffer in between still settles.
    // Only for a framebuffer that can be read through this library alone (every such read settles): memory the caller bound
    // (b32_fb_bind_device: a torch tensor, an RCCL gather) or shares with other ranks (b32_band_export / _import / _attach) is read behind
    // the library's back -- there the pending frame is settled here as before, so that an overflowing first frame is redrawn and its
    // capacities grow instead of every later frame of a clear / draw loop overflowing the same way unseen.
    const bool only_ours = !c->fb_external && !c->band_sync && !c->band_sync_own;
    if (!c->deep_async && c->frame_pending && c->pending_may_redraw && only_ours) c->pending_superseded = true;
    else { const int rc = settle_before_write(c); if (rc) return rc; }
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

// The presenter's copy without a host round trip per frame: the reference hands fb.pixels to the screen EVERY frame
// (game/renderer.rs:179-214); here the copy engine moves the frame into page-locked caller memory behind everything enqueued so far and
// a ticket tells when it has landed, so the host can enqueue frame i + 1 while frame i is drawn and copied.
void* b32_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (!bytes || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void b32_host_free(void* p) { if (p) (void)hipHostFree(p); }
int b32_fb_download_async(b32_ctx* c, uint8_t* rgba, uint64_t* ticket) {
    if (!c || !c->fb || !rgba || !ticket) return B32_E_ARG;
    (void)hipSetDevice(c->device);
    // (safe mode: a pending frame that may still need a redraw is settled first, as b32_fb_download does -- the frames of small meshes, what
    // a console frame is made of, never are; deep mode never blocks the host: a dropped frame is reported by b32_frame_finish)
    if (!c->deep_async) { const int rc = settle_pending(c); if (rc) return rc; }
    { const int rcf = flush_clear(c); if (rcf) return rcf; }
    const unsigned long long t = c->dl_seq + 1;
    hipEvent_t& ev = c->dl_ev[t % b32_ctx::DL_RING];
    if (!ev) HIPCHK(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    else if (t > b32_ctx::DL_RING) HIPCHK(c, hipEventSynchronize(ev));       // (the ticket that used this event, DL_RING downloads ago)
    const size_t px = (size_t)c->width * c->height;
    // a snapshot on the device first (two staging buffers, alternating): the frames that follow overwrite the framebuffer while the
    // snapshot crosses PCIe on a stream of its own
    if (!c->dl_stream) {
        HIPCHK(c, hipStreamCreateWithFlags(&c->dl_stream, hipStreamNonBlocking));
        for (hipEvent_t& e : c->dl_snap) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    if (px > c->dl_stage_px) {
        HIPCHK(c, hipStreamSynchronize(c->dl_stream));
        for (uint32_t*& q : c->dl_stage) { if (q) HIPCHK(c, hipFree(q)); q = nullptr; HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&q), px * 4)); }
        c->dl_stage_px = px;
    }
    const int k = (int)(t & 1u);
    // (the staging buffer's previous reader -- ticket t - 2's transfer -- must have left: the main stream waits for it, normally long done)
    if (t > 2) HIPCHK(c, hipStreamWaitEvent(c->stream, c->dl_ev[(t - 2) % b32_ctx::DL_RING], 0));
    HIPCHK(c, hipMemcpyAsync(c->dl_stage[k], c->fb, px * 4, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipEventRecord(c->dl_snap[k], c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->dl_stream, c->dl_snap[k], 0));
    HIPCHK(c, hipMemcpyAsync(rgba, c->dl_stage[k], px * 4, hipMemcpyDeviceToHost, c->dl_stream));
    HIPCHK(c, hipEventRecord(ev, c->dl_stream));
    c->dl_seq = t; *ticket = t;
    return B32_OK;
}
int b32_ticket_poll(b32_ctx* c, uint64_t ticket, int* done) {
    if (!c || !done || ticket == 0 || ticket > c->dl_seq) return B32_E_ARG;
    *done = 1;
    if (ticket + b32_ctx::DL_RING <= c->dl_seq) return B32_OK;               // (its event has been reused: it completed long ago)
    const hipError_t e = hipEventQuery(c->dl_ev[ticket % b32_ctx::DL_RING]);
    if (e == hipErrorNotReady) { *done = 0; (void)hipGetLastError(); return B32_OK; }
    if (e != hipSuccess) { c->last_hip = (int)e; return B32_E_HIP; }
    return B32_OK;
}
int b32_ticket_wait(b32_ctx* c, uint64_t ticket) {
    if (!c || ticket == 0 || ticket > c->dl_seq) return B32_E_ARG;
    if (ticket + b32_ctx::DL_RING <= c->dl_seq) return B32_OK;
    HIPCHK(c, hipEventSynchronize(c->dl_ev[ticket % b32_ctx::DL_RING]));
    return B32_OK;
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
    if (c && which == 10) return c->span_cover_frames;
    if (c && which == 11) return c->flag_join_frames;
    if (c && which == 12) return c->event_join_frames;
    if (c && which == 13) return c->poll_join_frames;
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
// The depth the library picks itself (b32_set_pipeline_depth(ctx, 0), the default): two frame sets, three for a NARROW band -- at most a sixth
// of the frame's rows, one rank of a frame sharded over six or more GPUs.  Such a rank still transforms the whole mesh and its frame is
// bound by the setup kernel, not by the fill: with three sets the setup kernels run back to back on the side stream instead of each waiting
// for the previous frame's fill to start (240 rows of C3, 1 of 8 ranks: 0.040 -> 0.034-0.035 ms per frame; 480 rows: the same either way; 960
// rows and the whole frame: two sets are faster -- profiles/r06_band_depth.txt).
static uint32_t auto_depth(const b32_ctx* c) {
    return (c->band_set && c->height && (uint64_t)(c->band_y1 - c->band_y0) * 6u <= c->height) ? 3u : 2u;
}
static int apply_depth(b32_ctx* c, uint32_t sets) {
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
extern "C" int b32_set_pipeline_depth(b32_ctx* c, uint32_t sets) {
    if (!c || sets == 1u || sets > 3u) return B32_E_ARG;
    c->n_sets_user = sets;
    return apply_depth(c, sets ? sets : auto_depth(c));
}
extern "C" int apply_depth_auto(b32_ctx* c) { return apply_depth(c, auto_depth(c)); }
extern "C" int b32_debug_inject(b32_ctx* c, uint32_t what) {
    if (!c || (what & ~3u)) return B32_E_ARG;
    c->inject |= what;
    return B32_OK;
}
extern "C" int b32_set_profiling_stride(b32_ctx* c, uint32_t every) {
    if (!c) return B32_E_ARG;
    c->prof_stride = every ? every : 1u;
    c->prof_seq = 0;
    return B32_OK;
}

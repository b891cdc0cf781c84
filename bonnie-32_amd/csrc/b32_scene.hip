// b32_scene.hip -- the C ABI, part 2: host -> device uploads (the drop-in calls' staged arena, resident scenes with Texture15 / Texture /
// index atlas + CLUT), scene slots, and the synchronous drop-in calls render_mesh[_15] / render_scene[_15].
#include "b32_host.h"

extern "C" {
// ------------------------------------------------------------------ scene upload
// Host -> device copy of an upload.  Inside a drop-in call (stage_active) the bytes are packed into the pinned arena and moved later
// by one kernel (stage_flush); a copy that does not fit, or any other caller, takes the stream's ordinary async copy.  The arena copy
// rounds the length up to 16 B: every destination has at least 15 B of slack (ensure() allocates one element more than asked for, texel
// offsets are multiples of 16 B).
int h2d(b32_ctx* c, void* dst, const void* src, size_t bytes) {
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
static_assert(sizeof(Ctrl) == 64 && sizeof(Stamps) == 64, "Ctrl and Stamps are read back through a 128-byte slot of the pinned arena");
bool stage_ensure(b32_ctx* c) {
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
int ensure_work(b32_ctx* c, uint32_t nf) {
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
    c->pos_valid = false; c->lit_valid = false; c->band_frames = 0;
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
    std::swap(c->d_pos12, sl->d_pos12); std::swap(c->cap_pos12, sl->cap_pos12); std::swap(c->pos_valid, sl->pos_valid); std::swap(c->band_frames, sl->band_frames); std::swap(c->lit_valid, sl->lit_valid);
    c->tex_sig.swap(sl->tex_sig); std::swap(c->tex_sig_valid, sl->tex_sig_valid);
    return B32_OK;
}


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

}  // extern "C"

// b32_batch.hip -- the C ABI, part 4: a frame of several meshes (b32_frame_begin / _add_scene / _end).
#include "b32_host.h"

extern "C" {
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
    m->direct_cap_opaque = 0; m->direct_ntiles = 0; m->direct_ok = true; m->pos_valid = false; m->lit_valid = false; m->band_frames = 0;
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
// the whole frame in one call: what scene.rs:180-261 loops over, as a table
int b32_frame_submit(b32_ctx* c, const B32Camera* cam, const B32Settings* st, b32_scene* const* slots, const B32MeshParams* params, uint32_t n) {
    if (!c || (n && !slots)) return B32_E_ARG;
    int rc = b32_frame_begin(c, cam, st);
    for (uint32_t i = 0; i < n && !rc; ++i) rc = b32_frame_add_scene(c, slots[i], params ? &params[i] : nullptr);
    if (rc) { c->batch_open = false; c->batch.clear(); return rc; }
    return b32_frame_end(c);
}
unsigned long long b32_batch_count(const b32_ctx* c, int which) { return (c && which >= 0 && which < 4) ? c->batch_stats[which] : 0ull; }

}  // extern "C"

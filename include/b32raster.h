/*
 * b32raster.h — C ABI of the MI355X-native bonnie-32 rasterizer hot path.
 *
 * This is the drop-in boundary for the reference's `render_mesh_15`
 * (reference: src/rasterizer/render.rs:2302-2310, re-exported at
 * src/rasterizer/mod.rs:63) and the `Framebuffer` it draws into
 * (render.rs:10-45).  The reference has no FFI seam today; these entry points
 * are exactly what a Rust `extern "C"` block for that path would bind (see
 * INTEGRATION.md for the Rust-side stub).
 *
 * Plain pointers and sizes only; no torch / HIP types in any signature
 * (streams and device pointers travel as `void*`).
 *
 * The same POD structs are consumed by the CPU oracle (oracle/b32_oracle.c),
 * which is test infrastructure and never linked into the product library.
 */
#ifndef B32RASTER_H
#define B32RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes (reference behaviour in brackets) ---------------------- */
#define B32_OK             0
#define B32_E_ARG         -1  /* null pointer / zero-size framebuffer                      */
#define B32_E_INDEX       -2  /* face.v* >= nv            [index panic, render.rs:2375-2377] */
#define B32_E_NAN_KEY     -3  /* NaN painter's key        [unwrap panic, render.rs:2531]     */
#define B32_E_HIP         -4  /* HIP runtime failure (b32_last_hip_error has the code)      */
#define B32_E_UNSUPPORTED -5  /* wireframe edge >= 2^30 px (i32 overflow in the reference); more than 65534 textures */
#define B32_E_NO_DEVICE   -6  /* no gfx950 device / kernels missing: the product path never falls back to CPU   */
#define B32_E_FRAME_DROPPED -7 /* deep asynchronous mode only (b32_set_async_depth): an EARLIER frame in flight ran out of buffer
                                  space and drew nothing (its framebuffer kept the cleared / previous contents); the most recent
                                  frame has been redrawn correctly */
#define B32_E_BAND_TIMEOUT -8  /* multi-GPU band exchange: a b32_band_wait / _wait_all / _acquire since the last b32_frame_finish gave up
                                  on another rank's epoch word (the frame may hold that rank's rows of an older frame) */

/* ---- enums mirrored as integers ------------------------------------------ */
/* BlendMode, types.rs:1380-1388 */
#define B32_BLEND_OPAQUE      0
#define B32_BLEND_AVERAGE     1
#define B32_BLEND_ADD         2
#define B32_BLEND_SUBTRACT    3
#define B32_BLEND_ADD_QUARTER 4
#define B32_BLEND_ERASE       5
/* ShadingMode, types.rs:1289-1294 */
#define B32_SHADE_NONE    0
#define B32_SHADE_FLAT    1
#define B32_SHADE_GOURAUD 2
/* LightType, types.rs:1297-1304 */
#define B32_LIGHT_DIRECTIONAL 0
#define B32_LIGHT_POINT       1
#define B32_LIGHT_SPOT        2   /* render.rs:1038-1058; f32::acos as the `libm` crate (musl acosf.c) computes it -- the reference's wasm32 target */

#define B32_NO_TEXTURE 0xFFFFFFFFu

/* ---- POD mirrors of the reference API types ------------------------------ */

/* Vertex, types.rs:947-959 (bone_index is editor-only and dropped). 36 B. */
typedef struct B32Vertex {
    float   pos[3];
    float   uv[2];
    float   normal[3];
    uint8_t r, g, b, blend;      /* Color{r,g,b,blend}, types.rs:721-726 */
} B32Vertex;

/* Face, types.rs:984-1002. 20 B. texture_id: Option<usize> -> B32_NO_TEXTURE = None. */
typedef struct B32Face {
    uint32_t v[3];
    uint32_t texture_id;
    uint8_t  black_transparent;
    uint8_t  blend_mode;
    uint8_t  editor_alpha;
    uint8_t  _pad;
} B32Face;

/* Texture15, types.rs:532-539. `pixels` is a HOST pointer to width*height Color15 (u16). */
typedef struct B32Texture15 {
    uint32_t        width, height;
    uint32_t        blend_mode;
    uint32_t        _pad;
    const uint16_t* pixels;
} B32Texture15;

/* Texture, types.rs:1166-1176 (the 8-bit-colour path): `pixels` is a HOST pointer to width*height Color values, 4 bytes
 * each = r, g, b, blend (Color{r,g,b,blend: BlendMode}, types.rs:721-726; blend == B32_BLEND_ERASE is a transparent texel). */
typedef struct B32Texture {
    uint32_t       width, height;
    uint32_t       blend_mode;
    uint32_t       _pad;
    const uint8_t* pixels;
} B32Texture;

/* IndexedAtlas + Clut (modeler/mesh_editor.rs:594-682, types.rs:390-397): one byte per texel for
 * both 4- and 8-bit depths; out-of-range index -> 0x0000.  Expanded with Clut::lookup semantics
 * exactly as IndexedAtlas::to_texture15 does before the rasterizer sees it (scene.rs:164). */
typedef struct B32IndexedTexture {
    uint32_t        width, height;
    uint32_t        blend_mode;
    uint32_t        clut_len;     /* 16 or 256 */
    const uint8_t*  indices;      /* HOST pointer, width*height */
    const uint16_t* clut;         /* HOST pointer, clut_len Color15 */
} B32IndexedTexture;

/* Camera, camera.rs:9-18: basis vectors are inputs (sin/cos stay on the host). */
typedef struct B32Camera {
    float position[3];
    float basis_x[3];
    float basis_y[3];
    float basis_z[3];
} B32Camera;

/* Light, types.rs:1306-1314 */
typedef struct B32Light {
    uint32_t type;
    float    position[3];
    float    direction[3];
    float    radius;
    float    angle;
    float    intensity;
    uint8_t  r, g, b, enabled;
} B32Light;

/* RasterSettings, types.rs:1392-1428 (low_resolution / stretch_to_fill are presentation-only). */
typedef struct B32Settings {
    uint8_t affine_textures;
    uint8_t use_zbuffer;
    uint8_t shading;
    uint8_t backface_cull;
    uint8_t backface_wireframe;
    uint8_t dithering;
    uint8_t wireframe_overlay;
    uint8_t use_rgb555;
    uint8_t use_fixed_point;
    uint8_t xray_mode;
    uint8_t has_ortho;           /* ortho_projection.is_some() */
    uint8_t _pad;
    float   ambient;
    float   ortho_zoom, ortho_center_x, ortho_center_y;
    uint32_t        n_lights;
    const B32Light* lights;      /* HOST pointer */
} B32Settings;

/* fog: Option<(f32,f32,f32,Color)> of render_mesh_15 (render.rs:2309); NULL pointer = None. */
typedef struct B32Fog {
    float   start, falloff, cull_distance;
    uint8_t r, g, b, blend;
} B32Fog;

/* RasterTimings, types.rs:1499-1514, plus the exact fragment-store count used for Mpixels/s. */
typedef struct B32Timings {
    float    transform_ms, fog_ms, cull_ms, sort_ms, draw_ms, wireframe_ms;
    uint32_t triangles_drawn;    /* opaque.len()+transparent.len(), render.rs:2545 */
    uint32_t tile_pairs;         /* (surface, 64x64 tile) pairs binned this frame (work unit of the coverage kernel) */
    uint64_t fragments;          /* pixel stores reached in rasterize_triangle_15 (render.rs:1671-1702) */
} B32Timings;

typedef struct b32_ctx b32_ctx;

/* ---- context -------------------------------------------------------------- */
/* One ctx = one device = one caller thread at a time.  Fails with B32_E_NO_DEVICE when no HIP
 * device is visible; there is no CPU fallback. */
int         b32_create(int device, b32_ctx** out);
void        b32_destroy(b32_ctx* ctx);
const char* b32_strerror(int code);
int         b32_last_hip_error(const b32_ctx* ctx);
/* Digest (16 hex digits) of the sources and compiler flags this library was built from (bonnie-32_amd/build.py: csrc_digest; no
 * reference counterpart).  bench.py and __graft_entry__.smoke() refuse to run a library whose digest is not the source tree's. */
const char* b32_build_digest(void);
/* Run on a caller-owned hipStream_t (e.g. torch's current stream); NULL = the ctx's own non-blocking stream.  The legacy default
 * stream is not NULL here: pass hipStreamLegacy ((hipStream_t)1) for it (torch reports its default stream as handle 0). */
int         b32_set_stream(b32_ctx* ctx, void* hip_stream);
int         b32_synchronize(b32_ctx* ctx);

/* ---- Framebuffer (render.rs:10-45) --------------------------------------- */
int b32_fb_resize(b32_ctx* ctx, uint32_t width, uint32_t height);           /* Framebuffer::resize :27-34: zero-filled on change, no-op on equal dimensions */
int b32_fb_new(b32_ctx* ctx, uint32_t width, uint32_t height);              /* Framebuffer::new :18-25: ALWAYS zero pixels and an f32::MAX z-buffer */
/* Framebuffer::clear :36-45 (rows of the band only when b32_set_band is active).  The clear is deferred inside the library: the draw
 * that follows folds it into its fused kernel when it can (painter's mode, no depth buffer allocated, same band) -- the frame then has
 * no clear launch and pixels nobody draws are written once -- and every other call that reads or writes the framebuffer, changes the
 * band, the binding or the stream, b32_synchronize and b32_frame_finish first turn it into the launches it stands for.  Through this
 * API the deferral is unobservable; code that reads a caller-bound device buffer (b32_fb_bind_device) DIRECTLY sees the clear after
 * the next draw, b32_synchronize or b32_frame_finish, in stream order. */
int b32_fb_clear(b32_ctx* ctx, uint8_t r, uint8_t g, uint8_t b, uint8_t blend);
int b32_fb_upload(b32_ctx* ctx, const uint8_t* rgba);                        /* host fb.pixels -> device */
int b32_fb_download(b32_ctx* ctx, uint8_t* rgba);                            /* device -> host fb.pixels */
/* The presenter's copy WITHOUT a host round trip per frame.  The reference hands fb.pixels to the screen every frame
 * (game/renderer.rs:179-214); b32_fb_download blocks the host until the frame is there.  b32_fb_download_async only ENQUEUES the copy --
 * behind everything enqueued on the context so far (a deferred clear is flushed; in safe mode a pending frame that may still need a
 * redraw is settled first, which frames of small meshes never do) -- into page-locked memory of the caller, and hands out a ticket:
 *     frame i:   b32_fb_clear; b32_frame_submit(...) or the draws; b32_fb_download_async(ctx, pinned[i & 1], &ticket[i & 1]);
 *                b32_ticket_wait(ctx, ticket[(i - 1) & 1]);   present pinned[(i - 1) & 1]      (frame i is being drawn meanwhile)
 * b32_ticket_poll never blocks (*done = 0 / 1); b32_ticket_wait blocks until the copy has landed.  At most 8 tickets are outstanding per
 * context: a ninth download first waits for the oldest.  `rgba` must stay valid until its ticket is done; memory from b32_host_alloc
 * (page-locked; NULL when it cannot be had) keeps the copy asynchronous, pageable memory works but may block the call.
 * Errors of the frames are reported by b32_frame_finish as ever. */
void* b32_host_alloc(size_t bytes);
void b32_host_free(void* p);
int b32_fb_download_async(b32_ctx* ctx, uint8_t* rgba, uint64_t* ticket);
int b32_ticket_poll(b32_ctx* ctx, uint64_t ticket, int* done);
int b32_ticket_wait(b32_ctx* ctx, uint64_t ticket);
/* Framebuffer::zbuffer (render.rs:12), used when settings.use_zbuffer: f32 per pixel, f32::MAX after new/resize/clear. */
int b32_zbuffer_download(b32_ctx* ctx, float* z);
int b32_zbuffer_upload(b32_ctx* ctx, const float* z);
/* Draw into caller-owned DEVICE memory (width*height*4 B, e.g. a torch uint8 tensor) instead of the
 * ctx-owned buffer; pass NULL to return to the ctx-owned buffer. */
int b32_fb_bind_device(b32_ctx* ctx, void* device_rgba, uint32_t width, uint32_t height);
int b32_fb_size(const b32_ctx* ctx, uint32_t* width, uint32_t* height);
/* Multi-GPU screen-band sharding: only rows [y0,y1) are filled by this ctx (default = whole frame). */
int b32_set_band(b32_ctx* ctx, uint32_t y0, uint32_t y1);

/* ---- render_mesh_15 (render.rs:2302-2638) -------------------------------- */
/* Drop-in form: host slices in, framebuffer stays device resident (read-modify-write).
 * Texture cache: the texel pool of the previous call is reused when every texture has the same host pointer, dimensions, blend mode and
 * 64-bit content hash (the texels are re-hashed on every call, so an in-place edit is seen); an edit that collides in the hash (2^-64
 * per edit, non-cryptographic) would draw stale texels -- b32_set_routes(B32_ROUTE_TEX_CACHE) uploads the texels on every call. */
int b32_render_mesh_15(b32_ctx* ctx,
                       const B32Vertex* vertices, uint32_t nv,
                       const B32Face* faces, uint32_t nf,
                       const B32Texture15* textures, uint32_t nt,
                       const B32Camera* camera, const B32Settings* settings,
                       const B32Fog* fog /* nullable */,
                       B32Timings* out /* nullable */);

/* Resident form (scene.rs:112-261 step-before): upload a mesh once, draw it many times. */
int b32_scene_upload(b32_ctx* ctx,
                     const B32Vertex* vertices, uint32_t nv,
                     const B32Face* faces, uint32_t nf,
                     const B32Texture15* textures, uint32_t nt);
/* Same, textures given as index atlas + CLUT; the expansion (Clut::lookup) runs on the device. */
int b32_scene_upload_indexed(b32_ctx* ctx,
                             const B32Vertex* vertices, uint32_t nv,
                             const B32Face* faces, uint32_t nf,
                             const B32IndexedTexture* textures, uint32_t nt);
int b32_render_scene_15(b32_ctx* ctx,
                        const B32Camera* camera, const B32Settings* settings,
                        const B32Fog* fog /* nullable */,
                        B32Timings* out /* nullable */);
/* Asynchronous draw of the resident scene: enqueue only, no host sync, no timings. Errors and counters
 * of the most recent frame are collected by b32_frame_finish (which synchronizes the stream). */
int b32_render_scene_15_async(b32_ctx* ctx,
                              const B32Camera* camera, const B32Settings* settings,
                              const B32Fog* fog /* nullable */);
int b32_frame_finish(b32_ctx* ctx, B32Timings* out /* nullable */);
/* How many frames of a LARGE scene (more than 8192 faces, or more than 2048 with a transparent pass) may be in flight.  Such a
 * frame can run out of tile-list space (the setup kernel bins it into fixed tile regions sized from the mesh; a region overflows when
 * far more of the mesh lands in one screen tile than the mean) or need the global depth sort; it then draws nothing and must be
 * redrawn by the host.
 *   deep = 0 (default): safe.  Enqueueing another frame, or any call that reads, writes or rebinds the framebuffer (b32_fb_download,
 *            b32_zbuffer_download, b32_fb_upload, b32_fb_clear*, b32_render_skybox_mesh, b32_draw_star_diamonds, b32_fb_bind_device,
 *            b32_set_stream, b32_present_nearest, b32_scene_upload*, b32_scene_swap, b32_set_band), first settles the pending frame (one host synchronisation, redraw if needed): no
 *            frame is ever lost, and none is redrawn on top of a later clear.  One exception that cannot be observed (round 5): a
 *            b32_fb_clear of the whole band behind a pending frame overwrites every pixel and depth that frame can have drawn, so the
 *            frame is marked superseded instead of settled and the draw that follows is enqueued behind it like in deep mode -- the
 *            reference's loop (clear, draw, clear, draw; game/renderer.rs:91-95) runs without a host synchronisation per frame in
 *            this default mode too.  Any call that READS the framebuffer in between still settles; errors of a superseded frame are
 *            still reported by the next b32_frame_finish.  The exception applies only to a framebuffer nobody can read behind the
 *            library's back: the library's own allocation (b32_fb_new / _resize), not exported to other ranks.  With caller-bound memory
 *            (b32_fb_bind_device) or a framebuffer shared by b32_band_export / _import / _attach the clear settles the pending frame as
 *            every other write does.  A superseded frame whose clear has meanwhile been executed (b32_synchronize) is never redrawn.
 *   deep = 1: throughput.  Frames are enqueued back to back with no host synchronisation (bench.py and the tools that call
 *            b32_set_async_depth(ctx, 1): static camera, capacities settled by a warm-up frame).  Only the most recent frame can be redrawn; if an earlier one was dropped,
 *            b32_frame_finish reports B32_E_FRAME_DROPPED -- never silently.  Consumers outside the library that read the bound
 *            framebuffer between two frames (an RCCL gather on the same stream) must accept that contract.  A b32_fb_clear issued
 *            after a draw is applied after that draw even when b32_frame_finish has to redraw it.
 * Frames of small meshes never overflow and are always enqueued without synchronisation. */
int b32_set_async_depth(b32_ctx* ctx, int deep);
/* Which internal route the frames of this context took since it was created (tests assert that the route they target really ran;
 * no reference counterpart).  which: 0 frames binned by the setup kernel into fixed tile regions (large meshes), 1 frames whose tile
 * lists were collected inside the fill kernel (small meshes), 2 frames binned by the counting-sort launches, 3 frames through the
 * keyed pipeline (global depth sort), 4 frames redrawn because a tile region overflowed, 5 frames redrawn through the global depth
 * sort, 6 frames redrawn after a pair-buffer overflow, 7 frames whose setup kernel ran on the second stream beside the previous frame's
 * fill (two frames in flight), 8 frames whose fused kernel sampled the 4/8-bit index atlas + CLUT from LDS (B32_ROUTE_LDS_ATLAS),
 * 9 frames whose wireframe phases went through the tile route (B32_ROUTE_WIRE_TILES), 10 frames whose opaque coverage was decided by
 * exact row intervals (B32_ROUTE_SPAN_COVER), 11 pipelined frames whose setup kernel was handed over to the fill by the flag / join kernel
 * pair, 12 by a cross-stream event (main and side stream of one priority), 13 those of 11 whose fused kernel polled the flag itself (the merged
 * draws of a batched frame: no launch and no event in front of the fill).
 * Unknown `which` or null ctx: 0. */
unsigned long long b32_route_count(const b32_ctx* ctx, int which);
/* Switch internal routes OFF for the frames enqueued from now on (no reference counterpart: the results are identical on every route;
 * the tests use it to keep the older pipelines covered, the timing tools to compare routes).  off_mask = 0 restores the default. */
#define B32_ROUTE_SORT_FREE   1u   /* the fused sort-free kernel -> keyed pipelines (global painter's sort / per-tile LDS sort)    */
#define B32_ROUTE_CUT_TILES   2u   /* tiles of 32 / 16 rows when a frame has few 64x64 tiles                                      */
#define B32_ROUTE_INLINE_BIN  4u   /* small meshes: tile lists collected inside the fill kernel -> binning launches                */
#define B32_ROUTE_DIRECT_BIN  8u   /* large meshes: binning inside the setup kernel -> counting-sort launches                      */
#define B32_ROUTE_WIDE_GROUPS 16u  /* 16-wave workgroups of the fused kernel when tiles are few -> always 8 waves                  */
#define B32_ROUTE_PACKED_STREAMS 32u /* resident large meshes: packed position / attribute streams for the setup kernel -> B32Vertex array */
#define B32_ROUTE_TEX_CACHE   128u /* drop-in calls: texture cache by (pointer, size, blend mode, 64-bit content hash) -> texels uploaded on every call */
#define B32_ROUTE_BATCH       256u /* b32_frame_end: runs of commuting meshes drawn as one merged mesh -> one draw per mesh           */
#define B32_ROUTE_LDS_ATLAS   512u /* one indexed texture (b32_scene_upload_indexed): index atlas + CLUT staged in LDS by every workgroup of the fused
                                    * kernel and looked up per shaded pixel (Clut::lookup, types.rs:390-397) whenever they fit beside the tile planes
                                    * -> expanded Color15 texels fetched from global memory                                          */
#define B32_ROUTE_WIRE_TILES  1024u /* wireframe phases (render.rs:2574-2635): edges binned to 64x16 tiles, first occurrences found in an LDS table per tile,
                                    * lines walked into an LDS bit plane -> one global first-occurrence table + one lane per whole line        */
#define B32_ROUTE_SPAN_COVER  2048u /* sort-free CHEAP painter's coverage: for surfaces with integer vertices, |area| <= 8192 and edges <= 512 px the reference's
                                    * toleranced inside test (render.rs:1536-1542) equals the closed integer triangle, so every row's passing pixels are one
                                    * interval with integer-quotient ends -- no per-pixel test; other surfaces keep the per-pixel form -> per-pixel form for all */
#define B32_ROUTE_STAGGER     4096u /* fused kernel, frames with more tiles than workgroup slots: the second workgroup of every CU starts 4 us late (the two then
                                    * run coverage against shading instead of in step) -> all workgroups start together */
#define B32_ROUTE_PIPELINE    64u  /* setup kernel of the next frame on a second stream beside the fill of the current one -> one stream */
int b32_set_routes(b32_ctx* ctx, uint32_t off_mask);
/* CHEAP coverage (inside test only, texel rule applied to the winner) is used while every texture has at most 1/den skippable texels
 * (default 64); applies to textures uploaded after the call.  den = 0: B32_E_ARG. */
int b32_set_cheap_threshold(b32_ctx* ctx, uint32_t den);

/* Several resident scenes per context (scene.rs:112-261 draws room after room, asset part after asset part, onto one
 * framebuffer every frame): a slot owns one uploaded scene's device buffers.  b32_scene_swap exchanges the context's current
 * resident scene with the slot's content (either side may be empty), so
 *     upload A; swap(sA);  upload B; swap(sB);                                 -- once
 *     clear;  swap(sA); render_async; swap(sA);  swap(sB); render_async; swap(sB);  frame_finish      -- every frame
 * draws both meshes with no upload and no host synchronisation between them.  Errors of ANY frame enqueued since the last
 * b32_frame_finish are reported by it (B32_E_INDEX / B32_E_NAN_KEY / B32_E_UNSUPPORTED; as in the reference, the failing mesh
 * draws nothing); its counters are those of the most recent frame.  A pending frame of a large scene (more than 8192 faces, or
 * more than 2048 with a transparent pass: it may need a redraw with grown buffers) is finished by b32_scene_swap before the exchange; an error of that frame is kept and
 * reported by the b32_frame_finish that ends the frame. */
typedef struct b32_scene b32_scene;
int b32_scene_create(b32_ctx* ctx, b32_scene** out);
void b32_scene_destroy(b32_ctx* ctx, b32_scene* slot);
int b32_scene_swap(b32_ctx* ctx, b32_scene* slot);

/* ---- a frame of several meshes (scene.rs:112-261) -------------------------------------------------------------------------------
 * The console's render step is one render_mesh_15 call per room and per asset part onto the same framebuffer, with ONE camera and
 * light list per frame and per-mesh ambient, fog (per room, scene.rs:189-205) and backface culling (per part: double_sided,
 * scene.rs:133-137).  These three calls take that sequence -- the meshes resident in scene slots -- and produce the framebuffer (and
 * depth buffer) the sequential calls produce, bit for bit, but draw every run of meshes whose draws commute as ONE merged mesh (one
 * setup + fill kernel pair instead of one per mesh; a 12-room 320x240 frame is launch-latency bound otherwise):
 *   z-buffer mode with RGB555 output: runs of meshes, each run ended by the first mesh that has a transparent pass (semi-transparent
 *   faces blend against what was drawn before them, so that mesh keeps its place); painter's mode, the 8-bit-colour path, x-ray,
 *   orthographic views and wireframe phases: mesh by mesh, exactly as b32_render_scene_15_async would.
 * b32_frame_begin copies camera, settings and lights; b32_frame_add_scene appends a slot (which must HOLD its scene: not swapped into
 * the context) with its per-mesh parameters (NULL: the base settings' ambient and backface flags, no fog); b32_frame_end enqueues the
 * frame; b32_frame_finish reports errors as for any asynchronous frame.  Difference to the sequential calls in the ERROR case only: a
 * vertex index out of range or a NaN sort key in one mesh of a merged run (the reference panics there) leaves the whole run undrawn.
 * At most 32 meshes are merged into one draw; longer runs are split.  Merged meshes are cached per context while the member slots'
 * contents stay the same (any b32_scene_upload* into a member rebuilds). */
typedef struct B32MeshParams {
    float   ambient;
    uint8_t backface_cull, backface_wireframe, has_fog, _pad;
    B32Fog  fog;
} B32MeshParams;
int b32_frame_begin(b32_ctx* ctx, const B32Camera* camera, const B32Settings* base_settings);
int b32_frame_add_scene(b32_ctx* ctx, b32_scene* slot, const B32MeshParams* params /* nullable */);
int b32_frame_end(b32_ctx* ctx);
/* The same frame in ONE call: begin, n x add_scene (params[i], or the base settings' values when params is NULL), end. */
int b32_frame_submit(b32_ctx* ctx, const B32Camera* camera, const B32Settings* base_settings, b32_scene* const* slots,
                     const B32MeshParams* params /* nullable */, uint32_t n);
/* which: 0 merged draws, 1 mesh-by-mesh draws, 2 merged meshes built, 3 frames ended -- since the context was created (tests). */
unsigned long long b32_batch_count(const b32_ctx* ctx, int which);

/* ---- the 8-bit-colour path: render_mesh (render.rs:1971-2264) + rasterize_triangle (render.rs:1202-1433) ----
 * What every caller of the reference runs when settings.use_rgb555 is false (scene.rs:163-169).  Same pipeline and settings
 * as render_mesh_15 except: Texture texels are Color values with a per-texel blend mode, no fog, no opaque/transparent
 * partition (one depth sort of all surfaces in painter's mode), every passing fragment writes depth in z-buffer mode. */
int b32_render_mesh(b32_ctx* ctx,
                    const B32Vertex* vertices, uint32_t nv,
                    const B32Face* faces, uint32_t nf,
                    const B32Texture* textures, uint32_t nt,
                    const B32Camera* camera, const B32Settings* settings,
                    B32Timings* out /* nullable */);
int b32_scene_upload_rgba(b32_ctx* ctx,
                          const B32Vertex* vertices, uint32_t nv,
                          const B32Face* faces, uint32_t nf,
                          const B32Texture* textures, uint32_t nt);
/* Draws the scene uploaded by b32_scene_upload_rgba; finish with b32_frame_finish like the _15 form. */
int b32_render_scene(b32_ctx* ctx, const B32Camera* camera, const B32Settings* settings, B32Timings* out /* nullable */);
int b32_render_scene_async(b32_ctx* ctx, const B32Camera* camera, const B32Settings* settings);

/* ---- the steps around the mesh draw that the reference runs on the same framebuffer ----------
 * so that a whole frame can stay device resident.  The procedural inputs that use sin/cos/powf (Skybox::generate_mesh,
 * world/geometry.rs:529; star directions and twinkle, render.rs:166-196) are computed by the caller, like the camera basis. */
typedef struct B32SkyVertex { float pos[3]; uint8_t r, g, b, blend; } B32SkyVertex;       /* SkyboxVertex{pos, color} */
/* Framebuffer::clear_gradient, render.rs:58-77 (top at y = 0, bottom at y = height-1; also resets the z-buffer) */
int b32_fb_clear_gradient(b32_ctx* ctx, uint8_t top_r, uint8_t top_g, uint8_t top_b, uint8_t top_blend,
                          uint8_t bottom_r, uint8_t bottom_g, uint8_t bottom_b, uint8_t bottom_blend);
/* Framebuffer::clear_transparent, render.rs:47-56 */
int b32_fb_clear_transparent(b32_ctx* ctx);
/* Step 1 of Framebuffer::render_skybox, render.rs:81-134: project (math.rs:117-136), cull and fill the vertex-coloured sphere with
 * rasterize_skybox_triangle (render.rs:251-298).  `faces` = 3 vertex indices per face, drawn in order. */
int b32_render_skybox_mesh(b32_ctx* ctx, const B32SkyVertex* vertices, uint32_t nv, const uint32_t* faces, uint32_t nf,
                           const B32Camera* camera);
/* draw_star_diamond, render.rs:199-240, for n stars in order: centre (cx, cy) = (screen.x as i32, screen.y as i32), colour rgb[3*i..]. */
int b32_draw_star_diamonds(b32_ctx* ctx, const int32_t* cx, const int32_t* cy, const uint8_t* rgb, uint32_t n, float size);
/* The presenter's upscale (game/renderer.rs:179-214: Texture2D::from_rgba8 + FilterMode::Nearest + dest_size): destination pixel
 * (x, y) shows source texel floor((x + 0.5) * w / dst_w), floor((y + 0.5) * h / dst_h).  Writes dst_w*dst_h RGBA8 to host memory. */
int b32_present_nearest(b32_ctx* ctx, uint32_t dst_w, uint32_t dst_h, uint8_t* rgba_out);

/* ---- stage taps (parity tests only; not on the frame path) ---------------- */
/* fixed::project_fixed (fixed.rs:424-441) + float depth (render.rs:2331-2345) for n positions. */
int b32_project_fixed_batch(b32_ctx* ctx, const float* pos_xyz, uint32_t n,
                            const B32Camera* camera, uint32_t width, uint32_t height,
                            int32_t* sx, int32_t* sy, float* z);
/* Draw order of the last frame: face index of every surviving surface, in the order drawn
 * (opaque sorted, then transparent sorted; render.rs:2518-2569). `cap` entries max; returns count via n. */
int b32_last_draw_order(b32_ctx* ctx, uint32_t* face_idx, uint32_t cap, uint32_t* n);
/* IEEE-754 f32 self-test of the device arithmetic the pipeline relies on (no FMA contraction,
 * correctly rounded / and sqrt, denormals kept): evaluates op(a[i], b[i], c[i]) on the GPU.
 * op: 0 a*b+c (two roundings), 1 a/b, 2 sqrt(a), 3 (a+b)/c, 4 acos(a) as the lighting code computes it (render.rs:1049),
 * 5 / 6 the bits of `a as i32` / `a as u32` with Rust's semantics (NaN -> 0, saturating; fixed.rs:126, render.rs:1455-1458, 1618),
 * 7 Fixed32::mul_fixed (fixed.rs:161-165) on the operands' bit patterns,
 * 8 the wireframe tile kernel's three-instruction depth parameter against k / N (render.rs:784): a[i] = N, out[i] = how many
 *   k in [0, N] give different bits (the kernel uses it for N < 16384; the test runs every such N and expects zeros). */
int b32_selftest_f32(b32_ctx* ctx, int op, const float* a, const float* b, const float* c,
                     float* out, uint32_t n);

/* The numeric literals of the reference AS THE DEVICE CODE HOLDS THEM (a kernel writes them out): `count` named constants
 * (names[i] = the key in tests/golden/ref_constants.json, which tests/golden/pin_constants.py derives from the reference text;
 * bits[i] = the f32 bit pattern when is_f32[i], else the integer), the UNR_TABLE the projection kernel indexes (fixed.rs:20-31,
 * 257 bytes) and PS1_DITHER_MATRIX as dither_offset() returns it, index (y & 3) * 4 + (x & 3) (render.rs:1150-1155).
 * Any of the output pointers may be NULL; at most `cap` constants are written. */
int b32_device_constants(b32_ctx* ctx, const char** names, uint32_t* bits, uint8_t* is_f32, uint32_t cap, uint32_t* count,
                         uint8_t* unr_table257, int32_t* dither16);

/* Per-kernel device time of the last finished frame (HIP events on the ctx stream), for bench.py.
 * names[i] points at static strings; returns the number of entries written (<= cap). */
int b32_last_kernel_times(b32_ctx* ctx, const char** names, float* ms, uint32_t cap);
/* HIP-event instrumentation of the frames enqueued from now on: 0 = none (default for the async path),
 * 1 = events around the coverage kernel (the dominant one), 2 = events around every phase
 * (setup, sort, bin, cover, shade). Averages over the frames between two
 * b32_frame_finish calls (last 64 at most) are returned by b32_last_kernel_times / B32Timings. */
int b32_set_profiling(b32_ctx* ctx, int level);
/* Instrument only every `every`-th frame (default 1: each one).  An event pair around a kernel costs the stream a few microseconds
 * per frame (the kernels of consecutive frames no longer run back to back); bench.py samples every 8th frame of its timed region. */
int b32_set_profiling_stride(b32_ctx* ctx, uint32_t every);
/* Shader clock the fused fill kernel of the last finished frame really ran at (no reference counterpart; instrumentation): workgroup 0
 * reads the shader-cycle counter and the 100 MHz wall clock when it starts and when it runs out of tiles.  *ghz = 0 when that frame had
 * no such kernel; *fill_ms (nullable) = the wall-clock span the cycles were counted over.  bench.py prices VALU issue with it. */
int b32_last_shader_clock(const b32_ctx* ctx, float* ghz, float* fill_ms);
/* Test tap (no reference counterpart): *host_bound = the faces of the resident scene that their own blend mode / editor alpha or their
 * texture's blend mode can put in the transparent pass (render.rs:2403-2415), counted on the host at upload -- what decides whether a
 * frame of a moderate mesh can ever need a redraw; *device_last = the surfaces the setup kernel of the last finished frame really
 * classified as transparent.  device_last <= host_bound must hold for every frame. */
int b32_transparent_counts(const b32_ctx* ctx, uint32_t* host_bound, uint32_t* device_last);
/* Test tap (no reference counterpart): fault injection for the failure paths that cannot be provoked from outside.  what = 1: the NEXT
 * frame that hands its setup kernel over by the flag / join kernel pair loses its flag (it publishes another epoch) and its join waits 2 ms
 * instead of 2 s -- the "setup kernel never arrived" path: the fill draws nothing but the folded clear, b32_frame_finish returns
 * B32_E_HIP, the frames after it are drawn normally.  what = 2: the NEXT fused fill kernel does not publish that it has started, and the gate
 * of the pipelined frame behind it -- which orders its setup kernel after the frame set's last reader by that word -- waits 2 ms instead of
 * 2 s, gives up, goes on and raises the same error: reported once by b32_frame_finish, frames right.  Other bits: B32_E_ARG. */
int b32_debug_inject(b32_ctx* ctx, uint32_t what);
/* Two frames in flight (no reference counterpart; see B32_ROUTE_PIPELINE): when a frame is enqueued while an earlier one is still
 * pending, its setup kernel runs on a second, low-priority stream of the context beside the earlier frame's fill kernel, on a second
 * set of per-face buffers.  permille > 0 holds that setup kernel back so that it runs beside the fill's thinning second half rather than
 * beside its busy start (started together the two kernels only slow each other down), measured on the fill's tile cursor:
 *   1001 .. 2000: until (permille - 1000) / 1000 of the tiles BEHIND the workgroups' first round have been handed out (1150 was the default
 *                 through round 5);
 *   1 .. 1000   : until the last tile has been handed out and (permille - 1) / 1000 of the workgroups have found the queue empty;
 *   0 (default) : no hold on the cursor.  Since round 6 the setup kernel waits anyway -- for the ORDER of the frame sets -- until the fill in
 *                 front of it has started (Events::fill_started, k_gate), which keeps it off that fill's first microseconds; a hold on top
 *                 of that only costs (C3 0.1007-0.1012 against 0.1024-0.1053 ms per frame with 1150, C5 0.1645-0.1653 against 0.1663-0.1670).
 * Results are identical either way.  permille > 2000: B32_E_ARG. */
int b32_set_pipeline_gate(b32_ctx* ctx, uint32_t permille);
/* How far ahead of its fill a pipelined setup kernel runs (no reference counterpart; render.rs:2364-2547 is one sequential call):
 *   sets = 2: k_setup(i + 1) beside the fill of frame i -- the fill of frame i + 1 starts behind a cross-stream event that is
 *             signalled only about when the fill before it ends (10-20 us per frame with little on the GPU);
 *   sets = 3: k_setup(i + 2) beside the fill of frame i, on a third set of per-face buffers: the setup kernel a fill waits for ended
 *             a whole fill earlier and fills run back to back on the main stream.  Measured on C3 (round 4): the hole closes, but the two
 *             kernels then share every CU all the time and the frame is bound by their summed VALU work: 0.127 against 0.122 ms.
 *   sets = 0 (the library's choice, the default): two sets; three while the context draws a NARROW band (b32_set_band: at most a sixth of the
 *             frame's rows -- one rank of a frame sharded over six or more GPUs): that rank still transforms the whole mesh, its frame is
 *             bound by the setup kernel, and with three sets the setup kernels run back to back (240 rows of the 1 M-triangle frame: 0.040 ->
 *             0.034-0.035 ms per frame).  b32_set_band switches when the band crosses that width (everything in flight ends first).
 * Settles a pending frame first.  Results are identical either way.  Other values: B32_E_ARG. */
int b32_set_pipeline_depth(b32_ctx* ctx, uint32_t sets);
/* B32Timings.fragments (the reference's pixel-store count, render.rs:1671-1702) is instrumentation, not an output of
 * render_mesh_15.  on = 0 (default): not counted (B32Timings.fragments = 0 unless the textures force exact coverage); the fill
 * may then resolve opaque visibility without fetching the texel of every overdrawn fragment and without a global depth
 * sort (identical framebuffer, see b32_fill.hip).  on = 1: every fragment is evaluated and counted exactly (painter's mode). */
int b32_set_fragment_counting(b32_ctx* ctx, int on);

/* ---- multi-GPU: the exchange step of a band-sharded frame (BASELINE config C4) ------------------------------------------------------
 * No reference counterpart: the reference draws on one CPU thread; its presenter reads `fb.pixels` of ONE process
 * (game/renderer.rs:179-214), so with the frame sharded by rows (b32_set_band, one rank per GPU) every band must end up in the ROOT
 * rank's framebuffer.  Two transports:
 *
 * (1) Shared framebuffer.  The root exports its library-owned framebuffer; a band rank binds it AS ITS OWN framebuffer -- the fill
 *     kernel of a band rank only writes the rows of its band, so they land directly in the root's HBM (over xGMI between two GPUs):
 *     no copy, no gather launch.  What remains is ordering, carried by one epoch word per rank behind the pixels of the same allocation:
 *
 *         root, once:      b32_fb_new(root, w, h);  b32_band_export(root, &share);           -> hand `share` (96 bytes) to every rank
 *         rank r, once:    b32_band_import(ctx, &share, r)      (another process)   or   b32_band_attach(ctx, root, r)   (same process)
 *                          b32_set_band(ctx, y0_r, y1_r);       b32_scene_upload...(ctx, ...)
 *         rank r, frame n: [b32_band_acquire(ctx, n - 1, us)]   b32_fb_clear(ctx, ...); b32_render_scene_15_async(ctx, ...);
 *                          b32_band_publish(ctx, n)
 *         root,  frame n:  b32_set_band(root, y0_0, y1_0) once; b32_fb_clear; b32_render_scene_15_async(root, ...);
 *                          b32_band_wait(root, r, n, us) for every r;   ... present / b32_fb_download ...;   [b32_band_release(root, n)]
 *
 *     Everything is enqueued on the contexts' streams; no call blocks the host.  publish(n) takes effect behind every kernel the rank
 *     enqueued before it; wait(r, n) holds the ROOT's stream until rank r has published a frame number >= n (wrap-safe compare) or
 *     timeout_us has passed -- a timeout is counted in b32_band_status AND makes the waiting context's next b32_frame_finish return
 *     B32_E_BAND_TIMEOUT, never silent.  release / acquire are the same in the other
 *     direction (the root has consumed frame n: a rank may overwrite its rows); a host that presents every frame before the ranks
 *     start the next one (e.g. behind its own barrier) does not need them.  Frame numbers start at 1 (the words start at 0).
 *     b32_frame_finish on a band rank still reports that rank's errors and counters; triangles_drawn is the whole mesh's on every rank.
 *     A band rank must not call b32_fb_clear* / sky / present functions outside its band: those honour b32_set_band like the draw.
 *
 * (2) RCCL.  b32_gather_bands_rccl: every rank draws into its OWN framebuffer and the rows travel by ncclSend / ncclRecv (grouped, on the
 *     context's stream) to `root`.  `nccl_comm` is the caller's ncclComm_t; librccl.so is loaded on first use (B32_E_UNSUPPORTED when
 *     it is not there).  y0 / y1: the band of every rank, the same arrays on every rank.
 *
 * Status: transport (1) is exercised by real processes sharing ONE GPU (tests: test_band_ranks_share_one_gpu[*-ipc]) and by the C++
 * harness inside one process; peer mappings between two GPUs and transport (2) are compiled and argument-checked but have not run. */
typedef struct B32BandShare {
    unsigned char mem[64];      /* HIP IPC handle of the root's framebuffer allocation (the epoch words sit in its tail) */
    uint32_t width, height;     /* the root's framebuffer */
    uint32_t device;            /* the root's HIP device ordinal (informational) */
    uint32_t reserved;
    uint64_t sync_offset;       /* byte offset of the epoch words inside the allocation */
    uint64_t reserved2;
} B32BandShare;
int b32_band_export(b32_ctx* root, B32BandShare* out);                        /* root; its framebuffer must be library-owned (b32_fb_new / _resize) */
int b32_band_import(b32_ctx* ctx, const B32BandShare* share, uint32_t rank);  /* band rank 1..63 in ANOTHER process: map + bind the root's framebuffer */
int b32_band_attach(b32_ctx* ctx, b32_ctx* root, uint32_t rank);              /* the same inside one process (one process driving several contexts / GPUs) */
int b32_band_close(b32_ctx* ctx);                                             /* unmap / unbind (also done by b32_destroy) */
int b32_band_publish(b32_ctx* ctx, uint32_t frame_no);                        /* band rank: "my rows of frame_no are complete", in stream order */
int b32_band_wait(b32_ctx* root, uint32_t rank, uint32_t frame_no, uint32_t timeout_us);   /* root: hold the stream until rank published >= frame_no */
int b32_band_release(b32_ctx* root, uint32_t frame_no);                       /* root: "frame_no has been consumed", in stream order */
/* root: b32_band_wait for every rank 1 .. nranks-1 and (release_after != 0) b32_band_release(frame_no) behind them, as ONE launch */
int b32_band_wait_all(b32_ctx* root, uint32_t nranks, uint32_t frame_no, uint32_t timeout_us, int release_after);
int b32_band_acquire(b32_ctx* ctx, uint32_t frame_no, uint32_t timeout_us);   /* band rank: hold the stream until the root released >= frame_no */
/* Host-side view of the epoch words (a blocking 4-KB copy): epochs[64] (nullable) = last published frame per rank, *root_epoch
 * (nullable) = last released frame, *timeouts (nullable) = waits that gave up since the export. */
int b32_band_status(b32_ctx* ctx, uint32_t* epochs, uint32_t* root_epoch, uint32_t* timeouts);
int b32_gather_bands_rccl(b32_ctx* ctx, void* nccl_comm, int rank, int nranks, int root, const uint32_t* y0, const uint32_t* y1);
/* The communicator for transport (2) made with the very librccl the library loaded (a host that links RCCL itself may pass its own
 * ncclComm_t instead): b32_rccl_unique_id = ncclGetUniqueId on ONE rank (128 bytes, handed to the others by whatever channel the host
 * has), b32_rccl_comm_create = ncclCommInitRank on the context's device (collective: every rank calls it), _destroy = ncclCommDestroy.
 * B32_E_UNSUPPORTED when librccl.so cannot be loaded; B32_E_HIP + b32_last_hip_error = the ncclResult_t otherwise. */
int b32_rccl_unique_id(unsigned char* id128);
int b32_rccl_comm_create(b32_ctx* ctx, const unsigned char* id128, int rank, int nranks, void** nccl_comm);
int b32_rccl_comm_destroy(void* nccl_comm);
/* Test tap (no reference counterpart): b32_gather_bands_rccl in which the root additionally sends its OWN band to itself and receives it
 * at row self_dst_y0 of the same framebuffer (the rows must not overlap the band).  With a 1-rank communicator this executes the whole
 * RCCL leg -- library load, the resolved entry points, the byte datatype, stream order behind the frame's kernels -- on a single GPU. */
int b32_gather_bands_rccl_loopback(b32_ctx* ctx, void* nccl_comm, int rank, int nranks, int root, const uint32_t* y0, const uint32_t* y1,
                                   uint32_t self_dst_y0);

#ifdef __cplusplus
}
#endif
#endif /* B32RASTER_H */

"""Host-side mirror of the reference rasterizer interface, running on the MI355X through the C ABI.

Same names and argument meaning as the reference:
    Framebuffer::{new, resize, clear}       src/rasterizer/render.rs:18-45
    render_mesh_15(fb, vertices, faces, textures, camera, settings, fog) -> RasterTimings   render.rs:2302-2310
Errors the reference turns into panics come back as B32Error (index out of range, NaN sort key).

There is no CPU path here: if libb32raster.so or a HIP device is missing, construction raises.
"""
import ctypes as C

import numpy as np

from . import abi
from . import rtypes as T


class B32Error(RuntimeError):
    def __init__(self, code, where=""):
        self.code = code
        msg = abi.load_library().b32_strerror(code).decode()
        super().__init__(f"{where}: {msg} (code {code})" if where else f"{msg} (code {code})")


def _chk(rc, where=""):
    if rc != abi.B32_OK:
        raise B32Error(rc, where)


class Context:
    """One b32_ctx: one GPU, one stream, one device-resident framebuffer and scene."""

    def __init__(self, device=0):
        self.lib = abi.load_library()
        h = C.c_void_p()
        _chk(self.lib.b32_create(device, C.byref(h)), "b32_create")
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.b32_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        """Run on a caller-owned hipStream_t.  torch reports its default stream as handle 0, which the C ABI reads as "the
        context's own (non-blocking) stream" -- work enqueued there is NOT ordered with torch's default stream.  So 0 is mapped
        to hipStreamLegacy (handle 1), the legacy default stream torch actually uses."""
        HIP_STREAM_LEGACY = 1
        _chk(self.lib.b32_set_stream(self.h, stream_ptr if stream_ptr else HIP_STREAM_LEGACY), "b32_set_stream")

    def use_own_stream(self):
        _chk(self.lib.b32_set_stream(self.h, None), "b32_set_stream")

    def synchronize(self):
        _chk(self.lib.b32_synchronize(self.h), "b32_synchronize")

    def set_profiling(self, level):
        _chk(self.lib.b32_set_profiling(self.h, level), "b32_set_profiling")

    def set_profiling_stride(self, every):
        """b32_set_profiling_stride: HIP events on every `every`-th frame only (an event pair per frame costs the stream microseconds)."""
        _chk(self.lib.b32_set_profiling_stride(self.h, int(every)), "b32_set_profiling_stride")

    def set_async_depth(self, deep):
        """b32_set_async_depth: 0 = safe (default), 1 = large-scene frames back to back, a dropped one is reported by finish()."""
        _chk(self.lib.b32_set_async_depth(self.h, int(deep)), "b32_set_async_depth")

    ROUTES = ("direct_bin", "inline_bin", "counting_sort", "keyed", "redraw_region", "redraw_global_sort", "redraw_pairs", "pipelined", "lds_atlas", "wire_tiles", "span_cover",
              "flag_join", "event_join", "poll_join")

    ROUTE_SORT_FREE, ROUTE_CUT_TILES, ROUTE_INLINE_BIN, ROUTE_DIRECT_BIN, ROUTE_WIDE_GROUPS, ROUTE_PACKED_STREAMS, ROUTE_PIPELINE, ROUTE_TEX_CACHE, ROUTE_BATCH, ROUTE_LDS_ATLAS, ROUTE_WIRE_TILES, ROUTE_SPAN_COVER, ROUTE_STAGGER = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096

    # ---- a frame of several meshes (scene.rs:112-261): b32_frame_begin / _add_scene / _end
    def frame_begin(self, camera, settings):
        """One camera, base settings and light list for the frame; the meshes follow with frame_add (resident scenes in slots)."""
        cam = camera.pack()
        st, keep = settings.pack()
        self._frame_keep = (cam, st, keep)
        _chk(self.lib.b32_frame_begin(self.h, C.byref(cam), C.byref(st)), "b32_frame_begin")

    def frame_add(self, scene, ambient=None, backface_cull=None, backface_wireframe=None, fog=None):
        """Append a detached ResidentScene with its per-mesh parameters (None: the base settings' value; fog None: no fog)."""
        st = self._frame_keep[1]
        p = abi.B32MeshParams()
        p.ambient = float(st.ambient if ambient is None else ambient)
        p.backface_cull = int(st.backface_cull if backface_cull is None else bool(backface_cull))
        p.backface_wireframe = int(st.backface_wireframe if backface_wireframe is None else bool(backface_wireframe))
        fg = T.pack_fog(fog)
        p.has_fog = 0 if fg is None else 1
        if fg is not None:
            p.fog = fg
        scene.detach()
        _chk(self.lib.b32_frame_add_scene(self.h, scene._slot, C.byref(p)), "b32_frame_add_scene")

    def frame_end(self):
        _chk(self.lib.b32_frame_end(self.h), "b32_frame_end")

    def frame_submit(self, table):
        """b32_frame_submit: the frame recorded by make_frame_table, in one call."""
        cam, st, _keep, slots, params, n = table
        _chk(self.lib.b32_frame_submit(self.h, C.byref(cam), C.byref(st), slots, C.cast(params, C.c_void_p), n), "b32_frame_submit")

    @staticmethod
    def make_frame_table(camera, settings, scenes, fogs=None, ambients=None):
        """Packs camera, base settings and a list of detached ResidentScenes (+ per-mesh fog / ambient) once, for frame_submit."""
        cam = camera.pack()
        st, keep = settings.pack()
        n = len(scenes)
        slots = (C.c_void_p * n)()
        params = (abi.B32MeshParams * n)()
        for i, sc in enumerate(scenes):
            sc.detach()
            slots[i] = sc._slot
            params[i].ambient = float(st.ambient if ambients is None or ambients[i] is None else ambients[i])
            params[i].backface_cull = int(st.backface_cull); params[i].backface_wireframe = int(st.backface_wireframe)
            fg = T.pack_fog(fogs[i]) if fogs is not None else None
            params[i].has_fog = 0 if fg is None else 1
            if fg is not None:
                params[i].fog = fg
        return cam, st, keep, slots, params, n

    # ---- the presenter's copy without a host round trip per frame (b32_fb_download_async + tickets)
    def host_alloc(self, nbytes):
        """b32_host_alloc: page-locked host memory as a numpy uint8 array (freed by host_free)."""
        p = self.lib.b32_host_alloc(int(nbytes))
        if not p:
            raise MemoryError("b32_host_alloc")
        arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(int(nbytes),))
        return arr, p

    def host_free(self, p):
        self.lib.b32_host_free(p)

    def download_async(self, host_ptr):
        t = C.c_uint64()
        _chk(self.lib.b32_fb_download_async(self.h, host_ptr, C.byref(t)), "b32_fb_download_async")
        return int(t.value)

    def ticket_wait(self, ticket):
        _chk(self.lib.b32_ticket_wait(self.h, int(ticket)), "b32_ticket_wait")

    def ticket_done(self, ticket):
        d = C.c_int()
        _chk(self.lib.b32_ticket_poll(self.h, int(ticket), C.byref(d)), "b32_ticket_poll")
        return bool(d.value)

    def finish(self) -> T.RasterTimings:
        """b32_frame_finish of whatever this context has in flight."""
        tm = abi.B32Timings()
        _chk(self.lib.b32_frame_finish(self.h, C.byref(tm)), "b32_frame_finish")
        return T.RasterTimings.from_c(tm)

    def batch_counts(self):
        return {n: int(self.lib.b32_batch_count(self.h, i)) for i, n in enumerate(("merged_draws", "single_draws", "merged_built", "frames"))}

    def set_pipeline_gate(self, permille):
        """b32_set_pipeline_gate: hold a pipelined setup kernel until the previous fill's tile cursor has come that far (include/b32raster.h)."""
        _chk(self.lib.b32_set_pipeline_gate(self.h, int(permille)), "b32_set_pipeline_gate")

    def last_shader_clock(self):
        """b32_last_shader_clock: (GHz, ms) the fused kernel of the last finished frame ran at / over; (0, 0) if it had none."""
        g, m = C.c_float(), C.c_float()
        _chk(self.lib.b32_last_shader_clock(self.h, C.byref(g), C.byref(m)), "b32_last_shader_clock")
        return float(g.value), float(m.value)

    def transparent_counts(self):
        """b32_transparent_counts: (host-side bound counted at upload, surfaces the last finished frame's setup kernel classified transparent)."""
        a, b = C.c_uint32(), C.c_uint32()
        _chk(self.lib.b32_transparent_counts(self.h, C.byref(a), C.byref(b)), "b32_transparent_counts")
        return int(a.value), int(b.value)

    def set_pipeline_depth(self, sets):
        """b32_set_pipeline_depth: 2 or 3 frame sets -- the setup kernel one or two frames ahead of the fill (include/b32raster.h)."""
        _chk(self.lib.b32_set_pipeline_depth(self.h, int(sets)), "b32_set_pipeline_depth")

    def set_routes(self, off_mask):
        """b32_set_routes: switch internal routes off (ROUTE_* bits); results are identical on every route."""
        _chk(self.lib.b32_set_routes(self.h, int(off_mask)), "b32_set_routes")

    def set_cheap_threshold(self, den):
        """b32_set_cheap_threshold: CHEAP coverage while every texture has at most 1/den skippable texels (default 64)."""
        _chk(self.lib.b32_set_cheap_threshold(self.h, int(den)), "b32_set_cheap_threshold")

    def route_counts(self):
        """b32_route_count: how many frames of this context took each internal route (tests assert the targeted one ran)."""
        return {n: int(self.lib.b32_route_count(self.h, i)) for i, n in enumerate(self.ROUTES)}

    def debug_inject(self, what):
        """b32_debug_inject: fault injection (1 = the next flag / join hand-over loses its flag; 2 = the next fused kernel does not publish its start)."""
        _chk(self.lib.b32_debug_inject(self.h, int(what)), "b32_debug_inject")

    def set_fragment_counting(self, on):
        _chk(self.lib.b32_set_fragment_counting(self.h, int(on)), "b32_set_fragment_counting")

    def last_kernel_times(self):
        names = (C.c_char_p * 8)()
        ms = (C.c_float * 8)()
        n = self.lib.b32_last_kernel_times(self.h, names, ms, 8)
        return {names[i].decode(): float(ms[i]) for i in range(n)}

    # ---- multi-GPU band exchange behind the C ABI (include/b32raster.h "multi-GPU", b32_gather.hip)
    BAND_SHARE_BYTES = 96

    def band_export(self) -> bytes:
        """b32_band_export (root): the 96-byte share of this context's library-owned framebuffer + epoch words, for the other ranks."""
        buf = C.create_string_buffer(self.BAND_SHARE_BYTES)
        _chk(self.lib.b32_band_export(self.h, C.cast(buf, C.c_void_p)), "b32_band_export")
        return buf.raw

    def band_import(self, share: bytes, rank):
        """b32_band_import (band rank, another process): map the root's framebuffer and draw into it; returns (width, height)."""
        assert len(share) == self.BAND_SHARE_BYTES
        buf = C.create_string_buffer(share, self.BAND_SHARE_BYTES)
        rc = self.lib.b32_band_import(self.h, C.cast(buf, C.c_void_p), int(rank))
        _chk(rc, f"b32_band_import (hip error {self.lib.b32_last_hip_error(self.h)})" if rc else "b32_band_import")
        w, h = np.frombuffer(share, np.uint32, 2, 64)
        return int(w), int(h)

    def band_attach(self, root: "Context", rank):
        _chk(self.lib.b32_band_attach(self.h, root.h, int(rank)), "b32_band_attach")

    def band_close(self):
        _chk(self.lib.b32_band_close(self.h), "b32_band_close")

    def band_publish(self, frame_no):
        _chk(self.lib.b32_band_publish(self.h, int(frame_no)), "b32_band_publish")

    def band_wait(self, rank, frame_no, timeout_us=2_000_000):
        _chk(self.lib.b32_band_wait(self.h, int(rank), int(frame_no), int(timeout_us)), "b32_band_wait")

    def band_wait_all(self, nranks, frame_no, timeout_us=2_000_000, release_after=True):
        _chk(self.lib.b32_band_wait_all(self.h, int(nranks), int(frame_no), int(timeout_us), 1 if release_after else 0), "b32_band_wait_all")

    def band_release(self, frame_no):
        _chk(self.lib.b32_band_release(self.h, int(frame_no)), "b32_band_release")

    def band_acquire(self, frame_no, timeout_us=2_000_000):
        _chk(self.lib.b32_band_acquire(self.h, int(frame_no), int(timeout_us)), "b32_band_acquire")

    def band_status(self):
        """b32_band_status: (published frame per rank [64], released frame of the root, waits that timed out)."""
        ep = (C.c_uint32 * 64)(); root = C.c_uint32(); to = C.c_uint32()
        _chk(self.lib.b32_band_status(self.h, ep, C.byref(root), C.byref(to)), "b32_band_status")
        return list(ep), int(root.value), int(to.value)

    # transport (2): RCCL behind the C ABI.  The communicator is made by the library itself (the librccl it loaded), so the host needs
    # no RCCL binding of its own.
    @staticmethod
    def rccl_unique_id() -> bytes:
        """b32_rccl_unique_id: ncclGetUniqueId on ONE rank; hand the 128 bytes to every other rank."""
        from . import abi
        buf = C.create_string_buffer(128)
        _chk(abi.load_library().b32_rccl_unique_id(C.cast(buf, C.c_void_p)), "b32_rccl_unique_id")
        return buf.raw

    def rccl_comm_create(self, unique_id: bytes, rank, nranks):
        """b32_rccl_comm_create: ncclCommInitRank on this context's device (collective); returns the opaque ncclComm_t."""
        assert len(unique_id) == 128
        buf = C.create_string_buffer(unique_id, 128)
        comm = C.c_void_p()
        rc = self.lib.b32_rccl_comm_create(self.h, C.cast(buf, C.c_void_p), int(rank), int(nranks), C.byref(comm))
        _chk(rc, f"b32_rccl_comm_create (ncclResult {self.lib.b32_last_hip_error(self.h)})" if rc else "b32_rccl_comm_create")
        return comm

    def rccl_comm_destroy(self, comm):
        _chk(self.lib.b32_rccl_comm_destroy(comm), "b32_rccl_comm_destroy")

    def gather_bands_rccl(self, comm, rank, nranks, root, bands, loopback_dst_y0=None):
        """b32_gather_bands_rccl: bands = [(y0, y1)] of every rank; enqueued on the context's stream behind the frame's kernels.
        loopback_dst_y0 (test tap): the root also sends its own band to itself, received at that row."""
        y0 = (C.c_uint32 * nranks)(*[b[0] for b in bands]); y1 = (C.c_uint32 * nranks)(*[b[1] for b in bands])
        if loopback_dst_y0 is None:
            rc = self.lib.b32_gather_bands_rccl(self.h, comm, int(rank), int(nranks), int(root), y0, y1)
        else:
            rc = self.lib.b32_gather_bands_rccl_loopback(self.h, comm, int(rank), int(nranks), int(root), y0, y1, int(loopback_dst_y0))
        _chk(rc, f"b32_gather_bands_rccl (ncclResult {self.lib.b32_last_hip_error(self.h)})" if rc == -4 else "b32_gather_bands_rccl")

    # ---- stage taps -----------------------------------------------------------------
    def project_fixed_batch(self, pos, camera: T.Camera, width, height):
        pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 3)
        n = len(pos)
        sx = np.zeros(n, np.int32); sy = np.zeros(n, np.int32); z = np.zeros(n, np.float32)
        cam = camera.pack()
        _chk(self.lib.b32_project_fixed_batch(self.h, pos.ctypes.data, n, C.byref(cam), width, height,
                                              sx.ctypes.data, sy.ctypes.data, z.ctypes.data), "project_fixed_batch")
        return sx, sy, z

    def device_constants(self):
        """({fixture key: value}, UNR table, 4x4 dither matrix [y & 3][x & 3]) read back from device code (b32_device_constants)."""
        cap = 256
        names = (C.c_char_p * cap)()
        bits = np.zeros(cap, np.uint32); isf = np.zeros(cap, np.uint8)
        n = C.c_uint32()
        unr = np.zeros(257, np.uint8); dither = np.zeros(16, np.int32)
        _chk(self.lib.b32_device_constants(self.h, names, bits.ctypes.data, isf.ctypes.data, cap, C.byref(n), unr.ctypes.data,
                                           dither.ctypes.data), "b32_device_constants")
        out = {}
        for i in range(min(n.value, cap)):
            out[names[i].decode()] = float(bits[i:i + 1].view(np.float32)[0]) if isf[i] else int(bits[i])
        return out, unr, dither.reshape(4, 4)

    def selftest_f32(self, op, a, b, c):
        a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
        c = np.ascontiguousarray(c, np.float32)
        out = np.zeros_like(a)
        _chk(self.lib.b32_selftest_f32(self.h, op, a.ctypes.data, b.ctypes.data, c.ctypes.data, out.ctypes.data, a.size))
        return out

    def last_draw_order(self, cap):
        buf = np.zeros(max(cap, 1), np.uint32)
        n = C.c_uint32()
        _chk(self.lib.b32_last_draw_order(self.h, buf.ctypes.data, cap, C.byref(n)), "last_draw_order")
        return buf[:min(n.value, cap)].copy()


class Framebuffer:
    """Framebuffer (render.rs:10-45), device resident. `pixels` downloads the RGBA8 bytes."""

    def __init__(self, width, height, ctx: Context = None, device=0):
        self.ctx = ctx or Context(device)
        # Framebuffer::new (render.rs:18-25): zero pixels and an f32::MAX z-buffer even when the ctx already held a frame of this size
        _chk(self.ctx.lib.b32_fb_new(self.ctx.h, width, height), "fb_new")
        self.width, self.height = width, height
        self.set_band(0, height)          # a new Framebuffer owns all its rows (a band set earlier on this ctx does not carry over)

    @staticmethod
    def new(width, height, ctx=None):
        return Framebuffer(width, height, ctx)

    def resize(self, width, height):
        _chk(self.ctx.lib.b32_fb_resize(self.ctx.h, width, height), "fb_resize")
        self.width, self.height = width, height

    def bind_device(self, device_ptr, width, height):
        """Draw into caller-owned device memory (e.g. torch uint8 tensor .data_ptr())."""
        _chk(self.ctx.lib.b32_fb_bind_device(self.ctx.h, device_ptr, width, height), "fb_bind_device")
        self.width, self.height = width, height

    def set_band(self, y0, y1):
        _chk(self.ctx.lib.b32_set_band(self.ctx.h, y0, y1), "set_band")

    def clear(self, color: T.Color):
        _chk(self.ctx.lib.b32_fb_clear(self.ctx.h, color.r, color.g, color.b, color.blend), "fb_clear")

    def clear_gradient(self, top: T.Color, bottom: T.Color):
        """Framebuffer::clear_gradient (render.rs:58-77)"""
        _chk(self.ctx.lib.b32_fb_clear_gradient(self.ctx.h, top.r, top.g, top.b, top.blend, bottom.r, bottom.g, bottom.b, bottom.blend), "fb_clear_gradient")

    def clear_transparent(self):
        """Framebuffer::clear_transparent (render.rs:47-56)"""
        _chk(self.ctx.lib.b32_fb_clear_transparent(self.ctx.h), "fb_clear_transparent")

    def render_skybox_mesh(self, vertices, faces, camera: T.Camera):
        """Step 1 of Framebuffer::render_skybox (render.rs:81-134): `vertices` (abi.SKY_VERTEX_DTYPE) and `faces` ([n,3] u32) are
        what Skybox::generate_mesh returned on the host."""
        v = np.ascontiguousarray(vertices, dtype=abi.SKY_VERTEX_DTYPE)
        f = np.ascontiguousarray(faces, dtype=np.uint32).reshape(-1, 3)
        cam = camera.pack()
        _chk(self.ctx.lib.b32_render_skybox_mesh(self.ctx.h, v.ctypes.data if len(v) else None, len(v),
                                                 f.ctypes.data if len(f) else None, len(f), C.byref(cam)), "render_skybox_mesh")

    def draw_star_diamonds(self, cx, cy, rgb, size):
        """draw_star_diamond (render.rs:199-240) for every star, in order."""
        cx = np.ascontiguousarray(cx, np.int32); cy = np.ascontiguousarray(cy, np.int32)
        rgb = np.ascontiguousarray(rgb, np.uint8).reshape(-1, 3)
        _chk(self.ctx.lib.b32_draw_star_diamonds(self.ctx.h, cx.ctypes.data, cy.ctypes.data, rgb.ctypes.data, len(cx), float(size)), "draw_star_diamonds")

    def present_nearest(self, dst_w, dst_h):
        """The presenter's nearest-neighbour upscale (game/renderer.rs:179-214) -> uint8 [dst_h, dst_w, 4]."""
        out = np.empty((dst_h, dst_w, 4), np.uint8)
        _chk(self.ctx.lib.b32_present_nearest(self.ctx.h, dst_w, dst_h, out.ctypes.data), "present_nearest")
        return out

    def upload(self, pixels):
        px = np.ascontiguousarray(pixels, dtype=np.uint8).reshape(-1)
        assert px.size == self.width * self.height * 4
        _chk(self.ctx.lib.b32_fb_upload(self.ctx.h, px.ctypes.data), "fb_upload")

    @property
    def pixels(self):
        out = np.empty(self.width * self.height * 4, np.uint8)
        _chk(self.ctx.lib.b32_fb_download(self.ctx.h, out.ctypes.data), "fb_download")
        return out

    @property
    def zbuffer(self):
        """Framebuffer::zbuffer (render.rs:12): f32 per pixel, f32::MAX where nothing was drawn in z-buffer mode."""
        out = np.empty(self.width * self.height, np.float32)
        _chk(self.ctx.lib.b32_zbuffer_download(self.ctx.h, out.ctypes.data), "zbuffer_download")
        return out

    def image(self):
        return self.pixels.reshape(self.height, self.width, 4)


def _geom(vertices, faces):
    v = np.ascontiguousarray(vertices, dtype=abi.VERTEX_DTYPE)
    f = np.ascontiguousarray(faces, dtype=abi.FACE_DTYPE)
    return v, f


def render_mesh_15(fb: Framebuffer, vertices, faces, textures, camera: T.Camera, settings: T.RasterSettings,
                   fog=None) -> T.RasterTimings:
    """render_mesh_15 (render.rs:2302-2310): host slices in, draws into the device framebuffer."""
    v, f = _geom(vertices, faces)
    tex_arr, _keep = T.pack_textures(textures)
    cam = camera.pack()
    st, _kl = settings.pack()
    fg = T.pack_fog(fog)
    tm = abi.B32Timings()
    rc = fb.ctx.lib.b32_render_mesh_15(fb.ctx.h, v.ctypes.data if len(v) else None, len(v),
                                       f.ctypes.data if len(f) else None, len(f),
                                       C.cast(tex_arr, C.c_void_p), len(textures), C.byref(cam), C.byref(st),
                                       C.byref(fg) if fg is not None else None, C.byref(tm))
    _chk(rc, "render_mesh_15")
    return T.RasterTimings.from_c(tm)


def render_mesh(fb: Framebuffer, vertices, faces, textures, camera: T.Camera, settings: T.RasterSettings) -> T.RasterTimings:
    """render_mesh (render.rs:1971-1978), the 8-bit-colour path every caller takes when `settings.use_rgb555` is false
    (scene.rs:163-169).  `textures` are rtypes.Texture (Color texels with per-texel blend modes)."""
    v, f = _geom(vertices, faces)
    tex_arr, _keep = T.pack_textures8(textures)
    cam = camera.pack()
    st, _kl = settings.pack()
    tm = abi.B32Timings()
    rc = fb.ctx.lib.b32_render_mesh(fb.ctx.h, v.ctypes.data if len(v) else None, len(v),
                                    f.ctypes.data if len(f) else None, len(f),
                                    C.cast(tex_arr, C.c_void_p), len(textures), C.byref(cam), C.byref(st), C.byref(tm))
    _chk(rc, "render_mesh")
    return T.RasterTimings.from_c(tm)


# aliases named in BASELINE.json's north_star (the reference's real entry point is render_mesh_15)
draw_mesh = render_mesh_15


class ResidentScene:
    """A mesh + textures kept in HBM across frames: upload once, draw many times."""

    def __init__(self, fb: Framebuffer, vertices, faces, textures=None, indexed_textures=None, textures8=None):
        self.fb = fb
        self.ctx = fb.ctx
        v, f = _geom(vertices, faces)
        self.fmt8 = textures8 is not None
        if textures8 is not None:                       # the 8-bit-colour path (render_mesh)
            arr, keep = T.pack_textures8(textures8)
            rc = self.ctx.lib.b32_scene_upload_rgba(self.ctx.h, v.ctypes.data if len(v) else None, len(v),
                                                    f.ctypes.data if len(f) else None, len(f),
                                                    C.cast(arr, C.c_void_p), len(textures8))
        elif indexed_textures is not None:
            arr, keep = T.pack_indexed_textures(indexed_textures)
            rc = self.ctx.lib.b32_scene_upload_indexed(self.ctx.h, v.ctypes.data if len(v) else None, len(v),
                                                       f.ctypes.data if len(f) else None, len(f),
                                                       C.cast(arr, C.c_void_p), len(indexed_textures))
        else:
            textures = textures or []
            arr, keep = T.pack_textures(textures)
            rc = self.ctx.lib.b32_scene_upload(self.ctx.h, v.ctypes.data if len(v) else None, len(v),
                                               f.ctypes.data if len(f) else None, len(f),
                                               C.cast(arr, C.c_void_p), len(textures))
        _chk(rc, "scene_upload")
        self.n_faces = len(f)
        self._packed = None
        self._slot = None

    def detach(self):
        """Move this scene into its own slot (b32_scene_swap) so that other scenes can be uploaded and drawn through the same context:
        every later render*/render_async swaps it in, enqueues, and swaps it out again -- no upload, no host sync."""
        if self._slot is None:
            h = C.c_void_p()
            _chk(self.ctx.lib.b32_scene_create(self.ctx.h, C.byref(h)), "scene_create")
            self._slot = h
            _chk(self.ctx.lib.b32_scene_swap(self.ctx.h, self._slot), "scene_swap")
        return self

    def _swap(self):
        if self._slot is not None:
            _chk(self.ctx.lib.b32_scene_swap(self.ctx.h, self._slot), "scene_swap")

    def close(self):
        if self._slot is not None:
            self.ctx.lib.b32_scene_destroy(self.ctx.h, self._slot)
            self._slot = None

    def _pack(self, camera, settings, fog):
        cam = camera.pack()
        st, kl = settings.pack()
        fg = T.pack_fog(fog)
        self._packed = (cam, st, kl, fg)
        return self._packed

    def render(self, camera, settings, fog=None) -> T.RasterTimings:
        cam, st, _kl, fg = self._pack(camera, settings, fog)
        tm = abi.B32Timings()
        self._swap()
        try:
            if self.fmt8:
                _chk(self.ctx.lib.b32_render_scene(self.ctx.h, C.byref(cam), C.byref(st), C.byref(tm)), "render_scene")
            else:
                _chk(self.ctx.lib.b32_render_scene_15(self.ctx.h, C.byref(cam), C.byref(st),
                                                      C.byref(fg) if fg is not None else None, C.byref(tm)), "render_scene_15")
        finally:
            self._swap()
        return T.RasterTimings.from_c(tm)

    def render_async(self, camera=None, settings=None, fog=None):
        """Enqueue only. With no arguments, re-enqueues the last packed camera/settings (no Python packing cost)."""
        if camera is not None:
            self._pack(camera, settings, fog)
        cam, st, _kl, fg = self._packed
        self._swap()
        try:
            if self.fmt8:
                _chk(self.ctx.lib.b32_render_scene_async(self.ctx.h, C.byref(cam), C.byref(st)), "render_scene_async")
            else:
                _chk(self.ctx.lib.b32_render_scene_15_async(self.ctx.h, C.byref(cam), C.byref(st),
                                                            C.byref(fg) if fg is not None else None), "render_scene_15_async")
        finally:
            self._swap()

    def finish(self) -> T.RasterTimings:
        tm = abi.B32Timings()
        _chk(self.ctx.lib.b32_frame_finish(self.ctx.h, C.byref(tm)), "frame_finish")
        return T.RasterTimings.from_c(tm)

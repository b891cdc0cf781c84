"""ctypes wrapper of the CPU oracle (oracle/b32_oracle.c).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

import bonnie32_amd as b32
from bonnie32_amd import abi, rtypes as T

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libb32oracle.so")
FAST_PATH = os.path.join(_HERE, "_build", "libb32oracle_fast.so")     # bench.py's CPU baseline: the reference's release profile
_lib = None
_fast = None


class B32OracleDump(C.Structure):
    _fields_ = [("sx", C.c_void_p), ("sy", C.c_void_p), ("sz", C.c_void_p), ("draw_order", C.c_void_p),
                ("n_drawn", C.c_uint32), ("n_opaque", C.c_uint32)]


def build(force=False):
    src = os.path.join(_HERE, "b32_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "b32raster.h")
    mk = os.path.join(_HERE, "Makefile")
    newest = max(os.path.getmtime(src), os.path.getmtime(hdr), os.path.getmtime(mk))
    if force or any(not os.path.exists(q) or os.path.getmtime(q) < newest for q in (LIB_PATH, FAST_PATH)):
        subprocess.run(["make", "-C", _HERE] + (["-B"] if force else []), check=True, capture_output=True)
    return LIB_PATH


def _load(path):
    if True:
        L = C.CDLL(path)
        P = C.c_void_p
        L.b32o_unr_table.restype = C.c_uint8; L.b32o_unr_table.argtypes = [C.c_uint32]
        L.b32o_fixed_from_f32.restype = C.c_int32; L.b32o_fixed_from_f32.argtypes = [C.c_float]
        L.b32o_fixed_mul.restype = C.c_int32; L.b32o_fixed_mul.argtypes = [C.c_int32, C.c_int32]
        L.b32o_fixed_div_unr.restype = C.c_int32; L.b32o_fixed_div_unr.argtypes = [C.c_int32, C.c_int32]
        L.b32o_fixed_to_f32.restype = C.c_float; L.b32o_fixed_to_f32.argtypes = [C.c_int32]
        L.b32o_project_fixed.restype = None
        L.b32o_project_fixed.argtypes = [P, P, P, P, P, C.c_uint32, C.c_uint32, P, P, P]
        L.b32o_color15_to_rgba.restype = None; L.b32o_color15_to_rgba.argtypes = [C.c_uint16, P]
        L.b32o_texture15_sample.restype = C.c_uint16
        L.b32o_texture15_sample.argtypes = [P, C.c_uint32, C.c_uint32, C.c_float, C.c_float]
        L.b32o_expand_indexed.restype = None; L.b32o_expand_indexed.argtypes = [P, C.c_uint32, P, C.c_uint32, P]
        L.b32o_blend_rgb555.restype = None
        L.b32o_blend_rgb555.argtypes = [C.c_uint8] * 6 + [C.c_uint32, P]
        L.b32o_dither_offset.restype = C.c_int32; L.b32o_dither_offset.argtypes = [C.c_uint32, C.c_uint32]
        L.b32o_dither_and_quantize.restype = None
        L.b32o_dither_and_quantize.argtypes = [C.c_uint8] * 3 + [C.c_uint32, C.c_uint32, P]
        L.b32o_unique_edges_selfcheck.restype = C.c_uint32; L.b32o_unique_edges_selfcheck.argtypes = [P, C.c_uint32]
        L.b32o_fb_clear.restype = None
        L.b32o_fb_clear.argtypes = [P, P, C.c_uint32, C.c_uint32] + [C.c_uint8] * 4
        L.b32o_render_mesh_15.restype = C.c_int
        L.b32o_render_mesh_15.argtypes = [P, P, C.c_uint32, C.c_uint32, P, C.c_uint32, P, C.c_uint32, P, C.c_uint32,
                                          P, P, P, P, P]
        L.b32o_render_mesh.restype = C.c_int
        L.b32o_render_mesh.argtypes = [P, P, C.c_uint32, C.c_uint32, P, C.c_uint32, P, C.c_uint32, P, C.c_uint32, P, P, P, P]
        L.b32o_fb_clear_gradient.restype = None; L.b32o_fb_clear_gradient.argtypes = [P, P, C.c_uint32, C.c_uint32, P, P]
        L.b32o_render_skybox_mesh.restype = C.c_int
        L.b32o_render_skybox_mesh.argtypes = [P, C.c_uint32, C.c_uint32, P, C.c_uint32, P, C.c_uint32, P]
        L.b32o_draw_star_diamond.restype = None
        L.b32o_draw_star_diamond.argtypes = [P, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_float, P]
        L.b32o_acosf.restype = C.c_float; L.b32o_acosf.argtypes = [C.c_float]
        L.b32o_set_row_band.restype = None; L.b32o_set_row_band.argtypes = [C.c_uint32, C.c_uint32]
        L.b32o_vec3_dot.restype = C.c_float; L.b32o_vec3_dot.argtypes = [P, P]
        L.b32o_vec3_cross.restype = None; L.b32o_vec3_cross.argtypes = [P, P, P]
        L.b32o_constant.restype = C.c_double; L.b32o_constant.argtypes = [C.c_char_p]
        L.b32o_constant_count.restype = C.c_uint32; L.b32o_constant_count.argtypes = []
        L.b32o_constant_name.restype = C.c_char_p; L.b32o_constant_name.argtypes = [C.c_uint32]
        L.b32o_set_threads.restype = None; L.b32o_set_threads.argtypes = [C.c_int]
    return L


def lib():
    """the CHECKER build (strict IEEE flags): what every parity test compares against"""
    global _lib
    if _lib is None:
        build()
        _lib = _load(LIB_PATH)
    return _lib


def fast_lib():
    """the CPU-BASELINE build (the reference's release profile: -O3 -flto, x86-64 baseline); frames are asserted byte-identical"""
    global _fast
    if _fast is None:
        build()
        _fast = _load(FAST_PATH)
    return _fast


def _f3(v):
    return (C.c_float * 3)(*[float(np.float32(x)) for x in v])


def project_fixed(world_pos, camera: T.Camera, width, height):
    """fixed::project_fixed (fixed.rs:424-441) -> (sx, sy, depth_f32)."""
    sx, sy, d = C.c_int32(), C.c_int32(), C.c_float()
    lib().b32o_project_fixed(_f3(world_pos), _f3(camera.position), _f3(camera.basis_x), _f3(camera.basis_y),
                             _f3(camera.basis_z), width, height, C.byref(sx), C.byref(sy), C.byref(d))
    return sx.value, sy.value, d.value


def constants():
    """{fixture key: value} of every named literal the C oracle computes with (b32o_constant)."""
    L = lib()
    names = [L.b32o_constant_name(i).decode() for i in range(L.b32o_constant_count())]
    return {n: L.b32o_constant(n.encode()) for n in names}


def unique_edges_selfcheck(tri_xyz):
    """0 when the hash-set edge de-duplication of the wireframe phases equals the reference's quadratic `any()` scan (render.rs:2589-2594)
    on these triangles (n x 9 floats: three screen vertices x, y, z), entry for entry; else 1 + the first differing index."""
    t = np.ascontiguousarray(tri_xyz, np.float32).reshape(-1, 9)
    return int(lib().b32o_unique_edges_selfcheck(t.ctypes.data, t.shape[0]))


def expand_indexed(indices, clut):
    """IndexedAtlas::to_texture15 texels (mesh_editor.rs:669-682) through Clut::lookup (types.rs:390-397): OOB index -> 0x0000."""
    idx = np.ascontiguousarray(indices, np.uint8).reshape(-1)
    cl = np.ascontiguousarray(clut, np.uint16).reshape(-1)
    out = np.zeros(idx.size, np.uint16)
    clp = cl if cl.size else np.zeros(1, np.uint16)
    lib().b32o_expand_indexed(idx.ctypes.data, idx.size, clp.ctypes.data, cl.size, out.ctypes.data)
    return out


class Framebuffer:
    """Framebuffer (render.rs:10-45) on the host, for the oracle."""

    def __init__(self, width, height):
        self.width, self.height = width, height
        self.pixels = np.zeros(width * height * 4, np.uint8)
        self.zbuffer = np.full(width * height, np.finfo(np.float32).max, np.float32)

    def clear(self, color: T.Color):
        lib().b32o_fb_clear(self.pixels.ctypes.data, self.zbuffer.ctypes.data, self.width, self.height,
                            color.r, color.g, color.b, color.blend)

    def image(self):
        return self.pixels.reshape(self.height, self.width, 4)

    def clear_gradient(self, top: T.Color, bottom: T.Color):
        t = (C.c_uint8 * 4)(top.r, top.g, top.b, top.blend); b = (C.c_uint8 * 4)(bottom.r, bottom.g, bottom.b, bottom.blend)
        lib().b32o_fb_clear_gradient(self.pixels.ctypes.data, self.zbuffer.ctypes.data, self.width, self.height, t, b)

    def render_skybox_mesh(self, vertices, faces, camera: T.Camera):
        v = np.ascontiguousarray(vertices, dtype=abi.SKY_VERTEX_DTYPE)
        f = np.ascontiguousarray(faces, dtype=np.uint32).reshape(-1, 3)
        cam = camera.pack()
        return lib().b32o_render_skybox_mesh(self.pixels.ctypes.data, self.width, self.height, v.ctypes.data, len(v), f.ctypes.data, len(f), C.byref(cam))

    def draw_star_diamonds(self, cx, cy, rgb, size):
        rgb = np.ascontiguousarray(rgb, np.uint8).reshape(-1, 3)
        for i in range(len(cx)):
            lib().b32o_draw_star_diamond(self.pixels.ctypes.data, self.width, self.height, int(cx[i]), int(cy[i]), float(size), rgb[i].ctypes.data)


def render_mesh(fb: Framebuffer, vertices, faces, textures, camera: T.Camera, settings: T.RasterSettings, dump=False):
    """render_mesh (render.rs:1971-2264), the 8-bit-colour path, on the CPU. `textures` are rtypes.Texture."""
    return render_mesh_15(fb, vertices, faces, textures, camera, settings, None, dump, _fmt8=True)


def render_mesh_15(fb: Framebuffer, vertices, faces, textures, camera: T.Camera, settings: T.RasterSettings,
                   fog=None, dump=False, _fmt8=False, fast=False, threads=1):
    """render_mesh_15 (render.rs:2302-2638) on the CPU. Returns (rc, RasterTimings[, dump dict]).
    fast: the baseline build (release profile) instead of the checker; threads > 1: the all-cores schedule (b32o_set_threads)."""
    L = fast_lib() if fast else lib()
    L.b32o_set_threads(int(threads))
    vertices = np.ascontiguousarray(vertices, dtype=abi.VERTEX_DTYPE)
    faces = np.ascontiguousarray(faces, dtype=abi.FACE_DTYPE)
    tex_arr, _keep_t = T.pack_textures8(textures) if _fmt8 else T.pack_textures(textures)
    cam = camera.pack()
    st, _keep_l = settings.pack()
    fg = T.pack_fog(fog)
    tm = abi.B32Timings()
    d = None
    if dump:
        d = B32OracleDump()
        sx = np.zeros(max(len(vertices), 1), np.int32); sy = np.zeros_like(sx)
        sz = np.zeros(max(len(vertices), 1), np.float32)
        order = np.zeros(max(len(faces), 1), np.uint32)
        d.sx, d.sy, d.sz, d.draw_order = sx.ctypes.data, sy.ctypes.data, sz.ctypes.data, order.ctypes.data
    common = (fb.pixels.ctypes.data, fb.zbuffer.ctypes.data, fb.width, fb.height,
              vertices.ctypes.data if len(vertices) else None, len(vertices),
              faces.ctypes.data if len(faces) else None, len(faces), C.cast(tex_arr, C.c_void_p), len(textures), C.byref(cam), C.byref(st))
    tail = (C.byref(tm), C.byref(d) if d is not None else None)
    try:
        if _fmt8:
            rc = L.b32o_render_mesh(*common, *tail)
        else:
            rc = L.b32o_render_mesh_15(*common, C.byref(fg) if fg is not None else None, *tail)
    finally:
        L.b32o_set_threads(1)
    t = T.RasterTimings.from_c(tm)
    if dump:
        return rc, t, {"sx": sx[:len(vertices)], "sy": sy[:len(vertices)], "sz": sz[:len(vertices)],
                       "draw_order": order[:d.n_drawn].copy(), "n_opaque": int(d.n_opaque)}
    return rc, t


# ---------------------------------------------------------------- all-cores CPU baseline (bench.py)
_pool_scene = None


def _band_worker(args):
    """One process of the all-cores baseline: draws the rows [y0, y1) of the shared scene into a private framebuffer."""
    y0, y1, reps = args
    sc = _pool_scene
    L = lib()
    fb = Framebuffer(sc.width, sc.height)
    L.b32o_set_row_band(y0, y1)
    import time
    t = 0.0
    for _ in range(reps):
        fb.clear(sc.clear_color)
        c0 = time.perf_counter()
        rc, tm = render_mesh_15(fb, sc.vertices, sc.faces, sc.textures, sc.camera, sc.settings, sc.fog)
        t += time.perf_counter() - c0
    L.b32o_set_row_band(0, 0xFFFFFFFF)
    row = sc.width * 4
    return y0, y1, t / reps, rc, fb.pixels[y0 * row:y1 * row].copy()


def render_all_cores(sc, n_procs, reps=1):
    """The same frame drawn by n_procs processes, one row band each (transform, cull and sort replicated in every process -- the
    reference rasterizer is single-threaded, this is the row-band-parallel "fair CPU" variant of SURVEY 8d).  Returns
    (seconds per frame = the slowest band, assembled RGBA bytes).  Processes are forked: the scene is inherited, not pickled."""
    import multiprocessing as mp
    global _pool_scene
    _pool_scene = sc
    lib()
    bands = []
    base, extra = divmod(sc.height, n_procs)
    y = 0
    for r in range(n_procs):
        h = base + (1 if r < extra else 0)
        bands.append((y, y + h, reps)); y += h
    with mp.get_context("fork").Pool(n_procs) as pool:
        res = pool.map(_band_worker, bands, chunksize=1)
    frame = np.zeros(sc.width * sc.height * 4, np.uint8)
    row = sc.width * 4
    worst = 0.0
    for y0, y1, t, rc, px in res:
        assert rc == 0
        frame[y0 * row:y1 * row] = px
        worst = max(worst, t)
    return worst, frame

"""ctypes mirror of include/b32raster.h (the C-ABI drop-in boundary for `render_mesh_15`,
reference: src/rasterizer/render.rs:2302-2310) and the loader of the HIP library.

The loader fails loudly when libb32raster.so is missing: there is no CPU fallback on the product path.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libb32raster.so")

# ---- error codes -----------------------------------------------------------
B32_OK, B32_E_ARG, B32_E_INDEX, B32_E_NAN_KEY, B32_E_HIP, B32_E_UNSUPPORTED, B32_E_NO_DEVICE, B32_E_FRAME_DROPPED, B32_E_BAND_TIMEOUT = 0, -1, -2, -3, -4, -5, -6, -7, -8
NO_TEXTURE = 0xFFFFFFFF

# BlendMode (types.rs:1380-1388)
OPAQUE, AVERAGE, ADD, SUBTRACT, ADD_QUARTER, ERASE = range(6)
# ShadingMode (types.rs:1289-1294)
SHADE_NONE, SHADE_FLAT, SHADE_GOURAUD = range(3)
# LightType (types.rs:1297-1304)
LIGHT_DIRECTIONAL, LIGHT_POINT, LIGHT_SPOT = range(3)

# ---- numpy dtypes with the exact C layout (used for bulk vertex/face arrays) ----
VERTEX_DTYPE = np.dtype([("pos", "<f4", 3), ("uv", "<f4", 2), ("normal", "<f4", 3),
                         ("r", "u1"), ("g", "u1"), ("b", "u1"), ("blend", "u1")])
FACE_DTYPE = np.dtype([("v", "<u4", 3), ("texture_id", "<u4"), ("black_transparent", "u1"),
                       ("blend_mode", "u1"), ("editor_alpha", "u1"), ("_pad", "u1")])
assert VERTEX_DTYPE.itemsize == 36 and FACE_DTYPE.itemsize == 20


class B32Texture15(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("blend_mode", C.c_uint32), ("_pad", C.c_uint32),
                ("pixels", C.c_void_p)]


class B32Texture(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("blend_mode", C.c_uint32), ("_pad", C.c_uint32), ("pixels", C.c_void_p)]


SKY_VERTEX_DTYPE = np.dtype([("pos", np.float32, 3), ("r", np.uint8), ("g", np.uint8), ("b", np.uint8), ("blend", np.uint8)])


class B32IndexedTexture(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("blend_mode", C.c_uint32), ("clut_len", C.c_uint32),
                ("indices", C.c_void_p), ("clut", C.c_void_p)]


class B32Camera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("basis_x", C.c_float * 3), ("basis_y", C.c_float * 3),
                ("basis_z", C.c_float * 3)]


class B32Light(C.Structure):
    _fields_ = [("type", C.c_uint32), ("position", C.c_float * 3), ("direction", C.c_float * 3),
                ("radius", C.c_float), ("angle", C.c_float), ("intensity", C.c_float),
                ("r", C.c_uint8), ("g", C.c_uint8), ("b", C.c_uint8), ("enabled", C.c_uint8)]


class B32Settings(C.Structure):
    _fields_ = [("affine_textures", C.c_uint8), ("use_zbuffer", C.c_uint8), ("shading", C.c_uint8),
                ("backface_cull", C.c_uint8), ("backface_wireframe", C.c_uint8), ("dithering", C.c_uint8),
                ("wireframe_overlay", C.c_uint8), ("use_rgb555", C.c_uint8), ("use_fixed_point", C.c_uint8),
                ("xray_mode", C.c_uint8), ("has_ortho", C.c_uint8), ("_pad", C.c_uint8),
                ("ambient", C.c_float), ("ortho_zoom", C.c_float), ("ortho_center_x", C.c_float),
                ("ortho_center_y", C.c_float), ("n_lights", C.c_uint32), ("lights", C.c_void_p)]


class B32Fog(C.Structure):
    _fields_ = [("start", C.c_float), ("falloff", C.c_float), ("cull_distance", C.c_float),
                ("r", C.c_uint8), ("g", C.c_uint8), ("b", C.c_uint8), ("blend", C.c_uint8)]


class B32MeshParams(C.Structure):
    _fields_ = [("ambient", C.c_float), ("backface_cull", C.c_uint8), ("backface_wireframe", C.c_uint8), ("has_fog", C.c_uint8),
                ("_pad", C.c_uint8), ("fog", B32Fog)]


class B32Timings(C.Structure):
    _fields_ = [("transform_ms", C.c_float), ("fog_ms", C.c_float), ("cull_ms", C.c_float), ("sort_ms", C.c_float),
                ("draw_ms", C.c_float), ("wireframe_ms", C.c_float), ("triangles_drawn", C.c_uint32),
                ("tile_pairs", C.c_uint32), ("fragments", C.c_uint64)]


# Every symbol include/b32raster.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("b32_create", C.c_int, [C.c_int, C.POINTER(_P)]),
    ("b32_destroy", None, [_P]),
    ("b32_strerror", C.c_char_p, [C.c_int]),
    ("b32_last_hip_error", C.c_int, [_P]),
    ("b32_set_stream", C.c_int, [_P, _P]),
    ("b32_synchronize", C.c_int, [_P]),
    ("b32_fb_resize", C.c_int, [_P, C.c_uint32, C.c_uint32]),
    ("b32_fb_new", C.c_int, [_P, C.c_uint32, C.c_uint32]),
    ("b32_set_async_depth", C.c_int, [_P, C.c_int]),
    ("b32_route_count", C.c_ulonglong, [_P, C.c_int]),
    ("b32_set_routes", C.c_int, [_P, C.c_uint32]),
    ("b32_set_cheap_threshold", C.c_int, [_P, C.c_uint32]),
    ("b32_fb_clear", C.c_int, [_P, C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint8]),
    ("b32_fb_upload", C.c_int, [_P, _P]),
    ("b32_fb_download", C.c_int, [_P, _P]),
    ("b32_host_alloc", C.c_void_p, [C.c_size_t]),
    ("b32_host_free", None, [C.c_void_p]),
    ("b32_fb_download_async", C.c_int, [_P, _P, C.POINTER(C.c_uint64)]),
    ("b32_ticket_poll", C.c_int, [_P, C.c_uint64, C.POINTER(C.c_int)]),
    ("b32_ticket_wait", C.c_int, [_P, C.c_uint64]),
    ("b32_zbuffer_download", C.c_int, [_P, _P]),
    ("b32_zbuffer_upload", C.c_int, [_P, _P]),
    ("b32_fb_bind_device", C.c_int, [_P, _P, C.c_uint32, C.c_uint32]),
    ("b32_fb_size", C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("b32_set_band", C.c_int, [_P, C.c_uint32, C.c_uint32]),
    ("b32_render_mesh_15", C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint32, _P, C.c_uint32, _P, _P, _P, _P]),
    ("b32_scene_upload", C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint32, _P, C.c_uint32]),
    ("b32_scene_upload_indexed", C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint32, _P, C.c_uint32]),
    ("b32_render_scene_15", C.c_int, [_P, _P, _P, _P, _P]),
    ("b32_render_scene_15_async", C.c_int, [_P, _P, _P, _P]),
    ("b32_frame_finish", C.c_int, [_P, _P]),
    ("b32_scene_create", C.c_int, [_P, C.POINTER(_P)]),
    ("b32_scene_destroy", None, [_P, _P]),
    ("b32_scene_swap", C.c_int, [_P, _P]),
    ("b32_frame_begin", C.c_int, [_P, _P, _P]),
    ("b32_frame_add_scene", C.c_int, [_P, _P, _P]),
    ("b32_frame_end", C.c_int, [_P]),
    ("b32_frame_submit", C.c_int, [_P, _P, _P, C.POINTER(_P), _P, C.c_uint32]),
    ("b32_batch_count", C.c_ulonglong, [_P, C.c_int]),
    ("b32_fb_clear_gradient", C.c_int, [_P] + [C.c_uint8] * 8),
    ("b32_fb_clear_transparent", C.c_int, [_P]),
    ("b32_render_skybox_mesh", C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint32, _P]),
    ("b32_draw_star_diamonds", C.c_int, [_P, _P, _P, _P, C.c_uint32, C.c_float]),
    ("b32_present_nearest", C.c_int, [_P, C.c_uint32, C.c_uint32, _P]),
    ("b32_render_mesh", C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint32, _P, C.c_uint32, _P, _P, _P]),
    ("b32_scene_upload_rgba", C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint32, _P, C.c_uint32]),
    ("b32_render_scene", C.c_int, [_P, _P, _P, _P]),
    ("b32_render_scene_async", C.c_int, [_P, _P, _P]),
    ("b32_project_fixed_batch", C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint32, C.c_uint32, _P, _P, _P]),
    ("b32_last_draw_order", C.c_int, [_P, _P, C.c_uint32, C.POINTER(C.c_uint32)]),
    ("b32_selftest_f32", C.c_int, [_P, C.c_int, _P, _P, _P, _P, C.c_uint32]),
    ("b32_last_kernel_times", C.c_int, [_P, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_uint32]),
    ("b32_device_constants", C.c_int, [_P, C.POINTER(C.c_char_p), _P, _P, C.c_uint32, C.POINTER(C.c_uint32), _P, _P]),
    ("b32_set_profiling", C.c_int, [_P, C.c_int]),
    ("b32_set_profiling_stride", C.c_int, [_P, C.c_uint32]),
    ("b32_set_pipeline_gate", C.c_int, [_P, C.c_uint32]),
    ("b32_set_pipeline_depth", C.c_int, [_P, C.c_uint32]),
    ("b32_last_shader_clock", C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    ("b32_transparent_counts", C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("b32_set_fragment_counting", C.c_int, [_P, C.c_int]),
    ("b32_debug_inject", C.c_int, [_P, C.c_uint32]),
    ("b32_build_digest", C.c_char_p, []),
    ("b32_band_export", C.c_int, [_P, C.c_void_p]),
    ("b32_band_import", C.c_int, [_P, C.c_void_p, C.c_uint32]),
    ("b32_band_attach", C.c_int, [_P, _P, C.c_uint32]),
    ("b32_band_close", C.c_int, [_P]),
    ("b32_band_publish", C.c_int, [_P, C.c_uint32]),
    ("b32_band_wait", C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint32]),
    ("b32_band_release", C.c_int, [_P, C.c_uint32]),
    ("b32_band_wait_all", C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]),
    ("b32_band_acquire", C.c_int, [_P, C.c_uint32, C.c_uint32]),
    ("b32_band_status", C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("b32_gather_bands_rccl", C.c_int, [_P, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("b32_gather_bands_rccl_loopback", C.c_int, [_P, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint32]),
    ("b32_rccl_unique_id", C.c_int, [C.c_void_p]),
    ("b32_rccl_comm_create", C.c_int, [_P, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    ("b32_rccl_comm_destroy", C.c_int, [C.c_void_p]),
]

_lib = None


def load_library(path=None):
    """dlopen libb32raster.so and type every entry point. Raises if the library was not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("B32_LIB") or LIB_PATH          # B32_LIB: an experiment build (tools/exp_variants.py); never set by the product
    if not os.path.exists(p):
        raise RuntimeError(
            f"{p} is missing: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()'). "
            "The bonnie-32 rasterizer path has no CPU fallback.")
    lib = C.CDLL(p)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    if path is None:
        _lib = lib
    return lib


def check_build_digest(lib=None):
    """The loaded library must have been compiled from THIS source tree (b32_build_digest() == build.csrc_digest()): a stale .so that
    travelled with the snapshot would otherwise be timed / tested in place of the sources next to it.  Returns the digest.  An
    experiment build named by B32_LIB (tools/exp_variants.py) is exempt: it is never the product."""
    from . import build as B
    lib = lib or load_library()
    have = (lib.b32_build_digest() or b"").decode("ascii", "replace")
    want = B.csrc_digest()
    if have != want and not os.environ.get("B32_LIB"):
        raise RuntimeError(f"libb32raster.so was built from other sources (library {have}, tree {want}): run __graft_entry__.build()")
    return have


def ptr(a):
    """Address of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data

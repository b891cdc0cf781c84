"""`.b32scene` -- one call of render_mesh_15 / render_mesh as a file: everything the reference's signature takes
(render.rs:2302-2310 / :1971-1978: vertices, faces, textures, camera, settings with lights, fog) plus the framebuffer it draws into
(size, clear colour) and, optionally, what the call must produce (frame / depth-buffer SHA-256, triangles_drawn, fragment stores).
The same bytes feed the CPU oracle, the GPU library (tests/cpp/mesh_harness.cpp through the C++ host mirror) and the Rust harness
that pins the oracle against the reference itself (tests/rust/pin_oracle).  Little-endian, no padding between sections:

  header  64 B   magic "B32SCENE", u32 version (1), u32 flags (1 = 8-bit-colour path: Texture texels of 4 B, call render_mesh;
                 2 = fog is Some; 4 = ortho_projection is Some; 8 = an expectation record follows the textures),
                 u32 width, height, nv, nf, nt, n_lights, u8 clear r, g, b, blend, 20 B reserved (zero)
  camera  48 B   f32 position[3], basis_x[3], basis_y[3], basis_z[3]                                         (camera.rs:9-18)
  settings 28 B  u8 affine_textures, use_zbuffer, shading (0 None, 1 Flat, 2 Gouraud), backface_cull, backface_wireframe, dithering,
                 wireframe_overlay, use_rgb555, use_fixed_point, xray_mode, 2 B zero; f32 ambient, ortho zoom, center_x, center_y
                                                                                                             (types.rs:1392-1428)
  fog     16 B   f32 start, falloff, cull_distance, u8 r, g, b, blend (zero when flags & 2 is clear)         (render.rs:2309)
  lights  n_lights x 44 B   u32 type (0 Directional, 1 Point, 2 Spot), f32 position[3], direction[3], radius, angle, intensity,
                 u8 r, g, b, enabled                                                                         (types.rs:1297-1314)
  vertices nv x 36 B   f32 pos[3], uv[2], normal[3], u8 r, g, b, blend                                       (types.rs:947-959)
  faces   nf x 20 B    u32 v0, v1, v2, texture_id (0xFFFFFFFF = None), u8 black_transparent, blend_mode, editor_alpha, 0
                                                                                                             (types.rs:984-1002)
  textures nt x { u32 width, height, blend_mode, texel_bytes (2: Color15 u16 / 4: Color r, g, b, blend); width*height texels }
  expectation 80 B (flags & 8)   u32 triangles_drawn, u32 zero, u64 fragments, 32 B SHA-256 of fb.pixels, 32 B SHA-256 of fb.zbuffer (f32 LE)
"""
import hashlib
import struct

import numpy as np

from . import abi, rtypes as T

MAGIC = b"B32SCENE"
VERSION = 1
F_FMT8, F_FOG, F_ORTHO, F_EXPECT = 1, 2, 4, 8


def write_scene(path, sc, expect=None, fmt8=None):
    """sc: a scenegen.Scene (or anything with its fields; `textures8` instead of `textures` on the 8-bit path).
    expect: None or dict(triangles_drawn, fragments, sha256 (hex), zbuffer_sha256 (hex))."""
    st = sc.settings
    if fmt8 is None:
        fmt8 = not st.use_rgb555
    texs = sc.textures8 if fmt8 else sc.textures
    fog = getattr(sc, "fog", None)
    flags = (F_FMT8 if fmt8 else 0) | (F_FOG if fog is not None else 0) | (F_ORTHO if st.ortho_projection is not None else 0) | (F_EXPECT if expect else 0)
    cc = sc.clear_color
    v = np.ascontiguousarray(sc.vertices, dtype=abi.VERTEX_DTYPE)
    f = np.ascontiguousarray(sc.faces, dtype=abi.FACE_DTYPE)
    out = [MAGIC, struct.pack("<8I", VERSION, flags, sc.width, sc.height, len(v), len(f), len(texs), len(st.lights)),
           bytes([cc.r, cc.g, cc.b, cc.blend]), bytes(20)]
    cam = sc.camera
    out.append(np.asarray(list(cam.position) + list(cam.basis_x) + list(cam.basis_y) + list(cam.basis_z), dtype="<f4").tobytes())
    ortho = st.ortho_projection if st.ortho_projection is not None else (0.0, 0.0, 0.0)
    out.append(bytes([int(st.affine_textures), int(st.use_zbuffer), int(st.shading), int(st.backface_cull), int(st.backface_wireframe),
                      int(st.dithering), int(st.wireframe_overlay), int(st.use_rgb555), int(st.use_fixed_point), int(st.xray_mode), 0, 0]))
    out.append(np.asarray([st.ambient, *ortho], dtype="<f4").tobytes())
    if fog is not None:
        start, falloff, cull, col = fog
        out.append(np.asarray([start, falloff, cull], dtype="<f4").tobytes() + bytes([col.r, col.g, col.b, col.blend]))
    else:
        out.append(bytes(16))
    for l in st.lights:
        out.append(struct.pack("<I", l.light_type) + np.asarray(list(l.position) + list(l.direction) + [l.radius, l.angle, l.intensity], dtype="<f4").tobytes()
                   + bytes([l.color.r, l.color.g, l.color.b, int(l.enabled)]))
    out.append(v.tobytes()); out.append(f.tobytes())
    for t in texs:
        px = np.ascontiguousarray(t.pixels, dtype=np.uint8 if fmt8 else "<u2")
        out.append(struct.pack("<4I", t.width, t.height, t.blend_mode, 4 if fmt8 else 2) + px.tobytes())
    if expect:
        out.append(struct.pack("<IIQ", int(expect["triangles_drawn"]), 0, int(expect["fragments"])) + bytes.fromhex(expect["sha256"]) + bytes.fromhex(expect["zbuffer_sha256"]))
    blob = b"".join(out)
    with open(path, "wb") as fh:
        fh.write(blob)
    return hashlib.sha256(blob).hexdigest()


class LoadedScene:
    """What read_scene returns: the fields of scenegen.Scene (+ textures8 / fmt8 / expect)."""

    @property
    def n_tris(self):
        return len(self.faces)


def read_scene(path):
    b = open(path, "rb").read()
    if b[:8] != MAGIC:
        raise ValueError("not a .b32scene file")
    ver, flags, w, h, nv, nf, nt, nl = struct.unpack_from("<8I", b, 8)
    if ver != VERSION:
        raise ValueError(f".b32scene version {ver}")
    o = 40
    sc = LoadedScene()
    sc.name = path; sc.width, sc.height = w, h
    sc.clear_color = T.Color(b[o], b[o + 1], b[o + 2], b[o + 3]); o = 64
    cam = np.frombuffer(b, "<f4", 12, o); o += 48
    sc.camera = T.Camera(tuple(map(float, cam[0:3])), tuple(map(float, cam[3:6])), tuple(map(float, cam[6:9])), tuple(map(float, cam[9:12])))
    s8 = b[o:o + 12]; o += 12
    amb, oz, ocx, ocy = (float(x) for x in np.frombuffer(b, "<f4", 4, o)); o += 16
    fog = None
    if flags & F_FOG:
        fs = np.frombuffer(b, "<f4", 3, o)
        fog = (float(fs[0]), float(fs[1]), float(fs[2]), T.Color(b[o + 12], b[o + 13], b[o + 14], b[o + 15]))
    o += 16
    lights = []
    for _ in range(nl):
        ty = struct.unpack_from("<I", b, o)[0]
        fl = np.frombuffer(b, "<f4", 9, o + 4)
        lights.append(T.Light(ty, tuple(map(float, fl[0:3])), tuple(map(float, fl[3:6])), float(fl[6]), float(fl[7]),
                              T.Color(b[o + 40], b[o + 41], b[o + 42]), float(fl[8]), bool(b[o + 43])))
        o += 44
    sc.settings = T.RasterSettings(affine_textures=bool(s8[0]), use_zbuffer=bool(s8[1]), shading=int(s8[2]), backface_cull=bool(s8[3]),
                                   backface_wireframe=bool(s8[4]), lights=lights, ambient=amb, dithering=bool(s8[5]), wireframe_overlay=bool(s8[6]),
                                   ortho_projection=(oz, ocx, ocy) if flags & F_ORTHO else None, use_rgb555=bool(s8[7]), use_fixed_point=bool(s8[8]),
                                   xray_mode=bool(s8[9]))
    sc.fog = fog
    sc.vertices = np.frombuffer(b, abi.VERTEX_DTYPE, nv, o).copy(); o += nv * abi.VERTEX_DTYPE.itemsize
    sc.faces = np.frombuffer(b, abi.FACE_DTYPE, nf, o).copy(); o += nf * abi.FACE_DTYPE.itemsize
    sc.fmt8 = bool(flags & F_FMT8)
    texs = []
    for _ in range(nt):
        tw, th, bl, tb = struct.unpack_from("<4I", b, o); o += 16
        if tb != (4 if sc.fmt8 else 2):
            raise ValueError("texel size does not match the pixel format of the file")
        if sc.fmt8:
            texs.append(T.Texture(tw, th, np.frombuffer(b, np.uint8, tw * th * 4, o).copy(), bl)); o += tw * th * 4
        else:
            texs.append(T.Texture15(tw, th, np.frombuffer(b, "<u2", tw * th, o).copy(), bl)); o += tw * th * 2
    sc.textures = [] if sc.fmt8 else texs
    sc.textures8 = texs if sc.fmt8 else []
    sc.indexed_textures = []
    sc.expect = None
    if flags & F_EXPECT:
        td, _z, fr = struct.unpack_from("<IIQ", b, o)
        sc.expect = {"triangles_drawn": td, "fragments": fr, "sha256": b[o + 16:o + 48].hex(), "zbuffer_sha256": b[o + 48:o + 80].hex()}
        o += 80
    if o != len(b):
        raise ValueError(f".b32scene: {len(b) - o} trailing bytes")
    return sc

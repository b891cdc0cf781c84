"""Host-side mirrors of the reference rasterizer's API types (src/rasterizer/types.rs, camera.rs), same
names and field meanings, plus the packers that flatten them into the C PODs of include/b32raster.h.

Pure data: importing this module needs neither the HIP library nor a GPU.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

from . import abi
from .abi import (OPAQUE, ERASE, SHADE_GOURAUD, SHADE_NONE, LIGHT_DIRECTIONAL, LIGHT_POINT, LIGHT_SPOT,
                  VERTEX_DTYPE, FACE_DTYPE, NO_TEXTURE)


@dataclass
class Color:
    """types.rs:721-726 Color{r,g,b,blend}."""
    r: int = 0
    g: int = 0
    b: int = 0
    blend: int = OPAQUE


@dataclass
class Texture15:
    """types.rs:532-539: width*height Color15 (u16), row-major, plus the STP blend mode."""
    width: int
    height: int
    pixels: np.ndarray  # uint16 [height*width]
    blend_mode: int = OPAQUE
    name: str = ""

    def __post_init__(self):
        self.pixels = np.ascontiguousarray(self.pixels, dtype=np.uint16).reshape(-1)
        assert self.pixels.size == self.width * self.height

    @staticmethod
    def checkerboard(width, height, color1, color2):
        """types.rs:702-711"""
        y, x = np.mgrid[0:height, 0:width]
        px = np.where(((x // 4) + (y // 4)) % 2 == 0, color1, color2).astype(np.uint16)
        return Texture15(width, height, px, OPAQUE, "checkerboard")


@dataclass
class Texture:
    """types.rs:1166-1176 (8-bit-colour path): width*height Color values = (r, g, b, blend) bytes, row-major."""
    width: int
    height: int
    pixels: np.ndarray  # uint8 [height*width, 4]
    blend_mode: int = OPAQUE
    name: str = ""

    def __post_init__(self):
        self.pixels = np.ascontiguousarray(self.pixels, dtype=np.uint8).reshape(-1, 4)
        assert self.pixels.shape[0] == self.width * self.height

    @staticmethod
    def checkerboard(width, height, color1, color2):
        """types.rs:1231-1240; colours are (r, g, b, blend) tuples."""
        y, x = np.mgrid[0:height, 0:width]
        m = (((x // 4) + (y // 4)) % 2 == 0)[..., None]
        px = np.where(m, np.array(color1, np.uint8), np.array(color2, np.uint8)).astype(np.uint8)
        return Texture(width, height, px, OPAQUE, "checkerboard")

    @staticmethod
    def from_texture15(t, stp_blend=OPAQUE):
        """Test helper (not a reference function): RGB555 texels widened to Color; 0x0000 -> Erase (what Texture::from_bytes
        makes of alpha 0, types.rs:1198-1223), STP-bit texels get `stp_blend` as their per-texel blend mode."""
        c = t.pixels.astype(np.uint32)
        def e(v5):
            return ((v5 << 3) | (v5 >> 2)) & 0xFF
        px = np.stack([e((c >> 10) & 31), e((c >> 5) & 31), e(c & 31), np.where(c == 0, ERASE, np.where(c & 0x8000, stp_blend, OPAQUE))], axis=1)
        return Texture(t.width, t.height, px.astype(np.uint8), t.blend_mode, t.name)


@dataclass
class IndexedTexture:
    """IndexedAtlas + Clut (modeler/mesh_editor.rs:594-606, types.rs:340-397): one byte per texel."""
    width: int
    height: int
    indices: np.ndarray  # uint8
    clut: np.ndarray     # uint16, 16 or 256 entries
    blend_mode: int = OPAQUE

    def __post_init__(self):
        self.indices = np.ascontiguousarray(self.indices, dtype=np.uint8).reshape(-1)
        self.clut = np.ascontiguousarray(self.clut, dtype=np.uint16).reshape(-1)

    def to_texture15(self, name="asset_part"):
        """IndexedAtlas::to_texture15 (mesh_editor.rs:669-682) with Clut::lookup (types.rs:390-397)."""
        idx = self.indices.astype(np.int64)
        ok = idx < self.clut.size
        lut = self.clut if self.clut.size else np.zeros(1, np.uint16)          # (an empty palette: every lookup is out of range)
        px = np.where(ok, lut[np.minimum(idx, lut.size - 1)], 0).astype(np.uint16)
        return Texture15(self.width, self.height, px, self.blend_mode, name)


@dataclass
class Camera:
    """camera.rs:9-18. The basis vectors are inputs of the path (sin/cos stay with the caller)."""
    position: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    basis_x: Tuple[float, float, float] = (1.0, 0.0, 0.0)
    basis_y: Tuple[float, float, float] = (0.0, 1.0, 0.0)
    basis_z: Tuple[float, float, float] = (0.0, 0.0, 1.0)

    def pack(self):
        c = abi.B32Camera()
        for name in ("position", "basis_x", "basis_y", "basis_z"):
            v = np.asarray(getattr(self, name), dtype=np.float32)
            getattr(c, name)[:] = [float(x) for x in v]
        return c


def _normalize_f32(v):
    """Vec3::normalize (math.rs:39-49) in f32 op order: sqrt((x*x + y*y) + z*z), then x/l."""
    v = np.asarray(v, dtype=np.float32)
    l = np.sqrt(np.float32(np.float32(v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]))
    if l == 0:
        return np.zeros(3, np.float32)
    return (v / l).astype(np.float32)


@dataclass
class Light:
    """types.rs:1306-1314"""
    light_type: int = LIGHT_DIRECTIONAL
    position: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    direction: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    radius: float = 0.0
    angle: float = 0.0
    color: Color = field(default_factory=lambda: Color(255, 255, 255))
    intensity: float = 1.0
    enabled: bool = True

    @staticmethod
    def directional(direction, intensity):
        """types.rs:1318-1326: the direction is normalized at construction."""
        return Light(LIGHT_DIRECTIONAL, direction=tuple(_normalize_f32(direction)), intensity=intensity)

    @staticmethod
    def point(position, radius, intensity):
        """types.rs:1329-1337"""
        return Light(LIGHT_POINT, position=tuple(position), radius=radius, intensity=intensity)

    @staticmethod
    def spot(position, direction, angle, radius, intensity):
        """types.rs:1355-1369: the direction is normalized at construction."""
        return Light(LIGHT_SPOT, position=tuple(position), direction=tuple(_normalize_f32(direction)), angle=angle, radius=radius,
                     intensity=intensity)


@dataclass
class RasterSettings:
    """types.rs:1392-1428; defaults are the reference's (types.rs:1475-1495)."""
    affine_textures: bool = True
    use_zbuffer: bool = True
    shading: int = SHADE_GOURAUD
    backface_cull: bool = True
    backface_wireframe: bool = True
    lights: List[Light] = field(default_factory=lambda: [Light.directional((-1.0, -1.0, -1.0), 0.7)])
    ambient: float = 0.3
    dithering: bool = True
    wireframe_overlay: bool = False
    ortho_projection: Optional[Tuple[float, float, float]] = None  # (zoom, center_x, center_y)
    use_rgb555: bool = True
    use_fixed_point: bool = True
    xray_mode: bool = False

    @staticmethod
    def game():
        """types.rs:1455-1460"""
        return RasterSettings(backface_wireframe=False)

    @staticmethod
    def benchmark():
        """SURVEY §8 benchmark configuration: affine + snap + RGB555 dither, painter's, no lights."""
        return RasterSettings(use_zbuffer=False, shading=SHADE_NONE, backface_wireframe=False, lights=[])

    def pack(self):
        """-> (B32Settings, keepalive). The lights array must outlive the call."""
        s = abi.B32Settings()
        s.affine_textures = int(self.affine_textures)
        s.use_zbuffer = int(self.use_zbuffer)
        s.shading = int(self.shading)
        s.backface_cull = int(self.backface_cull)
        s.backface_wireframe = int(self.backface_wireframe)
        s.dithering = int(self.dithering)
        s.wireframe_overlay = int(self.wireframe_overlay)
        s.use_rgb555 = int(self.use_rgb555)
        s.use_fixed_point = int(self.use_fixed_point)
        s.xray_mode = int(self.xray_mode)
        s.has_ortho = int(self.ortho_projection is not None)
        s.ambient = float(self.ambient)
        if self.ortho_projection is not None:
            s.ortho_zoom, s.ortho_center_x, s.ortho_center_y = [float(x) for x in self.ortho_projection]
        n = len(self.lights)
        arr = (abi.B32Light * max(n, 1))()
        for i, l in enumerate(self.lights):
            arr[i].type = l.light_type
            arr[i].position[:] = [float(np.float32(x)) for x in l.position]
            arr[i].direction[:] = [float(np.float32(x)) for x in l.direction]
            arr[i].radius = float(l.radius)
            arr[i].angle = float(l.angle)
            arr[i].intensity = float(l.intensity)
            arr[i].r, arr[i].g, arr[i].b = l.color.r, l.color.g, l.color.b
            arr[i].enabled = int(l.enabled)
        s.n_lights = n
        s.lights = C.cast(arr, C.c_void_p).value if n else None
        return s, arr


@dataclass
class RasterTimings:
    """types.rs:1499-1514 (+ the exact fragment-store count behind Mpixels/s)."""
    transform_ms: float = 0.0
    fog_ms: float = 0.0
    cull_ms: float = 0.0
    sort_ms: float = 0.0
    draw_ms: float = 0.0
    wireframe_ms: float = 0.0
    triangles_drawn: int = 0
    fragments: int = 0
    tile_pairs: int = 0

    @staticmethod
    def from_c(t):
        return RasterTimings(t.transform_ms, t.fog_ms, t.cull_ms, t.sort_ms, t.draw_ms, t.wireframe_ms,
                             int(t.triangles_drawn), int(t.fragments), int(t.tile_pairs))


def make_vertices(n):
    """Zeroed B32Vertex array with the reference defaults: color NEUTRAL (128,128,128) (types.rs:768)."""
    v = np.zeros(n, dtype=VERTEX_DTYPE)
    v["r"] = v["g"] = v["b"] = 128
    return v


def make_faces(n, texture_id=None):
    """Zeroed B32Face array with Face::new / Face::with_texture defaults (types.rs:1012-1036)."""
    f = np.zeros(n, dtype=FACE_DTYPE)
    f["texture_id"] = NO_TEXTURE if texture_id is None else texture_id
    f["black_transparent"] = 1
    f["blend_mode"] = OPAQUE
    f["editor_alpha"] = 255
    return f


def pack_fog(fog):
    """fog: None or (start, falloff, cull_distance, Color) as in render.rs:2309."""
    if fog is None:
        return None
    start, falloff, cull, col = fog
    f = abi.B32Fog()
    f.start, f.falloff, f.cull_distance = float(start), float(falloff), float(cull)
    f.r, f.g, f.b, f.blend = col.r, col.g, col.b, col.blend
    return f


def pack_textures(textures):
    """[Texture15] -> (B32Texture15 array, keepalive)."""
    n = len(textures)
    arr = (abi.B32Texture15 * max(n, 1))()
    for i, t in enumerate(textures):
        arr[i].width, arr[i].height, arr[i].blend_mode = t.width, t.height, t.blend_mode
        arr[i].pixels = t.pixels.ctypes.data if t.pixels.size else None
    return arr, list(textures)


def pack_textures8(textures):
    """[Texture] -> (B32Texture array, keepalive)."""
    n = len(textures)
    arr = (abi.B32Texture * max(n, 1))()
    for i, t in enumerate(textures):
        arr[i].width, arr[i].height, arr[i].blend_mode = t.width, t.height, t.blend_mode
        arr[i].pixels = t.pixels.ctypes.data if t.pixels.size else None
    return arr, list(textures)


def pack_indexed_textures(textures):
    n = len(textures)
    arr = (abi.B32IndexedTexture * max(n, 1))()
    for i, t in enumerate(textures):
        arr[i].width, arr[i].height, arr[i].blend_mode, arr[i].clut_len = t.width, t.height, t.blend_mode, t.clut.size
        arr[i].indices = t.indices.ctypes.data
        arr[i].clut = t.clut.ctypes.data
    return arr, list(textures)

"""Deterministic synthetic scenes for the rasterizer hot path.

PRNG = splitmix64 used as a counter-based generator: draw k of stream `seed` is mix(seed + (k+1)*GAMMA);
uniform floats are (x >> 40) * 2**-24 (exact in f32).  All geometry arithmetic is explicit numpy float32,
so the same seed gives bit-identical scenes on every box with this image.

Configs (BASELINE.json `configs`):
  C1  320x240,     2 000 tris, mean bbox  ~64 px, 64x64 4-bit atlas      (plumbing + golden fixture)
  C2  320x240,   100 000 tris, mean bbox  ~16 px, 64x64 4-bit atlas
  C3  2560x1920, 1 000 000 tris, mean bbox ~48 px, 256x256 8-bit atlas   (headline, HBM-roofline run)
  C4  = C3 sharded over 8 GPUs (screen bands)
  C5  2560x1920, 1 000 000 tris, mean bbox ~400 px, 64 discrete depths   (painter's-sort / overdraw stress)
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import abi
from .rtypes import (Camera, Color, IndexedTexture, Light, RasterSettings, Texture15, make_faces, make_vertices)

GAMMA = np.uint64(0x9E3779B97F4A7C15)
BASE_SEED = 0xB0771E32

CONFIGS = {
    "C1": dict(width=320, height=240, n_tris=2_000, bbox_px=64.0, atlas=64, clut=16, depths=0, config_id=1),
    "C2": dict(width=320, height=240, n_tris=100_000, bbox_px=16.0, atlas=64, clut=16, depths=0, config_id=2),
    "C3": dict(width=2560, height=1920, n_tris=1_000_000, bbox_px=48.0, atlas=256, clut=256, depths=0, config_id=3),
    "C5": dict(width=2560, height=1920, n_tris=1_000_000, bbox_px=400.0, atlas=256, clut=256, depths=64, config_id=5),
}
CONFIGS["C4"] = dict(CONFIGS["C3"])  # same scene, band-sharded over GPUs


def splitmix64(seed, n, offset=0):
    """n draws (uint64) of stream `seed`, starting at draw index `offset`."""
    with np.errstate(over="ignore"):
        k = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + k * GAMMA
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed, n, offset=0):
    """n floats in [0,1): (x >> 40) * 2^-24, exact in f32."""
    return ((splitmix64(seed, n, offset) >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


@dataclass
class Scene:
    name: str
    width: int
    height: int
    vertices: np.ndarray
    faces: np.ndarray
    textures: List[Texture15]
    indexed_textures: List[IndexedTexture]
    camera: Camera
    settings: RasterSettings
    clear_color: Color = field(default_factory=lambda: Color(20, 22, 28))  # game/renderer.rs:95
    fog: Optional[tuple] = None

    @property
    def n_tris(self):
        return len(self.faces)


def make_atlas(seed, size, clut_len, stp=False):
    """size x size index atlas (one byte/texel, uniform indices) + CLUT: entry 0 = 0x0000 (transparent),
    others uniform 15-bit with bit 15 clear (or set on every 4th entry when stp=True)."""
    idx = (splitmix64(seed ^ 0x7E87, size * size) % np.uint64(clut_len)).astype(np.uint8)
    clut = (splitmix64(seed ^ 0xC107, clut_len) & np.uint64(0x7FFF)).astype(np.uint16)
    clut[0] = 0
    if stp:
        clut[3::4] |= np.uint16(0x8000)
    return IndexedTexture(size, size, idx, clut, abi.OPAQUE)


def make_scene(config="C1", n_tris=None, seed=None, variant="bench", width=None, height=None, bbox_px=None):
    """Build one synthetic scene.

    variant: "bench"  -> SURVEY §8 benchmark settings (painter's, shading None)
             "gouraud"-> reference default light (directional (-1,-1,-1)*0.7, ambient 0.3), Gouraud
             "blend"  -> 10% of faces blend_mode=Average on an STP-bit CLUT + a second texture whose
                         blend_mode is Add (exercises the transparent partition and blend_rgb555)
             "float"  -> use_fixed_point=False (float projection, math.rs:117-136)
             "blend5" -> every BlendMode of blend_rgb555 (render.rs:1093-1145) both ways it can reach a pixel store: five extra
                         textures whose blend_mode is Average / Add / Subtract / AddQuarter / Erase over STP-bit CLUTs
                         (render.rs:1450-1452: the texture's mode wins when one is bound), and untextured faces carrying the
                         same five modes as the FACE blend mode with vertex colours dark enough that part of their pixels
                         quantise to black, get bit 15 (render.rs:1659-1661) and are blended
    """
    cfg = dict(CONFIGS[config])
    if n_tris is not None:
        cfg["n_tris"] = int(n_tris)
    if width is not None:
        cfg["width"], cfg["height"] = int(width), int(height)
    if bbox_px is not None:
        cfg["bbox_px"] = float(bbox_px)
    W, H, N = cfg["width"], cfg["height"], cfg["n_tris"]
    seed = (BASE_SEED + cfg["config_id"]) if seed is None else seed
    f32 = np.float32

    U = uniform01(seed, N * 24).reshape(N, 24)
    vs = f32((f32(min(W, H)) / f32(2.0)) * f32(0.75))
    px = U[:, 0] * f32(W)
    py = U[:, 1] * f32(H)
    if cfg["depths"]:
        level = np.floor(U[:, 2] * f32(cfg["depths"])).astype(np.float32)
        cz = f32(400.0) + level * f32(5600.0 / cfg["depths"])
    else:
        cz = f32(400.0) + U[:, 2] * f32(5600.0)
    k = (cz + f32(5.0)) / f32(4.0)                      # world units per projected unit at this depth
    cx = (px - f32(W / 2)) / vs * k
    cy = (py - f32(H / 2)) / vs * k
    r_world = f32(np.sqrt(cfg["bbox_px"])) / vs * k     # 3 uniform offsets in [-r,r] => E[bbox side] = r

    verts = make_vertices(3 * N)
    off = (U[:, 3:12] * f32(2.0) - f32(1.0)).reshape(N, 3, 3) * r_world[:, None, None]
    centre = np.stack([cx, cy, cz], axis=1)[:, None, :]
    verts["pos"] = (centre + off).astype(np.float32).reshape(3 * N, 3)
    verts["uv"] = (U[:, 12:18] * f32(3.0) - f32(1.0)).reshape(3 * N, 2)       # [-1, 2): exercises rem_euclid
    verts["normal"] = np.array([0.0, 0.0, -1.0], np.float32)
    col = (splitmix64(seed ^ 0xC0105, 9 * N) >> np.uint64(56)).astype(np.uint8).reshape(3 * N, 3)
    verts["r"], verts["g"], verts["b"] = col[:, 0], col[:, 1], col[:, 2]
    verts["blend"] = abi.OPAQUE

    faces = make_faces(N, texture_id=0)
    faces["v"] = np.arange(3 * N, dtype=np.uint32).reshape(N, 3)

    atlas = make_atlas(seed, cfg["atlas"], cfg["clut"], stp=(variant == "blend"))
    indexed = [atlas]
    settings = RasterSettings.benchmark()
    if variant == "gouraud":
        settings.shading = abi.SHADE_GOURAUD
        settings.lights = [Light.directional((-1.0, -1.0, -1.0), 0.7)]
        settings.ambient = 0.3
        nrm = (uniform01(seed ^ 0x4047, 9 * N).reshape(3 * N, 3) * f32(2.0) - f32(1.0)).astype(np.float32)
        verts["normal"] = nrm
    elif variant == "blend":
        sel = uniform01(seed ^ 0xB1E4D, N)
        faces["blend_mode"] = np.where(sel < 0.10, abi.AVERAGE, abi.OPAQUE).astype(np.uint8)
        t2 = make_atlas(seed ^ 0x2222, cfg["atlas"], cfg["clut"], stp=True)
        t2.blend_mode = abi.ADD
        indexed.append(t2)
        faces["texture_id"] = np.where((sel >= 0.10) & (sel < 0.15), 1, 0).astype(np.uint32)
        faces["black_transparent"] = np.where(sel > 0.9, 0, 1).astype(np.uint8)
        faces["editor_alpha"] = np.where((sel >= 0.15) & (sel < 0.18), 128, 255).astype(np.uint8)
    elif variant == "blend5":
        sel = uniform01(seed ^ 0xB1E4D, N)
        modes = [abi.AVERAGE, abi.ADD, abi.SUBTRACT, abi.ADD_QUARTER, abi.ERASE]
        tid = np.zeros(N, np.uint32)
        fbm = np.full(N, abi.OPAQUE, np.uint8)
        for k, m in enumerate(modes):
            t = make_atlas(seed ^ (0x3333 + k), cfg["atlas"], cfg["clut"], stp=True)
            t.blend_mode = m
            indexed.append(t)
            tid = np.where((sel >= 0.40 + 0.06 * k) & (sel < 0.46 + 0.06 * k), k + 1, tid).astype(np.uint32)
            face_m = (sel >= 0.70 + 0.05 * k) & (sel < 0.75 + 0.05 * k)          # untextured, face blend mode m
            tid = np.where(face_m, abi.NO_TEXTURE, tid).astype(np.uint32)
            fbm = np.where(face_m, m, fbm).astype(np.uint8)
        # textured faces with a face blend mode too (partition only: the bound texture's mode is the one applied)
        fbm = np.where((sel >= 0.30) & (sel < 0.40), abi.ADD_QUARTER, fbm).astype(np.uint8)
        faces["texture_id"] = tid
        faces["blend_mode"] = fbm
        faces["black_transparent"] = np.where(sel > 0.95, 0, 1).astype(np.uint8)
        # a fifth of the blend-textured faces also carry an editor alpha (set_pixel_with_editor_alpha_15, render.rs:567-591)
        faces["editor_alpha"] = np.where((sel >= 0.40) & (sel < 0.70) & (np.modf(sel * f32(997.0))[0] < 0.2), 100, 255).astype(np.uint8)
        dark = np.repeat(tid == abi.NO_TEXTURE, 3)
        dk = (splitmix64(seed ^ 0xDA4C, 9 * N) % np.uint64(12)).astype(np.uint8).reshape(3 * N, 3)
        for c, name in enumerate(("r", "g", "b")):
            verts[name] = np.where(dark, dk[:, c], verts[name]).astype(np.uint8)
    elif variant == "float":
        settings.use_fixed_point = False
    elif variant != "bench":
        raise ValueError(variant)

    textures = [t.to_texture15() for t in indexed]
    return Scene(f"{config}:{variant}:{N}", W, H, verts, faces, textures, indexed, Camera(), settings)


def wire_grid_scene(back_first=True, width=320, height=240, n=9):
    """Exercises the wireframe phases (render.rs:2574-2635) with the reference's default settings (z-buffer, Gouraud,
    back-face wireframe).  Two back-facing grids with SHARED vertices (so most edges repeat inside a grid) sit at depths
    1000 and 2000 on the same screen positions (so edges also repeat ACROSS the grids with different depths), and a
    front-facing quad at depth 1500 covers the left half: which grid comes first in face order decides whose depths the
    de-duplicated edges carry, i.e. whether they survive the `z < zbuffer` test over the quad."""
    from .rtypes import make_vertices, make_faces
    vs = (min(width, height) / 2.0) * 0.75

    def grid(cz, jitter):
        pts = []
        for j in range(n):
            for i in range(n):
                px = 20.0 + i * (width - 40.0) / (n - 1)
                py = 20.0 + j * (height - 40.0) / (n - 1)
                z = cz + jitter * ((i * 7 + j * 13) % 5)
                pts.append(((px - width / 2.0) / vs * (z + 5.0) / 4.0, (py - height / 2.0) / vs * (z + 5.0) / 4.0, z))
        tris = []
        for j in range(n - 1):
            for i in range(n - 1):
                a, b, c, d = j * n + i, j * n + i + 1, (j + 1) * n + i, (j + 1) * n + i + 1
                tris += [(a, b, c), (b, d, c)]      # clockwise on screen (y down) -> signed area > 0?  fixed below by probing
        return pts, tris

    near, tn = grid(1000.0, 3.0)
    far, tf = grid(2000.0, 0.0)
    quad = [(-1.0, -1.0), (0.0, -1.0), (0.0, 1.0), (-1.0, 1.0)]
    qz = 1500.0
    qpts = [((qx * width / 2.0) / vs * (qz + 5.0) / 4.0, (qy * height / 2.0) / vs * (qz + 5.0) / 4.0, qz) for qx, qy in quad]
    groups = [(far, tf), (near, tn)] if back_first else [(near, tn), (far, tf)]
    pts, tris, base = [], [], 0
    for g, t in groups:
        pts += g
        tris += [(a + base, c + base, b + base) for a, b, c in t]     # flipped winding: back-faces
        base += len(g)
    pts += qpts
    tris += [(base, base + 1, base + 2), (base, base + 2, base + 3)]
    v = make_vertices(len(pts))
    v["pos"] = np.array(pts, np.float32)
    v["uv"] = (v["pos"][:, :2] / 2000.0).astype(np.float32)
    v["normal"] = (0.0, 0.0, -1.0)
    v["r"], v["g"], v["b"], v["blend"] = 200, 180, 160, 0
    f = make_faces(len(tris))
    f["v"] = np.array(tris, np.uint32)
    f["texture_id"] = 0
    f["black_transparent"] = 1
    f["editor_alpha"] = 255
    tex = Texture15.checkerboard(32, 32, 0x7FFF, 0x3DEF)
    return Scene("wire-grid", width, height, v, f, [tex], [], Camera(), RasterSettings())

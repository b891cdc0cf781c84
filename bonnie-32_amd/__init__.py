"""bonnie-32_amd — MI355X-native (gfx950) implementation of bonnie-32's `src/rasterizer` hot path:
vertex transform + fixed-point snap, painter's-algorithm sort, affine textured fill with RGB555 dither
(reference: src/rasterizer/render.rs:2302-2638 `render_mesh_15`).

Layout:
  csrc/          hand-written HIP kernels + the C-ABI library (include/b32raster.h)
  abi.py         ctypes mirror of the C ABI, library loader (fails loudly if the .so is missing)
  rtypes.py      mirrors of the reference API types (Vertex/Face/Texture15/Camera/RasterSettings/...)
  rasterizer.py  Framebuffer + render_mesh_15 with the reference's signature, running on the GPU
  scenegen.py    deterministic synthetic scenes of BASELINE.md (C1..C5 + parity variants)
  parallel.py    screen-band sharding across ranks + the RCCL gather of band rows
  build.py       hipcc recipe for csrc/
"""
from . import abi, rtypes  # noqa: F401
from . import rtypes as types  # noqa: F401  (alias: mirrors the reference module name src/rasterizer/types.rs)
from .rtypes import (Camera, Color, IndexedTexture, Light, RasterSettings, RasterTimings, Texture, Texture15,  # noqa: F401
                    make_faces, make_vertices)

__all__ = ["abi", "types", "Camera", "Color", "IndexedTexture", "Light", "RasterSettings", "RasterTimings",
           "Texture", "Texture15", "make_faces", "make_vertices"]

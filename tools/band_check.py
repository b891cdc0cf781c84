import sys
sys.path.insert(0, ".")
import numpy as np
from bonnie32_amd import rasterizer as R, scenegen, parallel
from oracle import oracle as O
sc = scenegen.make_scene("C3")
ofb = O.Framebuffer(sc.width, sc.height); ofb.clear(sc.clear_color)
O.render_mesh_15(ofb, sc.vertices, sc.faces, sc.textures, sc.camera, sc.settings)
ctx = R.Context(0)
for N in (2, 4, 8):
    fb = R.Framebuffer(sc.width, sc.height, ctx)
    rs = R.ResidentScene(fb, sc.vertices, sc.faces, sc.textures)
    for r in range(N):
        y0, y1 = parallel.band_rows(sc.height, N, r)
        fb.set_band(y0, y1); fb.clear(sc.clear_color); rs.render(sc.camera, sc.settings)
    fb.set_band(0, sc.height)
    got = fb.pixels.reshape(sc.height, sc.width, 4); exp = ofb.pixels.reshape(sc.height, sc.width, 4)
    bad = (got != exp).any(axis=2)
    rows = np.nonzero(bad.any(axis=1))[0]
    print(N, "bad px", int(bad.sum()), "rows", rows[:10], rows[-5:] if len(rows) else "")

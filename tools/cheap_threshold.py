"""Where does CHEAP coverage (inside test only + runner-up repair) stop paying against EXACT (texel rule per fragment)?
C3 geometry with atlases whose CLUT has 1 transparent entry out of K; the threshold is forced either way with b32_set_cheap_threshold."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from bonnie32_amd import rasterizer as R, scenegen
for K in (64, 32, 16, 8, 4, 2):
    for den, label in ((1, "always CHEAP"), (1000000, "always EXACT")):
        sc = scenegen.make_scene("C3")
        t = sc.indexed_textures[0]
        t.indices = (t.indices % K).astype(np.uint8)          # index 0 -> CLUT entry 0 = transparent: 1/K of the texels
        ctx = R.Context(0)
        ctx.set_async_depth(1)      # timing loops: frames back to back (a dropped frame would be reported by finish())
        ctx.set_cheap_threshold(den)
        fb = R.Framebuffer(sc.width, sc.height, ctx)
        rs = R.ResidentScene(fb, sc.vertices, sc.faces, indexed_textures=sc.indexed_textures)
        fb.clear(sc.clear_color); rs.render(sc.camera, sc.settings)
        n = 50; ctx.synchronize(); t0 = time.perf_counter()
        for i in range(n):
            fb.clear(sc.clear_color); rs.render_async()
        rs.finish(); print(f"K={K} {label}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms/frame", flush=True)
        del rs, fb, ctx

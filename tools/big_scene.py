import sys, time
sys.path.insert(0, ".")
import numpy as np
from bonnie32_amd import rasterizer as R, scenegen
from oracle import oracle as O
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
sc = scenegen.make_scene("C3", n_tris=N, seed=123)
t0 = time.time(); ofb = O.Framebuffer(sc.width, sc.height); ofb.clear(sc.clear_color)
rc, otm = O.render_mesh_15(ofb, sc.vertices, sc.faces, sc.textures, sc.camera, sc.settings); tc = time.time() - t0
ctx = R.Context(0)
ctx.set_async_depth(1)      # timing loops: frames back to back (a dropped frame would be reported by finish())
fb = R.Framebuffer(sc.width, sc.height, ctx)
rs = R.ResidentScene(fb, sc.vertices, sc.faces, sc.textures)
for counting in (0, 1):
    ctx.set_fragment_counting(counting)
    fb.clear(sc.clear_color); tm = rs.render(sc.camera, sc.settings)
    ok = np.array_equal(fb.pixels, ofb.pixels)
    for i in range(4):           # (packed streams, second frame set: built on the first frames in flight -- warm-up, not timing)
        fb.clear(sc.clear_color); rs.render_async()
    rs.finish()
    n = 20; ctx.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        fb.clear(sc.clear_color); rs.render_async()
    rs.finish(); tg = (time.perf_counter() - t0) / n
    print(f"N={N} counting={counting} bit-exact={ok} tris_drawn={tm.triangles_drawn}/{otm.triangles_drawn} frags={tm.fragments}/{otm.fragments} gpu={tg*1e3:.3f} ms ({N/tg/1e6:.0f} Mtri/s) cpu={tc*1e3:.0f} ms")

"""C3 through EXACT coverage (fragment counting on = what textures with a colour key get): ms/frame."""
import sys, time
sys.path.insert(0, ".")
from bonnie32_amd import rasterizer as R, scenegen
import bonnie32_amd as b32
sc = scenegen.make_scene("C3")
ctx = R.Context(0)
ctx.set_async_depth(1)      # timing loops: frames back to back (a dropped frame would be reported by finish())
fb = R.Framebuffer(sc.width, sc.height, ctx)
rs = R.ResidentScene(fb, sc.vertices, sc.faces, indexed_textures=sc.indexed_textures)
for name, st, counting in (("painter CHEAP", sc.settings, 0), ("painter EXACT", sc.settings, 1), ("z-buffer EXACT", b32.RasterSettings(shading=0, lights=[], backface_wireframe=False), 1)):
    ctx.set_fragment_counting(counting)
    fb.clear(sc.clear_color); rs.render(sc.camera, st)
    for _ in range(4):          # (packed streams / second frame set are built on the first frames in flight: warm-up)
        fb.clear(sc.clear_color); rs.render_async()
    rs.finish()
    n = 50; ctx.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        fb.clear(sc.clear_color); rs.render_async()
    rs.finish(); t = (time.perf_counter() - t0) / n
    print(f"{name}: {t*1e3:.3f} ms/frame")

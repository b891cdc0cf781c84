"""Does the number of tiles per workgroup slot (1200 tiles over 512 slots = 2.34 rounds on C3) cost fill time?  One-stream k_cover time
of C3's scene restricted to its first `rows` rows (b32_fb_set_band), i.e. 40 x rows / 64 tiles of the same content density."""
import sys
sys.path.insert(0, ".")
from bonnie32_amd import rasterizer as R, scenegen
which = sys.argv[1] if len(sys.argv) > 1 else "C3"
sc = scenegen.make_scene(which)
ctx = R.Context(0); ctx.set_async_depth(1); ctx.set_routes(R.Context.ROUTE_PIPELINE)
fb = R.Framebuffer(sc.width, sc.height, ctx)
rs = R.ResidentScene(fb, sc.vertices, sc.faces, indexed_textures=sc.indexed_textures)
for rows in (1920, 1856, 1792, 1728, 1664, 1600, 1536, 1408, 1280, 1024, 832, 768, 640, 384):
    fb.set_band(0, rows)
    fb.clear(sc.clear_color); rs.render(sc.camera, sc.settings)
    for _ in range(5):
        fb.clear(sc.clear_color); rs.render_async()
    rs.finish()
    ctx.set_profiling(1)
    for _ in range(30):
        fb.clear(sc.clear_color); rs.render_async()
    rs.finish()
    t = ctx.last_kernel_times().get("cover"); ctx.set_profiling(0)
    nt = 40 * rows // 64
    print(f"rows {rows}: {nt} tiles = {nt / 512:.2f} rounds: k_cover {t * 1e3:.1f} us = {t * 1e3 / nt * 512:.1f} us per round-equivalent, {t * 1e6 / nt:.1f} ns per tile", flush=True)

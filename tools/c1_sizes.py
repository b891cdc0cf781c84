"""Fixed cost of a small frame: C1's scene with 50 ... 8 000 triangles (kernel times by HIP events, frames back to back)."""
import sys, time
sys.path.insert(0, ".")
from bonnie32_amd import rasterizer as R, scenegen
for n in (50, 500, 2000, 8000):
    sc = scenegen.make_scene("C1", n_tris=n)
    ctx = R.Context(0); ctx.set_async_depth(1)
    fb = R.Framebuffer(sc.width, sc.height, ctx)
    rs = R.ResidentScene(fb, sc.vertices, sc.faces, indexed_textures=sc.indexed_textures)
    for _ in range(10):
        fb.clear(sc.clear_color); rs.render_async(sc.camera, sc.settings)
    rs.finish()
    best = 1e9
    for rep in range(3):
        ctx.synchronize(); t0 = time.perf_counter()
        for _ in range(400):
            fb.clear(sc.clear_color); rs.render_async()
        rs.finish(); best = min(best, (time.perf_counter() - t0) / 400)
    print(f"C1 with {n} triangles: {best * 1e6:.1f} us per frame, routes {dict((k, v) for k, v in ctx.route_counts().items() if v)}", flush=True)
    ctx.close()

"""Upper bound of overlapping consecutive frames' FILLS (the fill of frame i + 1 in the thinning tail of frame i's): two contexts with a
framebuffer each on one GPU, frames issued alternately -- no ordering between the two at all.  Aggregate ms per frame against one context."""
import sys, time
sys.path.insert(0, ".")
from bonnie32_amd import rasterizer as R, scenegen
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
sc = scenegen.make_scene(cfg)
def make():
    ctx = R.Context(0); ctx.set_async_depth(1)
    fb = R.Framebuffer(sc.width, sc.height, ctx)
    rs = R.ResidentScene(fb, sc.vertices, sc.faces, indexed_textures=sc.indexed_textures)
    for _ in range(10):
        fb.clear(sc.clear_color); rs.render_async(sc.camera, sc.settings)
    rs.finish()
    return ctx, fb, rs
A = make(); B = make()
def run(ctxs, n):
    for c in ctxs: c[0].synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        c = ctxs[i % len(ctxs)]
        c[1].clear(sc.clear_color); c[2].render_async()
    for c in ctxs: c[2].finish()
    return (time.perf_counter() - t0) / n
for rep in range(3):
    one = run([A], 200); two = run([A, B], 400)
    print(f"{cfg}: one context {one * 1e3:.4f} ms per frame, two contexts alternating {two * 1e3:.4f} ms per frame ({(1 - two / one) * 100:+.1f} %)", flush=True)

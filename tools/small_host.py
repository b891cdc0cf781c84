"""Is a small frame's period the host's or the GPU's?  400 frames enqueued back to back: time of the enqueue loop alone (host), then of the
finish that follows (what the GPU still had to do)."""
import sys, time
sys.path.insert(0, ".")
from bonnie32_amd import rasterizer as R, scenegen
for cfg in ("C1", "C2"):
    sc = scenegen.make_scene(cfg)
    ctx = R.Context(0); ctx.set_async_depth(1)
    fb = R.Framebuffer(sc.width, sc.height, ctx)
    rs = R.ResidentScene(fb, sc.vertices, sc.faces, indexed_textures=sc.indexed_textures)
    for i in range(10):
        fb.clear(sc.clear_color); rs.render_async(sc.camera, sc.settings)
    rs.finish()
    for rep in range(3):
        ctx.synchronize(); t0 = time.perf_counter()
        for i in range(400):
            fb.clear(sc.clear_color); rs.render_async()
        t1 = time.perf_counter(); rs.finish(); t2 = time.perf_counter()
        print(f"{cfg}: enqueue loop {(t1 - t0) / 400 * 1e6:.1f} us per frame, finish {(t2 - t1) * 1e6:.0f} us, total {(t2 - t0) / 400 * 1e6:.1f} us per frame; routes {ctx.route_counts()['pipelined']} {ctx.route_counts().get('poll_join')}", flush=True)

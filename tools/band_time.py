"""Per-rank compute time of the band-sharded frame (no gather): one GPU renders only the rows rank 0 would own at N ranks."""
import sys, time
sys.path.insert(0, ".")
from bonnie32_amd import rasterizer as R, scenegen, parallel
sc = scenegen.make_scene("C3")
ctx = R.Context(0)
import os
if os.environ.get("EXP_DEPTH"): ctx.set_pipeline_depth(int(os.environ["EXP_DEPTH"]))
import os
if os.environ.get("EXP_ROUTES"):
    ctx.set_routes(int(os.environ["EXP_ROUTES"]))
ctx.set_async_depth(1)      # timing loops: frames back to back (a dropped frame would be reported by finish())
fb = R.Framebuffer(sc.width, sc.height, ctx)
rs = R.ResidentScene(fb, sc.vertices, sc.faces, indexed_textures=sc.indexed_textures)
for N in (1, 2, 4, 8):
    for r in sorted({0, N // 2}):
        y0, y1 = parallel.band_rows(sc.height, N, r)
        fb.set_band(y0, y1)
        fb.clear(sc.clear_color); rs.render(sc.camera, sc.settings)
        ctx.set_profiling(2)
        for i in range(20):
            fb.clear(sc.clear_color); rs.render_async()
        rs.finish(); kt = ctx.last_kernel_times(); ctx.set_profiling(0)
        t = 1e9
        for rep in range(3):                      # (best of three: the first stretch after the profiled frames is not steady state)
            n = 100; ctx.synchronize(); t0 = time.perf_counter()
            for i in range(n):
                fb.clear(sc.clear_color); rs.render_async()
            rs.finish(); t1 = (time.perf_counter() - t0) / n
            t = min(t, t1)
        print(f"N={N} rank {r} rows [{y0},{y1}): {t*1e3:.3f} ms/frame  phases {kt}  pipelined {ctx.route_counts()['pipelined']}")

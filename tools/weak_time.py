"""Weak-scaling series of SURVEY 8e on ONE GPU (no gather): at N ranks the scene has N x 125 k triangles on the fixed 2560x1920
frame and this GPU renders only the rows one rank would own -- per-rank compute time, i.e. the frame time of an N-GPU run minus
the RCCL gather.  efficiency(N) = t(1) / t(N) (per-GPU work is constant by construction)."""
import sys, time
sys.path.insert(0, ".")
from bonnie32_amd import rasterizer as R, scenegen, parallel
ctx = R.Context(0)
ctx.set_async_depth(1)      # timing loops: frames back to back (a dropped frame would be reported by finish())
base = None
for N in (1, 2, 4, 8):
    sc = scenegen.make_scene("C3", n_tris=125000 * N)
    fb = R.Framebuffer(sc.width, sc.height, ctx)
    rs = R.ResidentScene(fb, sc.vertices, sc.faces, indexed_textures=sc.indexed_textures)
    worst = 0.0
    for r in sorted({0, N // 2, N - 1}):
        y0, y1 = parallel.band_rows(sc.height, N, r)
        fb.set_band(y0, y1)
        fb.clear(sc.clear_color); rs.render(sc.camera, sc.settings)
        for _ in range(20):         # (packed vertex streams, second frame set: built on the first frames in flight -- warm-up, not timing)
            fb.clear(sc.clear_color); rs.render_async()
        rs.finish()
        n = 100; ctx.synchronize(); t0 = time.perf_counter()
        for i in range(n):
            fb.clear(sc.clear_color); rs.render_async()
        rs.finish(); t = (time.perf_counter() - t0) / n
        worst = max(worst, t)
        print(f"N={N} tris={sc.n_tris} rank {r} rows [{y0},{y1}): {t*1e3:.3f} ms/frame")
    base = base or worst
    print(f"N={N}: slowest rank {worst*1e3:.3f} ms -> {sc.n_tris / worst / 1e9:.2f} Gtri/s aggregate (without the gather), weak-scaling efficiency {base / worst:.2f}")

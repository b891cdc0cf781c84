"""Probe: ms per frame of resident C3 frames after a stretch of profiled frames (b32_set_profiling(2) then 0), with and without a band."""
import sys, time, os
sys.path.insert(0, ".")
from bonnie32_amd import rasterizer as R, scenegen, parallel
sc = scenegen.make_scene("C3")
for prof, band, sync_render in ((0, False, False), (2, False, False), (2, True, True), (0, True, True), (1, False, False)):
    ctx = R.Context(0); ctx.set_async_depth(1)
    fb = R.Framebuffer(sc.width, sc.height, ctx)
    rs = R.ResidentScene(fb, sc.vertices, sc.faces, indexed_textures=sc.indexed_textures)
    if band: fb.set_band(0, sc.height)
    fb.clear(sc.clear_color); rs.render(sc.camera, sc.settings)
    if prof:
        ctx.set_profiling(prof)
    for i in range(20):
        fb.clear(sc.clear_color); rs.render_async()
    rs.finish()
    if prof:
        ctx.set_profiling(0)
    for rep in range(3):
        n = 100; ctx.synchronize(); t0 = time.perf_counter()
        for i in range(n):
            fb.clear(sc.clear_color); rs.render_async()
        rs.finish(); t = (time.perf_counter() - t0) / n
        print(prof, band, rep, round(t*1e3,4), ctx.route_counts()["pipelined"])
    del rs, fb, ctx

import os, sys, time
sys.path.insert(0, "/root/repo")
from bonnie32_amd import rasterizer as R, scenegen
import bonnie32_amd as b32
sc = scenegen.make_scene("C2", n_tris=20000, width=320, height=240)
for routes in (0, 64):
    for name, st in (("painter", sc.settings), ("game", b32.RasterSettings.game())):
        ctx = R.Context(0); ctx.set_routes(routes)
        fb = R.Framebuffer(sc.width, sc.height, ctx)
        rs = R.ResidentScene(fb, sc.vertices, sc.faces, sc.textures)
        for _ in range(3): rs.render(sc.camera, st)
        for _ in range(20): rs.render_async(sc.camera, st)
        rs.finish()
        for rep in range(2):
            t0 = time.perf_counter()
            for i in range(200): rs.render_async(sc.camera, st)
            rs.finish(); ta = (time.perf_counter() - t0) / 200
            t0 = time.perf_counter()
            for i in range(200): rs.render_async(sc.camera, st); rs.finish()
            ts = (time.perf_counter() - t0) / 200
            ctx.set_profiling(2)
            for i in range(20): rs.render_async(sc.camera, st)
            rs.finish(); kt = ctx.last_kernel_times(); ctx.set_profiling(0)
            print(f"routes_off={routes} {name}: async {ta*1e3:.4f} sync {ts*1e3:.4f} phases { {k: round(v*1e3,1) for k,v in kt.items()} } routes {ctx.route_counts()}", flush=True)
        ctx.close()

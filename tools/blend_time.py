"""C3 with 10 % of its faces in the transparent pass: ms per frame and k_blend's share (per-phase events), for experiment builds."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bonnie32_amd import rasterizer as R, scenegen
import bonnie32_amd as b32
sc = scenegen.make_scene("C3", variant="blend")
st = b32.RasterSettings.benchmark()
ctx = R.Context(0); ctx.set_async_depth(1)
if os.environ.get("EXP_ROUTES"):
    ctx.set_routes(int(os.environ["EXP_ROUTES"]))
fb = R.Framebuffer(sc.width, sc.height, ctx)
rs = R.ResidentScene(fb, sc.vertices, sc.faces, sc.textures)
for _ in range(4):
    fb.clear(sc.clear_color); rs.render_async(sc.camera, st)
try:
    rs.finish()
except Exception as e:
    print("err", e)
best = 1e9
for rep in range(3):
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(100):
        fb.clear(sc.clear_color); rs.render_async()
    rs.finish(); best = min(best, (time.perf_counter() - t0) / 100)
ctx.set_profiling(2)
for _ in range(20):
    fb.clear(sc.clear_color); rs.render_async()
tm = rs.finish(); kt = ctx.last_kernel_times()
print(json.dumps({"ms": round(best * 1e3, 4), **{k: round(v * 1e3, 1) for k, v in kt.items()}}))

"""Transparent-pass frames whose surfaces all take the literal edge-walk replay (float projection: every surface is F_SLOW), C3 and C5
geometry with 10 % of the faces in the transparent pass: ms per frame, for A/B runs of experiment builds (B32_LIB)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bonnie32_amd as b32
from bonnie32_amd import rasterizer as R, scenegen
out = {}
ctx = R.Context(0); ctx.set_async_depth(1)
for cfg in ("C3", "C5"):
    sc = scenegen.make_scene(cfg, variant="blend")
    for name, fixed in (("fixed", True), ("float", False)):
        st = b32.RasterSettings.benchmark(); st.use_fixed_point = fixed
        fb = R.Framebuffer(sc.width, sc.height, ctx)
        rs = R.ResidentScene(fb, sc.vertices, sc.faces, sc.textures)
        for _ in range(3):
            fb.clear(sc.clear_color); rs.render_async(sc.camera, st)
        rs.finish()
        best = 1e9
        for rep in range(3):
            ctx.synchronize(); t0 = time.perf_counter()
            for _ in range(30):
                fb.clear(sc.clear_color); rs.render_async()
            rs.finish(); best = min(best, (time.perf_counter() - t0) / 30)
        out[f"{cfg}:{name}"] = round(best * 1e3, 4)
        del rs, fb
print(json.dumps(out))

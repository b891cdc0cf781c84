"""C1 / C2 (320x240) resident frames back to back: ms per frame, per-phase events, hash check -- for experiment builds (B32_LIB)."""
import hashlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bonnie32_amd import rasterizer as R, scenegen
H = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "hashes.json")))
out = {}
for cfg in ("C1", "C2"):
    sc = scenegen.make_scene(cfg)
    ctx = R.Context(0); ctx.set_async_depth(1)
    if os.environ.get("EXP_DEPTH"):
        ctx.set_pipeline_depth(int(os.environ["EXP_DEPTH"]))
    if os.environ.get("EXP_GATE"):
        ctx.set_pipeline_gate(int(os.environ["EXP_GATE"]))
    if os.environ.get("EXP_ROUTES"):
        ctx.set_routes(int(os.environ["EXP_ROUTES"]))
    fb = R.Framebuffer(sc.width, sc.height, ctx)
    rs = R.ResidentScene(fb, sc.vertices, sc.faces, indexed_textures=sc.indexed_textures)
    for _ in range(5):
        fb.clear(sc.clear_color); rs.render_async(sc.camera, sc.settings)
    rs.finish()
    best = 1e9
    for rep in range(3):
        ctx.synchronize(); t0 = time.perf_counter()
        for _ in range(400):
            fb.clear(sc.clear_color); rs.render_async()
        rs.finish(); best = min(best, (time.perf_counter() - t0) / 400)
    ok = hashlib.sha256(fb.pixels).hexdigest() == H[cfg]["sha256"]
    ctx.set_profiling(2)
    for _ in range(20):
        fb.clear(sc.clear_color); rs.render_async()
    tm = rs.finish(); kt = ctx.last_kernel_times(); ctx.set_profiling(0)
    out[cfg] = {"ms": round(best * 1e3, 4), "ok": ok, "pipelined": ctx.route_counts().get("pipelined"), "pairs": tm.tile_pairs, **{k: round(v * 1e3, 1) for k, v in kt.items()}}
print(json.dumps(out))

import sys, time, hashlib, json
sys.path.insert(0, ".")
from bonnie32_amd import rasterizer as R, scenegen
H = json.load(open("tests/golden/hashes.json"))
out = {}
for cfg in ("C1", "C2"):
    sc = scenegen.make_scene(cfg)
    ctx = R.Context(0); ctx.set_async_depth(1)
    fb = R.Framebuffer(sc.width, sc.height, ctx)
    rs = R.ResidentScene(fb, sc.vertices, sc.faces, indexed_textures=sc.indexed_textures)
    for i in range(10):
        fb.clear(sc.clear_color); rs.render_async(sc.camera, sc.settings)
    rs.finish()
    best = 1e9
    for rep in range(4):
        ctx.synchronize(); t0 = time.perf_counter()
        for i in range(400):
            fb.clear(sc.clear_color); rs.render_async()
        rs.finish(); best = min(best, (time.perf_counter() - t0) / 400)
    out[cfg] = (round(best * 1e3, 5), hashlib.sha256(fb.pixels).hexdigest() == H[cfg]["sha256"], ctx.route_counts()["pipelined"], ctx.route_counts()["flag_join"])
print(out)

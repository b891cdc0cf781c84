"""Two frames in flight on one GPU: two contexts (each with its own stream, framebuffer and resident copy of the scene) draw alternate
frames, so one frame's k_setup overlaps the other frame's k_cover (whose last fifth leaves most CUs idle).  Throughput only: the
latency of a frame does not change."""
import hashlib, json, os, sys, time
sys.path.insert(0, ".")
from bonnie32_amd import rasterizer as R, scenegen
H = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "hashes.json")))
for cfg in sys.argv[1:] or ["C3", "C5"]:
    sc = scenegen.make_scene(cfg)
    for nctx in (1, 2, 3):
        sets = []
        for k in range(nctx):
            ctx = R.Context(0); ctx.set_async_depth(1)
            fb = R.Framebuffer(sc.width, sc.height, ctx)
            rs = R.ResidentScene(fb, sc.vertices, sc.faces, indexed_textures=sc.indexed_textures)
            fb.clear(sc.clear_color); rs.render_async(sc.camera, sc.settings); rs.finish()
            sets.append((ctx, fb, rs))
        n = 300
        best = 1e9
        for rep in range(3):
            for ctx, fb, rs in sets: ctx.synchronize()
            t0 = time.perf_counter()
            for i in range(n):
                ctx, fb, rs = sets[i % nctx]
                fb.clear(sc.clear_color); rs.render_async()
            for ctx, fb, rs in sets: rs.finish()
            best = min(best, (time.perf_counter() - t0) / n)
        ok = all(hashlib.sha256(fb.pixels).hexdigest() == H[cfg]["sha256"] for ctx, fb, rs in sets)
        print(f"{cfg}: {nctx} frame(s) in flight: {best * 1e3:.4f} ms/frame, frames bit-exact: {ok}", flush=True)
        for ctx, fb, rs in sets: ctx.close()

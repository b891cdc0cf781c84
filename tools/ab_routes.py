"""A/B of route bits on resident scenes: frame time (frames back to back, deep asynchronous mode) with each mask of routes switched
OFF, every frame checked against the committed hash.  usage: ab_routes.py [--configs C3,C5] [--masks 0,2048] [--frames 200] [--one-stream]"""
import argparse, hashlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
from bonnie32_amd import rasterizer as R, scenegen

ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="C3,C5")
ap.add_argument("--masks", default="0,2048")
ap.add_argument("--frames", type=int, default=200)
ap.add_argument("--one-stream", action="store_true", help="also switch the two-frames-in-flight pipeline off (per-kernel effects, no overlap)")
ap.add_argument("--repeat", type=int, default=2)
args = ap.parse_args()
HASHES = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "hashes.json")))
for cfg in args.configs.split(","):
    variant = None
    name = cfg
    if ":" in cfg: name, variant = cfg.split(":")
    sc = scenegen.make_scene(name, variant=variant) if variant else scenegen.make_scene(name)
    for rep in range(args.repeat):
        for mask in [int(m, 0) for m in args.masks.split(",")]:
            ctx = R.Context(0)
            ctx.set_async_depth(1)
            ctx.set_routes(mask | (R.Context.ROUTE_PIPELINE if args.one_stream else 0))
            fb = R.Framebuffer(sc.width, sc.height, ctx)
            rs = R.ResidentScene(fb, sc.vertices, sc.faces, indexed_textures=sc.indexed_textures) if not variant else R.ResidentScene(fb, sc.vertices, sc.faces, sc.textures)
            for _ in range(8):
                fb.clear(sc.clear_color); rs.render_async(sc.camera, sc.settings, sc.fog)
            rs.finish()
            ctx.synchronize(); t0 = time.perf_counter()
            for _ in range(args.frames):
                fb.clear(sc.clear_color); rs.render_async()
            rs.finish(); ms = (time.perf_counter() - t0) / args.frames * 1e3
            sha = hashlib.sha256(fb.pixels).hexdigest()
            ok = sha == HASHES[cfg]["sha256"]
            print(f"{cfg:9s} routes_off={mask:5d} one_stream={int(args.one_stream)} {ms:.4f} ms/frame  hash_ok={ok}  routes={ctx.route_counts()}", flush=True)
            del rs, fb, ctx

"""Fixed cost of a short timed region: wall time of N frames after a synchronisation for several N, two frames in flight against one
stream (bench.py's conditions: torch stream, bound tensor) -> per-frame slope and per-run intercept."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bonnie32_amd import rasterizer as R, scenegen
sc = scenegen.make_scene("C3")
dev = torch.device("cuda", 0)
for routes in (0, R.Context.ROUTE_PIPELINE):
    ctx = R.Context(0); ctx.set_async_depth(1); ctx.set_routes(routes)
    if os.environ.get("EXP_GATE"):
        ctx.set_pipeline_gate(int(os.environ["EXP_GATE"]))
    stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream); ctx.set_stream(stream.cuda_stream)
    frame = torch.zeros(sc.width * sc.height * 4, dtype=torch.uint8, device=dev)
    fb = R.Framebuffer.__new__(R.Framebuffer); fb.ctx = ctx
    fb.bind_device(frame.data_ptr(), sc.width, sc.height)
    rs = R.ResidentScene(fb, sc.vertices, sc.faces, indexed_textures=sc.indexed_textures)
    mode = os.environ.get("PROBE_MODE", "")
    if "exact" in mode:                       # bench.py's warm-up: fragment counting on (EXACT coverage), then off
        ctx.set_fragment_counting(1)
        fb.clear(sc.clear_color); rs.render_async(sc.camera, sc.settings); rs.finish()
        for _ in range(4):
            fb.clear(sc.clear_color); rs.render_async()
        rs.finish()
        ctx.set_fragment_counting(0)
        fb.clear(sc.clear_color); rs.render_async(); rs.finish()
    else:
        for _ in range(8):
            fb.clear(sc.clear_color); rs.render_async(sc.camera, sc.settings)
        rs.finish()
    if "band" in mode:
        fb.set_band(0, sc.height)
    if "prof" in mode:
        ctx.set_profiling_stride(8); ctx.set_profiling(1)
    xs, ys = [], []
    for n in (2, 5, 10, 20, 40, 80, 160):
        best = 1e9; reps = []
        for rep in range(5):
            torch.cuda.synchronize(dev); t0 = time.perf_counter()
            for i in range(n):
                fb.clear(sc.clear_color); rs.render_async()
            torch.cuda.synchronize(dev); dt = time.perf_counter() - t0; best = min(best, dt); reps.append(round(dt / n * 1e3, 4))
            rs.finish()
        xs.append(n); ys.append(best * 1e3)
        if n == 20:
            print("   N=20 reps:", reps, flush=True)
    a, b = np.polyfit(xs, ys, 1)
    print(f"routes_off={routes}: " + " ".join(f"N={n}:{y / n:.4f}" for n, y in zip(xs, ys)) + f"  -> {a:.4f} ms per frame + {b:.3f} ms per run", flush=True)
    ctx.close()

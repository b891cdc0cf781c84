"""Per-step device times of the first frames after a synchronisation (bench.py's conditions: torch stream, bound tensor), two frames in
flight against one stream: where a short timed region loses time.  usage: step_probe.py [n_steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bonnie32_amd import rasterizer as R, scenegen
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
sc = scenegen.make_scene("C3")
dev = torch.device("cuda", 0)
for own in (0, 1):
    for routes in (0, R.Context.ROUTE_PIPELINE):
        ctx = R.Context(0); ctx.set_async_depth(1); ctx.set_routes(routes)
        stream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(stream)
        if not own:
            ctx.set_stream(stream.cuda_stream)
        frame = torch.zeros(sc.width * sc.height * 4, dtype=torch.uint8, device=dev)
        fb = R.Framebuffer.__new__(R.Framebuffer); fb.ctx = ctx
        fb.bind_device(frame.data_ptr(), sc.width, sc.height)
        rs = R.ResidentScene(fb, sc.vertices, sc.faces, indexed_textures=sc.indexed_textures)
        for _ in range(6):
            fb.clear(sc.clear_color); rs.render_async(sc.camera, sc.settings)
        rs.finish()
        for rep in range(2):
            torch.cuda.synchronize(dev); ctx.synchronize()
            t0 = time.perf_counter()
            host = []
            for i in range(n):
                h0 = time.perf_counter()
                fb.clear(sc.clear_color); rs.render_async()
                host.append((time.perf_counter() - h0) * 1e6)
            t_enq = time.perf_counter() - t0
            ctx.synchronize(); torch.cuda.synchronize(dev)
            t_all = time.perf_counter() - t0
            rs.finish()
            print(f"own_stream={own} routes_off={routes} rep={rep}: {t_all / n * 1e3:.4f} ms/step wall over {n} steps; host enqueue {t_enq / n * 1e6:.1f} us/step "
                  f"(first 4: {' '.join(f'{h:.0f}' for h in host[:4])}, median {sorted(host)[n // 2]:.0f})", flush=True)
        ctx.close()

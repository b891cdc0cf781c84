"""Which of bench.py's conditions costs time against the plain resident loop?  (own stream / torch stream, own framebuffer / bound torch
tensor, band set to the whole frame or not)"""
import os, sys, time
sys.path.insert(0, ".")
import torch
from bonnie32_amd import rasterizer as R, scenegen
sc = scenegen.make_scene("C3")
dev = torch.device("cuda:0")
CASES = ((0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 1))
for use_stream, bind, band in (CASES[:1] * 3 if os.environ.get("QUICK") else CASES):
    ctx = R.Context(0); ctx.set_async_depth(1)
    stream = torch.cuda.Stream(device=dev)
    if use_stream:
        ctx.set_stream(stream.cuda_stream)
    if bind:
        frame = torch.zeros(sc.height * sc.width * 4, dtype=torch.uint8, device=dev)
        fb = R.Framebuffer.__new__(R.Framebuffer); fb.ctx = ctx
        fb.bind_device(frame.data_ptr(), sc.width, sc.height)
    else:
        fb = R.Framebuffer(sc.width, sc.height, ctx)
    if band:
        fb.set_band(0, sc.height)
    rs = R.ResidentScene(fb, sc.vertices, sc.faces, indexed_textures=sc.indexed_textures)
    for i in range(5):
        fb.clear(sc.clear_color); rs.render_async(sc.camera, sc.settings); rs.finish()
    best = 1e9
    for rep in range(3):
        ctx.synchronize(); torch.cuda.synchronize(dev); t0 = time.perf_counter()
        for i in range(200):
            fb.clear(sc.clear_color); rs.render_async()
        rs.finish(); torch.cuda.synchronize(dev); best = min(best, (time.perf_counter() - t0) / 200)
    ctx.set_profiling(2)
    for i in range(20):
        fb.clear(sc.clear_color); rs.render_async()
    rs.finish(); kt = ctx.last_kernel_times(); ctx.set_profiling(0)
    print(f"torch stream {use_stream} bound tensor {bind} band set {band}: {best*1e3:.4f} ms/frame  " + " ".join(f"{k} {v*1e3:.1f}" for k, v in kt.items()), flush=True)

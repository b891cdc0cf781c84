"""Experiment: why does k_fill take longer in the bench loop than in a single frame?"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as g
g.build()
from bonnie32_amd import rasterizer as R, scenegen
sc = scenegen.make_scene("C3")
ctx = R.Context(0)
fb = R.Framebuffer(sc.width, sc.height, ctx)
rs = R.ResidentScene(fb, sc.vertices, sc.faces, indexed_textures=sc.indexed_textures)
def run(n, clear=True, label=""):
    ctx.set_profiling(2)
    for i in range(n):
        if clear: fb.clear(sc.clear_color)
        if i == 0: rs.render_async(sc.camera, sc.settings)
        else: rs.render_async()
    rs.finish()
    print(label, n, {k: round(v, 4) for k, v in ctx.last_kernel_times().items()})
    ctx.set_profiling(0)
run(1, label="single")
time.sleep(0.5)
run(1, label="single after sleep")
run(2, label="two")
run(5, label="five")
run(30, label="thirty")
run(30, clear=False, label="thirty noclear")
time.sleep(1.0)
run(1, label="single after sleep")
import torch
frame = torch.zeros(sc.width*sc.height*4, dtype=torch.uint8, device="cuda")
fb.bind_device(frame.data_ptr(), sc.width, sc.height)
run(1, label="torch fb single")
run(30, label="torch fb thirty")
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
run(1, label="torch stream single")
run(30, label="torch stream thirty")

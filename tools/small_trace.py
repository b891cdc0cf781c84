"""C1 or C2 resident frames back to back, for rocprofv3 --kernel-trace (tools/pipeline_trace.py show prints the timeline).
usage: small_trace.py C1|C2 [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bonnie32_amd import rasterizer as R, scenegen
cfg = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
sc = scenegen.make_scene(cfg)
ctx = R.Context(0); ctx.set_async_depth(1)
fb = R.Framebuffer(sc.width, sc.height, ctx)
rs = R.ResidentScene(fb, sc.vertices, sc.faces, indexed_textures=sc.indexed_textures)
for _ in range(5):
    fb.clear(sc.clear_color); rs.render_async(sc.camera, sc.settings)
rs.finish()
for _ in range(n):
    fb.clear(sc.clear_color); rs.render_async()
rs.finish()

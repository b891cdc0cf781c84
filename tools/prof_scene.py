"""Render one scenegen scene a few times (resident, default path) -- meant to be run under rocprofv3 --kernel-trace --stats.
usage: prof_scene.py CONFIG VARIANT [zbuffer]"""
import sys
sys.path.insert(0, ".")
import bonnie32_amd as b32
from bonnie32_amd import rasterizer as R, scenegen
cfg, variant = sys.argv[1], sys.argv[2]
sc = scenegen.make_scene(cfg, variant=variant)
if len(sys.argv) > 3:
    sc.settings.use_zbuffer = True
ctx = R.Context(0)
import os
if os.environ.get("EXP_ROUTES"):
    ctx.set_routes(int(os.environ["EXP_ROUTES"]))          # e.g. 64: one stream (no next-frame setup kernel beside the fill)
ctx.set_async_depth(1)      # timing loops: frames back to back (a dropped frame would be reported by finish())
fb = R.Framebuffer(sc.width, sc.height, ctx)
rs = R.ResidentScene(fb, sc.vertices, sc.faces, sc.textures)
for i in range(12):
    fb.clear(sc.clear_color); rs.render_async(sc.camera, sc.settings)
rs.finish()

"""One of bench_modes.py's settings on the C3 scene, a dozen resident frames -- meant to run under rocprofv3 --kernel-trace.
usage: mode_prof.py zbuffer|game|game8|blendz|default [routes_off]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bonnie32_amd as b32
from bonnie32_amd import rasterizer as R, scenegen
mode = sys.argv[1]
variant = "blend" if mode == "blendz" else "gouraud"
sc = scenegen.make_scene("C3", variant=variant)
st = {"zbuffer": b32.RasterSettings(shading=0, lights=[], backface_wireframe=False), "game": b32.RasterSettings.game(),
      "game8": b32.RasterSettings(backface_wireframe=False, use_rgb555=False), "default": b32.RasterSettings(),
      "blendz": b32.RasterSettings(shading=0, lights=[], backface_wireframe=False)}[mode]
ctx = R.Context(0)
if len(sys.argv) > 2:
    ctx.set_routes(int(sys.argv[2]))
ctx.set_async_depth(1)
fb = R.Framebuffer(sc.width, sc.height, ctx)
if mode == "game8":
    rs = R.ResidentScene(fb, sc.vertices, sc.faces, textures8=[b32.Texture.from_texture15(t) for t in sc.textures])
else:
    rs = R.ResidentScene(fb, sc.vertices, sc.faces, sc.textures)
for i in range(12):
    fb.clear(sc.clear_color); rs.render_async(sc.camera, st)
rs.finish()

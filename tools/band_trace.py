"""One rank's band of C4 (rows [y0, y1) of the C3 scene) back to back, for rocprofv3 --kernel-trace (tools/trace_cmd.sh): usage band_trace.py N rank [frames]"""
import sys
sys.path.insert(0, ".")
from bonnie32_amd import rasterizer as R, scenegen, parallel
N = int(sys.argv[1]); r = int(sys.argv[2]); n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
sc = scenegen.make_scene("C3")
ctx = R.Context(0); ctx.set_async_depth(1)
fb = R.Framebuffer(sc.width, sc.height, ctx)
rs = R.ResidentScene(fb, sc.vertices, sc.faces, indexed_textures=sc.indexed_textures)
y0, y1 = parallel.band_rows(sc.height, N, r)
fb.set_band(y0, y1)
for _ in range(8):
    fb.clear(sc.clear_color); rs.render_async(sc.camera, sc.settings)
rs.finish()
for _ in range(n):
    fb.clear(sc.clear_color); rs.render_async()
rs.finish()

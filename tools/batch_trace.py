"""The batched console frame alone (b32_frame_begin / _add_scene / _end), for rocprofv3 --kernel-trace: which kernels a frame costs.
usage: batch_trace.py [n_meshes] [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bonnie32_amd as b32
from bonnie32_amd import rasterizer as R, scenegen
n_meshes = int(sys.argv[1]) if len(sys.argv) > 1 else 12
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rng = np.random.default_rng(2024)
meshes = [scenegen.make_scene("C1", n_tris=int(rng.integers(300, 3000)), seed=1000 + i, variant=("blend" if i % 4 == 3 else "gouraud"),
                              bbox_px=float(os.environ.get("BBOX") or rng.choice([150.0, 400.0, 900.0]))) for i in range(n_meshes)]
st = b32.RasterSettings.game()
st.lights = [b32.Light.directional((-1.0, -1.0, -1.0), 0.7), b32.Light.point((0.0, -100.0, 1500.0), 3000.0, 1.2)]
fog = (1500.0, 3000.0, 5800.0, b32.Color(40, 50, 70))
ctx = R.Context(0)
if os.environ.get("ROUTES_OFF"):                      # e.g. 2 = B32_ROUTE_CUT_TILES off (64-row tiles)
    ctx.set_routes(int(os.environ["ROUTES_OFF"]))
fb = R.Framebuffer(meshes[0].width, meshes[0].height, ctx)
slots = [R.ResidentScene(fb, sc.vertices, sc.faces, sc.textures).detach() for sc in meshes]
def frame():
    fb.clear(b32.Color(10, 10, 30))
    ctx.frame_begin(meshes[0].camera, st)
    for rs in slots:
        ctx.frame_add(rs, fog=fog)
    ctx.frame_end()
    return ctx.finish()
frame(); frame()
t0 = time.perf_counter()
for _ in range(frames):
    tm = frame()
print(f"{(time.perf_counter() - t0) / frames * 1e3:.3f} ms per frame; routes {ctx.route_counts()} batch {ctx.batch_counts()}")

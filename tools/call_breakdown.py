"""Where the drop-in call's wall time goes: scene upload (host slices -> HBM) vs the synchronous draw, small meshes."""
import sys, time
sys.path.insert(0, ".")
from bonnie32_amd import rasterizer as R, scenegen
ctx = R.Context(0)
for n in (200, 2000):
    sc = scenegen.make_scene("C1", n_tris=n)
    fb = R.Framebuffer(sc.width, sc.height, ctx)
    for i in range(5):
        R.render_mesh_15(fb, sc.vertices, sc.faces, sc.textures, sc.camera, sc.settings)
    N = 300
    t0 = time.perf_counter()
    for i in range(N):
        rs = R.ResidentScene(fb, sc.vertices, sc.faces, sc.textures)
    ctx.synchronize(); t_up = (time.perf_counter() - t0) / N
    rs.render(sc.camera, sc.settings)
    t0 = time.perf_counter()
    for i in range(N):
        rs.render(sc.camera, sc.settings)
    t_draw = (time.perf_counter() - t0) / N
    t0 = time.perf_counter()
    for i in range(N):
        R.render_mesh_15(fb, sc.vertices, sc.faces, sc.textures, sc.camera, sc.settings)
    t_all = (time.perf_counter() - t0) / N
    print(f"{n} tris: upload {t_up*1e6:.1f} us, synchronous draw {t_draw*1e6:.1f} us, drop-in call {t_all*1e6:.1f} us")

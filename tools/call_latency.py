"""Wall time of the drop-in call (host slices in, per-call upload, synchronous) vs the resident call, small reference-sized meshes."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from bonnie32_amd import rasterizer as R, scenegen
import bonnie32_amd as b32
ctx = R.Context(0)
for cfg, n in (("C1", 2000), ("C1", 200), ("C3", 2000), ("C3", 200), ("C2", 20000)):      # C1 / C2: 64x64 texture; C3: 256x256 (128 KB)
    sc = scenegen.make_scene(cfg, n_tris=n, width=320, height=240, bbox_px=64.0 if cfg != "C2" else None)
    fb = R.Framebuffer(sc.width, sc.height, ctx)
    for st_name, st in (("painter", sc.settings), ("game()", b32.RasterSettings.game())):
        for i in range(5):
            R.render_mesh_15(fb, sc.vertices, sc.faces, sc.textures, sc.camera, st)
        N = 200
        t0 = time.perf_counter()
        for i in range(N):
            R.render_mesh_15(fb, sc.vertices, sc.faces, sc.textures, sc.camera, st)
        t_drop = (time.perf_counter() - t0) / N
        rs = R.ResidentScene(fb, sc.vertices, sc.faces, sc.textures)
        for _ in range(3):          # (first launches of a kernel instantiation, packed streams on the second frame: warm-up)
            rs.render(sc.camera, st)
        t0 = time.perf_counter()
        for i in range(N):
            rs.render_async(sc.camera, st)
        rs.finish(); t_res = (time.perf_counter() - t0) / N
        t0 = time.perf_counter()
        for i in range(N):
            rs.render_async(sc.camera, st); rs.finish()
        t_sync = (time.perf_counter() - t0) / N
        print(f"{cfg} {n:6d} tris {sc.width}x{sc.height} {st_name:8s}: drop-in {t_drop*1e3:.3f} ms/call, resident async {t_res*1e3:.3f}, resident + finish {t_sync*1e3:.3f}")

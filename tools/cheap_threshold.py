"""Where does CHEAP coverage (inside test only + runner-up repair) stop paying against EXACT (texel rule per fragment)?
C3 geometry with atlases whose CLUT has 1 transparent entry out of K."""
import os, sys, time, subprocess
sys.path.insert(0, ".")
if len(sys.argv) > 1:
    K = int(sys.argv[1])
    import numpy as np
    from bonnie32_amd import rasterizer as R, scenegen
    sc = scenegen.make_scene("C3")
    t = sc.indexed_textures[0]
    t.indices = (t.indices % K).astype(np.uint8)          # index 0 -> CLUT entry 0 = transparent: 1/K of the texels
    ctx = R.Context(0)
    ctx.set_async_depth(1)      # timing loops: frames back to back (a dropped frame would be reported by finish())
    fb = R.Framebuffer(sc.width, sc.height, ctx)
    rs = R.ResidentScene(fb, sc.vertices, sc.faces, indexed_textures=sc.indexed_textures)
    fb.clear(sc.clear_color); rs.render(sc.camera, sc.settings)
    n = 50; ctx.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        fb.clear(sc.clear_color); rs.render_async()
    rs.finish(); print(f"K={K} B32_CHEAP_DEN={os.environ.get('B32_CHEAP_DEN')}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms/frame")
else:
    for K in (64, 32, 16, 8, 4, 2):
        for den in ("1", "1000000"):                        # 1: always CHEAP, 1000000: always EXACT
            subprocess.run([sys.executable, __file__, str(K)], env=dict(os.environ, B32_CHEAP_DEN=den))

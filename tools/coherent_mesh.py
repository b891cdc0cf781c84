"""k_setup's tile binning on a COHERENT mesh: the C3 scene with its faces ordered by the screen tile they land in (what a real, spatially
ordered mesh looks like: neighbours in memory are neighbours on screen), against the scene's own random face order."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from bonnie32_amd import rasterizer as R, scenegen
sc = scenegen.make_scene("C3")
pos = sc.vertices["pos"].reshape(-1, 3, 3)
c = pos.mean(axis=1)
k = (c[:, 2] + 5.0) / 4.0
vs = min(sc.width, sc.height) / 2 * 0.75
px = c[:, 0] / k * vs + sc.width / 2; py = c[:, 1] / k * vs + sc.height / 2
def morton3(p, bits=10):
    q = ((p - p.min(axis=0)) / (np.ptp(p, axis=0) + 1e-9) * ((1 << bits) - 1)).astype(np.uint64)
    code = np.zeros(len(p), np.uint64)
    for b in range(bits):
        for d in range(3):
            code |= ((q[:, d] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + d)
    return code
for name, order in (("random order", np.arange(len(c))), ("tile order", np.lexsort((px // 64, py // 64))), ("scanline order", np.lexsort((px, py // 8))),
                    ("3-D Morton order of the centroids (camera-independent)", np.argsort(morton3(c), kind="stable")),
                    ("Morton order of the direction from the origin, then depth", np.argsort(morton3(np.stack([c[:, 0] / c[:, 2], c[:, 1] / c[:, 2], np.zeros(len(c))], 1)), kind="stable"))):
    v = sc.vertices.reshape(-1, 3)[order].reshape(-1).copy()
    f = sc.faces.copy(); f["v"] = np.arange(3 * len(order), dtype=np.uint32).reshape(-1, 3)
    ctx = R.Context(0); ctx.set_async_depth(1)
    fb = R.Framebuffer(sc.width, sc.height, ctx)
    rs = R.ResidentScene(fb, v, f, indexed_textures=sc.indexed_textures)
    for i in range(3):
        fb.clear(sc.clear_color); rs.render_async(sc.camera, sc.settings); rs.finish()
    n = 100; ctx.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        fb.clear(sc.clear_color); rs.render_async()
    rs.finish(); t = (time.perf_counter() - t0) / n
    ctx.set_profiling(2)
    for i in range(20):
        fb.clear(sc.clear_color); rs.render_async()
    rs.finish(); kt = ctx.last_kernel_times(); ctx.set_profiling(0)
    print(f"{name}: {t * 1e3:.3f} ms/frame  " + " ".join(f"{a} {b * 1e3:.1f}" for a, b in kt.items()), "routes", {a: b for a, b in ctx.route_counts().items() if b}, flush=True)
    ctx.close()

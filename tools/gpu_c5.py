"""Full-size C5 (painter's-sort / overdraw stress): parity vs oracle on both paths + timings."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as g
g.build()
from bonnie32_amd import rasterizer as R, scenegen
from oracle import oracle as O
cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
sc = scenegen.make_scene(cfg)
ofb = O.Framebuffer(sc.width, sc.height); ofb.clear(sc.clear_color)
t0 = time.time(); rc, otm, d = O.render_mesh_15(ofb, sc.vertices, sc.faces, sc.textures, sc.camera, sc.settings, dump=True); tc = time.time() - t0
print(f"oracle {cfg}: {tc:.1f}s drawn={otm.triangles_drawn} frags={otm.fragments}")
ctx = R.Context(0)
ctx.set_async_depth(1)      # timing loops: frames back to back (a dropped frame would be reported by finish())
fb = R.Framebuffer(sc.width, sc.height, ctx)
rs = R.ResidentScene(fb, sc.vertices, sc.faces, indexed_textures=sc.indexed_textures)
for mode in (1, 0):
    ctx.set_fragment_counting(mode)
    fb.clear(sc.clear_color); tm = rs.render(sc.camera, sc.settings)
    ok = np.array_equal(fb.pixels, ofb.pixels)
    order_ok = np.array_equal(ctx.last_draw_order(len(sc.faces)), d["draw_order"])
    ctx.set_profiling(2)
    for i in range(10):
        fb.clear(sc.clear_color); rs.render_async(sc.camera, sc.settings) if i == 0 else rs.render_async()
    t = rs.finish(); kt = ctx.last_kernel_times(); ctx.set_profiling(0)
    tot = sum(kt.values())
    print(f"mode={'exact' if mode else 'fast '} parity={ok} order={order_ok} frags={tm.fragments}/{otm.fragments} pairs={tm.tile_pairs} phases_ms={ {k: round(v,3) for k,v in kt.items()} } total={tot:.3f}ms -> {sc.n_tris/tot/1e3:.0f} Mtri/s, {otm.fragments/tot/1e6:.1f} Gpix/s; cpu {tc*1e3:.0f} ms")

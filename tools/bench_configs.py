"""Fills BASELINE.md section 3: every config on 1 GPU (default fast path, scene resident in HBM, HIP-event kernel time summed
per frame + wall time over back-to-back frames) next to the CPU oracle (1 thread) on the same box.  Prints a markdown table."""
import json, sys, time
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as g
g.build()
from bonnie32_amd import rasterizer as R, scenegen
from oracle import oracle as O

ctx = R.Context(0)
ctx.set_async_depth(1)      # timing loops: frames back to back (a dropped frame would be reported by finish())
rows = []
for cfg in ["C1", "C2", "C3", "C5"]:
    sc = scenegen.make_scene(cfg)
    ofb = O.Framebuffer(sc.width, sc.height)
    reps, tcpu = 0, 0.0
    while tcpu < 4.0 and reps < 20:
        ofb.clear(sc.clear_color); t0 = time.perf_counter()
        rc, otm = O.render_mesh_15(ofb, sc.vertices, sc.faces, sc.textures, sc.camera, sc.settings, fast=True); tcpu += time.perf_counter() - t0; reps += 1
    tcpu /= reps
    fb = R.Framebuffer(sc.width, sc.height, ctx)
    rs = R.ResidentScene(fb, sc.vertices, sc.faces, indexed_textures=sc.indexed_textures)
    ctx.set_fragment_counting(1); fb.clear(sc.clear_color); tm = rs.render(sc.camera, sc.settings)
    exact_ok = np.array_equal(fb.pixels, ofb.pixels) and tm.fragments == otm.fragments
    ctx.set_fragment_counting(0)
    for i in range(5):
        fb.clear(sc.clear_color); rs.render_async(sc.camera, sc.settings); 
    rs.finish()
    fast_ok = np.array_equal(fb.pixels, ofb.pixels)
    n = 200
    ctx.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        fb.clear(sc.clear_color); rs.render_async()
    rs.finish(); tg = (time.perf_counter() - t0) / n
    ctx.set_profiling(2)
    for i in range(20):
        fb.clear(sc.clear_color); rs.render_async()
    rs.finish(); kt = ctx.last_kernel_times(); ctx.set_profiling(0)
    alg = 36 * len(sc.vertices) + 20 * sc.n_tris + 16 * otm.triangles_drawn + 8 * sc.width * sc.height + sum(t.width * t.height * 2 for t in sc.textures)
    rows.append((cfg, f"{sc.width}x{sc.height}", sc.n_tris, otm.triangles_drawn, otm.fragments, tcpu * 1e3, sc.n_tris / tcpu / 1e6, otm.fragments / tcpu / 1e6,
                 tg * 1e3, sc.n_tris / tg / 1e6, otm.fragments / tg / 1e6, alg / tg / 8e12, exact_ok and fast_ok, sum(kt.values())))
print("| Config | Frame | Triangles (drawn) | Fragments | CPU oracle 1 thread: ms, Mtri/s, Mpix/s | 1x MI355X: ms/frame, Mtri/s, Mpix/s | B_alg / t / 8 TB/s | bit-exact | GPU/CPU |")
print("|---|---|---|---|---|---|---|---|---|")
for r in rows:
    print(f"| {r[0]} | {r[1]} | {r[2]:,} ({r[3]:,}) | {r[4]:,} | {r[5]:.1f}, {r[6]:.2f}, {r[7]:.1f} | {r[8]:.3f} (kernels {r[13]:.3f}), {r[9]:.0f}, {r[10]:.0f} | {r[11]*100:.1f} % | {'yes' if r[12] else 'NO'} | {r[5]/r[8]:.0f}x |")

"""The all-cores CPU baseline (oracle.render_all_cores: one row band per process) at several process counts, and the share of a frame
that every process repeats (transform, cull, surface setup, sort = a frame with an empty band)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from bonnie32_amd import scenegen
from oracle import oracle as O
sc = scenegen.make_scene(sys.argv[1] if len(sys.argv) > 1 else "C3")
fb = O.Framebuffer(sc.width, sc.height); fb.clear(sc.clear_color)
L = O.lib()
for name, band in (("whole frame", (0, 0xFFFFFFFF)), ("empty band (replicated part only)", (0, 0))):
    L.b32o_set_row_band(*band)
    O.render_mesh_15(fb, sc.vertices, sc.faces, sc.textures, sc.camera, sc.settings, sc.fog)
    t0 = time.perf_counter(); O.render_mesh_15(fb, sc.vertices, sc.faces, sc.textures, sc.camera, sc.settings, sc.fog); t = time.perf_counter() - t0
    print(f"1 process, {name}: {t*1e3:.1f} ms")
L.b32o_set_row_band(0, 0xFFFFFFFF)
fb.clear(sc.clear_color); O.render_mesh_15(fb, sc.vertices, sc.faces, sc.textures, sc.camera, sc.settings, sc.fog)
for n in (2, 4, 8, 16, 32, 64):
    t, frame = O.render_all_cores(sc, n, reps=2)
    print(f"{n} processes: {t*1e3:.1f} ms per frame (slowest band), identical: {np.array_equal(frame, fb.pixels)}")

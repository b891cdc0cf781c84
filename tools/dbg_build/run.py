import sys, ctypes as C
sys.path.insert(0, ".")
import numpy as np
from bonnie32_amd import abi
abi._lib = None
lib = abi.load_library.__wrapped__ if hasattr(abi.load_library, "__wrapped__") else None
abi.LIB_PATH = "tools/dbg_build/csrc/libb32dbg.so"
L = abi.load_library()
from bonnie32_amd import rasterizer as R, scenegen
sc = scenegen.make_scene(sys.argv[1] if len(sys.argv) > 1 else "C3")
ctx = R.Context(0)
fb = R.Framebuffer(sc.width, sc.height, ctx)
rs = R.ResidentScene(fb, sc.vertices, sc.faces, indexed_textures=sc.indexed_textures)
names = ["tex-stage", "tile-fetch", "clear+bar", "phaseA", "barA(wait)", "phaseB", "B2+C", "end-bar"]
for mode in (1, 0):
    ctx.set_fragment_counting(mode)
    for i in range(3):
        fb.clear(sc.clear_color); rs.render_async(sc.camera, sc.settings); rs.finish()
    ctx.set_profiling(2); fb.clear(sc.clear_color); rs.render_async(); rs.finish(); kt = ctx.last_kernel_times(); ctx.set_profiling(0)
    n = 256 * 16 * 8
    buf = np.zeros(n, np.uint64)
    L.b32_debug_read.argtypes = [C.c_void_p, C.c_int]
    L.b32_debug_read(buf.ctypes.data, n)
    T = buf.reshape(256, 16, 8).astype(np.float64)
    tot = T.sum(axis=2)
    print("mode", "exact" if mode else "cheap", "fill ms", round(kt.get("fill", 0), 4), "cycles/wave mean", int(tot.mean()), "max", int(tot.max()))
    for i, nme in enumerate(names):
        print(f"   {nme:12s} mean {T[:,:,i].mean():10.0f}  ({100*T[:,:,i].mean()/tot.mean():5.1f}%)   max-wave {T[:,:,i].max():10.0f}")

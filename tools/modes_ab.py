"""GPU ms per frame of the C3 scene under several settings, no CPU oracle: for A/B runs of experiment builds (B32_LIB) in one gpurun call.
usage: modes_ab.py [mode ...]   modes: painter zbuffer game blend blendz game8 default gamex zx"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bonnie32_amd as b32
from bonnie32_amd import rasterizer as R, scenegen
modes = sys.argv[1:] or ["painter", "zbuffer", "game"]
ctx = R.Context(0); ctx.set_async_depth(1)
if os.environ.get("EXP_DEPTH"):
    ctx.set_pipeline_depth(int(os.environ["EXP_DEPTH"]))
if os.environ.get("EXP_GATE"):
    ctx.set_pipeline_gate(int(os.environ["EXP_GATE"]))
out = {}
scenes = {}
for mode in modes:
    variant = "blend" if mode.startswith("blend") else "gouraud"
    if variant not in scenes:
        scenes[variant] = scenegen.make_scene("C3", variant=variant)
    sc = scenes[variant]
    st = {"painter": b32.RasterSettings.benchmark(), "blend": b32.RasterSettings.benchmark(),
          "zbuffer": b32.RasterSettings(shading=0, lights=[], backface_wireframe=False), "game": b32.RasterSettings.game(),
          "game8": b32.RasterSettings(backface_wireframe=False, use_rgb555=False), "default": b32.RasterSettings(), "gamex": b32.RasterSettings.game(), "zx": b32.RasterSettings(shading=0, lights=[], backface_wireframe=False),
          "blendz": b32.RasterSettings(shading=0, lights=[], backface_wireframe=False)}[mode]
    ctx.set_fragment_counting(1 if mode in ("gamex", "zx") else 0)      # (gamex / zx: EXACT coverage, as colour-keyed textures force it)
    fb = R.Framebuffer(sc.width, sc.height, ctx)
    if mode == "game8":
        rs = R.ResidentScene(fb, sc.vertices, sc.faces, textures8=[b32.Texture.from_texture15(t) for t in sc.textures])
    else:
        rs = R.ResidentScene(fb, sc.vertices, sc.faces, sc.textures)
    for i in range(5):
        fb.clear(sc.clear_color); rs.render_async(sc.camera, st)
    rs.finish()
    best = 1e9
    for rep in range(4):
        ctx.synchronize(); t0 = time.perf_counter()
        for i in range(100):
            fb.clear(sc.clear_color); rs.render_async()
        rs.finish(); best = min(best, (time.perf_counter() - t0) / 100)
    out[mode] = round(best * 1e3, 4)
    del rs, fb
print(json.dumps(out))

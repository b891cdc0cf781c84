"""C3 scene under the settings the reference's callers actually use (not only the BASELINE painter's configuration):
ms/frame on 1 GPU (resident scene, default library path), bit-exactness against the CPU oracle, CPU oracle time."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as g
g.build()
import bonnie32_amd as b32
from bonnie32_amd import rasterizer as R, scenegen
from oracle import oracle as O

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ctx = R.Context(0)
import os
if os.environ.get("EXP_ROUTES"):
    ctx.set_routes(int(os.environ["EXP_ROUTES"]))
ctx.set_async_depth(1)      # timing loops: frames back to back (a dropped frame would be reported by finish())
if os.environ.get("EXP_GATE"): ctx.set_pipeline_gate(int(os.environ["EXP_GATE"]))      # (experiments: see b32_set_pipeline_gate)
if os.environ.get("EXP_DEPTH"): ctx.set_pipeline_depth(int(os.environ["EXP_DEPTH"]))
base = scenegen.make_scene("C3", n_tris=N, variant="gouraud")
tex8 = [b32.Texture.from_texture15(t) for t in base.textures]
MODES = [
    ("painter (BASELINE cfg)", b32.RasterSettings.benchmark(), False),
    ("z-buffer, no lights", b32.RasterSettings(shading=0, lights=[], backface_wireframe=False), False),
    ("RasterSettings::game()", b32.RasterSettings.game(), False),
    ("RasterSettings::default()", b32.RasterSettings(), False),
    ("painter, 10 % faces in the transparent pass", b32.RasterSettings.benchmark(), "blend"),
    ("game(), 10 % faces in the transparent pass", b32.RasterSettings(shading=0, lights=[], backface_wireframe=False), "blend"),
    ("8-bit painter", b32.RasterSettings(use_zbuffer=False, shading=0, lights=[], backface_wireframe=False, use_rgb555=False), True),
    ("8-bit game()", b32.RasterSettings(backface_wireframe=False, use_rgb555=False), True),
]
blend = scenegen.make_scene("C3", n_tris=N, variant="blend")
print("| settings | GPU ms/frame | Mtri/s | CPU oracle ms | bit-exact |")
print("|---|---|---|---|---|")
for name, st, f8 in MODES:
    scene = base
    if f8 == "blend":
        scene, f8 = blend, False
    ofb = O.Framebuffer(base.width, base.height); ofb.clear(base.clear_color)
    t0 = time.perf_counter()
    skip_cpu = False      # (round 4: the oracle de-duplicates the wireframe edges through a hash set proven equal to the reference's O(n^2) scan)
    if skip_cpu:
        rc, otm = 0, None
    elif f8:
        rc, otm = O.render_mesh(ofb, base.vertices, base.faces, tex8, base.camera, st)
    else:
        rc, otm = O.render_mesh_15(ofb, scene.vertices, scene.faces, scene.textures, scene.camera, st)
    tcpu = time.perf_counter() - t0
    assert rc == 0
    fb = R.Framebuffer(base.width, base.height, ctx)
    rs = R.ResidentScene(fb, base.vertices, base.faces, textures8=tex8) if f8 else R.ResidentScene(fb, scene.vertices, scene.faces, scene.textures)
    fb.clear(base.clear_color); rs.render(base.camera, st)
    ok = skip_cpu or (np.array_equal(fb.pixels, ofb.pixels) and (not st.use_zbuffer or np.array_equal(fb.zbuffer.view(np.uint32), ofb.zbuffer.view(np.uint32))))
    for i in range(3):
        fb.clear(base.clear_color); rs.render_async(base.camera, st)
    rs.finish()
    n = 50
    ctx.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        fb.clear(base.clear_color); rs.render_async()
    rs.finish(); tg = (time.perf_counter() - t0) / n
    print(f"| {name} | {tg*1e3:.3f} | {N/tg/1e6:.0f} | {'(skipped: O(n^2) dedup)' if skip_cpu else f'{tcpu*1e3:.0f}'} | {'n/a' if skip_cpu else ('yes' if ok else 'NO')} |")

"""Repro hunt: batched frames whose base settings have backface_wireframe on (soak seed 3101 #1517)."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bonnie32_amd as b32
from bonnie32_amd import rasterizer as R, scenegen
from oracle import oracle as O
W, H = 640, 480
fails = 0
for trial in range(40):
    rng = np.random.default_rng(500 + trial)
    names = [("C2", "bench", 9000), ("C2", "blend", 7), ("C2", "bench", 300), ("C2", "gouraud", 9000), ("C2", "blend5", 300)]
    meshes = [scenegen.make_scene(c, n_tris=n, seed=int(rng.integers(1 << 30)), variant=v, width=W, height=H, bbox_px=float(rng.choice([60.0, 400.0, 900.0]))) for c, v, n in names]
    st = b32.RasterSettings.game(); st.use_zbuffer = True; st.backface_wireframe = True
    st.shading = int(rng.integers(0, 3)); st.lights = [b32.Light.directional((-1.0, -1.0, -1.0), 0.7)][:int(rng.integers(0, 2))]
    per = [dict(ambient=float(rng.uniform(0.0, 0.6)), backface_cull=bool(rng.integers(4) > 0), fog=None) for _ in meshes]
    cam = b32.Camera(); cam.position = (float(rng.normal(0, 30)), float(rng.normal(0, 30)), float(rng.normal(0, 60)))
    ofb = O.Framebuffer(W, H); ofb.clear(b32.Color(1, 2, 3))
    for sc, p in zip(meshes, per):
        s2 = copy.copy(st); s2.ambient = p["ambient"]; s2.backface_cull = p["backface_cull"]
        assert O.render_mesh_15(ofb, sc.vertices, sc.faces, sc.textures, cam, s2, None)[0] == 0
    res = {}
    for mode in ("batched", "routes_batch_off", "slots_async", "slots_sync"):
        ctx = R.Context(0)
        if mode == "routes_batch_off":
            ctx.set_routes(R.Context.ROUTE_BATCH)
        fb = R.Framebuffer(W, H, ctx); fb.clear(b32.Color(1, 2, 3))
        slots = [R.ResidentScene(fb, sc.vertices, sc.faces, sc.textures).detach() for sc in meshes]
        if mode in ("batched", "routes_batch_off"):
            ctx.frame_begin(cam, st)
            for rs, p in zip(slots, per):
                ctx.frame_add(rs, **p)
            ctx.frame_end(); ctx.finish()
        else:
            for rs, p in zip(slots, per):
                s2 = copy.copy(st); s2.ambient = p["ambient"]; s2.backface_cull = p["backface_cull"]
                if mode == "slots_async":
                    rs.render_async(cam, s2)
                else:
                    rs.render(cam, s2)
            slots[-1].finish()
        res[mode] = bool(np.array_equal(fb.pixels, ofb.pixels) and np.array_equal(fb.zbuffer.view(np.uint32), ofb.zbuffer.view(np.uint32)))
        ctx.close()
    if not all(res.values()):
        fails += 1
        print("trial", trial, res, [p["backface_cull"] for p in per], flush=True)
print("fails", fails)

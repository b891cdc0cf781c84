"""Randomised parity soak: random scenes x random settings (both pixel formats, every mode) GPU vs CPU oracle, bit-exact.
usage: soak.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
import bonnie32_amd as b32
from bonnie32_amd import rasterizer as R, scenegen, abi
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = R.Context(0)
t_end = time.time() + budget
n = fails = drawn = refused = 0

def sky_case():
    """clear_gradient + skybox sphere + star sprites + nearest upscale with a random camera orientation; returns ok."""
    from tests.test_sky_present import sky_mesh
    W, H = [(320, 240), (640, 480), (97, 333), (1000, 64)][rng.integers(4)]
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 2] = -q[:, 2]
    q = q.astype(np.float32)
    pos = tuple(float(x) for x in rng.normal(0, 100, 3).astype(np.float32))
    cam = b32.Camera(position=pos, basis_x=tuple(map(float, q[:, 0])), basis_y=tuple(map(float, q[:, 1])), basis_z=tuple(map(float, q[:, 2])))
    verts, faces = sky_mesh(pos, int(rng.choice([8, 24, 48])), int(rng.choice([6, 16, 32])), seed=int(rng.integers(1 << 20)))
    top = b32.Color(*[int(x) for x in rng.integers(0, 256, 3)]); bot = b32.Color(*[int(x) for x in rng.integers(0, 256, 3)])
    k = int(rng.integers(0, 300))
    cx = rng.integers(-4, W + 4, k); cy = rng.integers(-4, H + 4, k); rgb = rng.integers(0, 256, (k, 3)); size = float(rng.choice([0.5, 1.0, 2.0, 3.7]))
    ofb = O.Framebuffer(W, H); ofb.clear_gradient(top, bot)
    fb = R.Framebuffer(W, H, ctx); fb.clear_gradient(top, bot)
    if ofb.render_skybox_mesh(verts, faces, cam) != 0:
        return True
    fb.render_skybox_mesh(verts, faces, cam)
    if k:
        ofb.draw_star_diamonds(cx, cy, rgb, size); fb.draw_star_diamonds(cx, cy, rgb, size)
    ok = np.array_equal(fb.pixels, ofb.pixels)
    dw, dh = int(rng.integers(1, 2000)), int(rng.integers(1, 1200))
    out = fb.present_nearest(dw, dh)
    sx = ((2 * np.arange(dw) + 1) * W) // (2 * dw); sy = ((2 * np.arange(dh) + 1) * H) // (2 * dh)
    return ok and np.array_equal(out, ofb.image()[sy][:, sx])

def slots_case():
    """A multi-mesh frame through scene slots (b32_scene_swap): 2-5 meshes of random size (across the inline-binning limits) and
    random per-mesh settings, RGB555 and 8-bit mixed, two frames with different cameras, persistent depth; sometimes a mesh with a
    bad vertex index in the middle (the finish reports it, that mesh draws nothing)."""
    W, H = [(320, 240), (333, 197), (640, 480), (64, 64)][rng.integers(4)]
    k = int(rng.integers(2, 6))
    items = []
    for i in range(k):
        ntri = int(rng.choice([5, 300, 2048, 2500, 8192, 9000]))
        sc = scenegen.make_scene(str(rng.choice(["C1", "C2"])), n_tris=ntri, seed=int(rng.integers(1 << 30)), variant=str(rng.choice(["bench", "gouraud", "blend", "blend5"])),
                                 width=W, height=H, bbox_px=float(rng.choice([60.0, 900.0, 20000.0])))
        st = sc.settings
        st.use_zbuffer = bool(rng.integers(2)); st.backface_cull = bool(rng.integers(2)); st.affine_textures = bool(rng.integers(4))
        f8 = bool(rng.integers(4) == 0)
        st.use_rgb555 = not f8
        tex8 = [b32.Texture.from_texture15(t, int(rng.choice([0, 1, 3]))) for t in sc.textures] if f8 else None
        bad = bool(rng.integers(8) == 0)
        faces = sc.faces.copy()
        if bad:
            faces["v"][int(rng.integers(len(faces))), int(rng.integers(3))] = len(sc.vertices) + 5
        items.append((sc, st, tex8, faces, bad))
    ctx.set_fragment_counting(int(rng.integers(2)))
    fb = R.Framebuffer(W, H, ctx)
    ofb = O.Framebuffer(W, H)
    slots = [R.ResidentScene(fb, sc.vertices, faces, None if tex8 else sc.textures, textures8=tex8).detach() for sc, st, tex8, faces, bad in items]
    ok = True
    try:
        for frame in range(2):
            cam = b32.Camera(); cam.position = (float(rng.normal(0, 30)), float(rng.normal(0, 30)), float(rng.normal(0, 60)))
            col = b32.Color(int(rng.integers(256)), int(rng.integers(256)), int(rng.integers(256)))
            ofb.clear(col); fb.clear(col)
            want_rc = 0
            for (sc, st, tex8, faces, bad), rs in zip(items, slots):
                rc = (O.render_mesh(ofb, sc.vertices, faces, tex8, cam, st) if tex8 else O.render_mesh_15(ofb, sc.vertices, faces, sc.textures, cam, st))[0]
                if rc and not want_rc:
                    want_rc = rc
                rs.render_async(cam, st)
            try:
                slots[-1].finish(); got_rc = 0
            except R.B32Error as e:
                got_rc = e.code
            ok = ok and (got_rc != 0) == (want_rc != 0) and np.array_equal(fb.pixels, ofb.pixels) and np.array_equal(fb.zbuffer.view(np.uint32), ofb.zbuffer.view(np.uint32))
    finally:
        for rs in slots:
            rs.close()
    if not ok:
        print("   slots:", [(sc.name, st.use_zbuffer, tex8 is not None, bad) for sc, st, tex8, faces, bad in items], W, H, flush=True)
    return ok

def batch_case():
    """b32_frame_begin / _add_scene / _end with random meshes, per-mesh ambient / fog / backface_cull, random base settings (z-buffer or
    painter's, lights or not, sometimes x-ray / 8-bit-free wireframe bases that force the mesh-by-mesh fallback) against the oracle's
    sequential render_mesh_15 calls; two frames, the second with another camera (merged meshes come from the cache)."""
    import copy
    W, H = [(320, 240), (333, 197), (640, 480)][rng.integers(3)]
    k = int(rng.integers(2, 9))
    meshes = [scenegen.make_scene(str(rng.choice(["C1", "C2"])), n_tris=int(rng.choice([7, 300, 2500, 9000])), seed=int(rng.integers(1 << 30)),
                                  variant=str(rng.choice(["bench", "gouraud", "gouraud", "blend", "blend5"])), width=W, height=H,
                                  bbox_px=float(rng.choice([60.0, 400.0, 900.0]))) for _ in range(k)]
    st = b32.RasterSettings.game()
    st.use_zbuffer = bool(rng.integers(4) > 0)
    st.shading = int(rng.integers(0, 3))
    st.lights = [b32.Light.directional((-1.0, -1.0, -1.0), 0.7), b32.Light.point((0.0, -100.0, 1500.0), 3000.0, 1.2)][:int(rng.integers(0, 3))]
    st.xray_mode = bool(rng.integers(12) == 0)
    st.affine_textures = bool(rng.integers(4) > 0)
    st.backface_wireframe = bool(rng.integers(10) == 0)
    fogs = [None, (1500.0, 3000.0, 5800.0, b32.Color(40, 50, 70)), (800.0, 2500.0, 5000.0, b32.Color(90, 20, 20))]
    per = [dict(ambient=float(rng.uniform(0.0, 0.6)), backface_cull=bool(rng.integers(4) > 0), fog=fogs[int(rng.integers(3))]) for _ in range(k)]
    ctx.set_fragment_counting(int(rng.integers(2)))
    fb = R.Framebuffer(W, H, ctx)
    ofb = O.Framebuffer(W, H)
    slots = [R.ResidentScene(fb, sc.vertices, sc.faces, sc.textures).detach() for sc in meshes]
    ok = True
    try:
        for frame in range(2):
            cam = b32.Camera(); cam.position = (float(rng.normal(0, 30)), float(rng.normal(0, 30)), float(rng.normal(0, 60)))
            col = b32.Color(int(rng.integers(256)), int(rng.integers(256)), int(rng.integers(256)))
            ofb.clear(col); fb.clear(col)
            for sc, p in zip(meshes, per):
                s2 = copy.copy(st); s2.ambient = p["ambient"]; s2.backface_cull = p["backface_cull"]
                assert O.render_mesh_15(ofb, sc.vertices, sc.faces, sc.textures, cam, s2, p["fog"])[0] == 0
            ctx.frame_begin(cam, st)
            for rs, p in zip(slots, per):
                ctx.frame_add(rs, **p)
            ctx.frame_end(); ctx.finish()
            ok = ok and np.array_equal(fb.pixels, ofb.pixels) and np.array_equal(fb.zbuffer.view(np.uint32), ofb.zbuffer.view(np.uint32))
    finally:
        for rs in slots:
            rs.close()
    if not ok:
        print("   batch:", [(sc.name, sc.n_tris) for sc in meshes], W, H, "z", st.use_zbuffer, "xray", st.xray_mode, "wire", st.backface_wireframe, flush=True)
    return ok

def pipeline_case():
    """Two frames in flight: a frame with more tiles than workgroup slots (so that the setup kernel of frame i+1 goes to the side stream),
    deep asynchronous mode, several cameras back to back without a clear in between (z-buffer: colour and depth accumulate; painter's:
    later frames overwrite), random gate; against the oracle's sequential calls."""
    W, H = [(2560, 1440), (1920, 1440), (2048, 2048)][rng.integers(3)]
    sc = scenegen.make_scene("C3", n_tris=int(rng.choice([9000, 30000, 70000])), seed=int(rng.integers(1 << 30)),
                             variant=str(rng.choice(["bench", "gouraud", "blend"])), width=W, height=H, bbox_px=float(rng.choice([60.0, 200.0, 700.0])))
    st = sc.settings
    st.use_zbuffer = bool(rng.integers(2)); st.backface_cull = bool(rng.integers(3) > 0)
    cams = [b32.Camera(position=(float(rng.normal(0, 40)), float(rng.normal(0, 40)), float(rng.normal(0, 200)))) for _ in range(int(rng.integers(3, 7)))]
    ofb = O.Framebuffer(W, H); ofb.clear(sc.clear_color)
    for cam in cams:
        assert O.render_mesh_15(ofb, sc.vertices, sc.faces, sc.textures, cam, st, None, fast=True)[0] == 0
    c2 = R.Context(0)
    try:
        c2.set_async_depth(1); c2.set_pipeline_gate(int(rng.choice([0, 1, 300, 1000, 1150, 1600, 2000]))); c2.set_pipeline_depth(int(rng.choice([2, 3])))
        c2.set_fragment_counting(int(rng.integers(2)))
        fb = R.Framebuffer(W, H, c2); fb.clear(sc.clear_color)
        rs = R.ResidentScene(fb, sc.vertices, sc.faces, sc.textures)
        rs.render_async(cams[0], st); rs.finish()               # (deep mode: list regions settled by a first synchronous frame)
        for cam in cams[1:]:
            rs.render_async(cam, st)
        rs.finish()
        ok = np.array_equal(fb.pixels, ofb.pixels) and (not st.use_zbuffer or np.array_equal(fb.zbuffer.view(np.uint32), ofb.zbuffer.view(np.uint32)))
        ok = ok and c2.route_counts()["pipelined"] >= 1
    finally:
        c2.close()
    if not ok:
        print("   pipeline:", sc.name, W, H, "z", st.use_zbuffer, len(cams), flush=True)
    return ok

while time.time() < t_end:
    n += 1
    if rng.integers(25) == 0 and not os.environ.get("SOAK_FORCE"):
        drawn += 1
        if not pipeline_case():
            fails += 1
            print(f"FAIL #{n} two frames in flight", flush=True)
        continue
    if rng.integers(10) == 0 and not os.environ.get("SOAK_FORCE"):
        drawn += 1
        if not batch_case():
            fails += 1
            print(f"FAIL #{n} batched frame", flush=True)
        continue
    if rng.integers(10) == 0 and not os.environ.get("SOAK_FORCE"):
        drawn += 1
        if not slots_case():
            fails += 1
            print(f"FAIL #{n} scene slots", flush=True)
        continue
    if rng.integers(12) == 0:
        drawn += 1
        if not sky_case():
            fails += 1
            print(f"FAIL #{n} sky/gradient/stars/present", flush=True)
        continue
    cfg = rng.choice(["C1", "C2", "C5"])
    W, H = [(320, 240), (333, 197), (640, 480), (64, 64), (1280, 720), (97, 801)][rng.integers(6)]
    ntri = int(rng.choice([1, 7, 300, 2500, 20000]))
    if os.environ.get("SOAK_FORCE"):                     # screen-filling triangles: keep the CPU oracle's work bounded
        ntri = min(ntri, 300); W, H = min(W, 640), min(H, 480)
    variant = rng.choice(["bench", "gouraud", "blend", "blend5", "float"])        # blend5: all five blend_rgb555 modes, both as texture and as face mode
    sc = scenegen.make_scene(cfg, n_tris=ntri, seed=int(rng.integers(1 << 30)), variant=variant, width=W, height=H,
                             bbox_px=float(rng.choice([4.0, 60.0, 900.0, 20000.0])))
    force = os.environ.get("SOAK_FORCE", "")            # "orthoz": every scene hostile + orthographic + z-buffer (signed-zero depths)
    hostile = bool(rng.integers(5) == 0 or force)
    if hostile:                                           # hostile geometry: huge / degenerate / near-plane coordinates, wild UVs
        v = sc.vertices
        k = max(1, len(v) // 20)
        idx = rng.choice(len(v), k, replace=False)
        v["pos"][idx, int(rng.integers(3))] = (10.0 ** rng.uniform(3, 30, k)) * rng.choice([-1, 1], k)
        idx = rng.choice(len(v), k, replace=False)
        v["pos"][idx, 2] = rng.choice([0.1, 0.100001, 0.0999, -3.0, 0.0, 1e-3], k)
        idx = rng.choice(len(v), k, replace=False)
        v["uv"][idx] = rng.choice([1e9, -1e9, 0.0, 1.0, -1.0, 123456.789], (k, 2))
        if len(v) >= 6:
            v["pos"][0:3] = v["pos"][3:6]
    st = sc.settings
    st.use_zbuffer = bool(rng.integers(2)); st.affine_textures = bool(rng.integers(4) > 0); st.dithering = bool(rng.integers(4) > 0)
    st.backface_cull = bool(rng.integers(3) > 0)
    # (no wireframe on hostile coordinates: the oracle walks lines literally, 2^29 steps per edge would take minutes)
    st.backface_wireframe = bool(rng.integers(4) == 0) and ntri <= 2500 and not hostile
    st.wireframe_overlay = bool(rng.integers(10) == 0) and ntri <= 2500 and not hostile
    st.xray_mode = bool(rng.integers(8) == 0)
    if force == "orthoz":
        st.use_zbuffer = True; st.xray_mode = False
    if rng.integers(6) == 0 or force == "orthoz":
        st.ortho_projection = (float(rng.choice([0.02, 0.1, 1.0])), float(rng.normal(0, 50)), float(rng.normal(0, 50)))
        sc.camera.position = (0.0, 0.0, float(rng.choice([0.0, 2500.0])))
    if rng.integers(3) == 0 and variant != "gouraud":
        st.shading = int(rng.integers(1, 3))
        st.lights = [b32.Light.directional((float(rng.normal()), float(rng.normal()), float(rng.normal()) + 0.1), float(rng.uniform(0.2, 1.5))),
                     b32.Light.point((float(rng.normal(0, 500)), float(rng.normal(0, 500)), float(rng.uniform(300, 3000))), float(rng.uniform(500, 5000)), 1.2),
                     # spot light, direction normalized (Light::spot) or over-long (acos of |dot| > 1 is NaN: the reference's lit branch)
                     (b32.Light.spot if rng.integers(2) else (lambda p, d, a, r, i: b32.Light(b32.abi.LIGHT_SPOT, position=p, direction=d, angle=a, radius=r, intensity=i)))(
                         (float(rng.normal(0, 600)), float(rng.normal(0, 600)), float(rng.uniform(-300, 2500))),
                         (float(rng.normal(0, 0.5)), float(rng.normal(0, 0.5)), float(rng.uniform(0.3, 1.2))),
                         float(rng.uniform(0.05, 3.2)), float(rng.uniform(800, 8000)), float(rng.uniform(0.5, 2.0)))][:int(rng.integers(1, 4))]
        if rng.integers(3) == 0:
            st.lights = st.lights[::-1]
        st.ambient = float(rng.uniform(0.0, 0.6))
    fog = None
    fmt8 = bool(rng.integers(3) == 0)
    if not fmt8 and rng.integers(4) == 0:
        fog = (float(rng.uniform(300, 2000)), float(rng.choice([0.0, 800.0, 3000.0])), float(rng.uniform(2000, 7000)), b32.Color(int(rng.integers(256)), 90, 120))
    if rng.integers(4) == 0:
        sc.faces["editor_alpha"][::int(rng.integers(2, 9))] = int(rng.choice([0, 1, 128, 254]))
    if rng.integers(4) == 0:
        sc.faces["black_transparent"][::3] = 0
    counting = int(rng.integers(2))
    ofb = O.Framebuffer(W, H); ofb.clear(sc.clear_color)
    if fmt8:
        st.use_rgb555 = False
        stp = int(rng.choice([0, 1, 2, 3, 4]))
        tex8 = [b32.Texture.from_texture15(t, stp) for t in sc.textures]
        rc, otm = O.render_mesh(ofb, sc.vertices, sc.faces, tex8, sc.camera, st)
    else:
        rc, otm = O.render_mesh_15(ofb, sc.vertices, sc.faces, sc.textures, sc.camera, st, fog)
    ctx.set_fragment_counting(counting)
    fb = R.Framebuffer(W, H, ctx); fb.clear(sc.clear_color)
    bands = [(0, H)] if rng.integers(3) else [(0, H // 3), (H // 3, H // 3 + 37), (H // 3 + 37, H)]
    desc = f"#{n} {cfg} {W}x{H} tris={ntri} {variant} fmt8={fmt8} z={st.use_zbuffer} xray={st.xray_mode} ortho={st.ortho_projection is not None} wire={st.backface_wireframe}/{st.wireframe_overlay} shading={st.shading} fog={fog is not None} counting={counting} bands={len(bands)}"
    try:
        # (half of the RGB555 scenes whose textures are the scene's own index atlases go up indexed: device-side CLUT expansion, and --
        # with one texture -- CLUT + index bytes sampled from LDS wherever they fit beside the tile planes, B32_ROUTE_LDS_ATLAS)
        as_indexed = (not fmt8) and len(sc.indexed_textures) == len(sc.textures) and rng.integers(2) == 0 and \
            all(np.array_equal(a.to_texture15().pixels, t.pixels) and a.blend_mode == t.blend_mode for a, t in zip(sc.indexed_textures, sc.textures))
        desc += f" indexed={as_indexed}"
        rs = R.ResidentScene(fb, sc.vertices, sc.faces, textures8=tex8) if fmt8 else \
            (R.ResidentScene(fb, sc.vertices, sc.faces, None, sc.indexed_textures) if as_indexed else R.ResidentScene(fb, sc.vertices, sc.faces, sc.textures))
        grc = 0
        for b0, b1 in bands:
            fb.set_band(b0, b1)
            if len(bands) > 1:
                fb.clear(sc.clear_color)
            tm = rs.render(sc.camera, st, None if fmt8 else fog)
        fb.set_band(0, H)
    except R.B32Error as e:
        grc = e.code
    ok = grc == rc
    drawn += rc == 0; refused += rc != 0
    if ok and rc == 0:
        ok = np.array_equal(fb.pixels, ofb.pixels) and tm.triangles_drawn == otm.triangles_drawn
        if ok and st.use_zbuffer:
            ok = np.array_equal(fb.zbuffer.view(np.uint32), ofb.zbuffer.view(np.uint32))
        if ok and tm.fragments and len(bands) == 1:
            ok = tm.fragments == otm.fragments
    if ok and rc == 0 and len(bands) == 1 and rng.integers(10) < 4:
        # a second mesh on the same framebuffer through the drop-in call: read-modify-write of colours and of the persistent depth buffer
        sc2 = scenegen.make_scene(str(rng.choice(["C1", "C2"])), n_tris=int(rng.choice([7, 300, 2500])), seed=int(rng.integers(1 << 30)),
                                  variant=str(rng.choice(["bench", "blend", "gouraud"])), width=W, height=H, bbox_px=float(rng.choice([60.0, 900.0, 20000.0])))
        st2 = sc2.settings
        st2.use_zbuffer = bool(rng.integers(2)); st2.backface_cull = bool(rng.integers(2)); st2.use_rgb555 = not fmt8
        desc += f" + mesh2 {sc2.name} z={st2.use_zbuffer}"
        try:
            if fmt8:
                t2 = [b32.Texture.from_texture15(t, int(rng.choice([0, 1, 3]))) for t in sc2.textures]
                rc2, otm2 = O.render_mesh(ofb, sc2.vertices, sc2.faces, t2, sc2.camera, st2)
                tm2 = R.render_mesh(fb, sc2.vertices, sc2.faces, t2, sc2.camera, st2)
            else:
                rc2, otm2 = O.render_mesh_15(ofb, sc2.vertices, sc2.faces, sc2.textures, sc2.camera, st2)
                tm2 = R.render_mesh_15(fb, sc2.vertices, sc2.faces, sc2.textures, sc2.camera, st2)
            ok = rc2 == 0 and np.array_equal(fb.pixels, ofb.pixels) and tm2.triangles_drawn == otm2.triangles_drawn
            if ok and (st.use_zbuffer or st2.use_zbuffer):
                ok = np.array_equal(fb.zbuffer.view(np.uint32), ofb.zbuffer.view(np.uint32))
            st = st2 if st2.use_zbuffer else st
        except R.B32Error as e:
            ok = False; grc = e.code
    if not ok:
        fails += 1
        print("FAIL", desc, "rc", rc, grc, flush=True)
        if rc == 0 and grc == 0:
            got = fb.pixels.reshape(H, W, 4); exp = ofb.pixels.reshape(H, W, 4)
            bad = (got != exp).any(axis=2)
            zbad = (fb.zbuffer.view(np.uint32) != ofb.zbuffer.view(np.uint32)).reshape(H, W) if st.use_zbuffer else np.zeros((H, W), bool)
            ys, xs = np.nonzero(bad | zbad)
            print(f"   pixels {int(bad.sum())} zbuf {int(zbad.sum())} tris {tm.triangles_drawn}/{otm.triangles_drawn} first {list(zip(ys[:4].tolist(), xs[:4].tolist()))}", flush=True)
            for y, x in list(zip(ys[:3].tolist(), xs[:3].tolist())):
                print(f"   ({y},{x}) gpu {got[y, x].tolist()} z {fb.zbuffer.reshape(H, W)[y, x]!r}  cpu {exp[y, x].tolist()} z {ofb.zbuffer.reshape(H, W)[y, x]!r}", flush=True)
        if os.environ.get("SOAK_STOP"):
            break
print(f"soak: {n} scenes ({drawn} drawn, {refused} refused by both sides), {fails} failures")

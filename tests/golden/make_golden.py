"""Regenerates tests/golden/*.  Run from the repo root:  python tests/golden/make_golden.py

The reference (Rust) cannot be built or run in this image, and its own tests pin no pixel, so these vectors are
produced by the CPU oracle (oracle/b32_oracle.c) after it has been cross-checked against the independent numpy
restatement (oracle/np_model.py).  They pin the oracle against regressions and travel to the GPU box as data.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bonnie32_amd as b32  # noqa: E402
from bonnie32_amd import scenegen  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests.golden.ref_fixtures import cube_scene  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def fog_scene():
    sc = scenegen.make_scene("C1", n_tris=3000, bbox_px=300.0, seed=77)
    sc.fog = (1000.0, 3000.0, 5500.0, b32.Color(90, 100, 120))
    sc.settings.shading = b32.abi.SHADE_FLAT
    sc.settings.lights = [b32.Light.point((100.0, -50.0, 900.0), 2500.0, 1.5), b32.Light.directional((0.3, -1.0, 0.2), 0.5)]
    sc.settings.backface_cull = False
    return sc


def spot_lights():
    """Spot lights (render.rs:1038-1058): one built like Light::spot (normalized direction), one with an over-long direction so that
    |dot| > 1 makes acos NaN for surfaces near its axis (the reference then takes the lit branch with NaN -> min(NaN, 1) = 1)."""
    return [b32.Light.spot((0.0, 0.0, -200.0), (0.1, -0.05, 1.0), 0.5, 7000.0, 1.6),
            b32.Light(b32.abi.LIGHT_SPOT, position=(800.0, -300.0, 2500.0), direction=(-0.9, 0.2, 0.6), angle=1.2, radius=3000.0,
                      color=b32.Color(255, 120, 60), intensity=1.0),
            b32.Light.directional((0.2, 1.0, 0.3), 0.15)]


def spot_scene(flat=False, zbuf=False):
    sc = scenegen.make_scene("C1", n_tris=2500, variant="gouraud", seed=71, bbox_px=500.0)
    sc.settings.lights = spot_lights()
    sc.settings.ambient = 0.15
    if flat:
        sc.settings.shading = b32.abi.SHADE_FLAT
    sc.settings.use_zbuffer = zbuf
    return sc


def needle_scene(n=160, seed=5, width=320, height=240):
    """Needles millions of pixels long with their tip on the screen (the far end sits next to the camera plane, where the projection
    blows up): doubled area up to 2^23, so the -1e-4 tolerance of the inside test (render.rs:1536-1542) is worth a few hundred edge
    units, and the pixels straight BEYOND the tip -- outside the triangle's own bounding box -- still pass it.  The reference never
    looks at them: its loops stop at the bounding box, which is therefore part of the semantics.  (Integer-snapped coordinates
    below 2^22 with products below 2^24: these surfaces take the closed-form walk, not the literal replay.)"""
    # (the 256-entry CLUT of C3 has too few black texels for EXACT coverage to be chosen: with fragment counting off this is CHEAP coverage)
    sc = scenegen.make_scene("C3", n_tris=n, seed=seed, bbox_px=100.0, width=width, height=height)
    rng = np.random.default_rng(seed)
    W, H = sc.width, sc.height
    vs = np.float32((np.float32(min(W, H)) / np.float32(2.0)) * np.float32(0.75))
    pos = sc.vertices["pos"].reshape(n, 3, 3)
    for i in range(n):
        x0 = int(rng.integers(8, W - 8)); y0 = int(rng.integers(8, H - 8))
        side = -1.0 if rng.integers(2) else 1.0                  # the far end lies to the left or to the right
        zt = np.float32(rng.uniform(300, 3000)); zf = np.float32(0.2)
        dy = sorted(int(v) for v in rng.choice(np.arange(-3, 4), 2, replace=False))
        tip = [(x0 - W / 2) / vs * (zt + 5) / 4, (y0 - H / 2) / vs * (zt + 5) / 4, zt]
        far = [[side * float(rng.uniform(3.0e4, 5.0e4)), (y0 + d - H / 2) / vs * (zf + 5) / 4, zf] for d in dy]
        far[1][0] = far[0][0]
        tri = np.array([tip, far[0], far[1]], np.float32)
        pos[i] = tri[[0, 2, 1]] if rng.integers(2) else tri
    sc.settings.backface_cull = False
    return sc


def persp_scene():
    sc = scenegen.make_scene("C1", seed=31, bbox_px=400.0)
    sc.settings.affine_textures = False          # perspective-correct UVs (render.rs:1568-1579)
    return sc


def zbuf_scene(variant="bench", seed=17):
    sc = scenegen.make_scene("C1", seed=seed, variant=variant, bbox_px=150.0)
    sc.settings.use_zbuffer = True               # the reference's default / RasterSettings::game() (types.rs:1455-1495)
    return sc


def blend5_scene(zbuf=False):
    """All five PS1 blend modes (blend_rgb555, render.rs:1093-1145) as texture blend mode over STP texels and as face blend mode
    of untextured faces, editor alpha on part of them (scenegen variant "blend5")."""
    sc = scenegen.make_scene("C1", variant="blend5", seed=91, bbox_px=250.0)
    sc.settings.use_zbuffer = zbuf
    return sc


def ortho_scene():
    sc = scenegen.make_scene("C1", seed=41, variant="blend", bbox_px=900.0)
    sc.camera.position = (30.0, -20.0, 3000.0)   # part of the scene ends up at negative camera depth: no near cull in ortho
    sc.settings.ortho_projection = (0.05, 10.0, -15.0)
    return sc


def xray_scene(zbuf):
    sc = scenegen.make_scene("C1", seed=43, variant="blend", bbox_px=200.0)
    sc.settings.xray_mode = True
    sc.settings.use_zbuffer = zbuf
    return sc


def default_settings_scene():
    sc = scenegen.make_scene("C1", seed=47, variant="gouraud", bbox_px=200.0)
    sc.settings = b32.RasterSettings()            # reference defaults: z-buffer, Gouraud + directional light, back-face wireframe
    return sc


def wire_painter_scene(overlay):
    sc = scenegen.make_scene("C1", seed=53, bbox_px=300.0)
    sc.settings.backface_wireframe = True
    sc.settings.wireframe_overlay = overlay
    return sc


def cube_default_scene():
    sc = cube_scene()
    sc.settings = b32.RasterSettings()            # far-side faces are back-faces: their wireframe must fail the depth test
    return sc


SCENES = {
    "C1:ortho": ortho_scene,
    "C1:xray": lambda: xray_scene(False),
    "C1:xray-zbuf": lambda: xray_scene(True),
    "C1:default-settings": default_settings_scene,
    "C1:wire-painter": lambda: wire_painter_scene(False),
    "C1:wire-overlay": lambda: wire_painter_scene(True),
    "cube:default": cube_default_scene,
    "wire-grid:far-first": lambda: scenegen.wire_grid_scene(True),
    "wire-grid:near-first": lambda: scenegen.wire_grid_scene(False),
    "C1:spot-gouraud": spot_scene,
    "C1:spot-flat-zbuf": lambda: spot_scene(True, True),
    "needles": needle_scene,
    "C1:persp": persp_scene,
    "C1:zbuf": zbuf_scene,
    "C1:zbuf-blend": lambda: zbuf_scene("blend", 19),
    "C1:zbuf-gouraud": lambda: zbuf_scene("gouraud", 23),
    "C1:blend5": blend5_scene,
    "C1:zbuf-blend5": lambda: blend5_scene(True),
    "C1": lambda: scenegen.make_scene("C1"),
    "C1:gouraud": lambda: scenegen.make_scene("C1", variant="gouraud"),
    "C1:blend": lambda: scenegen.make_scene("C1", variant="blend"),
    "C1:float": lambda: scenegen.make_scene("C1", variant="float"),
    "cube": cube_scene,
    "fog-flat-point-nocull": fog_scene,
    "C2": lambda: scenegen.make_scene("C2"),
    "C2:blend": lambda: scenegen.make_scene("C2", variant="blend"),
    "C3:100k": lambda: scenegen.make_scene("C3", n_tris=100_000),
    "C5:20k": lambda: scenegen.make_scene("C5", n_tris=20_000),
    "C3": lambda: scenegen.make_scene("C3"),       # BASELINE configs[2] at full size (1 M triangles @ 2560x1920)
    "C5": lambda: scenegen.make_scene("C5"),       # BASELINE configs[4] at full size (65.5 M fragments)
    "C3:blend": lambda: scenegen.make_scene("C3", variant="blend"),   # C3 with 10 % of its faces in the transparent pass (bench.py `configs`)
}


# ---- real content: frames of the reference's OWN sample assets -- the five OBJ meshes as the OBJ importer's preview
# submits them (shared vertices, computed normals, RasterSettings::default()) and rooms of the sample levels with their textures, UVs,
# vertex colours, ambient and fog (RasterSettings::game() / painter's).  The scenes are DATA: tools/make_real_scenes.py laid them out in
# the build container from /root/reference/assets following the reference's producers, and committed them as .b32scene files; here
# they are only read back.
REAL_DIR = os.path.join(OUT, "scenes", "real")


def real_scene(name):
    from bonnie32_amd import scenefile
    sc = scenefile.read_scene(os.path.join(REAL_DIR, name + ".b32scene"))
    sc.name = "real:" + name
    side = os.path.join(REAL_DIR, name + ".indexed.npz")       # (asset parts: the reference's 4-bit index bytes + palette beside the expanded texels)
    if os.path.exists(side):
        z = np.load(side)
        sc.indexed_textures = [b32.IndexedTexture(int(z["width"]), int(z["height"]), z["indices"], z["clut"], int(z["blend_mode"]))]
    return sc


REAL = sorted(json.load(open(os.path.join(REAL_DIR, "manifest.json")))) if os.path.exists(os.path.join(REAL_DIR, "manifest.json")) else []
for _n in REAL:
    SCENES["real:" + _n] = (lambda n=_n: real_scene(n))


def rgba_scene(name="C1", stp_blend=0, seed=61, variant="bench", settings=None, alpha_every=0, bbox_px=200.0, **kw):
    """A scenegen scene with its RGB555 atlas widened to the 8-bit path's Texture (per-texel blend modes on STP texels)."""
    sc = scenegen.make_scene(name, seed=seed, variant=variant, bbox_px=bbox_px, **kw)
    sc.textures8 = [b32.Texture.from_texture15(t, stp_blend) for t in sc.textures]
    if stp_blend:                                  # mix all four PS1 modes over the STP texels, keyed by texel position
        for t in sc.textures8:
            stp = t.pixels[:, 3] == stp_blend
            t.pixels[stp, 3] = 1 + (np.arange(len(t.pixels))[stp] % 4)
    if settings is not None:
        sc.settings = settings
    if alpha_every:
        sc.faces["editor_alpha"][::alpha_every] = 140
        sc.faces["editor_alpha"][1::alpha_every * 3] = 0
    sc.settings.use_rgb555 = False
    return sc


def cube8_scene():
    sc = cube_scene()
    sc.textures8 = [b32.Texture.checkerboard(32, 32, (255, 255, 255, 0), (120, 120, 120, 0))]
    sc.settings = b32.RasterSettings(use_rgb555=False)
    return sc


def zb_settings(**kw):
    return b32.RasterSettings(use_rgb555=False, **kw)


SCENES8 = {
    "8:C1": lambda: rgba_scene(),
    "8:C1-gouraud": lambda: rgba_scene(variant="gouraud", seed=62),
    "8:C1-blend": lambda: rgba_scene(stp_blend=1, variant="blend", seed=63),
    "8:C1-alpha": lambda: rgba_scene(stp_blend=1, variant="blend", seed=64, alpha_every=5),
    "8:C1-default": lambda: rgba_scene(variant="gouraud", seed=65, settings=zb_settings()),
    "8:C1-zbuf-blend-alpha": lambda: rgba_scene(stp_blend=1, variant="blend", seed=66, alpha_every=4, settings=zb_settings(backface_wireframe=False)),
    "8:C1-persp-float": lambda: rgba_scene(variant="float", seed=67, settings=zb_settings(affine_textures=False, use_fixed_point=False, use_zbuffer=False, backface_wireframe=False, shading=1)),
    "8:C1-xray-zbuf": lambda: rgba_scene(stp_blend=1, variant="blend", seed=68, settings=zb_settings(xray_mode=True)),
    "8:C1-ortho-overlay": lambda: rgba_scene(seed=69, bbox_px=900.0, settings=zb_settings(ortho_projection=(0.05, 10.0, -15.0), wireframe_overlay=True)),
    "8:C1-spot": lambda: rgba_scene(variant="gouraud", seed=72, settings=zb_settings(lights=spot_lights(), ambient=0.2, backface_wireframe=False)),
    "8:cube": cube8_scene,
    "8:C2": lambda: rgba_scene("C2", seed=70, bbox_px=None),
}


def render8(sc):
    fb = O.Framebuffer(sc.width, sc.height)
    fb.clear(sc.clear_color)
    rc, tm, d = O.render_mesh(fb, sc.vertices, sc.faces, sc.textures8, sc.camera, sc.settings, dump=True)
    assert rc == 0
    return fb, tm, d


def render(sc):
    fb = O.Framebuffer(sc.width, sc.height)
    fb.clear(sc.clear_color)
    rc, tm, d = O.render_mesh_15(fb, sc.vertices, sc.faces, sc.textures, sc.camera, sc.settings, sc.fog, dump=True)
    assert rc == 0
    return fb, tm, d


def main():
    hashes = {}
    for name, mk in SCENES.items():
        sc = mk()
        fb, tm, d = render(sc)
        hashes[name] = {"sha256": hashlib.sha256(fb.pixels).hexdigest(), "zbuffer_sha256": hashlib.sha256(fb.zbuffer.tobytes()).hexdigest(),
                        "triangles_drawn": tm.triangles_drawn,
                        "fragments": tm.fragments, "width": sc.width, "height": sc.height,
                        "draw_order_sha256": hashlib.sha256(d["draw_order"].tobytes()).hexdigest(),
                        "scene_sha256": hashlib.sha256(sc.vertices.tobytes() + sc.faces.tobytes() + b"".join(t.pixels.tobytes() for t in sc.textures[:1])).hexdigest()}
        if name == "C1":
            np.savez_compressed(os.path.join(OUT, "c1_frame.npz"), rgba=fb.pixels, sx=d["sx"], sy=d["sy"],
                                sz_bits=d["sz"].view(np.uint32), draw_order=d["draw_order"])
        if name == "cube":
            np.savez_compressed(os.path.join(OUT, "cube_frame.npz"), rgba=fb.pixels, draw_order=d["draw_order"])
        print(name, hashes[name]["sha256"][:16], tm.triangles_drawn, tm.fragments)
    for name, mk in SCENES8.items():
        sc = mk()
        fb, tm, d = render8(sc)
        hashes[name] = {"sha256": hashlib.sha256(fb.pixels).hexdigest(), "zbuffer_sha256": hashlib.sha256(fb.zbuffer.tobytes()).hexdigest(),
                        "triangles_drawn": tm.triangles_drawn, "fragments": tm.fragments, "width": sc.width, "height": sc.height,
                        "draw_order_sha256": hashlib.sha256(d["draw_order"].tobytes()).hexdigest(),
                        "scene_sha256": hashlib.sha256(sc.vertices.tobytes() + sc.faces.tobytes() + sc.textures8[0].pixels.tobytes()).hexdigest()}
        print(name, hashes[name]["sha256"][:16], tm.triangles_drawn, tm.fragments)
    json.dump(hashes, open(os.path.join(OUT, "hashes.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()

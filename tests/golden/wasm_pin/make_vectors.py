"""Frames rendered by THE REFERENCE'S OWN COMPILED CODE -> tests/golden/wasm_pin/*.b32scene (expectation = what that code produced).

/root/reference/docs/bonnie-engine.wasm is a build of the crate older than the source tree beside it (no RGB555 / fixed-point path
yet; README.md here lists what differs).  Its `render_mesh` is the ancestor of today's 8-bit-colour `render_mesh`
(render.rs:1971-2259) and shares with `render_mesh_15` the camera transform, float projection, near / backface culling, the painter's
sort, triangle setup, the inside test, affine UVs, `Texture::sample`, vertex-colour interpolation, modulation, lighting
(`shade_multi_light_color` incl. the wasm32 `acosf`) and the blended stores.  The scenes below stay inside what both versions define
identically, so the CURRENT-source restatement (oracle/b32_oracle.c, 8-bit path) must reproduce the old binary's frames bit for bit.

Runs only where /root/reference and node exist (this container); the fixtures it writes travel, the module does not.
usage: python tests/golden/wasm_pin/make_vectors.py [--check]"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from bonnie32_amd import abi, rtypes as T, scenefile, scenegen     # noqa: E402
import wasm_tools                                                   # noqa: E402

WASM = "/root/reference/docs/bonnie-engine.wasm"
NODE_FLAGS = ["--experimental-wasm-anyref", "--experimental-wasm-bulk-memory"]
EXPORTS = {"pin_render_mesh": "rasterizer6render11render_mesh17h", "pin_fb_new": "rasterizer6render11Framebuffer3new17h",
           "pin_malloc": "Dlmalloc$LT$A$GT$6malloc17h", "pin_acosf": "=acosf"}


class PinScene:
    pass


def make(name, seed, n_tris, width, height, *, bbox=40.0, shading=abi.SHADE_NONE, lights=(), ambient=0.3, cull=True, ntex=2, tex_size=32,
         untextured_every=0, oob_every=0, blends=True, cam_pos=(0.0, 0.0, 0.0), clear=(20, 22, 28), rot=None, ortho=None, wire_front=False, near_faces=True, edge=False, tex_shapes=None):
    """Random triangles in front of an identity camera.  Two things differ between the old build and today's source in painter's mode,
    and the depths are drawn so that neither can show: (1) the old build sorts by the LARGEST camera-space z of a face, today's source by
    the mean of z + 5 (render.rs:2155-2160 with math.rs:133); (2) the old build's stores still test and write the z-buffer after the sort
    (linear z), today's painter's mode stores unconditionally (render.rs:1413-1421).  Every face gets a depth slot of its own, 2 units
    wide (z = 2 * slot + three offsets in [0, 1.5], multiples of 1/4): either key orders the faces by slot, all sums are exact, and a
    face drawn later is nearer at every pixel, so the old build's depth test never rejects what the painter's order draws.  (Equal keys
    -- the stability of the sort -- can therefore not be pinned by this module; the draw-order tests cover them.)"""
    rng = np.random.default_rng(seed)
    v = T.make_vertices(3 * n_tris)
    f = T.make_faces(n_tris)
    vs = min(width, height) / 2 * 0.75
    slot = 20 + rng.permutation(n_tris)
    offs = rng.integers(0, 7, (n_tris, 3)) * 0.25
    cz = 2.0 * slot
    cx = (rng.uniform(-0.1, 1.1, n_tris) * width - width / 2) / vs * (cz + 5) / 4
    cy = (rng.uniform(-0.1, 1.1, n_tris) * height - height / 2) / vs * (cz + 5) / 4
    r = bbox / vs * (cz + 5) / 4 / 2
    off = rng.uniform(-1, 1, (n_tris, 3, 3))
    if edge:                                               # hostile geometry: what the saturating casts, the area reject and the walk must survive
        kind = rng.random(n_tris)
        off[kind < 0.05] *= 400.0                          # screen coordinates far outside the frame (bounding box clamps, `as usize` of negatives)
        needle = (kind >= 0.05) & (kind < 0.09)
        off[needle, 1] = off[needle, 0] + rng.uniform(-1e-3, 1e-3, (needle.sum(), 3))
        flat = (kind >= 0.09) & (kind < 0.11)
        off[flat, 2] = off[flat, 1] = off[flat, 0]          # zero area
    pos = np.stack([cx, cy, cz], 1)[:, None, :] + off * r[:, None, None]
    pos[..., 2] = cz[:, None] + offs
    near = (rng.random(n_tris) < 0.02) & near_faces                       # a few faces through the near plane (cam z <= 0.1 rejects, render.rs:2053)
    pos[near, 0, 2] = np.round(rng.uniform(-3.0, 0.0, near.sum()) * 4) / 4
    if ortho is not None:                                  # project_ortho (math.rs:140-148): x = (cam.x - cx) * zoom + w/2, y flipped
        zoom, ocx, ocy = ortho
        pos[..., 0] = (pos[..., 0] * 4 / (cz[:, None] + 5) * vs) / zoom + ocx
        pos[..., 1] = -(pos[..., 1] * 4 / (cz[:, None] + 5) * vs) / zoom + ocy
    basis = np.eye(3)
    if rot is not None:                                    # camera basis = rows of a rotation; the mesh is placed in camera space
        ax, ay = rot
        rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
        ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
        basis = (rx @ ry).astype(np.float32).astype(np.float64)
        pos = pos @ basis                                   # world = sum_k cam_k * basis_k  (basis orthonormal up to f32 rounding)
    v["pos"] = pos.reshape(-1, 3).astype(np.float32) + np.float32(cam_pos)
    v["uv"] = rng.uniform(-1.0, 2.0, (3 * n_tris, 2)).astype(np.float32)
    if edge:                                               # Texture::sample with NaN / infinite / huge / exactly integral coordinates
        k2 = rng.random(3 * n_tris)
        v["uv"][k2 < 0.01, 0] = np.nan
        v["uv"][(k2 >= 0.01) & (k2 < 0.02), 1] = np.inf
        v["uv"][(k2 >= 0.02) & (k2 < 0.03), 0] = -np.inf
        big = (k2 >= 0.03) & (k2 < 0.06)
        v["uv"][big] *= np.float32(3.0e8)
        whole = (k2 >= 0.06) & (k2 < 0.10)
        v["uv"][whole] = np.round(v["uv"][whole])
    nrm = rng.normal(size=(3 * n_tris, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    v["normal"] = nrm.astype(np.float32)
    v["r"], v["g"], v["b"] = (rng.integers(0, 256, 3 * n_tris, dtype=np.uint8) for _ in range(3))
    v["blend"] = 0
    f["v"] = np.arange(3 * n_tris, dtype=np.uint32).reshape(-1, 3)
    f["texture_id"] = rng.integers(0, max(ntex, 1), n_tris)
    if untextured_every:
        f["texture_id"][::untextured_every] = 0xFFFFFFFF
    if oob_every:
        f["texture_id"][1::oob_every] = ntex + 3             # textures.get(id) -> None -> untextured (render.rs:2174-2176)
    f["black_transparent"] = 0; f["blend_mode"] = 0; f["editor_alpha"] = 255
    texs = []
    for t in range(ntex):
        tw, th = tex_shapes[t] if tex_shapes else (tex_size, tex_size)
        px = rng.integers(0, 256, (tw * th, 4), dtype=np.uint8)
        # per-texel blend byte: texture 0 opaque with some Erase texels (skipped, Color::is_transparent), the others every mode
        px[:, 3] = np.where(rng.random(tw * th) < 0.1, T.ERASE, 0) if (t == 0 or not blends) else rng.integers(0, 6, tw * th)
        texs.append(T.Texture(tw, th, px, T.OPAQUE, f"t{t}"))
    sc = PinScene()
    sc.name = name; sc.width, sc.height = width, height
    sc.vertices, sc.faces = v, f
    sc.textures, sc.textures8, sc.indexed_textures = [], texs, []
    sc.camera = T.Camera(position=cam_pos, basis_x=tuple(map(float, basis[0])), basis_y=tuple(map(float, basis[1])), basis_z=tuple(map(float, basis[2])))
    sc.settings = T.RasterSettings(affine_textures=True, use_zbuffer=False, shading=shading, backface_cull=cull, backface_wireframe=False,
                                   lights=list(lights), ambient=ambient, dithering=False, wireframe_overlay=wire_front, ortho_projection=ortho, use_rgb555=False,
                                   use_fixed_point=False, xray_mode=False)
    sc.clear_color = T.Color(*clear)
    sc.fog = None
    return sc


def scenes():
    L = T.Light
    warm = L.directional((-1.0, -1.0, -1.0), 0.7)
    blue = L.directional((0.3, -0.5, 1.0), 0.9); blue.color = T.Color(80, 120, 255)
    point = L.point((30.0, -20.0, 300.0), 900.0, 1.4); point.color = T.Color(255, 200, 120)
    spot = L.spot((0.0, 0.0, 0.0), (0.05, -0.02, 1.0), 0.6, 2500.0, 1.8)
    off = L.directional((0.0, 0.0, 1.0), 5.0); off.enabled = False
    return [
        make("plain_64x48", 11, 60, 64, 48, bbox=24.0),
        make("plain_320x240", 12, 2000, 320, 240, bbox=40.0, untextured_every=7, oob_every=11),
        # (no scene with backface_cull = false: today's source swaps v2 / v3 of a rendered back face, render.rs:2085-2110, the old
        # build keeps the order -- same pixels up to the rounding of the swapped sums, i.e. not bit for bit)
        make("medium_160x120", 13, 500, 160, 120, bbox=50.0, ntex=3),
        make("flat_lights_320x240", 14, 1200, 320, 240, shading=abi.SHADE_FLAT, lights=[warm, blue, off], ambient=0.25, untextured_every=5),
        make("gouraud_lights_320x240", 15, 1200, 320, 240, shading=abi.SHADE_GOURAUD, lights=[warm, point], ambient=0.2),
        make("gouraud_spot_acos_320x240", 16, 1500, 320, 240, shading=abi.SHADE_GOURAUD, lights=[spot, point], ambient=0.1, bbox=60.0),
        make("camera_offset_256x256", 17, 800, 256, 256, cam_pos=(12.5, -7.25, -30.0), bbox=70.0),
        make("opaque_big_tris_320x240", 18, 300, 320, 240, bbox=220.0, blends=False, ntex=1, tex_size=64),
        make("rotated_camera_320x240", 19, 1500, 320, 240, rot=(0.31, -0.47), cam_pos=(100.0, 50.0, -20.0), shading=abi.SHADE_GOURAUD, lights=[warm], bbox=50.0),
        # (today's source skips the near-plane test in ortho views, render.rs:2052; the old build does not: no face behind the plane)
        make("ortho_320x240", 20, 600, 320, 240, ortho=(2.5, 10.0, -4.0), bbox=50.0, near_faces=False),
        make("wire_overlay_160x120", 22, 60, 160, 120, wire_front=True, bbox=60.0),
        make("hostile_320x240", 23, 2500, 320, 240, edge=True, ntex=4, tex_shapes=[(37, 19), (1, 1), (64, 8), (5, 128)], bbox=45.0, untextured_every=13),
        make("small_tris_512x384", 24, 12000, 512, 384, bbox=7.0, ntex=3, tex_shapes=[(16, 16), (128, 128), (3, 7)]),
        make("hostile_lit_256x192", 25, 1500, 256, 192, edge=True, shading=abi.SHADE_GOURAUD, lights=[warm, spot], ntex=2, tex_shapes=[(9, 9), (200, 3)]),
    ]


def run_reference(sc, workdir, patched):
    p = os.path.join(workdir, sc.name + ".b32scene")
    scenefile.write_scene(p, sc, fmt8=True)
    out = os.path.join(workdir, sc.name + ".rgba")
    r = subprocess.run(["node", *NODE_FLAGS, os.path.join(HERE, "wasm_driver.js"), patched, p, out], capture_output=True, text=True, timeout=600)
    if r.returncode:
        raise RuntimeError(f"{sc.name}: {r.stderr[-2000:]}")
    info = json.loads(r.stdout.strip().splitlines()[-1])
    return np.fromfile(out, np.uint8), info


def main():
    check = "--check" in sys.argv
    raw = open(WASM, "rb").read()
    patched_bytes, idx = wasm_tools.add_exports(raw, EXPORTS)
    manifest = {"module": "docs/bonnie-engine.wasm", "module_sha256": hashlib.sha256(raw).hexdigest(), "module_bytes": len(raw),
                "exported_for_the_pin": idx, "node": subprocess.run(["node", "--version"], capture_output=True, text=True).stdout.strip(),
                "node_flags": NODE_FLAGS, "scenes": {}}
    with tempfile.TemporaryDirectory() as wd:
        patched = os.path.join(wd, "patched.wasm")
        open(patched, "wb").write(patched_bytes)
        for sc in scenes():
            frame, info = run_reference(sc, wd, patched)
            sha = hashlib.sha256(frame.tobytes()).hexdigest()
            drawn = int(info["timings_words"][5]) if len(info["timings_words"]) > 5 else 0
            manifest["scenes"][sc.name] = {"sha256": sha, "width": sc.width, "height": sc.height, "triangles": len(sc.faces),
                                           "changed_pixels": int((frame.reshape(-1, 4) != np.array([*[sc.clear_color.r, sc.clear_color.g, sc.clear_color.b], 255], np.uint8)).any(1).sum())}
            dst = os.path.join(HERE, sc.name + ".b32scene")
            if check:
                old = scenefile.read_scene(dst)
                assert old.expect["sha256"] == sha, sc.name
                continue
            scenefile.write_scene(dst, sc, fmt8=True, expect={"triangles_drawn": 0, "fragments": 0, "sha256": sha, "zbuffer_sha256": "00" * 32})
            if sc.width * sc.height <= 64 * 48:
                np.save(os.path.join(HERE, sc.name + ".frame.npy"), frame.reshape(sc.height, sc.width, 4))
            print(sc.name, sha[:16], manifest["scenes"][sc.name]["changed_pixels"], "px drawn", info)
        # f32::acos of the wasm32 target: the module's own acosf over special cases, the algorithm's branch points and random arguments
        rng = np.random.default_rng(7)
        xs = np.concatenate([np.float32([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 1e-9, -1e-9, 2.0 ** -26, 2.0 ** -27, 1.0000001, -1.0000001, 2.0, -2.0,
                                         np.inf, -np.inf, np.nan, 0.49999997, 0.50000006, -0.49999997, -0.50000006, 0.99999994, -0.99999994]),
                             rng.uniform(-1, 1, 6000).astype(np.float32), (1 - np.abs(rng.normal(0, 1e-3, 1000))).astype(np.float32),
                             (-1 + np.abs(rng.normal(0, 1e-3, 1000))).astype(np.float32), (0.5 + rng.normal(0, 1e-4, 500)).astype(np.float32),
                             (-0.5 + rng.normal(0, 1e-4, 500)).astype(np.float32), rng.normal(0, 1e-6, 500).astype(np.float32)]).astype(np.float32)
        xs.tofile(os.path.join(wd, "acos_in.f32"))
        r = subprocess.run(["node", *NODE_FLAGS, os.path.join(HERE, "wasm_driver.js"), patched, "--acosf", os.path.join(wd, "acos_in.f32"),
                            os.path.join(wd, "acos_out.f32")], capture_output=True, text=True, timeout=600)
        if r.returncode:
            raise RuntimeError(r.stderr[-2000:])
        ys = np.fromfile(os.path.join(wd, "acos_out.f32"), np.float32)
        if check:
            old = np.load(os.path.join(HERE, "acosf_kat.npz"))
            assert np.array_equal(old["x_bits"], xs.view(np.uint32)) and np.array_equal(old["acos_bits"], ys.view(np.uint32))
        else:
            np.savez_compressed(os.path.join(HERE, "acosf_kat.npz"), x_bits=xs.view(np.uint32), acos_bits=ys.view(np.uint32))
        manifest["acosf_kat"] = {"n": int(len(xs)), "sha256": hashlib.sha256(ys.tobytes()).hexdigest()}
    if not check:
        json.dump(manifest, open(os.path.join(HERE, "manifest.json"), "w"), indent=1)
    print("ok")


if __name__ == "__main__":
    main()

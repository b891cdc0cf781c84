"""Real-content parity fixtures from the reference's OWN assets.  BUILD CONTAINER ONLY: it reads /root/reference/assets
and writes data -- `.b32scene` files (inputs of one render_mesh_15 call + what the CPU oracle draws from them) under
tests/golden/scenes/real/.  Only those files travel; nothing of the reference's text is copied.

What it lays out, following the reference's own producers (each function cites the lines it follows):
  * the five sample meshes assets/samples/meshes/*.obj exactly as the OBJ importer's preview submits them: ObjImporter::parse
    (modeler/obj_import.rs:22-203: one vertex per (v, vt, vn) triple, fan triangulation with v1 / v2 swapped), the import scale
    (obj_importer.rs:231, main.rs:1126-1128), compute_face_normals (obj_import.rs:468-503), EditableMesh::to_render_data
    (mesh_editor.rs:1587-1618: shared vertices, texture_id None), the orbit camera of set_preview / draw_orbit_preview
    (obj_importer.rs:265-312, 789-809) and its call `render_mesh_15(fb, &vertices, &faces, &[], &camera, &RasterSettings::default(), None)`
    into a 640x480 framebuffer cleared to (25, 25, 35) (obj_importer.rs:812-840);
  * rooms of the sample levels assets/samples/levels/*.ron (brotli, decoded by node's zlib): Room::to_render_data_with_textures
    (world/geometry.rs:2839-3400: floors / ceilings split in two triangles, cardinal and diagonal walls, world-aligned and projected UVs,
    front / back / both normal modes), the level's textures loaded as Texture::from_file + to_15 (rasterizer/types.rs:1080-1107,
    1267-1283: alpha 0 -> 0x0000, else r >> 3 | g >> 3 | b >> 3) and resolved BY NAME (game/renderer.rs:104-112), per-room ambient and fog
    (scene.rs:211-229, 264-276), RasterSettings::game() (types.rs:1455-1460), clear colour (20, 22, 28) (game/renderer.rs:95).
    The camera of a room scene is this script's choice (the reference's is the player's): it stands inside the room.

usage: python tools/make_real_scenes.py          (writes the files + manifest; then python tests/golden/make_golden.py for hashes.json)
"""
import json
import math
import os
import re
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bonnie32_amd as b32                       # noqa: E402
from bonnie32_amd import scenefile, scenegen     # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden", "scenes", "real")
f32 = np.float32
SECTOR_SIZE = f32(1024.0)                        # world/geometry.rs:10


# ------------------------------------------------------------------ Vec3 in f32 with the reference's expression order (math.rs:23-49)
def v3(x, y, z):
    return np.array([x, y, z], f32)


def dot(a, b):
    return f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))


def cross(a, b):
    return v3(f32(a[1] * b[2]) - f32(a[2] * b[1]), f32(a[2] * b[0]) - f32(a[0] * b[2]), f32(a[0] * b[1]) - f32(a[1] * b[0]))


def normalize(a):
    l = f32(np.sqrt(dot(a, a)))
    if l == 0.0:
        return v3(0, 0, 0)
    return v3(a[0] / l, a[1] / l, a[2] / l)


def camera_from_rotation(position, rotation_x, rotation_y):
    """Camera::update_basis (camera.rs:76-91); sin / cos through f32 like the reference's f32::sin / cos (libm: the basis is INPUT data)."""
    rx, ry = f32(rotation_x), f32(rotation_y)
    bz = v3(f32(np.cos(rx)) * f32(np.sin(ry)), -f32(np.sin(rx)), f32(np.cos(rx)) * f32(np.cos(ry)))
    bx = normalize(cross(v3(0.0, -1.0, 0.0), bz))
    by = cross(bz, bx)
    return b32.Camera(tuple(float(v) for v in position), tuple(float(v) for v in bx), tuple(float(v) for v in by), tuple(float(v) for v in bz))


# ------------------------------------------------------------------ OBJ preview
def parse_obj(text):
    """ObjImporter::parse, obj_import.rs:22-203.  Returns (positions/uv/normal per vertex, triangles)."""
    pos, tcs, nrm = [], [], []
    verts, cache, faces = [], {}, []

    def index(s, count):
        i = int(s)
        assert i != 0
        return i - 1 if i > 0 else count + i

    for line in text.splitlines():
        parts = line.strip().split()
        if not parts or parts[0].startswith("#"):
            continue
        if parts[0] == "v":
            pos.append([f32(p) for p in parts[1:4]])
        elif parts[0] == "vt":
            tcs.append([f32(p) for p in parts[1:3]])
        elif parts[0] == "vn":
            nrm.append([f32(p) for p in parts[1:4]])
        elif parts[0] == "f":
            fv = []
            for spec in parts[1:]:
                sp = spec.split("/")
                key = (index(sp[0], len(pos)), index(sp[1], len(tcs)) if len(sp) > 1 and sp[1] else None,
                       index(sp[2], len(nrm)) if len(sp) > 2 and sp[2] else None)
                if key not in cache:
                    cache[key] = len(verts)
                    verts.append((pos[key[0]], tcs[key[1]] if key[1] is not None else [f32(0), f32(0)],
                                  nrm[key[2]] if key[2] is not None else [f32(0), f32(0), f32(0)]))
                fv.append(cache[key])
            for i in range(1, len(fv) - 1):
                faces.append((fv[0], fv[i + 1], fv[i]))         # (v1 / v2 swapped: OBJ is CCW, the rasterizer wants CW)
    return verts, faces


def obj_preview_scene(name, settings, scale=1024.0, yaw=0.8, pitch=0.3, width=640, height=480):
    verts, tris = parse_obj(open(os.path.join(REF, "assets", "samples", "meshes", name + ".obj")).read())
    P = np.array([v[0] for v in verts], f32) * f32(scale)       # main.rs:1126-1128
    UV = np.array([v[1] for v in verts], f32)
    N = np.array([v[2] for v in verts], f32)
    # compute_face_normals, obj_import.rs:468-503: a vertex without a normal takes the normal of the FIRST face that names it
    for (a, b, c) in tris:
        n = normalize(cross(P[b] - P[a], P[c] - P[a]))
        for vi in (a, b, c):
            if N[vi][0] == 0.0 and N[vi][1] == 0.0 and N[vi][2] == 0.0:
                N[vi] = n
    # update_preview_camera, obj_importer.rs:281-312
    mn, mx = P.min(axis=0), P.max(axis=0)
    center = v3((mn[0] + mx[0]) / f32(2), (mn[1] + mx[1]) / f32(2), (mn[2] + mx[2]) / f32(2))
    size = mx - mn
    diag = f32(np.sqrt(f32(f32(size[0] * size[0]) + f32(size[1] * size[1])) + f32(size[2] * size[2])))
    dist = f32(max(diag, f32(2048.0))) * f32(2.0)
    # draw_orbit_preview, obj_importer.rs:789-809
    cp, sp_, cy, sy = f32(np.cos(f32(pitch))), f32(np.sin(f32(pitch))), f32(np.cos(f32(yaw))), f32(np.sin(f32(yaw)))
    cam_pos = center + v3(f32(dist * cp) * sy, dist * sp_, f32(dist * cp) * cy)
    d = center - cam_pos
    n = d * f32(f32(1.0) / f32(np.sqrt(dot(d, d))))
    cam = camera_from_rotation(cam_pos, f32(np.arcsin(-n[1])), f32(np.arctan2(n[0], n[2])))
    v = b32.rtypes.make_vertices(len(P))
    v["pos"] = P; v["uv"] = UV; v["normal"] = N
    v["r"] = v["g"] = v["b"] = 128; v["blend"] = b32.abi.OPAQUE                      # Color::NEUTRAL, types.rs:771
    f = b32.rtypes.make_faces(len(tris))
    f["v"] = np.array(tris, np.uint32)
    f["texture_id"] = b32.abi.NO_TEXTURE; f["black_transparent"] = 1; f["blend_mode"] = b32.abi.OPAQUE; f["editor_alpha"] = 255
    return scenegen.Scene("real:" + name, width, height, v, f, [], [], cam, settings, clear_color=b32.Color(25, 25, 35))


# ------------------------------------------------------------------ RON (the subset serde writes for a Level)
TOKEN = re.compile(r'\s*(?:("(?:[^"\\]|\\.)*")|([A-Za-z_][A-Za-z0-9_]*)|(-?(?:\d+\.?\d*(?:[eE][-+]?\d+)?|inf|NaN))|(.))')


def parse_ron(text):
    toks = [m.groups() for m in TOKEN.finditer(text) if any(g is not None for g in m.groups())]
    toks = [t for t in toks if not (t[3] is not None and t[3].isspace())]
    pos = [0]

    def peek(k=0):
        return toks[pos[0] + k] if pos[0] + k < len(toks) else (None, None, None, None)

    def eat(ch=None):
        t = toks[pos[0]]; pos[0] += 1
        if ch is not None:
            assert t[3] == ch, (t, ch, pos[0])
        return t

    def seq(close):
        out = []
        while peek()[3] != close:
            out.append(value())
            if peek()[3] == ",":
                eat(",")
        eat(close)
        return out

    def paren():                                    # after '(' : a struct (name: value, ...) or a tuple
        if peek()[1] is not None and peek(1)[3] == ":":
            d = {}
            while peek()[3] != ")":
                k = eat()[1]; eat(":")
                d[k] = value()
                if peek()[3] == ",":
                    eat(",")
            eat(")")
            return d
        return seq(")")

    def value():
        s, ident, num, ch = eat()
        if s is not None:
            return json.loads(s)
        if num is not None:
            return float(num) if any(c in num for c in ".eEin") else int(num)
        if ident is not None:
            if ident == "true":
                return True
            if ident == "false":
                return False
            if peek()[3] == "(":
                eat("(")
                inner = paren()
                if ident == "Some":
                    return inner[0] if isinstance(inner, list) and len(inner) == 1 else inner
                return {"__variant__": ident, "value": inner}
            return None if ident == "None" else ident
        if ch == "(":
            return paren()
        if ch == "[":
            return seq("]")
        if ch == "{":
            d = {}
            while peek()[3] != "}":
                k = value(); eat(":"); d[k] = value()
                if peek()[3] == ",":
                    eat(",")
            eat("}")
            return d
        raise ValueError((s, ident, num, ch, pos[0]))

    v = value()
    assert pos[0] == len(toks), (pos[0], len(toks))
    return v


def load_level(name):
    src = os.path.join(REF, "assets", "samples", "levels", name + ".ron")
    raw = open(src, "rb").read()
    if raw[:1] not in (b"(", b" ", b"\n", b"\r", b"\t"):       # brotli (mesh_editor.rs:1563-1575 does the same sniffing)
        raw = subprocess.run(["node", "-e", "process.stdout.write(require('zlib').brotliDecompressSync(require('fs').readFileSync(process.argv[1])))", src],
                             capture_output=True, check=True).stdout
    return parse_ron(raw.decode("utf-8"))


# ------------------------------------------------------------------ Room::to_render_data_with_textures (world/geometry.rs:2839-3400)
BLEND = {"Opaque": 0, "Average": 1, "Add": 2, "Subtract": 3, "AddQuarter": 4, "Erase": 5}


def vec2(d):
    return np.array([d["x"], d["y"]], f32)


class Mesh:
    def __init__(self):
        self.v, self.f = [], []

    def vertex(self, pos, uv, normal, col):
        self.v.append((pos, uv, normal, (col["r"], col["g"], col["b"], BLEND[col["blend"]])))

    def face(self, a, b, c, tex, bt, blend):
        self.f.append((a, b, c, tex, 1 if bt else 0, BLEND[blend]))


def room_mesh(room, resolve):
    m = Mesh()
    px, py, pz = f32(room["position"]["x"]), f32(room["position"]["y"]), f32(room["position"]["z"])

    def horizontal(face, bx, bz, gx, gz, is_floor):             # geometry.rs:2906-3049
        h1 = [f32(h) for h in face["heights"]]
        h2 = [f32(h) for h in (face.get("heights_2") or face["heights"])]

        def corners(h):
            return [v3(bx, py + h[0], bz), v3(bx + SECTOR_SIZE, py + h[1], bz), v3(bx + SECTOR_SIZE, py + h[2], bz + SECTOR_SIZE), v3(bx, py + h[3], bz + SECTOR_SIZE)]
        c1, c2 = corners(h1), corners(h2)
        t1, w1 = resolve(face["texture"])
        t2, w2 = resolve(face.get("texture_2") or face["texture"])
        s1, s2 = f32(32.0) / f32(w1), f32(32.0) / f32(w2)

        def default_uv(s):
            uo, vo = f32(gx) * s, f32(gz) * s
            return [np.array([uo, vo], f32), np.array([uo + s, vo], f32), np.array([uo + s, vo + s], f32), np.array([uo, vo + s], f32)]
        uv1 = [vec2(u) for u in face["uv"]] if face.get("uv") else default_uv(s1)
        uv2_src = face.get("uv_2") or face.get("uv")
        uv2 = [vec2(u) for u in uv2_src] if uv2_src else (uv1 if w1 == w2 else default_uv(s2))
        col1 = face["colors"]; col2 = face.get("colors_2") or face["colors"]
        mode = face.get("normal_mode", "Front")
        front, back = mode != "Back", mode != "Front"
        nwse = face.get("split_direction", "NwSe") == "NwSe"
        tri1 = (0, 1, 2) if nwse else (0, 1, 3)
        tri2 = (0, 2, 3) if nwse else (1, 2, 3)

        def fnormal(c):
            e1, e2 = c[1] - c[0], c[3] - c[0]
            return normalize(cross(e2, e1)) if is_floor else normalize(cross(e1, e2))
        n1, n2 = fnormal(c1), fnormal(c2)
        bt, blend = face.get("black_transparent", True), face.get("blend_mode", "Opaque")

        def tri(c, idx, uv, col, n, tex, flip):
            b = len(m.v)
            for k in idx:
                m.vertex(c[k], uv[k], n, col[k])
            if flip:
                m.face(b, b + 2, b + 1, tex, bt, blend)
            else:
                m.face(b, b + 1, b + 2, tex, bt, blend)
        if front:
            tri(c1, tri1, uv1, col1, n1, t1, not is_floor)
        if back:
            tri(c1, tri1, uv1, col1, n1 * f32(-1.0), t1, is_floor)
        if front:
            tri(c2, tri2, uv2, col2, n2, t2, not is_floor)
        if back:
            tri(c2, tri2, uv2, col2, n2 * f32(-1.0), t2, is_floor)

    def wall_uvs(wall, corner_u, s):                            # geometry.rs:3158-3196 / 3288-3322
        base = [vec2(u) for u in wall["uv"]] if wall.get("uv") else [np.array([corner_u[0], s], f32), np.array([corner_u[1], s], f32),
                                                                       np.array([corner_u[2], 0.0], f32), np.array([corner_u[3], 0.0], f32)]
        if wall.get("uv_projection", "Default") == "Projected":
            wh = [py + f32(h) for h in wall["heights"]]
            return [np.array([base[i][0], f32(f32(-wh[i]) / SECTOR_SIZE) * s], f32) for i in range(4)]
        return base

    def quad(wall, corners, normal, uvs, tex):                  # geometry.rs:3198-3224
        mode = wall.get("normal_mode", "Front")
        bt, blend = wall.get("black_transparent", True), wall.get("blend_mode", "Opaque")
        if mode != "Back":
            b = len(m.v)
            for i in range(4):
                m.vertex(corners[i], uvs[i], normal, wall["colors"][i])
            m.face(b, b + 2, b + 1, tex, bt, blend); m.face(b, b + 3, b + 2, tex, bt, blend)
        if mode != "Front":
            b = len(m.v)
            for i in range(4):
                m.vertex(corners[i], uvs[i], normal * f32(-1.0), wall["colors"][i])
            m.face(b, b + 1, b + 2, tex, bt, blend); m.face(b, b + 2, b + 3, tex, bt, blend)

    def cardinal(wall, bx, bz, gx, gz, direction):              # geometry.rs:3051-3157
        h = [py + f32(x) for x in wall["heights"]]
        S = SECTOR_SIZE
        if direction == "North":
            c, n = [v3(bx, h[0], bz), v3(bx + S, h[1], bz), v3(bx + S, h[2], bz), v3(bx, h[3], bz)], v3(0, 0, 1)
        elif direction == "East":
            c, n = [v3(bx + S, h[0], bz), v3(bx + S, h[1], bz + S), v3(bx + S, h[2], bz + S), v3(bx + S, h[3], bz)], v3(-1, 0, 0)
        elif direction == "South":
            c, n = [v3(bx + S, h[0], bz + S), v3(bx, h[1], bz + S), v3(bx, h[2], bz + S), v3(bx + S, h[3], bz + S)], v3(0, 0, -1)
        else:
            c, n = [v3(bx, h[0], bz + S), v3(bx, h[1], bz), v3(bx, h[2], bz), v3(bx, h[3], bz + S)], v3(1, 0, 0)
        tex, w = resolve(wall["texture"])
        s = f32(32.0) / f32(w)
        u = (f32(gx) if direction in ("North", "South") else f32(gz)) * s
        quad(wall, c, n, wall_uvs(wall, [u, u + s, u + s, u], s), tex)

    def diagonal(wall, bx, bz, gx, is_nwse):                    # geometry.rs:3227-3345
        hh = [f32(x) for x in wall["heights"]]
        S = SECTOR_SIZE
        r = f32(f32(1.0) / f32(np.sqrt(f32(2.0))))
        if is_nwse:
            c = [v3(bx + S, py + hh[1], bz + S), v3(bx, py + hh[0], bz), v3(bx, py + hh[3], bz), v3(bx + S, py + hh[2], bz + S)]
            n = v3(r, 0, -r)
        else:
            c = [v3(bx, py + hh[1], bz + S), v3(bx + S, py + hh[0], bz), v3(bx + S, py + hh[3], bz), v3(bx, py + hh[2], bz + S)]
            n = v3(r, 0, r)
        tex, w = resolve(wall["texture"])
        s = f32(32.0) / f32(w)
        u = f32(gx) * s
        quad(wall, c, n, wall_uvs(wall, [u, u + s, u + s, u], s), tex)

    for gx, col in enumerate(room["sectors"]):                  # iter_sectors, geometry.rs:2828-2835
        for gz, sec in enumerate(col):
            if sec is None:
                continue
            bx, bz = px + f32(gx) * SECTOR_SIZE, pz + f32(gz) * SECTOR_SIZE
            if sec.get("floor"):
                horizontal(sec["floor"], bx, bz, gx, gz, True)
            if sec.get("ceiling"):
                horizontal(sec["ceiling"], bx, bz, gx, gz, False)
            for d, key in (("North", "walls_north"), ("East", "walls_east"), ("South", "walls_south"), ("West", "walls_west")):
                for wall in sec.get(key, []):
                    cardinal(wall, bx, bz, gx, gz, d)
            for wall in sec.get("walls_nwse", []):
                diagonal(wall, bx, bz, gx, True)
            for wall in sec.get("walls_nesw", []):
                diagonal(wall, bx, bz, gx, False)
    return m


def level_textures(level):
    """Every texture the level names, as the game loads them (Texture::from_file + to_15), in (pack, name) order; resolved by NAME."""
    from PIL import Image
    refs = set()

    def walk(o):
        if isinstance(o, dict):
            if set(o.keys()) >= {"pack", "name"} and isinstance(o.get("name"), str):
                refs.add((o["pack"], o["name"]))
            for v in o.values():
                walk(v)
        elif isinstance(o, list):
            for v in o:
                walk(v)
    walk(level["rooms"])
    texs, names = [], []
    for pack, name in sorted(refs):
        p = next((q for q in (os.path.join(REF, "assets", "samples", "texture-packs", pack, name + ext) for ext in (".png", ".PNG")) if os.path.exists(q)), None)
        if not name or p is None:
            continue
        rgba = np.asarray(Image.open(p).convert("RGBA"), np.uint8)
        h, w = rgba.shape[:2]
        c15 = ((rgba[..., 0].astype(np.uint16) >> 3) << 10) | ((rgba[..., 1].astype(np.uint16) >> 3) << 5) | (rgba[..., 2].astype(np.uint16) >> 3)
        c15[rgba[..., 3] == 0] = 0                                                    # types.rs:1094-1096, 1267-1275
        texs.append(b32.Texture15(w, h, c15.reshape(-1).astype(np.uint16), b32.abi.OPAQUE)); names.append(name)
    return texs, names


def room_scene(level_name, room_idx, cam_pos, rot_x, rot_y, settings_of, width=320, height=240, only_used=True):
    level = load_level(level_name)
    texs, names = level_textures(level)
    room = level["rooms"][room_idx]

    def resolve(ref):                                           # game/renderer.rs:104-112
        if not ref or not ref.get("name"):                      # TextureRef::is_valid
            return 0, 64
        for i, n in enumerate(names):
            if n == ref["name"]:
                return i, texs[i].width
        return 0, 64                                            # `.unwrap_or((0, 64))`, geometry.rs:2941
    m = room_mesh(room, resolve)
    if only_used:                                               # keep the file small: only the textures this room's faces name, ids remapped
        used = sorted({f[3] for f in m.f if f[3] < len(texs)})
        remap = {t: i for i, t in enumerate(used)}
        for t in {f[3] for f in m.f if f[3] >= len(texs)}:       # (no such texture: `textures.get(id)` is None, the face is drawn untextured -- render.rs:2554-2556)
            remap[t] = len(used) + (t - len(texs))
        texs = [texs[t] for t in used]
        m.f = [(a, b, c, remap[t], bt, bl) for (a, b, c, t, bt, bl) in m.f]
    v = b32.rtypes.make_vertices(len(m.v))
    v["pos"] = np.array([x[0] for x in m.v], f32); v["uv"] = np.array([x[1] for x in m.v], f32); v["normal"] = np.array([x[2] for x in m.v], f32)
    cols = np.array([x[3] for x in m.v], np.uint8)
    v["r"], v["g"], v["b"], v["blend"] = cols[:, 0], cols[:, 1], cols[:, 2], cols[:, 3]
    f = b32.rtypes.make_faces(len(m.f))
    ff = np.array(m.f, np.int64)
    f["v"] = ff[:, 0:3].astype(np.uint32); f["texture_id"] = ff[:, 3].astype(np.uint32)
    f["black_transparent"] = ff[:, 4]; f["blend_mode"] = ff[:, 5]; f["editor_alpha"] = 255
    st = settings_of()
    st.ambient = float(f32(room.get("ambient", 0.5)))           # scene.rs:211-215
    fog = None
    rf = room.get("fog")
    if rf and rf.get("enabled"):                                # build_room_fog, scene.rs:264-276 (`as u8` saturates)
        r, g, b = (int(min(max(f32(c) * f32(255.0), 0), 255)) for c in rf["color"])
        start, fall, off = f32(rf["start"]), f32(rf.get("falloff", 30000.0)), f32(rf.get("cull_offset", 0.0))
        fog = (float(start), float(fall), float(f32(start + fall) + off), b32.Color(r, g, b))
    cam = camera_from_rotation(cam_pos, rot_x, rot_y)
    sc = scenegen.Scene(f"real:{level_name}-room{room_idx}", width, height, v, f, texs, [], cam, st, clear_color=b32.Color(20, 22, 28))
    sc.fog = fog
    return sc


def inside(level_name, room_idx, gx, gz, eye=610.0):
    """A camera position standing on sector (gx, gz) of the room at the level's camera height (player_settings.camera_height)."""
    room = load_level(level_name)["rooms"][room_idx]
    sec = room["sectors"][gx][gz]
    floor = max(sec["floor"]["heights"]) if sec and sec.get("floor") else 0.0
    p = room["position"]
    return (p["x"] + (gx + 0.5) * 1024.0, p["y"] + floor + eye, p["z"] + (gz + 0.5) * 1024.0)


# ------------------------------------------------------------------ asset mesh parts with the reference's indexed user textures (scene.rs:75-170)
def load_ron_file(path):
    raw = open(path, "rb").read()
    if raw[:1] not in (b"(", b" ", b"\n", b"\r", b"\t"):
        raw = subprocess.run(["node", "-e", "process.stdout.write(require('zlib').brotliDecompressSync(require('fs').readFileSync(process.argv[1])))", path],
                             capture_output=True, check=True).stdout
    return parse_ron(raw.decode("utf-8"))


def user_textures():
    """assets/samples/textures/*.ron: UserTexture { id, width, height, depth, indices (one byte per texel), palette (Color15 words), blend_mode }"""
    out = {}
    d = os.path.join(REF, "assets", "samples", "textures")
    for fn in sorted(os.listdir(d)):
        if fn.endswith(".ron"):
            t = load_ron_file(os.path.join(d, fn))
            out[t["id"]] = t
    return out


def asset_part_scene(part_idx, settings_of, yaw=0.8, pitch=0.3, width=320, height=240, texture_name=None, fog=None):
    """One iteration of render_asset_parts (scene.rs:114-170): EditableMesh::to_render_data_textured (mesh_editor.rs:1623-1653: n-gon faces
    fan-triangulated, texture_id Some(0)), resolve_part_texture (scene.rs:75-104: the UserTexture's indices + palette as IndexedAtlas + Clut),
    `atlas.to_texture15(&clut, ..)` (mesh_editor.rs:669-682, Clut::lookup types.rs:390-397) -- here handed over as the INDEXED texture, the
    expansion being the library's (b32_scene_upload_indexed) -- and per-part back-face settings (scene.rs:134-138).  texture_name: pair the
    part with another of the sample textures (texture_001 / _002 carry STP palette entries; their own blend mode does not reach the
    rasterizer: to_texture15 always says Opaque)."""
    asset = load_ron_file(os.path.join(REF, "assets", "samples", "assets", "asset_003.ron"))
    mesh_comp = next(c for c in asset["components"] if isinstance(c, dict) and c.get("__variant__") == "Mesh")
    part = mesh_comp["value"]["parts"][part_idx]
    texs = user_textures()
    ref = part["texture_ref"]
    tex = texs[ref["value"][0]] if isinstance(ref, dict) and ref.get("__variant__") == "Id" else None
    if texture_name:
        tex = next(t for t in texs.values() if t["name"] == texture_name)
    assert tex is not None
    mv = part["mesh"]["vertices"]
    v = b32.rtypes.make_vertices(len(mv))
    v["pos"] = np.array([[x["pos"]["x"], x["pos"]["y"], x["pos"]["z"]] for x in mv], f32)
    v["uv"] = np.array([[x["uv"]["x"], x["uv"]["y"]] for x in mv], f32)
    v["normal"] = np.array([[x["normal"]["x"], x["normal"]["y"], x["normal"]["z"]] for x in mv], f32)
    v["r"] = [x["color"]["r"] for x in mv]; v["g"] = [x["color"]["g"] for x in mv]; v["b"] = [x["color"]["b"] for x in mv]
    v["blend"] = [BLEND[x["color"]["blend"]] for x in mv]
    tris = []
    for fc in part["mesh"]["faces"]:                            # EditFace::triangulate, mesh_editor.rs:96-110
        vs = fc["vertices"]
        for i in range(1, len(vs) - 1):
            tris.append((vs[0], vs[i], vs[i + 1], fc["texture_id"] if fc.get("texture_id") is not None else 0,
                         1 if fc.get("black_transparent", True) else 0, BLEND[fc.get("blend_mode", "Opaque")]))
    f = b32.rtypes.make_faces(len(tris))
    ff = np.array(tris, np.int64)
    f["v"] = ff[:, 0:3].astype(np.uint32); f["texture_id"] = ff[:, 3].astype(np.uint32)
    f["black_transparent"] = ff[:, 4]; f["blend_mode"] = ff[:, 5]; f["editor_alpha"] = 255
    st = settings_of()
    double_sided = bool(part.get("double_sided", False))
    st.backface_cull = (not double_sided) and st.backface_cull          # scene.rs:134-138
    st.backface_wireframe = (not double_sided) and st.backface_wireframe
    it = b32.IndexedTexture(int(tex["width"]), int(tex["height"]), np.array(tex["indices"], np.uint8), np.array(tex["palette"], np.uint16), b32.abi.OPAQUE)
    P = v["pos"]
    mn, mx = P.min(axis=0), P.max(axis=0)
    center = (mn + mx) / f32(2)
    dist = f32(max(float(np.sqrt(((mx - mn) ** 2).sum())), 2048.0)) * f32(1.2)
    cp, sp_, cy, sy = f32(np.cos(f32(pitch))), f32(np.sin(f32(pitch))), f32(np.cos(f32(yaw))), f32(np.sin(f32(yaw)))
    cam_pos = center + v3(f32(dist * cp) * sy, dist * sp_, f32(dist * cp) * cy)
    d = center - cam_pos
    n = d * f32(f32(1.0) / f32(np.sqrt(dot(d, d))))
    cam = camera_from_rotation(cam_pos, f32(np.arcsin(-n[1])), f32(np.arctan2(n[0], n[2])))
    sc = scenegen.Scene(f"real:asset3-part{part_idx}", width, height, v, f, [it.to_texture15()], [it], cam, st, clear_color=b32.Color(20, 22, 28))
    sc.fog = fog
    return sc


def painter():
    return b32.RasterSettings.benchmark()


SCENES = {
    # the OBJ importer's preview call, verbatim settings (RasterSettings::default(): z-buffer, Gouraud + directional light, back-face wireframe)
    "obj-clockwork": lambda: obj_preview_scene("clockwork", b32.RasterSettings()),
    "obj-crawler": lambda: obj_preview_scene("crawler", b32.RasterSettings()),
    "obj-ghost": lambda: obj_preview_scene("ghost", b32.RasterSettings()),
    "obj-ps1_figure": lambda: obj_preview_scene("ps1_figure", b32.RasterSettings()),
    "obj-warrior": lambda: obj_preview_scene("warrior", b32.RasterSettings()),
    # the same meshes (shared vertices) under the benchmark's painter's settings and under RasterSettings::game(), other orbit angles
    "obj-warrior-painter": lambda: obj_preview_scene("warrior", painter(), yaw=2.4, pitch=-0.2),
    "obj-ghost-game": lambda: obj_preview_scene("ghost", b32.RasterSettings.game(), yaw=-1.1, pitch=0.6, width=320, height=240),
    "obj-clockwork-painter-2560": lambda: obj_preview_scene("clockwork", painter(), yaw=0.3, pitch=0.1, width=2560, height=1920),
    # rooms of the sample levels with their own textures, UVs, vertex colours, ambient and fog; RasterSettings::game() and painter's
    "dungeon-room0-game": lambda: room_scene("Dungeon", 0, inside("Dungeon", 0, 3, 9), 0.15, 0.4, b32.RasterSettings.game),
    "dungeon-room0-painter": lambda: room_scene("Dungeon", 0, inside("Dungeon", 0, 2, 4), -0.1, 2.9, painter),
    "cave-room0-game": lambda: room_scene("Cave", 0, inside("Cave", 0, 2, 2), 0.1, 0.9, b32.RasterSettings.game),
    "cathedral-room0-game-640": lambda: room_scene("Cathedral", 0, inside("Cathedral", 0, 4, 4), 0.05, 0.7, b32.RasterSettings.game, width=640, height=480),
    "sewers-room0-painter": lambda: room_scene("Sewers", 0, inside("Sewers", 0, 1, 1), 0.2, 1.3, painter),
    # the sample asset's mesh parts (quads, fan-triangulated) with the reference's own 4-bit indexed textures + palettes, as render_asset_parts
    # submits them: one call per part, per-part back-face settings; the texture reaches the library as index bytes + CLUT
    "asset3-part0-game": lambda: asset_part_scene(0, b32.RasterSettings.game),
    "asset3-part1-game": lambda: asset_part_scene(1, b32.RasterSettings.game, yaw=2.2, pitch=0.5),
    "asset3-part2-painter": lambda: asset_part_scene(2, painter, yaw=-0.6, pitch=0.2, width=640, height=480),
    "asset3-part0-stp-palette-painter": lambda: asset_part_scene(0, painter, yaw=1.4, pitch=0.4, texture_name="texture_001"),
}


def main():
    from oracle import oracle as O
    import hashlib
    os.makedirs(OUT, exist_ok=True)
    manifest = {}
    for name, mk in SCENES.items():
        sc = mk()
        fb = O.Framebuffer(sc.width, sc.height); fb.clear(sc.clear_color)
        rc, tm, d = O.render_mesh_15(fb, sc.vertices, sc.faces, sc.textures, sc.camera, sc.settings, sc.fog, dump=True)
        assert rc == 0, (name, rc)
        exp = {"triangles_drawn": tm.triangles_drawn, "fragments": tm.fragments, "sha256": hashlib.sha256(fb.pixels).hexdigest(),
               "zbuffer_sha256": hashlib.sha256(fb.zbuffer.tobytes()).hexdigest()}
        fn = name + ".b32scene"
        digest = scenefile.write_scene(os.path.join(OUT, fn), sc, exp)
        if sc.indexed_textures:        # the texture as the reference holds it -- index bytes + palette: a sidecar (the .b32scene carries the expanded Texture15)
            it = sc.indexed_textures[0]
            np.savez_compressed(os.path.join(OUT, name + ".indexed.npz"), width=it.width, height=it.height, indices=it.indices, clut=it.clut, blend_mode=it.blend_mode)
        lit = int((fb.pixels.reshape(-1, 4)[:, :3] != np.array([sc.clear_color.r, sc.clear_color.g, sc.clear_color.b], np.uint8)).any(axis=1).sum())
        manifest[name] = {"file": fn, "file_sha256": digest, "width": sc.width, "height": sc.height, "vertices": int(len(sc.vertices)), "faces": int(len(sc.faces)),
                          "textures": len(sc.textures), "triangles_drawn": tm.triangles_drawn, "fragments": tm.fragments, "pixels_drawn": lit, **{k: exp[k] for k in ("sha256",)}}
        print(f"{name:32s} {len(sc.vertices):6d} v {len(sc.faces):6d} f {len(sc.textures):3d} tex  drawn {tm.triangles_drawn:6d}  fragments {tm.fragments:8d}  pixels {lit:7d}  {os.path.getsize(os.path.join(OUT, fn)) // 1024} KB")
    json.dump(manifest, open(os.path.join(OUT, "manifest.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()

"""np_model.py — second, independent restatement of bonnie-32's `render_mesh_15` in numpy float32 / int64.

TEST INFRASTRUCTURE.  Purpose: the reference is Rust and cannot be built here, and it holds no test that pins a pixel; only an older
compiled build (docs/bonnie-engine.wasm) runs, which pins what the 8-bit and the RGB555 path share (tests/golden/wasm_pin/, this file
reproduces those frames too) and leaves the fixed-point snap and the RGB555 tail "parity unpinned" (SURVEY §8c).  This file is a second reading of the same Rust text, written in a different style
from oracle/b32_oracle.c (whole-bbox vectorised per triangle, `np.add.accumulate` for the incremental edge walk,
vectorised UNR division on int64 arrays, `argsort(kind="stable")` for the painter's order).  tests/ require the two
restatements to agree bit-for-bit on whole frames; a misreading would have to be made identically in both to survive.

Every function cites the reference file:line it follows (paths relative to /root/reference).
Scope: perspective projection (fixed-point or float), painter's and z-buffer mode, affine and perspective-correct textures, shading None/Flat/Gouraud with
directional and point lights, fog, blend modes, editor alpha.
"""
import json
import os

import numpy as np

f32 = np.float32

# Every numeric literal of the algorithm comes from tests/golden/ref_constants.json, which tests/golden/pin_constants.py derives from
# the reference's own text (file:line recorded per entry): this restatement holds no hand-typed copy of them.
_FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "ref_constants.json")
REF = {k: v["value"] for k, v in json.load(open(_FIXTURE)).items()}


def K(name):
    """literal `name` of the reference: f32 for float literals, int otherwise"""
    v = REF[name]
    return f32(v) if isinstance(v, float) else int(v)

I32_MIN, I32_MAX = -(2 ** 31), 2 ** 31 - 1


# ---------------------------------------------------------------- Rust numeric-cast semantics on arrays
def as_i32(x):
    """`f32 as i32`: truncate, saturate, NaN -> 0."""
    x = np.asarray(x, dtype=np.float32)
    with np.errstate(invalid="ignore"):
        t = np.trunc(x.astype(np.float64))
    t = np.where(np.isnan(t), 0.0, t)
    return np.clip(t, I32_MIN, I32_MAX).astype(np.int64)


def as_usize(x):
    x = np.asarray(x, dtype=np.float32)
    t = np.trunc(x.astype(np.float64))
    t = np.where(np.isnan(t), 0.0, t)
    return np.clip(t, 0, 2.0 ** 63).astype(np.int64)


def as_u8(x):
    x = np.asarray(x, dtype=np.float32)
    t = np.trunc(x.astype(np.float64))
    t = np.where(np.isnan(t), 0.0, t)
    return np.clip(t, 0, 255).astype(np.int64)


def wrap32(x):
    """two's-complement wrap of int64 values to i32 (release-mode / wrapping_* arithmetic)."""
    return ((np.asarray(x, dtype=np.int64) + 2 ** 31) % 2 ** 32) - 2 ** 31


# ---------------------------------------------------------------- fixed.rs
UNR_TABLE = np.array([max(0, (K("unr.numerator") // (i + K("unr.index_offset")) + K("unr.round_add")) // K("unr.round_div") - K("unr.subtract"))
                      for i in range(K("unr.entries"))], dtype=np.int64)                                      # :20-31
FRAC = K("fixed.frac_bits")


def fx_from_f32(x):                       # Fixed32::from_f32 :125-127
    return as_i32(np.asarray(x, np.float32) * f32(1 << FRAC))


def fx_mul(a, b):                         # mul_fixed :161-165  (i64 product, arithmetic shift, truncating narrow)
    return wrap32((np.asarray(a, np.int64) * np.asarray(b, np.int64)) >> FRAC)


def fx_div_unr(n, d):                     # div_unr :178-230
    n = np.asarray(n, np.int64); d = np.asarray(d, np.int64)
    out = np.zeros(np.broadcast(n, d).shape, np.int64)
    n, d = np.broadcast_arrays(n, d)
    nz = d != 0
    num = np.abs(n[nz]).astype(np.uint64)
    den = np.abs(d[nz]).astype(np.uint64)
    neg = (n[nz] < 0) != (d[nz] < 0)
    bl = np.zeros(den.shape, np.int64)                  # bit length of den (1..32)
    t = den.copy()
    for s in (16, 8, 4, 2, 1):
        m = t >= (np.uint64(1) << np.uint64(s))
        bl += np.where(m, s, 0)
        t = np.where(m, t >> np.uint64(s), t)
    bl += (t > 0).astype(np.int64)
    z = (32 - bl).astype(np.uint64)
    d16 = ((den << z) >> np.uint64(K("div_unr.d16_shift"))).astype(np.int64)
    idx = np.minimum((d16 - K("div_unr.index_bias")) >> K("div_unr.index_shift"), K("div_unr.index_max"))
    u = UNR_TABLE[idx] + K("div_unr.u_add")
    nr1 = (K("div_unr.nr1_const") - d16 * u) >> K("div_unr.nr1_shift")
    nr2 = ((K("div_unr.nr2_const") + nr1 * u) >> K("div_unr.nr2_shift")).astype(np.uint64)
    raw = num * nr2                                     # < 2^64 for every i32 numerator
    shift = (np.uint64(K("div_unr.shift_base")) - z)
    mag = (raw + (np.uint64(1) << (shift - np.uint64(1)))) >> shift
    mag = np.minimum(mag, np.uint64(I32_MAX)).astype(np.int64)
    out[nz] = np.where(neg, -mag, mag)
    return out


def project_fixed(pos, cam, width, height):
    """fixed::project_fixed :424-441 -> (sx, sy) int64 arrays."""
    p = [fx_from_f32(pos[:, i]) for i in range(3)]
    c = [fx_from_f32(f32(cam.position[i])) for i in range(3)]
    rel = [wrap32(p[i] - c[i]) for i in range(3)]                               # :370 (wrapping_sub :244-246)
    def dot(b):
        bb = [fx_from_f32(f32(b[i])) for i in range(3)]
        return wrap32(wrap32(fx_mul(rel[0], bb[0]) + fx_mul(rel[1], bb[1])) + fx_mul(rel[2], bb[2]))   # :311-313
    cx, cy, cz = dot(cam.basis_x), dot(cam.basis_y), dot(cam.basis_z)
    distance, scale = fx_from_f32(K("project_fixed.distance")), fx_from_f32(K("project_fixed.scale"))              # :396-397
    vs = fx_from_f32(f32(f32(min(width, height)) / K("project_fixed.viewport_div")) * K("project_fixed.viewport_frac"))   # :398
    half_w, half_h = wrap32((width // 2) << FRAC), wrap32((height // 2) << FRAC)   # :399-400
    denom = wrap32(cz + distance)
    absd = np.where(denom < 0, wrap32(-denom), denom)                           # i32::abs (wraps at MIN)
    small = absd < K("project_fixed.denom_guard")                               # :406-408
    safe = np.where(small, 4096, denom)
    px = fx_div_unr(fx_mul(cx, scale), safe)
    py = fx_div_unr(fx_mul(cy, scale), safe)
    sx = wrap32(fx_mul(px, vs) + half_w) >> FRAC
    sy = wrap32(fx_mul(py, vs) + half_h) >> FRAC
    return np.where(small, half_w >> FRAC, sx), np.where(small, half_h >> FRAC, sy)


# ---------------------------------------------------------------- math.rs
def dot3(a, b):                            # Vec3::dot :23-25  (x*ox + y*oy) + z*oz, all f32
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def normalize3(v):                         # :39-49
    v = np.asarray(v, np.float32)
    l = np.sqrt(dot3(v, v))
    if l == 0:
        return np.zeros(3, np.float32)
    return (v / l).astype(np.float32)


def rmin(a, b):
    return np.fmin(a, b)                   # f32::min ignores NaN


def rmax(a, b):
    return np.fmax(a, b)


# ---------------------------------------------------------------- colour helpers (render.rs / types.rs)
DITHER = np.array(REF["dither.matrix"], np.int64)                                                # render.rs:1150-1155


def expand5(v):                            # render.rs:1161-1163
    return ((v << K("expand5.shl")) | (v >> K("expand5.shr"))) & 0xFF


def blend555(front, back, mode):           # render.rs:1093-1145 on [...,3] uint arrays
    f5, b5 = front >> K("blend555.in_shift"), back >> K("blend555.in_shift")
    hi = K("blend555.clamp_hi")
    if mode == 1:
        r = np.minimum((b5 + f5) // K("blend555.average_div"), hi)
    elif mode == 2:
        r = np.minimum(b5 + f5, hi)
    elif mode == 3:
        r = np.maximum(b5 - f5, K("blend555.clamp_lo"))
    elif mode == 4:
        r = np.minimum(b5 + f5 // K("blend555.quarter_div"), hi)
    elif mode == 5:
        r = b5
    else:
        r = f5
    return r << K("blend555.out_shift")


def acosf(x):
    """f32::acos as the `libm` crate / musl acosf.c computes it (see oracle/b32_oracle.c: b32o_acosf), scalar f32."""
    x = f32(x)
    pio2_hi, pio2_lo = f32(1.5707962513e+00), f32(7.5497894159e-08)
    pS0, pS1, pS2, qS1 = f32(1.6666586697e-01), f32(-4.2743422091e-02), f32(-8.6563630030e-03), f32(-7.0662963390e-01)

    def R(z):
        p = f32(z * f32(pS0 + f32(z * f32(pS1 + f32(z * pS2)))))
        q = f32(f32(1.0) + f32(z * qS1))
        return f32(p / q)
    hx = int(np.float32(x).view(np.uint32))
    ix = hx & 0x7FFFFFFF
    if ix >= 0x3F800000:
        if ix == 0x3F800000:
            return f32(f32(2.0) * pio2_hi) if hx >> 31 else f32(0.0)
        return f32(np.nan)
    if ix < 0x3F000000:
        if ix <= 0x32800000:
            return pio2_hi
        return f32(pio2_hi - f32(x - f32(pio2_lo - f32(x * R(f32(x * x))))))
    if hx >> 31:
        z = f32(f32(f32(1.0) + x) * f32(0.5)); s = f32(np.sqrt(z))
        w = f32(f32(R(z) * s) - pio2_lo)
        return f32(f32(2.0) * f32(pio2_hi - f32(s + w)))
    z = f32(f32(f32(1.0) - x) * f32(0.5)); s = f32(np.sqrt(z))
    df = np.uint32(int(np.float32(s).view(np.uint32)) & 0xFFFFF000).view(np.float32)
    c = f32(f32(z - f32(df * df)) / f32(s + df))
    w = f32(f32(R(z) * s) + c)
    return f32(f32(2.0) * f32(df + w))


def shade_multi(normal, wpos, lights, ambient):
    """shade_multi_light_color render.rs:1013-1071 for one vertex (f32 scalars)."""
    t = np.array([ambient, ambient, ambient], np.float32)
    for l in lights:
        if not l.enabled:
            continue
        if l.light_type == 0:
            neg = (np.asarray(l.direction, np.float32) * f32(-1.0)).astype(np.float32)
            contrib = f32(rmax(dot3(normal, neg), f32(0.0)) * f32(l.intensity))
        elif l.light_type == 1:
            to_light = (np.asarray(l.position, np.float32) - wpos).astype(np.float32)
            dist = np.sqrt(dot3(to_light, to_light))
            if dist > f32(l.radius) or dist < K("light.min_dist"):
                contrib = f32(0.0)
            else:
                att = f32(1.0) - (dist / f32(l.radius))
                ndl = rmax(dot3(normal, normalize3(to_light)), f32(0.0))
                contrib = f32(f32(f32(ndl * f32(l.intensity)) * att) * att)
        elif l.light_type == 2:                                  # Spot, render.rs:1038-1058
            to_light = (np.asarray(l.position, np.float32) - wpos).astype(np.float32)
            dist = np.sqrt(dot3(to_light, to_light))
            if dist > f32(l.radius) or dist < K("light.min_dist"):
                contrib = f32(0.0)
            else:
                to_surface = normalize3(to_light)
                neg = (to_surface * f32(-1.0)).astype(np.float32)
                with np.errstate(invalid="ignore"):
                    spot_angle = acosf(dot3(neg, np.asarray(l.direction, np.float32)))
                    if spot_angle > f32(l.angle):
                        contrib = f32(0.0)
                    else:                                        # (NaN angle lands here, as in the reference)
                        att = f32(1.0) - (dist / f32(l.radius))
                        edge = f32(1.0) - f32(spot_angle / f32(l.angle))
                        ndl = rmax(dot3(normal, to_surface), f32(0.0))
                        contrib = f32(f32(f32(f32(ndl * f32(l.intensity)) * att) * att) * edge)
        else:
            raise ValueError("not a LightType")
        col = np.array([l.color.r, l.color.g, l.color.b], np.float32) / K("light.color_div")
        t = (t + contrib * col).astype(np.float32)
    return rmin(t, f32(1.0)).astype(np.float32)


# ---------------------------------------------------------------- render_mesh_15, render.rs:2302-2572
def render_mesh(pixels, width, height, vertices, faces, textures, camera, settings, zbuffer=None):
    """render_mesh (render.rs:1971-2264): the 8-bit-colour path; `textures` are rtypes.Texture."""
    return render_mesh_15(pixels, width, height, vertices, faces, textures, camera, settings, None, zbuffer, fmt8=True)


def render_mesh_15(pixels, width, height, vertices, faces, textures, camera, settings, fog=None, zbuffer=None, fmt8=False):
    """Draw into `pixels` (uint8 [H*W*4]) (and `zbuffer` f32 [H*W] when settings.use_zbuffer); returns dict(triangles_drawn,
    fragments, draw_order, sx, sy)."""
    img = pixels.reshape(height, width, 4)
    zb = zbuffer.reshape(height, width) if settings.use_zbuffer else None
    pos = vertices["pos"].astype(np.float32)
    nv = len(vertices)
    cpos = np.asarray(camera.position, np.float32)
    B = [np.asarray(b, np.float32) for b in (camera.basis_x, camera.basis_y, camera.basis_z)]
    rel = (pos - cpos).astype(np.float32)
    cam = np.stack([dot3(rel, B[0]), dot3(rel, B[1]), dot3(rel, B[2])], axis=1).astype(np.float32)   # math.rs:103-109
    ortho = settings.ortho_projection
    if ortho is not None:                                                                            # :2323-2328, math.rs:140-148
        zoom, ocx, ocy = (f32(x) for x in ortho)
        x = ((cam[:, 0] - ocx).astype(np.float32) * zoom).astype(np.float32) + f32(width) / f32(2.0)
        y = ((-(cam[:, 1] - ocy)).astype(np.float32) * zoom).astype(np.float32) + f32(height) / f32(2.0)
        scr = np.stack([x, y, cam[:, 2]], axis=1).astype(np.float32)
        sx = sy = None
    elif settings.use_fixed_point:                                                                   # :2329-2345
        sx, sy = project_fixed(pos, camera, width, height)
        scr = np.stack([sx.astype(np.float32), sy.astype(np.float32), cam[:, 2] + K("mesh.distance")], axis=1)
    else:                                                                                            # math.rs:117-136
        vs = f32(f32(min(width, height)) / K("project.viewport_div")) * K("project.viewport_frac")
        denom = cam[:, 2] + K("project.distance")
        us = f32(K("project.distance") - K("project.us_sub"))                                         # math.rs:121-122
        with np.errstate(divide="ignore", invalid="ignore"):
            x = (cam[:, 0] * us) / denom * vs + f32(width) / f32(2.0)
            y = (cam[:, 1] * us) / denom * vs + f32(height) / f32(2.0)
        tiny = np.abs(denom) < K("project.denom_guard")
        scr = np.stack([np.where(tiny, f32(width) / f32(2.0), x), np.where(tiny, f32(height) / f32(2.0), y),
                        np.where(tiny, cam[:, 2], denom)], axis=1).astype(np.float32)
        sx = sy = None
    if len(faces) and faces["v"].max() >= nv:
        raise IndexError("vertex index out of range")                                                # :2375-2377

    surfaces = []
    back_wires, front_wires = [], []
    for fi, face in enumerate(faces):
        i0, i1, i2 = (int(k) for k in face["v"])
        cz = cam[[i0, i1, i2], 2]
        if ortho is None and (cz <= K("near_plane")).any():                                                 # :2381-2385
            continue
        v1, v2, v3 = scr[i0], scr[i1], scr[i2]
        signed_area = (v2[0] - v1[0]) * (v3[1] - v1[1]) - (v3[0] - v1[0]) * (v2[1] - v1[1])          # :2393
        back = signed_area <= 0
        tid = int(face["texture_id"])
        tex = textures[tid] if tid < len(textures) else None                                         # textures.get(id)
        transp = ((tex is not None and tex.blend_mode != 0) or face["blend_mode"] != 0 or face["editor_alpha"] < 255)
        if fmt8:
            transp = False                                # render_mesh computes the flag but never partitions (render.rs:2175-2184)
        cols = [np.array([vertices[i]["r"], vertices[i]["g"], vertices[i]["b"], vertices[i]["blend"]], np.int64) for i in (i0, i1, i2)]
        if fog is not None:                                                                          # :2419-2442
            start, falloff, cull, fc = fog
            if (cz > f32(cull)).all():
                continue
            fcol = np.array([fc.r, fc.g, fc.b, fc.blend], np.int64)
            for k in range(3):
                z = cz[k]
                if z <= f32(start):
                    fac = f32(0.0)
                elif f32(falloff) <= 0:
                    fac = f32(1.0)
                else:
                    fac = rmin((z - f32(start)) / f32(falloff), f32(1.0))
                if fac <= 0:
                    pass
                elif fac >= 1:
                    cols[k] = fcol.copy()
                else:
                    inv = f32(1.0) - fac
                    rgb = as_u8(cols[k][:3].astype(np.float32) * inv + fcol[:3].astype(np.float32) * fac)
                    cols[k] = np.array([rgb[0], rgb[1], rgb[2], 0], np.int64)
        if back and not settings.xray_mode:                                                          # :2445-2449
            back_wires.append((v1, v2, v3))
        if not back and settings.wireframe_overlay:                                                  # :2509-2511
            front_wires.append((v1, v2, v3))
        if back and settings.backface_cull and not settings.xray_mode:                               # :2451-2453
            continue
        order = (i0, i2, i1) if back else (i0, i1, i2)
        corder = (0, 2, 1) if back else (0, 1, 2)
        sgn = f32(-1.0) if back else f32(1.0)
        surfaces.append(dict(
            v=[scr[i] for i in order], w=[pos[i] for i in order],
            wn=[(vertices["normal"][i].astype(np.float32) * sgn).astype(np.float32) if back else vertices["normal"][i].astype(np.float32) for i in order],
            uv=[vertices["uv"][i].astype(np.float32) for i in order], vc=[cols[k] for k in corder],
            face=fi, tex=tex, black_tr=bool(face["black_transparent"]), transp=bool(transp),
            blend=int(tex.blend_mode if tex is not None else face["blend_mode"]), alpha=int(face["editor_alpha"])))

    # sort, :2518-2545
    key = np.array([f32(f32(s["v"][0][2] + s["v"][1][2]) + s["v"][2][2]) / f32(3.0) for s in surfaces], np.float32)
    opaque = [i for i, s in enumerate(surfaces) if not s["transp"]]
    transp = [i for i, s in enumerate(surfaces) if s["transp"]]
    def painter(ids):
        ids = np.array(ids, np.int64)
        if len(ids) >= 2 and np.isnan(key[ids]).any():
            raise FloatingPointError("NaN sort key")                                                 # unwrap panic :2531
        return ids[np.argsort(-key[ids], kind="stable")] if len(ids) else ids
    # opaque surfaces are sorted only in painter's mode (render.rs:2535); the transparent pass always is
    draw = (list(opaque) if settings.use_zbuffer else list(painter(opaque))) + list(painter(transp))
    n_opaque = len(opaque)

    fragments = 0
    if not settings.wireframe_overlay:                                                               # :2550
        for k, si in enumerate(draw):
            if fmt8:
                fragments += _rasterize8(img, width, height, surfaces[si], settings, zb)
            else:
                fragments += _rasterize(img, width, height, surfaces[si], settings, zb, skip_z_write=k >= n_opaque)
    # wireframe phases, :2574-2635
    zfull = zbuffer.reshape(height, width) if zbuffer is not None else None
    if settings.backface_cull and settings.backface_wireframe:
        for e in _unique_edges(back_wires):
            draw_line(img, width, height, e, (80, 80, 100), zfull, depth_test=True)
    if settings.wireframe_overlay and front_wires:
        for e in _unique_edges(front_wires):
            draw_line(img, width, height, e, (200, 200, 220), None, depth_test=False)
    return dict(triangles_drawn=len(surfaces), fragments=fragments, draw_order=np.array([surfaces[i]["face"] for i in draw], np.uint32),
                sx=sx, sy=sy, sz=scr[:, 2])


def _as_i32(x):
    """Rust `f32 as i32`: truncating, saturating, NaN -> 0."""
    x = float(x)
    if x != x:
        return 0
    return int(max(-2147483648.0, min(2147483647.0, np.trunc(x))))


def _unique_edges(tris):
    """render.rs:2578-2596: edges as screen integers, direction-normalised, first occurrence kept (with its depths)."""
    seen, out = set(), []
    for tri in tris:
        for j in range(3):
            a, b = tri[j], tri[(j + 1) % 3]
            e = (_as_i32(a[0]), _as_i32(a[1]), f32(a[2]), _as_i32(b[0]), _as_i32(b[1]), f32(b[2]))
            if not ((e[0], e[1]) < (e[3], e[4])):
                e = (e[3], e[4], e[5], e[0], e[1], e[2])
            k = (e[0], e[1], e[3], e[4])
            if k not in seen:
                seen.add(k)
                out.append(e)
    return out


def line_points(x0, y0, x1, y1):
    """Pixels of the reference's Bresenham (render.rs:716-750 / 771-817) in closed form: the k-th plotted point, k = 0..N.
    For an x-major line x advances every iteration and y has advanced floor((2*ady*k + adx) / (2*adx)) times (round half up),
    symmetrically for y-major lines.  Independent of the literal loop in oracle/b32_oracle.c; tests compare the two."""
    adx, ady = abs(x1 - x0), abs(y1 - y0)
    sx, sy = (1 if x0 < x1 else -1), (1 if y0 < y1 else -1)
    n = max(adx, ady)
    k = np.arange(n + 1, dtype=np.int64)
    if n == 0:
        return np.array([x0], np.int64), np.array([y0], np.int64), k
    if adx >= ady:
        return x0 + sx * k, y0 + sy * ((2 * ady * k + adx) // (2 * adx)), k
    return x0 + sx * ((2 * adx * k + ady) // (2 * ady)), y0 + sy * k, k


def draw_line(img, width, height, e, rgb, zb, depth_test):
    x0, y0, z0, x1, y1, z1 = e
    if max(abs(x1 - x0), abs(y1 - y0)) >= 1 << 30:
        raise OverflowError("Bresenham state overflows i32 in the reference")
    xs, ys, k = line_points(x0, y0, x1, y1)
    on = (xs >= 0) & (xs < width) & (ys >= 0) & (ys < height)
    xs, ys, k = xs[on], ys[on], k[on]
    if depth_test:
        total = f32(max(abs(x1 - x0), max(abs(y1 - y0), 1)))
        step = np.minimum(k, 1 << 24).astype(np.float32)                  # step += 1.0 saturates at 2^24 in f32
        with np.errstate(invalid="ignore", over="ignore"):
            t = (step / total).astype(np.float32)
            z = (f32(z0) + (t * (f32(z1) - f32(z0))).astype(np.float32)).astype(np.float32)
            cur = zb[ys, xs] if zb is not None else np.full(len(xs), np.finfo(np.float32).max, np.float32)
            ok = z < cur
        xs, ys = xs[ok], ys[ok]
    img[ys, xs, 0], img[ys, xs, 1], img[ys, xs, 2], img[ys, xs, 3] = rgb[0], rgb[1], rgb[2], 255


def _rasterize(img, width, height, s, st, zb=None, skip_z_write=False):
    """rasterize_triangle_15, render.rs:1440-1714, whole bbox at once (one triangle touches a pixel once, so the sequential
    z-buffer semantics hold within the call)."""
    v1, v2, v3 = s["v"]
    min_x = int(as_usize(rmax(rmin(rmin(v1[0], v2[0]), v3[0]), f32(0.0))))                            # :1455-1458
    max_x = int(as_usize(rmin(rmax(rmax(v1[0], v2[0]), v3[0]) + f32(1.0), f32(width))))
    min_y = int(as_usize(rmax(rmin(rmin(v1[1], v2[1]), v3[1]), f32(0.0))))
    max_y = int(as_usize(rmin(rmax(rmax(v1[1], v2[1]), v3[1]) + f32(1.0), f32(height))))
    if min_x >= max_x or min_y >= max_y:
        return 0
    area = (v2[1] - v3[1]) * (v1[0] - v3[0]) + (v3[0] - v2[0]) * (v1[1] - v3[1])                      # :1500
    if abs(area) < K("fill.area_eps"):
        return 0
    inv_area = f32(1.0) / area
    a0, b0, a1, b1 = v2[1] - v3[1], v3[0] - v2[0], v3[1] - v1[1], v1[0] - v3[0]                       # :1507-1510
    w0s = a0 * (f32(min_x) - v3[0]) + b0 * (f32(min_y) - v3[1])                                       # :1517-1518
    w1s = a1 * (f32(min_x) - v3[0]) + b1 * (f32(min_y) - v3[1])
    nx, ny = max_x - min_x, max_y - min_y
    # incremental walk (:1706-1712) as sequential f32 accumulations: rows first, then along each row
    def walk(start, row_step, col_step):
        rows = np.add.accumulate(np.concatenate([[start], np.full(ny - 1, row_step, np.float32)]).astype(np.float32), dtype=np.float32)
        grid = np.empty((ny, nx), np.float32)
        grid[:, 0] = rows
        if nx > 1:
            grid[:, 1:] = col_step
        return np.add.accumulate(grid, axis=1, dtype=np.float32)
    w0, w1 = walk(w0s, b0, a0), walk(w1s, b1, a1)
    bcx = (w0 * inv_area).astype(np.float32)
    bcy = (w1 * inv_area).astype(np.float32)
    bcz = ((f32(1.0) - bcx) - bcy).astype(np.float32)                                                 # :1538
    E = K("fill.err")
    inside = (bcx >= E) & (bcy >= E) & (bcz >= E)                                                     # :1542
    if st.xray_mode:
        zb = None                                                                                     # :1553: no depth test, no depth write
    zgrid = None
    if zb is not None:                                                                                # :1546-1560
        izs = [f32(1.0) / f32(vv[2]) for vv in (v1, v2, v3)]
        inv_zi = ((bcx * izs[0] + bcy * izs[1]).astype(np.float32) + bcz * izs[2]).astype(np.float32)
        with np.errstate(divide="ignore", invalid="ignore"):
            zgrid = (f32(1.0) / inv_zi).astype(np.float32)
            inside = inside & ~(zgrid >= zb[min_y:max_y, min_x:max_x])
    uv = s["uv"]
    tex = s["tex"]
    if tex is not None:
        if st.affine_textures:
            u = ((bcx * uv[0][0] + bcy * uv[1][0]) + bcz * uv[2][0]).astype(np.float32)               # :1565-1566
            v = ((bcx * uv[0][1] + bcy * uv[1][1]) + bcz * uv[2][1]).astype(np.float32)
        else:                                                                                         # :1546-1579
            iz = [f32(1.0) / f32(vv[2]) for vv in (v1, v2, v3)]
            inv_z = ((bcx * iz[0] + bcy * iz[1]).astype(np.float32) + bcz * iz[2]).astype(np.float32)
            def over_z(c):
                t0 = ((bcx * uv[0][c]).astype(np.float32) * iz[0]).astype(np.float32)
                t1 = ((bcy * uv[1][c]).astype(np.float32) * iz[1]).astype(np.float32)
                t2 = ((bcz * uv[2][c]).astype(np.float32) * iz[2]).astype(np.float32)
                return ((t0 + t1).astype(np.float32) + t2).astype(np.float32)
            with np.errstate(divide="ignore", invalid="ignore"):
                u = (over_z(0) / inv_z).astype(np.float32)
                v = (over_z(1) / inv_z).astype(np.float32)
        if tex.width == 0 or tex.height == 0 or tex.pixels.size == 0:
            texel = np.zeros(u.shape, np.int64)
        else:                                                                                         # types.rs:671-681
            def wrapc(c, n):
                r = np.fmod(c, f32(1.0)).astype(np.float32)
                r = np.where(r < 0, (r + f32(1.0)).astype(np.float32), r)
                return np.minimum(as_usize(r * f32(n)), n - 1)
            tx = wrapc(u, tex.width)
            ty = wrapc((f32(1.0) - v).astype(np.float32), tex.height)
            texel = tex.pixels.astype(np.int64)[ty * tex.width + tx]
    else:
        texel = np.full(bcx.shape, K("color15.white"), np.int64)
    black = (texel & ~K("color15.semi_bit") & 0xFFFF) == 0
    if s["black_tr"]:
        drawn = inside & ~black                                                                       # :1592-1608
    else:
        drawn = inside.copy()
        texel = np.where(texel == K("color15.transparent"), K("color15.black_drawable"), texel)
    if s["alpha"] == 0:                                                                               # :1664-1669
        return 0
    if zb is not None and s["alpha"] == 255:                                                          # :1681-1683 `z < zbuffer`
        with np.errstate(invalid="ignore"):
            drawn = drawn & (zgrid < zb[min_y:max_y, min_x:max_x])
    ys, xs = np.nonzero(drawn)
    if len(ys) == 0:
        return 0
    if zb is not None and not skip_z_write:                                                           # :1686-1688, render.rs:603-605
        zb[ys + min_y, xs + min_x] = zgrid[ys, xs]
    bx, by, bz, tx_ = bcx[ys, xs], bcy[ys, xs], bcz[ys, xs], texel[ys, xs]
    px, py = xs + min_x, ys + min_y
    chans = [(tx_ >> 10) & 31, (tx_ >> 5) & 31, tx_ & 31]
    out5 = []
    for i in range(3):
        tex8 = expand5(chans[i])
        vert = as_u8((bx * f32(s["vc"][0][i]) + by * f32(s["vc"][1][i])).astype(np.float32) + bz * f32(s["vc"][2][i]))   # :1618-1620
        m = np.minimum((tex8 * vert) // K("fill.modulate_div"), K("fill.modulate_max"))               # :1624-1626
        if st.shading != 0:
            if "_sh" not in s:
                _prep_shades(s, st)
            if st.shading == 1:
                sv = np.full(bx.shape, s["_sh"][0][i], np.float32)
            else:
                sv = ((bx * s["_sh"][0][i] + by * s["_sh"][1][i]).astype(np.float32) + bz * s["_sh"][2][i]).astype(np.float32)
            lo, hi = K("fill.shade_clamp_lo"), K("fill.shade_clamp_hi")
            svc = np.where(sv < lo, lo, np.where(sv > hi, hi, sv)).astype(np.float32)                # clamp propagates NaN
            m = as_u8(rmin((m.astype(np.float32) * svc).astype(np.float32), f32(255.0)))              # :1643-1645
        out5.append(m)
    vc = s["vc"]
    needs_dither = st.dithering and (st.shading == 2 or tex is not None or not np.array_equal(vc[0], vc[1]) or not np.array_equal(vc[1], vc[2]))
    if needs_dither:                                                                                  # :1173-1182
        off = DITHER[py & 3, px & 3]
        q = [np.clip((c + off) >> K("dither.shift"), K("dither.clamp_lo"), K("dither.clamp_hi")) for c in out5]
    else:
        q = [c >> K("fill.nodither_shift") for c in out5]
    all_black = (q[0] == 0) & (q[1] == 0) & (q[2] == 0)
    semi = ((tx_ & K("color15.semi_bit")) != 0) | all_black                                                          # :1659-1661
    front = np.stack([expand5(q[0]), expand5(q[1]), expand5(q[2])], axis=1)                           # Color15::r8 etc.
    back = img[py, px, :3].astype(np.int64)
    mode, alpha = s["blend"], s["alpha"]
    do_blend = semi & (mode != 0)
    ps1 = np.where(do_blend[:, None], blend555(front, back, mode), front)
    if st.xray_mode:                                                                                  # render.rs:507-526
        res = (front + back) // 2
    elif alpha < 255:                                                                                 # render.rs:567-591
        res = (ps1 * alpha + back * (255 - alpha)) // 255
    else:
        res = ps1                                                                                     # :445-502
    img[py, px, :3] = res.astype(np.uint8)
    img[py, px, 3] = 255
    return int(len(ys))


def blend8(front, back, mode):
    """Color::blend_with (types.rs:886-936) on int64 arrays [n,3]; mode scalar, not Erase/Opaque-specific shortcuts."""
    if mode == 1:
        return (back + front) // 2
    if mode == 2:
        return np.minimum(back + front, 255)
    if mode == 3:
        return np.maximum(back - front, 0)
    if mode == 4:
        return np.minimum(back + front // 4, 255)
    if mode == 5:
        return np.zeros_like(front)
    return front


def _rasterize8(img, width, height, s, st, zb=None):
    """rasterize_triangle (render.rs:1202-1433), whole bbox at once."""
    v1, v2, v3 = s["v"]
    min_x = int(as_usize(rmax(rmin(rmin(v1[0], v2[0]), v3[0]), f32(0.0))))
    max_x = int(as_usize(rmin(rmax(rmax(v1[0], v2[0]), v3[0]) + f32(1.0), f32(width))))
    min_y = int(as_usize(rmax(rmin(rmin(v1[1], v2[1]), v3[1]), f32(0.0))))
    max_y = int(as_usize(rmin(rmax(rmax(v1[1], v2[1]), v3[1]) + f32(1.0), f32(height))))
    if min_x >= max_x or min_y >= max_y:
        return 0
    area = (v2[1] - v3[1]) * (v1[0] - v3[0]) + (v3[0] - v2[0]) * (v1[1] - v3[1])
    if abs(area) < K("fill8.area_eps"):
        return 0
    inv_area = f32(1.0) / area
    a0, b0, a1, b1 = v2[1] - v3[1], v3[0] - v2[0], v3[1] - v1[1], v1[0] - v3[0]
    w0s = a0 * (f32(min_x) - v3[0]) + b0 * (f32(min_y) - v3[1])
    w1s = a1 * (f32(min_x) - v3[0]) + b1 * (f32(min_y) - v3[1])
    nx, ny = max_x - min_x, max_y - min_y

    def walk(start, row_step, col_step):
        rows = np.add.accumulate(np.concatenate([[start], np.full(ny - 1, row_step, np.float32)]).astype(np.float32), dtype=np.float32)
        grid = np.empty((ny, nx), np.float32)
        grid[:, 0] = rows
        if nx > 1:
            grid[:, 1:] = col_step
        return np.add.accumulate(grid, axis=1, dtype=np.float32)
    w0, w1 = walk(w0s, b0, a0), walk(w1s, b1, a1)
    bcx = (w0 * inv_area).astype(np.float32)
    bcy = (w1 * inv_area).astype(np.float32)
    bcz = ((f32(1.0) - bcx) - bcy).astype(np.float32)
    E = K("fill8.err")
    inside = (bcx >= E) & (bcy >= E) & (bcz >= E)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        iz = [f32(1.0) / f32(vv[2]) for vv in (v1, v2, v3)]
        inv_z = ((bcx * iz[0] + bcy * iz[1]).astype(np.float32) + bcz * iz[2]).astype(np.float32)
        zgrid = (f32(1.0) / inv_z).astype(np.float32)
        zsub = zb[min_y:max_y, min_x:max_x] if zb is not None else None
        if zb is not None and not st.xray_mode:                                                       # :1312-1319
            inside = inside & ~(zgrid >= zsub)
    uv, tex = s["uv"], s["tex"]
    if tex is not None:
        if st.affine_textures:
            u = ((bcx * uv[0][0] + bcy * uv[1][0]) + bcz * uv[2][0]).astype(np.float32)
            v = ((bcx * uv[0][1] + bcy * uv[1][1]) + bcz * uv[2][1]).astype(np.float32)
        else:
            def over_z(c):
                t0 = ((bcx * uv[0][c]).astype(np.float32) * iz[0]).astype(np.float32)
                t1 = ((bcy * uv[1][c]).astype(np.float32) * iz[1]).astype(np.float32)
                t2 = ((bcz * uv[2][c]).astype(np.float32) * iz[2]).astype(np.float32)
                return ((t0 + t1).astype(np.float32) + t2).astype(np.float32)
            with np.errstate(divide="ignore", invalid="ignore"):
                u = (over_z(0) / inv_z).astype(np.float32)
                v = (over_z(1) / inv_z).astype(np.float32)
        if tex.width == 0 or tex.height == 0 or tex.pixels.size == 0:
            texel = np.zeros(u.shape + (4,), np.int64); texel[..., 3] = 5
        else:
            def wrapc(c, n):
                r = np.fmod(c, f32(1.0)).astype(np.float32)
                r = np.where(r < 0, (r + f32(1.0)).astype(np.float32), r)
                return np.minimum(as_usize(r * f32(n)), n - 1)
            tx = wrapc(u, tex.width)
            ty = wrapc((f32(1.0) - v).astype(np.float32), tex.height)
            texel = tex.pixels.astype(np.int64)[ty * tex.width + tx]
    else:
        texel = np.zeros(bcx.shape + (4,), np.int64); texel[..., :3] = 255                            # Color::WHITE
    drawn = inside & (texel[..., 3] != 5)                                                             # is_transparent :1348
    if s["alpha"] == 0:
        return 0
    alpha = s["alpha"]
    if zb is not None:
        with np.errstate(invalid="ignore"):
            drawn = drawn & (~(zgrid >= zsub) if alpha < 255 else (zgrid < zsub))                    # render.rs:387 vs :432, :1407
    ys, xs = np.nonzero(drawn)
    if len(ys) == 0:
        return 0
    if zb is not None:
        zb[ys + min_y, xs + min_x] = zgrid[ys, xs]                                                   # every passing fragment writes depth
    bx, by, bz, tx_ = bcx[ys, xs], bcy[ys, xs], bcz[ys, xs], texel[ys, xs]
    px, py = xs + min_x, ys + min_y
    cols = []
    for i in range(3):
        vert = as_u8((bx * f32(s["vc"][0][i]) + by * f32(s["vc"][1][i])).astype(np.float32) + bz * f32(s["vc"][2][i]))
        m = np.minimum((tx_[:, i] * vert) // 128, 255)                                                # modulate, types.rs:801-808
        if st.shading != 0:
            if "_sh" not in s:
                _prep_shades(s, st)
            if st.shading == 1:
                sv = np.full(bx.shape, s["_sh"][0][i], np.float32)
            else:
                sv = ((bx * s["_sh"][0][i] + by * s["_sh"][1][i]).astype(np.float32) + bz * s["_sh"][2][i]).astype(np.float32)
            m = as_u8(rmin((m.astype(np.float32) * sv).astype(np.float32), f32(255.0)))               # shade_color_rgb :1074-1081 (no clamp)
        cols.append(m)
    vc = s["vc"]
    needs_dither = st.dithering and (st.shading == 2 or tex is not None or not np.array_equal(vc[0], vc[1]) or not np.array_equal(vc[1], vc[2]))
    if needs_dither:                                                                                  # apply_dither :1186-1197
        off = DITHER[py & 3, px & 3]
        cols = [np.clip((c + off) >> K("dither8.shift"), 0, K("dither8.clamp_hi")) << K("dither8.expand_shift") for c in cols]
    front = np.stack(cols, axis=1).astype(np.int64)
    back = img[py, px, :3].astype(np.int64)
    mode = tx_[:, 3]
    ps1 = front.copy()
    a_out = np.full(len(ys), 255, np.int64)
    for mval in (1, 2, 3, 4, 5):
        sel = mode == mval
        if sel.any():
            ps1[sel] = blend8(front[sel], back[sel], mval)
            if mval == 5:
                a_out[sel] = 0
    if alpha < 255:                                                                                   # render.rs:356-366
        a = f32(alpha) / f32(255.0)
        inv_a = f32(1.0) - a
        res = as_u8((ps1.astype(np.float32) * a).astype(np.float32) + (back.astype(np.float32) * inv_a).astype(np.float32))
        a_out[:] = 255
    else:
        res = ps1
    img[py, px, :3] = res.astype(np.uint8)
    img[py, px, 3] = a_out.astype(np.uint8)
    return int(len(ys))


def _prep_shades(s, st):
    if st.shading == 1:                                                                               # :1466-1469
        third = f32(1.0) / f32(3.0)
        center = (((s["w"][0] + s["w"][1]).astype(np.float32) + s["w"][2]).astype(np.float32) * third).astype(np.float32)
        n = normalize3((((s["wn"][0] + s["wn"][1]).astype(np.float32) + s["wn"][2]).astype(np.float32) * third).astype(np.float32))
        sh = shade_multi(n, center, st.lights, f32(st.ambient))
        s["_sh"] = [sh, sh, sh]
    else:                                                                                             # :1475-1483
        s["_sh"] = [shade_multi(s["wn"][k], s["w"][k], st.lights, f32(st.ambient)) for k in range(3)]

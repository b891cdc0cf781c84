"""Known-answer tests that pin the CPU oracle (oracle/b32_oracle.c) against the reference's OWN unit tests and the
public specs its comments name.  Reference: /root/reference/src/rasterizer/{fixed.rs:473-549, math.rs:779-807,
render.rs:1093-1182, types.rs:20-227}.  CPU only."""
import ctypes as C

import numpy as np
import pytest

import bonnie32_amd as b32


def fx(oracle, f):
    return oracle.lib().b32o_fixed_from_f32(C.c_float(f))


def to_f(oracle, v):
    return oracle.lib().b32o_fixed_to_f32(v)


def test_fixed32_precision(oracle):                      # fixed.rs:478-489 test_fixed32_precision
    assert fx(oracle, 1.0) == 4096
    assert fx(oracle, 0.5) == 2048
    assert fx(oracle, 0.001) > 0


def test_fixed32_mul(oracle):                            # fixed.rs:492-497
    r = oracle.lib().b32o_fixed_mul(fx(oracle, 2.0), fx(oracle, 3.0))
    assert abs(to_f(oracle, r) - 6.0) < 0.01


def test_unr_division_has_error(oracle):                 # fixed.rs:500-510
    r = oracle.lib().b32o_fixed_div_unr(fx(oracle, 10.0), fx(oracle, 3.0))
    assert abs(to_f(oracle, r) - 10.0 / 3.0) < 0.1


def test_unr_division_basic(oracle):                     # fixed.rs:513-531
    L = oracle.lib()
    assert abs(to_f(oracle, L.b32o_fixed_div_unr(fx(oracle, 10.0), fx(oracle, 2.0))) - 5.0) < 0.01
    assert abs(to_f(oracle, L.b32o_fixed_div_unr(fx(oracle, -6.0), fx(oracle, 2.0))) + 3.0) < 0.01
    assert abs(to_f(oracle, L.b32o_fixed_div_unr(fx(oracle, 7.5), fx(oracle, 1.0))) - 7.5) < 0.1


def test_projection_outputs_integers(oracle):            # fixed.rs:534-548
    x, y, _ = oracle.project_fixed((1.234, 2.567, 5.0), b32.Camera(), 320, 240)
    assert -1000 < x < 1000 and -1000 < y < 1000


def test_vec3_ops(oracle):                               # math.rs:784-799 (dot = 32, cross z = 1)
    L = oracle.lib()
    a = (C.c_float * 3)(1, 2, 3); b = (C.c_float * 3)(4, 5, 6)
    assert L.b32o_vec3_dot(a, b) == 32.0
    out = (C.c_float * 3)()
    L.b32o_vec3_cross((C.c_float * 3)(1, 0, 0), (C.c_float * 3)(0, 1, 0), out)
    assert list(out) == [0.0, 0.0, 1.0]


def test_unr_table_matches_psx_spx(oracle):
    """fixed.rs:18-19 names the psx-spx GTE table: first 16, last 5 and the checksum of the 257 entries."""
    t = [oracle.lib().b32o_unr_table(i) for i in range(257)]
    assert t[:16] == [0xFF, 0xFD, 0xFB, 0xF9, 0xF7, 0xF5, 0xF3, 0xF1, 0xEF, 0xEE, 0xEC, 0xEA, 0xE8, 0xE6, 0xE4, 0xE3]
    assert t[-5:] == [1, 1, 0, 0, 0]
    assert sum(t) == 25186
    assert t == [max(0, (0x40000 // (i + 0x100) + 1) // 2 - 0x101) for i in range(257)]


def test_div_unr_and_project_vectors(oracle):
    """Integer answers derived independently during the survey with a scratch model."""
    L = oracle.lib()
    assert L.b32o_fixed_div_unr(40960, 12288) == 13653
    assert L.b32o_fixed_div_unr(40960, 8192) == 20480
    assert L.b32o_fixed_div_unr(-24576, 8192) == -12288
    assert L.b32o_fixed_div_unr(30720, 4096) == 30720
    assert L.b32o_fixed_div_unr(123, 0) == 0
    assert oracle.project_fixed((1.234, 2.567, 5.0), b32.Camera(), 320, 240)[:2] == (204, 212)
    assert oracle.project_fixed((1.234, 2.567, 5.0), b32.Camera(), 2560, 1920)[:2] == (1635, 1699)
    assert oracle.project_fixed((-300.5, 120.25, 2000.0), b32.Camera(position=(10, 20, -30)), 320, 240)[:2] == (105, 137)
    # near-zero denominator -> screen centre (fixed.rs:406-408)
    assert oracle.project_fixed((3.0, 4.0, -5.0), b32.Camera(), 320, 240)[:2] == (160, 120)


def test_project_fixed_matches_numpy_model(oracle):
    from oracle import np_model as M
    rng = np.random.default_rng(5)
    pos = (rng.standard_normal((4000, 3)) * np.array([3000, 3000, 4000])).astype(np.float32)
    pos[:50] *= 1e4                                       # saturating conversions / wrapping dot products
    cam = b32.Camera(position=(12.5, -7.25, 3.0), basis_x=(0.8, 0.0, -0.6), basis_y=(0.0, 1.0, 0.0), basis_z=(0.6, 0.0, 0.8))
    sx, sy = M.project_fixed(pos, cam, 640, 480)
    for i in range(len(pos)):
        x, y, _ = oracle.project_fixed(pos[i], cam, 640, 480)
        assert (x, y) == (sx[i], sy[i]), i


def test_dither_matrix_is_psx_spx(oracle):               # render.rs:1150-1155
    m = [[oracle.lib().b32o_dither_offset(x, y) for x in range(4)] for y in range(4)]
    assert m == [[-4, 0, -3, 1], [2, -2, 3, -1], [-3, 1, -4, 0], [3, -1, 2, -2]]
    out = (C.c_uint8 * 3)()
    oracle.lib().b32o_dither_and_quantize(0, 255, 130, 0, 0, out)      # offset -4: clamp low, (251>>3)=31, (126>>3)=15
    assert list(out) == [0, 31, 15]
    oracle.lib().b32o_dither_and_quantize(250, 5, 7, 2, 1, out)        # offset +3
    assert list(out) == [31, 1, 1]


@pytest.mark.parametrize("mode,expect", [(0, (16, 24, 248)), (1, (64, 64, 120)), (2, (136, 128, 248)),
                                         (3, (104, 80, 0)), (4, (120, 104, 56)), (5, (120, 104, 0))])
def test_blend_rgb555(oracle, mode, expect):             # render.rs:1093-1145: 5-bit maths, result << 3 (no bit replication)
    out = (C.c_uint8 * 3)()
    oracle.lib().b32o_blend_rgb555(20, 30, 250, 123, 110, 5, mode, out)
    assert tuple(out) == expect


def test_color15_to_rgba(oracle):                        # types.rs:220-226
    out = (C.c_uint8 * 4)()
    oracle.lib().b32o_color15_to_rgba(0x0000, out); assert list(out) == [0, 0, 0, 0]
    oracle.lib().b32o_color15_to_rgba(0x7FFF, out); assert list(out) == [255, 255, 255, 255]
    oracle.lib().b32o_color15_to_rgba(0x8000, out); assert list(out) == [0, 0, 0, 255]
    oracle.lib().b32o_color15_to_rgba((1 << 10) | (2 << 5) | 31, out); assert list(out) == [8, 16, 255, 255]


def test_texture_sample_wrapping(oracle):                # types.rs:671-681
    px = np.arange(16, dtype=np.uint16) + 1
    s = lambda u, v: oracle.lib().b32o_texture15_sample(px.ctypes.data, 4, 4, C.c_float(u), C.c_float(v))
    assert s(0.0, 0.0) == 1 and s(0.99, 0.99) == 16
    assert s(-0.25, 0.0) == 4                                   # rem_euclid: -0.25 -> 0.75
    assert s(1.5, 2.25) == px[1 * 4 + 2]
    assert s(-1e-10, 0.0) == 4                                  # rem_euclid gives exactly 1.0 -> clamped to width-1
    assert s(float("nan"), 0.0) == 1                            # NaN as usize == 0
    assert oracle.lib().b32o_texture15_sample(None, 0, 0, C.c_float(0.5), C.c_float(0.5)) == 0


def test_indexed_expansion(oracle):                      # types.rs:390-397, mesh_editor.rs:669-682
    idx = np.array([0, 1, 15, 16, 255], np.uint8)
    clut = (np.arange(16) + 100).astype(np.uint16)
    out = np.zeros(5, np.uint16)
    oracle.lib().b32o_expand_indexed(idx.ctypes.data, 5, clut.ctypes.data, 16, out.ctypes.data)
    assert list(out) == [100, 101, 115, 0, 0]
    t = b32.IndexedTexture(5, 1, idx, clut).to_texture15()
    assert list(t.pixels) == list(out)


def test_error_codes(oracle):
    sc_v = b32.make_vertices(3); sc_f = b32.make_faces(1)
    sc_v["pos"] = [[0, 0, 10], [1, 0, 10], [0, 1, 10]]
    sc_f["v"][0] = (0, 1, 7)
    fb = oracle.Framebuffer(32, 32)
    st = b32.RasterSettings.benchmark()
    assert oracle.render_mesh_15(fb, sc_v, sc_f, [], b32.Camera(), st)[0] == b32.abi.B32_E_INDEX
    sc_f["v"][0] = (0, 1, 2)
    v2 = np.concatenate([sc_v, sc_v]); f2 = np.concatenate([sc_f, sc_f]); f2["v"][1] = (3, 4, 5)
    v2["pos"][3:6] = [[0, 0, 20], [0, 1, 20], [1, 0, 20]]      # front-facing (y down)
    v2["pos"][0:3] = [[0, 0, 10], [0, 1, 10], [1, 0, 10]]
    v2["pos"][4, 2] = np.nan
    st_nc = b32.RasterSettings.benchmark(); st_nc.backface_cull = False
    assert oracle.render_mesh_15(fb, v2, f2, [], b32.Camera(), st_nc)[0] == b32.abi.B32_E_NAN_KEY
    # a single surface never compares keys: no panic in the reference (sort_by on a 1-element slice)
    assert oracle.render_mesh_15(fb, v2[3:6], sc_f, [], b32.Camera(), st_nc)[0] == 0
    st2 = b32.RasterSettings()                                  # reference defaults (z-buffer, back-face wireframe) are in scope
    assert oracle.render_mesh_15(fb, sc_v, sc_f, [], b32.Camera(), st2)[0] == 0
    st2.lights = [b32.Light.spot((0, 0, 0), (0, 0, 1), 0.5, 50.0, 1.0)]
    assert oracle.render_mesh_15(fb, sc_v, sc_f, [], b32.Camera(), st2)[0] == 0                            # spot lights are in scope
    st2.lights = [b32.Light(7, position=(0, 0, 0), direction=(0, 0, 1), radius=50.0, angle=0.5)]
    assert oracle.render_mesh_15(fb, sc_v, sc_f, [], b32.Camera(), st2)[0] == b32.abi.B32_E_ARG           # not a LightType


def test_acosf_is_the_published_musl_algorithm(oracle):
    """f32::acos of the spot-light cone test (render.rs:1047).  The oracle, oracle/np_model.py and the GPU all use the musl /
    `libm`-crate acosf (what the reference's wasm32 build executes).  Known answers of that algorithm's special cases, agreement
    of the two restatements bit for bit, and its published accuracy (< 1 ulp of the true value)."""
    from oracle import np_model as M
    L = oracle.lib()
    bits = lambda f: int(np.float32(f).view(np.uint32))
    assert bits(L.b32o_acosf(1.0)) == 0 and bits(L.b32o_acosf(-1.0)) == 0x40490FDA          # 0 and 2 * pio2_hi
    assert bits(L.b32o_acosf(0.0)) == 0x3FC90FDA and bits(L.b32o_acosf(1e-9)) == 0x3FC90FDA  # pio2_hi below 2^-26
    assert np.isnan(L.b32o_acosf(1.0000001)) and np.isnan(L.b32o_acosf(-2.0)) and np.isnan(L.b32o_acosf(float("nan")))
    rng = np.random.default_rng(5)
    xs = np.concatenate([rng.uniform(-1, 1, 3000), np.cos(rng.uniform(0, np.pi, 3000)), 1 - 10 ** rng.uniform(-8, 0, 1500),
                         -1 + 10 ** rng.uniform(-8, 0, 1500), [0.5, -0.5, 0.49999997, -0.50000006]]).astype(np.float32)
    xs = np.clip(xs, -1, 1)
    worst = 0.0
    for x in xs:
        a = np.float32(L.b32o_acosf(float(x)))
        b = M.acosf(x)
        assert bits(a) == bits(b), (float(x), float(a), float(b))
        true = np.arccos(np.float64(x))
        worst = max(worst, abs(float(a) - true) / float(np.spacing(np.float32(true))))
    assert worst < 1.0, worst


def test_bresenham_closed_form_equals_literal_loop(oracle):
    """The GPU (b32_wire.hip) and oracle/np_model.py use the closed form of the reference's Bresenham state
    (render.rs:716-750); oracle/b32_oracle.c runs the loop literally.  Same pixels, and same per-pixel depths through
    draw_line_3d (render.rs:771-817), on random lines including steep, flat, reversed, degenerate and off-screen ones."""
    import ctypes as C
    from oracle import np_model as M
    L = oracle.lib()
    L.b32o_draw_line.restype = C.c_int
    L.b32o_draw_line.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32] + [C.c_int32] * 4 + [C.c_uint8] * 3
    L.b32o_draw_line_3d.restype = C.c_int
    L.b32o_draw_line_3d.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_float,
                                    C.c_int32, C.c_int32, C.c_float] + [C.c_uint8] * 3
    rng = np.random.default_rng(5)
    W, H = 97, 61
    zb = (rng.random(W * H) * 100).astype(np.float32)
    for it in range(1500):
        x0, y0, x1, y1 = [int(v) for v in rng.integers(-60, 160, 4)]
        if it % 7 == 0: x1 = x0
        if it % 11 == 0: y1 = y0
        if it % 13 == 0: x1, y1 = x0 + (y1 - y0), y0 + (y1 - y0)      # exact diagonal
        z0, z1 = np.float32(rng.random() * 100), np.float32(rng.random() * 100)
        a = np.zeros(W * H * 4, np.uint8); b = np.zeros((H, W, 4), np.uint8)
        assert L.b32o_draw_line(a.ctypes.data, W, H, x0, y0, x1, y1, 1, 2, 3) == 0
        M.draw_line(b, W, H, (x0, y0, z0, x1, y1, z1), (1, 2, 3), None, False)
        assert np.array_equal(a.reshape(H, W, 4), b), (x0, y0, x1, y1)
        a[:] = 0; b[:] = 0
        assert L.b32o_draw_line_3d(a.ctypes.data, zb.ctypes.data, W, H, x0, y0, z0, x1, y1, z1, 4, 5, 6) == 0
        M.draw_line(b, W, H, (x0, y0, z0, x1, y1, z1), (4, 5, 6), zb.reshape(H, W), True)
        assert np.array_equal(a.reshape(H, W, 4), b), (x0, y0, x1, y1)
    # a line whose Bresenham state overflows i32 in the reference is refused, not walked
    assert L.b32o_draw_line(a.ctypes.data, W, H, -(1 << 30), 0, 5, 5, 1, 2, 3) == b32.abi.B32_E_UNSUPPORTED


def _steps_inside_closed_form(e, cx0, cx1, cy0, cy1):
    """line_k_range_exact of b32_wire.hip, statement by statement, in Python integers: the steps of the line whose pixel lies inside
    the rectangle, as one interval (None: no step)"""
    x0, y0, x1, y1 = e
    adx, ady = abs(x1 - x0), abs(y1 - y0)
    xmajor = adx >= ady
    N, m0, lo, hi = (adx, x0, cx0, cx1) if xmajor else (ady, y0, cy0, cy1)
    forward = (x0 < x1) if xmajor else (y0 < y1)
    k_lo, k_hi = (max(0, lo - m0), min(N, hi - m0)) if forward else (max(0, m0 - hi), min(N, m0 - lo))
    if k_lo > k_hi:
        return None
    dmaj, dmin = N, (ady if xmajor else adx)
    n0, nlo, nhi = (y0, cy0, cy1) if xmajor else (x0, cx0, cx1)
    up = (y0 < y1) if xmajor else (x0 < x1)
    ja, jb = (nlo - n0, nhi - n0) if up else (n0 - nhi, n0 - nlo)
    if jb < 0 or ja > dmin:
        return None
    ja, jb = max(ja, 0), min(jb, dmin)
    if dmin > 0:
        d = 2 * dmin
        if ja >= 1:
            k_lo = max(k_lo, (dmaj * (2 * ja - 1) + d - 1) // d)
        k_hi = min(k_hi, (dmaj * (2 * jb + 1) + d - 1) // d - 1)
        assert dmaj * (2 * jb + 1) + d - 1 < 2 ** 31                       # (the device computes this in 32 bits)
    return (k_lo, k_hi) if k_lo <= k_hi else None


def test_wire_depth_parameter_by_reciprocal_equals_the_division():
    """k_wire_tile computes draw_line_3d's `t = step / total_steps` (render.rs:784) as q = k * r, t = fma(fma(-q, N, k), r, q) with
    r = 1.0f / N (wire_t_fast, b32_device.h).  Emulated here in float64 (every product below is exact in 53 bits; only the last sum is
    rounded twice) for EVERY pair 0 <= k <= N < 16384 -- the range the kernel uses it for -- against numpy's correctly rounded float32
    division.  The device itself is checked by b32_selftest_f32 op 8 (tests/test_gpu_parity.py::test_device_f32_semantics)."""
    k_all = np.arange(0, 16384, dtype=np.float64)
    for n0 in range(1, 16384, 128):
        N = np.arange(n0, min(n0 + 128, 16384), dtype=np.float32)[:, None]
        r = (np.float32(1.0) / N)
        k = k_all[None, :]
        q = (k * r.astype(np.float64)).astype(np.float32)
        rem = (k - q.astype(np.float64) * N.astype(np.float64)).astype(np.float32)
        t = (q.astype(np.float64) + rem.astype(np.float64) * r.astype(np.float64)).astype(np.float32)
        want = k.astype(np.float32) / N
        use = k <= N.astype(np.float64)
        assert np.array_equal(t[use], want[use]), n0


def test_wire_tile_clip_closed_form_equals_literal_loop():
    """k_wire_tile (b32_wire.hip) walks only the steps of a line whose pixel lies inside its tile, found in closed form from both axes;
    the literal loop of draw_line / draw_line_3d (render.rs:716-750, 771-817) visits the same steps, and they are one interval.
    Step numbers are the loop's iteration count = the reference's `step` (render.rs:779, 805-812)."""
    import random
    rnd = random.Random(5)
    for it in range(30000):
        R = rnd.choice([8, 40, 200, 1000, 16000])
        x0, y0 = rnd.randint(-R, R + 64), rnd.randint(-R, R + 16)
        x1, y1 = x0 + rnd.randint(-min(R, 16383), min(R, 16383)), y0 + rnd.randint(-min(R, 16383), min(R, 16383))
        if rnd.random() < 0.1: x1 = x0
        if rnd.random() < 0.1: y1 = y0
        if rnd.random() < 0.05: y1 = y0 + (x1 - x0)
        if (x0, y0) > (x1, y1) and it % 3:                                     # (wire_edge's direction, and the other one)
            x0, y0, x1, y1 = x1, y1, x0, y0
        cx0 = rnd.choice([0, 64, 640]); cx1 = cx0 + rnd.choice([63, 30, 0])
        cy0 = rnd.choice([0, 16, 5, 480]); cy1 = cy0 + rnd.choice([15, 7, 0])
        dx, dy = abs(x1 - x0), -abs(y1 - y0)
        sx, sy = (1 if x0 < x1 else -1), (1 if y0 < y1 else -1)
        err, x, y, k, ks = dx + dy, x0, y0, 0, []
        while True:
            if cx0 <= x <= cx1 and cy0 <= y <= cy1:
                ks.append(k)
            if x == x1 and y == y1:
                break
            e2 = 2 * err
            if e2 >= dy: err += dy; x += sx
            if e2 <= dx: err += dx; y += sy
            k += 1
        want = (ks[0], ks[-1]) if ks else None
        assert not ks or ks == list(range(ks[0], ks[-1] + 1)), (x0, y0, x1, y1)
        assert _steps_inside_closed_form((x0, y0, x1, y1), cx0, cx1, cy0, cy1) == want, ((x0, y0, x1, y1), (cx0, cx1, cy0, cy1))


# ---------------------------------------------------------------------------------------------------------------------------
# Constants pinned to the reference's own text (tests/golden/pin_constants.py -> tests/golden/ref_constants.json)
def _ref_constants():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_constants.json")))


def _expect(entry):
    """the value a `const X: f32 = <literal>` / integer literal of the reference holds"""
    v = entry["value"]
    if isinstance(v, float):
        assert int(np.float32(v).view(np.uint32)) == entry["f32_bits"]
        return float(np.float32(v))
    return v


def test_fixture_is_current():
    """Where the reference is present (the build container) the committed fixture must be exactly what the script derives from the
    reference text today; on the GPU box (no /root/reference) only its shape is checked."""
    import importlib.util
    import os
    ref = _ref_constants()
    assert len(ref) >= 60 and all("source" in e and e["source"].startswith("src/rasterizer/") for e in ref.values())
    spec = importlib.util.spec_from_file_location("pin_constants", os.path.join(os.path.dirname(__file__), "golden", "pin_constants.py"))
    pin = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pin)
    if not os.path.isdir(pin.SRC):
        pytest.skip("reference text not on this box: the committed fixture is the pin")
    assert pin.derive() == ref


def test_oracle_constants_are_the_reference_text(oracle):
    """Every named literal the C oracle computes with (b32o_constant: the K_* macros its code uses) equals the literal found in the
    reference text; the UNR table it indexes and the dither matrix it adds are the ones the reference's generator / table give."""
    ref = _ref_constants()
    got = oracle.constants()
    scalar_keys = {k for k, e in ref.items() if not isinstance(e["value"], list)}
    assert set(got) == scalar_keys, (sorted(set(got) ^ scalar_keys))
    for k in sorted(scalar_keys):
        assert got[k] == _expect(ref[k]), (k, got[k], ref[k])
    L = oracle.lib()
    assert [L.b32o_unr_table(i) for i in range(257)] == ref["unr.table"]["value"]
    assert [[L.b32o_dither_offset(x, y) for x in range(4)] for y in range(4)] == ref["dither.matrix"]["value"]
    # the periodic extension the fill uses: [y & 3][x & 3]
    assert L.b32o_dither_offset(6, 9) == ref["dither.matrix"]["value"][1][2]


def test_np_model_constants_are_the_reference_text():
    """oracle/np_model.py holds no literal of its own: it loads the fixture.  Spot-check that the loaded values are the ones in use."""
    from oracle import np_model as M
    ref = _ref_constants()
    assert M.UNR_TABLE.tolist() == ref["unr.table"]["value"] and M.DITHER.tolist() == ref["dither.matrix"]["value"]
    assert float(M.K("fill.err")) == float(np.float32(ref["fill.err"]["value"])) and M.K("div_unr.nr1_const") == 0x2000080
    src = open(M.__file__).read()
    for lit in ("0.0001", "0.00001", "0x7FC0", "0x2000080", "0x101", "262144", "0x40000"):
        assert lit not in src, f"np_model.py still carries its own copy of {lit}"


def test_wire_edge_hash_set_equals_the_quadratic_scan(oracle):
    """The wireframe phases keep the FIRST occurrence of every screen-space edge, found by `unique_edges.iter().any(..)` before each push
    (render.rs:2589-2594): O(n^2).  The oracle's hash set must produce the same list, entry for entry and with the first occurrence's
    depths, so that RasterSettings::default() frames of a million triangles can be checked at all.  20 000 triangles on a coarse grid:
    shared edges, reversed duplicates, degenerate edges, edges differing in one coordinate only, hostile coordinates."""
    rng = np.random.default_rng(20)
    n = 20_000
    t = np.zeros((n, 9), np.float32)
    t[:, [0, 1, 3, 4, 6, 7]] = rng.integers(-3, 40, (n, 6)).astype(np.float32)            # few distinct points: most edges repeat
    t[:, [2, 5, 8]] = rng.uniform(5.0, 900.0, (n, 3)).astype(np.float32)                  # every occurrence has its own depths
    t[::7, 3:6] = t[::7, 0:3]                                                               # zero-length edges
    t[::11] = t[::11][:, [3, 4, 5, 0, 1, 2, 6, 7, 8]]                                       # the same triangle wound the other way
    t[::501, 0] = [3e9, -3e9, np.nan, np.inf][0]                                            # `as i32` saturates
    t[1::501, 1] = np.nan
    t[2::501, 4] = -np.inf
    assert oracle.unique_edges_selfcheck(t) == 0
    # (below 64 triangles unique_edges() IS the quadratic scan: sizes at and just above that threshold exercise the hash set)
    for m in (0, 50, 63, 64, 65, 100):
        assert oracle.unique_edges_selfcheck(t[:m]) == 0
    # every edge distinct, and every edge the same
    u = t.copy(); u[:, 0] = np.arange(n, dtype=np.float32) * 3; u[:, 3] = u[:, 0] + 1; u[:, 6] = u[:, 0] + 2
    assert oracle.unique_edges_selfcheck(u) == 0
    assert oracle.unique_edges_selfcheck(np.tile(t[:1], (5000, 1))) == 0

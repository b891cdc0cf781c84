"""Span coverage (b32_fill.hip: span_edge / span_interval / span_trip, B32_ROUTE_SPAN_COVER) replaces the reference's per-pixel inside
test (render.rs:1536-1542, evaluated on rounded f32 values with a -1e-4 tolerance) by ONE interval per row whose ends are integer
quotients.  Two claims carry it, both checked here by brute force in numpy f32:

  (1) for an integer triangle with |area| <= 8192 the toleranced float test passes exactly on the pixels of the closed integer
      triangle  s*w0 >= 0, s*w1 >= 0, |area| - s*w0 - s*w1 >= 0  -- over every pixel of a window around the triangle, not only its box;
  (2) the device's interval arithmetic (reciprocal of the step, shifted by half a step, one fma, ceil / floor) yields exactly that
      set on every row, with the hardware's approximate reciprocal modelled as the exact one moved by up to +-2 ulp (v_rcp_f32 is
      specified to 1 ulp) and the fma as an exactly rounded one.

The device side is compared with the oracle frame by frame in tests/test_gpu_parity.py (every painter's scene runs with the route on
and off); this file is the CPU-side proof the kernel's eligibility rule rests on."""
import numpy as np
import pytest

F = np.float32
ERR = F(-0.0001)
MAX_EXT = 512
MAX_AREA = 8192


def float_inside(v, xs, ys):
    """The reference's test on a grid of pixels: closed-form edge values (exact integers here, equal to its accumulation),
    bc = w * inv_area in f32, bc_z = (1 - bc_x) - bc_y, all three >= -1e-4."""
    (x1, y1), (x2, y2), (x3, y3) = [(F(a), F(b)) for a, b in v]
    area = F(F((y2 - y3) * (x1 - x3)) + F((x3 - x2) * (y1 - y3)))            # render.rs:1500
    inv_area = F(F(1.0) / area)
    a0, b0, a1, b1 = F(y2 - y3), F(x3 - x2), F(y3 - y1), F(x1 - x3)          # :1507-1510
    X, Y = np.meshgrid(xs.astype(F), ys.astype(F))
    dx, dy = (X - x3).astype(F), (Y - y3).astype(F)
    w0 = ((a0 * dx).astype(F) + (b0 * dy).astype(F)).astype(F)
    w1 = ((a1 * dx).astype(F) + (b1 * dy).astype(F)).astype(F)
    bx = (w0 * inv_area).astype(F)
    by = (w1 * inv_area).astype(F)
    bz = ((F(1.0) - bx).astype(F) - by).astype(F)
    return (bx >= ERR) & (by >= ERR) & (bz >= ERR)


def int_inside(v, xs, ys):
    (x1, y1), (x2, y2), (x3, y3) = [(int(a), int(b)) for a, b in v]
    area = (y2 - y3) * (x1 - x3) + (x3 - x2) * (y1 - y3)
    s = -1 if area < 0 else 1
    X, Y = np.meshgrid(xs.astype(np.int64), ys.astype(np.int64))
    e0 = s * ((y2 - y3) * (X - x3) + (x3 - x2) * (Y - y3))
    e1 = s * ((y3 - y1) * (X - x3) + (x1 - x3) * (Y - y3))
    e2 = abs(area) - e0 - e1
    return (e0 >= 0) & (e1 >= 0) & (e2 >= 0)


def rcp_model(v, ulps):
    r = F(F(1.0) / F(v))
    for _ in range(abs(ulps)):
        r = np.nextafter(r, F(np.inf) if ulps > 0 else F(-np.inf), dtype=F)
    return F(r)


def fma32(a, b, c):
    """Exactly rounded f32 fma for the magnitudes in play (|a b| < 2^51, c a small f32): the f64 sum is exact or off by far less
    than the distance to an f32 rounding boundary that could matter for the ceil / floor that follows."""
    return F(np.float64(a) * np.float64(b) + np.float64(c))


def span_edge(G, ulps):
    """Mirror of the device function."""
    G = F(G)
    if G == 0:
        return F(-1073741824.0), F(64.0)
    r = rcp_model(G, ulps)
    return (r, F(F(-0.5) * r)) if G > 0 else (r, F(F(F(-0.5) * r) + F(1.0)))


def span_rows(v, cx0, cx1, cy0, cy1, ulps):
    """Mirror of the kernel: per-surface part at the first pixel of the clipped box, per-row part by one fma per edge.  Returns
    a boolean grid [cy1 - cy0, cx1 - cx0]."""
    (x1, y1), (x2, y2), (x3, y3) = [(F(a), F(b)) for a, b in v]
    a0, b0, a1, b1 = F(y2 - y3), F(x3 - x2), F(y3 - y1), F(x1 - x3)
    area = F(F(a0 * F(x1 - x3)) + F(b0 * F(y1 - y3)))
    inv = F(F(1.0) / area)
    sgn = F(-1.0) if inv < 0 else F(1.0)
    G0, G1 = F(sgn * a0), F(sgn * a1)
    G2 = F(-F(G0 + G1))
    H0, H1 = F(sgn * b0), F(sgn * b1)
    A = F(abs(F(F(a0 * b1) - F(b0 * a1))))
    dx, dy = F(F(cx0) - x3), F(F(cy0) - y3)
    E0o = F(sgn * F(F(a0 * dx) + F(b0 * dy)))
    E1o = F(sgn * F(F(a1 * dx) + F(b1 * dy)))
    d = [span_edge(G, ulps) for G in (G0, G1, G2)]
    wlen = F(cx1 - cx0)
    out = np.zeros((cy1 - cy0, cx1 - cx0), dtype=bool)
    for row in range(cy1 - cy0):
        E0, E1 = fma32(H0, F(row), E0o), fma32(H1, F(row), E1o)
        E2 = F(F(A - E0) - E1)
        lo, hi = F(0.0), wlen
        for E, (r, c) in zip((E0, E1, E2), d):
            vv = fma32(-E, r, c)
            if r > 0:
                lo = max(lo, F(np.ceil(vv)))
            else:
                hi = min(hi, F(np.floor(vv)))
        n = int(hi - lo)
        if n > 0:
            out[row, int(lo):int(lo) + n] = True
    return out


def eligible(v):
    (x1, y1), (x2, y2), (x3, y3) = [(int(a), int(b)) for a, b in v]
    area = (y2 - y3) * (x1 - x3) + (x3 - x2) * (y1 - y3)
    ext = max(abs(y2 - y3), abs(x3 - x2), abs(y3 - y1), abs(x1 - x3), abs(y2 - y1), abs(x1 - x2))
    return 1 <= abs(area) <= MAX_AREA and ext <= MAX_EXT


def random_triangles(rng, n):
    out = []
    while len(out) < n:
        kind = rng.integers(0, 5)
        c = rng.integers(-200, 1200, size=2)
        if kind == 0:      # small blobs (the benchmark's shape)
            v = c + rng.integers(-8, 9, size=(3, 2))
        elif kind == 1:    # medium
            v = c + rng.integers(-45, 46, size=(3, 2))
        elif kind == 2:    # slivers: long and thin, area near the limit or near 1
            d = rng.integers(-500, 501, size=2)
            v = np.array([c, c + d, c + d // 2 + rng.integers(-3, 4, size=2)])
        elif kind == 3:    # axis-aligned edges (G == 0 rows / columns)
            w, h = rng.integers(1, 90, size=2)
            v = np.array([c, c + [w, 0], c + [rng.integers(0, w + 1), h]])
            if rng.integers(0, 2):
                v = v[:, ::-1]
        else:              # area exactly at or just below the limit
            w = int(rng.integers(1, 400))
            h = MAX_AREA // w
            v = np.array([c, c + [w, 0], c + [rng.integers(-20, 20), h]])
        v = v[rng.permutation(3)]
        if eligible(v):
            out.append(v)
    return out


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_toleranced_float_test_is_the_closed_integer_triangle(seed):
    rng = np.random.default_rng(seed)
    for v in random_triangles(rng, 400):
        x0, x1 = v[:, 0].min() - 6, v[:, 0].max() + 7
        y0, y1 = v[:, 1].min() - 6, v[:, 1].max() + 7
        # a window of at most 96 x 96 around a random part of the triangle's surroundings
        if x1 - x0 > 96:
            x0 = int(rng.integers(x0, x1 - 96)); x1 = x0 + 96
        if y1 - y0 > 96:
            y0 = int(rng.integers(y0, y1 - 96)); y1 = y0 + 96
        xs, ys = np.arange(x0, x1), np.arange(y0, y1)
        assert np.array_equal(float_inside(v, xs, ys), int_inside(v, xs, ys)), v


def test_area_limit_is_where_the_equivalence_ends():
    """Just beyond the limit the tolerance does admit pixels outside the integer triangle (which is why such surfaces keep the
    per-pixel form): a right triangle of |area| 10100 > 1 / 1e-4."""
    v = np.array([[0, 0], [101, 0], [0, 100]])          # |area| = 10100, coprime edge steps: an edge value of -1 exists
    xs, ys = np.arange(-3, 106), np.arange(-3, 104)
    assert not eligible(v)
    assert not np.array_equal(float_inside(v, xs, ys), int_inside(v, xs, ys))


@pytest.mark.parametrize("ulps", [-2, -1, 0, 1, 2])
def test_row_intervals_are_the_passing_set(ulps):
    rng = np.random.default_rng(100 + ulps)
    for v in random_triangles(rng, 160):
        # the clipped box the kernel walks: the reference's box (render.rs:1455-1458) cut to a random 64-wide tile
        bx0, bx1 = int(v[:, 0].min()), int(v[:, 0].max()) + 1
        by0, by1 = int(v[:, 1].min()), int(v[:, 1].max()) + 1
        tx = int(rng.integers(bx0 - 40, bx1)); ty = int(rng.integers(by0 - 40, by1))
        cx0, cx1 = max(bx0, tx), min(bx1, tx + 64)
        cy0, cy1 = max(by0, ty), min(by1, ty + 64)
        if cx0 >= cx1 or cy0 >= cy1:
            continue
        want = float_inside(v, np.arange(cx0, cx1), np.arange(cy0, cy1))
        got = span_rows(v, cx0, cx1, cy0, cy1, ulps)
        assert np.array_equal(got, want), (v, cx0, cx1, cy0, cy1)

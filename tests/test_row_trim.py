"""The row trimming of the coverage walk (b32_fill.hip: row_trim) must never exclude a pixel that passes the reference's inside
test (render.rs:1536-1542, evaluated on rounded f32 values).  This restates both in numpy f32 and checks the containment on
random, thin, huge and degenerate integer triangles, with the hardware's approximate reciprocal modelled as the exact one moved
by up to +-2 ulp (v_rcp_f32 is specified to 1 ulp)."""
import numpy as np
import pytest

F = np.float32
ERR = F(-0.0001)


def inside_row(w0, w1, a0, a1, inv_area, n):
    """The reference's walk along one row: w += a per pixel (exact integers here), bc = w * inv_area, bc_z = 1 - bc_x - bc_y."""
    x = np.arange(n, dtype=np.float32)
    W0 = (w0 + a0 * x).astype(F)
    W1 = (w1 + a1 * x).astype(F)
    bx = (W0 * inv_area).astype(F)
    by = (W1 * inv_area).astype(F)
    bz = ((F(1.0) - bx).astype(F) - by).astype(F)
    return (bx >= ERR) & (by >= ERR) & (bz >= ERR)


def rcp_model(v, ulps):
    r = (F(1.0) / v).astype(F) if isinstance(v, np.ndarray) else F(F(1.0) / v)
    for _ in range(abs(ulps)):
        r = np.nextafter(r, F(np.inf) if ulps > 0 else F(-np.inf), dtype=F)
    return F(r)


def row_trim(w0, w1, a0, a1, inv_area, n, ulps):
    """Mirror of the device function (same f32 operation order)."""
    A = rcp_model(F(abs(inv_area)), ulps)
    if not (A >= F(0.5) and A < F(1048576.0)):
        return 0, n
    s = F(-1.0) if inv_area < 0 else F(1.0)
    T = F(F(1.02e-4) * A)
    E = [F(F(s * w0) + T), F(F(s * w1) + T), F(F(A + T) - F(s * F(w0 + w1)))]
    G = [F(s * a0), F(s * a1), F(-F(F(s * a0) + F(s * a1)))]
    flo, fhi = F(0.0), F(n)
    for e, g in zip(E, G):
        if g > 0:
            r = F(F(-e) * rcp_model(g, ulps))
            flo = max(flo, F(np.ceil(F(r - F(0.01)))))
        elif g < 0:
            r = F(F(-e) * rcp_model(g, ulps))
            fhi = min(fhi, F(F(np.floor(F(r + F(0.01)))) + F(1.0)))
        elif e < 0:
            fhi = F(0.0)
    flo = min(flo, F(n))
    fhi = max(fhi, flo)
    return int(flo), int(fhi)


def tri_rows(v, width=4096, height=4096):
    """Per-row parameters exactly as k_setup / phase_a_rows derive them for an integer triangle; None if culled or not closed-form."""
    (x1, y1), (x2, y2), (x3, y3) = [(F(a), F(b)) for a, b in v]
    area = F(F((y2 - y3) * (x1 - x3)) + F((x3 - x2) * (y1 - y3)))
    if abs(area) < F(0.00001):
        return None
    inv_area = F(F(1.0) / area)
    a0, b0, a1, b1 = F(y2 - y3), F(x3 - x2), F(y3 - y1), F(x1 - x3)
    min_x = int(max(min(x1, x2, x3), 0)); max_x = int(min(max(x1, x2, x3) + 1, width))
    min_y = int(max(min(y1, y2, y3), 0)); max_y = int(min(max(y1, y2, y3) + 1, height))
    if min_x >= max_x or min_y >= max_y:
        return None
    # the closed-form guard of k_setup (every product / sum an exact integer below 2^24 over the bbox)
    for dx in (F(min_x) - x3, F(max_x - 1) - x3):
        for dy in (F(min_y) - y3, F(max_y - 1) - y3):
            vals = [a0 * dx, b0 * dy, a1 * dx, b1 * dy, a0 * dx + b0 * dy, a1 * dx + b1 * dy]
            if not all(abs(float(t)) < 2 ** 24 for t in vals):
                return None
    rows = []
    for y in range(min_y, max_y):
        dx, dy = F(min_x) - x3, F(y) - y3
        rows.append((F(F(a0 * dx) + F(b0 * dy)), F(F(a1 * dx) + F(b1 * dy))))
    return a0, a1, inv_area, max_x - min_x, rows


def gen_triangles(rng, count):
    out = []
    for i in range(count):
        kind = i % 6
        if kind == 0:    # small, like the benchmark
            c = rng.integers(0, 2000, 2); v = [c + rng.integers(-8, 9, 2) for _ in range(3)]
        elif kind == 1:  # thin slivers: huge barycentrics inside the bbox
            c = rng.integers(0, 2000, 2); d = rng.integers(-60, 61, 2)
            v = [c, c + d, c + d + rng.integers(-1, 2, 2)]
        elif kind == 2:  # large
            v = [rng.integers(-500, 4500, 2) for _ in range(3)]
        elif kind == 3:  # areas around the 1e-4 * area = 0.5, 1 thresholds (area ~ 5000 .. 20000)
            c = rng.integers(0, 2000, 2); v = [c, c + [rng.integers(60, 200), rng.integers(-3, 4)], c + [rng.integers(-3, 4), rng.integers(60, 200)]]
        elif kind == 4:  # axis-aligned edges (zero coefficients)
            c = rng.integers(0, 2000, 2); w, h = rng.integers(1, 40, 2); v = [c, c + [w, 0], c + [0, h]]
        else:            # partly off-screen
            v = [rng.integers(-300, 300, 2) for _ in range(3)]
        if rng.random() < 0.5:
            v = [v[0], v[2], v[1]]      # both windings (negative inv_area: rendered back-faces)
        out.append([(int(p[0]), int(p[1])) for p in v])
    return out


@pytest.mark.parametrize("ulps", [-2, 0, 2])
def test_trim_never_drops_a_passing_pixel(ulps):
    rng = np.random.default_rng(1234 + ulps)
    rows_total = kept = passing = bbox = 0
    for v in gen_triangles(rng, 1500):
        t = tri_rows(v)
        if t is None:
            continue
        a0, a1, inv_area, n, rows = t
        n = min(n, 64)                                   # rows are clipped to a 64-pixel tile
        for (w0, w1) in rows[:96]:
            ins = inside_row(w0, w1, a0, a1, inv_area, n)
            lo, hi = row_trim(w0, w1, a0, a1, inv_area, n, ulps)
            assert 0 <= lo <= hi <= n
            outside = np.ones(n, bool); outside[lo:hi] = False
            assert not (ins & outside).any(), (v, float(w0), float(w1), lo, hi, np.nonzero(ins)[0])
            rows_total += 1; kept += hi - lo; passing += int(ins.sum()); bbox += n
    assert rows_total > 10000
    # and it must be worth having: the kept interval is far smaller than the bbox row and close to the passing pixels
    assert kept < 0.75 * bbox and kept < 1.6 * passing + rows_total


def test_trim_is_tight_on_small_triangles():
    rng = np.random.default_rng(7)
    kept = passing = 0
    for v in gen_triangles(rng, 1200)[0::6]:
        t = tri_rows(v)
        if t is None:
            continue
        a0, a1, inv_area, n, rows = t
        for (w0, w1) in rows:
            ins = inside_row(w0, w1, a0, a1, inv_area, n)
            lo, hi = row_trim(w0, w1, a0, a1, inv_area, n, 0)
            kept += hi - lo; passing += int(ins.sum())
    assert kept == passing          # area < 4900: the three conditions are exactly the integer tests w >= 0

"""Verbose GPU-vs-oracle diagnostics (development aid; the graded tests live in tests/)."""
import hashlib
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import __graft_entry__ as g  # noqa: E402

g.build()
from bonnie32_amd import rasterizer as R, scenegen  # noqa: E402
from oracle import oracle as O  # noqa: E402

ctx = R.Context(0)
ctx.set_fragment_counting(1)


def check(name, sc, resident=False, indexed=False):
    ofb = O.Framebuffer(sc.width, sc.height); ofb.clear(sc.clear_color)
    t0 = time.time()
    rc, ot, dump = O.render_mesh_15(ofb, sc.vertices, sc.faces, sc.textures, sc.camera, sc.settings, sc.fog, dump=True)
    t_cpu = time.time() - t0
    fb = R.Framebuffer(sc.width, sc.height, ctx); fb.clear(sc.clear_color)
    t0 = time.time()
    try:
        if resident:
            rs = R.ResidentScene(fb, sc.vertices, sc.faces, sc.textures if not indexed else None,
                                 sc.indexed_textures if indexed else None)
            t = rs.render(sc.camera, sc.settings, sc.fog)
        else:
            t = R.render_mesh_15(fb, sc.vertices, sc.faces, sc.textures, sc.camera, sc.settings, sc.fog)
    except R.B32Error as e:
        print(f"[{name}] GPU error {e} (oracle rc={rc})")
        return False
    t_gpu = time.time() - t0
    kern_exact = ctx.last_kernel_times()
    got = fb.image(); exp = ofb.image()
    # same frame with fragment counting off (CHEAP coverage + repair where eligible): identical pixels required
    ctx.set_fragment_counting(0)
    fb2 = R.Framebuffer(sc.width, sc.height, ctx); fb2.clear(sc.clear_color)
    if resident:
        rs2 = R.ResidentScene(fb2, sc.vertices, sc.faces, sc.textures if not indexed else None, sc.indexed_textures if indexed else None)
        rs2.render(sc.camera, sc.settings, sc.fog)
    else:
        R.render_mesh_15(fb2, sc.vertices, sc.faces, sc.textures, sc.camera, sc.settings, sc.fog)
    cheap_kern = ctx.last_kernel_times()
    cheap_bad = int((fb2.image() != exp).any(axis=2).sum())
    ctx.set_fragment_counting(1)
    diff = (got != exp).any(axis=2)
    order = ctx.last_draw_order(len(sc.faces))
    order_ok = np.array_equal(order, dump["draw_order"])
    ok = (not diff.any()) and order_ok and t.triangles_drawn == ot.triangles_drawn and t.fragments == ot.fragments and cheap_bad == 0
    print(f"[{name}] {'OK ' if ok else 'BAD'} px_mismatch={int(diff.sum())} order_ok={order_ok} drawn gpu/cpu={t.triangles_drawn}/{ot.triangles_drawn} "
          f"frags gpu/cpu={t.fragments}/{ot.fragments} cpu={t_cpu*1e3:.1f}ms gpu_call={t_gpu*1e3:.1f}ms "
          f"cheap_px_mismatch={cheap_bad} fill exact/cheap ms={kern_exact.get('fill', 0):.3f}/{cheap_kern.get('fill', 0):.3f}")
    if diff.any():
        ys, xs = np.nonzero(diff)
        for i in range(min(5, len(ys))):
            print(f"    ({xs[i]},{ys[i]}) gpu={got[ys[i], xs[i]]} cpu={exp[ys[i], xs[i]]}")
        print(f"    bbox of mismatches x[{xs.min()},{xs.max()}] y[{ys.min()},{ys.max()}]")
    if not order_ok:
        n = min(len(order), len(dump["draw_order"]))
        bad = np.nonzero(order[:n] != dump["draw_order"][:n])[0]
        print(f"    order len gpu/cpu {len(order)}/{len(dump['draw_order'])} first bad idx {bad[:5]}")
    return ok


# 0. device f32 semantics
rng = np.random.default_rng(1)
a = rng.standard_normal(4096).astype(np.float32) * np.float32(1e3)
b = rng.standard_normal(4096).astype(np.float32)
c = rng.standard_normal(4096).astype(np.float32) * np.float32(1e-3)
a[:8] = [1e-40, 3e-39, 1.0, 16777216.0, 1e38, -1e-45, 0.1, 3.0]
b[:8] = [0.5, 0.25, 3.0, 1.0, 10.0, 0.5, 0.2, 7.0]
print("selftest mul+add", np.array_equal(ctx.selftest_f32(0, a, b, c), (a * b).astype(np.float32) + c),
      "div", np.array_equal(ctx.selftest_f32(1, a, b, c), a / b),
      "sqrt", np.array_equal(ctx.selftest_f32(2, np.abs(a), b, c), np.sqrt(np.abs(a))),
      "(a+b)/c", np.array_equal(ctx.selftest_f32(3, a, b, c), (a + b) / c))

# 1. project_fixed stage tap
sc = scenegen.make_scene("C2")
pos = sc.vertices["pos"][:50000]
sx, sy, z = ctx.project_fixed_batch(pos, sc.camera, sc.width, sc.height)
exp = np.array([O.project_fixed(p, sc.camera, sc.width, sc.height)[:2] for p in pos[:5000]])
print("project_fixed", np.array_equal(sx[:5000], exp[:, 0]), np.array_equal(sy[:5000], exp[:, 1]))

results = []
results.append(check("C1", scenegen.make_scene("C1")))
results.append(check("cube", __import__("tests.golden.ref_fixtures", fromlist=["cube_scene"]).cube_scene()))
results.append(check("C1-resident-indexed", scenegen.make_scene("C1"), resident=True, indexed=True))
results.append(check("C1-gouraud", scenegen.make_scene("C1", variant="gouraud")))
results.append(check("C1-blend", scenegen.make_scene("C1", variant="blend")))
results.append(check("C1-float", scenegen.make_scene("C1", variant="float")))
results.append(check("C2", scenegen.make_scene("C2")))
results.append(check("C2-blend", scenegen.make_scene("C2", variant="blend")))
if "--big" in sys.argv:
    results.append(check("C3-100k", scenegen.make_scene("C3", n_tris=100_000), resident=True))
    results.append(check("C3", scenegen.make_scene("C3"), resident=True))
    results.append(check("C5-50k", scenegen.make_scene("C5", n_tris=50_000), resident=True))
print("ALL OK" if all(results) else "FAILURES", results)

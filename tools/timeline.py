"""Per-tile timeline of the fused coverage kernel (needs an experiment build with -DB32_TIMELINE: tools/exp_variants.py build tl -DB32_TIMELINE,
then on the GPU box  B32_LIB=bonnie-32_amd/csrc/exp_tl.so python tools/timeline.py [C3|C5]).  Prints how busy the 512 workgroup slots are over
the kernel's duration and the split coverage / shading per tile."""
import ctypes as C, os, sys
sys.path.insert(0, ".")
import numpy as np
from bonnie32_amd import rasterizer as R, scenegen
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
sc = scenegen.make_scene(cfg)
ctx = R.Context(0); ctx.set_async_depth(1)
fb = R.Framebuffer(sc.width, sc.height, ctx)
rs = R.ResidentScene(fb, sc.vertices, sc.faces, indexed_textures=sc.indexed_textures)
for i in range(6):
    fb.clear(sc.clear_color); rs.render_async(sc.camera, sc.settings)
rs.finish()
buf = np.zeros(1 + 4 * 8192 + 64, np.uint64)
lib = ctx.lib
lib.b32_debug_timeline.restype = C.c_int
assert lib.b32_debug_timeline(C.c_void_p(buf.ctypes.data), C.c_uint(buf.size)) == 1
n = int(buf[0]); e = buf[1:1 + 4 * n].reshape(n, 4)
wg = (e[:, 0] >> np.uint64(32)) & np.uint64(0xFFFF); tile = e[:, 0] & np.uint64(0xFFFFFFFF); nop = e[:, 0] >> np.uint64(48)
t0 = e[:, 1].astype(np.int64); t1 = e[:, 2].astype(np.int64); t2 = e[:, 3].astype(np.int64)
base = t0.min(); t0 -= base; t1 -= base; t2 -= base
tick = 0.01    # wall_clock64: 100 MHz -> 10 ns
print(f"{cfg}: {n} tiles, kernel span {t2.max() * tick:.1f} us; per tile: coverage {np.mean(t1 - t0) * tick:.1f} us, shading {np.mean(t2 - t1) * tick:.1f} us, "
      f"total {np.mean(t2 - t0) * tick:.1f} (min {np.min(t2 - t0) * tick:.1f}, max {np.max(t2 - t0) * tick:.1f}) us; list entries per tile {nop.mean():.0f}")
per_wg = {}
for w, a, b in zip(wg.tolist(), t0.tolist(), t2.tolist()):
    per_wg.setdefault(w, []).append((a, b))
cnt = np.bincount([len(v) for v in per_wg.values()])
print("tiles per workgroup:", {i: int(c) for i, c in enumerate(cnt) if c})
end = np.array([max(b for a, b in v) for v in per_wg.values()]) * tick
print(f"workgroup finish times: min {end.min():.1f}, median {np.median(end):.1f}, max {end.max():.1f} us")
span = t2.max()
for lo in range(0, 100, 10):
    a, b = span * lo // 100, span * (lo + 10) // 100
    busy = sum(max(0, min(y, b) - max(x, a)) for v in per_wg.values() for x, y in v)
    print(f"  {lo:3d}-{lo + 10:3d} % of the kernel: {busy / ((b - a) * 512) * 100:5.1f} % of the 512 slots in a tile")
# order of tile duration vs start time
order = np.argsort(t0)
third = n // 3
for k, name in enumerate(("first", "middle", "last")):
    sel = order[k * third:(k + 1) * third]
    print(f"  {name} third of the tiles (by start): coverage {np.mean((t1 - t0)[sel]) * tick:.1f} us, shading {np.mean((t2 - t1)[sel]) * tick:.1f} us")
# ---- what makes a tile slow?
dur = (t2 - t0) * tick
print("tile time percentiles (us):", {p: round(float(np.percentile(dur, p)), 1) for p in (1, 10, 25, 50, 75, 90, 99, 100)})
print("by XCD (wg % 8):", [round(float(dur[(wg % np.uint64(8)) == np.uint64(x)].mean()), 1) for x in range(8)])
tx = (tile % np.uint64((sc.width + 63) // 64)).astype(np.int64); tyr = (tile // np.uint64((sc.width + 63) // 64)).astype(np.int64)
print("by tile column (8 groups):", [round(float(dur[(tx * 8 // ((sc.width + 63) // 64)) == g].mean()), 1) for g in range(8)])
print("by tile row (8 groups):", [round(float(dur[(tyr * 8 // (tyr.max() + 1)) == g].mean()), 1) for g in range(8)])
print("corr(time, list entries) =", round(float(np.corrcoef(dur, nop.astype(np.float64))[0, 1]), 3),
      " entries percentiles:", {p: int(np.percentile(nop, p)) for p in (1, 50, 99, 100)})
# order within the workgroup
kth = np.zeros(n, np.int64)
seen = {}
for i in np.argsort(t0):
    w = int(wg[i]); kth[i] = seen.get(w, 0); seen[w] = kth[i] + 1
print("by position in the workgroup's sequence:", {int(k): (round(float(dur[kth == k].mean()), 1), int((kth == k).sum())) for k in np.unique(kth)})
slow = np.argsort(dur)[-12:]
print("slowest tiles: (tile, wg, start us, cover us, shade us, entries)")
for i in slow:
    print("   ", int(tile[i]), int(wg[i]), round(t0[i] * tick, 1), round((t1[i] - t0[i]) * tick, 1), round((t2[i] - t1[i]) * tick, 1), int(nop[i]))
# same-CU partner: workgroups b and b + 256 share a CU if dispatch is breadth-first
# ---- gaps between consecutive tiles of one workgroup (shaded stamp of tile k -> coverage start of tile k + 1): the per-tile fixed cost
gaps = []
for w, v in per_wg.items():
    v = sorted(v)
    gaps += [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
first = [min(a for a, b in v) for v in per_wg.values()]
if gaps:
    g = np.array(gaps) * tick
    print(f"gap between a workgroup's tiles: mean {g.mean():.2f} us, median {np.median(g):.2f}, p90 {np.percentile(g, 90):.2f}; first tile starts {np.mean(first) * tick:.2f} us after the earliest")

# ---- sub-phase shader-clock sums (per wave; -DB32_TIMELINE builds): averages per tile and per wave
acc = buf[1 + 4 * 8192:].astype(np.float64)
ghz = float(os.environ.get("B32_GHZ", "2.35"))
if acc.sum() > 0:
    tiles, waves = float(n), 8.0
    us = lambda cyc: cyc / (ghz * 1e3)
    b, rounds, drains, rows = acc[4], acc[5], acc[6], acc[7]
    print(f"coverage, per wave and tile (us at {ghz} GHz): batches {b / tiles / waves:.2f}, rounds {rounds / tiles / waves:.2f}, drains {drains / tiles / waves:.2f}, row items {rows / tiles / waves:.1f}")
    print(f"   grab -> record arrived {us(acc[0]) / tiles / waves:.2f}   setup + scan {us(acc[1]) / tiles / waves:.2f}   rounds {us(acc[2]) / tiles / waves:.2f}"
          f"   drains {us(acc[3]) / tiles / waves:.2f}   wait at the barrier {us(acc[8]) / tiles / waves:.2f}   tile header {us(acc[9]) / tiles / waves:.2f}")
    if rounds: print(f"   per round {us(acc[2]) / rounds:.3f} us, per drain {us(acc[3]) / max(drains, 1):.3f} us, per batch load {us(acc[0]) / max(b, 1):.3f} us")
    steps = acc[20]
    if steps:
        print(f"shading, per step ({steps / tiles / waves:.2f} steps per wave and tile): winners + records arrived {us(acc[16]) / steps:.3f}   address math {us(acc[17]) / steps:.3f}"
              f"   texels arrived {us(acc[18]) / steps:.3f}   colour + stores {us(acc[19]) / steps:.3f} us")

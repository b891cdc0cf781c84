"""Timeline of two frames in flight: `run` renders N resident frames back to back (meant to run under rocprofv3 --kernel-trace),
`show results.db` prints start / end of the last frames' kernels relative to the first of them (us) -- which setup kernel ran beside
which fill.  usage: pipeline_trace.py run CONFIG GATE_PERMILLE [ROUTES_OFF] | pipeline_trace.py show results.db [n_kernels]"""
import os
import sqlite3
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(cfg, gate, routes):
    from bonnie32_amd import rasterizer as R, scenegen
    sc = scenegen.make_scene(cfg)
    ctx = R.Context(0); ctx.set_async_depth(1)
    ctx.set_pipeline_gate(gate)
    if os.environ.get("EXP_DEPTH"):
        ctx.set_pipeline_depth(int(os.environ["EXP_DEPTH"]))
    if routes:
        ctx.set_routes(routes)
    fb = R.Framebuffer(sc.width, sc.height, ctx)
    rs = R.ResidentScene(fb, sc.vertices, sc.faces, indexed_textures=sc.indexed_textures)
    for _ in range(3):
        fb.clear(sc.clear_color); rs.render_async(sc.camera, sc.settings); rs.finish()
    n = int(os.environ.get("TRACE_FRAMES", "24"))
    import time
    for rep in range(2):                 # (the second region is the one to read: `show` prints the last kernels)
        ctx.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            fb.clear(sc.clear_color); rs.render_async()
        ctx.synchronize(); t1 = time.perf_counter()
        print(f"host: {n} frames in {(t1 - t0) * 1e6:.1f} us = {(t1 - t0) / n * 1e6:.2f} us per frame")
        rs.finish()


def show(path, n):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else ("kernel_name" if "kernel_name" in cols else cols[0])
    extra = [c for c in ("queue_id", "stream_id", "queue") if c in cols]
    rows = list(cur.execute(f"select {name_col}, start, end{''.join(', ' + e for e in extra)} from kernels order by start"))
    rows = rows[-n:]
    t0 = rows[0][1]
    for r in rows:
        nm = r[0].split("(")[0]
        nm = nm if len(nm) < 60 else nm[:57] + "..."
        print(f"{(r[1] - t0) / 1e3:9.1f} {(r[2] - t0) / 1e3:9.1f}  {(r[2] - r[1]) / 1e3:7.1f} us  {' '.join(str(x) for x in r[3:])}  {nm}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 0)
    else:
        show(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 24)

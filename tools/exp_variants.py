"""Experiment harness: times several builds of the library against each other on one GPU box.
  here  : python tools/exp_variants.py build <tag> [-DFLAG ...]     -> bonnie-32_amd/csrc/exp_<tag>.so (git-ignored, travels with gpurun)
  on GPU: python tools/exp_variants.py run [tag ...]                 (default: every exp_*.so + the product build as "base")
Each variant runs in its own process (B32_LIB): C3 and C5, default path, 200 / 100 frames back to back, per-phase HIP-event times,
frame hash against tests/golden/hashes.json."""
import glob
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "bonnie-32_amd", "csrc")


def one():
    import numpy as np
    from bonnie32_amd import rasterizer as R, scenegen
    H = json.load(open(os.path.join(ROOT, "tests", "golden", "hashes.json")))
    out = {}
    for cfg, n in (("C3", 200), ("C5", 100)):
        sc = scenegen.make_scene(cfg)
        ctx = R.Context(0); ctx.set_async_depth(1)
        if os.environ.get("EXP_GATE"):
            ctx.set_pipeline_gate(int(os.environ["EXP_GATE"]))
        if os.environ.get("EXP_DEPTH"):
            ctx.set_pipeline_depth(int(os.environ["EXP_DEPTH"]))
        if os.environ.get("EXP_ROUTES"):
            ctx.set_routes(int(os.environ["EXP_ROUTES"]))          # e.g. base@EXP_ROUTES=64: the product build without two frames in flight
        if os.environ.get("EXP_SETTINGS"):                         # base@EXP_SETTINGS=game | default: the reference's RasterSettings::game() / ::default() (no golden hash: "ok" is false)
            from bonnie32_amd import rtypes
            sc.settings = rtypes.RasterSettings.game() if os.environ["EXP_SETTINGS"] == "game" else rtypes.RasterSettings()
        fb = R.Framebuffer(sc.width, sc.height, ctx)
        rs = R.ResidentScene(fb, sc.vertices, sc.faces, indexed_textures=sc.indexed_textures)
        def fin():
            try:
                rs.finish()
            except Exception as e:
                out.setdefault("errors", []).append(str(e)[:60])
        for i in range(5):
            fb.clear(sc.clear_color); rs.render_async(sc.camera, sc.settings); fin()
        best = 1e9
        for rep in range(3):
            ctx.synchronize(); t0 = time.perf_counter()
            for i in range(n):
                fb.clear(sc.clear_color); rs.render_async()
            fin(); best = min(best, (time.perf_counter() - t0) / n)
        ok = hashlib.sha256(fb.pixels).hexdigest() == H[cfg]["sha256"]
        ctx.set_profiling(2)
        for i in range(20):
            fb.clear(sc.clear_color); rs.render_async()
        try:
            rs.finish()
        except Exception as e:                       # (experiment builds that break the frame on purpose still report their kernel times)
            out.setdefault("errors", []).append(str(e)[:60])
        kt = ctx.last_kernel_times(); ctx.set_profiling(0)
        out[cfg] = {"ms": round(best * 1e3, 4), "ok": ok, "pipelined": ctx.route_counts().get("pipelined"), **{k: round(v * 1e3, 1) for k, v in kt.items()}}
    print(json.dumps(out))


def main():
    if sys.argv[1] == "build":
        from bonnie32_amd import build as B
        tag = sys.argv[2]
        B.build(force=True, out=os.path.join(CSRC, f"exp_{tag}.so"), extra=sys.argv[3:])
        print("built", tag, sys.argv[3:])
    elif sys.argv[1] == "one":
        one()
    else:
        tags = sys.argv[2:] or (["base"] + sorted(os.path.basename(f)[4:-3] for f in glob.glob(os.path.join(CSRC, "exp_*.so"))))
        for full in tags:                                  # "tag" or "tag@VAR=value@VAR2=value": environment switches of an experiment
            tag, *sets = full.split("@")
            env = dict(os.environ)
            for kv in sets:
                k, v = kv.split("=", 1); env[k] = v
            if tag != "base":
                env["B32_LIB"] = os.path.join(CSRC, f"exp_{tag}.so")
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=env, capture_output=True, text=True, timeout=600)
            line = r.stdout.strip().split("\n")[-1] if r.stdout.strip() else ("ERR " + r.stderr[-400:])
            print(f"{full:32s} {line}", flush=True)


if __name__ == "__main__":
    main()

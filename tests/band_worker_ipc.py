"""One band rank of tests/test_gpu_parity.py::test_band_ranks_share_one_gpu[*-ipc]: BASELINE config C4's exchange step through the C ABI
ONLY (include/b32raster.h "multi-GPU"; no torch, no torch.distributed in this process).  The parent (rank 0, the root) owns the
framebuffer and hands this process its 96-byte share; the worker maps it (b32_band_import), binds its band (b32_set_band), draws
frame after frame straight into the root's memory and publishes each (b32_band_publish); the root's stream waits on the epoch words.
Frames alternate between two scenes so that a stale band would be seen.

usage: band_worker_ipc.py <share file> <rank> <world> <n_tris_a> <n_tris_b> <frames> [acquire]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bonnie32_amd import rasterizer as R, scenegen          # noqa: E402
from bonnie32_amd.bands import band_rows                    # noqa: E402


def main():
    share = open(sys.argv[1], "rb").read()
    rank, world, na, nb, frames = (int(a) for a in sys.argv[2:7])
    use_acquire = len(sys.argv) > 7 and sys.argv[7] == "acquire"
    assert "torch" not in sys.modules
    ctx = R.Context(0)
    ctx.set_async_depth(1)
    W, H = ctx.band_import(share, rank)
    fb = R.Framebuffer.__new__(R.Framebuffer)
    fb.ctx = ctx; fb.width, fb.height = W, H
    y0, y1 = band_rows(H, world, rank)
    fb.set_band(y0, y1)
    scenes = []
    for n in (na, nb):
        sc = scenegen.make_scene("C3", n_tris=n, width=W, height=H)
        scenes.append((sc, R.ResidentScene(fb, sc.vertices, sc.faces, indexed_textures=sc.indexed_textures).detach()))
    for f in range(1, frames + 1):
        sc, rs = scenes[f % 2]
        if use_acquire and f > 1:
            ctx.band_acquire(f - 1)                # the root has consumed the previous frame: its rows may be overwritten
        fb.clear(sc.clear_color)
        rs.render_async(sc.camera, sc.settings, sc.fog)
        ctx.band_publish(f)
        if not use_acquire:
            # lock-step with the parent through the filesystem-free channel we have: wait until the root has RELEASED this frame
            # (host poll of the epoch words) before drawing the next one
            t0 = time.time()
            while ctx.band_status()[1] < f:
                if time.time() - t0 > 60:
                    raise SystemExit(f"rank {rank}: root never released frame {f}")
                time.sleep(0.0005)
    tm = scenes[frames % 2][1].finish()
    ctx.synchronize()
    print(f"BAND_IPC_WORKER_OK rank={rank} rows={y0}:{y1} drawn={tm.triangles_drawn}", flush=True)
    ctx.band_close()


if __name__ == "__main__":
    main()

"""One rank of the multi-process GPU band test (tests/test_gpu_parity.py::test_band_ranks_share_one_gpu): the data path of BASELINE
config C4 -- every rank binds the real HIP context to its screen band (b32_set_band), renders the resident scene into its rows of a
device tensor, and the rows travel to rank 0 through bonnie32_amd.parallel -- with the ranks as separate PROCESSES sharing GPU 0
and gloo as the transport (band rows staged through the host: gloo has no device gather; RCCL itself needs one GPU per rank).
Launched by torch.distributed.run; rank 0 checks every assembled frame against the CPU oracle and prints BAND_WORKER_OK."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np      # noqa: E402
import torch            # noqa: E402
import torch.distributed as dist   # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    from bonnie32_amd import rasterizer as R, scenegen, parallel
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    for n_tris in (100_000, None):
        sc = scenegen.make_scene("C3", n_tris=n_tris)
        W, H = sc.width, sc.height
        y0, y1 = parallel.band_rows(H, world, rank)
        want = None
        if rank == 0:
            from oracle import oracle as O
            ofb = O.Framebuffer(W, H); ofb.clear(sc.clear_color)
            rc, otm = O.render_mesh_15(ofb, sc.vertices, sc.faces, sc.textures, sc.camera, sc.settings, sc.fog, fast=True)
            assert rc == 0
            want = ofb.pixels
        sets = []
        for k in range(2):                      # two framebuffers / contexts, like bench.py's pipelined gather
            ctx = R.Context(0)
            ctx.set_async_depth(1)
            ctx.set_stream(stream.cuda_stream)
            frame = torch.full((W * H * 4,), 0xAB, dtype=torch.uint8, device=dev)       # rows of other ranks are poisoned
            fb = R.Framebuffer.__new__(R.Framebuffer)
            fb.ctx = ctx
            fb.bind_device(frame.data_ptr(), W, H)
            fb.set_band(y0, y1)
            rs = R.ResidentScene(fb, sc.vertices, sc.faces, indexed_textures=sc.indexed_textures)
            fb.clear(sc.clear_color); rs.render_async(sc.camera, sc.settings, sc.fog)
            tm = rs.finish()
            sets.append((ctx, fb, rs, frame))
        # ---- synchronous gather (parallel.gather_bands)
        ctx, fb, rs, frame = sets[0]
        for rep in range(2):
            fb.clear(sc.clear_color); rs.render_async()
            tm = rs.finish()
            torch.cuda.synchronize(dev)
            host = frame.cpu()
            parallel.gather_bands(host, W, H, world, rank)
            if rank == 0:
                got = host.numpy()
                assert tm.triangles_drawn == otm.triangles_drawn
                assert np.array_equal(got, want), f"synchronous gather: {int((got != want).sum())} bytes differ (tris={n_tris}, world={world})"
        # ---- pipelined: frames alternate between the two framebuffers, frame i's gather is waited for when its buffer is redrawn
        pending = [None, None]
        hosts = [None, None]

        def settle(k):
            if pending[k] is not None:
                pending[k][0].wait(); pending[k] = None
                if rank == 0:
                    got = hosts[k].numpy()
                    assert np.array_equal(got, want), f"pipelined gather: {int((got != want).sum())} bytes differ (tris={n_tris}, world={world})"
        for i in range(6):
            k = i % 2
            ctx, fb, rs, frame = sets[k]
            settle(k)
            fb.clear(sc.clear_color); rs.render_async()
            torch.cuda.synchronize(dev)
            hosts[k] = frame.cpu()
            pending[k] = parallel.gather_bands_async(hosts[k], W, H, world, rank)
        settle(0); settle(1)
        for ctx, fb, rs, frame in sets:
            rs.finish()
        if rank == 0:
            print(f"band_worker: tris={sc.n_tris} world={world} sha256={hashlib.sha256(want).hexdigest()[:16]} ok", flush=True)
        for ctx, fb, rs, frame in sets:
            ctx.close()
        dist.barrier()
    if rank == 0:
        print("BAND_WORKER_OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

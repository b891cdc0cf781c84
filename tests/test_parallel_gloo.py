"""N > 1 path on CPU: world_size-2 gloo processes each render their screen band (the CPU oracle stands in for the
GPU band renderer) and gather the rows to rank 0; the assembled frame must equal the single-process frame."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, height, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bonnie32_amd import parallel, scenegen
    from oracle import oracle as O
    sc = scenegen.make_scene("C1", width=320, height=height)
    fb = O.Framebuffer(sc.width, sc.height); fb.clear(sc.clear_color)
    O.render_mesh_15(fb, sc.vertices, sc.faces, sc.textures, sc.camera, sc.settings)
    y0, y1 = parallel.band_rows(sc.height, world, rank)
    # a rank only owns its band: everything outside is poisoned so that a wrong gather shows up
    frame = torch.full((sc.width * sc.height * 4,), 0xAB, dtype=torch.uint8)
    row = sc.width * 4
    frame[y0 * row:y1 * row] = torch.from_numpy(fb.pixels[y0 * row:y1 * row].copy())
    parallel.gather_bands(frame, sc.width, sc.height, world, rank)
    if rank == 0:
        np.save(out_path, frame.numpy())
    # ---- the overlapped form bench.py uses at N > 1: frames alternate between two buffers, the gather of one is only waited for when
    # its buffer is about to be redrawn (two frames later); every assembled frame must equal the single-rank frame
    W, H = sc.width, sc.height
    if len({b[1] - b[0] for b in (parallel.band_rows(H, world, r) for r in range(world))}) == 1:
        bufs = [torch.zeros(W * H * 4, dtype=torch.uint8), torch.zeros(W * H * 4, dtype=torch.uint8)]
        pending = [None, None]

        def settle(k):
            if pending[k] is not None:
                pending[k][0].wait(); pending[k] = None
                if rank == 0:
                    assert np.array_equal(bufs[k].numpy(), fb.pixels), "overlapped frame differs"
        for i in range(5):
            k = i % 2
            settle(k)
            bufs[k].fill_(7 + i)                                   # "clear": stale rows must not survive
            bufs[k][y0 * row:y1 * row] = torch.from_numpy(fb.pixels[y0 * row:y1 * row].copy())
            pending[k] = parallel.gather_bands_async(bufs[k], W, H, world, rank)
        settle(0); settle(1)
    else:
        try:
            parallel.gather_bands_async(torch.zeros(W * H * 4, dtype=torch.uint8), W, H, world, rank)
            raise AssertionError("ragged bands must be refused")
        except ValueError:
            pass
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("height", [240, 241])      # equal bands (one gather into views) and ragged bands (padded)
def test_band_gather_world2(tmp_path, height):
    sys.path.insert(0, ROOT)
    from bonnie32_amd import scenegen
    from oracle import oracle as O
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, _free_port(), height, out), nprocs=2, join=True)
    sc = scenegen.make_scene("C1", width=320, height=height)
    fb = O.Framebuffer(sc.width, sc.height); fb.clear(sc.clear_color)
    O.render_mesh_15(fb, sc.vertices, sc.faces, sc.textures, sc.camera, sc.settings)
    assert np.array_equal(np.load(out), fb.pixels)

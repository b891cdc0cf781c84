"""Debug: N processes on one GPU, each renders its band like bench.py does; every rank checks its own rows against the oracle."""
import os, sys
sys.path.insert(0, ".")
import numpy as np, torch, torch.distributed as dist
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
from bonnie32_amd import rasterizer as R, scenegen, parallel
from oracle import oracle as O
sc = scenegen.make_scene("C3")
W, H = sc.width, sc.height
ofb = O.Framebuffer(W, H); ofb.clear(sc.clear_color)
O.render_mesh_15(ofb, sc.vertices, sc.faces, sc.textures, sc.camera, sc.settings)
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = R.Context(0)
ctx.set_async_depth(1)      # timing loops: frames back to back (a dropped frame would be reported by finish())
stream = torch.cuda.Stream(device=dev) if os.environ.get('OWN_TORCH_STREAM') else torch.cuda.current_stream(dev)
torch.cuda.set_stream(stream)
ctx.set_stream(stream.cuda_stream)
frame = torch.zeros(W * H * 4, dtype=torch.uint8, device=dev)
fb = R.Framebuffer.__new__(R.Framebuffer); fb.ctx = ctx
fb.bind_device(frame.data_ptr(), W, H)
y0, y1 = parallel.band_rows(H, world, rank)
fb.set_band(y0, y1)
rs = R.ResidentScene(fb, sc.vertices, sc.faces, indexed_textures=sc.indexed_textures)
for counting in (1, 0, 0):
    ctx.set_fragment_counting(counting)
    fb.clear(sc.clear_color); rs.render_async(sc.camera, sc.settings, sc.fog); tm = rs.finish()
    got = frame.cpu().numpy().reshape(H, W, 4)[y0:y1]
    exp = ofb.pixels.reshape(H, W, 4)[y0:y1]
    bad = (got != exp).any(axis=2)
    print(f"rank {rank} band [{y0},{y1}) counting={counting}: bad px {int(bad.sum())} tris {tm.triangles_drawn} pairs {tm.tile_pairs}", flush=True)
dist.barrier()

def step():
    fb.clear(sc.clear_color); rs.render_async()
    host = frame.cpu()
    parallel.gather_bands(host, W, H, world, rank)
    if rank == 0:
        frame.copy_(host)

def check(tag):
    if rank == 0:
        got = frame.cpu().numpy().reshape(H, W, 4); exp = ofb.pixels.reshape(H, W, 4)
        bad = (got != exp).any(axis=2); rows = np.nonzero(bad.any(axis=1))[0]
        print(f"rank 0 {tag}: bad px {int(bad.sum())} rows {rows.min() if len(rows) else '-'}..{rows.max() if len(rows) else '-'}", flush=True)
    dist.barrier()

for lvl in [int(x) for x in os.environ.get('LVLS', '0,1,2').split(',')]:
    ctx.set_profiling(lvl)
    for _ in range(3):
        step()
    tm = rs.finish()
    check(f"profiling {lvl}")

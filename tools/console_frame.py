"""A frame the way the console's render step issues it (scene.rs:112-261): Framebuffer::clear, then one render_mesh_15 call per room /
asset part onto the same 320x240 framebuffer (z-buffer, Gouraud + lights, fog), through the DROP-IN call with host slices -- per-call
upload, synchronous -- then the download of the 320x240 frame the presenter consumes.  GPU vs the CPU oracle, bit-exact check of the final frame."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
import bonnie32_amd as b32
from bonnie32_amd import rasterizer as R, scenegen
from oracle import oracle as O

n_meshes = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(2024)
meshes = []
for i in range(n_meshes):
    sc = scenegen.make_scene("C1", n_tris=int(rng.integers(300, 3000)), seed=1000 + i, variant=("blend" if i % 4 == 3 else "gouraud"),
                             bbox_px=float(rng.choice([150.0, 400.0, 900.0])))
    meshes.append(sc)
st = b32.RasterSettings.game()
st.lights = [b32.Light.directional((-1.0, -1.0, -1.0), 0.7), b32.Light.point((0.0, -100.0, 1500.0), 3000.0, 1.2)]
fog = (1500.0, 3000.0, 5800.0, b32.Color(40, 50, 70))
W, H = meshes[0].width, meshes[0].height
clear = b32.Color(10, 10, 30)

ofb = O.Framebuffer(W, H)
def cpu_frame():
    ofb.clear(clear)
    for sc in meshes:
        rc, _ = O.render_mesh_15(ofb, sc.vertices, sc.faces, sc.textures, sc.camera, st, fog)
        assert rc == 0
cpu_frame(); t0 = time.perf_counter()
for _ in range(5): cpu_frame()
t_cpu = (time.perf_counter() - t0) / 5

ctx = R.Context(0)
if os.environ.get("EXP_ROUTES"):
    ctx.set_routes(int(os.environ["EXP_ROUTES"]))
fb = R.Framebuffer(W, H, ctx)
def gpu_frame():
    fb.clear(clear)
    for sc in meshes:
        R.render_mesh_15(fb, sc.vertices, sc.faces, sc.textures, sc.camera, st, fog)
    return fb.pixels            # what the presenter hands to Texture2D::from_rgba8 (game/renderer.rs:179)
for _ in range(3): gpu_frame()
N = 50; t0 = time.perf_counter()
for _ in range(N): gpu_frame()
t_gpu = (time.perf_counter() - t0) / N
# the same frame with the rooms resident (one scene slot each, b32_scene_swap): no upload, no host sync between the meshes
slots = [R.ResidentScene(fb, sc.vertices, sc.faces, sc.textures).detach() for sc in meshes]
def gpu_frame_resident(first=False):
    fb.clear(clear)
    for sc, rs in zip(meshes, slots):
        if first: rs.render_async(sc.camera, st, fog)
        else: rs.render_async()
    slots[-1].finish()
    return fb.pixels
gpu_frame_resident(True); gpu_frame_resident()
t0 = time.perf_counter()
for _ in range(N): gpu_frame_resident()
t_res = (time.perf_counter() - t0) / N
ok_res = np.array_equal(fb.pixels, ofb.pixels) and np.array_equal(fb.zbuffer.view(np.uint32), ofb.zbuffer.view(np.uint32))
# runs of consecutive meshes WITHOUT a transparent pass merged into one resident mesh each: in z-buffer mode the result is identical
# (the depth test is order-independent, ties go to the first face in order exactly like the sequential strict `z < zbuffer`); a mesh
# with transparent faces must keep its place in the sequence (its blended pixels depend on what was drawn before it)
def merge(run):
    nv = 0; V = []; F = []; T = []
    for sc in run:
        f = sc.faces.copy(); f["v"] += nv
        f["texture_id"] = np.where(f["texture_id"] == 0xFFFFFFFF, 0xFFFFFFFF, f["texture_id"] + len(T)).astype(np.uint32)
        V.append(sc.vertices); F.append(f); T += list(sc.textures); nv += len(sc.vertices)
    return np.concatenate(V), np.concatenate(F), T
groups, run = [], []
for sc in meshes:
    if sc.name.split(":")[1] == "blend":
        if run: groups.append(merge(run)); run = []
        groups.append((sc.vertices, sc.faces, sc.textures))
    else:
        run.append(sc)
if run: groups.append(merge(run))
gslots = [R.ResidentScene(fb, v, f, t).detach() for v, f, t in groups]
def gpu_frame_merged(first=False):
    fb.clear(clear)
    for rs in gslots:
        if first: rs.render_async(meshes[0].camera, st, fog)
        else: rs.render_async()
    gslots[-1].finish()
    return fb.pixels
gpu_frame_merged(True); gpu_frame_merged()
t0 = time.perf_counter()
for _ in range(N): gpu_frame_merged()
t_mrg = (time.perf_counter() - t0) / N
ok_mrg = np.array_equal(fb.pixels, ofb.pixels) and np.array_equal(fb.zbuffer.view(np.uint32), ofb.zbuffer.view(np.uint32))
# the library's own batching (b32_frame_begin / _add_scene / _end): the same sequence of meshes, runs that commute drawn as one merged mesh
def gpu_frame_batched():
    fb.clear(clear)
    ctx.frame_begin(meshes[0].camera, st)
    for rs in slots:
        ctx.frame_add(rs, fog=fog)
    ctx.frame_end()
    ctx.finish()
    return fb.pixels
gpu_frame_batched(); gpu_frame_batched()
t0 = time.perf_counter()
for _ in range(N): gpu_frame_batched()
t_bat = (time.perf_counter() - t0) / N
ok_bat = np.array_equal(fb.pixels, ofb.pixels) and np.array_equal(fb.zbuffer.view(np.uint32), ofb.zbuffer.view(np.uint32))
bc = ctx.batch_counts()
# without the download of the frame (a presenter on the device: b32_present_nearest / a bound tensor)
def gpu_frame_batched_nodl():
    fb.clear(clear)
    ctx.frame_begin(meshes[0].camera, st)
    for rs in slots:
        ctx.frame_add(rs, fog=fog)
    ctx.frame_end()
    ctx.finish()
t0 = time.perf_counter()
for _ in range(N): gpu_frame_batched_nodl()
t_bat2 = (time.perf_counter() - t0) / N
# ... and without a host round trip per frame either (b32_set_async_depth(1): frames enqueued back to back, one b32_frame_finish at the end;
# what a device-side presenter sees -- the host enqueues frame i + 1 while the GPU draws frame i)
ctx.set_async_depth(1)
def gpu_frame_batched_async():
    fb.clear(clear)
    ctx.frame_begin(meshes[0].camera, st)
    for rs in slots:
        ctx.frame_add(rs, fog=fog)
    ctx.frame_end()
for _ in range(3): gpu_frame_batched_async()
ctx.finish()
t0 = time.perf_counter()
for _ in range(N): gpu_frame_batched_async()
ctx.finish()
t_bat3 = (time.perf_counter() - t0) / N
ok_bat3 = np.array_equal(fb.pixels, ofb.pixels) and np.array_equal(fb.zbuffer.view(np.uint32), ofb.zbuffer.view(np.uint32))
ctx.set_async_depth(0)
# ---- the console's render step as the reference runs it -- EVERY frame reaches host memory (game/renderer.rs:179-214) -- without a host
# round trip per frame: the whole frame's table in one call (b32_frame_submit), the copy into page-locked memory enqueued behind the last
# kernel (b32_fb_download_async), and the presenter waits for the ticket of the PREVIOUS frame while this one is drawn.  Library default
# (safe) mode: frames of small meshes never need settling.
table = ctx.make_frame_table(meshes[0].camera, st, slots, fogs=[fog] * len(slots))
nbytes = W * H * 4
bufs = [ctx.host_alloc(nbytes) for _ in range(2)]
tickets = [0, 0]
presented = []
def gpu_frame_ticketed(i, keep=False):
    fb.clear(clear)
    ctx.frame_submit(table)
    tickets[i & 1] = ctx.download_async(bufs[i & 1][1])
    if i > 0:
        ctx.ticket_wait(tickets[(i - 1) & 1])             # the presenter's frame: bufs[(i - 1) & 1]
        if keep: presented.append(bufs[(i - 1) & 1][0].copy())
for i in range(4): gpu_frame_ticketed(i)
ctx.ticket_wait(tickets[3 & 1]); ctx.finish()
t0 = time.perf_counter()
for i in range(N): gpu_frame_ticketed(i)
ctx.ticket_wait(tickets[(N - 1) & 1])
t_tick = (time.perf_counter() - t0) / N
ctx.finish()
for i in range(6): gpu_frame_ticketed(i, keep=True)
ctx.ticket_wait(tickets[5 & 1]); presented.append(bufs[5 & 1][0].copy()); ctx.finish()
ok_tick = len(presented) == 6 and all(np.array_equal(p, ofb.pixels) for p in presented)
for _, p in bufs: ctx.host_free(p)
gpu_frame()
ok = np.array_equal(fb.pixels, ofb.pixels) and np.array_equal(fb.zbuffer.view(np.uint32), ofb.zbuffer.view(np.uint32))
tris = sum(sc.n_tris for sc in meshes)
print(f"console frame: {n_meshes} meshes, {tris} triangles, {W}x{H}, game() + point light + fog: CPU oracle {t_cpu*1e3:.2f} ms, "
      f"GPU drop-in calls + frame download {t_gpu*1e3:.3f} ms ({t_cpu/t_gpu:.1f}x), {t_gpu/n_meshes*1e6:.0f} us per mesh, bit-exact: {ok}; "
      f"rooms resident in scene slots {t_res*1e3:.3f} ms ({t_cpu/t_res:.1f}x), {t_res/n_meshes*1e6:.0f} us per mesh, bit-exact: {ok_res}; "
      f"opaque runs merged ({len(groups)} draws) {t_mrg*1e3:.3f} ms ({t_cpu/t_mrg:.1f}x), bit-exact: {ok_mrg}; "
      f"b32_frame_begin/add_scene/end ({bc['merged_draws'] // (N + 2)} merged draws per frame) {t_bat*1e3:.3f} ms with the frame download, {t_bat2*1e3:.3f} ms without ({t_cpu/t_bat2:.1f}x), bit-exact: {ok_bat}; frames back to back without a host round trip each {t_bat3*1e3:.3f} ms ({t_cpu/t_bat3:.1f}x), bit-exact: {ok_bat3}; "
      f"EVERY frame delivered to page-locked host memory, one b32_frame_submit + b32_fb_download_async per frame, the presenter one frame behind (b32_ticket_wait) "
      f"{t_tick*1e3:.3f} ms per frame ({t_cpu/t_tick:.1f}x), all presented frames bit-exact: {ok_tick}")

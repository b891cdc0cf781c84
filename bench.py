#!/usr/bin/env python
"""bench.py — headline benchmark of the bonnie-32 rasterizer hot path on MI355X.

One step = one whole frame of the hot path over a synthetic scene whose inputs are already resident in HBM:
Framebuffer::clear + render_mesh_15 (vertex transform + snap, cull/setup, painter's sort, tile binning, textured
RGB555-dither fill) and, for N > 1, the gather of the screen bands to rank 0.
Workload at N = 1: BASELINE.json configs[2] "C3" — 2560x1920, 1M-triangle synthetic scene, 8-bit 256x256 atlas.

Prints ONE JSON line on rank 0 (see the driver contract), with `roofline` (dominant kernel vs the 8 TB/s HBM peak)
and `cpu_baseline` (the CPU oracle = C port of the reference, 1 thread, timed on this box's host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C3", help="scene config of bonnie32_amd.scenegen (C1,C2,C3,C5)")
    ap.add_argument("--tris", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--check", action="store_true", help="also verify the final frame against the oracle (slow at C3)")
    ap.add_argument("--weak-series", action="store_true",
                    help="also time the weak-scaling point of SURVEY 8e (N x 125 k tris on the same frame); always on for N > 1")
    ap.add_argument("--sync-gather", action="store_true",
                    help="N > 1: gather every frame before the next one starts (default: frames alternate between two framebuffers and the "
                         "gather of one overlaps the rendering of the next, after a run-time self-check against the synchronous frame)")
    ap.add_argument("--pipeline-debug", action="store_true",
                    help="with --dist-backend gloo: run the overlapped-gather logic too (band rows staged through the host)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (default). gloo = debug only: ranks may share one GPU, band rows are staged through the host")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the rasterizer has no CPU path")
    if args.dist_backend == "gloo":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    import __graft_entry__ as g
    if rank == 0:
        g.build()
    if world > 1:
        dist.barrier()
    from bonnie32_amd import rasterizer as R, scenegen, parallel

    sc = scenegen.make_scene(args.config, n_tris=args.tris)
    W, H, NF = sc.width, sc.height, sc.n_tris

    ctx = R.Context(local_rank)
    ctx.set_async_depth(1)      # frames back to back without a host synchronisation (static camera, capacities settled by the warm-up
                                # frames); a frame dropped for lack of buffer space would be REPORTED by finish(), never silent
    # one explicit stream for everything of this rank: the rasterizer's kernels, torch's copies and the RCCL gather are ordered by it
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    frame = torch.zeros(W * H * 4, dtype=torch.uint8, device=dev)       # the Framebuffer's pixels, in HBM
    fb = R.Framebuffer.__new__(R.Framebuffer)
    fb.ctx = ctx
    fb.bind_device(frame.data_ptr(), W, H)
    y0, y1 = parallel.band_rows(H, world, rank)
    fb.set_band(y0, y1)
    # inputs resident in HBM before the timed region (index atlas + CLUT are expanded on the device)
    rs = R.ResidentScene(fb, sc.vertices, sc.faces, indexed_textures=sc.indexed_textures)

    def step(first=False):
        fb.clear(sc.clear_color)
        if first:
            rs.render_async(sc.camera, sc.settings, sc.fog)
        else:
            rs.render_async()
        if world > 1:
            if args.dist_backend == "nccl":
                parallel.gather_bands(frame, W, H, world, rank)
            else:       # debug: same gather logic on a host copy
                host = frame.cpu()
                parallel.gather_bands(host, W, H, world, rank)
                if rank == 0:
                    frame.copy_(host)

    ctx.set_fragment_counting(1)          # warmup frames count the reference's pixel stores exactly (Mpixels/s numerator)
    # warmup (also settles buffer capacities: finish() grows the pair buffers if the first frame overflowed them)
    step(first=True)
    tm = rs.finish()
    for _ in range(max(args.warmup - 1, 0)):
        step()
    tm = rs.finish()

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    exact_fragments = tm.fragments        # counted exactly during warmup (fragment counting on)
    ctx.set_fragment_counting(0)          # instrumentation off for the timed region (identical framebuffer)
    step(); rs.finish()

    # ---- N > 1: overlap the band gather with the next frame.  Frames alternate between two framebuffers (two contexts on the same
    # stream, each with the scene resident); the gather of frame i is asynchronous (RCCL's own stream, started after frame i's kernels)
    # and is only waited for when its buffer is about to be redrawn, two frames later.  Every frame is still cleared, rendered and
    # gathered in full.  Guarded by a run-time self-check: pipelined frames must equal the synchronous frame on rank 0, on every
    # rank's agreement, otherwise the synchronous path is timed.
    sets = [(fb, rs, frame)]
    pending = [None, None]
    pipelined = False

    def make_second_set():
        ctx2 = R.Context(local_rank)
        ctx2.set_async_depth(1)
        ctx2.set_stream(stream.cuda_stream)
        frame2 = torch.zeros_like(frame)
        fb2 = R.Framebuffer.__new__(R.Framebuffer)
        fb2.ctx = ctx2
        fb2.bind_device(frame2.data_ptr(), W, H)
        fb2.set_band(y0, y1)
        rs2 = R.ResidentScene(fb2, sc.vertices, sc.faces, indexed_textures=sc.indexed_textures)
        return fb2, rs2, frame2

    def pstep(i, ssets, scene, first=False):
        k = i % len(ssets)
        fbk, rsk, frk = ssets[k]
        if pending[k] is not None:                      # this buffer's previous frame must have left before it is redrawn
            pending[k][0].wait(); pending[k] = None
        fbk.clear(scene.clear_color)
        if first:
            rsk.render_async(scene.camera, scene.settings, scene.fog)
        else:
            rsk.render_async()
        if args.dist_backend == "nccl":
            pending[k] = parallel.gather_bands_async(frk, W, H, world, rank)
        else:               # debug: the same alternation and waits, rows staged through a host copy (gloo has no device gather)
            host = frk.cpu()
            work, keep = parallel.gather_bands_async(host, W, H, world, rank)

            class _HostWait:
                def wait(self, work=work, host=host, frk=frk):
                    work.wait()
                    if rank == 0:
                        frk.copy_(host)
            pending[k] = (_HostWait(), keep)

    def pdrain():
        for k in range(2):
            if pending[k] is not None:
                pending[k][0].wait(); pending[k] = None

    if world > 1 and (args.dist_backend == "nccl" or args.pipeline_debug) and not args.sync_gather and H % world == 0:
        ok = 1
        try:
            sets.append(make_second_set())
            torch.cuda.synchronize(dev)
            ref = frame.clone() if rank == 0 else None            # the synchronous frame (assembled by the last step())
            for i in range(4):
                pstep(i, sets, sc, first=(i < 2))
            pdrain()
            sets[0][1].finish(); sets[1][1].finish()
            torch.cuda.synchronize(dev)
            if rank == 0 and not (torch.equal(sets[0][2], ref) and torch.equal(sets[1][2], ref)):
                ok = 0
        except Exception as e:                                    # noqa: BLE001 -- any failure means: time the synchronous path
            print(f"# rank {rank}: pipelined gather unavailable ({e!r}), timing the synchronous path", file=sys.stderr)
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=dev if args.dist_backend == "nccl" else torch.device("cpu"))
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        pipelined = bool(flag.item())
        if not pipelined:
            pending[0] = pending[1] = None
            sets = sets[:1]
            step(); rs.finish()

    # ---- timed region: exactly K steps, HIP events around the dominant kernel on the stream it runs on
    ctx.set_profiling(1)
    sync_all()
    t0 = time.perf_counter()
    if pipelined:
        for i in range(args.steps):
            pstep(i, sets, sc)
        pdrain()
    else:
        for _ in range(args.steps):
            step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    tm = rs.finish()
    if pipelined:
        sets[1][1].finish()
    cover_ms = ctx.last_kernel_times().get("cover", None)     # HIP events around k_cover on the stream it runs on
    ctx.set_profiling(0)

    rdev = dev if args.dist_backend == "nccl" else torch.device("cpu")
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=rdev)
    frags = torch.tensor([float(exact_fragments)], dtype=torch.float64, device=rdev)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        dist.all_reduce(frags, op=dist.ReduceOp.SUM)
    elapsed = float(elapsed.item())
    fragments = int(frags.item())
    ms_per_step = elapsed / args.steps * 1e3

    # per-phase device times (separate untimed pass, events around every phase)
    ctx.set_profiling(2)
    for _ in range(10):
        step()
    rs.finish()
    phases = ctx.last_kernel_times()
    ctx.set_profiling(0)

    # ---- weak-scaling point (north_star's >= 0.7 target): N x 125 k triangles on the same 2560x1920 frame, so every
    # rank's band keeps the fragments and binned triangles of the 1-GPU 125 k scene.  Reported beside the headline (which is the
    # fixed C3 scene, i.e. strong scaling, as `metric` states); raw times only, the driver derives efficiencies.
    final_frame = frame.cpu().numpy() if (args.check and rank == 0) else None      # the C3 frame, before the weak series redraws
    weak = None
    if world > 1 or args.weak_series:
        per_gpu = 125000
        wsc = scenegen.make_scene(args.config, n_tris=per_gpu * world)
        wrs = R.ResidentScene(fb, wsc.vertices, wsc.faces, indexed_textures=wsc.indexed_textures)

        def wstep(first=False):
            fb.clear(wsc.clear_color)
            if first:
                wrs.render_async(wsc.camera, wsc.settings, wsc.fog)
            else:
                wrs.render_async()
            if world > 1:
                if args.dist_backend == "nccl":
                    parallel.gather_bands(frame, W, H, world, rank)
                else:
                    host = frame.cpu()
                    parallel.gather_bands(host, W, H, world, rank)
                    if rank == 0:
                        frame.copy_(host)

        wsets = [(fb, wrs, frame)]
        if pipelined:       # the same overlap as the headline: the second framebuffer's context gets the weak scene too
            wsets.append((sets[1][0], R.ResidentScene(sets[1][0], wsc.vertices, wsc.faces, indexed_textures=wsc.indexed_textures), sets[1][2]))
            for i in range(2 + 2 * max(args.warmup - 1, 0)):
                pstep(i, wsets, wsc, first=(i < 2))
            pdrain()
            wsets[0][1].finish(); wsets[1][1].finish()
        else:
            wstep(first=True); wrs.finish()
            for _ in range(max(args.warmup - 1, 0)):
                wstep()
            wrs.finish()
        sync_all()
        w0 = time.perf_counter()
        if pipelined:
            for i in range(args.steps):
                pstep(i, wsets, wsc)
            pdrain()
        else:
            for _ in range(args.steps):
                wstep()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)
        w1 = time.perf_counter()
        wrs.finish()
        if pipelined:
            wsets[1][1].finish()
        wel = torch.tensor([w1 - w0], dtype=torch.float64, device=rdev)
        if world > 1:
            dist.all_reduce(wel, op=dist.ReduceOp.MAX)
        w_ms = float(wel.item()) / args.steps * 1e3
        # the 1-GPU point of the same series, on rank 0's GPU alone: 125 k triangles, whole frame, no gather
        one_ms = None
        if rank == 0:
            osc = scenegen.make_scene(args.config, n_tris=per_gpu)
            fb.set_band(0, H)
            ors = R.ResidentScene(fb, osc.vertices, osc.faces, indexed_textures=osc.indexed_textures)
            fb.clear(osc.clear_color); ors.render_async(osc.camera, osc.settings, osc.fog); ors.finish()
            for _ in range(max(args.warmup - 1, 0)):
                fb.clear(osc.clear_color); ors.render_async()
            ors.finish(); torch.cuda.synchronize(dev)
            o0 = time.perf_counter()
            for _ in range(args.steps):
                fb.clear(osc.clear_color); ors.render_async()
            ors.finish(); torch.cuda.synchronize(dev)
            one_ms = (time.perf_counter() - o0) / args.steps * 1e3
            fb.set_band(y0, y1)
        if world > 1:
            dist.barrier()
        if rank == 0:
            weak = {"scaling": "weak", "tris_per_gpu": per_gpu, "tris": per_gpu * world, "ms_per_step": round(w_ms, 5),
                    "value": round(per_gpu * world / (w_ms * 1e-3) / 1e6, 3), "unit": "Mtriangles/s",
                    "one_gpu_ms_per_step": round(one_ms, 5), "one_gpu_value": round(per_gpu / (one_ms * 1e-3) / 1e6, 3)}

    if rank == 0:
        mtri = NF / (ms_per_step * 1e-3) / 1e6
        mpix = fragments / (ms_per_step * 1e-3) / 1e6
        # ALGORITHMIC bytes.  Frame: SURVEY 8d B_alg.  Dominant kernel k_cover (coverage + shading of the
        # default fast path), per launch, every input counted once:
        #   per (surface, tile) pair: 4 B surface id + 4 B painter's key + 64 B of the surface record (edges, bbox, uv, flags)
        #   per surface: 16 B more of its record for shading (vertex colours)
        #   per band pixel: 4 B framebuffer write;  + the texture once
        tex_bytes = sum(t.width * t.height * 2 for t in sc.textures)
        alg_frame = 36 * len(sc.vertices) + 20 * NF + 16 * tm.triangles_drawn + 8 * W * H + tex_bytes   # SURVEY 8d B_alg
        roofline = None
        if cover_ms:
            alg_cover = 72 * tm.tile_pairs + 16 * tm.triangles_drawn + 4 * W * (y1 - y0) + tex_bytes
            ach = alg_cover / (cover_ms * 1e-3) / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tpath):
                try:
                    traffic = json.load(open(tpath)).get(f"{args.config}:k_cover") if world == 1 else None   # measured for the whole frame on 1 GPU
                except Exception:
                    traffic = None
            roofline = {"kernel": "k_cover", "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                        "kernel_ms": round(cover_ms, 4), "algorithmic_bytes": alg_cover, "units": {"tile_pairs": tm.tile_pairs, "surfaces": tm.triangles_drawn, "pixels": W * (y1 - y0)},
                        "frame_algorithmic_bytes": alg_frame,
                        "frame_frac": round(alg_frame / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
        cpu = None
        cpu_all = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle as O
            ofb = O.Framebuffer(W, H)
            reps, t_cpu = 0, 0.0
            while t_cpu < args.cpu_seconds and reps < 50:
                ofb.clear(sc.clear_color)
                c0 = time.perf_counter()
                rc, otm = O.render_mesh_15(ofb, sc.vertices, sc.faces, sc.textures, sc.camera, sc.settings, sc.fog)
                t_cpu += time.perf_counter() - c0
                reps += 1
            per = t_cpu / reps
            cpu = {"value": round(NF / per / 1e6, 4), "unit": "Mtriangles/s", "cores": 1, "kind": "port",
                   "mpixels_per_s": round(otm.fragments / per / 1e6, 3), "ms_per_frame": round(per * 1e3, 2),
                   "sample": f"{reps} full frames of the same {args.config} scene ({NF} tris @ {W}x{H}), oracle/b32_oracle.c, 1 thread"}
            # the "fair CPU" variant of SURVEY 8d beside it: the same port on all host cores, one row band per process (transform, cull
            # and sort replicated in every process, like on the GPU ranks; the reference itself is single-threaded)
            try:
                # (more processes are not always faster: on the GPU box 16 row bands take 342 ms, 32 take 783 -- the replicated part is
                # memory-bound -- so a few counts are tried and the fastest is the baseline)
                ncpu = os.cpu_count() or 1
                best = None
                for cand in sorted({max(1, min(c, ncpu, H)) for c in (8, 16, 32)}):
                    t_c, frame_c = O.render_all_cores(sc, cand, reps=2)
                    if best is None or t_c < best[0]:
                        best = (t_c, frame_c, cand)
                t_all, frame_all, cores = best
                cpu_all = {"value": round(NF / t_all / 1e6, 4), "unit": "Mtriangles/s", "cores": cores, "kind": "port",
                           "mpixels_per_s": round(otm.fragments / t_all / 1e6, 3), "ms_per_frame": round(t_all * 1e3, 2),
                           "identical_to_single_core_frame": bool(np.array_equal(frame_all, ofb.pixels)),
                           "sample": f"2 frames of the same scene, {cores} processes x one row band each (slowest band; fastest of 8 / 16 / 32 processes on {ncpu} host CPUs), oracle/b32_oracle.c"}
            except Exception as e:                                  # noqa: BLE001 -- an extra, never a reason to lose the bench line
                cpu_all = {"error": repr(e)}
        if args.check:
            from oracle import oracle as O
            cfb = O.Framebuffer(W, H); cfb.clear(sc.clear_color)
            O.render_mesh_15(cfb, sc.vertices, sc.faces, sc.textures, sc.camera, sc.settings, sc.fog)
            got = final_frame
            ok = np.array_equal(got, cfb.pixels)
            print("# parity vs oracle:", "bit-exact" if ok else "MISMATCH", file=sys.stderr)
            if not ok:
                bad = (got.reshape(H, W, 4) != cfb.pixels.reshape(H, W, 4)).any(axis=2)
                rows = np.nonzero(bad.any(axis=1))[0]
                print(f"#   {int(bad.sum())} pixels differ, rows {rows.min()}..{rows.max()} ({len(rows)} rows)", file=sys.stderr)
        line = {
            "metric": "Mtriangles/s + Mpixels/s, 1M-tri synthetic scene @ 2560x1920",
            "value": round(mtri, 3), "unit": "Mtriangles/s",
            "mpixels_per_s": round(mpix, 2),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32+i32 (exact-order f32 setup/barycentrics, integer snap and RGB555 colour tail)",
            "data": "synthetic",
            "config": {"workload": f"{args.config}: {NF} tris @ {W}x{H}, 256x256 8-bit atlas, affine+snap+RGB555 dither, painter's",
                       "triangles_drawn": tm.triangles_drawn, "fragments": fragments,
                       "parallelism": (f"screen bands x{world}, RCCL gather " + ("overlapped with the next frame (two framebuffers)" if pipelined else "after every frame")) if world > 1 else "single GPU"},
            "phases_ms": {k: round(v, 4) for k, v in phases.items()},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        if weak is not None:
            line["weak_series"] = weak
        if cpu_all is not None:
            line["cpu_all_cores"] = cpu_all
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Host side of the multi-GPU exchange step THROUGH THE C ABI (include/b32raster.h "multi-GPU", b32_gather.hip): what a Rust host
would write around its per-rank `render_mesh_15` loop, in the two transports the library offers.  No torch here: the only thing a
transport needs from the host's own channel is `bcast(payload_or_None) -> payload` (rank 0's bytes handed to every rank once, at set-up).

One process per GPU; rank 0 is the root, the process whose `fb.pixels` the presenter reads (game/renderer.rs:179-214).

    ex = SharedFramebufferExchange(ctx, W, H, rank, world, bcast)    # or RcclExchange(...)
    fb.set_band(*band_rows(H, world, rank))
    every frame:   ex.begin();  fb.clear(...);  scene.render_async(...);  ex.end()

Every call only enqueues on the context's stream; nothing blocks the host between frames.
"""
from .bands import band_rows


class SharedFramebufferExchange:
    """Transport (1): the root exports its library-owned framebuffer, band ranks map it (HIP IPC) and draw their rows straight into the
    root's HBM; one epoch word per rank orders the frames on the device (publish / wait, release / acquire)."""
    name = "shm"

    def __init__(self, ctx, width, height, rank, world, bcast, timeout_us=2_000_000):
        from . import rasterizer as R
        self.ctx, self.rank, self.world, self.timeout_us = ctx, rank, world, int(timeout_us)
        self.n = 0
        share = None
        if rank == 0:
            R._chk(ctx.lib.b32_fb_new(ctx.h, width, height), "b32_fb_new")       # the root's framebuffer must be the library's own allocation
            share = ctx.band_export()
        share = bcast(share)
        if rank > 0:
            ctx.band_import(share, rank)

    def begin(self):
        """Before the rank's clear + draw of the next frame: a band rank may only overwrite its rows once the root has consumed them."""
        self.n = (self.n + 1) & 0xFFFFFFFF
        if self.rank > 0 and self.n > 1:
            self.ctx.band_acquire(self.n - 1, self.timeout_us)

    def end(self):
        """Behind the rank's draw: publish (band rank) / wait for every rank and release the frame (root), in stream order."""
        if self.rank > 0:
            self.ctx.band_publish(self.n)
        else:
            self.ctx.band_wait_all(self.world, self.n, self.timeout_us, release_after=True)

    def timeouts(self):
        """Waits that gave up since the export (every rank reads the same shared word)."""
        return self.ctx.band_status()[2]

    def close(self):
        if self.rank > 0:
            self.ctx.band_close()


class RcclExchange:
    """Transport (2): every rank draws into its OWN framebuffer; the band rows travel to the root by one grouped ncclSend / ncclRecv per
    frame on the context's stream (b32_gather_bands_rccl).  The communicator is made by the library (b32_rccl_comm_create)."""
    name = "rccl"

    def __init__(self, ctx, width, height, rank, world, bcast):
        from . import rasterizer as R
        self.ctx, self.rank, self.world = ctx, rank, world
        uid = bcast(R.Context.rccl_unique_id() if rank == 0 else None)
        self.comm = ctx.rccl_comm_create(uid, rank, world)
        self.bands = [band_rows(height, world, r) for r in range(world)]

    def begin(self):
        pass

    def end(self):
        self.ctx.gather_bands_rccl(self.comm, self.rank, self.world, 0, self.bands)

    def timeouts(self):
        return 0

    def close(self):
        if self.comm is not None:
            self.ctx.rccl_comm_destroy(self.comm)
            self.comm = None

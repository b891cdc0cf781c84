"""Screen-band sharding of one frame across ranks and the gather of band rows.

One process per GPU.  Rank r owns the contiguous rows [y0, y1) of the frame; transform/cull/setup run on every
rank (inputs are replicated in each GPU's HBM), everything after is band-local; the only exchange step of the path
is one gather of W*(y1-y0)*4 bytes per rank to rank 0 (RCCL over xGMI on GPUs; gloo in the CPU tests).
The functions take any torch.distributed backend and any uint8 tensors, so the same code is covered by the
world_size-2 gloo tests with the CPU oracle standing in for the band renderer.
"""
import torch
import torch.distributed as dist


from .bands import band_rows  # noqa: E402,F401  (the row partition, torch-free)


def gather_bands(frame: torch.Tensor, width, height, world_size, rank, dst=0, group=None):
    """Collect every rank's band rows into `frame` on rank `dst`.

    frame: flat uint8 tensor of width*height*4 bytes on every rank; rank r has rendered rows band_rows(r) into it.
    Equal bands use one gather straight into views of dst's frame; ragged bands are padded to the tallest band.
    """
    if world_size == 1:
        return frame
    row = width * 4
    bands = [band_rows(height, world_size, r) for r in range(world_size)]
    sizes = [(b[1] - b[0]) * row for b in bands]
    y0, y1 = bands[rank]
    mine = frame[y0 * row:y1 * row]
    if len(set(sizes)) == 1:
        out = [frame[b[0] * row:b[1] * row] for b in bands] if rank == dst else None
        if rank == dst:
            # gather forbids aliasing of input and output: stage dst's own band
            out[dst] = torch.empty_like(mine)
        dist.gather(mine, out, dst=dst, group=group)
        return frame
    mx = max(sizes)
    send = torch.zeros(mx, dtype=frame.dtype, device=frame.device)
    send[:mine.numel()] = mine
    out = [torch.empty(mx, dtype=frame.dtype, device=frame.device) for _ in range(world_size)] if rank == dst else None
    dist.gather(send, out, dst=dst, group=group)
    if rank == dst:
        for r, b in enumerate(bands):
            if r != dst:
                frame[b[0] * row:b[1] * row] = out[r][:sizes[r]]
    return frame


def gather_bands_async(frame: torch.Tensor, width, height, world_size, rank, dst=0, group=None):
    """Non-blocking form of gather_bands for equal bands: returns (work, keepalive) -- `work.wait()` makes the current stream (or the
    host, for CPU tensors) wait for the gather; `keepalive` must be held until then.  Returns None when there is nothing to do, and
    raises ValueError for ragged bands (use gather_bands).  The collective starts after everything already enqueued on the current
    stream, so a frame rendered on that stream is complete before its rows travel; the NEXT frame, rendered into another buffer, overlaps
    with the transfer."""
    if world_size == 1:
        return None
    row = width * 4
    bands = [band_rows(height, world_size, r) for r in range(world_size)]
    if len({(b[1] - b[0]) for b in bands}) != 1:
        raise ValueError("gather_bands_async needs equal bands")
    y0, y1 = bands[rank]
    mine = frame[y0 * row:y1 * row]
    out = None
    if rank == dst:
        out = [frame[b[0] * row:b[1] * row] for b in bands]
        out[dst] = torch.empty_like(mine)          # gather forbids aliasing of input and output: stage dst's own band
    work = dist.gather(mine, out, dst=dst, group=group, async_op=True)
    return work, (mine, out)

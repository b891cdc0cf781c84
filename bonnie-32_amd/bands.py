"""Row partition of a band-sharded frame: no torch, so that a host that only speaks the C ABI (tests/band_worker_ipc.py, the
C++ mirror) can share it with bonnie32_amd.parallel."""


def band_rows(height, world_size, rank):
    """Rows [y0, y1) owned by `rank`: a balanced contiguous partition (sizes differ by at most one row)."""
    base, extra = divmod(height, world_size)
    y0 = rank * base + min(rank, extra)
    y1 = y0 + base + (1 if rank < extra else 0)
    return y0, y1

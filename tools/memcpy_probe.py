"""hipMemcpyAsync host-to-device: enqueue cost and stream time per copy, pageable vs pinned source, small sizes (why the drop-in
calls move their uploads with one copy kernel out of a pinned arena instead of five SDMA copies)."""
import ctypes as C, time, numpy as np
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
hip.hipSetDevice(0)
st = C.c_void_p(); hip.hipStreamCreate(C.byref(st))
for size in (4096, 72000, 400000):
    d = C.c_void_p(); hip.hipMalloc(C.byref(d), size)
    pin = C.c_void_p(); hip.hipHostMalloc(C.byref(pin), size, 0)
    page = np.zeros(size, np.uint8)
    for name, src in (("pageable", page.ctypes.data), ("pinned", pin.value)):
        for _ in range(20): hip.hipMemcpyAsync(d, C.c_void_p(src), C.c_size_t(size), 1, st)
        hip.hipStreamSynchronize(st)
        N = 500; t0 = time.perf_counter()
        for _ in range(N): hip.hipMemcpyAsync(d, C.c_void_p(src), C.c_size_t(size), 1, st)
        t1 = time.perf_counter(); hip.hipStreamSynchronize(st); t2 = time.perf_counter()
        print(f"{size:7d} B {name:8s}: enqueue {(t1-t0)/N*1e6:6.1f} us/call, drained after {(t2-t0)/N*1e6:6.1f} us/call")

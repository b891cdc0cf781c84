"""What this MI355X sustains on plain streaming kernels (torch elementwise ops), to put k_setup's 3.5 TB/s in perspective."""
import torch, time
dev = torch.device("cuda", 0)
def bench(fn, nbytes, name, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"{name}: {us:.1f} us, {nbytes / us / 1e6:.2f} TB/s")
for mb in (64, 128, 512):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    y = torch.empty_like(x)
    bench(lambda: y.copy_(x), 2 * n * 4, f"copy {mb} MB -> {mb} MB")
    bench(lambda: torch.add(x, 1.0, out=y), 2 * n * 4, f"add  {mb} MB -> {mb} MB")
    bench(lambda: y.fill_(1.0), n * 4, f"fill {mb} MB")
    bench(lambda: x.sum(), n * 4, f"sum  {mb} MB (read only)")

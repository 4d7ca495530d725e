"""Bandwidth calibration of the GPU box (plain torch kernels): what a copy / write-only / read-only
stream actually reaches, as context for the roofline fractions in bench.py."""
import torch, time
dev = torch.device("cuda")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n
for mb in (256, 822, 2048):
    n = mb*1024*1024//4
    x = torch.randn(n, device=dev); y = torch.empty_like(x)
    t = timeit(lambda: y.copy_(x)); print(f"copy  {mb}MB: {2*n*4/t/1e12:.2f} TB/s")
    t = timeit(lambda: y.fill_(1.0)); print(f"fill  {mb}MB: {n*4/t/1e12:.2f} TB/s")
    t = timeit(lambda: x.sum()); print(f"sum   {mb}MB: {n*4/t/1e12:.2f} TB/s")
    t = timeit(lambda: torch.add(x, 1.0, out=y)); print(f"add   {mb}MB: {2*n*4/t/1e12:.2f} TB/s")
    z = torch.empty(2*n, device=dev)
    t = timeit(lambda: torch.cat([x,x], out=z)); print(f"1r2w  {mb}MB: {3*n*4/t/1e12:.2f} TB/s (read n, write 2n; source re-read hits cache)")

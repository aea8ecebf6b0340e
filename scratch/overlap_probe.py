"""Does an HBM-bound kernel (Adam) overlap an MFMA-bound one (2592^2 conv) when issued on another stream?  scratch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raygun_amd import ops
dev = torch.device("cuda")
x = ops.to_cl(torch.randn(1, 2592, 19, 19, 19, device=dev))
w = torch.randn(2592, 2592, 3, 3, 3, device=dev).contiguous(memory_format=torch.channels_last_3d) * 0.01
p = torch.randn(60_000_000, device=dev); g = torch.randn_like(p); m = torch.zeros_like(p); v = torch.zeros_like(p)
side = torch.cuda.Stream()
def conv(n):
    for _ in range(n): ops.conv3d(x, w)
def adam(n):
    for i in range(n): ops.adam_step_(p, g, m, v, 1e-3, 0.9, 0.999, 1e-8, 0.0, i + 1)
def timed(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
conv(2); adam(2)
tc = timed(lambda: conv(10)); ta = timed(lambda: adam(10))
def both():
    with torch.cuda.stream(side):
        adam(10)
    conv(10)
tb = timed(both)
print(f"conv x10 {tc:.2f} ms, adam x10 {ta:.2f} ms, concurrent {tb:.2f} ms (serial sum {tc + ta:.2f})")

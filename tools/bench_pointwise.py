"""HBM-bound ops of the CycleGAN step one by one (GPU box): forward and backward time per op and shape (HIP events, median of 20), with the
algorithmic bytes each pass has to move and the rate that makes.   usage: python tools/bench_pointwise.py"""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raygun_amd import ops
dev = torch.device("cuda")


def timed(fn, reps=20):
    for _ in range(3): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) for a, b in ev) * 1e3   # us


def mb(t):
    n, c, d, h, w = t.shape
    return n * d * h * w * ops.voxel_stride_of(c) * 2 / 1e6


def report(name, shape, us_f, bytes_f, us_b, bytes_b):
    print(f"{name:34s} {str(shape):28s} fwd {us_f:7.1f} us {bytes_f:7.1f} MB {bytes_f / us_f:6.2f} TB/s | bwd {us_b:7.1f} us {bytes_b:7.1f} MB {bytes_b / us_b:6.2f} TB/s", flush=True)


for c, s in ((256, 32), (128, 64), (64, 128), (128, 32), (256, 16)):
    x = ops.to_cl(torch.randn(1, c, s, s, s, device=dev)).requires_grad_(True)
    for act in ("relu", "none"):
        y = ops.instance_norm(x, act=act); g = ops.to_cl(torch.randn_like(y.float()))
        f = timed(lambda: ops.instance_norm(x, act=act))
        b = timed(lambda: torch.autograd.grad(y, x, g, retain_graph=True))
        report(f"instance_norm[{act}]", tuple(x.shape), f, 3 * mb(x), b, (4 if act == "none" else 5) * mb(x))
    r = ops.to_cl(torch.randn(1, c, s, s, s, device=dev))
    y = ops.instance_norm(x, residual=r); g = ops.to_cl(torch.randn_like(y.float()))
    f = timed(lambda: ops.instance_norm(x, residual=r)); b = timed(lambda: torch.autograd.grad(y, x, g, retain_graph=True))
    report("instance_norm[+res]", tuple(x.shape), f, 4 * mb(x), b, 4 * mb(x))
    p = 1 if s <= 64 else 3
    y = ops.pad3d(x, p); g = ops.to_cl(torch.randn_like(y.float()))
    f = timed(lambda: ops.pad3d(x, p)); b = timed(lambda: torch.autograd.grad(y, x, g, retain_graph=True))
    report(f"pad3d[reflect {p}]", tuple(x.shape), f, mb(x) + mb(y), b, mb(x) + mb(y))
    y = ops.add(x, r)
    f = timed(lambda: ops.add(x, r))
    report("add", tuple(x.shape), f, 3 * mb(x), 1.0, 0.0)
# the 7x7 in-plane taps <-> 49 channels maps of the generator's 7^3 stem / tail (1-channel side 134 or 128 wide)
one = ops.to_cl(torch.randn(1, 1, 134, 134, 134, device=dev))
many = ops.to_cl(torch.randn(1, 49, 134, 128, 128, device=dev))
with torch.no_grad():
    sp = lambda: ops._DiagSpread.apply(one, 7, 7, 1, 128, 128)
    f = timed(sp); y = sp()
    report("diag_spread 7x7", tuple(y.shape), f, mb(one) + mb(y), 1.0, 0.0)
    big = ops.to_cl(torch.randn(1, 49, 128, 134, 134, device=dev))
    sm = lambda: ops._DiagSum.apply(big, 7, 7, 1, 128, 128, None)
    f = timed(sm); y = sm()
    report("diag_sum 7x7 (+1)", tuple(big.shape), f, mb(big) + mb(y), 1.0, 0.0)
    sm2 = lambda: ops._DiagSum.apply(many, 7, 7, -1, 134, 134, None)
    f = timed(sm2); y = sm2()
    report("diag_sum 7x7 (-1, adjoint)", tuple(many.shape), f, mb(many) + mb(y), 1.0, 0.0)

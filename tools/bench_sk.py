"""The stream-K schedule of igemm5 on one layer (GPU box): forward + backward-data under RG_SK settings.  usage: python tools/bench_sk.py cin cout size k [RG_SK values]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raygun_amd import ops
cin, cout, s, k = [int(v) for v in sys.argv[1:5]]
opts = [tuple(int(q) for q in (v + ':0').split(':')[:2]) for v in sys.argv[5:]] or [(0, 0), (-1, 0)]   # RG_SK[:RG_GEMM5]
dev = torch.device("cuda")
x = ops.to_cl(torch.randn(1, cin, s, s, s, device=dev)).requires_grad_(True)
w = (torch.randn(cout, cin, k, k, k, device=dev) * 0.05); b = torch.zeros(cout, device=dev)
for rep in range(3):
    for o in opts:
        with ops.option("RG_SK", o[0]), ops.option("RG_GEMM5", o[1]):
            y = ops.conv3d(x, w, b, 1, 0, "none"); g = torch.randn_like(y)
            for _ in range(3):
                ops.profile_start(); y = ops.conv3d(x, w, b, 1, 0, "none"); y.backward(g); pr = ops.profile_stop()
        if rep:
            print(f"RG_SK={o}: " + " | ".join(f"{n}: {d['ms'] * 1e3:.0f} us {d['flops'] / d['ms'] / 1e9:.0f} TF" for n, d in pr.items()), flush=True)

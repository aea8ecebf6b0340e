"""One convolution layer, forward + backward, per-kernel time (GPU box).  usage: python tools/bench_one_conv.py cin cout size k [pad] [opt RG_GEMM5 values...]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raygun_amd import ops
cin, cout, s, k = [int(v) for v in sys.argv[1:5]]
pad = int(sys.argv[5]) if len(sys.argv) > 5 else 0
opts = [int(v) for v in sys.argv[6:]] or [0]
dev = torch.device("cuda")
x = ops.to_cl(torch.randn(1, cin, s, s, s, device=dev)).requires_grad_(True)
w = (torch.randn(cout, cin, k, k, k, device=dev) * 0.05).requires_grad_(True); b = torch.zeros(cout, device=dev)
for rep in range(2):
    for o in opts:
        with ops.option("RG_GEMM5", o):
            y = ops.conv3d(x, w, b, 1, pad, "none"); g = torch.randn_like(y)
            for _ in range(3):
                ops.profile_start(); y = ops.conv3d(x, w, b, 1, pad, "none"); y.backward(g); pr = ops.profile_stop()
        if rep:
            print(f"RG_GEMM5={o}: " + " | ".join(f"{n}: {d['ms'] * 1e3:.0f} us {d['flops'] / d['ms'] / 1e9:.0f} TF" for n, d in pr.items()), flush=True)

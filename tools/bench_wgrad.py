"""Weight-gradient microbench (GPU box): python3 tools/bench_wgrad.py — the wide U-Net layers, variants interleaved in one process."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raygun_amd import ops
dev = torch.device("cuda")
shapes = [(2592, 2592, 19, 3), (432, 432, 44, 3), (2592, 3456, 17, 2), (432, 2592, 21, 3)]
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1", "0"])]
for cin, cout, s, k in shapes:
    x = ops.to_cl(torch.randn(1, cin, s, s, s, device=dev).to(torch.bfloat16))
    w = (torch.randn(cout, cin, k, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    y = ops.conv3d(x, w, None, 1, 0, "none"); g = torch.randn_like(y)
    res = {}
    for rnd in range(3):
        for v in variants:
            with ops.option("RG_KWD", v):
                torch.autograd.grad(y, w, g, retain_graph=True); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): torch.autograd.grad(y, w, g, retain_graph=True)
                e1.record(); torch.cuda.synchronize()
                res.setdefault(v, []).append(e0.elapsed_time(e1) / 5)
    flops = 2.0 * cin * cout * k ** 3 * (s - k + 1) ** 3
    print(f"{cin}->{cout} @{s} k{k}: " + " | ".join(f"KWD={v}: {min(t):.3f} ms {flops / min(t) / 1e9:.0f} TF/s" for v, t in res.items()), flush=True)

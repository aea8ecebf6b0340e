import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raygun_amd import ops
dev = torch.device("cuda")
for gs, fs, target in [((9, 9, 9), (20, 20, 20), (18, 18, 18)), ((8, 9, 8), (16, 17, 15), (13, 15, 11))]:
    cf, cg, cout = 16, 64, 24
    torch.manual_seed(0)
    f = torch.randn(1, cf, *fs, device=dev).bfloat16().float().requires_grad_()
    g = torch.randn(1, cg, *gs, device=dev).bfloat16().float().requires_grad_()
    w = (torch.randn(cout, cf + cg, 3, 3, 3, device=dev) * 0.05).bfloat16().float().requires_grad_()
    b = torch.randn(cout, device=dev).requires_grad_()
    up = torch.nn.functional.interpolate(g, scale_factor=2, mode="nearest")
    x = torch.cat([ops.center_crop(f, target), ops.center_crop(up, target)], 1)
    ref = torch.tanh(torch.nn.functional.conv3d(x, w, b))
    go = torch.randn_like(ref).bfloat16().float()
    rg = torch.autograd.grad(ref, (f, g, w, b), go)
    got = ops.upcat_conv3d(f, g, target, w, b, "tanh")
    gg = torch.autograd.grad(got, (f, g, w, b), go)
    print("out", (got.float() - ref).abs().max().item(), ref.abs().max().item())
    for n, a, r in zip("fgwb", gg, rg):
        print(n, tuple(a.shape), (a.float() - r).abs().max().item(), r.abs().max().item(), flush=True)
    # pieces: skip-only and up-only
    w0 = w.detach().clone(); w0[:, cf:] = 0
    print("skip-only fwd", (ops.upcat_conv3d(f, g, target, w0, None).float() - torch.nn.functional.conv3d(x, w0)).abs().max().item())
    w1 = w.detach().clone(); w1[:, :cf] = 0
    print("up-only fwd", (ops.upcat_conv3d(f, g, target, w1, None).float() - torch.nn.functional.conv3d(x, w1)).abs().max().item())

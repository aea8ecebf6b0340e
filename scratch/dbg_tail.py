import sys, os, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raygun_amd import ops
torch.manual_seed(0)
dev = torch.device('cuda')
bf = lambda t: t.to(torch.bfloat16).float()
for (cin, mid, P, k, pad, sp, act) in [(16, 16, 1, 7, 3, (10, 11, 12), "tanh"), (64, 64, 1, 7, 0, (12, 12, 14), "none"), (24, 32, 2, 3, 1, (6, 7, 8), "none")]:
    x = bf(torch.randn(2, cin, *sp)); wa = bf(torch.randn(mid, cin, k, k, k) * (1.0 / (cin * k ** 3) ** 0.5)); ba = torch.randn(mid) * 0.1
    wb = bf(torch.randn(P, mid, 1, 1, 1) * (1.0 / mid ** 0.5)); bb = torch.randn(P) * 0.1
    afn = {"none": lambda t: t, "tanh": torch.tanh}[act]
    ref_in = [t.clone().requires_grad_(True) for t in (x, wa, ba, wb, bb)]
    yr = afn(F.conv3d(F.conv3d(ref_in[0], ref_in[1], ref_in[2], 1, pad), ref_in[3], ref_in[4]))
    g = torch.randn_like(yr); yr.backward(g)
    my_in = [t.clone().to(dev).requires_grad_(True) for t in (x, wa, ba, wb, bb)]
    y = ops.conv3d_linear_pair(my_in[0], my_in[1], my_in[2], my_in[3], my_in[4], pad, act, 0.0)
    y = ops.leave(y) if hasattr(ops, "leave") else y
    y.float().backward(g.to(dev).to(y.dtype)) if y.shape == g.shape else None
    print(cin, mid, P, k, 'shape', tuple(y.shape), tuple(yr.shape), 'y err', ((y.float().cpu() - yr).abs().max() / yr.abs().max()).item())
    for name, a, b in zip("x wa ba wb bb".split(), my_in, ref_in):
        print('   grad', name, ((a.grad.float().cpu() - b.grad).abs().max() / (b.grad.abs().max() + 1e-9)).item())

import sys, os, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raygun_amd import ops, _lib
torch.manual_seed(0)
dev = torch.device('cuda')
def bf(t): return t.to(torch.bfloat16).float()
for (cin, cout, k, s, p, sp) in [(1, 64, 4, 2, 1, (12, 12, 12)), (4, 64, (4,4,1), (2,2,1), 0, (14,14,6)), (4, 64, (4,4,1), 1, 0, (14,14,6)), (8, 64, 3, 2, 1, (9,9,9)), (72, 64, 3, 2, 1, (9,9,9))]:
    kk = (k,)*3 if isinstance(k, int) else k
    x = bf(torch.randn(2, cin, *sp)); w = bf(torch.randn(cout, cin, *kk) * 0.1); b = torch.randn(cout)
    y = ops.conv3d(x.to(dev), w.to(dev), b.to(dev), s, p, "none", 0.0)
    y = ops.from_cl(y) if hasattr(ops, 'from_cl') else y
    yr = F.conv3d(x, w, b, s, p)
    yy = y.float().cpu()
    err = (yy - yr).abs()
    print(cin, cout, k, s, p, 'kernel', _lib.lib().rg_last_kernel_name().decode() if hasattr(_lib.lib(), 'rg_last_kernel_name') else '', 'maxerr', err.max().item(), 'ref max', yr.abs().max().item())
    if err.max() > 0.1:
        bad = (err > 0.1)
        print('  bad frac', bad.float().mean().item(), 'bad by n', bad.float().mean(dim=(1,2,3,4)).tolist(), 'by d', bad.float().mean(dim=(0,1,3,4)).tolist())
        print('  by w', bad.float().mean(dim=(0,1,2,3)).tolist())

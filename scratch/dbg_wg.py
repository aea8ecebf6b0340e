import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device('cuda')
buf = torch.zeros(64 * 16 * 8, dtype=torch.int64, device=dev)
os.environ["RG_WGDBG"] = str(buf.data_ptr())
from raygun_amd import ops
shapes = {0: (3024, 432, 34), 1: (2592, 2592, 19), 2: (432, 432, 44)}
for idx in [int(v) for v in sys.argv[1].split(',')]:
    cin, cout, s = shapes[idx]
    x = ops.to_cl(torch.randn(1, cin, s, s, s, device=dev).to(torch.bfloat16))
    w = (torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05).requires_grad_(True); b = torch.zeros(cout, device=dev)
    y = ops.conv3d(x, w, b, 1, 0, 'relu'); g = torch.randn_like(y)
    for rep in range(2):
        buf.zero_(); y = ops.conv3d(x, w, b, 1, 0, 'relu'); y.backward(g); torch.cuda.synchronize()
    d = buf.view(64, 16, 8).cpu()
    wg = d[:8, :9]  # first 8 workgroups, 9 waves
    steps = wg[0, 0, 4].item()
    print(f"{cin}->{cout}@{s}: steps/WG {steps}")
    for nm, k in (("dma wait", 0), ("barrier", 1), ("dma issue", 2), ("compute", 3)):
        v = wg[:, :, k].float() / max(steps, 1)
        print(f"   {nm:12s} per step: mean {v.mean():8.0f}  min {v.min():8.0f}  max {v.max():8.0f}   by wave(0..8) of WG0: {[int(t) for t in v[0]]}")

import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device('cuda')
buf = torch.zeros(64 * 4 * 4, dtype=torch.int64, device=dev)
os.environ["RG_IG4DBG"] = str(buf.data_ptr())
from raygun_amd import ops
shapes = {0: (3024, 432, 34), 1: (2592, 2592, 19), 2: (432, 432, 44)}
for idx in [int(v) for v in sys.argv[1].split(',')]:
    cin, cout, s = shapes[idx]
    x = ops.to_cl(torch.randn(1, cin, s, s, s, device=dev).to(torch.bfloat16)).requires_grad_(True)
    w = (torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05).requires_grad_(True); b = torch.zeros(cout, device=dev)
    for which in ("fwd", "dgrad"):
        y = ops.conv3d(x, w, b, 1, 0, 'relu'); g = torch.randn_like(y)
        torch.cuda.synchronize(); buf.zero_()
        if which == "fwd":
            y = ops.conv3d(x, w, b, 1, 0, 'relu')
        else:
            buf.zero_(); y.backward(g)
        torch.cuda.synchronize()
        d = buf.view(64, 4, 4).cpu()
        nk = 27 * ((cin if which == 'fwd' else cout) + 63) // 64
        if nk == 0: print(which, "no data"); continue
        v = d[:16].float() / nk
        print(f"{cin}->{cout}@{s} {which}: steps {nk}  barrier {v[:,:,0].mean():.0f} (min {v[:,:,0].min():.0f} max {v[:,:,0].max():.0f})  store_b+load_b {v[:,:,1].mean():.0f} (store_b alone {v[:,:,3].mean():.0f})  mma-phase {v[:,:,2].mean():.0f}  | waves of WG0: {[[int(t) for t in v[0,w]] for w in range(4)]}")

"""The generator's 7^3 stem (1 -> 64 at 128^3, pad 3) forward + backward under both folds: per-kernel times (GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raygun_amd import ops
dev = torch.device("cuda")
torch.manual_seed(0)
x0 = torch.randn(1, 1, 128, 128, 128, device=dev)
w = (torch.randn(64, 1, 7, 7, 7, device=dev) * 0.05).requires_grad_(True); b = torch.zeros(64, device=dev, requires_grad=True)
outs = {}
for fold in ("0", "1"):
    os.environ["RAYGUN_AMD_FOLD_DH"] = fold
    for rep in range(3):
        x = x0.clone().requires_grad_(True)
        ops.profile_start()
        xp = ops.pad3d(ops.enter(x), 3, "reflect")
        y = ops.conv3d(xp, w, b, 1, 0, "none")
        yl = ops.leave(y)
        g = torch.ones_like(yl)
        gw, gx = torch.autograd.grad(yl, (w, x), g)
        pr = ops.profile_stop()
    outs[fold] = (yl.detach(), gw.detach(), gx.detach())
    tot = sum(d["ms"] for d in pr.values())
    print(f"FOLD_DH={fold}: conv-kernel total {tot:.3f} ms  " + " | ".join(f"{n}: {d['ms']*1e3:.0f} us" for n, d in pr.items()), flush=True)
for i, nm in enumerate(("y", "gw", "gx")):
    a, c = outs["0"][i].float(), outs["1"][i].float()
    print(nm, "max rel diff", float((a - c).abs().max() / c.abs().max()))
for fold in ("0", "1", "0", "1"):
    os.environ["RAYGUN_AMD_FOLD_DH"] = fold
    def once():
        x = x0.clone().requires_grad_(True)
        yl = ops.leave(ops.conv3d(ops.pad3d(ops.enter(x), 3, "reflect"), w, b, 1, 0, "none"))
        return torch.autograd.grad(yl, (w, x), torch.ones_like(yl))
    for _ in range(3): once()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): once()
    e1.record(); torch.cuda.synchronize()
    print(f"FOLD_DH={fold}: whole stem forward + backward {e0.elapsed_time(e1) / 10:.3f} ms", flush=True)

"""f16-storage parity of config U with and without the Winograd path (GPU box): python3 tools/f16_wino_probe.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raygun_amd import ops, synth
from raygun_amd.torch.models import MTLSDModel
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
def l2rel(a, b):
    a = a.detach().cpu().numpy() if hasattr(a, "detach") else a
    return float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))
dev = torch.device("cuda")
for wino in ("0", "1"):
    os.environ["RAYGUN_AMD_WINOGRAD"] = wino
    for size, name in ((100, "mtlsd_100"), (196, "mtlsd_196_sampled")):
        fx = np.load(os.path.join(G, name + ".npz"))
        m = MTLSDModel(); m.load_state_dict(synth.fill_state_dict(m.state_dict(), 51)); m = m.to(dev).eval()
        raw = torch.from_numpy(synth.uniform((1, 1, size, size, size), 52)).to(dev)
        with torch.no_grad(), ops.precision("f16"):
            affs, lsds = m(raw)
        if size == 100:
            print("wino", wino, name, "affs %.3e lsds %.3e" % (l2rel(affs, fx["affs"]), l2rel(lsds, fx["lsds"])))
        else:
            flat = torch.cat([affs.reshape(-1), lsds.reshape(-1)])
            got = flat[torch.from_numpy(fx["out_idx"]).to(dev)].cpu().numpy()
            print("wino", wino, name, "sampled %.3e max %.3e" % (l2rel(got, fx["out_val"]), float(np.abs(got - fx["out_val"]).max())))

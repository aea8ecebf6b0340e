"""Affinity error of config U against the reference fixtures under bf16 and f16 storage.  GPU box, repo root."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raygun_amd import ops, synth
from raygun_amd.torch.models import MTLSDModel

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
dev = torch.device("cuda")
m = MTLSDModel()
m.load_state_dict(synth.fill_state_dict(m.state_dict(), 51))
m = m.to(dev).eval()
for size, name in ((100, "mtlsd_100"), (196, "mtlsd_196_sampled")):
    fx = np.load(os.path.join(G, name + ".npz"))
    raw = torch.from_numpy(synth.uniform((1, 1, size, size, size), 52)).to(dev)
    for kind in ("bf16", "f16"):
        with torch.no_grad(), ops.precision(kind):
            affs, lsds = m(raw)
        if size == 100:
            got = np.concatenate([affs.cpu().numpy().reshape(-1), lsds.cpu().numpy().reshape(-1)])
            want = np.concatenate([fx["affs"].reshape(-1), fx["lsds"].reshape(-1)])
        else:
            flat = torch.cat([affs.reshape(-1), lsds.reshape(-1)])
            got = flat[torch.from_numpy(fx["out_idx"]).to(dev)].cpu().numpy(); want = fx["out_val"]
        d = got.astype(np.float64) - want
        print(f"{name} {kind}: rel-L2 {np.sqrt((d**2).sum()/(want.astype(np.float64)**2).sum()):.3e}  max-abs {np.abs(d).max():.3e}  "
              f"max-rel {np.abs(d / want).max():.3e}  finite {np.isfinite(got).all()}", flush=True)

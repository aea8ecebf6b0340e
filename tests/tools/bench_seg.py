"""Timing of rg_mws_segment vs the C oracle (scratch).  usage: python scratch/bench_seg.py [size ...]"""
import sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import mws as O
from raygun_amd import segment as S, synth

def affs_for(n, kind):
    shape = (n, n, n)
    if kind == "noise":
        return synth.uniform((20,) + shape, 5, 0, 1).astype(np.float32)
    z, y, x = np.meshgrid(*[np.arange(s, dtype=np.float32) for s in shape], indexing="ij")
    f = np.sin(z / 9.1 + 1) * np.sin(y / 11.3) * np.sin(x / 8.7 + 1.0)
    base = (np.abs(f) > 0.2).astype(np.float32) * 0.8 + 0.1
    return (0.8 * base[None] + 0.2 * synth.uniform((20,) + shape, 21, 0, 1)).astype(np.float32)

nb = O.neighborhood(8).astype(np.int32)
for n in [int(a) for a in sys.argv[1:]] or [64, 128]:
    for kind in ("blobs", "noise"):
        affs = affs_for(n, kind)
        w, m = O.prepare(affs.copy())
        wt = torch.from_numpy(w).cuda(); mt = torch.from_numpy(m.astype(np.uint8)).cuda()
        S.compute_mws_segmentation(wt[:, :16, :16, :16].contiguous(), nb, 3, (10, 10, 10), mask=mt[:16, :16, :16].contiguous())
        S.compute_mws_segmentation(wt, nb, 3, (10, 10, 10), mask=mt)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        seg = S.compute_mws_segmentation(wt, nb, 3, (10, 10, 10), mask=mt)
        torch.cuda.synchronize(); tg = time.perf_counter() - t0
        line = f"n={n} {kind}: gpu {tg*1e3:.1f} ms = {n**3/tg/1e6:.2f} Mvox/s ({n**3*88/tg/1e9:.2f} GB/s) segments={int(seg.max())} masked={1-m.mean():.2f}"
        if n <= int(os.environ.get("RG_ORACLE_MAX", "160")):
            t0 = time.perf_counter(); ref = O.compute_mws_segmentation(w, nb, 3, (10, 10, 10), m); tc = time.perf_counter() - t0
            line += f" | cpu {tc*1e3:.0f} ms = {n**3/tc/1e6:.2f} Mvox/s equal={bool((ref == seg).all())}"
        print(line, flush=True)

"""Mutex watershed on uint8-quantised affinities (what the reference's predict writes): does the tie-heavy order hurt? (GPU box)"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from raygun_amd import segment as S
dev = torch.device("cuda"); nb = S.build_neighborhood()
for size in (128, 192, 256):
    base = bench.structured_affs(size, 0, dev)
    for name, affs in (("float", base), ("uint8 levels", torch.round(base * 255) / 255), ("6-bit levels", torch.round(base * 63) / 63)):
        st = {}
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            w, m = S.prepare_weights(affs, 3, 0.5, 1.0)
            lab = S.compute_mws_segmentation(w, nb, 3, strides=[10, 10, 10], mask=m, as_tensor=True, stats=st if rep else None)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{size}^3 {name:13s}: {dt*1e3:9.2f} ms  segments {int(lab.max())}  {st}", flush=True)

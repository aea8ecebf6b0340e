"""Persistent-grid size / window size sweep of the mutex watershed (diagnostic switches RG_MWS_WGS, RG_MWS_WMUL): structured and pure-noise 256^3.  GPU box."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from raygun_amd import ops, segment as S
dev = torch.device("cuda")
nb = S.build_neighborhood()
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cases = {"blobs": bench.structured_affs(size, 0, dev), "noise": torch.rand((20, size, size, size), generator=torch.Generator(device=dev).manual_seed(5), device=dev)}
for wgs, wmul in ((128, 4), (256, 2), (256, 4), (192, 4), (128, 8), (128, 2), (64, 8), (256, 1)):
    out = []
    for name, affs in cases.items():
        with ops.option("RG_MWS_WGS", wgs), ops.option("RG_MWS_WMUL", wmul):
            w, m = S.prepare_weights(affs, 3, 0.5, 1.0)
            ts = []
            for rep in range(4):
                st = {}
                torch.cuda.synchronize(); t0 = time.perf_counter()
                S.compute_mws_segmentation(w, nb, 3, strides=[10, 10, 10], mask=m, as_tensor=True, stats=st if rep == 3 else None)
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        out.append(f"{name} {min(ts[:3]) * 1e3:.2f} ms rounds {st['rounds']} visits {st['window_visits'] / 1e6:.1f}M A/B {st['phase_ms'][0]:.1f}/{st['phase_ms'][1]:.1f}")
    print(f"wgs {wgs} wmul {wmul}: " + " | ".join(out), flush=True)

"""The adversarial input of the mutex watershed: the UNSHAPED output of the random-initialised network (uint8 affinities in a narrow band around 128: few
distinct keys, a random mask) — round-kernel statistics per size.  GPU box: python tools/mws_raw_case.py [sizes...]"""
import sys, os, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from raygun_amd import predict as P, segment as S
from raygun_amd.torch.models import MTLSDModel
dev = torch.device("cuda")
torch.manual_seed(1234)
with torch.device(dev):
    model = MTLSDModel(num_affs=20).eval()
nb = S.build_neighborhood()
for size in [int(v) for v in sys.argv[1:]] or [300, 404]:
    vol = bench.config5_volume((size,) * 3, 0, size, dev)
    dests, nblk = P.predict_volume(model, vol, input_shape=[196] * 3, output_shape=[104] * 3, out_dtypes=[np.uint8] * 2, rank=0, world=1)
    a8 = dests[0]
    inner = a8[:, 46:-46, 46:-46, 46:-46]
    vals, cnt = torch.unique(inner, return_counts=True)
    print(f"size {size}: {nblk} blocks; affinity bytes {int(vals.min())}..{int(vals.max())} ({len(vals)} distinct), mean {float(inner.float().mean()):.1f}", flush=True)
    _, m = S.prepare_weights_u8(a8, 1.0 / 255.0, 3, 0.5, 1.0, device=dev, mask_only=True)
    print(f"   mask fraction {float(m.float().mean()):.3f}", flush=True)
    w = S.U8Affinities(a8, 1.0 / 255.0, 3)
    for rep in range(2):
        st = {}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lab = S.compute_mws_segmentation(w, nb, 3, strides=[10, 10, 10], mask=m, device=dev, as_tensor=True, stats=st)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"   {dt * 1e3:.1f} ms, segments {int(lab.max())}, {st}", flush=True)

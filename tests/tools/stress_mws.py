"""Randomised bit-exactness stress of rg_mws_segment against the C oracle (scratch; GPU box)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import mws as O
from raygun_amd.segment import compute_mws_segmentation
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
NB = [np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [2, 0, 0], [0, 2, 0], [0, 0, 2], [0, 1, 1], [0, 1, -1]], dtype=np.int32),
      O.neighborhood(8).astype(np.int32),
      np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [3, 0, 0], [0, 3, 0], [0, 0, 3], [0, 4, 4], [0, 4, -4], [2, 2, 2]], dtype=np.int32)]
bad = 0; t0 = time.time(); n = 0
while time.time() - t0 < float(sys.argv[2]) if len(sys.argv) > 2 else 60:
    nb = NB[rng.integers(len(NB))]
    shape = tuple(int(v) for v in rng.integers(8, 44, 3))
    strides = tuple(int(v) for v in rng.integers(1, 5, 3))
    kind = rng.integers(4)
    w = rng.random((len(nb),) + shape, dtype=np.float32)
    if kind == 1:  # blobs
        z, y, x = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
        f = np.sin(z / 3.1 + 1) * np.sin(y / 4.3) * np.sin(x / 3.7 + 1.0)
        base = ((np.abs(f) > 0.25) * 0.8 + 0.1).astype(np.float32)
        w = (0.7 * base[None] + 0.3 * w).astype(np.float32); w[3:] = 1 - w[3:]
    elif kind == 2:  # many ties
        w = (np.round(w * 4) / 4).astype(np.float32)
    elif kind == 3:  # strong repulsion
        w[3:] = (0.5 + 0.5 * w[3:]).astype(np.float32)
    mask = (rng.random(shape) > 0.15) if rng.integers(2) else None
    ref = O.compute_mws_segmentation(w, nb, 3, strides, mask)
    got = compute_mws_segmentation(w, nb, 3, strides, mask=mask)
    n += 1
    if not (ref == got).all():
        bad += 1
        print("MISMATCH", shape, strides, kind, mask is not None, len(nb), int((ref != got).sum()), flush=True)
print(f"{n} volumes, {bad} mismatches", flush=True)

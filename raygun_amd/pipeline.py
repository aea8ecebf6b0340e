"""BASELINE config 5 on one GPU, end to end in HBM: block-wise prediction of an integer volume followed by the mutex watershed of the
predicted affinities — the reference's `raygun-predict` then `raygun-segment` flow (src/raygun/predict.py:154-170 writing the
affinity dataset, src/raygun/segment.py:139-230 reading it back) without the round trip through the container: a 2048^3 / 8 source
part is 1 GB, its 20-channel uint8 affinities 21 GB, both a small fraction of the 288 GB of an MI355X.

    labels, info = predict_and_segment(model, volume)          # volume: (Z, Y, X) uint8 tensor on the GPU

The affinities stay what the reference stores for integer destinations (worker.py:133-144: value * iinfo.max, clamped, cast), are
rescaled to [0, 1] for segment.py's arithmetic (`max_affinity_value` 1.0 on float affinities) and go through rg_mws_prepare and the
(slab) mutex watershed.  Across ranks: every rank predicts and segments its own Z part (predict: static block partition, no
collective; segment: segment_slabs with z0 / global_shape — one boundary exchange)."""
import time

import numpy as np
import torch

from . import predict as P
from . import segment as S


def predict_and_segment(model, volume, input_shape=(196, 196, 196), output_shape=(104, 104, 104), sep=3, mask_thresh=0.5, strides=(10, 10, 10),
                        n_diagonals=8, aff_output=0, precision=None, n_slabs=None, z0=0, global_shape=None, timings=None):
    """volume: integer (Z, Y, X) tensor resident on the GPU.  `model(x)` returns the affinity prediction as output `aff_output`
    (MTLSDModel: (affs, lsds)).  Returns (labels int64 tensor on the device — same bits as the reference's uint64 —, dict with the
    uint8 affinity tensor, block count and phase times).  With `global_shape`, `volume` is this rank's Z part starting at plane z0."""
    if not (torch.is_tensor(volume) and volume.is_cuda):
        raise RuntimeError("raygun_amd.predict_and_segment keeps the whole flow in HBM: pass the volume as a CUDA/HIP tensor")
    dev = volume.device
    t0 = time.perf_counter()
    dests, nblk = P.predict_volume(model, volume, input_shape=list(input_shape), output_shape=list(output_shape),
                                   out_dtypes=None if aff_output is None else [np.uint8] * (aff_output + 1) + [np.uint8] * 8, precision=precision)
    affs8 = dests[aff_output]
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    nb = S.build_neighborhood(None, n_diagonals)
    if affs8.shape[0] != len(nb):
        raise ValueError(f"the model predicts {affs8.shape[0]} affinity channels, the neighbourhood has {len(nb)} offsets")
    S._stage(None, dev)
    # uint8 destinations hold value * 255: taken back to [0, 1] and through segment.py:188-194 INSIDE the kernels that read them — neither the
    # float32 affinities nor the float32 weights (4 bytes per value: 87 GB each for a 2048^3 / 8 part) are ever materialised; allocating
    # them cost more than the whole watershed.  Bit for bit the float route (tests/test_mws.py).
    _, m = S.prepare_weights_u8(affs8, 1.0 / 255.0, sep, mask_thresh, 1.0, device=dev, mask_only=True)
    w = S.U8Affinities(affs8, 1.0 / 255.0, sep)
    S._stage("mask (rg_mws_prepare_u8)", dev)
    need = -(-(w.shape[0] * affs8[0].numel()) // S.SINGLE_PASS_EDGE_IDS)
    if global_shape is not None or max(need, n_slabs or 1) > 1:
        labels = S.segment_slabs(w, nb, sep, strides=list(strides), mask=m, n_slabs=max(need, n_slabs or 1), device=dev, as_tensor=True,
                                 z0=z0, global_shape=global_shape)
    else:
        labels = S.compute_mws_segmentation(w, nb, sep, strides=list(strides), mask=m, device=dev, as_tensor=True)
    torch.cuda.synchronize(dev)
    t2 = time.perf_counter()
    info = dict(affinities=affs8, blocks=nblk, predict_s=t1 - t0, segment_s=t2 - t1, total_s=t2 - t0)
    if timings is not None:
        timings.update({k: v for k, v in info.items() if k.endswith("_s")})
    return labels, info

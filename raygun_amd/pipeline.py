"""BASELINE config 5, end to end in HBM: block-wise prediction of an integer volume followed by the mutex watershed of the
predicted affinities — the reference's `raygun-predict` then `raygun-segment` flow (src/raygun/predict.py:154-170 writing the
affinity dataset, src/raygun/segment.py:139-230 reading it back) without the round trip through the container: a 2048^3 / 8 source
part is 1 GB, its 20-channel uint8 affinities 21 GB, both a small fraction of the 288 GB of an MI355X.

    labels, info = predict_and_segment(model, volume)          # volume: (Z, Y, X) uint8 tensor on the GPU

The affinities stay what the reference stores for integer destinations (worker.py:133-144: value * iinfo.max, clamped, cast), are
rescaled to [0, 1] for segment.py's arithmetic (`max_affinity_value` 1.0 on float affinities) and go through rg_mws_prepare and the
(slab) mutex watershed.

ONE volume over several ranks (`shard_plan`): the Z axis is cut on the BLOCK grid of the global volume, so that the ranks' write ROIs
tile exactly what a single process would have written — rank r owns the label planes `own`, loads the source planes `src` (= `own`
plus the `context` planes of read halo either side that its first / last block row needs), predicts every block of that part (no
collective) and segments its `own` planes through segment_slabs(z0, global_shape) with the one boundary exchange.  The result equals
the one-process result with the same slab bounds bit for bit (tests/test_multirank_gpu.py)."""
import time

import numpy as np
import torch

from . import predict as P
from . import segment as S


def shard_plan(global_shape, world, input_shape=(196, 196, 196), output_shape=(104, 104, 104)):
    """Z partition of one (Z, Y, X) volume over `world` ranks on the block grid of predict.py:73-108: block rows k = 0 .. nz-1 read the
    planes [k*w, k*w + r) and write [c + k*w, c + (k+1)*w) (r / w / c = read / write / context extent along Z).  Rank i gets the
    contiguous rows [k0, k1) and
        src = [k0*w, (k1-1)*w + r)      the source planes it loads (its blocks' read ROIs: halo included)
        own = [c + k0*w, c + k1*w)      the label planes it owns, widened to plane 0 on the first and to Z on the last rank (the
                                        border no block writes: zero affinities, masked out, as in the one-process run).
    Returns one dict(rank, rows, src, own) per rank; raises when there are fewer block rows than ranks."""
    read, write, context = P.roi_geometry(3, list(input_shape), list(output_shape))
    Z = int(global_shape[0])
    r, w, c = read[0], write[0], context[0]
    nz = (Z - r) // w + 1 if Z >= r else 0
    if nz < world:
        raise ValueError(f"a volume of {Z} planes has {nz} block rows of {r} -> {w} planes: fewer than {world} ranks")
    plan = []
    for i in range(world):
        k0, k1 = i * nz // world, (i + 1) * nz // world
        plan.append(dict(rank=i, rows=(k0, k1), src=(k0 * w, (k1 - 1) * w + r),
                         own=(0 if i == 0 else c + k0 * w, Z if i == world - 1 else c + k1 * w)))
    return plan


def predict_and_segment(model, volume, input_shape=(196, 196, 196), output_shape=(104, 104, 104), sep=3, mask_thresh=0.5, strides=(10, 10, 10),
                        n_diagonals=8, aff_output=0, precision=None, n_slabs=None, bounds=None, z0=0, global_shape=None, own=None, group=None,
                        timings=None):
    """volume: integer (Z, Y, X) tensor resident on the GPU.  `model(x)` returns the affinity prediction as output `aff_output`
    (MTLSDModel: (affs, lsds)).  Returns (labels int64 tensor on the device — same bits as the reference's uint64 —, dict with the
    uint8 affinity tensor, block count and phase times).

    Several ranks, one volume: `volume` is this rank's `src` planes of `shard_plan` (read halo included), `z0` the global index of its
    first plane, `own` the global label planes this rank owns and `global_shape` the whole volume's shape; the labels returned are the
    `own` planes.  `n_slabs` / `bounds` cut a part (or, in one process, the volume) into sub-slabs; `bounds` are plane indices local to
    the owned part."""
    if not (torch.is_tensor(volume) and volume.is_cuda):
        raise RuntimeError("raygun_amd.predict_and_segment keeps the whole flow in HBM: pass the volume as a CUDA/HIP tensor")
    if global_shape is not None:
        if own is None:
            raise ValueError("raygun_amd.predict_and_segment: a rank's part needs `own` (the label planes it owns) next to z0 / global_shape — "
                             "take src / own from pipeline.shard_plan: a part without its read halo leaves 2 x context unpredicted planes at every rank boundary")
        lo, hi = int(own[0]) - int(z0), int(own[1]) - int(z0)
        if not (0 <= lo < hi <= volume.shape[0]) or tuple(global_shape[1:]) != tuple(volume.shape[1:]):
            raise ValueError(f"own planes {tuple(own)} do not lie inside the part [{z0}, {z0 + volume.shape[0]}) of a {tuple(global_shape)} volume")
    dev = volume.device
    t0 = time.perf_counter()
    # rank = 0, world = 1: the part is already this rank's share — its blocks are not sharded a second time
    dests, nblk = P.predict_volume(model, volume, input_shape=list(input_shape), output_shape=list(output_shape),
                                   out_dtypes=None if aff_output is None else [np.uint8] * (aff_output + 1) + [np.uint8] * 8, precision=precision,
                                   rank=0, world=1)
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
    if global_shape is not None:   # the owned planes of the part: a Z view of every channel (addressed through the channel stride, never copied)
        affs8, m = affs8[:, lo:hi], m[lo:hi].contiguous()
    w = S.U8Affinities(affs8, 1.0 / 255.0, sep)
    S._stage("mask (rg_mws_prepare_u8)", dev)
    need = -(-(w.shape[0] * affs8[0].numel()) // S.SINGLE_PASS_EDGE_IDS)
    if global_shape is not None or bounds is not None or max(need, n_slabs or 1) > 1:
        labels = S.segment_slabs(w, nb, sep, strides=list(strides), mask=m, n_slabs=max(need, n_slabs or 1), bounds=bounds, device=dev, as_tensor=True,
                                 z0=0 if global_shape is None else int(own[0]), global_shape=global_shape, group=group)
    else:
        labels = S.compute_mws_segmentation(w, nb, sep, strides=list(strides), mask=m, device=dev, as_tensor=True)
    torch.cuda.synchronize(dev)
    t2 = time.perf_counter()
    info = dict(affinities=affs8, blocks=nblk, predict_s=t1 - t0, segment_s=t2 - t1, total_s=t2 - t0)
    if timings is not None:
        timings.update({k: v for k, v in info.items() if k.endswith("_s")})
    return labels, info

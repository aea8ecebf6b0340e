"""Mutex-watershed segmentation on the GPU — drop-in for the mutex path of raygun.segment
(reference: src/raygun/segment.py:139-230 `mutex_segment`, :233-248 `segment`).

`compute_mws_segmentation` has affogato's call signature as used at segment.py:197-199 and runs
rg_mws_segment (edge keys -> device radix sort -> wavefront-level union-find with mutex lists).  The seeded
watershed + waterz agglomeration branch of segment.py (:46-136) is a different algorithm and out of scope
(SURVEY.md §2 #13b): `segment(config)` with mutex=False raises.

Volumes are taken from the config's zarr container (the `zarr` package when importable, else raygun_amd.zarrio), or
passed in memory (`affs=` keyword / config["affs"] as an ndarray or a .npy path) — the reference reads the whole array
into RAM anyway (segment.py:183-186).
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import rg_mws_params
from .read_config import read_config

DEFAULT_NEIGHBORHOOD = [[1, 0, 0], [0, 1, 0], [0, 0, 1], [2, 0, 0], [0, 2, 0], [0, 0, 2],
                        [4, 0, 0], [0, 4, 0], [0, 0, 4], [8, 0, 0], [0, 8, 0], [0, 0, 8]]


def build_neighborhood(neighborhood=None, n_diagonals=8):
    """segment.py:144-157,173-181: 12 axis offsets + n in-plane diagonals (int8)."""
    nb = np.array(DEFAULT_NEIGHBORHOOD if neighborhood is None else neighborhood)
    if n_diagonals > 0:
        t = np.linspace(0, np.pi, num=n_diagonals, endpoint=False)
        pos_diag = np.round(n_diagonals * np.sin(t))
        neg_diag = np.round(n_diagonals * np.cos(t))
        nb = np.concatenate([nb, np.stack([0 * pos_diag, pos_diag, neg_diag], axis=-1)]).astype(np.int8)
    return nb


def compute_mws_segmentation(weights, offsets, number_of_attractive_channels, strides=None, randomize_strides=False,
                             invert_repulsive_weights=True, bias_cut=0., mask=None, algorithm="kruskal", device=None,
                             as_tensor=False, origin=(0, 0, 0)):
    """GPU mutex watershed with affogato's call signature (affogato.segmentation.compute_mws_segmentation as published:
    weights, offsets, number_of_attractive_channels, strides=None, randomize_strides=False, invert_repulsive_weights=True,
    bias_cut=0., mask=None, algorithm='kruskal'; the reference's call site is segment.py:197-199).

    weights: (n_offsets, Z, Y, X) float32 (numpy or torch), offsets: (n_offsets, 3), mask: (Z, Y, X) bool or None (True =
    valid).  As in affogato the weights are transformed before they are ranked, in float32: the repulsive channels become
    `w * -1 + 1` when `invert_repulsive_weights` (the default — so the reference, which already passes 1 - affinity there,
    ranks its repulsive edges by the twice-inverted value), then `bias_cut` is added to the attractive channels.  Unlike
    affogato the caller's array is left untouched.  Only algorithm='kruskal' is built ('prim' yields the same partition for
    distinct weights; its tie order is affogato-internal).  Returns a uint64 ndarray (Z, Y, X), 0 = masked out; labels are
    1 + rank of the cluster's first voxel in raster order — affogato returns union-find root ids (+1 with a mask), so
    only the PARTITION is the portable claim (parity unpinned).  as_tensor=True keeps the labels on the device
    (int64 tensor, same bits).  `origin`: global coordinate of voxel (0,0,0) for the stride grid (slabs of a larger volume)."""
    if randomize_strides:
        raise NotImplementedError("raygun_amd: randomize_strides is not built (the reference passes strides only)")
    if algorithm != "kruskal":
        if algorithm == "prim":
            raise NotImplementedError("raygun_amd: algorithm='prim' is not built (the reference uses the default, 'kruskal')")
        raise AssertionError("Unsupported algorithm, %s" % algorithm)  # affogato asserts
    dev = torch.device(device if device is not None else "cuda")
    if dev.type != "cuda":
        raise RuntimeError("raygun_amd.compute_mws_segmentation runs on the MI355X only (no CPU fallback)")
    w = torch.as_tensor(np.ascontiguousarray(weights, dtype=np.float32) if isinstance(weights, np.ndarray) else weights)
    w = w.to(dev, torch.float32).contiguous()
    if w.dim() != 4 or w.shape[0] != len(offsets):
        raise ValueError(f"weights must be (n_offsets, Z, Y, X); got {tuple(w.shape)} for {len(offsets)} offsets")
    off = torch.as_tensor(np.ascontiguousarray(offsets, dtype=np.int32)).to(dev)
    m = None
    if mask is not None:
        m = torch.as_tensor(np.ascontiguousarray(mask, dtype=np.uint8) if isinstance(mask, np.ndarray) else mask).to(dev, torch.uint8).contiguous()
        if tuple(m.shape) != tuple(w.shape[1:]):
            raise ValueError("mask shape must equal the spatial shape of the weights")
    p = rg_mws_params()
    for i in range(3):
        p.shape[i] = w.shape[1 + i]
        p.strides[i] = 1 if strides is None else int(strides[i])
        p.origin[i] = int(origin[i])
    p.n_offsets, p.n_attractive, p.randomize_strides = w.shape[0], int(number_of_attractive_channels), 0
    p.invert_repulsive_weights, p.bias_cut = int(bool(invert_repulsive_weights)), float(bias_cut)
    labels = torch.zeros(tuple(w.shape[1:]), dtype=torch.int64, device=dev)
    if labels.numel() == 0:
        return labels if as_tensor else labels.cpu().numpy().astype(np.uint64)
    L = _lib.lib()
    nbytes = L.rg_mws_workspace_bytes(C.byref(p))
    if nbytes < 0:
        raise RuntimeError("raygun_amd.compute_mws_segmentation: volume too large for one segment "
                           "(n_offsets * voxels must stay below 2^32) — segment it in blocks (segment_blocks)")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(L.rg_mws_segment(C.c_void_p(w.data_ptr()), C.c_void_p(off.data_ptr()),
                                    C.c_void_p(m.data_ptr()) if m is not None else None, C.c_void_p(labels.data_ptr()),
                                    C.byref(p), C.c_void_p(ws.data_ptr()), nbytes,
                                    C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "rg_mws_segment")
    if as_tensor:
        return labels  # all-ones labels (= -1) would mean the device pass made no progress; callers that keep the tensor check
    out = labels.cpu().numpy().astype(np.uint64)
    if out.flat[0] == np.uint64(0xFFFFFFFFFFFFFFFF):
        raise RuntimeError("raygun_amd.compute_mws_segmentation: the device pass made no progress (internal error)")
    return out


def prepare_weights(affs, sep, mask_thresh, max_affinity_value, device=None):
    """segment.py:188-194 on the device (rg_mws_prepare): mask = channel mean > mask_thresh * max_affinity_value,
    weights = 1 - affs with the attractive channels flipped back — rounded exactly as numpy's float32 arithmetic.
    affs: (n_offsets, Z, Y, X) float32 ndarray or tensor.  Returns device tensors (weights float32, mask uint8)."""
    dev = torch.device(device if device is not None else "cuda")
    if dev.type != "cuda":
        raise RuntimeError("raygun_amd.prepare_weights runs on the MI355X only (no CPU fallback)")
    a = torch.as_tensor(np.ascontiguousarray(affs, dtype=np.float32) if isinstance(affs, np.ndarray) else affs)
    a = a.to(dev, torch.float32).contiguous()
    if a.dim() != 4:
        raise ValueError(f"affs must be (n_offsets, Z, Y, X); got {tuple(a.shape)}")
    w = torch.empty_like(a)
    m = torch.empty(tuple(a.shape[1:]), dtype=torch.uint8, device=dev)
    if m.numel():
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().rg_mws_prepare(C.c_void_p(a.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(m.data_ptr()),
                                                 m.numel(), a.shape[0], int(sep), float(np.float32(mask_thresh * max_affinity_value)),
                                                 C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "rg_mws_prepare")
    return w, m


def segment_blocks(weights, offsets, sep, strides, mask, block_shape, rank=0, world=1):
    """Independent mutex watershed per block (no stitching: labels are made unique by a per-block offset).  With
    world > 1 the blocks are dealt to ranks statically (parallel.shard_blocks) and each rank returns its own blocks'
    labels, zeros elsewhere: 'replicas only' in SURVEY.md §8e terms."""
    from .parallel import shard_blocks
    Z, Y, X = weights.shape[1:]
    bz, by, bx = block_shape
    blocks = [(z, y, x) for z in range(0, Z, bz) for y in range(0, Y, by) for x in range(0, X, bx)]
    out = np.zeros((Z, Y, X), dtype=np.uint64)
    for bi in shard_blocks(len(blocks), rank, world):
        z, y, x = blocks[bi]
        sl = (slice(z, min(z + bz, Z)), slice(y, min(y + by, Y)), slice(x, min(x + bx, X)))
        seg = compute_mws_segmentation(weights[(slice(None),) + sl], offsets, sep, strides, mask=None if mask is None else mask[sl], origin=(z, y, x))
        out[sl] = np.where(seg > 0, seg + np.uint64(bi) * np.uint64(bz * by * bx), 0)
    return out


def mutex_segment(config_path, affs=None):
    """segment.py:139-230.  Returns the uint64 label volume; writes it next to the affinities when the input came
    from a zarr container and config['save'] is not False."""
    seg_config = {"aff_ds": "pred_affs", "max_affinity_value": 1.0, "sep": 3, "neighborhood": DEFAULT_NEIGHBORHOOD,
                  "n_diagonals": 8, "mask_thresh": 0.5}
    seg_config.update(read_config(config_path))
    neighborhood = build_neighborhood(seg_config["neighborhood"], seg_config["n_diagonals"])
    f = None
    if affs is None:
        affs = seg_config.get("affs")
    if isinstance(affs, str):
        affs = np.load(affs)
    if affs is None:
        try:
            import zarr
        except ImportError:
            from . import zarrio as zarr  # zarr v2 directory stores with stdlib codecs (see zarrio.py)
        f = zarr.open(seg_config["file"], "a")
        affs = f[seg_config["aff_ds"]][:]
    if not torch.is_tensor(affs):
        affs = np.asarray(affs, dtype=np.float32)
    weights, mask = prepare_weights(affs, seg_config["sep"], seg_config["mask_thresh"], seg_config["max_affinity_value"])
    seg = compute_mws_segmentation(weights, neighborhood, seg_config["sep"], strides=[10, 10, 10], mask=mask)
    if ("save" in seg_config and not seg_config["save"]) or f is None:
        return seg
    dest = seg_config.get("dest_dataset", f"mutex_{'{:.2f}'.format(seg_config['mask_thresh'])}")
    f[dest] = seg
    f[dest].attrs["offset"] = f[seg_config["aff_ds"]].attrs["offset"]
    f[dest].attrs["resolution"] = f[seg_config["aff_ds"]].attrs["resolution"]
    return seg


def segment(config_path=None, affs=None):
    """segment.py:233-296 — only the mutex branch is built."""
    cfg = read_config(config_path)
    if cfg.get("mutex", False):
        return mutex_segment(config_path, affs=affs)
    raise NotImplementedError("raygun_amd.segment: the seeded-watershed + waterz branch (segment.py:250-296) is out of "
                              "scope; set \"mutex\": true")

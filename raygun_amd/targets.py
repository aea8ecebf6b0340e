"""Train-time targets of the MTLSD system, computed on the GPU from a label crop — the three nodes the reference's data pipe runs on
the CPU for every batch (reference: src/raygun/io/MTLSDDataPipe.py:99-131):

    add_affinities(labels, neighborhood, labels_mask)      gp.AddAffinities              (:107-114)
    balance_labels(affs, mask)                             gp.BalanceLabels              (:126-130)
    lsd_targets(labels, sigma, voxel_size, downsample)     lsd's AddLocalShapeDescriptor (:100-105)

All three call libraygun_hip.so kernels (csrc/targets.hip); CPU tensors raise (the CPU restatement, oracle/targets.py, is test
infrastructure).  Parity unpinned against gunpowder / lsd themselves (not installed, not vendored): the published algorithms restated."""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _labels(labels, what):
    if not torch.is_tensor(labels) or not labels.is_cuda:
        raise RuntimeError(f"raygun_amd.targets.{what}: expected a CUDA/HIP tensor (the HIP path has no CPU fallback)")
    if labels.dim() != 3:
        raise ValueError(f"raygun_amd.targets.{what}: labels must be (Z, Y, X), got {tuple(labels.shape)}")
    return labels.to(torch.int32).contiguous()      # (gunpowder casts the labels to int32 as well)


def _roi(shape, out_offset, out_shape):
    off = (0, 0, 0) if out_offset is None else tuple(int(v) for v in out_offset)
    osh = tuple(int(s) - o for s, o in zip(shape, off)) if out_shape is None else tuple(int(v) for v in out_shape)
    return off, osh


def add_affinities(labels, neighborhood, labels_mask=None, out_offset=None, out_shape=None):
    """-> (affs float32 (K, *out_shape), affs_mask float32 | None).  affs[e][p] = 1 where labels[p] == labels[p + neighborhood[e]] and both
    are foreground, 0 elsewhere and where p + neighborhood[e] leaves the array; the mask is the same graph of `labels_mask`.  The ROI
    [out_offset, out_offset + out_shape) selects the part gunpowder would crop out after growing the request by the neighbourhood."""
    lab = _labels(labels, "add_affinities")
    dev = lab.device
    nb = torch.as_tensor(np.ascontiguousarray(np.asarray(neighborhood).reshape(-1, 3), dtype=np.int32)).to(dev)
    off, osh = _roi(lab.shape, out_offset, out_shape)
    K = int(nb.shape[0])
    affs = torch.empty((K,) + osh, dtype=torch.float32, device=dev)
    m = am = None
    if labels_mask is not None:
        m = labels_mask.to(dev, torch.uint8).contiguous()
        if tuple(m.shape) != tuple(lab.shape):
            raise ValueError("labels_mask must have the labels' shape")
        am = torch.empty_like(affs)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().rg_add_affinities(C.c_void_p(lab.data_ptr()), C.c_void_p(m.data_ptr()) if m is not None else None, C.c_void_p(nb.data_ptr()), K,
                                                _lib.i3(lab.shape), _lib.i3(off), _lib.i3(osh), C.c_void_p(affs.data_ptr()),
                                                C.c_void_p(am.data_ptr()) if am is not None else None, _stream(dev)), "rg_add_affinities")
    return affs, am


def balance_labels(labels, mask=None, clipmin=0.05, clipmax=0.95):
    """-> float32 loss weights shaped like `labels` (0 / 1 values): mask * w[label] with w[c] = 1 / 2 / clip(share of class c among the
    masked-in voxels) — gp.BalanceLabels with its defaults (one slab, two classes)."""
    if not torch.is_tensor(labels) or not labels.is_cuda:
        raise RuntimeError("raygun_amd.targets.balance_labels: expected a CUDA/HIP tensor (the HIP path has no CPU fallback)")
    lab = labels.to(torch.float32).contiguous()
    dev = lab.device
    m = None if mask is None else mask.to(dev, torch.float32).contiguous()
    if m is not None and m.shape != lab.shape:
        raise ValueError("mask must have the labels' shape")
    scale = torch.empty_like(lab)
    L = _lib.lib()
    nb = int(L.rg_balance_labels_workspace_bytes())
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(L.rg_balance_labels(C.c_void_p(lab.data_ptr()), C.c_void_p(m.data_ptr()) if m is not None else None, lab.numel(), float(clipmin),
                                       float(clipmax), C.c_void_p(scale.data_ptr()), C.c_void_p(ws.data_ptr()), nb, _stream(dev)), "rg_balance_labels")
    return scale


def gaussian_kernel1d(sigma, truncate=3.0):
    """The 1-D kernel scipy.ndimage.gaussian_filter uses (order 0): radius int(truncate * sigma + 0.5), weights normalised to sum 1."""
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (float(sigma) * float(sigma)) * x ** 2)
    return phi / phi.sum()


def lsd_targets(labels, sigma, voxel_size=(1, 1, 1), downsample=1, out_offset=None, out_shape=None):
    """-> float32 (10, *out_shape): lsd's local shape descriptors (Gaussian mode) of the segments in `labels` — mean offset z y x,
    variance z y x, Pearson coefficients zy zx yx, mass — evaluated on the grid subsampled by `downsample`, normalised into [0, 1]
    and zero on the background.  sigma (scalar or per axis) and voxel_size in world units (MTLSD default: sigma 300, downsample 2)."""
    lab = _labels(labels, "lsd_targets")
    dev = lab.device
    df = int(downsample)
    vs = np.asarray(voxel_size, dtype=np.float64)
    sig = np.full(3, float(sigma)) if np.isscalar(sigma) else np.asarray(sigma, dtype=np.float64)
    sub_vs = vs * df
    ws = [gaussian_kernel1d(s / v) for s, v in zip(sig, sub_vs)]
    wd = [torch.as_tensor(np.ascontiguousarray(w, dtype=np.float32)).to(dev) for w in ws]
    off, osh = _roi(lab.shape, out_offset, out_shape)
    out = torch.empty((10,) + osh, dtype=torch.float32, device=dev)
    f3 = C.c_float * 3
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().rg_lsd_targets(C.c_void_p(lab.data_ptr()), _lib.i3(lab.shape), _lib.i3(off), _lib.i3(osh), df, C.c_void_p(wd[0].data_ptr()),
                                             C.c_void_p(wd[1].data_ptr()), C.c_void_p(wd[2].data_ptr()), _lib.i3([len(w) // 2 for w in ws]),
                                             f3(*[float(v) for v in sub_vs]), f3(*[float(v) for v in sig]), C.c_void_p(out.data_ptr()), _stream(dev)),
                   "rg_lsd_targets")
    return out

"""The default `raygun-segment` route on the GPU — drop-in for the seeded-watershed + agglomeration half of raygun.segment
(reference: src/raygun/segment.py:15-44 `watershed_from_boundary_distance`, :47-108 `watershed_from_affinities`, :116-136 `get_segmentation`;
`segment` :233-296 dispatches here unless the config says "mutex": true).

Same function names, arguments and return conventions as the reference (numpy in -> numpy uint64 out), the work done by rg_ws_* (csrc/watershed.hip):

    scipy.ndimage.distance_transform_edt   -> rg_ws_edt      exact, kept as integer squares (their order is the order of the distances)
    scipy.ndimage.maximum_filter + ==      -> rg_ws_maxima   scipy's window and reflected borders
    scipy.ndimage.label                    -> rg_ws_label    6-connectivity, numbered by first voxel in raster order
    skimage.segmentation.watershed         -> rg_ws_flood    BUILD-DEFINED tie order: minimum spanning forest rooted at the seeds (see the kernel file);
                                                             skimage's own result hangs on heap insertion order — parity unpinned, oracle/watershed.py
    waterz.agglomerate                     -> rg_ws_rag_* (region graph + 256-bin affinity histograms on the GPU), rg_ws_agglomerate (the sequential
                                              merge queue, host C++ inside the library), rg_ws_relabel — restated from waterz's published sources
                                              (OneMinus<HistogramQuantileAffinity<RegionGraphType, 75, ScoreValue, 256, false>>), parity unpinned

There is no CPU fallback: without the HIP library or a GPU every function raises."""
import ctypes as C

import numpy as np
import torch

from . import _lib

MAX_FRAGMENT_ID = (1 << 28) - 1     # the region-graph records pack two fragment ids and a histogram bin into 64 bits


def _dev(device=None):
    dev = torch.device(device if device is not None else "cuda")
    if dev.type != "cuda":
        raise RuntimeError("raygun_amd.watershed runs on the MI355X only (no CPU fallback)")
    return dev


def _st(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _shape3(shape):
    return _lib.i3([int(s) for s in shape])


def _to(dev, a, dtype):
    t = torch.as_tensor(np.ascontiguousarray(a) if isinstance(a, np.ndarray) else a)
    return t.to(dev, dtype).contiguous()


def boundary_mask(affs, max_affinity_value=1.0, device=None):
    """segment.py:93 — np.mean(affs, axis=0) > 0.5 * max_affinity_value, in numpy's float32 arithmetic.  affs: (K, Z, Y, X) float32.  Returns a uint8 device tensor."""
    dev = _dev(device)
    a = _to(dev, affs, torch.float32)
    if a.dim() != 4:
        raise ValueError(f"affs must be (channels, Z, Y, X); got {tuple(a.shape)}")
    m = torch.empty(tuple(a.shape[1:]), dtype=torch.uint8, device=dev)
    if m.numel():
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().rg_ws_mask(C.c_void_p(a.data_ptr()), a.shape[0], m.numel(), float(np.float32(0.5 * max_affinity_value)),
                                             C.c_void_p(m.data_ptr()), _st(dev)), "rg_ws_mask")
    return m


def squared_distance_transform(mask, device=None):
    """scipy.ndimage.distance_transform_edt(mask) ** 2 as exact int32 (segment.py:94): squared distance to the nearest False voxel of the volume."""
    dev = _dev(device)
    m = _to(dev, mask, torch.uint8)
    if m.dim() != 3:
        raise ValueError("mask must be (Z, Y, X)")
    if m.numel() and bool(m.all().item()):
        raise ValueError("raygun_amd.watershed: the mask has no boundary (False) voxel — the distance transform is undefined")
    d2 = torch.empty(tuple(m.shape), dtype=torch.int32, device=dev)
    if m.numel():
        L, sh = _lib.lib(), _shape3(m.shape)
        ws = torch.empty(int(L.rg_ws_edt_workspace_bytes(sh)), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.rg_ws_edt(C.c_void_p(m.data_ptr()), sh, C.c_void_p(d2.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(), _st(dev)), "rg_ws_edt")
    return d2


def seeds_from_distance(d2, min_seed_distance=10):
    """segment.py:25-27 on the squared distances: (int32 labels of the maxima's connected components, their number)."""
    dev = d2.device
    L, sh = _lib.lib(), _shape3(d2.shape)
    maxima = torch.empty(tuple(d2.shape), dtype=torch.uint8, device=dev)
    seeds = torch.empty(tuple(d2.shape), dtype=torch.int32, device=dev)
    n = torch.zeros(1, dtype=torch.int32, device=dev)
    if d2.numel() == 0:
        return seeds, 0
    ws = torch.empty(int(max(L.rg_ws_maxima_workspace_bytes(sh), L.rg_ws_label_workspace_bytes(sh))), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(L.rg_ws_maxima(C.c_void_p(d2.data_ptr()), sh, int(min_seed_distance), C.c_void_p(maxima.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(), _st(dev)),
                   "rg_ws_maxima")
        _lib.check(L.rg_ws_label(C.c_void_p(maxima.data_ptr()), sh, C.c_void_p(seeds.data_ptr()), C.c_void_p(n.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(), _st(dev)),
                   "rg_ws_label")
    return seeds, int(n.item())


def flood(d2, seeds, mask, stats=None):
    """Seeded watershed of (max distance - distance) inside `mask` from the int32 `seeds` (segment.py:36-38), build-defined tie order.  int32 tensor."""
    dev = d2.device
    L, sh = _lib.lib(), _shape3(d2.shape)
    out = torch.empty(tuple(d2.shape), dtype=torch.int32, device=dev)
    if d2.numel() == 0:
        return out
    ws = torch.empty(int(L.rg_ws_flood_workspace_bytes(sh)), dtype=torch.uint8, device=dev)
    rounds = C.c_int32(0)
    with torch.cuda.device(dev):
        _lib.check(L.rg_ws_flood(C.c_void_p(d2.data_ptr()), C.c_void_p(seeds.data_ptr()), C.c_void_p(mask.data_ptr()), sh, C.c_void_p(out.data_ptr()), C.byref(rounds),
                                 C.c_void_p(ws.data_ptr()), ws.numel(), _st(dev)), "rg_ws_flood")
    if stats is not None:
        stats["rounds"] = int(rounds.value)
    return out


def _fragments(d2, mask, id_offset, min_seed_distance, stats=None):
    seeds, n = seeds_from_distance(d2, min_seed_distance)
    if n == 0:
        return None, id_offset, None
    if id_offset:
        seeds = torch.where(seeds != 0, seeds + int(id_offset), seeds)     # seeds[seeds != 0] += id_offset (index shift, no arithmetic on data)
    return flood(d2, seeds, mask, stats), n + id_offset, seeds


def watershed_from_boundary_distance(boundary_distances, boundary_mask, return_seeds=False, id_offset=0, min_seed_distance=10, as_tensor=False, device=None):
    """segment.py:16-44.  `boundary_distances`: the Euclidean distance transform (float array, as the reference passes it) or an int32 tensor of its SQUARES
    (what squared_distance_transform returns).  Returns (fragments uint64, n + id_offset[, seeds uint64])."""
    dev = _dev(device)
    if torch.is_tensor(boundary_distances) and boundary_distances.dtype == torch.int32:
        d2 = boundary_distances.to(dev).contiguous()
    else:
        d = np.asarray(boundary_distances.cpu() if torch.is_tensor(boundary_distances) else boundary_distances, dtype=np.float64)
        d2 = torch.from_numpy(np.rint(d * d).astype(np.int32)).to(dev)      # exact: an EDT's squares are integers
    m = _to(dev, boundary_mask, torch.uint8)
    frag, n, seeds = _fragments(d2, m, id_offset, min_seed_distance)
    conv = (lambda t: t.to(torch.int64)) if as_tensor else (lambda t: t.cpu().numpy().astype(np.uint64))
    if frag is None:
        z = torch.zeros(tuple(d2.shape), dtype=torch.int64, device=dev) if as_tensor else np.zeros(tuple(d2.shape), dtype=np.uint64)
        return z, id_offset
    ret = (conv(frag), n)
    if return_seeds:
        ret = ret + (conv(seeds),)
    return ret


def watershed_from_affinities(affs, max_affinity_value=1.0, fragments_in_xy=False, return_seeds=False, min_seed_distance=10, labels_mask=None, as_tensor=False,
                              device=None, stats=None):
    """segment.py:47-108, both branches.  affs: (K, Z, Y, X) float32 (numpy or tensor)."""
    dev = _dev(device)
    a = _to(dev, affs, torch.float32)
    conv = (lambda t: t.to(torch.int64)) if as_tensor else (lambda t: t.cpu().numpy().astype(np.uint64))
    if fragments_in_xy:
        # per section: mean of the two in-plane channels (0.5 * (a1 + a2) == (a1 + a2) / 2 exactly), 2-D distance transform, 2-D watershed, ids offset section by section
        Z = a.shape[1]
        frags = torch.zeros(tuple(a.shape[1:]), dtype=torch.int32, device=dev)
        seeds_all = torch.zeros_like(frags)
        id_offset = 0
        lm = None if labels_mask is None else _to(dev, np.asarray(labels_mask).astype(bool), torch.uint8)
        for z in range(Z):
            bm = boundary_mask(a[1:3, z:z + 1], max_affinity_value, dev)
            d2 = squared_distance_transform(bm, dev)
            if lm is not None:
                bm = bm * (lm if lm.dim() == 2 else lm[z])[None]
            f, id_offset, s = _fragments(d2, bm, id_offset, min_seed_distance)
            if f is not None:
                frags[z], seeds_all[z] = f[0], s[0]
        ret = (conv(frags), id_offset)
        return ret + ((conv(seeds_all),) if return_seeds else ())
    bm = boundary_mask(a, max_affinity_value, dev)
    d2 = squared_distance_transform(bm, dev)
    frag, n, seeds = _fragments(d2, bm, 0, min_seed_distance, stats)
    if frag is None:
        z = torch.zeros(tuple(d2.shape), dtype=torch.int64, device=dev) if as_tensor else np.zeros(tuple(d2.shape), dtype=np.uint64)
        return z, 0
    ret = (conv(frag), n)
    return ret + ((conv(seeds),) if return_seeds else ())


def agglomerate(affs, fragments, thresholds, as_tensor=False, device=None, stats=None):
    """waterz.agglomerate(affs=affs, fragments=fragments, thresholds=thresholds, scoring_function="OneMinus<HistogramQuantileAffinity<RegionGraphType, 75,
    ScoreValue, 256, false>>") — the reference's call, segment.py:126-131.  Yields one uint64 (Z, Y, X) segmentation per threshold (ascending); a segment carries
    the smallest fragment id it contains."""
    dev = _dev(device)
    a = _to(dev, affs, torch.float32)
    if a.dim() != 4 or a.shape[0] != 3:
        raise ValueError(f"waterz agglomerates (3, Z, Y, X) nearest-neighbour affinities; got {tuple(a.shape)}")
    f = torch.as_tensor(np.ascontiguousarray(fragments).astype(np.int64) if isinstance(fragments, np.ndarray) else fragments).to(dev)
    if tuple(f.shape) != tuple(a.shape[1:]):
        raise ValueError("fragments must have the affinities' spatial shape")
    n = int(f.max().item()) if f.numel() else 0
    if n > MAX_FRAGMENT_ID:
        raise ValueError(f"raygun_amd.watershed: fragment ids up to {n} exceed {MAX_FRAGMENT_ID}")
    f32 = f.to(torch.int32).contiguous()
    L, sh = _lib.lib(), _shape3(f.shape)
    thr = np.ascontiguousarray([np.float32(t) for t in thresholds], dtype=np.float32)
    keys = counts = None
    if f.numel():
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.rg_ws_rag_count(C.c_void_p(f32.data_ptr()), sh, C.c_void_p(cnt.data_ptr()), _st(dev)), "rg_ws_rag_count")
            cap = int(cnt.item())
            rec = torch.empty(max(cap, 1), dtype=torch.int64, device=dev)
            _lib.check(L.rg_ws_rag_emit(C.c_void_p(f32.data_ptr()), C.c_void_p(a.data_ptr()), sh, C.c_void_p(rec.data_ptr()), cap, C.c_void_p(cnt.data_ptr()), _st(dev)),
                       "rg_ws_rag_emit")
        if int(cnt.item()) != cap:
            raise RuntimeError("raygun_amd.watershed: region-graph record count changed between the two passes (internal error)")
        # sorted unique records + multiplicities = the region graph's edges with their sparse histograms (ids < 2^28: the int64 order is the uint64 order)
        keys, counts = torch.unique(rec[:cap], sorted=True, return_counts=True)
        keys, counts = keys.cpu().numpy().astype(np.uint64), counts.cpu().numpy().astype(np.int64)
    nrec = 0 if keys is None else len(keys)
    lut = np.zeros((len(thr), n + 1), dtype=np.int64)
    if len(thr):
        _lib.check(L.rg_ws_agglomerate(keys.ctypes.data_as(C.c_void_p) if nrec else None, counts.ctypes.data_as(C.c_void_p) if nrec else None, nrec, n,
                                       thr.ctypes.data_as(C.c_void_p), len(thr), lut.ctypes.data_as(C.c_void_p)), "rg_ws_agglomerate")
    if stats is not None:
        stats.update(records=nrec, fragments=n)
    lut_d = torch.from_numpy(lut).to(dev)
    for t in range(len(thr)):
        out = torch.empty(tuple(f.shape), dtype=torch.int64, device=dev)
        if f.numel():
            with torch.cuda.device(dev):
                _lib.check(L.rg_ws_relabel(C.c_void_p(f32.data_ptr()), C.c_void_p(lut_d[t].data_ptr()), C.c_void_p(out.data_ptr()), f.numel(), _st(dev)), "rg_ws_relabel")
        yield out if as_tensor else out.cpu().numpy().astype(np.uint64)


def get_segmentation(affinities, thresholds, labels_mask=None, max_affinity_value=None, as_tensor=False, device=None):
    """segment.py:116-136: fragments from the affinities, then one agglomerated segmentation per threshold."""
    dev = _dev(device)
    a = _to(dev, affinities, torch.float32)
    if max_affinity_value is None:
        max_affinity_value = float(a.max().item())
    fragments = watershed_from_affinities(a, max_affinity_value=max_affinity_value, labels_mask=labels_mask, as_tensor=True, device=dev)[0]
    if not isinstance(thresholds, list):
        thresholds = [thresholds]
    return [seg for seg in agglomerate(a, fragments, thresholds, as_tensor=as_tensor, device=dev)]

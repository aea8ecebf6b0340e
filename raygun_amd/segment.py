"""Mutex-watershed segmentation on the GPU — drop-in for the mutex path of raygun.segment
(reference: src/raygun/segment.py:139-230 `mutex_segment`, :233-248 `segment`).

`compute_mws_segmentation` has affogato's call signature as used at segment.py:197-199 and runs
rg_mws_segment (edge keys -> device radix sort -> wavefront-level union-find with mutex lists).  The DEFAULT route of
segment.py (:46-136: seeded watershed on the boundary-distance transform + waterz agglomeration) lives in
raygun_amd/watershed.py; `segment(config)` dispatches to it unless the config says "mutex": true.

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

_SEG_TIMING = bool(os.environ.get("RAYGUN_AMD_SEG_TIMING"))   # diagnostics: stage times of segment_slabs / pipeline on stderr (synchronises)
_seg_t = [0.0]


def _stage(tag, dev=None):
    if not _SEG_TIMING:
        return
    import sys
    import time
    torch.cuda.synchronize(dev)
    now = time.perf_counter()
    if tag is not None:
        sys.stderr.write(f"[seg] {tag}: {(now - _seg_t[0]) * 1e3:.1f} ms  (allocated {torch.cuda.memory_allocated(dev) / 2**30:.1f} GiB, reserved {torch.cuda.memory_reserved(dev) / 2**30:.1f} GiB)\n")
    _seg_t[0] = time.perf_counter()



def build_neighborhood(neighborhood=None, n_diagonals=8):
    """segment.py:144-157,173-181: 12 axis offsets + n in-plane diagonals (int8)."""
    nb = np.array(DEFAULT_NEIGHBORHOOD if neighborhood is None else neighborhood)
    if n_diagonals > 0:
        t = np.linspace(0, np.pi, num=n_diagonals, endpoint=False)
        pos_diag = np.round(n_diagonals * np.sin(t))
        neg_diag = np.round(n_diagonals * np.cos(t))
        nb = np.concatenate([nb, np.stack([0 * pos_diag, pos_diag, neg_diag], axis=-1)]).astype(np.int8)
    return nb


class U8Affinities:
    """The block runner's uint8 affinities (n_offsets, Z, Y, X) on the GPU standing in for the float32 weight array of segment.py:188-199:
    weights[k] = 1 - u8 * scale, flipped back for k < sep (rg_mws_prepare's arithmetic, applied inside the edge kernels).  Accepted wherever
    compute_mws_segmentation / segment_slabs take `weights`; Z-slabs are views of `t` (each channel contiguous), never copies."""

    def __init__(self, t, scale, sep):
        if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.uint8 and t.dim() == 4):
            raise RuntimeError("raygun_amd.U8Affinities: a (n_offsets, Z, Y, X) uint8 tensor on the MI355X expected")
        if t.stride(3) != 1 or t.stride(2) != t.shape[3] or t.stride(1) != t.shape[2] * t.shape[3]:
            t = t.contiguous()
        self.t, self.scale, self.sep = t, float(np.float32(scale)), int(sep)

    @property
    def shape(self):
        return self.t.shape

    def slab(self, z0, z1):
        return U8Affinities(self.t[:, z0:z1], self.scale, self.sep)


def compute_mws_segmentation(weights, offsets, number_of_attractive_channels, strides=None, randomize_strides=False,
                             invert_repulsive_weights=True, bias_cut=0., mask=None, algorithm="kruskal", device=None,
                             as_tensor=False, origin=(0, 0, 0), stats=None):
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
    u8 = weights if isinstance(weights, U8Affinities) else None
    if u8 is not None:
        w = u8.t
    else:
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
        raise RuntimeError("raygun_amd.compute_mws_segmentation: volume too large for one pass "
                           "(n_offsets * voxels must stay below 2^32) — use segment_slabs (Z-slabs + one exchange step)")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        if u8 is not None:
            _lib.check(L.rg_mws_segment_u8(C.c_void_p(w.data_ptr()), u8.scale, u8.sep, int(w.stride(0)), C.c_void_p(off.data_ptr()),
                                           C.c_void_p(m.data_ptr()) if m is not None else None, C.c_void_p(labels.data_ptr()),
                                           C.byref(p), C.c_void_p(ws.data_ptr()), nbytes,
                                           C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "rg_mws_segment_u8")
        else:
            _lib.check(L.rg_mws_segment(C.c_void_p(w.data_ptr()), C.c_void_p(off.data_ptr()),
                                        C.c_void_p(m.data_ptr()) if m is not None else None, C.c_void_p(labels.data_ptr()),
                                        C.byref(p), C.c_void_p(ws.data_ptr()), nbytes,
                                        C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "rg_mws_segment")
        if stats is not None:  # diagnostics of the round kernel (synchronises)
            buf = (C.c_uint64 * 8)()
            _lib.check(L.rg_mws_stats(C.c_void_p(ws.data_ptr()), C.byref(p), buf, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "rg_mws_stats")
            stats.update(edges=int(buf[0]), rounds=int(buf[1]), window_visits=int(buf[2]), phase_ms=[round(int(buf[3 + i]) / 1e5, 3) for i in range(3)],
                         mutex_nodes=int(buf[6]))
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


def prepare_weights_u8(affs8, scale, sep, mask_thresh, max_affinity_value, device=None, mask_only=False):
    """prepare_weights(affs8.float() * scale, ...) bit for bit, without the float32 affinities: affs8 is the (n_offsets, Z, Y, X) uint8 tensor the
    block runner leaves on the device (worker.py:133-144 stores value * 255 for integer destinations; scale = 1 / 255 takes it back to [0, 1])."""
    dev = torch.device(device if device is not None else "cuda")
    if dev.type != "cuda" or not (torch.is_tensor(affs8) and affs8.is_cuda and affs8.dtype == torch.uint8):
        raise RuntimeError("raygun_amd.prepare_weights_u8: a uint8 tensor on the MI355X expected (no CPU fallback)")
    a = affs8.contiguous()
    if a.dim() != 4:
        raise ValueError(f"affs must be (n_offsets, Z, Y, X); got {tuple(a.shape)}")
    w = None if mask_only else torch.empty(tuple(a.shape), dtype=torch.float32, device=dev)   # mask_only: the weights stay implicit (U8Affinities)
    m = torch.empty(tuple(a.shape[1:]), dtype=torch.uint8, device=dev)
    if m.numel():
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().rg_mws_prepare_u8(C.c_void_p(a.data_ptr()), float(np.float32(scale)), C.c_void_p(w.data_ptr()) if w is not None else None, C.c_void_p(m.data_ptr()),
                                                    m.numel(), a.shape[0], int(sep), float(np.float32(mask_thresh * max_affinity_value)),
                                                    C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "rg_mws_prepare_u8")
    return w, m


SINGLE_PASS_EDGE_IDS = 2 ** 32 - 2 ** 24   # one rg_mws_segment pass numbers its edges n_offsets * voxels < 2^32


def _mws_params(shape, n_off, n_attr, strides, invert, bias, origin):
    p = rg_mws_params()
    for i in range(3):
        p.shape[i], p.strides[i], p.origin[i] = int(shape[i]), 1 if strides is None else int(strides[i]), int(origin[i])
    p.n_offsets, p.n_attractive, p.randomize_strides = int(n_off), int(n_attr), 0
    p.invert_repulsive_weights, p.bias_cut = int(bool(invert)), float(bias)
    return p


def slab_bounds(Z, n_slabs, min_thickness=1):
    """n_slabs Z-slabs of (nearly) equal thickness: [0, b1, ..., Z]."""
    n_slabs = max(1, min(int(n_slabs), Z // max(1, min_thickness)))
    return [round(i * Z / n_slabs) for i in range(n_slabs + 1)]


def segment_slabs(weights, offsets, number_of_attractive_channels, strides=None, mask=None, n_slabs=None, bounds=None,
                  invert_repulsive_weights=True, bias_cut=0., device=None, as_tensor=False, z0=0, global_shape=None, group=None):
    """Mutex watershed of a volume decomposed into Z-slabs, with ONE exchange step: the path past
    the single-pass limit (n_offsets * voxels < 2^32) and across GPUs (BASELINE config 5).  No reference counterpart — affogato
    segments one in-memory array; the result is DEFINED by this decomposition and restated on the CPU by oracle.mws.slab_mws
    (bit-exact test: tests/test_mws.py).  It equals the single-pass result for one slab.

      1. first level: the exact mutex watershed of every slab on its own (rg_mws_segment; stride grid in global coordinates);
      2. exchange: the H = max|dz| label planes either side of every slab boundary, as global cluster ids;
      3. second level: a mutex watershed (rg_mws_graph) over the first-level clusters that reach a boundary zone, with the
         grid edges that cross a boundary (ranked exactly like the first level ranks them) and, ahead of all of them, the
         mutexes the first level left between such clusters inside a slab.  Clusters outside the zones have no attractive
         cross edge, can never merge and are left out — exact, not an approximation of the definition;
      4. relabel: 1 + rank of the cluster's first voxel in global raster order, 0 = masked out (rg_mws_relabel).

    One process, several slabs: pass the whole (n_offsets, Z, Y, X) volume and `n_slabs` / `bounds`.  Several ranks
    (torch.distributed initialised; backend nccl = RCCL, or gloo with the payloads staged through the host): pass THIS rank's
    part of the volume, `z0` = its first global plane and `global_shape`; parts are ordered by rank.  A rank's part is itself cut
    into `n_slabs` sub-slabs (default: as many as the single-pass limit n_offsets * voxels < 2^32 asks for — config 5's
    2048^3 / 8 ranks needs 6 per rank), or at the LOCAL plane indices `bounds`; the global decomposition is the concatenation
    of all ranks' sub-slabs, and the result equals the one-process result with those bounds.  Step 2 is then a neighbour
    send/recv at rank boundaries, step 3 runs replicated on every rank from an all-gather of the (small) edge records.
    Returns this process's labels (uint64 ndarray, or int64 tensor)."""
    import torch.distributed as dist
    from . import parallel
    dev = torch.device(device if device is not None else "cuda")
    if dev.type != "cuda":
        raise RuntimeError("raygun_amd.segment_slabs runs on the MI355X only (no CPU fallback)")
    u8 = weights if isinstance(weights, U8Affinities) else None
    if u8 is not None:
        w = u8.t
    else:
        w = torch.as_tensor(np.ascontiguousarray(weights, dtype=np.float32) if isinstance(weights, np.ndarray) else weights).to(dev, torch.float32).contiguous()
    off_np = np.ascontiguousarray(offsets, dtype=np.int32)
    off = torch.as_tensor(off_np).to(dev)
    K, Zl, Y, X = w.shape
    n_attr = int(number_of_attractive_channels)
    m = None
    if mask is not None:
        m = torch.as_tensor(np.ascontiguousarray(mask, dtype=np.uint8) if isinstance(mask, np.ndarray) else mask).to(dev, torch.uint8).contiguous()
    distributed = global_shape is not None
    world = dist.get_world_size(group) if distributed else 1
    rank = dist.get_rank(group) if distributed else 0
    H = max(1, int(np.abs(off_np[:, 0]).max()))
    zbase = int(z0) if distributed else 0
    Zg = int(global_shape[0]) if distributed else Zl
    if bounds is None:
        auto = -(-(K * Zl * Y * X) // SINGLE_PASS_EDGE_IDS)          # sub-slabs the single-pass limit asks for
        bounds = slab_bounds(Zl, max(int(n_slabs or 1), auto), H)
    mine = [(zbase + int(a), int(b - a)) for a, b in zip(bounds[:-1], bounds[1:])]
    bad = any(t < H for _, t in mine) and (len(mine) > 1 or distributed)
    if distributed:   # every rank must reach the collectives below or none: agree on validity first
        flags = parallel.all_gather_ints([int(bad), len(mine)], group)
        bad = any(f[0] for f in flags)
        nsub = [f[1] for f in flags]
    if bad:
        raise ValueError(f"slabs must be at least max|dz| = {H} planes thick")
    L = _lib.lib()
    st = lambda: C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)  # noqa: E731
    # ---- 1. first level
    _stage(None, dev)
    labs, counts = [], []
    for zs, t in mine:
        zl = zs - zbase
        lab = compute_mws_segmentation(u8.slab(zl, zl + t) if u8 is not None else w[:, zl:zl + t], off_np, n_attr, strides, mask=None if m is None else m[zl:zl + t],
                                       invert_repulsive_weights=invert_repulsive_weights, bias_cut=bias_cut, device=dev,
                                       as_tensor=True, origin=(zs, 0, 0))
        if int(lab.view(-1)[0].item()) == -1:
            raise RuntimeError("raygun_amd.segment_slabs: the device pass made no progress (internal error)")
        labs.append(lab); counts.append(int(lab.max().item()))
        _stage(f"first level, slab z={zs} t={t}", dev)
    if distributed:   # global slab list = every rank's sub-slabs, in rank order
        per_rank = parallel.all_gather_ints(counts, group)
        assert [len(c) for c in per_rank] == nsub
        all_counts = [c for cs in per_rank for c in cs]
        first = sum(nsub[:rank])
        my_ids = list(range(first, first + len(mine)))
    else:
        all_counts, my_ids = counts, list(range(len(mine)))
    S = len(all_counts)
    bases = np.concatenate([[0], np.cumsum(all_counts)]).astype(np.int64)
    if S == 1:
        out = labs[0]
        return out if as_tensor else out.cpu().numpy().astype(np.uint64)
    # ---- 2. boundary zones as global ids
    def glob(lab, sid):
        return torch.where(lab > 0, lab + int(bases[sid]), lab)
    top = {sid: glob(lab[:H], sid).contiguous() for sid, lab in zip(my_ids, labs)}      # first H planes
    bot = {sid: glob(lab[-H:], sid).contiguous() for sid, lab in zip(my_ids, labs)}     # last H planes
    nb_above = {sid: top[sid + 1] for sid in my_ids if sid + 1 in top}
    nb_below = {sid: bot[sid - 1] for sid in my_ids if sid - 1 in bot}
    if distributed:   # the zones either side of a RANK boundary travel once, neighbour to neighbour
        below, above = parallel.neighbor_exchange(top[my_ids[0]] if rank > 0 else None, bot[my_ids[-1]] if rank < world - 1 else None, group)
        if below is not None:
            nb_below[my_ids[0]] = below
        if above is not None:
            nb_above[my_ids[-1]] = above
    # ---- 3a. this process's edge records
    per_dir = {d: int(sum(abs(int(dz)) for dz in off_np[:, 0] if dz * d > 0)) * Y * X for d in (1, -1)}
    rec = dict(wkey=[], gid=[], gu=[], gv=[], attr=[])
    mut = []
    zone_ids = []
    Ng = Zg * Y * X
    for sid, lab, (zs, t) in zip(my_ids, labs, mine):
        zl = zs - zbase
        p = _mws_params((t, Y, X), K, n_attr, strides, invert_repulsive_weights, bias_cut, (zs, 0, 0))
        wsl = w[:, zl:zl + t] if u8 is not None else w[:, zl:zl + t].contiguous()   # (uint8: addressed in place through its channel stride)
        for d, nz in ((1, nb_above.get(sid)), (-1, nb_below.get(sid))):
            if nz is None or per_dir[d] == 0 or (d == 1 and sid == S - 1) or (d == -1 and sid == 0):
                continue
            cap = per_dir[d]
            wk = torch.empty(cap, dtype=torch.int32, device=dev); gid = torch.empty(cap, dtype=torch.int64, device=dev)
            gu = torch.empty(cap, dtype=torch.int64, device=dev); gv = torch.empty(cap, dtype=torch.int64, device=dev)
            at = torch.empty(cap, dtype=torch.uint8, device=dev); cnt = torch.zeros(1, dtype=torch.int32, device=dev)
            with torch.cuda.device(dev):
                if u8 is not None:
                    _lib.check(L.rg_mws_cross_edges_u8(C.c_void_p(wsl.data_ptr()), u8.scale, u8.sep, int(wsl.stride(0)), C.c_void_p(lab.data_ptr()), int(bases[sid]),
                                                       C.c_void_p(nz.data_ptr()), H, d, Ng, C.c_void_p(off.data_ptr()), C.byref(p), C.c_void_p(wk.data_ptr()),
                                                       C.c_void_p(gid.data_ptr()), C.c_void_p(gu.data_ptr()), C.c_void_p(gv.data_ptr()), C.c_void_p(at.data_ptr()),
                                                       C.c_void_p(cnt.data_ptr()), cap, st()), "rg_mws_cross_edges_u8")
                else:
                    _lib.check(L.rg_mws_cross_edges(C.c_void_p(wsl.data_ptr()), C.c_void_p(lab.data_ptr()), int(bases[sid]), C.c_void_p(nz.data_ptr()), H, d, Ng,
                                                    C.c_void_p(off.data_ptr()), C.byref(p), C.c_void_p(wk.data_ptr()), C.c_void_p(gid.data_ptr()),
                                                    C.c_void_p(gu.data_ptr()), C.c_void_p(gv.data_ptr()), C.c_void_p(at.data_ptr()), C.c_void_p(cnt.data_ptr()),
                                                    cap, st()), "rg_mws_cross_edges")
            n = int(cnt.item())
            if n > cap:
                raise RuntimeError("raygun_amd.segment_slabs: cross-edge capacity exceeded (internal error)")
            for k_, v_ in (("wkey", wk), ("gid", gid), ("gu", gu), ("gv", gv), ("attr", at)):
                rec[k_].append(v_[:n])
        # second-level nodes of this slab: clusters with a voxel in a boundary zone
        zl_ids = torch.unique(torch.cat([lab[:H].reshape(-1) if sid > 0 else lab.new_zeros(0), lab[-H:].reshape(-1) if sid < S - 1 else lab.new_zeros(0)]))
        zl_ids = zl_ids[zl_ids > 0]
        zone_ids.append(zl_ids + int(bases[sid]))
        is_node = torch.zeros(all_counts[sid] + 1, dtype=torch.uint8, device=dev)
        is_node[zl_ids] = 1
        nrep = K - n_attr
        if nrep > 0 and zl_ids.numel() > 1:
            sz, sy, sx = (1, 1, 1) if strides is None else [int(v) for v in strides]
            cap = max(1, ((t + sz - 1) // sz + 1) * ((Y + sy - 1) // sy) * ((X + sx - 1) // sx) * nrep)
            pairs = torch.empty(cap, dtype=torch.int64, device=dev); cnt = torch.zeros(1, dtype=torch.int32, device=dev)
            with torch.cuda.device(dev):
                _lib.check(L.rg_mws_slab_mutexes(C.c_void_p(lab.data_ptr()), C.c_void_p(is_node.data_ptr()), C.c_void_p(off.data_ptr()), C.byref(p),
                                                 C.c_void_p(pairs.data_ptr()), C.c_void_p(cnt.data_ptr()), cap, st()), "rg_mws_slab_mutexes")
            n = int(cnt.item())
            if n > cap:
                raise RuntimeError("raygun_amd.segment_slabs: mutex-pair capacity exceeded (internal error)")
            pr = torch.unique(pairs[:n])
            mut.append(torch.stack([(pr >> 32) + int(bases[sid]), (pr & 0xffffffff) + int(bases[sid])], 1))
    _stage("zones + cross edges + slab mutexes", dev)
    cat = lambda xs, dt: torch.cat(xs) if xs else torch.zeros(0, dtype=dt, device=dev)  # noqa: E731
    wkey, gid, gu, gv, attr = cat(rec["wkey"], torch.int32), cat(rec["gid"], torch.int64), cat(rec["gu"], torch.int64), cat(rec["gv"], torch.int64), cat(rec["attr"], torch.uint8)
    mutp = torch.cat(mut) if mut else torch.zeros((0, 2), dtype=torch.int64, device=dev)
    nodes = torch.cat(zone_ids) if zone_ids else torch.zeros(0, dtype=torch.int64, device=dev)
    if distributed:  # replicate the (small) second-level input on every rank
        wkey, gid, gu, gv, attr, mutp, nodes = [parallel.all_gather_cat(t, group) for t in (wkey, gid, gu, gv, attr, mutp, nodes)]
    # ---- 3b. second level on compact node indices (sorted global ids: the smallest index is the smallest id)
    nodes = torch.unique(nodes)
    n_nodes = int(nodes.numel())
    rep_g = nodes.clone()
    if n_nodes > 0 and (wkey.numel() or mutp.numel()):
        nm = int(mutp.shape[0])
        eu = torch.searchsorted(nodes, torch.cat([mutp[:, 0], gu])).to(torch.int32)
        ev = torch.searchsorted(nodes, torch.cat([mutp[:, 1], gv])).to(torch.int32)
        wk2 = torch.cat([torch.zeros(nm, dtype=torch.int32, device=dev), wkey])
        gid2 = torch.cat([torch.arange(nm, dtype=torch.int64, device=dev), gid])
        at2 = torch.cat([torch.zeros(nm, dtype=torch.uint8, device=dev), attr])
        ne = int(wk2.numel())
        nbytes = L.rg_mws_graph_workspace_bytes(n_nodes, ne)
        if nbytes < 0:
            raise RuntimeError("raygun_amd.segment_slabs: second-level graph too large")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        rep = torch.empty(n_nodes, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.rg_mws_graph(C.c_void_p(wk2.data_ptr()), C.c_void_p(gid2.data_ptr()), C.c_void_p(eu.data_ptr()), C.c_void_p(ev.data_ptr()),
                                      C.c_void_p(at2.data_ptr()), ne, n_nodes, C.c_void_p(rep.data_ptr()), C.c_void_p(ws.data_ptr()), nbytes, st()),
                       "rg_mws_graph")
        if n_nodes and int(rep.min().item()) < 0:
            raise RuntimeError("raygun_amd.segment_slabs: the second-level pass made no progress (internal error)")
        rep_g = nodes[rep.long()]
    nonrep = nodes[rep_g != nodes]                       # sorted: clusters that merged into a smaller id
    _stage(f"second level ({n_nodes} nodes, {int(wkey.numel())} cross edges, {int(mutp.shape[0])} mutexes)", dev)
    # ---- 4. relabel: final label = rank of the cluster's representative among all representatives
    outs = []
    for sid, lab in zip(my_ids, labs):
        g = torch.arange(all_counts[sid] + 1, dtype=torch.int64, device=dev) + int(bases[sid])
        r = g.clone()
        if n_nodes:
            pos = torch.searchsorted(nodes, g).clamp_(max=n_nodes - 1)
            hit = nodes[pos] == g
            r = torch.where(hit, rep_g[pos], g)
        lut = r - torch.searchsorted(nonrep, r)
        lut[0] = 0
        with torch.cuda.device(dev):
            _lib.check(L.rg_mws_relabel(C.c_void_p(lab.data_ptr()), C.c_void_p(lut.data_ptr()), lab.numel(), st()), "rg_mws_relabel")
        outs.append(lab)
    out = torch.cat(outs, 0) if len(outs) > 1 else outs[0]
    _stage("relabel + concatenate", dev)
    return out if as_tensor else out.cpu().numpy().astype(np.uint64)


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
    n_vox = int(np.prod(weights.shape[1:]))
    n_slabs = seg_config.get("n_slabs") or -(-(n_vox * weights.shape[0]) // SINGLE_PASS_EDGE_IDS)   # one pass holds n_offsets * voxels < 2^32 edge ids
    if n_slabs > 1:   # beyond one pass (or on request): Z-slabs + one exchange step; the result is defined by that decomposition
        seg = segment_slabs(weights, neighborhood, seg_config["sep"], strides=[10, 10, 10], mask=mask, n_slabs=n_slabs)
    else:
        seg = compute_mws_segmentation(weights, neighborhood, seg_config["sep"], strides=[10, 10, 10], mask=mask)
    if ("save" in seg_config and not seg_config["save"]) or f is None:
        return seg
    dest = seg_config.get("dest_dataset", f"mutex_{'{:.2f}'.format(seg_config['mask_thresh'])}")
    f[dest] = seg
    f[dest].attrs["offset"] = f[seg_config["aff_ds"]].attrs["offset"]
    f[dest].attrs["resolution"] = f[seg_config["aff_ds"]].attrs["resolution"]
    return seg


def segment(config_path=None, affs=None):
    """segment.py:233-296 — `python -m raygun_amd.segment seg.json` = raygun-segment.  "mutex": true -> mutex_segment; otherwise the reference's default
    route: fragments by the seeded watershed on the boundary-distance transform, one waterz-style agglomeration per threshold (raygun_amd.watershed), the
    segmentations written next to the affinities as `pred_seg_<threshold>` with their offset / resolution and listed in the neuroglancer view script.
    `affs` (keyword, or config["affs"] as an array / .npy path) segments an in-memory (3, Z, Y, X) array instead of the container's; returns the list of
    segmentations (uint64 arrays)."""
    if config_path is None:
        import sys
        config_path = sys.argv[1]
    seg_config = {"aff_ds": "pred_affs", "thresholds": [t for t in np.arange(0.1, 0.9, 0.1)], "mutex": False, "max_affinity_value": 1.0, "labels_mask": None}
    seg_config.update(read_config(config_path))
    if seg_config["mutex"]:
        return mutex_segment(config_path, affs=affs)
    from . import watershed
    thresholds = seg_config["thresholds"]
    if affs is None:
        affs = seg_config.get("affs")
    if isinstance(affs, str):
        affs = np.load(affs)
    if affs is not None:   # in memory: nothing to skip, nothing to write unless a container is named as well
        segs = watershed.get_segmentation(np.asarray(affs, dtype=np.float32) if not torch.is_tensor(affs) else affs, thresholds=list(thresholds),
                                          labels_mask=seg_config["labels_mask"], max_affinity_value=seg_config["max_affinity_value"])
        if "file" not in seg_config or ("save" in seg_config and not seg_config["save"]):
            return segs
    file, aff_ds = seg_config["file"], seg_config["aff_ds"]
    done = True
    for thresh in thresholds:
        done = done and os.path.exists(os.path.join(file, "pred_seg_%.2f" % thresh))
    try:
        import zarr
    except ImportError:
        from . import zarrio as zarr  # zarr v2 directory stores with stdlib codecs (see zarrio.py)
    f = zarr.open(file, "a")
    segs = None if affs is None else segs
    if not done:
        if segs is None:
            prediction = np.asarray(f[aff_ds][:]).astype(np.float32)
            segs = watershed.get_segmentation(prediction, thresholds=list(thresholds), labels_mask=seg_config["labels_mask"],
                                              max_affinity_value=seg_config["max_affinity_value"])
        where = os.path.dirname(config_path) if isinstance(config_path, str) else file      # (a dict config has no directory: beside the container)
        view_script = os.path.join(where, f"view_{os.path.basename(file).rstrip('.n5').rstrip('.zarr')}.ng")
        for thresh, seg in zip(thresholds, segs):
            dest_dataset = f'pred_seg_{"{:.2f}".format(thresh)}'
            f[dest_dataset] = seg
            f[dest_dataset].attrs["offset"] = f[aff_ds].attrs["offset"]
            f[dest_dataset].attrs["resolution"] = f[aff_ds].attrs["resolution"]
            if not os.path.exists(view_script):
                with open(view_script, "w") as v:
                    v.write(f"neuroglancer -f {file} -d {dest_dataset} ")
            else:
                with open(view_script, "a") as v:
                    v.write(f"{dest_dataset} ")
    return segs


if __name__ == "__main__":
    segment()

"""ORACLE (test infrastructure) — Python face of oracle/mws.c plus the numpy restatement of the preprocessing in
raygun.segment.mutex_segment (reference: src/raygun/segment.py:139-199).  PARITY UNPINNED (see oracle/mws.c)."""
import ctypes as C

import numpy as np

from . import build_oracle

DEFAULT_NEIGHBORHOOD = [[1, 0, 0], [0, 1, 0], [0, 0, 1], [2, 0, 0], [0, 2, 0], [0, 0, 2],
                        [4, 0, 0], [0, 4, 0], [0, 0, 4], [8, 0, 0], [0, 8, 0], [0, 0, 8]]


def neighborhood(n_diagonals=8, base=None):
    """segment.py:144-157,173-181."""
    nb = np.array(DEFAULT_NEIGHBORHOOD if base is None else base)
    if n_diagonals > 0:
        t = np.linspace(0, np.pi, num=n_diagonals, endpoint=False)
        pos, neg = np.round(n_diagonals * np.sin(t)), np.round(n_diagonals * np.cos(t))
        nb = np.concatenate([nb, np.stack([0 * pos, pos, neg], axis=-1)]).astype(np.int8)
    return nb


def prepare(affs, sep=3, mask_thresh=0.5, max_affinity_value=1.0):
    """segment.py:188-194: mask = mean(affs) > thresh*max ; weights = 1-affs, attractive channels flipped back."""
    mask = np.mean(affs, axis=0) > mask_thresh * max_affinity_value
    w = 1 - affs
    w[:sep] = w[:sep] * -1
    w[:sep] = w[:sep] + 1
    return w.astype(np.float32), mask


def affogato_weights(weights, n_attractive, invert_repulsive_weights=True, bias_cut=0.):
    """What affogato's Python wrapper does to the weights before ranking them (published source, not vendored here:
    `if invert_repulsive_weights: weights[n:] *= -1; weights[n:] += 1` then `weights[:n] += bias_cut`), in float32."""
    w = np.array(weights, dtype=np.float32, copy=True)
    if invert_repulsive_weights:
        w[n_attractive:] *= -1
        w[n_attractive:] += 1
    w[:n_attractive] += np.float32(bias_cut)
    return w


def compute_mws_segmentation(weights, offsets, n_attractive, strides=(1, 1, 1), mask=None, invert_repulsive_weights=True,
                             bias_cut=0., origin=(0, 0, 0)):
    """affogato's call (defaults included: the reference passes neither invert_repulsive_weights nor bias_cut,
    segment.py:197-199) on the C restatement, which ranks the transformed weights as given."""
    lib = C.CDLL(build_oracle.build())
    w = np.ascontiguousarray(affogato_weights(weights, n_attractive, invert_repulsive_weights, bias_cut))
    off = np.ascontiguousarray(offsets, dtype=np.int32)
    shape = (C.c_int32 * 3)(*w.shape[1:])
    st = (C.c_int32 * 3)(*[int(s) for s in strides])
    org = (C.c_int32 * 3)(*[int(s) for s in origin])
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    out = np.zeros(w.shape[1:], dtype=np.uint64)
    rc = lib.mws_oracle2(w.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), C.c_int32(off.shape[0]),
                         C.c_int32(n_attractive), shape, st, org, None if m is None else m.ctypes.data_as(C.c_void_p),
                         out.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise MemoryError("mws oracle")
    return out


def same_partition(a, b):
    """True iff two label volumes describe the same partition (0 = background in both) — the portable parity claim."""
    a, b = np.asarray(a).reshape(-1), np.asarray(b).reshape(-1)
    if a.shape != b.shape or not np.array_equal(a == 0, b == 0):
        return False
    pairs = np.unique(np.stack([a, b], 1), axis=0)
    return len(np.unique(pairs[:, 0])) == len(pairs) == len(np.unique(pairs[:, 1]))


def mws_python(weights, offsets, n_attractive, strides=(1, 1, 1), mask=None, invert_repulsive_weights=True, bias_cut=0.):
    """Independent pure-Python Kruskal (explicit cluster/mutex sets) for tiny volumes: checks the C restatement."""
    weights = affogato_weights(weights, n_attractive, invert_repulsive_weights, bias_cut)
    Z, Y, X = weights.shape[1:]
    N = Z * Y * X
    edges = []
    for k, (oz, oy, ox) in enumerate(np.asarray(offsets).tolist()):
        for z in range(Z):
            for y in range(Y):
                for x in range(X):
                    z2, y2, x2 = z + oz, y + oy, x + ox
                    if not (0 <= z2 < Z and 0 <= y2 < Y and 0 <= x2 < X):
                        continue
                    if k >= n_attractive and (z % strides[0] or y % strides[1] or x % strides[2]):
                        continue
                    if mask is not None and not (mask[z, y, x] and mask[z2, y2, x2]):
                        continue
                    u, v = (z * Y + y) * X + x, (z2 * Y + y2) * X + x2
                    edges.append((-float(weights[k, z, y, x]), k * N + u, k, u, v))
    edges.sort()
    cluster = {i: {i} for i in range(N)}
    of = list(range(N))
    mutex = {i: set() for i in range(N)}
    for _, _, k, u, v in edges:
        a, b = of[u], of[v]
        if a == b:
            continue
        if k < n_attractive:
            if b in mutex[a]:
                continue
            if len(cluster[a]) < len(cluster[b]):
                a, b = b, a
            for i in cluster[b]:
                of[i] = a
            cluster[a] |= cluster.pop(b)
            for c in mutex.pop(b):
                mutex[c].discard(b); mutex[c].add(a); mutex[a].add(c)
        else:
            mutex[a].add(b); mutex[b].add(a)
    labels = np.zeros(N, dtype=np.uint64)
    seen = {}
    for i in range(N):
        if mask is not None and not mask.reshape(-1)[i]:
            continue
        r = of[i]
        if r not in seen:
            seen[r] = len(seen) + 1
        labels[i] = seen[r]
    return labels.reshape(Z, Y, X)


def _order_desc(w):
    """The 32-bit key both passes rank by: ascending key == descending float32 weight (raygun_amd/csrc/mws.hip:order_desc)."""
    b = np.ascontiguousarray(w, dtype=np.float32).view(np.uint32)
    b = np.where(b & np.uint32(0x80000000), ~b, b | np.uint32(0x80000000))
    return ~b


def graph_mws(n_nodes, wkey, gid, eu, ev, attr):
    lib = C.CDLL(build_oracle.build())
    wkey = np.ascontiguousarray(wkey, np.uint32); gid = np.ascontiguousarray(gid, np.uint64)
    eu = np.ascontiguousarray(eu, np.int64); ev = np.ascontiguousarray(ev, np.int64); attr = np.ascontiguousarray(attr, np.uint8)
    rep = np.zeros(n_nodes, np.int64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    if lib.mws_graph_oracle(C.c_int64(n_nodes), C.c_int64(len(wkey)), p(wkey), p(gid), p(eu), p(ev), p(attr), p(rep)) != 0:
        raise MemoryError("graph mws oracle")
    return rep


def slab_mws(weights, offsets, n_attractive, strides, mask, bounds, invert_repulsive_weights=True, bias_cut=0.):
    """CPU restatement of the two-level slab decomposition (the oracle for volumes segmented in Z-slabs, SURVEY.md
    8e): first level = the grid oracle per slab [bounds[s], bounds[s+1]) (stride grid in global coordinates), second level =
    graph_mws over ALL first-level clusters with (a) every valid grid edge whose ends lie in different slabs, ranked like the
    grid pass, and (b) a pre-existing mutex (wkey 0) for every valid repulsive edge inside a slab whose ends carry different
    labels.  No node filtering here — the GPU path's restriction to boundary-zone clusters must not change the result.
    Labels: 1 + rank of the cluster's first voxel in global raster order."""
    w = np.ascontiguousarray(weights, np.float32)
    K, Z, Y, X = w.shape
    off = np.asarray(offsets, np.int64)
    tw = affogato_weights(w, n_attractive, invert_repulsive_weights, bias_cut)
    lab = np.zeros((Z, Y, X), np.int64)
    base = 0
    slab_of = np.zeros(Z, np.int64)
    for s in range(len(bounds) - 1):
        z0, z1 = bounds[s], bounds[s + 1]
        l = compute_mws_segmentation(w[:, z0:z1], offsets, n_attractive, strides, None if mask is None else mask[z0:z1],
                                     invert_repulsive_weights, bias_cut, origin=(z0, 0, 0)).astype(np.int64)
        lab[z0:z1] = np.where(l > 0, l + base, 0)
        base += int(l.max())
        slab_of[z0:z1] = s
    n_nodes = base + 1                                          # node 0 = background, never touched
    zz, yy, xx = np.meshgrid(np.arange(Z), np.arange(Y), np.arange(X), indexing="ij")
    N = Z * Y * X
    wk, gid, eu, ev, at = [], [], [], [], []
    for k in range(K):
        z2, y2, x2 = zz + off[k, 0], yy + off[k, 1], xx + off[k, 2]
        ok = (z2 >= 0) & (z2 < Z) & (y2 >= 0) & (y2 < Y) & (x2 >= 0) & (x2 < X)
        if k >= n_attractive:
            ok &= (zz % strides[0] == 0) & (yy % strides[1] == 0) & (xx % strides[2] == 0)
        u = np.nonzero(ok.reshape(-1))[0]
        v = (z2.reshape(-1)[u] * Y + y2.reshape(-1)[u]) * X + x2.reshape(-1)[u]
        lu, lv = lab.reshape(-1)[u], lab.reshape(-1)[v]
        keep = (lu > 0) & (lv > 0)
        cross = slab_of[u // (Y * X)] != slab_of[v // (Y * X)]
        sel = keep & cross
        wk.append(_order_desc(tw[k].reshape(-1)[u[sel]])); gid.append((k * N + u[sel]).astype(np.uint64))
        eu.append(lu[sel]); ev.append(lv[sel]); at.append(np.full(int(sel.sum()), 1 if k < n_attractive else 0, np.uint8))
        if k >= n_attractive:
            sel = keep & ~cross & (lu != lv)
            wk.append(np.zeros(int(sel.sum()), np.uint32)); gid.append((k * N + u[sel]).astype(np.uint64))
            eu.append(lu[sel]); ev.append(lv[sel]); at.append(np.zeros(int(sel.sum()), np.uint8))
    rep = graph_mws(n_nodes, np.concatenate(wk), np.concatenate(gid), np.concatenate(eu), np.concatenate(ev), np.concatenate(at))
    is_rep = rep == np.arange(n_nodes)
    rank = np.cumsum(is_rep) - 1                               # node 0 is its own rep: rank 0 = background
    return rank[rep][lab].astype(np.uint64)

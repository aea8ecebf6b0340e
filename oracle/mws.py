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

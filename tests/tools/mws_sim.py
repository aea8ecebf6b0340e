"""CPU simulation of the round-based mutex watershed rules (scratch): checks the early-execution rules against the oracle."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import mws as O
from raygun_amd import synth

def canon(labels_root, mask):
    out = np.zeros(labels_root.shape, np.uint64); seen = {}
    for i, r in enumerate(labels_root):
        if mask is not None and not mask.flat[i]: continue
        if r not in seen: seen[r] = len(seen) + 1
        out[i] = seen[r]
    return out

def sim(w, offsets, n_attr, strides, mask, WIN=512, early_rep=True, quiet=True, mutex_exc=True, QMAX=48):
    Z, Y, X = w.shape[1:]; N = Z * Y * X
    edges = []
    for k, (oz, oy, ox) in enumerate(np.asarray(offsets).tolist()):
        for z in range(Z):
            for y in range(Y):
                for x in range(X):
                    z2, y2, x2 = z + oz, y + oy, x + ox
                    if not (0 <= z2 < Z and 0 <= y2 < Y and 0 <= x2 < X): continue
                    if k >= n_attr and (z % strides[0] or y % strides[1] or x % strides[2]): continue
                    u = (z * Y + y) * X + x; v = (z2 * Y + y2) * X + x2
                    if mask is not None and (not mask.flat[u] or not mask.flat[v]): continue
                    edges.append((-float(w[k, z, y, x]), k * N + u, u, v, k < n_attr))
    edges.sort(key=lambda e: (e[0], e[1]))
    parent = list(range(N)); mtx = [[] for _ in range(N)]
    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]; x = parent[x]
        return x
    def has_mutex(a, b):
        if len(mtx[a]) > len(mtx[b]): a, b = b, a
        return any(find(p) == b for p in mtx[a])
    win = []; nxt = 0; rounds = 0
    while win or nxt < len(edges):
        rounds += 1
        while len(win) < WIN and nxt < len(edges):
            e = edges[nxt]; win.append([nxt, e[2], e[3], e[4]]); nxt += 1
        # phase A
        claim = {}
        live = []
        for e in win:
            a, b = find(e[1]), find(e[2])
            if a == b: continue
            if e[3] and mtx[a] and mtx[b] and has_mutex(a, b): continue
            e[1], e[2] = a, b
            claim[a] = min(claim.get(a, 1 << 60), e[0]); claim[b] = min(claim.get(b, 1 << 60), e[0])
            live.append(e)
        # phase B: decisions on the phase-A state
        snapshot_find = {}
        def quiet_ok(c, other, slot):
            if not mtx[c]: return True
            if not quiet or len(mtx[c]) > QMAX: return False
            for p in mtx[c]:
                rp = find(p)
                if rp == other: return False
                if claim.get(rp, 1 << 60) < slot:
                    if not mutex_exc or not mtx[other] or not has_mutex(other, rp): return False
            return True
        merges = []; reps = []; keep = []
        for e in live:
            slot, a, b, attr = e
            fa, fb = claim[a] == slot, claim[b] == slot
            if attr:
                child = -1
                if fa and fb: child = b if len(mtx[b]) <= len(mtx[a]) else a
                elif fb and quiet_ok(b, a, slot): child = b
                elif fa and quiet_ok(a, b, slot): child = a
                if child >= 0: merges.append((child, a if child == b else b))
                else: keep.append(e)
            else:
                if (fa and fb) or (early_rep and (fa or fb)): reps.append((a, b))
                else: keep.append(e)
        assert len(keep) < len(win), "stalled"
        for c, p in merges: parent[c] = p
        for c, p in merges:
            if mtx[c]:
                r = find(c); mtx[r].extend(mtx[c]); mtx[c] = []
        for a, b in reps:
            mtx[find(a)].append(b); mtx[find(b)].append(a)
        win = keep
    roots = np.array([find(i) for i in range(N)])
    return canon(roots, mask), rounds

if __name__ == "__main__":
    NB8 = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [2, 0, 0], [0, 2, 0], [0, 0, 2], [0, 1, 1], [0, 1, -1]], dtype=np.int32)
    shape = (10, 10, 10); strides = (1, 1, 1)
    for seed in range(3):
        w = synth.uniform((8,) + shape, 11 + seed, 0, 1)
        ref = O.compute_mws_segmentation(w, NB8, 3, strides, None, invert_repulsive_weights=False).reshape(-1)
        for kw in (dict(early_rep=False, quiet=False), dict(early_rep=True, quiet=False), dict(early_rep=False, quiet=True, mutex_exc=False),
                   dict(early_rep=False, quiet=True, mutex_exc=True), dict()):
            got, rounds = sim(w, NB8, 3, strides, None, **kw)
            print(seed, kw, "rounds", rounds, "diff", int((got != ref).sum()), flush=True)
    shape = (16, 16, 16)
    w = synth.uniform((8,) + shape, 11, 0, 1)
    ref = O.compute_mws_segmentation(w, NB8, 3, strides, None, invert_repulsive_weights=False).reshape(-1)
    for WIN in (65536, 4096):
        got, rounds = sim(w, NB8, 3, strides, None, WIN=WIN)
        print("16^3 WIN", WIN, "rounds", rounds, "diff", int((got != ref).sum()), flush=True)

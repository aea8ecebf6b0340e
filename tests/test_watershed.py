"""The default raygun-segment route (reference: src/raygun/segment.py:15-136, 233-296): seeded watershed on the boundary-distance transform + waterz-style
agglomeration.

CPU (`-m "not gpu"`): the oracle's pieces against what IS installed — scipy pins the distance transform / maximum filter / labelling; the two watershed
restatements (skimage's heap flood, the build-defined spanning-forest order) are checked against first principles on small cases and against each other.
GPU (`-m gpu`): every rg_ws_* pass bit for bit against scipy / the oracle, the whole chain, and `segment(config)` through a container."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import watershed as OW
from raygun_amd import synth


def _volume(shape, seed, touching=True):
    """(3, Z, Y, X) float32 nearest-neighbour affinities with object structure: high inside smooth cells, low where |f| is small (the boundaries)."""
    z, y, x = np.meshgrid(*[np.arange(s, dtype=np.float32) for s in shape], indexing="ij")
    f = np.sin(z / 5.1 + seed) * np.sin(y / 6.3) * np.sin(x / 4.7 + 1.0)
    strength = np.clip(np.abs(f) * (3.0 if touching else 1.5), 0, 1)
    noise = synth.uniform((3,) + tuple(shape), 40 + seed, 0, 1)
    return (np.stack([np.minimum(strength, np.roll(strength, 1, ax)) for ax in range(3)]) * 0.85 + 0.15 * noise).astype(np.float32)


# ------------------------------------------------------------------------------------------------------------------------------ CPU
def test_oracle_edt_is_scipys_and_exact():
    bm = OW.boundary_mask(_volume((20, 24, 28), 1))
    d2, d = OW.squared_edt(bm)
    bg = np.argwhere(~bm)
    for p in np.argwhere(bm)[::97]:                                   # brute force on a sample
        assert d2[tuple(p)] == ((bg - p) ** 2).sum(1).min()
    assert (d2[~bm] == 0).all() and d2.dtype == np.int32


def test_oracle_msf_watershed_equals_prim_from_a_virtual_root():
    """ws_msf (Kruskal with markers) == Prim grown from all seeds at once under the same strict edge order — the minimum spanning forest is unique."""
    import heapq
    rng = np.random.default_rng(5)
    shape = (6, 7, 8)
    d2 = rng.integers(0, 6, size=shape).astype(np.int32)              # few levels: ties everywhere, the order must carry the result
    mask = rng.random(shape) > 0.15
    markers = np.zeros(shape, np.int32)
    for k, p in enumerate(rng.choice(np.flatnonzero(mask), 5, replace=False)):
        markers.flat[p] = k + 1
    got = OW.msf_watershed(d2, markers, mask)
    N, (Z, Y, X) = d2.size, shape
    step = (Y * X, X, 1)
    cap = (1 << 30) - 1

    def edges(i):
        z, y, x = np.unravel_index(i, shape)
        for a, (c, n) in enumerate(((z, Z), (y, Y), (x, X))):
            for s in (-1, 1):
                if 0 <= c + s < n:
                    j = i + s * step[a]
                    if mask.flat[j]:
                        lo, hi = sorted((int(d2.flat[i]), int(d2.flat[j])))
                        yield (cap - lo, cap - hi, 3 * min(i, j) + a), j
    lab = (markers * mask).ravel().copy()
    heap = [(k, j, int(lab[i])) for i in np.flatnonzero(lab) for k, j in edges(int(i))]
    heapq.heapify(heap)
    while heap:
        k, j, l = heapq.heappop(heap)
        if lab[j]:
            continue
        lab[j] = l
        for k2, j2 in edges(int(j)):
            if not lab[j2]:
                heapq.heappush(heap, (k2, j2, l))
    assert np.array_equal(got.ravel(), lab)


def test_oracle_skimage_flood_restatement_first_principles():
    """1-D profile: the flood pops the smallest value of the frontier; a voxel is labelled when its first neighbour pops."""
    image = np.array([[[0, 5, 1, 6, 2, 7, 0]]], dtype=np.float64)
    markers = np.array([[[1, 0, 0, 0, 0, 0, 2]]], dtype=np.int32)
    out = OW.skimage_watershed(image, markers, np.ones_like(markers, bool))
    assert out.ravel().tolist() == [1, 1, 1, 1, 1, 2, 2]
    out = OW.skimage_watershed(image, markers, np.array([[[1, 1, 1, 0, 1, 1, 1]]], bool))      # the mask cuts the line: voxel 4 is reached from the right only
    assert out.ravel().tolist() == [1, 1, 1, 0, 2, 2, 2]


def test_oracle_restatements_agree_up_to_the_meeting_surfaces():
    """The build-defined order and the restated skimage flood are the same minimax-path assignment: they may differ only where floods meet at equal level."""
    affs = _volume((40, 44, 48), 2)
    bm = OW.boundary_mask(affs)
    d2, d = OW.squared_edt(bm)
    seeds, n = OW.seeds_from_distance(d, 6)
    a = OW.msf_watershed(d2, seeds, bm)
    b = OW.skimage_watershed(d.max() - d, seeds, bm)
    assert n > 20 and ((a > 0) == (b > 0)).all() and set(np.unique(a)) == set(np.unique(b))
    differ = (a != b).sum() / bm.sum()
    assert differ < 0.15, differ
    # a differing voxel always has a neighbour of the other label in BOTH results' union: it lies on a meeting surface
    seedvox = (seeds * bm) > 0
    assert (a[seedvox] == b[seedvox]).all()


def test_oracle_agglomeration_merges_in_score_order():
    """Three fragments in a row: 1|2 share a strong boundary (affinity 0.9 -> score 1 - 229/255), 2|3 a weak one (0.2 -> 1 - 51/255): thresholds pick them apart."""
    frag = np.zeros((1, 4, 12), np.uint64)
    frag[..., :4], frag[..., 4:8], frag[..., 8:] = 1, 2, 3
    affs = np.full((3, 1, 4, 12), 0.95, np.float32)
    affs[2, :, :, 4] = 0.9
    affs[2, :, :, 8] = 0.2
    edges, hist = OW.region_graph(affs, frag)
    assert edges == [(1, 2), (2, 3)] and hist[0][229] == 4 and hist[1][51] == 4 and hist.sum() == 8
    segs = list(OW.agglomerate(affs, frag, [0.05, 0.5, 0.9]))
    assert [len(np.unique(s)) for s in segs] == [3, 2, 1]
    assert np.array_equal(np.unique(segs[1]), [1, 3]) and (segs[2] == 1).all()       # the smaller id survives


def test_oracle_agglomeration_histograms_add_up_when_regions_merge():
    """2 merges into 1; both touched 3: the two boundary histograms are summed and the 75 % quantile of the SUM decides the next merge."""
    frag = np.zeros((1, 2, 9), np.uint64)
    frag[0, 0, :3], frag[0, 0, 3:6], frag[0, 0, 6:] = 1, 2, 3
    frag[0, 1, :] = 3
    affs = np.full((3, 1, 2, 9), 0.99, np.float32)
    affs[2, 0, 0, 3] = 0.95            # 1|2: merges first
    affs[1, 0, 1, 0:3] = 0.30          # 1|3 along y: three weak samples
    affs[1, 0, 1, 3:6] = 0.80          # 2|3 along y: three strong samples
    affs[2, 0, 0, 6] = 0.80            # 2|3 along x
    segs = list(OW.agglomerate(affs, frag, [0.1, 0.3, 0.75]))
    # after 1+2 the edge (1,3) holds {0.30 x3, 0.80 x4}: its 75 % quantile is 0.80, score 0.2 — but, as in waterz, the edge keeps its OLD queue position (score
    # 0.70 of {0.30 x3}) until it reaches the top, where it is re-scored and pushed back ("new edges can only be more expensive" is waterz's own assumption):
    # nothing merges at 0.3, everything at 0.75
    assert [len(np.unique(s)) for s in segs] == [2, 2, 1]


# ------------------------------------------------------------------------------------------------------------------------------ GPU
gpu = pytest.mark.gpu


@gpu
@pytest.mark.parametrize("shape,seed", [((40, 44, 48), 2), ((1, 70, 90), 3), ((33, 17, 129), 4), ((70, 9, 10), 5)])
def test_gpu_mask_edt_maxima_labels_equal_scipy(shape, seed):
    from scipy.ndimage import distance_transform_edt, label, maximum_filter
    from raygun_amd import watershed as W
    affs = _volume(shape, seed)
    bm = W.boundary_mask(affs)
    ref_bm = np.mean(affs, axis=0) > 0.5
    assert np.array_equal(bm.cpu().numpy().astype(bool), ref_bm)
    d2 = W.squared_distance_transform(bm)
    d = distance_transform_edt(ref_bm)
    assert np.array_equal(np.sqrt(d2.cpu().numpy().astype(np.float64)), d)            # the distance transform, exactly
    for size in (10, 3, 7):
        seeds, n = W.seeds_from_distance(d2, size)
        want, nw = label(maximum_filter(d, size) == d)
        assert n == nw and np.array_equal(seeds.cpu().numpy(), want), size


@gpu
def test_gpu_edt_far_from_any_boundary():
    """Distances up to the volume's extent: a single background voxel in a corner, one in the middle of a face."""
    from scipy.ndimage import distance_transform_edt
    from raygun_amd import watershed as W
    m = np.ones((50, 60, 140), bool)
    m[0, 0, 0] = m[49, 30, 139] = False
    d2 = W.squared_distance_transform(m).cpu().numpy()
    assert np.array_equal(np.sqrt(d2.astype(np.float64)), distance_transform_edt(m))
    with pytest.raises(ValueError):
        W.squared_distance_transform(np.ones((4, 4, 4), bool))


@gpu
@pytest.mark.parametrize("shape,seed,msd", [((40, 44, 48), 2, 6), ((64, 64, 64), 3, 10), ((1, 90, 120), 4, 5), ((24, 80, 72), 6, 4)])
def test_gpu_seeded_watershed_bit_exact_vs_the_build_defined_oracle(shape, seed, msd):
    from raygun_amd import watershed as W
    affs = _volume(shape, seed)
    frag, n, seeds = W.watershed_from_affinities(affs, min_seed_distance=msd, return_seeds=True)
    want, nw, wseeds = OW.watershed_from_affinities(affs, min_seed_distance=msd, return_seeds=True)
    assert n == nw and frag.dtype == np.uint64 and np.array_equal(seeds, wseeds)
    assert np.array_equal(frag, want), f"{int((frag != want).sum())} of {want.size} fragment labels differ"
    assert len(np.unique(frag)) > 10
    # against the restated skimage flood: same set of labels, same support, a thin set of different voxels
    sk = OW.watershed_from_affinities(affs, min_seed_distance=msd, flood="skimage")[0]
    assert ((sk > 0) == (frag > 0)).all() and (sk != frag).sum() < 0.15 * (frag > 0).sum()


@gpu
def test_gpu_watershed_from_boundary_distance_signature_and_xy_mode():
    from scipy.ndimage import distance_transform_edt
    from raygun_amd import watershed as W
    affs = _volume((6, 60, 64), 7)
    bm = np.mean(affs, axis=0) > 0.5
    d = distance_transform_edt(bm)
    ret = W.watershed_from_boundary_distance(d, bm, return_seeds=True, id_offset=100, min_seed_distance=5)          # the reference's call (float distances)
    d2, _ = OW.squared_edt(bm)
    want = OW.watershed_from_boundary_distance(d, bm, d2, return_seeds=True, id_offset=100, min_seed_distance=5)
    assert ret[1] == want[1] and np.array_equal(ret[0], want[0]) and np.array_equal(ret[2], want[2]) and ret[0][ret[0] > 0].min() > 100
    z = W.watershed_from_boundary_distance(np.zeros((2, 5, 5)), np.zeros((2, 5, 5), bool))      # one all-background component is a seed outside the mask
    assert z[0].sum() == 0
    got = W.watershed_from_affinities(affs, fragments_in_xy=True, min_seed_distance=5, return_seeds=True)
    ref = OW.watershed_from_affinities(affs, fragments_in_xy=True, min_seed_distance=5, return_seeds=True)
    assert got[1] == ref[1] and np.array_equal(got[0], ref[0]) and np.array_equal(got[2], ref[2])


@gpu
@pytest.mark.parametrize("shape,seed", [((40, 44, 48), 2), ((56, 60, 64), 8)])
def test_gpu_agglomeration_equals_the_oracle(shape, seed):
    from raygun_amd import watershed as W
    affs = _volume(shape, seed)
    frag = OW.watershed_from_affinities(affs, min_seed_distance=3)[0]      # over-segmented: cells split into several touching fragments
    thresholds = [0.05, 0.1, 0.15, 0.2, 0.3, 0.5]
    st = {}
    got = list(W.agglomerate(affs, frag, thresholds, stats=st))
    want = list(OW.agglomerate(affs, frag, thresholds))
    counts = [len(np.unique(s)) for s in want]
    assert counts[0] > counts[-1] > 0 and st["records"] > 100, counts        # merges actually happen along the way
    for t, g, w in zip(thresholds, got, want):
        assert g.dtype == np.uint64 and np.array_equal(g, w), (t, int((g != w).sum()))
    with pytest.raises(ValueError):
        list(W.agglomerate(np.zeros((4,) + shape, np.float32), frag, [0.5]))


@gpu
def test_gpu_get_segmentation_and_segment_config_through_a_container(tmp_path):
    from raygun_amd import segment as S, watershed as W, zarrio
    shape = (36, 40, 44)
    affs = _volume(shape, 9)
    thresholds = [0.05, 0.2, 0.5]
    segs = W.get_segmentation(affs, thresholds, max_affinity_value=1.0)
    want = OW.get_segmentation(affs, thresholds, max_affinity_value=1.0)
    assert len(segs) == 3 and all(np.array_equal(a, b) for a, b in zip(segs, want))
    path = str(tmp_path / "vol.zarr")
    a = zarrio.prepare_ds(path, "pred_affs", shape, (4, 4, 4), np.float32, write_size=(18, 20, 22), num_channels=3, offset=(40, 0, 8), compressor=None)
    a[...] = affs
    cfg = tmp_path / "seg.json"
    cfg.write_text(json.dumps({"file": path, "thresholds": thresholds}))
    out = S.segment(str(cfg))
    assert all(np.array_equal(a_, b_) for a_, b_ in zip(out, want))
    for t, w in zip(thresholds, want):
        ds = zarrio.open_ds(path, "pred_seg_%.2f" % t)
        assert np.array_equal(ds[...], w) and list(ds.attrs["offset"]) == [40, 0, 8] and list(ds.attrs["resolution"]) == [4, 4, 4]
    view = (tmp_path / "view_vol.ng").read_text()
    assert view.startswith(f"neuroglancer -f {path} -d pred_seg_0.05 ") and view.rstrip().endswith("pred_seg_0.50")
    assert S.segment(str(cfg)) is None                    # everything exists: the reference skips the work (and returns nothing)

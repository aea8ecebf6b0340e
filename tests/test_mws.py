"""Mutex watershed: oracle self-consistency (CPU) and GPU-vs-oracle bit-exactness (labels are canonical: 1 + rank of the
cluster's first voxel in raster order, 0 = masked out)."""
import numpy as np
import pytest

from oracle import mws as O
from raygun_amd import synth

NB8 = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [2, 0, 0], [0, 2, 0], [0, 0, 2], [0, 1, 1], [0, 1, -1]], dtype=np.int32)


def blobs(shape, seed):
    """Smooth structured affinities: high inside blobs, low on boundaries (plus hash noise)."""
    z, y, x = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
    f = np.sin(z / 3.1 + seed) * np.sin(y / 4.3) * np.sin(x / 3.7 + 1.0)
    base = (np.abs(f) > 0.25).astype(np.float32) * 0.8 + 0.1
    return base


def test_c_oracle_equals_python_oracle():
    for shape in [(3, 4, 5), (6, 6, 6), (5, 7, 4)]:
        w = synth.uniform((8,) + shape, 3, 0, 1)
        mask = synth.uniform(shape, 4, 0, 1) > 0.15
        for st in [(1, 1, 1), (2, 2, 2)]:
            for m in (mask, None):
                assert (O.compute_mws_segmentation(w, NB8, 3, st, m) == O.mws_python(w, NB8, 3, st, m)).all()


def test_oracle_edge_cases():
    w = synth.uniform((8, 4, 4, 4), 5, 0, 1)
    allmasked = np.zeros((4, 4, 4), bool)
    assert O.compute_mws_segmentation(w, NB8, 3, (1, 1, 1), allmasked).max() == 0
    # strong attraction, no repulsion in range -> one segment
    w1 = np.ones((8, 4, 4, 4), np.float32); w1[3:] = 0.0
    seg = O.compute_mws_segmentation(w1, NB8, 3, (1, 1, 1), None)
    assert seg.min() == 1 and seg.max() == 1
    # repulsion stronger than attraction everywhere -> every voxel its own segment, numbered in raster order
    w2 = np.full((8, 3, 3, 3), 0.1, np.float32); w2[3:] = 0.9
    nb = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 0, 0], [0, 1, 0], [0, 0, 1], [0, 1, 1], [1, 1, 0]], dtype=np.int32)
    seg = O.compute_mws_segmentation(w2, nb, 3, (1, 1, 1), None)
    assert (seg.reshape(-1) == np.arange(1, 28)).all()
    nb20 = O.neighborhood(8)
    assert nb20.shape == (20, 3) and nb20[12:].tolist() == [[0, 0, 8], [0, 3, 7], [0, 6, 6], [0, 7, 3], [0, 8, 0], [0, 7, -3], [0, 6, -6], [0, 3, -7]]


@pytest.mark.gpu
@pytest.mark.parametrize("shape,strides,use_mask,structured", [
    ((6, 6, 6), (1, 1, 1), True, False), ((9, 10, 11), (2, 2, 2), True, False), ((16, 16, 16), (1, 1, 1), False, False),
    ((24, 20, 28), (3, 2, 4), True, True), ((32, 32, 32), (10, 10, 10), True, True), ((1, 40, 40), (1, 4, 4), False, True),
    # more edges than one window of the round-based pass holds; dense mutexes (stride 1) exercise the early-merge rules
    ((48, 48, 48), (1, 1, 1), False, False), ((40, 48, 56), (2, 2, 2), True, True), ((64, 64, 64), (4, 4, 4), True, False),
    ((56, 56, 56), (1, 1, 1), True, True), ((72, 72, 72), (10, 10, 10), False, True)])
def test_gpu_labels_bit_exact(shape, strides, use_mask, structured):
    """Bit-exact against the build's own CPU restatement (oracle/mws.c — PARITY UNPINNED against affogato itself, whose label
    values are union-find internals; the partition is the portable claim and is asserted separately), for both settings of
    affogato's invert_repulsive_weights (default True = what segment.py:197-199 gets) and a non-zero bias_cut."""
    from raygun_amd.segment import compute_mws_segmentation
    w = synth.uniform((8,) + shape, 11, 0, 1)
    if structured:
        w = (0.7 * blobs(shape, 2)[None] + 0.3 * w).astype(np.float32)  # raw affinities: high = same object, every channel
    mask = (synth.uniform(shape, 12, 0, 1) > 0.2) if use_mask else None
    keep = w.copy()
    for inv, bias in ((True, 0.0), (False, 0.0), (True, 0.125)):
        ww = w.copy()
        if structured and not inv:
            ww[3:] = 1 - ww[3:]      # the caller inverts instead
        ref = O.compute_mws_segmentation(ww, NB8, 3, strides, mask, invert_repulsive_weights=inv, bias_cut=bias)
        st = {}
        got = compute_mws_segmentation(ww, NB8, 3, strides, mask=mask, invert_repulsive_weights=inv, bias_cut=bias, stats=st)
        assert got.dtype == np.uint64 and got.shape == ref.shape
        # rg_mws_stats: every edge enters the window at least once, in at least one round
        assert st["rounds"] >= 1 and st["window_visits"] >= st["edges"] > 0 and len(st["phase_ms"]) == 3
        assert (got == ref).all(), f"invert={inv} bias={bias}: {int((got != ref).sum())} of {ref.size} labels differ"
        assert O.same_partition(got, ref)
    assert np.array_equal(w, keep)  # the caller's weights are not modified (affogato modifies them in place)


def test_affogato_weight_transform_and_partition_helper():
    """oracle.affogato_weights = the float32 arithmetic of affogato's wrapper; the reference pipeline (segment.py:188-199)
    therefore ranks repulsive edges by fl(1 - fl(1 - a)), attractive ones by fl(1 - fl(1 - a)) too (segment.py:193-194)."""
    a = synth.uniform((5, 3, 3, 3), 1, 0, 1).astype(np.float32)
    w = O.affogato_weights(a, 2)
    assert np.array_equal(w[:2], a[:2]) and np.array_equal(w[2:], (a[2:] * np.float32(-1)) + np.float32(1))
    assert np.array_equal(O.affogato_weights(a, 2, False, 0.25)[:2], a[:2] + np.float32(0.25))
    ww, _ = O.prepare(a.copy(), 2)
    both = O.affogato_weights(ww, 2)
    assert np.array_equal(both[2:], np.float32(1) - (np.float32(1) - a[2:]))
    l1 = np.array([[1, 1, 2], [0, 3, 3]]); l2 = np.array([[7, 7, 4], [0, 9, 9]]); l3 = np.array([[7, 7, 7], [0, 9, 9]])
    assert O.same_partition(l1, l2) and not O.same_partition(l1, l3) and not O.same_partition(l1, np.where(l1 == 0, 5, l1))
    tiny = synth.uniform((8, 3, 4, 3), 2, 0, 1)
    for inv in (True, False):
        assert (O.compute_mws_segmentation(tiny, NB8, 3, (1, 1, 1), None, inv, 0.1) == O.mws_python(tiny, NB8, 3, (1, 1, 1), None, inv, 0.1)).all()


@pytest.mark.gpu
def test_gpu_mutex_segment_reference_pipeline():
    """segment.py:139-199 end to end on a synthetic 20-channel affinity volume (12 axis + 8 diagonal offsets, sep 3,
    strides 10, mask from the channel mean) against the oracle's restatement of the same preprocessing."""
    from raygun_amd.segment import build_neighborhood, mutex_segment
    shape = (24, 40, 40)
    affs = (0.75 * blobs(shape, 1)[None] + 0.25 * synth.uniform((20,) + shape, 21, 0, 1)).astype(np.float32)
    seg = mutex_segment({"mutex": True, "save": False}, affs=affs.copy())
    ww, m = O.prepare(affs.copy())
    ref = O.compute_mws_segmentation(ww, O.neighborhood(8).astype(np.int32), 3, (10, 10, 10), m)
    assert (build_neighborhood() == O.neighborhood(8)).all()
    assert (seg == ref).all() and seg.max() > 1 and (seg[~m] == 0).all()


@pytest.mark.gpu
def test_gpu_prepare_bit_exact():
    """rg_mws_prepare == the numpy float32 arithmetic of segment.py:188-194 (oracle.mws.prepare), bit for bit, including
    voxels whose channel mean sits right at the threshold."""
    from raygun_amd.segment import prepare_weights
    shape = (9, 14, 21)
    affs = synth.uniform((20,) + shape, 31, 0, 1).astype(np.float32)
    affs[:, 0] = 0.5          # mean exactly at the threshold: not > 0.5
    affs[:, 1, 0] = np.float32(0.5) + np.float32(2.0 ** -24)
    for sep, thresh, mx in ((3, 0.5, 1.0), (5, 0.25, 2.0)):
        ww, m = O.prepare(affs.copy(), sep, thresh, mx)
        w, mk = prepare_weights(affs.copy(), sep, thresh, mx)
        assert (w.cpu().numpy().view(np.uint32) == ww.view(np.uint32)).all()
        assert (mk.cpu().numpy().astype(bool) == m).all()


@pytest.mark.gpu
def test_gpu_prepare_from_uint8_bit_exact():
    """rg_mws_prepare_u8 (the block runner's uint8 affinities, scaled inside the kernel) == the numpy float32 arithmetic of segment.py:188-194 on
    `affs8.astype(float32) * float32(1/255)` — oracle.mws.prepare — bit for bit, and == the float kernel on the same float array."""
    import torch
    from raygun_amd.segment import prepare_weights, prepare_weights_u8
    shape = (7, 13, 19)
    rng = np.random.default_rng(8)
    a8 = rng.integers(0, 256, size=(20,) + shape, dtype=np.uint8)
    a8[:, 0, 0, :4] = np.array([127, 128, 0, 255], np.uint8)     # channel means right around the threshold, and the range ends
    scale = np.float32(1.0 / 255.0)
    af = a8.astype(np.float32) * scale
    for sep, thresh in ((3, 0.5), (5, 0.25)):
        ww, m = O.prepare(af.copy(), sep, thresh, 1.0)
        w, mk = prepare_weights_u8(torch.from_numpy(a8).cuda(), float(scale), sep, thresh, 1.0)
        assert (w.cpu().numpy().view(np.uint32) == ww.view(np.uint32)).all()
        assert (mk.cpu().numpy().astype(bool) == m).all()
        w2, mk2 = prepare_weights(af.copy(), sep, thresh, 1.0)
        assert torch.equal(w, w2) and torch.equal(mk, mk2)
    with pytest.raises(RuntimeError):
        prepare_weights_u8(torch.from_numpy(a8), float(scale), 3, 0.5, 1.0)       # host tensor: no CPU fallback


@pytest.mark.gpu
def test_gpu_segment_from_uint8_affinities_equals_the_float_route():
    """U8Affinities (rg_mws_segment_u8 / rg_mws_cross_edges_u8: the preprocessing of segment.py:188-194 applied inside the edge kernels, slabs addressed
    in place) gives the labels of the float route — rg_mws_prepare_u8's weights through rg_mws_segment — bit for bit: one pass, forced Z-slabs,
    and against the CPU oracle on the oracle's own prepare() of the same values.  uint8 values tie constantly: the tie order is exercised too."""
    import torch
    from raygun_amd.segment import (U8Affinities, build_neighborhood, compute_mws_segmentation, prepare_weights_u8, segment_slabs)
    nb = build_neighborhood(None, 8)
    shape = (30, 26, 34)
    rng = np.random.default_rng(12)
    zz, yy, xx = np.meshgrid(*[np.arange(n) for n in shape], indexing="ij")
    blobs = (np.sin(zz / 3.1) * np.sin(yy / 2.7 + 1) * np.sin(xx / 3.3) > -0.2)
    a8 = np.stack([np.where(blobs, 200, 30) + rng.integers(-25, 26, size=shape) for _ in range(len(nb))]).clip(0, 255).astype(np.uint8)
    t8 = torch.from_numpy(a8).cuda()
    scale = np.float32(1.0 / 255.0)
    w, m = prepare_weights_u8(t8, float(scale), 3, 0.5, 1.0)
    _, m2 = prepare_weights_u8(t8, float(scale), 3, 0.5, 1.0, mask_only=True)
    assert torch.equal(m, m2)
    src = U8Affinities(t8, float(scale), 3)
    ref = compute_mws_segmentation(w, nb, 3, [3, 3, 3], mask=m)
    got = compute_mws_segmentation(src, nb, 3, [3, 3, 3], mask=m)
    assert got.dtype == np.uint64 and (got == ref).all()
    ww, mm = O.prepare(a8.astype(np.float32) * scale, 3, 0.5, 1.0)
    assert (got == O.compute_mws_segmentation(ww, nb, 3, [3, 3, 3], mm)).all()
    ref_s = segment_slabs(w, nb, 3, strides=[3, 3, 3], mask=m, n_slabs=3)
    got_s = segment_slabs(src, nb, 3, strides=[3, 3, 3], mask=m, n_slabs=3)
    assert (got_s == ref_s).all() and got_s.max() > 3
    with pytest.raises(RuntimeError):
        U8Affinities(torch.from_numpy(a8), float(scale), 3)          # host tensor


@pytest.mark.gpu
def test_gpu_ties_and_empty():
    from raygun_amd.segment import compute_mws_segmentation
    # every weight equal: the tie order (ascending edge id) decides everything
    w = np.full((8, 5, 6, 7), 0.5, np.float32)
    assert (compute_mws_segmentation(w, NB8, 3, (1, 1, 1), mask=None) == O.compute_mws_segmentation(w, NB8, 3, (1, 1, 1), None)).all()
    m0 = np.zeros((5, 6, 7), bool)
    assert compute_mws_segmentation(w, NB8, 3, (1, 1, 1), mask=m0).max() == 0
    with pytest.raises(NotImplementedError):
        compute_mws_segmentation(w, NB8, 3, algorithm="prim")
    with pytest.raises(ValueError):
        compute_mws_segmentation(w[:5], NB8, 3)


def test_slab_oracle_reduces_to_single_pass():
    """oracle.slab_mws (the CPU restatement of the two-level slab decomposition) with ONE slab is the plain
    mutex watershed; with several slabs it is a different, decomposition-defined partition."""
    w = synth.uniform((8, 12, 10, 9), 3, 0, 1)
    m = synth.uniform((12, 10, 9), 4, 0, 1) > 0.15
    one = O.compute_mws_segmentation(w, NB8, 3, (2, 2, 2), m)
    assert np.array_equal(O.slab_mws(w, NB8, 3, (2, 2, 2), m, [0, 12]), one)
    two = O.slab_mws(w, NB8, 3, (2, 2, 2), m, [0, 6, 12])
    assert np.array_equal(two == 0, one == 0) and two.max() > 1
    # the graph oracle on a hand-made case: a-b attract strongly, b-c attract weakly but a|c repel in between
    rep = O.graph_mws(3, [10, 30, 20], [0, 1, 2], [0, 1, 0], [1, 2, 2], [1, 1, 0])
    assert rep.tolist() == [0, 0, 2]


@pytest.mark.gpu
@pytest.mark.parametrize("shape,strides,bounds,use_mask", [
    ((48, 40, 44), (2, 2, 2), [0, 24, 48], True), ((48, 40, 44), (2, 2, 2), [0, 10, 31, 48], False),
    ((40, 36, 36), (1, 1, 1), [0, 20, 40], True), ((64, 48, 48), (4, 4, 4), [0, 16, 32, 48, 64], True)])
def test_gpu_slab_decomposition_bit_exact(shape, strides, bounds, use_mask):
    """segment_slabs (first level per slab, boundary-zone exchange, second-level mutex watershed over the clusters that reach
    a zone, global relabel — all on the GPU) against oracle.slab_mws, which runs the second level over ALL clusters and ALL
    intra-slab mutexes: bit-exact, so the GPU path's node filter is exact too."""
    from raygun_amd.segment import compute_mws_segmentation, segment_slabs
    w = (0.7 * blobs(shape, 2)[None] + 0.3 * synth.uniform((8,) + shape, 11, 0, 1)).astype(np.float32)
    mask = (synth.uniform(shape, 12, 0, 1) > 0.2) if use_mask else None
    ref = O.slab_mws(w, NB8, 3, strides, mask, bounds)
    got = segment_slabs(w, NB8, 3, strides, mask=mask, bounds=bounds)
    assert got.dtype == np.uint64 and (got == ref).all(), f"{int((got != ref).sum())} of {ref.size} labels differ"
    # one slab == the single pass
    assert (segment_slabs(w, NB8, 3, strides, mask=mask, n_slabs=1) == compute_mws_segmentation(w, NB8, 3, strides, mask=mask)).all()


@pytest.mark.gpu
def test_gpu_mutex_segment_in_slabs_reference_pipeline():
    """mutex_segment with "n_slabs": 2 (what volumes beyond the single-pass limit get automatically): the reference's 20-offset /
    sep 3 / stride 10 / masked pipeline on a 96^3 volume in 2 slabs against the oracle with the identical decomposition."""
    from raygun_amd.segment import mutex_segment, slab_bounds
    shape = (96, 96, 96)
    affs = (0.75 * blobs(shape, 1)[None] + 0.25 * synth.uniform((20,) + shape, 21, 0, 1)).astype(np.float32)
    seg = mutex_segment({"mutex": True, "save": False, "n_slabs": 2}, affs=affs.copy())
    ww, m = O.prepare(affs.copy())
    assert slab_bounds(96, 2, 8) == [0, 48, 96]
    ref = O.slab_mws(ww, O.neighborhood(8).astype(np.int32), 3, (10, 10, 10), m, [0, 48, 96])
    assert (seg == ref).all() and seg.max() > 1 and (seg[~m] == 0).all()

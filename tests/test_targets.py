"""Train-time target generation (reference call sites src/raygun/io/MTLSDDataPipe.py:99-131): the numpy / scipy
restatement of gunpowder's AddAffinities / BalanceLabels and lsd's AddLocalShapeDescriptor (oracle/targets.py, PARITY UNPINNED:
the packages are not installed here) pinned by hand-derived values on the CPU, and the HIP kernels (raygun_amd.targets) against
that oracle on the GPU: affinities, masks and balance weights bit for bit, LSD offsets / mass within 3e-6 and variances / correlations within 1e-4 absolute
(float32 window sums against scipy's separable float32 passes)."""
import numpy as np
import pytest
import torch

from oracle import targets as O
from raygun_amd import synth


def _labels(shape, seed, n_seg=7, bg=0.2):
    """Blocky random segments: nearest-seed labelling of a few random points, some voxels background."""
    rng = np.random.default_rng(seed)
    pts = rng.uniform(0, 1, (n_seg, 3)) * np.array(shape)
    g = np.stack(np.meshgrid(*[np.arange(s) for s in shape], indexing="ij"), -1).astype(np.float32)
    lab = (np.linalg.norm(g[..., None, :] - pts, axis=-1).argmin(-1) + 1).astype(np.int64) * 1000 + 17      # sparse ids
    lab[synth.uniform(shape, seed + 1, 0, 1) < bg] = 0
    return lab


NB = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [2, 0, 0], [0, 3, 0], [0, 0, 4], [0, 2, -3], [-1, 0, 0], [0, -2, 1]], dtype=np.int32)


def test_oracle_affinities_and_balance_by_hand():
    seg = np.array([[[1, 1, 2, 0, 2]]])                      # one line along x
    aff = O.seg_to_affgraph(seg, [[0, 0, 1], [0, 0, -1], [0, 0, 2]])
    assert aff[0, 0, 0].tolist() == [1, 0, 0, 0, 0]          # p and p+1 equal and foreground; last voxel has no partner
    assert aff[1, 0, 0].tolist() == [0, 1, 0, 0, 0]          # offset -1: first voxel has no partner
    assert aff[2, 0, 0].tolist() == [0, 0, 1, 0, 0]          # 2 == 2 across the background voxel
    labels = np.array([1, 1, 1, 0, 0, 0, 0, 0, 1, 0], dtype=np.float32)
    mask = np.array([1, 1, 1, 1, 1, 1, 1, 1, 0, 0], dtype=np.float32)
    w = O.balance_labels(labels, mask)                        # masked-in: 3 ones, 5 zeros of 8
    assert np.allclose(w[:3], 1 / 2 / (3 / 8)) and np.allclose(w[3:8], 1 / 2 / (5 / 8)) and (w[8:] == 0).all()
    rare = np.zeros(100, np.float32); rare[0] = 1            # 1 % positives: the share is clipped at 5 %
    assert np.isclose(O.balance_labels(rare)[0], 1 / 2 / 0.05) and np.isclose(O.balance_labels(rare)[1], 1 / 2 / 0.95)


def test_oracle_lsd_of_a_half_space():
    """A segment filling the half space x < 20 of a 24 x 24 x 40 volume, sigma 2 voxels: deep inside (5 sigma from the face and from the
    array's edge) the mass is ~1, the mean offset ~0 and the variances ~sigma^2 (normalised: ~1); at the face the mass is ~0.6 and the
    mean offset along x points into the segment (measured 0.176 = -1.3 voxels)."""
    seg = np.zeros((24, 24, 40), np.int64); seg[:, :, :20] = 5
    d = O.lsd_targets(seg, sigma=2.0, voxel_size=(1, 1, 1), downsample=1)
    c = d[:, 12, 12, 10]
    assert abs(c[9] - 1) < 1e-3 and np.allclose(c[:3], 0.5, atol=2e-3) and np.allclose(c[3:6], 1.0, atol=0.02) and np.allclose(c[6:9], 0.5, atol=1e-3)
    f = d[:, 12, 12, 19]
    assert 0.5 < f[9] < 0.65 and f[2] < 0.5 - 0.1 and abs(f[0] - 0.5) < 1e-3 and f[5] < f[3]
    assert (d[:, :, :, 20:] == 0).all()                      # background
    k = O.gaussian_kernel1d(2.0)
    assert len(k) == 13 and abs(k.sum() - 1) < 1e-12 and k[6] == k.max()


@pytest.mark.gpu
@pytest.mark.parametrize("shape,off,osh,use_mask", [((20, 24, 28), (0, 0, 0), None, True), ((26, 22, 30), (3, 2, 4), (18, 17, 20), False),
                                                     ((9, 40, 40), (1, 8, 8), (7, 24, 24), True)])
def test_gpu_affinities_and_balance_bit_exact(shape, off, osh, use_mask):
    from raygun_amd import targets as T
    dev = torch.device("cuda")
    lab = _labels(shape, 3)
    msk = (synth.uniform(shape, 5, 0, 1) > 0.15) if use_mask else None
    osh = osh or tuple(s - o for s, o in zip(shape, off))
    ra, rm = O.add_affinities(lab, NB, None if msk is None else msk.astype(np.int32), off, osh)
    ga, gm = T.add_affinities(torch.from_numpy(lab).to(dev), NB, None if msk is None else torch.from_numpy(msk).to(dev), off, osh)
    assert ga.dtype == torch.float32 and tuple(ga.shape) == (len(NB),) + tuple(osh)
    assert np.array_equal(ga.cpu().numpy(), ra) and (gm is None) == (rm is None)
    if rm is not None:
        assert np.array_equal(gm.cpu().numpy(), rm)
    rw = O.balance_labels(ra, rm)
    gw = T.balance_labels(ga, gm)
    assert np.array_equal(gw.cpu().numpy().view(np.uint32), rw.view(np.uint32)), float(np.abs(gw.cpu().numpy() - rw).max())
    # a class that never occurs among the masked-in voxels keeps weight 0; all-masked-out input gives all zeros
    z = torch.zeros(1000, device=dev)
    assert float(T.balance_labels(z).unique()) == pytest.approx(1 / 2 / 0.95)
    assert float(T.balance_labels(z + 1, torch.zeros(1000, device=dev)).abs().sum()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("shape,off,osh,sigma,vs,df", [((40, 44, 48), (0, 0, 0), None, 3.0, (1, 1, 1), 1),
                                                       ((56, 60, 52), (10, 12, 8), (36, 36, 36), 120.0, (30, 30, 30), 2),
                                                       ((30, 64, 64), (4, 16, 16), (22, 32, 32), (80.0, 40.0, 40.0), (40, 8, 8), 2)])
def test_gpu_lsd_targets_vs_oracle(shape, off, osh, sigma, vs, df):
    """lsd's descriptors: the kernel's direct window sums against the oracle's per-segment scipy gaussian filters — whole volume and an
    interior ROI, isotropic and anisotropic voxels, with and without sub-sampling (MTLSD default: sigma 300 at 30 nm voxels,
    downsample 2 = sigma 5 sub voxels; here 120 / 30 / 2 = 2 for the oracle's sake)."""
    from raygun_amd import targets as T
    dev = torch.device("cuda")
    lab = _labels(shape, 7, n_seg=9, bg=0.1)
    osh = osh or tuple(s - o for s, o in zip(shape, off))
    ref = O.lsd_targets(lab, sigma, vs, df, off, osh)
    got = T.lsd_targets(torch.from_numpy(lab).to(dev), sigma, vs, df, off, osh).cpu().numpy()
    assert got.shape == ref.shape == (10,) + tuple(osh) and got.dtype == np.float32
    err = np.abs(got - ref).max(axis=(1, 2, 3))
    # mean offsets and mass: a few float32 ulps.  Variances / Pearson coefficients: the oracle follows the package's E[c c'] - E[c] E[c']
    # on ABSOLUTE float32 coordinates (cancellation ~3e-5 at these extents), the kernel sums coordinates relative to the voxel
    assert err[[0, 1, 2, 9]].max() <= 3e-6 and err[3:9].max() <= 1e-4, err
    assert (got[:, lab[tuple(slice(o, o + s) for o, s in zip(off, osh))] == 0] == 0).all()


@pytest.mark.gpu
def test_mtlsd_batches_from_label_sources(tmp_path):
    """MTLSDCropBatches (train/data.py): an MTLSD config with `sources` trains from label volumes — crops, normalisation and the GPU
    targets arrive under the names MTLSDModel.forward / WeightedMSELoss_MTLSD.forward take, with the oracle's values."""
    from raygun_amd import zarrio
    from raygun_amd.torch.train.data import MTLSDCropBatches
    dev = torch.device("cuda")
    shape = (64, 72, 80)
    raw = (synth.uniform(shape, 11, 0, 1) * 255).astype(np.uint8)
    lab = _labels(shape, 12, n_seg=12, bg=0.05)
    msk = (synth.uniform(shape, 13, 0, 1) > 0.02).astype(np.uint8)
    path = str(tmp_path / "gt.zarr")
    for name, arr in (("volumes/raw", raw), ("volumes/labels", lab.astype(np.uint64)), ("volumes/mask", msk)):
        d = zarrio.prepare_ds(path, name, shape, (30, 30, 30), arr.dtype, write_size=(32, 32, 32), compressor=None)
        d[...] = arr
    nb = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [2, 0, 0], [0, 2, 0], [0, 0, 2]])
    it = iter(MTLSDCropBatches([{"path": path, "raw": "volumes/raw", "labels": "volumes/labels", "labels_mask": "volumes/mask"}],
                               (36, 36, 36), (20, 20, 20), nb, (30, 30, 30), {"sigma": 60, "downsample": 2}, 2, dev, seed=1))
    b = next(it)
    assert {k: tuple(v.shape) for k, v in b.items()} == {"raw": (2, 1, 36, 36, 36), "gt_affs": (2, 6, 20, 20, 20), "affs_weights": (2, 6, 20, 20, 20),
                                                         "gt_lsds": (2, 10, 20, 20, 20), "lsds_weights": (2, 10, 20, 20, 20)}
    assert float(b["raw"].min()) >= -1 and float(b["raw"].max()) <= 1 and all(v.dtype == torch.float32 and v.is_cuda for v in b.values())
    assert set(np.unique(b["gt_affs"].cpu().numpy())) <= {0.0, 1.0} and float(b["affs_weights"].max()) <= 10.0 + 1e-6
    # locate the first sample's crop in the volume through its raw values and recompute its targets with the oracle
    x = ((b["raw"][0, 0].cpu().numpy() + 1) / 2 * 255).round().astype(np.uint8)
    hits = [(z, y, xx) for z in range(shape[0] - 35) for y in range(shape[1] - 35) for xx in range(shape[2] - 35)
            if raw[z, y, xx] == x[0, 0, 0] and np.array_equal(raw[z:z + 36, y:y + 36, xx:xx + 36], x)]
    assert len(hits) == 1
    o = [h + 8 for h in hits[0]]
    ra, rm = O.add_affinities(lab, nb, msk.astype(np.int32), o, (20, 20, 20))
    assert np.array_equal(b["gt_affs"][0].cpu().numpy(), ra)
    assert np.allclose(b["affs_weights"][0].cpu().numpy(), O.balance_labels(ra, rm), rtol=1e-6)

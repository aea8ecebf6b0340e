"""TEST INFRASTRUCTURE — CPU restatement (numpy / scipy) of the train-time target generation the reference's MTLSD data pipe runs
on every batch (reference call sites: src/raygun/io/MTLSDDataPipe.py:99-131):

    AddLocalShapeDescriptor(labels, gt_lsds, lsds_mask, sigma, downsample)      lsd.train.gp            (:100-105)
    gp.AddAffinities(neighborhood, labels, gt_affs, labels_mask, affs_mask)     gunpowder               (:107-114)
    gp.BalanceLabels(gt_affs, affs_weights, affs_mask)                          gunpowder               (:126-130)

PARITY UNPINNED: gunpowder (fork rhoadesScholar/gunpowder@raygun, unpinned branch, setup.cfg:30) and lsd (funkelab/lsd, unpinned)
are not installed here and not vendored under /root/reference, and no reference test holds vectors for these nodes.  The functions
below restate the algorithms those packages publish — gunpowder/nodes/add_affinities.py `seg_to_affgraph`,
gunpowder/nodes/balance_labels.py `BalanceLabels.__balance`, lsd/train/local_shape_descriptor.py `LsdExtractor.get_descriptors` /
`__get_stats` (gaussian mode) — as literally as they can be recalled; the GPU kernels (raygun_amd/csrc/targets.hip) are held to
THESE functions.  Only tests/ may import this module.
"""
import numpy as np


def seg_to_affgraph(seg, nhood):
    """aff[e, p] = (seg[p] == seg[p + nhood[e]]) * (seg[p] > 0) * (seg[p + nhood[e]] > 0) where both voxels exist, else 0 (int32)."""
    seg = np.asarray(seg)
    nhood = np.asarray(nhood, dtype=np.int64)
    shape = seg.shape
    aff = np.zeros((nhood.shape[0],) + shape, dtype=np.int32)
    for e in range(nhood.shape[0]):
        a = tuple(slice(max(0, -int(o)), min(s, s - int(o))) for o, s in zip(nhood[e], shape))
        b = tuple(slice(max(0, int(o)), min(s, s + int(o))) for o, s in zip(nhood[e], shape))
        aff[(e,) + a] = (seg[a] == seg[b]) * (seg[a] > 0) * (seg[b] > 0)
    return aff


def add_affinities(labels, neighborhood, labels_mask=None, out_offset=(0, 0, 0), out_shape=None, dtype=np.float32):
    """gp.AddAffinities on an in-memory label array: affinities over the whole array, cropped to [out_offset, out_offset +
    out_shape) (gunpowder requests the labels grown by the neighbourhood's extent and crops the result to the requested ROI);
    the affinity mask is the same graph of `labels_mask`.  Returns (affs dtype, affs_mask dtype | None)."""
    labels = np.asarray(labels).astype(np.int32)
    out_shape = tuple(labels.shape) if out_shape is None else tuple(out_shape)
    crop = (slice(None),) + tuple(slice(o, o + s) for o, s in zip(out_offset, out_shape))
    affs = seg_to_affgraph(labels, neighborhood)[crop].astype(dtype)
    mask = None
    if labels_mask is not None:
        mask = seg_to_affgraph(np.asarray(labels_mask).astype(np.int32), neighborhood)[crop].astype(dtype)
    return affs, mask


def balance_labels(labels, mask=None, num_classes=2, clipmin=0.05, clipmax=0.95):
    """gp.BalanceLabels with one slab: error scale = mask * w[label], w[c] = 1 / num_classes / clip(fraction of class c among the
    masked-in samples)."""
    labels = np.asarray(labels)
    scale = np.ones(labels.shape, dtype=np.float32)
    if mask is not None:
        scale *= np.asarray(mask)
    lab = labels.astype(np.int64)
    masked_in = scale.sum()
    classes, counts = np.unique(lab[np.nonzero(scale)], return_counts=True)
    fracs = counts.astype(float) / masked_in if masked_in > 0 else np.zeros(counts.size)
    np.clip(fracs, clipmin, clipmax, fracs)
    w_sparse = 1.0 / float(num_classes) / fracs
    w = np.zeros(num_classes)
    w[classes] = w_sparse
    scale *= np.take(w, lab)
    return scale


def _outer(a):
    k = a.shape[0]
    return np.stack([a[i] * a[j] for i in range(k) for j in range(k)])


def lsd_targets(labels, sigma, voxel_size=(1, 1, 1), downsample=1, out_offset=(0, 0, 0), out_shape=None):
    """lsd's LsdExtractor.get_descriptors (3-D, mode "gaussian") on an in-memory label array: per segment, Gaussian-weighted local
    statistics of the segment's voxels on the `downsample`d grid — mean offset (3), variance (3), Pearson coefficients zy / zx / yx
    (3), mass (1) — nearest-neighbour upsampled, masked by the segment, normalised into [0, 1].  `sigma` in world units (scalar or
    per axis).  Returns float32 (10, *out_shape) for the ROI [out_offset, out_offset + out_shape) of the array."""
    from scipy.ndimage import gaussian_filter
    seg = np.asarray(labels)
    vs = np.asarray(voxel_size, dtype=np.float64)
    sig = np.full(3, sigma, dtype=np.float64) if np.isscalar(sigma) else np.asarray(sigma, dtype=np.float64)
    df = int(downsample)
    out_shape = tuple(seg.shape) if out_shape is None else tuple(out_shape)
    roi = tuple(slice(o, o + s) for o, s in zip(out_offset, out_shape))
    sub_vs = vs * df
    sub_shape = tuple(-(-s // df) for s in seg.shape)
    sub_sigma_voxel = sig / sub_vs
    grid = np.meshgrid(*[np.arange(0, n * v, v)[:n] for n, v in zip(sub_shape, sub_vs)], indexing="ij")
    coords = np.array(grid, dtype=np.float32)
    agg = lambda a: gaussian_filter(a, sigma=sub_sigma_voxel, mode="constant", cval=0.0, truncate=3.0)  # noqa: E731
    desc = np.zeros((10,) + out_shape, dtype=np.float32)
    entries = [0, 4, 8, 1, 2, 5]
    rep = lambda a: np.repeat(np.repeat(np.repeat(a, df, 1), df, 2), df, 3)[(slice(None),) + tuple(slice(0, s) for s in seg.shape)]  # noqa: E731
    for label in np.unique(seg):
        if label == 0:
            continue
        mask = (seg == label).astype(np.float32)
        sub_mask = mask[::df, ::df, ::df]
        masked_coords = coords * sub_mask
        count = agg(sub_mask)
        count[count == 0] = 1
        mean = np.array([agg(masked_coords[d]) for d in range(3)])
        mean /= count
        mean_offset = mean - coords
        outer = _outer(masked_coords)
        cov = np.array([agg(outer[d]) for d in entries])
        cov /= count
        cov -= _outer(mean)[entries]
        variance, pearson = cov[[0, 1, 2]], cov[[3, 4, 5]]
        variance[variance < 1e-3] = 1e-3
        pearson[0] /= np.sqrt(variance[0] * variance[1])
        pearson[1] /= np.sqrt(variance[0] * variance[2])
        pearson[2] /= np.sqrt(variance[1] * variance[2])
        variance[0] /= sig[0] ** 2; variance[1] /= sig[1] ** 2; variance[2] /= sig[2] ** 2
        sub_desc = np.concatenate([mean_offset, variance, pearson, count[None]])
        desc += (rep(sub_desc) * mask[None])[(slice(None),) + roi]
    desc[[0, 1, 2]] = desc[[0, 1, 2]] / sig.astype(np.float32)[:, None, None, None] * 0.5 + 0.5
    desc[[6, 7, 8]] = desc[[6, 7, 8]] * 0.5 + 0.5
    desc[[0, 1, 2, 6, 7, 8]] *= (seg[roi] != 0)
    np.clip(desc, 0.0, 1.0, out=desc)
    return desc


def gaussian_kernel1d(sigma, truncate=3.0):
    """scipy.ndimage's 1-D Gaussian kernel (order 0): radius int(truncate * sigma + 0.5), weights normalised to sum 1 (float64)."""
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (float(sigma) * float(sigma)) * x ** 2)
    return phi / phi.sum()

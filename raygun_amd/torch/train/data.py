"""Batch sources for BaseSystem.train().  The reference feeds its train step from gunpowder graphs (io/*DataPipe.py —
out of scope); what reaches the model is a dict of arrays named after the model's forward arguments and the
loss's extra arguments (BaseTrain.py:42-60).  Two sources produce such dicts here:

  * SyntheticBatches — counter-hash uniform volumes (raygun_amd.synth), the benchmark / smoke-test input;
  * ZarrCropBatches — random crops of the config's `sources` for the CycleGAN (CycleDataPipe: Normalize to [0,1] then
    IntensityScaleShift(2, -1), io/CycleDataPipe.py:46-60), read through raygun_amd.zarrio.
  * MTLSDCropBatches — random crops of raw / labels / labels_mask from the config's `sources` with the MTLSD targets (LSDs,
    affinities, class-balance weights: io/MTLSDDataPipe.py:99-131) generated ON THE GPU by raygun_amd.targets;
    the reference's elastic / intensity augmentations are gunpowder nodes and stay out of scope.
"""
import numpy as np
import torch

from ... import synth


class SyntheticBatches:
    """Endless iterable of batch dicts on `device`; rank-distinct seeds under data parallelism."""

    def __init__(self, specs, device, seed=42, rank=0):
        self.specs, self.device, self.seed = specs, device, seed + 1000 * rank

    def __iter__(self):
        it = 0
        while True:
            b = {}
            for i, (name, (shape, lo, hi)) in enumerate(self.specs.items()):
                b[name] = torch.from_numpy(synth.uniform(tuple(shape), self.seed + 17 * it + i, lo, hi)).to(self.device)
            it += 1
            yield b


class ZarrCropBatches:
    """real_A / real_B crops of side_length^ndims voxels at uniformly random positions, scaled to [-1, 1]."""

    def __init__(self, sources, side_length, ndims, batch_size, device, seed=42, rank=0):
        from ... import zarrio
        self.arrays = {k: zarrio.open_ds(v["path"], v["real_name"]) for k, v in sources.items()}
        self.side, self.ndims, self.bs, self.device = int(side_length), ndims, batch_size, device
        self.rng = np.random.default_rng(seed + 1000 * rank)

    def _crop(self, a):
        shp = (1,) * (3 - self.ndims) + (self.side,) * self.ndims
        lo = [int(self.rng.integers(0, n - s + 1)) for n, s in zip(a.shape[-3:], shp)]
        x = a[tuple(slice(o, o + s) for o, s in zip(lo, shp))].astype(np.float32)
        if np.issubdtype(a.dtype, np.integer):
            x /= float(np.iinfo(a.dtype).max)
        x = x * 2 - 1
        return x[0] if self.ndims == 2 else x

    def __iter__(self):
        while True:
            yield {f"real_{k}": torch.from_numpy(np.stack([self._crop(a)[None] for _ in range(self.bs)])).to(self.device)
                   for k, a in self.arrays.items()}



class MTLSDCropBatches:
    """Batches for the MTLSD system from label volumes (io/MTLSDDataPipe.py): per sample a random location (RandomLocation, with
    `min_masked` of the label mask when the config asks for it), raw normalised to [0, 1] then scaled to [-1, 1]
    (gp.Normalize + IntensityScaleShift(2, -1), :146,:132), and on the GPU: gt_lsds / lsds_weights (AddLocalShapeDescriptor),
    gt_affs (AddAffinities over the label crop grown by the neighbourhood's extent and the LSD context) and affs_weights
    (BalanceLabels over the affinity mask).  Yields the names MTLSDModel.forward / WeightedMSELoss_MTLSD.forward take.

    Several sources are JOINED as the reference joins them (io/MTLSDDataPipe.py:163-164, gp.MergeProvider): each source contributes the
    datasets it names, together they form one provider, and a dataset named by two sources is an error (gunpowder asserts the same)."""

    def __init__(self, sources, input_shape, output_shape, neighborhood, voxel_size, lsd_kwargs, batch_size, device, seed=42, rank=0,
                 random_location_kwargs=None):
        from ... import zarrio
        merged = {}
        for src in (sources if isinstance(sources, (list, tuple)) else [sources]):
            for k in ("raw", "labels", "labels_mask"):
                if k in src:
                    if k in merged:
                        raise ValueError(f"MTLSD sources: {k!r} is provided by more than one source (the reference merges its sources with "
                                         "gp.MergeProvider, which takes every array from exactly one provider)")
                    merged[k] = zarrio.open_ds(src["path"], src[k])
        if "raw" not in merged or "labels" not in merged:
            raise ValueError("the MTLSD sources must name at least a 'raw' and a 'labels' dataset")
        self.srcs = [merged]
        self.inp, self.out = tuple(int(v) for v in input_shape), tuple(int(v) for v in output_shape)
        self.nb = np.asarray(neighborhood, dtype=np.int32).reshape(-1, 3)
        self.vs = tuple(float(v) for v in voxel_size)
        self.lsd = dict(lsd_kwargs) if lsd_kwargs else None
        self.bs, self.device = int(batch_size), device
        self.rng = np.random.default_rng(seed + 1000 * rank)
        self.min_masked = float((random_location_kwargs or {}).get("min_masked", 0.0))
        # label context: the neighbourhood's extent, and 3 sigma of the LSD window (in voxels, rounded up to the sub-sampling grid)
        ctx = np.abs(self.nb).max(0)
        if self.lsd:
            df = int(self.lsd.get("downsample", 1))
            sig = np.full(3, float(self.lsd["sigma"])) if np.isscalar(self.lsd["sigma"]) else np.asarray(self.lsd["sigma"], dtype=float)
            r = np.array([int(3.0 * s / (v * df) + 0.5) * df + df for s, v in zip(sig, self.vs)])
            ctx = np.maximum(ctx, r)
        self.ctx = tuple(int(c) for c in ctx)

    def _sample(self):
        from ... import targets as T
        a = self.srcs[int(self.rng.integers(0, len(self.srcs)))]
        shp = a["raw"].shape[-3:]
        pad = tuple((i - o) // 2 for i, o in zip(self.inp, self.out))              # output ROI inside the input ROI
        for _ in range(64):
            lo = [int(self.rng.integers(0, n - s + 1)) for n, s in zip(shp, self.inp)]
            olo = [l + p for l, p in zip(lo, pad)]
            if "labels_mask" in a and self.min_masked > 0:
                m = a["labels_mask"][tuple(slice(o, o + s) for o, s in zip(olo, self.out))]
                if float((np.asarray(m) > 0).mean()) < self.min_masked:
                    continue
            break
        raw = a["raw"][tuple(slice(o, o + s) for o, s in zip(lo, self.inp))]
        x = np.asarray(raw).astype(np.float32)
        if np.issubdtype(np.asarray(raw).dtype, np.integer):
            x /= float(np.iinfo(np.asarray(raw).dtype).max)
        x = x * 2 - 1
        # label crop grown by the context, clipped to the volume (outside = background, as gp.Pad would supply)
        glo = [max(0, o - c) for o, c in zip(olo, self.ctx)]
        ghi = [min(n, o + s + c) for n, o, s, c in zip(shp, olo, self.out, self.ctx)]
        sl = tuple(slice(l, h) for l, h in zip(glo, ghi))
        lab = torch.from_numpy(np.ascontiguousarray(np.asarray(a["labels"][sl]).astype(np.int64))).to(self.device)
        msk = torch.from_numpy(np.ascontiguousarray(np.asarray(a["labels_mask"][sl]) > 0)).to(self.device) if "labels_mask" in a else None
        off = [o - l for o, l in zip(olo, glo)]
        affs, amask = T.add_affinities(lab, self.nb, msk, off, self.out)
        b = {"raw": torch.from_numpy(x[None]).to(self.device), "gt_affs": affs, "affs_weights": T.balance_labels(affs, amask)}
        if self.lsd:
            # (the sub-sampling grid is anchored at the crop's origin, as it is at the batch's label array in the reference)
            b["gt_lsds"] = T.lsd_targets(lab, self.lsd["sigma"], self.vs, int(self.lsd.get("downsample", 1)), off, self.out)
            fg = (lab[tuple(slice(o, o + s) for o, s in zip(off, self.out))] != 0).to(torch.float32)   # lsd's mask: labels != 0, on the uncast ids
            b["lsds_weights"] = fg[None].expand(10, *self.out).contiguous()
        return b

    def __iter__(self):
        while True:
            samples = [self._sample() for _ in range(self.bs)]
            yield {k: torch.stack([s[k] for s in samples]) for k in samples[0]}

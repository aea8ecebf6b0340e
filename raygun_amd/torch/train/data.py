"""Batch sources for BaseSystem.train().  The reference feeds its train step from gunpowder graphs (io/*DataPipe.py —
out of scope); what reaches the model is a dict of arrays named after the model's forward arguments and the
loss's extra arguments (BaseTrain.py:42-60).  Two sources produce such dicts here:

  * SyntheticBatches — counter-hash uniform volumes (raygun_amd.synth), the benchmark / smoke-test input;
  * ZarrCropBatches — random crops of the config's `sources` for the CycleGAN (CycleDataPipe: Normalize to [0,1] then
    IntensityScaleShift(2, -1), io/CycleDataPipe.py:46-60), read through raygun_amd.zarrio.
The MTLSD targets (affinities, LSDs, balance weights from a label volume: io/MTLSDDataPipe.py:99-131) are SURVEY.md 8(f4),
not built: an MTLSD config with `sources` and no "synthetic_data" raises.
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

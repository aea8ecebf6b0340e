"""MTLSD (U-Net affinities + LSDs) system — mirror of raygun.torch.systems.MTLSD (reference: systems/MTLSD.py:19-107,
defaults from torch/default_configs/default_MTLSD_conf.json)."""
import numpy as np

from .BaseSystem import BaseSystem
from ..losses import WeightedMSELoss_MTLSD
from ..models import MTLSDModel
from ..optimizers import get_base_optimizer

_NEIGHBORHOOD = [[1, 0, 0], [0, 1, 0], [0, 0, 1], [2, 0, 0], [0, 2, 0], [0, 0, 2],
                 [4, 0, 0], [0, 4, 0], [0, 0, 4], [8, 0, 0], [0, 8, 0], [0, 0, 8]]


def diagonal_offsets(n_diagonals):
    """In-plane diagonal offsets added to the neighbourhood (systems/MTLSD.py:32-45 == segment.py:173-181)."""
    t = np.linspace(0, np.pi, num=n_diagonals, endpoint=False)
    pos, neg = np.round(n_diagonals * np.sin(t)), np.round(n_diagonals * np.cos(t))
    return np.stack([0 * pos, pos, neg], axis=-1)


class MTLSD(BaseSystem):
    DEFAULTS = dict(
        framework="torch", system="MTLSD", trainer_base="BaseTrain", ndims=3, batch_size=1, loss_kwargs={},
        model_kwargs={"unet_kwargs": {"input_nc": 1, "ngf": 12, "fmap_inc_factor": 6,
                                      "downsample_factors": [[2, 2, 2], [2, 2, 2], [2, 2, 2]], "constant_upsample": True},
                      "num_affs": 20},
        neighborhood=_NEIGHBORHOOD, n_diagonals=8, optim_type="Adam", optim_kwargs={"lr": 0.5e-4},
        voxel_size=[1, 1, 1], input_shape=[196, 196, 196], output_shape=[104, 104, 104], num_epochs=20000,
        model_path="./models/", model_name="model", random_seed=42)

    def __init__(self, config=None):
        super().__init__(config=config)
        if self.ndims is None:
            self.ndims = int(sum(np.array(self.voxel_size) == np.min(self.voxel_size)))
        self.neighborhood = np.array(self.neighborhood)
        if self.n_diagonals > 0:
            self.neighborhood = np.concatenate([self.neighborhood, diagonal_offsets(self.n_diagonals)]).astype(np.int8)
        self.voxel_size = tuple(self.voxel_size)
        self.output_size = tuple(o * v for o, v in zip(self.output_shape, self.voxel_size))
        self.input_size = tuple(i * v for i, v in zip(self.input_shape, self.voxel_size))

    def setup_networks(self):
        pass

    def setup_model(self):
        kw = dict(self.model_kwargs)
        uk = dict(kw["unet_kwargs"])
        uk["downsample_factors"] = [tuple(f) for f in uk["downsample_factors"]]
        kw["unet_kwargs"] = uk
        self.model = MTLSDModel(**kw)

    def setup_optimization(self, grad_sync=None):
        self.optimizer = get_base_optimizer(self.optim_type)(params=self.model.parameters(), **self.optim_kwargs)
        self.loss = WeightedMSELoss_MTLSD(**self.loss_kwargs)
        self.grad_sync = grad_sync

    def batch_specs(self):
        """Names = MTLSDModel.forward's argument + WeightedMSELoss_MTLSD.forward's extras (MTLSDModel.py:52, loss :27-35)."""
        n = self.batch_size
        i, o = tuple(self.input_shape), tuple(self.output_shape)
        na = self.model_kwargs.get("num_affs", 3)
        return {"raw": ((n, 1) + i, -1.0, 1.0), "gt_lsds": ((n, 10) + o, 0.0, 1.0), "lsds_weights": ((n, 10) + o, 1.0, 1.0),
                "gt_affs": ((n, na) + o, 0.0, 1.0), "affs_weights": ((n, na) + o, 1.0, 1.0)}

    def source_batches(self, rank):
        """Training from the config's `sources` (io/MTLSDDataPipe.py): random crops + GPU target generation (train/data.py)."""
        from ..train.data import MTLSDCropBatches
        return MTLSDCropBatches(self.sources, self.input_shape, self.output_shape, self.neighborhood, self.voxel_size, getattr(self, "lsd_kwargs", None),
                                self.batch_size, self.device, getattr(self, "random_seed", None) or 42, rank, getattr(self, "random_location_kwargs", None))

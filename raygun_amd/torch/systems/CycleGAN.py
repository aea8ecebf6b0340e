"""CycleGAN system — mirror of raygun.torch.systems.CycleGAN (reference: systems/CycleGAN.py:19-269,
defaults from torch/default_configs/default_cycleGAN_conf.json:1-63)."""
import itertools

import numpy as np

from .BaseSystem import BaseSystem
from ..losses import LinkCycleLoss, SplitCycleLoss
from ..models import CycleModel
from ..optimizers import BaseDummyOptimizer, get_base_optimizer


def _g_lambda():
    return {s: {"l1_loss": {"cycled": 10, "identity": 0.5}, "gan_loss": {"fake": 1, "cycled": 0}} for s in "AB"}


class CycleGAN(BaseSystem):
    DEFAULTS = dict(
        common_voxel_size=None, ndims=None, gnet_type="unet",
        gnet_kwargs={"input_nc": 1, "ngf": 64, "padding_type": "same"},
        dnet_type="patchdiscriminator",
        dnet_kwargs={"input_nc": 1, "downsampling_kw": 2, "kw": 3, "n_layers": 3, "ngf": 64},
        loss_type="link",
        loss_kwargs={"g_lambda_dict": _g_lambda(),
                     "d_lambda_dict": {s: {"real": 1, "fake": 1, "cycled": 0} for s in "AB"}},
        sampling_bottleneck=False,
        g_optim_type="Adam", g_optim_kwargs={"betas": [0.9, 0.999], "lr": 1e-5, "weight_decay": 0},
        d_optim_type="Adam", d_optim_kwargs={"betas": [0.9, 0.999], "lr": 1e-5, "weight_decay": 0},
        interp_order=None, side_length=64, batch_size=1, num_workers=11, cache_size=50, spawn_subprocess=False,
        num_epochs=20000, log_every=20, save_every=2000, model_path="./models/", model_name="CycleGAN",
        tensorboard_path="./tensorboard/", verbose=True, checkpoint=None, pretrain_gnet=False, random_seed=42,
        trainer_base="CycleTrain", freeze_norms_at=None, snapshot_every=None)

    def __init__(self, config=None):
        super().__init__(config=config)
        if self.common_voxel_size is None:
            raise RuntimeError("raygun_amd.CycleGAN: 'common_voxel_size' must be given (the reference reads it from the "
                               "source dataset through daisy, systems/CycleGAN.py:28-33)")
        self.common_voxel_size = tuple(self.common_voxel_size)
        if self.ndims is None:
            self.ndims = int(sum(np.array(self.common_voxel_size) == np.min(self.common_voxel_size)))

    def setup_networks(self):
        """CycleGAN.py:133-138 — get_network pops keys, so hand it copies."""
        self.netG1 = self.get_network(self.gnet_type, dict(self.gnet_kwargs))
        self.netG2 = self.get_network(self.gnet_type, dict(self.gnet_kwargs))
        self.netD1 = self.get_network(self.dnet_type, dict(self.dnet_kwargs))
        self.netD2 = self.get_network(self.dnet_type, dict(self.dnet_kwargs))

    def setup_model(self):
        """CycleGAN.py:140-165."""
        if not hasattr(self, "netG1"):
            self.setup_networks()
        scale_factor_A = scale_factor_B = None
        if self.sampling_bottleneck:
            scale_factor_A = tuple(np.divide(self.common_voxel_size, self.A_voxel_size)[-self.ndims:])
            if not any(s < 1 for s in scale_factor_A):
                scale_factor_A = None
            scale_factor_B = tuple(np.divide(self.common_voxel_size, self.B_voxel_size)[-self.ndims:])
            if not any(s < 1 for s in scale_factor_B):
                scale_factor_B = None
        self.model = CycleModel(self.netG1, self.netG2, scale_factor_A, scale_factor_B,
                                split=self.loss_type.lower() == "split", freeze_norms_at=self.freeze_norms_at)

    def setup_optimization(self, grad_sync=None):
        """CycleGAN.py:167-233."""
        self.optimizer_D = get_base_optimizer(self.d_optim_type)(
            itertools.chain(self.netD1.parameters(), self.netD2.parameters()), **self.d_optim_kwargs)
        scheduler = getattr(self, "scheduler", None)
        scheduler_kwargs = getattr(self, "scheduler_kwargs", {}) if scheduler is not None else {}
        if self.loss_type.lower() == "link":
            self.optimizer_G = get_base_optimizer(self.g_optim_type)(
                itertools.chain(self.netG1.parameters(), self.netG2.parameters()), **self.g_optim_kwargs)
            self.optimizer = BaseDummyOptimizer(optimizer_G=self.optimizer_G, optimizer_D=self.optimizer_D,
                                                scheduler=scheduler, scheduler_kwargs=scheduler_kwargs)
            self.loss = LinkCycleLoss(self.netD1, self.netG1, self.netD2, self.netG2, self.optimizer_G, self.optimizer_D,
                                      self.ndims, grad_sync=grad_sync, **self.loss_kwargs)
        elif self.loss_type.lower() == "split":
            self.optimizer_G1 = get_base_optimizer(self.g_optim_type)(self.netG1.parameters(), **self.g_optim_kwargs)
            self.optimizer_G2 = get_base_optimizer(self.g_optim_type)(self.netG2.parameters(), **self.g_optim_kwargs)
            self.optimizer = BaseDummyOptimizer(optimizer_G1=self.optimizer_G1, optimizer_G2=self.optimizer_G2,
                                                optimizer_D=self.optimizer_D, scheduler=scheduler,
                                                scheduler_kwargs=scheduler_kwargs)
            self.loss = SplitCycleLoss(self.netD1, self.netG1, self.netD2, self.netG2, self.optimizer_G1, self.optimizer_G2,
                                       self.optimizer_D, self.ndims, grad_sync=grad_sync, **self.loss_kwargs)
        else:
            raise NotImplementedError("Unexpected Loss Style. Accepted options are 'cycle' or 'split'")

    def batch_specs(self):
        shp = (self.batch_size, self.gnet_kwargs.get("input_nc", 1)) + (self.side_length,) * self.ndims
        return {"real_A": (shp, -1.0, 1.0), "real_B": (shp, -1.0, 1.0)}

    def source_batches(self, rank):
        """Random crops of sources["A"/"B"] ({"path", "real_name"}, systems/CycleGAN.py:236-248), scaled as CycleDataPipe does."""
        from ..train.data import ZarrCropBatches
        return ZarrCropBatches(self.sources, self.side_length, self.ndims, self.batch_size, self.device,
                               getattr(self, "random_seed", None) or 42, rank)

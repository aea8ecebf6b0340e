"""CycleGAN 'link' train step — drop-in for raygun.torch.losses.LinkCycleLoss (reference: losses/LinkCycleLoss.py:10-176).

forward() IS the train step: generator backward (both sides, retain_graph) + one optimiser step over G1 u G2,
then discriminator backward + step.  `grad_sync` (optional callable(list_of_params)) is invoked right before each
optimiser step — the data-parallel hook (RCCL all-reduce) that has no counterpart in the reference."""
import logging

import torch

from .BaseCompetentLoss import BaseCompetentLoss
from ...utils import passing_locals

logger = logging.Logger(__name__, "INFO")


def _default_g_lambda():
    return {s: {"l1_loss": {"cycled": 10, "identity": 0}, "gan_loss": {"fake": 1, "cycled": 0}} for s in "AB"}


def _default_d_lambda():
    return {s: {"real": 1, "fake": 1, "cycled": 0} for s in "AB"}


def _params_of(opt):
    return [p for g in opt.param_groups for p in g["params"]]


class LinkCycleLoss(BaseCompetentLoss):
    def __init__(self, netD1, netG1, netD2, netG2, optimizer_G, optimizer_D, dims, l1_loss=None,
                 g_lambda_dict=None, d_lambda_dict=None, gan_mode="lsgan", grad_sync=None, **kwargs):
        if l1_loss is None:
            l1_loss = torch.nn.SmoothL1Loss()
        if g_lambda_dict is None:
            g_lambda_dict = _default_g_lambda()
        if d_lambda_dict is None:
            d_lambda_dict = _default_d_lambda()
        super().__init__(**passing_locals(locals()))
        self.data_dict = {}

    def _sync(self, opt):
        if self.grad_sync is not None:
            self.grad_sync(_params_of(opt))

    def backward_D(self, side, dnet, data_dict):
        """LinkCycleLoss.py:43-59."""
        loss = 0
        for key, lambda_ in self.d_lambda_dict[side].items():
            if lambda_ != 0:
                this_loss = self.gan_loss(dnet(data_dict[key].detach()), key == "real")
                self.loss_dict.update({f"Discriminator_{side}/{key}": this_loss})
                loss += lambda_ * this_loss
        self._backward(loss)
        return loss

    def backward_Ds(self, data_dict, n_loop=5):
        """LinkCycleLoss.py:61-85."""
        self.set_requires_grad([self.netG1, self.netG2], False)
        self.optimizer_D.zero_grad(set_to_none=True)
        if self.gan_mode.lower() == "wgangp":
            for _ in range(n_loop):
                loss_D1 = self.backward_D("B", self.netD1, data_dict["B"])
                loss_D2 = self.backward_D("A", self.netD2, data_dict["A"])
                self._sync(self.optimizer_D)
                self._step(self.optimizer_D)
                self.clamp_weights(self.netD1)
                self.clamp_weights(self.netD2)
        else:
            loss_D1 = self.backward_D("B", self.netD1, data_dict["B"])
            loss_D2 = self.backward_D("A", self.netD2, data_dict["A"])
            self._sync(self.optimizer_D)
            self._step(self.optimizer_D)
        self.set_requires_grad([self.netG1, self.netG2], True)
        return loss_D1, loss_D2

    def _retain_graph(self):
        return True  # LinkCycleLoss.py:113

    def backward_G(self, side, gnet, dnet, data_dict):
        """LinkCycleLoss.py:87-114."""
        loss = 0
        real = data_dict["real"]
        for fcn_name, lambdas in self.g_lambda_dict[side].items():
            for key, lambda_ in lambdas.items():
                if lambda_ != 0:
                    if key == "identity" and key not in data_dict:
                        data_dict["identity"] = gnet(real)
                    pred = data_dict[key]
                    if fcn_name == "l1_loss":
                        if real.size()[-self.dims:] != pred.size()[-self.dims:]:
                            this_loss = self.l1(self.crop(real, pred.size()[-self.dims:]), pred)
                        else:
                            this_loss = self.l1(real, pred)
                    elif fcn_name == "gan_loss":
                        this_loss = self.gan_loss(dnet(pred), True)
                    else:
                        this_loss = getattr(self, fcn_name)(real, pred)
                    self.loss_dict.update({f"{fcn_name}/{key}_{side}": this_loss})
                    loss += lambda_ * this_loss
        self._backward(loss, retain_graph=self._retain_graph())
        return loss

    def backward_Gs(self, data_dict):
        """LinkCycleLoss.py:116-135."""
        self.set_requires_grad([self.netD1, self.netD2], False)
        self.optimizer_G.zero_grad(set_to_none=True)
        loss_G1 = self.backward_G("B", self.netG1, self.netD1, data_dict["B"])
        loss_G2 = self.backward_G("A", self.netG2, self.netD2, data_dict["A"])
        self._sync(self.optimizer_G)
        self._step(self.optimizer_G)
        self.set_requires_grad([self.netD1, self.netD2], True)
        return loss_G1, loss_G2

    def forward(self, real_A, fake_A, cycled_A, real_B, fake_B, cycled_B):
        """LinkCycleLoss.py:137-176."""
        self.data_dict.update({"real_A": real_A, "fake_A": fake_A, "cycled_A": cycled_A,
                               "real_B": real_B, "fake_B": fake_B, "cycled_B": cycled_B})
        if real_A.size()[-self.dims:] != fake_B.size()[-self.dims:]:
            real_A = self.crop(real_A, fake_A.size()[-self.dims:])
            real_B = self.crop(real_B, fake_B.size()[-self.dims:])
        data_dict = {"A": {"real": real_A, "fake": fake_A, "cycled": cycled_A},
                     "B": {"real": real_B, "fake": fake_B, "cycled": cycled_B}}
        loss_G1, loss_G2 = self.backward_Gs(data_dict)
        loss_D1, loss_D2 = self.backward_Ds(data_dict)
        # (the reference stores float(loss): four host reads that stall the launch queue at the end of EVERY step — 8 ms of idle GPU per 75-ms step at
        # 128^3, the host restarting from an empty queue; the entries stay 0-dim tensors here, like the per-term entries above, and whoever logs them
        # — add_log, float(v) — pays for the read when it logs)
        self.loss_dict.update({"Total_Loss/D1": loss_D1.detach(), "Total_Loss/D2": loss_D2.detach(),
                               "Total_Loss/G1": loss_G1.detach(), "Total_Loss/G2": loss_G2.detach()})
        total_loss = loss_G1 + loss_G2 + loss_D1 + loss_D2
        total_loss.backward = lambda: None  # the Train node's own loss.backward() must be a no-op (:173)
        logger.info(self.loss_dict)
        return total_loss

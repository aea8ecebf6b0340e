"""CycleGAN 'split' train step — drop-in for raygun.torch.losses.SplitCycleLoss (reference: losses/SplitCycleLoss.py:10-194).
One optimiser per generator; G1 is stepped before G2's backward; plain backward() (no retain_graph)."""
import torch

from .LinkCycleLoss import LinkCycleLoss, _default_d_lambda, _default_g_lambda
from .BaseCompetentLoss import BaseCompetentLoss
from ...utils import passing_locals


class SplitCycleLoss(LinkCycleLoss):
    def __init__(self, netD1, netG1, netD2, netG2, optimizer_G1, optimizer_G2, optimizer_D, dims, l1_loss=None,
                 g_lambda_dict=None, d_lambda_dict=None, gan_mode="lsgan", grad_sync=None, **kwargs):
        if l1_loss is None:
            l1_loss = torch.nn.SmoothL1Loss()
        if g_lambda_dict is None:
            g_lambda_dict = _default_g_lambda()
        if d_lambda_dict is None:
            d_lambda_dict = _default_d_lambda()
        BaseCompetentLoss.__init__(self, **passing_locals(locals()))
        self.data_dict = {}

    def _retain_graph(self):
        return False  # SplitCycleLoss.py:113

    def backward_Gs(self, data_dict):
        """SplitCycleLoss.py:116-151."""
        self.set_requires_grad([self.netD1, self.netD2], False)
        self.set_requires_grad([self.netG1], True)
        self.set_requires_grad([self.netG2], False)
        self.optimizer_G1.zero_grad(set_to_none=True)
        loss_G1 = self.backward_G("B", self.netG1, self.netD1, data_dict["B"])
        self._sync(self.optimizer_G1)
        self._step(self.optimizer_G1)
        self.set_requires_grad([self.netG2], True)
        self.set_requires_grad([self.netG1], False)
        self.optimizer_G2.zero_grad(set_to_none=True)
        loss_G2 = self.backward_G("A", self.netG2, self.netD2, data_dict["A"])
        self._sync(self.optimizer_G2)
        self._step(self.optimizer_G2)
        self.set_requires_grad([self.netG1, self.netD1, self.netD2], True)
        return loss_G1, loss_G2

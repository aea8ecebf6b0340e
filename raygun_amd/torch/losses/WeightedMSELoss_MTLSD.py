"""Masked weighted MSE of the affinity/LSD heads — drop-in for raygun.torch.losses.WeightedMSELoss_MTLSD
(reference: losses/WeightedMSELoss_MTLSD.py:6-85).  One fused value+gradient kernel per head; the data-dependent
branch of _calc_loss (:15-23) is evaluated on the device, without the reference's host sync."""
from .BaseCompetentLoss import BaseCompetentLoss
from ... import ops
from ...utils import passing_locals


class WeightedMSELoss_MTLSD(BaseCompetentLoss):
    def __init__(self, **kwargs):
        super().__init__(**passing_locals(locals()))
        self.data_dict = {}

    def _calc_loss(self, pred, target, weights):
        return ops.weighted_mse(pred, target, weights)

    def forward(self, pred_lsds=None, gt_lsds=None, lsds_weights=None, pred_affs=None, gt_affs=None, affs_weights=None):
        self.data_dict.update({"pred_lsds": pred_lsds.detach(), "gt_lsds": gt_lsds.detach(), "lsds_weights": lsds_weights.detach(),
                               "pred_affs": pred_affs.detach(), "gt_affs": gt_affs.detach(), "affs_weights": affs_weights.detach()})
        lsd_loss = self._calc_loss(pred_lsds, gt_lsds, lsds_weights)
        aff_loss = self._calc_loss(pred_affs, gt_affs, affs_weights)
        self.loss_dict = {"LSDs": lsd_loss.detach(), "Affinities": aff_loss.detach()}
        return lsd_loss + aff_loss

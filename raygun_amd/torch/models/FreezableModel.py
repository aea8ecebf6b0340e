"""models/FreezableModel.py:8-32 of the reference (norm freezing after N steps, running-stat histograms)."""
import torch

from .BaseModel import BaseModel
from ..networks.utils import get_running_norm_stats, set_norm_mode
from ...utils import passing_locals


class FreezableModel(BaseModel):
    def __init__(self, freeze_norms_at=None, **kwargs):
        super().__init__(**passing_locals(locals()))

    def set_norm_modes(self, mode: str = "train"):
        for net in self.nets:
            set_norm_mode(net, mode)

    def add_log(self, writer, step):
        means, vars_ = [], []
        for net in self.nets:
            mean, var = get_running_norm_stats(net)
            if mean is not None:
                means.append(mean)
                vars_.append(var)
        if len(means) > 0:
            for tag, values in {"means": torch.cat(means), "vars": torch.cat(vars_)}.items():
                writer.add_histogram(tag, values, global_step=step)

    def update_status(self, step):
        if self.freeze_norms_at is not None and step >= self.freeze_norms_at:
            self.set_norm_modes(mode="fix_norms")

"""FreezableModel — the reference's models/FreezableModel.py:8-32 contract: `freeze_norms_at` (an iteration count, or None), `set_norm_modes`,
`add_log(writer, step)` (histograms of the running normalisation statistics, tags "means" / "vars") and `update_status(step)`."""
import torch

from .BaseModel import BaseModel
from ..networks.utils import get_running_norm_stats, set_norm_mode
from ...utils import passing_locals


class FreezableModel(BaseModel):
    def __init__(self, freeze_norms_at=None, **kwargs):
        super().__init__(**passing_locals(locals()))

    def set_norm_modes(self, mode: str = "train"):
        """Put every network of the model into `mode` ("train" | "eval" | "fix_norms": training mode with the norm layers frozen)."""
        for net in self.nets:
            set_norm_mode(net, mode)

    def _running_stats(self):
        """(all running means, all running variances) over the networks that track them, each one flat tensor — or None."""
        per_net = [get_running_norm_stats(net) for net in self.nets]
        tracked = [(m, v) for m, v in per_net if m is not None]
        if not tracked:
            return None
        return tuple(torch.cat(column) for column in zip(*tracked))

    def add_log(self, writer, step):
        stats = self._running_stats()
        if stats is None:       # no network keeps running statistics (the hot-path configs: InstanceNorm without tracking)
            return
        writer.add_histogram("means", stats[0], global_step=step)
        writer.add_histogram("vars", stats[1], global_step=step)

    def update_status(self, step):
        due = self.freeze_norms_at is not None and step >= self.freeze_norms_at
        if due:
            self.set_norm_modes(mode="fix_norms")

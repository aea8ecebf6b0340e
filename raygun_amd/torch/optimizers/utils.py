"""optimizers/utils.py:4-9 of the reference; 'Adam' resolves to the fused HIP Adam (same hyper-parameters/state keys)."""
import torch

from .FusedAdam import FusedAdam


def get_base_optimizer(optim):
    if isinstance(optim, str):
        if optim == "Adam":
            return FusedAdam
        return getattr(torch.optim, optim)
    return optim

"""Minimal train-step driver reproducing the call order of the reference's gp.torch.Train node
(torch/train/BaseTrain.py:63-75,167-178; reconstructed in SURVEY.md §3.1/§3.2), without gunpowder:

    optimizer.zero_grad(); outs = model(**inputs); loss = loss_fn(*loss_inputs); loss.backward(); optimizer.step()
    model.update_status(step); loss.update_status(step); checkpoint every save_every

Batches are dicts of arrays named after the model's forward arguments plus the loss's extra arguments
(BaseTrain.py:42-60).  Data-parallel gradient averaging (no reference counterpart) hooks in through `grad_sync`.
"""
import inspect
import os

import torch


def _names(fn):
    return [p for p in inspect.signature(fn).parameters if p not in ("self", "args", "kwargs")]


def train_step(model, loss_fn, optimizer, batch, grad_sync=None):
    """One iteration.  Returns (loss, outputs dict)."""
    if hasattr(optimizer, "zero_grad"):
        optimizer.zero_grad()
    in_names = _names(model.forward)
    outs = model(**{k: batch[k] for k in in_names if k in batch})
    outs = outs if isinstance(outs, (tuple, list)) else (outs,)
    named = dict(zip(model.output_arrays, outs))
    pool = {**batch, **named}
    loss = loss_fn(*[pool[k] for k in _names(loss_fn.forward)])
    loss.backward()  # a no-op for the CycleGAN losses, which own their backward+step (LinkCycleLoss.py:173)
    if grad_sync is not None and isinstance(optimizer, torch.optim.Optimizer):
        grad_sync([p for g in optimizer.param_groups for p in g["params"]])
    optimizer.step()
    return loss, named


class Trainer:
    def __init__(self, system):
        self.system = system
        self.iteration = getattr(system, "iteration", 0) or 0

    def train(self, iterations, batches=None):
        s = self.system
        if batches is None:
            raise RuntimeError("raygun_amd.Trainer: pass an iterable of batches (the gunpowder data pipes are out of scope)")
        last = None
        for batch in batches:
            if self.iteration >= iterations:
                break
            loss, outs = train_step(s.model, s.loss, s.optimizer, batch, getattr(s, "grad_sync", None))
            self.iteration += 1
            for obj in (s.model, s.loss):
                if hasattr(obj, "update_status"):
                    obj.update_status(self.iteration)
            if getattr(s, "save_every", None) and self.iteration % s.save_every == 0:
                self.save_checkpoint()
            last = (loss, outs)
        return last

    def save_checkpoint(self):
        s = self.system
        os.makedirs(os.path.dirname(s.checkpoint_basename) or ".", exist_ok=True)
        path = f"{s.checkpoint_basename}_checkpoint_{self.iteration}"
        torch.save({"model_state_dict": s.model.state_dict(), "iteration": self.iteration}, path)
        return path

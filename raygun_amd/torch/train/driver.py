"""Train-step driver reproducing the call order of the reference's gp.torch.Train node
(torch/train/BaseTrain.py:63-75,167-178; reconstructed in SURVEY.md 3.1/3.2), without gunpowder:

    optimizer.zero_grad(); outs = model(**inputs); loss = loss_fn(*loss_inputs); loss.backward(); optimizer.step()
    model.update_status(step); loss.update_status(step); checkpoint every save_every

Batches are dicts of arrays named after the model's forward arguments plus the loss's extra arguments
(BaseTrain.py:42-60).  Data-parallel gradient averaging (no reference counterpart) hooks in through `grad_sync`.

Checkpoints are what gunpowder's Train node writes — a torch.save'd dict with `model_state_dict` and
`optimizer_state_dict` under <checkpoint_basename>_checkpoint_<iteration> (the schema BaseSystem.load_saved_model reads,
BaseSystem.py:118-121; PARITY UNPINNED beyond those two keys: the gunpowder fork is not under /root/reference) — plus
`iteration` and, for the CycleGAN whose BaseDummyOptimizer is a parameter-less Module (state_dict() == {}), the wrapped
optimisers' and schedulers' state under `optimizers` / `schedulers`, so a resumed run continues with its Adam moments.
"""
import contextlib
import inspect
import os

import torch

from ... import ops


def _names(fn):
    return [p for p in inspect.signature(fn).parameters if p not in ("self", "args", "kwargs")]


def train_step(model, loss_fn, optimizer, batch, grad_sync=None, scaler=None):
    """One iteration.  Returns (loss, outputs dict).  `scaler` (amp.LossScaler): f16 storage with dynamic loss scaling — the
    config's "precision": "f16".  MTLSD leaves backward and step to this driver; the CycleGAN losses own both and take the scaler as
    `loss_fn.scaler` (every backward of theirs is scaled, every optimiser step checked)."""
    if hasattr(optimizer, "zero_grad"):
        optimizer.zero_grad()
    own_step = not isinstance(optimizer, torch.optim.Optimizer)   # the CycleGAN losses own backward + step (LinkCycleLoss.py:87-135)
    if hasattr(optimizer, "overlap_backward") and not own_step:
        # this step is ONE backward followed by step(): FusedAdam may update every parameter as soon as its gradient exists, under the rest of the
        # backward pass — unless the gradients are still to be checked (loss scaler) or averaged across ranks afterwards.  Opt-in
        # (RAYGUN_AMD_ADAM_OVERLAP=1): on MI355X it buys nothing, the convolution kernels hold the whole register file
        from ... import parallel
        optimizer.overlap_backward(scaler is None and (grad_sync is None or parallel.world_size() == 1)
                                   and os.environ.get("RAYGUN_AMD_ADAM_OVERLAP", "0") == "1")
    if scaler is not None and own_step:
        if not hasattr(loss_fn, "_backward"):
            raise RuntimeError("raygun_amd.train_step: loss scaling with a loss that owns backward() and step() needs raygun_amd's "
                               "BaseCompetentLoss (its _backward / _step go through the scaler)")
        loss_fn.scaler = scaler
    in_names = _names(model.forward)
    with (ops.precision("f16") if scaler is not None else contextlib.nullcontext()):
        outs = model(**{k: batch[k] for k in in_names if k in batch})
        outs = outs if isinstance(outs, (tuple, list)) else (outs,)
        named = dict(zip(model.output_arrays, outs))
        pool = {**batch, **named}
        missing = [k for k in _names(loss_fn.forward) if k not in pool]
        if missing:
            raise KeyError(f"raygun_amd.train_step: the loss takes {missing}, which neither the batch nor the model provides "
                           f"(an MTLSD config without `lsd_kwargs` yields no gt_lsds / lsds_weights); batch keys: {sorted(batch)}")
        loss = loss_fn(*[pool[k] for k in _names(loss_fn.forward)])
    (scaler.scale(loss) if scaler is not None and not own_step else loss).backward()  # a no-op for the CycleGAN losses, which own their backward+step (LinkCycleLoss.py:173)
    if grad_sync is not None and isinstance(optimizer, torch.optim.Optimizer):
        grad_sync([p for g in optimizer.param_groups for p in g["params"]])
    if scaler is not None and not own_step:
        scaler.step(optimizer)
    else:
        optimizer.step()
    return loss, named


def _named_optimizers(optimizer):
    """The real optimisers behind the system's `optimizer`: itself, or the attributes of a BaseDummyOptimizer."""
    if isinstance(optimizer, torch.optim.Optimizer):
        return {"optimizer": optimizer}
    return {k: v for k, v in vars(optimizer).items() if isinstance(v, torch.optim.Optimizer)}


class Trainer:
    def __init__(self, system):
        self.system = system
        self.iteration = getattr(system, "iteration", 0) or 0
        self.scaler = None
        prec = str(getattr(system, "precision", "bf16") or "bf16").lower()
        if prec in ("f16", "fp16", "half", "float16"):
            from ... import amp
            self.scaler = amp.LossScaler(**(getattr(system, "loss_scale_kwargs", None) or {}))
        elif prec not in ("bf16", "bfloat16"):
            raise ValueError(f"raygun_amd.Trainer: precision {prec!r} (expected \"bf16\" or \"f16\")")

    def train(self, iterations, batches=None):
        s = self.system
        if batches is None:
            batches = getattr(s, "batches", None)
        if batches is None:
            raise RuntimeError("raygun_amd.Trainer: no batch source (pass `batches=` or give the config `sources` / "
                               "\"synthetic_data\": true; the gunpowder data pipes are out of scope)")
        last = None
        # a full garbage collection over everything the imports created costs ~70 ms of host time and can starve the GPU for one
        # step: what exists now is long-lived, keep it out of the collector's way
        import gc
        gc.collect(); gc.freeze()
        for batch in batches:
            if self.iteration >= iterations:
                break
            loss, outs = train_step(s.model, s.loss, s.optimizer, batch, getattr(s, "grad_sync", None), self.scaler)
            self.iteration += 1
            for obj in (s.model, s.loss):
                if hasattr(obj, "update_status"):
                    obj.update_status(self.iteration)
            if getattr(s, "save_every", None) and self.iteration % s.save_every == 0 and self._is_writer():
                self.save_checkpoint()
            last = (loss, outs)
        return last

    def _is_writer(self):
        d = getattr(self.system, "distributed", None)
        return d is None or d[0] == 0

    def save_checkpoint(self):
        s = self.system
        os.makedirs(os.path.dirname(s.checkpoint_basename) or ".", exist_ok=True)
        path = f"{s.checkpoint_basename}_checkpoint_{self.iteration}"
        ck = {"model_state_dict": s.model.state_dict(), "optimizer_state_dict": s.optimizer.state_dict(), "iteration": self.iteration}
        if self.scaler is not None:
            ck["loss_scaler"] = self.scaler.state_dict()
        if not isinstance(s.optimizer, torch.optim.Optimizer):
            ck["optimizers"] = {k: o.state_dict() for k, o in _named_optimizers(s.optimizer).items()}
            ck["schedulers"] = {k: sch.state_dict() for k, sch in getattr(s.optimizer, "schedulers", {}).items()}
        torch.save(ck, path)
        return path

    def load_checkpoint(self, path):
        """Model weights, optimiser moments, scheduler state and the iteration counter of a checkpoint written above (a
        reference checkpoint restores what it holds: model + optimizer_state_dict)."""
        s = self.system
        ck = torch.load(path, map_location=None if torch.cuda.is_available() else torch.device("cpu"), weights_only=False)
        s.model.load_state_dict(ck["model_state_dict"] if "model_state_dict" in ck else ck)
        if isinstance(s.optimizer, torch.optim.Optimizer):
            if ck.get("optimizer_state_dict"):
                s.optimizer.load_state_dict(ck["optimizer_state_dict"])
        else:
            for k, o in _named_optimizers(s.optimizer).items():
                if k in ck.get("optimizers", {}):
                    o.load_state_dict(ck["optimizers"][k])
            for k, sch in getattr(s.optimizer, "schedulers", {}).items():
                if k in ck.get("schedulers", {}):
                    sch.load_state_dict(ck["schedulers"][k])
        if self.scaler is not None and "loss_scaler" in ck:
            self.scaler.load_state_dict(ck["loss_scaler"])
        self.iteration = int(ck.get("iteration", getattr(s, "iteration", 0) or 0))
        return ck

"""f16-storage training with dynamic loss scaling (no reference counterpart — the reference trains in fp32; this is the mode in
which the TRAINED precision meets north_star's "within 1e-3": IEEE half keeps 11 significant bits where bf16 keeps 8).

    scaler = amp.LossScaler()
    with ops.precision("f16"):                 # activations, packed weights and activation gradients are stored as half
        loss = loss_fn(*model(x))
    scaler.scale(loss).backward()              # every gradient carries the factor `scale` (half's range: 6e-8 .. 65504)
    scaler.step(optimizer)                     # inf / NaN in any gradient -> the step is skipped and the scale halved, else
                                               # FusedAdam runs with 1/scale folded into its kernel; the scale doubles every
                                               # `growth_interval` clean steps

Master weights, Adam moments and weight gradients stay fp32 (as in the bf16 mode); only the 16-bit storage type changes.  The scale
policy is torch.cuda.amp.GradScaler's (init 2^16, x2 / x0.5, interval 2000); the inf check is one read-only HIP pass over the
gradients (rg_nonfinite_check) and costs one host<-device read of a 4-byte flag per step, as GradScaler's does."""
import torch

from . import ops


class _ScaleGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, loss, scale):
        ctx.scale = scale
        return loss.view_as(loss)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.scale, None


class LossScaler:
    def __init__(self, init_scale=2.0 ** 16, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, min_scale=1.0, max_scale=2.0 ** 24):
        self.scale_value = float(init_scale)
        self.growth_factor, self.backoff_factor, self.growth_interval = float(growth_factor), float(backoff_factor), int(growth_interval)
        self.min_scale, self.max_scale = float(min_scale), float(max_scale)
        self.clean_steps = 0        # consecutive steps without inf / NaN
        self.skipped = 0            # steps dropped so far
        self._flag = None

    def scale(self, loss):
        """loss, whose backward multiplies the incoming gradient by the current scale."""
        return _ScaleGrad.apply(loss, self.scale_value)

    def found_inf(self, params):
        """One read-only pass over the gradients of `params`; True if any holds an inf / NaN.  (After a data-parallel gradient
        all-reduce every rank sees the same sums, hence the same answer.)"""
        grads = [p.grad for p in params if p.grad is not None]
        if not grads:
            return False
        if self._flag is None or self._flag.device != grads[0].device:
            self._flag = torch.zeros(1, dtype=torch.float32, device=grads[0].device)
        self._flag.zero_()
        for g in grads:
            ops.nonfinite_check_(g, self._flag)
        return bool(self._flag.item() != 0.0)

    def step(self, optimizer):
        """optimizer.step() with the gradients divided by the scale — or nothing, if a gradient overflowed — then the scale update.
        Returns True when the step was applied."""
        if not hasattr(optimizer, "grad_scale"):
            raise RuntimeError("raygun_amd.amp.LossScaler: needs raygun_amd's FusedAdam (1/scale is folded into its kernel)")
        bad = self.found_inf([p for g in optimizer.param_groups for p in g["params"]])
        if bad:
            self.skipped += 1
            self.clean_steps = 0
            self.scale_value = max(self.min_scale, self.scale_value * self.backoff_factor)
            return False
        optimizer.grad_scale = 1.0 / self.scale_value
        try:
            optimizer.step()
        finally:
            optimizer.grad_scale = 1.0
        self.clean_steps += 1
        if self.clean_steps >= self.growth_interval:
            self.clean_steps = 0
            self.scale_value = min(self.max_scale, self.scale_value * self.growth_factor)
        return True

    def state_dict(self):
        return {"scale": self.scale_value, "clean_steps": self.clean_steps, "skipped": self.skipped}

    def load_state_dict(self, d):
        self.scale_value, self.clean_steps, self.skipped = float(d["scale"]), int(d.get("clean_steps", 0)), int(d.get("skipped", 0))

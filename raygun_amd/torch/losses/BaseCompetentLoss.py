"""Common loss plumbing — reference: losses/BaseCompetentLoss.py:8-76 (validation hook kept as an optional callback)."""
import torch

from .GANLoss import GANLoss
from ... import ops
from ...utils import passing_locals


class BaseCompetentLoss(torch.nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        kwargs = passing_locals(locals())
        for key, value in kwargs.items():
            setattr(self, key, value)
        if hasattr(self, "gan_mode"):
            self.gan_loss = GANLoss(gan_mode=self.gan_mode)
        self.loss_dict = {}
        self.scaler = None   # amp.LossScaler (f16-storage training, config "precision": "f16"): set by the train driver; None = plain bf16 step

    def _backward(self, loss, **kw):
        """loss.backward(**kw) — through the loss scaler when one is set (no reference counterpart: the reference trains in fp32)."""
        (self.scaler.scale(loss) if self.scaler is not None else loss).backward(**kw)

    def _step(self, optimizer):
        """optimizer.step(), or the scaler's checked step (skipped, scale halved, when a gradient overflowed)."""
        if self.scaler is not None:
            self.scaler.step(optimizer)
        else:
            optimizer.step()

    def set_requires_grad(self, nets, requires_grad=False):
        """BaseCompetentLoss.py:20-31."""
        if not isinstance(nets, list):
            nets = [nets]
        for net in nets:
            if net is not None:
                for param in net.parameters():
                    param.requires_grad = requires_grad

    def crop(self, x, shape):
        """Centre crop (BaseCompetentLoss.py:33-42)."""
        x_target_size = x.size()[: -self.dims] + shape
        offset = tuple((a - b) // 2 for a, b in zip(x.size(), x_target_size))
        slices = tuple(slice(o, o + s) for o, s in zip(offset, x_target_size))
        return x[slices]

    def clamp_weights(self, net, min=-0.01, max=0.01):
        for module in net.model:
            if hasattr(module, "weight") and hasattr(module.weight, "data"):
                module.weight.data = module.weight.data.clamp(min, max)
                ops.invalidate_packed(module.weight)

    def l1(self, real, pred):
        """SmoothL1(beta=1, mean) runs as the fused kernel; any other user-supplied criterion is called as is."""
        fn = self.l1_loss
        if isinstance(fn, torch.nn.SmoothL1Loss) and fn.beta == 1.0 and fn.reduction == "mean" and pred.is_cuda:
            return ops.smooth_l1(pred, real)
        return fn(real, pred)

    def add_log(self, writer, step):
        """BaseCompetentLoss.py:50-70."""
        for key, loss in self.loss_dict.items():
            writer.add_scalar(key, loss, step)
        for name, data in self.data_dict.items():
            img = data[0].squeeze() if len(data.shape) > 3 else data.squeeze()
            if len(img.shape) == 3:
                img = img[img.shape[0] // 2]
            if (img.min() < 0) and (img.min() >= -1.0) and (img.max() <= 1.0):
                img = (img * 0.5) + 0.5
            writer.add_image(name, img, global_step=step, dataformats="HW")

    def update_status(self, step):
        """BaseCompetentLoss.py:72-76 — the reference launches raygun.evaluation.run_validation (out of scope here);
        a callable stored in validation_config['run'] is invoked instead when present."""
        cfg = getattr(self, "validation_config", None)
        if cfg and step % cfg["validate_every"] == 0 and callable(cfg.get("run")):
            cfg["run"](cfg, step)

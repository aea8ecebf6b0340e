"""GAN objectives — drop-in for raygun.torch.losses.GANLoss (reference: losses/GANLoss.py:5-64).
every mode is one fused value+gradient kernel (rg_loss_fwd_bwd kind 0 lsgan, 3 vanilla, 4 wgangp)."""
import torch

from ... import ops


class GANLoss(torch.nn.Module):
    def __init__(self, gan_mode, target_real_label=1.0, target_fake_label=0.0):
        super().__init__()
        self.register_buffer("real_label", torch.tensor(target_real_label))
        self.register_buffer("fake_label", torch.tensor(target_fake_label))
        self._labels = (float(target_real_label), float(target_fake_label))
        self.gan_mode = gan_mode
        if gan_mode == "lsgan":
            self.loss = torch.nn.MSELoss()
        elif gan_mode == "vanilla":
            self.loss = torch.nn.BCEWithLogitsLoss()
        elif gan_mode in ["wgangp"]:
            self.loss = None
        else:
            raise NotImplementedError("gan mode %s not implemented" % gan_mode)

    def get_target_tensor(self, prediction, target_is_real):
        target_tensor = self.real_label if target_is_real else self.fake_label
        return target_tensor.to(prediction.device).expand_as(prediction)

    def __call__(self, prediction, target_is_real):
        if self.gan_mode == "lsgan":
            return ops.mse_const(prediction, self._labels[0] if target_is_real else self._labels[1])
        if not prediction.is_cuda:
            # plain torch modules on host tensors: only the CPU suite's host-logic tests of the loss classes come this way (vanilla /
            # wgangp, as in the reference); lsgan above has no such branch — the HIP ops raise on CPU tensors
            if self.gan_mode == "vanilla":
                return self.loss(prediction, self.get_target_tensor(prediction, target_is_real))
            return -prediction.mean() if target_is_real else prediction.mean()
        if self.gan_mode == "vanilla":
            return ops.bce_logits_const(prediction, self._labels[0] if target_is_real else self._labels[1])
        return ops.mean_signed(prediction, -1.0 if target_is_real else 1.0)

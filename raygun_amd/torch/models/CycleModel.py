"""CycleGAN forward graph — drop-in for raygun.torch.models.CycleModel (reference: models/CycleModel.py:8-89).

The four generator passes run inside one raygun_amd scope, so fake_B / fake_A stay channels-last bf16 between
the two generators; the four outputs are returned as fp32 NCDHW like the reference's."""
import torch
import torch.nn.functional as F

from .FreezableModel import FreezableModel
from ... import hipnn, ops
from ...utils import passing_locals


class CycleModel(FreezableModel):
    def __init__(self, netG1, netG2, scale_factor_A=None, scale_factor_B=None, split=False, **kwargs):
        output_arrays = ["fake_B", "cycled_B", "fake_A", "cycled_A"]
        nets = [netG1, netG2]
        super().__init__(**passing_locals(locals()))
        self.cycle = True
        self.crop_pad = None

    def sampling_bottleneck(self, array, scale_factor):
        """CycleModel.py:24-30 — trilinear down/up (torch op; off in every BASELINE config)."""
        array = ops.leave(array) if ops.is_cl(array) else array
        size = array.shape[-len(scale_factor):]
        mode = {2: "bilinear", 3: "trilinear"}[len(size)]
        down = F.interpolate(array, scale_factor=scale_factor, mode=mode, align_corners=True)
        return F.interpolate(down, size=size, mode=mode, align_corners=True)

    def set_crop_pad(self, crop_pad, ndims):
        self.crop_pad = (slice(None, None, None),) * 2 + (slice(crop_pad, -crop_pad),) * ndims

    def _one_direction(self, real, net_a, net_b, scale_factor):
        fake = net_a(real)
        if self.crop_pad is not None:
            fake = fake[self.crop_pad]
        if scale_factor:
            fake = self.sampling_bottleneck(fake, scale_factor)
        cycled = None
        if self.cycle:
            cycled = net_b(fake.detach() if self.split else fake)  # CycleModel.py:52-57
            if self.crop_pad is not None:
                cycled = cycled[self.crop_pad]
        return fake, cycled

    @hipnn.scoped
    def forward(self, real_A=None, real_B=None):
        assert real_A is not None or real_B is not None, "Must have some real input to generate outputs)"
        fake_B = cycled_A = fake_A = cycled_B = None
        if real_A is not None:
            fake_B, cycled_A = self._one_direction(real_A, self.netG1, self.netG2, self.scale_factor_B)
        if real_B is not None:
            fake_A, cycled_B = self._one_direction(real_B, self.netG2, self.netG1, self.scale_factor_A)
        return fake_B, cycled_B, fake_A, cycled_A

"""Residual U-Net — drop-in for raygun.torch.networks.ResidualUNet (reference: src/raygun/torch/networks/ResidualUNet.py:364-701).

The reference file repeats UNet.py and changes three things, which is all this module adds to raygun_amd's UNet:
  * ConvPass takes `final` (:22): the block that produces the network output omits its last activation (:94-99,123-126);
    the level-0 right ConvPass of every head is built that way (:636);
  * the network keeps `padding_type` and an instance of the activation (:499-507);
  * forward (:689-701) adds the (centre-cropped, for valid padding) INPUT to every head's output and applies the activation:
    y_i = act(crop(x) + y_i) — x is broadcast over the channels when input_nc == 1.
Same constructor signature, attributes and state_dict keys as the reference class; every layer runs on the gfx950 kernels
(the residual ConvPass's grouped 1x1 map as a dense 1x1 convolution, UNet.ConvPass._init_map)."""
import torch

from ... import hipnn, ops
from .UNet import ConvDownsample, ConvPass, MaxDownsample, UNet, Upsample, _activation_instance  # noqa: F401  (same names as the reference module)


class ResidualUNet(UNet):
    _final_output_block = True

    def __init__(self, input_nc, ngf, fmap_inc_factor, downsample_factors, kernel_size_down=None, kernel_size_up=None,
                 activation="ReLU", output_nc=None, num_heads=1, constant_upsample=False, downsample_method="max",
                 padding_type="valid", residual=False, norm_layer=None, add_noise=False):
        super().__init__(input_nc, ngf, fmap_inc_factor, downsample_factors, kernel_size_down, kernel_size_up, activation,
                         output_nc, num_heads, constant_upsample, downsample_method, padding_type, residual, norm_layer,
                         add_noise)
        self.padding_type = padding_type
        self.activation = _activation_instance(activation)

    def crop(self, x, shape):
        return ops.center_crop(x, tuple(shape))

    @hipnn.scoped
    def forward(self, x):
        x = ops.enter(x)
        y = self.rec_forward(self.num_levels - 1, x)
        if self.padding_type.lower() == "valid":
            x = self.crop(x, tuple(y[0].shape[-3:]))
        a = hipnn.act_of(self.activation)
        for i in range(self.num_heads):
            xi = x
            if xi.shape[1] != y[i].shape[1]:      # torch broadcasting of a 1-channel input over the output channels
                xi = ops.enter(ops.leave(xi).expand(-1, y[i].shape[1], -1, -1, -1).contiguous())
            s = ops.add(xi, y[i])
            y[i] = ops.activation(s, a[0], a[1]) if a is not None else hipnn.passthrough(self.activation, s)
        if self.num_heads == 1:
            return y[0]
        return y

"""Johnson-style ResNet generator on the gfx950 kernels — drop-in for raygun.torch.networks.ResNet
(reference: src/raygun/torch/networks/ResNet.py:226-473; layer order, `.model` indices and state_dict keys kept)."""
import functools

import torch
import torch.nn as nn

from ... import hipnn, ops
from .utils import NoiseBlock, ParameterizedNoiseBlock


def _padder(padding_type, width, nd=3):
    """ResNet.py:264-278 / :388-401 (3-D), :44-58 / :171-184 (2-D) -> (pad modules, conv padding)."""
    pt = padding_type.lower()
    if pt in ("reflect", "same"):
        return [hipnn.layers(nd)[3](width)], 0
    if pt == "replicate":
        return [{3: hipnn.ReplicationPad3d, 2: hipnn.ReplicationPad2d}[nd](width)], 0
    if pt == "zeros":
        return [], width
    if pt == "valid":
        return [], "valid"
    raise NotImplementedError("padding [%s] is not implemented" % padding_type)


class ResnetBlock3D(nn.Module):
    """ResNet.py:347-443: [pad, conv3, norm, act, (dropout), pad, conv3, norm] + skip (cropped when valid)."""
    _raygun_amd = True
    _nd = 3

    def __init__(self, dim, padding_type, norm_layer, use_dropout, use_bias, activation=torch.nn.ReLU):
        super().__init__()
        self.conv_block = self.build_conv_block(dim, padding_type, norm_layer, use_dropout, use_bias, activation)
        self.padding_type = padding_type

    def build_conv_block(self, dim, padding_type, norm_layer, use_dropout, use_bias, activation=torch.nn.ReLU):
        norm_layer, activation = hipnn.hip_layer(norm_layer), hipnn.hip_layer(activation)
        Conv = hipnn.layers(self._nd)[0]
        _, p = _padder(padding_type, 1, self._nd)
        conv_block = list(_padder(padding_type, 1, self._nd)[0])
        conv_block += [Conv(dim, dim, kernel_size=3, padding=p, bias=use_bias), norm_layer(dim), activation()]
        if use_dropout:
            conv_block += [nn.Dropout(0.2)]
        conv_block += list(_padder(padding_type, 1, self._nd)[0])
        conv_block += [Conv(dim, dim, kernel_size=3, padding=p, bias=use_bias), norm_layer(dim)]
        return hipnn.FusedSequential(*conv_block)

    def crop(self, x, shape):
        return ops.center_crop(x, tuple(shape))

    @hipnn.scoped
    def forward(self, x):
        x = ops.enter(x)
        if self.padding_type == "valid":
            res = self.conv_block.run(x)
            return ops.add(self.crop(x, res.size()[-3:]), res)
        # skip add fused into the last InstanceNorm's apply kernel
        return self.conv_block.run(x, residual=x)


class ResnetBlock2D(ResnetBlock3D):
    """ResNet.py:130-223 — the same block on Conv2d/InstanceNorm2d twins (tensors travel as (N,C,1,H,W))."""
    _nd = 2


class ResnetGenerator3D(nn.Module):
    """ResNet.py:226-344."""
    _raygun_amd = True
    _raygun_tail_ok = True  # forward is exactly self.model(input): RaygunSequential may split off the last conv

    def __init__(self, input_nc=1, output_nc=None, ngf=64, norm_layer=torch.nn.InstanceNorm3d, use_dropout=False,
                 n_blocks=6, padding_type="reflect", activation=torch.nn.ReLU, add_noise=False, n_downsampling=2):
        assert n_blocks >= 0
        nn.Module.__init__(self)
        self._build(3, input_nc, output_nc, ngf, norm_layer, use_dropout, n_blocks, padding_type, activation, add_noise, n_downsampling)

    def _build(self, nd, input_nc, output_nc, ngf, norm_layer, use_dropout, n_blocks, padding_type, activation, add_noise,
               n_downsampling):
        Conv, ConvT = hipnn.layers(nd)[:2]
        Block = ResnetBlock3D if nd == 3 else ResnetBlock2D
        use_bias = hipnn.is_instance_norm_factory(norm_layer, nd)
        norm_layer, activation = hipnn.hip_layer(norm_layer), hipnn.hip_layer(activation)
        if output_nc is None or nd == 2:  # the 2-D generator always ends in Conv2d(ngf, ngf, 7) (ResNet.py:118)
            output_nc = ngf
        _, p = _padder(padding_type, 3, nd)
        updown_p = 0 if padding_type.lower() == "valid" else 1

        model = list(_padder(padding_type, 3, nd)[0])
        model += [Conv(input_nc, ngf, kernel_size=7, padding=p, bias=use_bias), norm_layer(ngf), activation()]
        for i in range(n_downsampling):
            mult = 2 ** i
            model += [Conv(ngf * mult, ngf * mult * 2, kernel_size=3, stride=2, padding=updown_p, bias=use_bias),
                      norm_layer(ngf * mult * 2), activation()]
        mult = 2 ** n_downsampling
        for i in range(n_blocks):
            model += [Block(ngf * mult, padding_type=padding_type.lower(), norm_layer=norm_layer,
                            use_dropout=use_dropout, use_bias=use_bias, activation=activation)]
        if add_noise == "param":
            model += [ParameterizedNoiseBlock()]
        elif add_noise:
            model += [NoiseBlock()]
        for i in range(n_downsampling):
            mult = 2 ** (n_downsampling - i)
            model += [ConvT(ngf * mult + (i == 0 and (add_noise is not False)), int(ngf * mult / 2),
                            kernel_size=3, stride=2, padding=updown_p, output_padding=updown_p, bias=use_bias),
                      norm_layer(int(ngf * mult / 2)), activation()]
        model += list(_padder(padding_type, 3, nd)[0])
        model += [Conv(ngf, output_nc, kernel_size=7, padding=p)]
        self.model = hipnn.FusedSequential(*model)

    @hipnn.scoped
    def forward(self, input):
        return self.model.run(ops.enter(input))


class ResnetGenerator2D(ResnetGenerator3D):
    """ResNet.py:6-127."""

    def __init__(self, input_nc=1, output_nc=None, ngf=64, norm_layer=torch.nn.InstanceNorm2d, use_dropout=False,
                 n_blocks=6, padding_type="reflect", activation=torch.nn.ReLU, add_noise=False, n_downsampling=2):
        assert n_blocks >= 0
        nn.Module.__init__(self)
        self._build(2, input_nc, output_nc, ngf, norm_layer, use_dropout, n_blocks, padding_type, activation, add_noise, n_downsampling)


class ResNet(ResnetGenerator2D, ResnetGenerator3D):
    """ResNet.py:446-473 — dispatch on ndims."""

    def __init__(self, ndims, **kwargs):
        if ndims == 2:
            ResnetGenerator2D.__init__(self, **kwargs)
        elif ndims == 3:
            ResnetGenerator3D.__init__(self, **kwargs)
        else:
            raise ValueError(ndims, "Only 2D or 3D currently implemented. Feel free to contribute more!")

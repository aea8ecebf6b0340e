"""PatchGAN discriminator on the gfx950 kernels — drop-in for raygun.torch.networks.NLayerDiscriminator
(reference: src/raygun/torch/networks/NLayerDiscriminator.py:104-202)."""
import torch
import torch.nn as nn

from ... import hipnn, ops


class NLayerDiscriminator3D(nn.Module):
    """NLayerDiscriminator.py:104-180."""
    _raygun_amd = True

    def __init__(self, input_nc=1, ngf=64, n_layers=3, norm_layer=torch.nn.BatchNorm3d, kw=4, downsampling_kw=None):
        nn.Module.__init__(self)
        self._build(3, input_nc, ngf, n_layers, norm_layer, kw, downsampling_kw)

    def _build(self, nd, input_nc, ngf, n_layers, norm_layer, kw, downsampling_kw):
        Conv = hipnn.layers(nd)[0]
        use_bias = hipnn.is_instance_norm_factory(norm_layer, nd)
        norm_layer = hipnn.hip_layer(norm_layer)
        if downsampling_kw is None:
            downsampling_kw = kw
        padw, ds_kw = 1, downsampling_kw
        sequence = [Conv(input_nc, ngf, kernel_size=ds_kw, stride=2, padding=padw), hipnn.LeakyReLU(0.2, True)]
        nf_mult = 1
        for n in range(1, n_layers):
            nf_mult_prev, nf_mult = nf_mult, min(2 ** n, 8)
            sequence += [Conv(ngf * nf_mult_prev, ngf * nf_mult, kernel_size=ds_kw, stride=2, padding=padw, bias=use_bias),
                         norm_layer(ngf * nf_mult), hipnn.LeakyReLU(0.2, True)]
        nf_mult_prev, nf_mult = nf_mult, min(2 ** n_layers, 8)
        sequence += [Conv(ngf * nf_mult_prev, ngf * nf_mult, kernel_size=kw, stride=1, padding=padw, bias=use_bias),
                     norm_layer(ngf * nf_mult), hipnn.LeakyReLU(0.2, True)]
        sequence += [Conv(ngf * nf_mult, 1, kernel_size=kw, stride=1, padding=padw)]
        self.model = hipnn.FusedSequential(*sequence)

    @hipnn.scoped
    def forward(self, input):
        return self.model.run(ops.enter(input))


class NLayerDiscriminator2D(NLayerDiscriminator3D):
    """NLayerDiscriminator.py:5-101 (incl. the FOV property, :79-96)."""

    def __init__(self, input_nc=1, ngf=64, n_layers=3, norm_layer=torch.nn.BatchNorm2d, kw=4, downsampling_kw=None):
        nn.Module.__init__(self)
        self._build(2, input_nc, ngf, n_layers, norm_layer, kw, downsampling_kw)

    @property
    def FOV(self):
        k, s = [], []
        for layer in self.model:
            if hasattr(layer, "kernel_size"):
                k += [layer.kernel_size[-1]]
                s += [layer.stride[-1]]
        r = 1
        for l in range(len(k) - 1, 0, -1):
            r = s[l] * r + (k[l] - s[l])
        return r


class NLayerDiscriminator(NLayerDiscriminator2D, NLayerDiscriminator3D):
    """NLayerDiscriminator.py:183-202."""

    def __init__(self, ndims, **kwargs):
        if ndims == 2:
            NLayerDiscriminator2D.__init__(self, **kwargs)
        elif ndims == 3:
            NLayerDiscriminator3D.__init__(self, **kwargs)
        else:
            raise ValueError(ndims, "Only 2D or 3D currently implemented. Feel free to contribute more!")

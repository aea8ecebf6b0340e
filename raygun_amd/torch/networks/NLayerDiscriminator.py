"""PatchGAN discriminator on the gfx950 kernels — drop-in for raygun.torch.networks.NLayerDiscriminator
(reference: src/raygun/torch/networks/NLayerDiscriminator.py:104-202)."""
import torch
import torch.nn as nn

from ... import hipnn, ops


class NLayerDiscriminator3D(nn.Module):
    """NLayerDiscriminator.py:104-180."""
    _raygun_amd = True

    def __init__(self, input_nc=1, ngf=64, n_layers=3, norm_layer=torch.nn.BatchNorm3d, kw=4, downsampling_kw=None):
        super().__init__()
        use_bias = hipnn.is_instance_norm_factory(norm_layer, 3)
        norm_layer = hipnn.hip_layer(norm_layer)
        if downsampling_kw is None:
            downsampling_kw = kw
        padw, ds_kw = 1, downsampling_kw
        sequence = [hipnn.Conv3d(input_nc, ngf, kernel_size=ds_kw, stride=2, padding=padw), hipnn.LeakyReLU(0.2, True)]
        nf_mult = 1
        for n in range(1, n_layers):
            nf_mult_prev, nf_mult = nf_mult, min(2 ** n, 8)
            sequence += [hipnn.Conv3d(ngf * nf_mult_prev, ngf * nf_mult, kernel_size=ds_kw, stride=2, padding=padw, bias=use_bias),
                         norm_layer(ngf * nf_mult), hipnn.LeakyReLU(0.2, True)]
        nf_mult_prev, nf_mult = nf_mult, min(2 ** n_layers, 8)
        sequence += [hipnn.Conv3d(ngf * nf_mult_prev, ngf * nf_mult, kernel_size=kw, stride=1, padding=padw, bias=use_bias),
                     norm_layer(ngf * nf_mult), hipnn.LeakyReLU(0.2, True)]
        sequence += [hipnn.Conv3d(ngf * nf_mult, 1, kernel_size=kw, stride=1, padding=padw)]
        self.model = hipnn.FusedSequential(*sequence)

    @hipnn.scoped
    def forward(self, input):
        return self.model.run(ops.enter(input))


class NLayerDiscriminator(NLayerDiscriminator3D):
    """NLayerDiscriminator.py:183-202."""

    def __init__(self, ndims, **kwargs):
        if ndims == 3:
            NLayerDiscriminator3D.__init__(self, **kwargs)
        elif ndims == 2:
            raise NotImplementedError("raygun_amd: the 2-D HIP path is not built yet (SURVEY.md §8f rank 1)")
        else:
            raise ValueError(ndims, "Only 2D or 3D currently implemented. Feel free to contribute more!")

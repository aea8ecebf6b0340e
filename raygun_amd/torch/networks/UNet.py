"""3-D U-Net on the gfx950 kernels — drop-in for raygun.torch.networks.UNet
(reference: src/raygun/torch/networks/UNet.py; constructor signatures, attributes, state_dict keys
and error behaviour preserved; citations per class).

Differences that are deliberate:
  * forward runs hand-written HIP (bf16 storage, fp32 accumulate) and needs a CUDA/HIP tensor;
  * nearest upsample + crop_to_factor + centre-crop of the skip + concat is one kernel
    (ops.upcat) instead of four tensor ops (UNet.py:338-351);
  * 2-D / 4-D kernels are not built on this path yet: kernel rank 3 only.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from ... import hipnn, ops
from .utils import NoiseBlock, ParameterizedNoiseBlock


def _activation_instance(activation):
    """UNet.py:41-47: str -> torch.nn name, callable -> instance, None -> Identity (HIP twin where one exists)."""
    if activation is None:
        return nn.Identity()
    if isinstance(activation, str):
        return hipnn.hip_layer(activation)()
    return hipnn.hip_layer(activation)()


def _conv_cls(dims):
    if dims == 3:
        return hipnn.Conv3d
    if dims == 2:
        return hipnn.Conv2d  # the 3-D kernels on a (N,C,1,H,W) view
    raise RuntimeError("%dD convolution not implemented" % dims)  # UNet.py:84-85


class ConvPass(nn.Module):
    """UNet.py:12-117.  conv_pass is an nn.Sequential subclass: conv_pass[i] exposes the same layers."""
    _raygun_amd = True

    def __init__(self, input_nc, output_nc, kernel_sizes, activation, padding="valid", residual=False,
                 padding_mode="reflect", norm_layer=None, final=False):
        """`final` (ResidualUNet.py:22,94-99,123-126): the block is the network's output — its last activation is omitted."""
        super().__init__()
        self.activation = _activation_instance(activation)
        self.residual = residual
        self.padding = padding
        self.final = final
        norm_layer = hipnn.hip_layer(norm_layer)
        layers = []
        for i, kernel_size in enumerate(kernel_sizes):
            self.dims = len(kernel_size)
            conv = _conv_cls(self.dims)
            layers.append(conv(input_nc, output_nc, tuple(int(k) for k in kernel_size), padding=padding, padding_mode=padding_mode))
            if residual and i == 0:
                groups = input_nc if input_nc < output_nc else output_nc
                # grouped 1x1 map of the residual variant (UNet.py:70-83): a torch module for its parameter / state_dict key;
                # forward() evaluates it as a dense 1x1 convolution on the HIP kernel (block-sparse weight, see _init_map)
                self.x_init_map = getattr(nn, f"Conv{self.dims}d")(input_nc, output_nc, tuple(np.ones(self.dims, dtype=int)), padding=padding,
                                            padding_mode=padding_mode, bias=False, groups=groups)
            if norm_layer is not None:
                layers.append(norm_layer(output_nc))
            if not ((residual and i == (len(kernel_sizes) - 1)) or (final and i == (len(kernel_sizes) - 1))):
                layers.append(self.activation)
            input_nc = output_nc
        self.conv_pass = hipnn.FusedSequential(*layers)

    def crop(self, x, shape):
        return ops.center_crop(x, tuple(shape))

    @hipnn.scoped
    def forward(self, x):
        if not self.residual:
            return self.conv_pass(x)
        res = self.conv_pass(x)
        init_x = self._init_map(x)
        if self.padding.lower() == "valid":
            init_x = self.crop(init_x, res.size()[-3:] if res.dim() == 5 else res.size()[-self.dims:])
        y = ops.add(init_x, res)
        if self.final:
            return y
        a = hipnn.act_of(self.activation)
        return ops.activation(y, a[0], a[1]) if a is not None else hipnn.passthrough(self.activation, y)

    def _init_map(self, x):
        """x_init_map(x): the grouped, bias-free 1x1 convolution as a DENSE 1x1 convolution whose (Cout, Cin) weight is zero outside
        the groups — output channel o of group g = o // (Cout / groups) sees input channels [g * Cin/groups, (g+1) * Cin/groups).
        The scatter is differentiable, so the grouped parameter gets its gradient; the arithmetic runs on the igemm kernels."""
        m = self.x_init_map
        if not x.is_cuda or any(k != 1 for k in m.kernel_size):
            return ops.enter(hipnn.passthrough(m, x))
        cout, ipg = m.weight.shape[0], m.weight.shape[1]
        cin, opg = ipg * m.groups, cout // m.groups
        cols = (torch.arange(cout, device=x.device) // opg)[:, None] * ipg + torch.arange(ipg, device=x.device)[None, :]
        dense = torch.zeros(cout, cin, dtype=m.weight.dtype, device=x.device).scatter(1, cols, m.weight.reshape(cout, ipg))
        return ops.conv3d(ops.enter(x), dense.reshape(cout, cin, 1, 1, 1), None)


class ConvDownsample(nn.Module):
    """UNet.py:120-188 — strided valid conv + norm + activation."""
    _raygun_amd = True

    def __init__(self, input_nc, output_nc, kernel_sizes, downsample_factor, activation, padding="valid",
                 padding_mode="reflect", norm_layer=None):
        super().__init__()
        self.activation = _activation_instance(activation)
        self.padding = padding
        self.dims = len(kernel_sizes)
        norm_layer = hipnn.hip_layer(norm_layer)
        layers = [_conv_cls(self.dims)(input_nc, output_nc, tuple(int(k) for k in kernel_sizes),
                                       stride=tuple(int(f) for f in downsample_factor), padding="valid", padding_mode=padding_mode)]
        if norm_layer is not None:
            layers.append(norm_layer(output_nc))
        layers.append(self.activation)
        self.conv_pass = hipnn.FusedSequential(*layers)

    @hipnn.scoped
    def forward(self, x):
        return self.conv_pass(x)


class MaxDownsample(nn.Module):
    """UNet.py:191-238 — flexible=True floors sizes that the factor does not divide."""
    _raygun_amd = True

    def __init__(self, downsample_factor, flexible=True):
        super().__init__()
        self.dims = len(downsample_factor)
        self.downsample_factor = downsample_factor
        self.flexible = flexible
        if self.dims == 2:
            f = tuple(int(v) for v in downsample_factor)
            self.down = hipnn.MaxPool2d(f, stride=f)
        else:
            f = tuple(int(v) for v in downsample_factor)[-3:]
            self.down = hipnn.MaxPool3d(f, stride=f)

    @hipnn.scoped
    def forward(self, x):
        if self.flexible:
            try:
                return self.down(x)
            except Exception:
                self.check_mismatch(x.size())
                raise
        self.check_mismatch(x.size())
        return self.down(x)

    def check_mismatch(self, size):
        for d in range(1, self.dims + 1):
            if size[-d] % self.downsample_factor[-d] != 0:
                raise RuntimeError("Can not downsample shape %s with factor %s, mismatch in spatial dimension %d"
                                   % (size, self.downsample_factor, self.dims - d))
        return


class Upsample(nn.Module):
    """UNet.py:241-351."""
    _raygun_amd = True

    def __init__(self, scale_factor, mode=None, input_nc=None, output_nc=None, crop_factor=None,
                 next_conv_kernel_sizes=None):
        super().__init__()
        if crop_factor is not None:
            assert next_conv_kernel_sizes is not None, "crop_factor and next_conv_kernel_sizes have to be given together"
        self.crop_factor = crop_factor
        self.next_conv_kernel_sizes = next_conv_kernel_sizes
        self.dims = len(scale_factor)
        self.scale_factor = tuple(int(s) for s in scale_factor)
        self.mode = mode
        if self.dims not in (2, 3):
            raise NotImplementedError("raygun_amd.Upsample: 2-D and 3-D only")
        if mode == "transposed_conv":
            self.up = hipnn.layers(self.dims)[1](input_nc, output_nc, kernel_size=self.scale_factor, stride=self.scale_factor)
        else:
            if mode != "nearest":
                raise NotImplementedError(f"raygun_amd.Upsample: mode {mode!r} is not built (nearest / transposed_conv)")
            self.up = nn.Upsample(scale_factor=self.scale_factor, mode=mode)  # parameter-free; fused in forward

    def target_shape(self, spatial_shape, factor, kernel_sizes):
        """crop_to_factor (UNet.py:275-325): largest n with n*f + c <= s, c = sum(ks-1)."""
        convolution_crop = tuple(sum(ks[d] - 1 for ks in kernel_sizes) for d in range(len(factor)))
        ns = (int(math.floor(float(s - c) / f)) for s, c, f in zip(spatial_shape, convolution_crop, factor))
        target = tuple(n * f + c for n, c, f in zip(ns, convolution_crop, factor))
        if target != tuple(spatial_shape):
            assert all(t > c for t, c in zip(target, convolution_crop)), (
                "Feature map with shape %s is too small to ensure translation equivariance with factor %s and following "
                "convolutions %s" % (tuple(spatial_shape), factor, kernel_sizes))
        return target

    def crop(self, x, shape):
        return ops.center_crop(x, tuple(shape))

    @hipnn.scoped
    def forward(self, f_left, g_out):
        f_left, g_out = ops.enter(f_left), ops.enter(g_out)  # 2-D tensors travel as (N,C,1,H,W)
        lift = (1,) * (3 - self.dims)
        if self.mode == "transposed_conv":
            g_up, factor = self.up(g_out), (1, 1, 1)
            up_shape = tuple(g_up.shape[-3:])
        else:
            g_up, factor = g_out, lift + self.scale_factor
            up_shape = tuple(a * b for a, b in zip(g_out.shape[-3:], factor))
        target = up_shape
        if self.crop_factor is not None:
            target = self.target_shape(up_shape, lift + tuple(self.crop_factor),
                                       [lift + tuple(ks) for ks in self.next_conv_kernel_sizes])
        return ops.upcat(f_left, g_up, factor, target)


class UNet(nn.Module):
    """UNet.py:354-666 — same constructor, attributes (.ndims/.num_levels/.num_heads/.input_nc/.output_nc),
    module tree (l_conv / l_down / r_up / r_conv) and recursion."""
    _raygun_amd = True

    def __init__(self, input_nc, ngf, fmap_inc_factor, downsample_factors, kernel_size_down=None, kernel_size_up=None,
                 activation="ReLU", output_nc=None, num_heads=1, constant_upsample=False, downsample_method="max",
                 padding_type="valid", residual=False, norm_layer=None, add_noise=False):
        super().__init__()
        self.ndims = len(downsample_factors[0])
        self.num_levels = len(downsample_factors) + 1
        self.num_heads = num_heads
        self.input_nc = input_nc
        self.output_nc = output_nc if output_nc else ngf
        self.residual = residual
        if add_noise == "param":
            self.noise_layer = ParameterizedNoiseBlock()
        elif add_noise:
            self.noise_layer = NoiseBlock()
        else:
            self.noise_layer = None
        if kernel_size_down is None:
            kernel_size_down = [[(3,) * self.ndims, (3,) * self.ndims]] * self.num_levels
        if kernel_size_up is None:
            kernel_size_up = [[(3,) * self.ndims, (3,) * self.ndims]] * (self.num_levels - 1)

        # UNet.py:506-522: crop factors for translation equivariance (valid padding only)
        crop_factors = []
        factor_product = None
        for factor in downsample_factors[::-1]:
            if padding_type.lower() == "valid":
                factor_product = list(factor) if factor_product is None else [f * ff for f, ff in zip(factor, factor_product)]
            elif padding_type.lower() == "same":
                factor_product = None
            else:
                raise RuntimeError(f"Invalid padding_type option: {padding_type}")
            crop_factors.append(factor_product)
        crop_factors = crop_factors[::-1]

        is_max = downsample_method.lower() == "max"
        self.l_conv = nn.ModuleList([
            ConvPass(input_nc if level == 0 else ngf * fmap_inc_factor ** (level - int(is_max)),
                     ngf * fmap_inc_factor ** level, kernel_size_down[level], activation=activation,
                     padding=padding_type, residual=self.residual, norm_layer=norm_layer)
            for level in range(self.num_levels)])
        self.dims = self.l_conv[0].dims

        if is_max:
            self.l_down = nn.ModuleList([MaxDownsample(downsample_factors[level]) for level in range(self.num_levels - 1)])
        elif downsample_method.lower() == "convolve":
            self.l_down = nn.ModuleList([
                ConvDownsample(ngf * fmap_inc_factor ** level, ngf * fmap_inc_factor ** (level + 1),
                               kernel_size_down[level][0], downsample_factors[level], activation=activation,
                               padding=padding_type, norm_layer=norm_layer)
                for level in range(self.num_levels - 1)])
        else:
            raise RuntimeError(f'Unknown downsampling method {downsample_method}. Use "max" or "convolve" instead.')

        self.r_up = nn.ModuleList([nn.ModuleList([
            Upsample(downsample_factors[level], mode="nearest" if constant_upsample else "transposed_conv",
                     input_nc=ngf * fmap_inc_factor ** (level + 1) + (level == 1 and (add_noise is not False)),
                     output_nc=ngf * fmap_inc_factor ** (level + 1), crop_factor=crop_factors[level],
                     next_conv_kernel_sizes=kernel_size_up[level])
            for level in range(self.num_levels - 1)]) for _ in range(num_heads)])

        self.r_conv = nn.ModuleList([nn.ModuleList([
            ConvPass(ngf * fmap_inc_factor ** level + ngf * fmap_inc_factor ** (level + 1),
                     ngf * fmap_inc_factor ** level if output_nc is None or level != 0 else output_nc,
                     kernel_size_up[level], activation=activation, padding=padding_type, residual=self.residual,
                     norm_layer=norm_layer, **({"final": level == 0} if self._final_output_block else {}))
            for level in range(self.num_levels - 1)]) for _ in range(num_heads)])

    def rec_forward(self, level, f_in):
        i = self.num_levels - level - 1
        f_left = self.l_conv[i](f_in)
        if level == 0:
            if self.noise_layer is not None:
                f_left = ops.enter(hipnn.passthrough(self.noise_layer, f_left))
            fs_out = [f_left] * self.num_heads
        else:
            if f_left.is_cuda and f_left.requires_grad and ops.is_cl(f_left):
                f_left, f_down = ops.fork(f_left)  # two consumers: their gradients are summed by one HIP pass
            else:
                f_down = f_left
            g_in = self.l_down[i](f_down)
            gs_out = self.rec_forward(level - 1, g_in)
            fs_out = []
            for h in range(self.num_heads):
                fused = self._subpixel_up(self.r_up[h][i], self.r_conv[h][i], f_left, gs_out[h])
                fs_out.append(fused if fused is not None else self.r_conv[h][i](self.r_up[h][i](f_left, gs_out[h])))
        return fs_out

    @staticmethod
    def _subpixel_up(up, rc, f_left, g_out):
        """Nearest x2 upsample + crop + concat + the first 3x3x3 valid conv (+ activation) of the right ConvPass as
        ops.upcat_conv3d (27 -> 8 taps on the upsampled channels), then the rest of the ConvPass.  None when the pattern does
        not apply (other modes / factors / kernels, norm layers, residual passes, few upsampled channels) or with
        RAYGUN_AMD_SUBPIXEL_UP=0."""
        if os.environ.get("RAYGUN_AMD_SUBPIXEL_UP", "1") == "0" or not f_left.is_cuda:
            return None
        if up.mode != "nearest" or up.dims != 3 or up.scale_factor != (2, 2, 2) or rc.residual or str(rc.padding).lower() != "valid":
            return None
        mods = list(rc.conv_pass)
        if len(mods) < 2 or not isinstance(mods[0], hipnn.Conv3d) or hipnn.act_of(mods[1]) is None:
            return None
        conv = mods[0]
        if (tuple(conv.kernel_size) != (3, 3, 3) or tuple(conv.stride) != (1, 1, 1) or conv.groups != 1 or any(d != 1 for d in conv.dilation)
                or conv.out_channels % 8 or g_out.shape[1] < 64 or g_out.dim() != 5):
            return None
        f_left, g_out = ops.enter(f_left), ops.enter(g_out)
        up_shape = tuple(2 * v for v in g_out.shape[-3:])
        target = up_shape
        if up.crop_factor is not None:
            target = up.target_shape(up_shape, tuple(up.crop_factor), [tuple(ks) for ks in up.next_conv_kernel_sizes])
        if any(t < 3 for t in target) or any(a < t for a, t in zip(f_left.shape[-3:], target)):
            return None
        a = hipnn.act_of(mods[1])
        with hipnn.scope():
            y = ops.upcat_conv3d(f_left, g_out, target, conv.weight, conv.bias, a[0], a[1])
            return rc.conv_pass.run(y, start=2)

    _final_output_block = False   # ResidualUNet: the level-0 right ConvPass is built with final=True (ResidualUNet.py:636)

    @hipnn.scoped
    def forward(self, x):
        y = self.rec_forward(self.num_levels - 1, ops.enter(x))
        if self.num_heads == 1:
            return y[0]
        return y

"""ORACLE — CPU restatement (plain PyTorch fp32) of raygun's 3D/2D-conv hot path.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this file; the product package (raygun_amd/) never does and
fails loudly when its HIP library is missing.  Parity status: PINNED — every class here is
checked against the reference's own modules (imported from /root/reference in the build
container by tests/golden/make_golden.py) through the fixtures in tests/golden/*.npz, and
tests/test_oracle_vs_reference.py re-checks them live whenever /root/reference is present.

Each class cites the reference lines it restates (paths relative to
/root/reference/src/raygun/torch/).  This is a restatement, not a copy: the layer *order*,
parameter names and state_dict keys are reproduced so checkpoints interchange
(SURVEY.md §5 "Checkpoint / resume", Appendix B), the code is written independently.
"""
import functools
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def _nd(name, nd):
    return getattr(nn, f"{name}{nd}d")


def _act(activation):
    # networks/UNet.py:41-47 — str -> torch.nn class, class/partial -> instance, None -> Identity
    if activation is None:
        return nn.Identity()
    if isinstance(activation, str):
        return getattr(nn, activation)()
    return activation()


def center_crop(x, shape):
    # networks/UNet.py:97-106, :327-336 ; ResNet.py:422-434 ; losses/BaseCompetentLoss.py:33-42
    nd = len(shape)
    off = [(a - b) // 2 for a, b in zip(x.shape[-nd:], shape)]
    idx = (Ellipsis,) + tuple(slice(o, o + s) for o, s in zip(off, shape))
    return x[idx]


# ----------------------------------------------------------------------------- U-Net
class ConvPass(nn.Module):
    """networks/UNet.py:12-117."""

    def __init__(self, input_nc, output_nc, kernel_sizes, activation, padding="valid",
                 residual=False, padding_mode="reflect", norm_layer=None):
        super().__init__()
        self.activation = _act(activation)
        self.residual = residual
        self.padding = padding
        layers = []
        for i, ks in enumerate(kernel_sizes):
            self.dims = len(ks)
            if self.dims not in (2, 3):
                raise RuntimeError("%dD convolution not implemented" % self.dims)
            conv = _nd("Conv", self.dims)
            layers.append(conv(input_nc, output_nc, tuple(ks), padding=padding, padding_mode=padding_mode))
            if residual and i == 0:
                groups = input_nc if input_nc < output_nc else output_nc
                self.x_init_map = conv(input_nc, output_nc, (1,) * self.dims, padding=padding,
                                       padding_mode=padding_mode, bias=False, groups=groups)
            if norm_layer is not None:
                layers.append(norm_layer(output_nc))
            if not (residual and i == len(kernel_sizes) - 1):
                layers.append(self.activation)
            input_nc = output_nc
        self.conv_pass = nn.Sequential(*layers)

    def forward(self, x):
        if not self.residual:
            return self.conv_pass(x)
        res = self.conv_pass(x)
        init_x = self.x_init_map(x)
        if self.padding.lower() == "valid":
            init_x = center_crop(init_x, tuple(res.shape[-self.dims:]))
        return self.activation(init_x + res)


class ConvDownsample(nn.Module):
    """networks/UNet.py:120-188 (strided valid conv + norm + act)."""

    def __init__(self, input_nc, output_nc, kernel_sizes, downsample_factor, activation,
                 padding="valid", padding_mode="reflect", norm_layer=None):
        super().__init__()
        self.activation = _act(activation)
        self.padding = padding
        self.dims = len(kernel_sizes)
        layers = [_nd("Conv", self.dims)(input_nc, output_nc, tuple(kernel_sizes),
                                         stride=tuple(downsample_factor), padding="valid",
                                         padding_mode=padding_mode)]
        if norm_layer is not None:
            layers.append(norm_layer(output_nc))
        layers.append(self.activation)
        self.conv_pass = nn.Sequential(*layers)

    def forward(self, x):
        return self.conv_pass(x)


class MaxDownsample(nn.Module):
    """networks/UNet.py:191-238.  flexible=True floors odd sizes (torch MaxPool default)."""

    def __init__(self, downsample_factor, flexible=True):
        super().__init__()
        self.dims = len(downsample_factor)
        self.downsample_factor = tuple(downsample_factor)
        self.flexible = flexible
        self.down = _nd("MaxPool", self.dims)(self.downsample_factor, stride=self.downsample_factor)

    def check_mismatch(self, size):
        for d in range(1, self.dims + 1):
            if size[-d] % self.downsample_factor[-d] != 0:
                raise RuntimeError("Can not downsample shape %s with factor %s, mismatch in spatial dimension %d"
                                   % (size, self.downsample_factor, self.dims - d))

    def forward(self, x):
        if not self.flexible:
            self.check_mismatch(x.size())
        return self.down(x)


class Upsample(nn.Module):
    """networks/UNet.py:241-351: up -> crop_to_factor -> centre-crop skip -> cat([skip, up])."""

    def __init__(self, scale_factor, mode=None, input_nc=None, output_nc=None,
                 crop_factor=None, next_conv_kernel_sizes=None):
        super().__init__()
        if crop_factor is not None:
            assert next_conv_kernel_sizes is not None
        self.crop_factor = crop_factor
        self.next_conv_kernel_sizes = next_conv_kernel_sizes
        self.dims = len(scale_factor)
        if mode == "transposed_conv":
            self.up = _nd("ConvTranspose", self.dims)(input_nc, output_nc, kernel_size=tuple(scale_factor),
                                                      stride=tuple(scale_factor))
        else:
            self.up = nn.Upsample(scale_factor=tuple(scale_factor), mode=mode)

    def target_shape(self, spatial):
        # UNet.py:275-325: largest n with n*f + c <= s ;  c = sum(ks-1)
        conv_crop = [sum(ks[d] - 1 for ks in self.next_conv_kernel_sizes) for d in range(self.dims)]
        out = []
        for s, c, f in zip(spatial, conv_crop, self.crop_factor):
            n = int(math.floor(float(s - c) / f))
            out.append(n * f + c)
        if tuple(out) != tuple(spatial):
            assert all(t > c for t, c in zip(out, conv_crop)), "Feature map too small for translation equivariance"
        return tuple(out)

    def forward(self, f_left, g_out):
        g_up = self.up(g_out)
        if self.crop_factor is not None:
            g_up = center_crop(g_up, self.target_shape(tuple(g_up.shape[-self.dims:])))
        f_crop = center_crop(f_left, tuple(g_up.shape[-self.dims:]))
        return torch.cat([f_crop, g_up], dim=1)


class UNet(nn.Module):
    """networks/UNet.py:354-666 (noise layers omitted: SURVEY §2 #4 keeps them torch passthrough)."""

    def __init__(self, input_nc, ngf, fmap_inc_factor, downsample_factors, kernel_size_down=None,
                 kernel_size_up=None, activation="ReLU", output_nc=None, num_heads=1,
                 constant_upsample=False, downsample_method="max", padding_type="valid",
                 residual=False, norm_layer=None, add_noise=False):
        super().__init__()
        assert not add_noise, "oracle: noise blocks are out of scope"
        self.ndims = len(downsample_factors[0])
        self.num_levels = len(downsample_factors) + 1
        self.num_heads = num_heads
        self.input_nc = input_nc
        self.output_nc = output_nc if output_nc else ngf
        self.residual = residual
        L = self.num_levels
        if kernel_size_down is None:
            kernel_size_down = [[(3,) * self.ndims, (3,) * self.ndims]] * L
        if kernel_size_up is None:
            kernel_size_up = [[(3,) * self.ndims, (3,) * self.ndims]] * (L - 1)
        # UNet.py:506-522 crop factors = cumulative products from the bottom (valid only)
        crop_factors, prod = [], None
        for f in downsample_factors[::-1]:
            if padding_type.lower() == "valid":
                prod = list(f) if prod is None else [a * b for a, b in zip(f, prod)]
            elif padding_type.lower() == "same":
                prod = None
            else:
                raise RuntimeError(f"Invalid padding_type option: {padding_type}")
            crop_factors.append(prod)
        crop_factors = crop_factors[::-1]
        is_max = downsample_method.lower() == "max"
        f = fmap_inc_factor
        self.l_conv = nn.ModuleList([
            ConvPass(input_nc if lv == 0 else ngf * f ** (lv - int(is_max)), ngf * f ** lv,
                     kernel_size_down[lv], activation=activation, padding=padding_type,
                     residual=residual, norm_layer=norm_layer) for lv in range(L)])
        self.dims = self.l_conv[0].dims
        if is_max:
            self.l_down = nn.ModuleList([MaxDownsample(downsample_factors[lv]) for lv in range(L - 1)])
        elif downsample_method.lower() == "convolve":
            self.l_down = nn.ModuleList([
                ConvDownsample(ngf * f ** lv, ngf * f ** (lv + 1), kernel_size_down[lv][0],
                               downsample_factors[lv], activation=activation, padding=padding_type,
                               norm_layer=norm_layer) for lv in range(L - 1)])
        else:
            raise RuntimeError(f'Unknown downsampling method {downsample_method}. Use "max" or "convolve" instead.')
        self.r_up = nn.ModuleList([nn.ModuleList([
            Upsample(downsample_factors[lv], mode="nearest" if constant_upsample else "transposed_conv",
                     input_nc=ngf * f ** (lv + 1), output_nc=ngf * f ** (lv + 1),
                     crop_factor=crop_factors[lv], next_conv_kernel_sizes=kernel_size_up[lv])
            for lv in range(L - 1)]) for _ in range(num_heads)])
        self.r_conv = nn.ModuleList([nn.ModuleList([
            ConvPass(ngf * f ** lv + ngf * f ** (lv + 1),
                     ngf * f ** lv if output_nc is None or lv != 0 else output_nc,
                     kernel_size_up[lv], activation=activation, padding=padding_type,
                     residual=residual, norm_layer=norm_layer)
            for lv in range(L - 1)]) for _ in range(num_heads)])

    def rec_forward(self, level, f_in):
        i = self.num_levels - level - 1
        f_left = self.l_conv[i](f_in)
        if level == 0:
            return [f_left] * self.num_heads
        g_in = self.l_down[i](f_left)
        gs = self.rec_forward(level - 1, g_in)
        rights = [self.r_up[h][i](f_left, gs[h]) for h in range(self.num_heads)]
        return [self.r_conv[h][i](rights[h]) for h in range(self.num_heads)]

    def forward(self, x):
        y = self.rec_forward(self.num_levels - 1, x)
        return y[0] if self.num_heads == 1 else y


# ----------------------------------------------------------------------------- ResNet generator
def _is_instance_norm(norm_layer, nd):
    target = _nd("InstanceNorm", nd)
    if isinstance(norm_layer, functools.partial):
        return norm_layer.func == target
    return norm_layer == target


def _padder(padding_type, nd, width):
    """ResNet.py:264-278 / :388-401: returns (pad modules, conv padding arg)."""
    pt = padding_type.lower()
    if pt in ("reflect", "same"):
        return [_nd("ReflectionPad", nd)(width)], 0
    if pt == "replicate":
        return [_nd("ReplicationPad", nd)(width)], 0
    if pt == "zeros":
        return [], width
    if pt == "valid":
        return [], "valid"
    raise NotImplementedError("padding [%s] is not implemented" % padding_type)


class ResnetBlock(nn.Module):
    """ResNet.py:347-443 (3D) / :127-223 (2D): [pad,conv3,norm,act,(dropout),pad,conv3,norm] + skip."""

    def __init__(self, nd, dim, padding_type, norm_layer, use_dropout, use_bias, activation=nn.ReLU):
        super().__init__()
        self.nd = nd
        self.padding_type = padding_type
        pad, p = _padder(padding_type, nd, 1)
        conv = _nd("Conv", nd)
        blk = list(_padder(padding_type, nd, 1)[0])
        blk += [conv(dim, dim, kernel_size=3, padding=p, bias=use_bias), norm_layer(dim), activation()]
        if use_dropout:
            blk += [nn.Dropout(0.2)]
        blk += list(_padder(padding_type, nd, 1)[0])
        blk += [conv(dim, dim, kernel_size=3, padding=p, bias=use_bias), norm_layer(dim)]
        self.conv_block = nn.Sequential(*blk)

    def forward(self, x):
        res = self.conv_block(x)
        if self.padding_type == "valid":
            return center_crop(x, tuple(res.shape[-self.nd:])) + res
        return x + res


class ResnetGenerator(nn.Module):
    """ResNet.py:226-344 (3D) and :6-124 (2D).

    2D quirk kept: the last 7x7 conv outputs `ngf`, not `output_nc` (ResNet.py:118); the
    3D one outputs `output_nc` (:338).  Through get_network both end in an appended 1x1 conv.
    """

    def __init__(self, nd, input_nc=1, output_nc=None, ngf=64, norm_layer=None, use_dropout=False,
                 n_blocks=6, padding_type="reflect", activation=nn.ReLU, add_noise=False, n_downsampling=2):
        super().__init__()
        assert n_blocks >= 0 and not add_noise
        if nd not in (2, 3):
            raise ValueError(nd, "Only 2D or 3D currently implemented. Feel free to contribute more!")
        if norm_layer is None:
            norm_layer = _nd("InstanceNorm", nd)
        use_bias = _is_instance_norm(norm_layer, nd)
        if output_nc is None:
            output_nc = ngf
        conv, convT = _nd("Conv", nd), _nd("ConvTranspose", nd)
        pads, p = _padder(padding_type, nd, 3)
        updown_p = 0 if padding_type.lower() == "valid" else 1
        m = list(_padder(padding_type, nd, 3)[0])
        m += [conv(input_nc, ngf, kernel_size=7, padding=p, bias=use_bias), norm_layer(ngf), activation()]
        for i in range(n_downsampling):
            mult = 2 ** i
            m += [conv(ngf * mult, ngf * mult * 2, kernel_size=3, stride=2, padding=updown_p, bias=use_bias),
                  norm_layer(ngf * mult * 2), activation()]
        mult = 2 ** n_downsampling
        for _ in range(n_blocks):
            m += [ResnetBlock(nd, ngf * mult, padding_type.lower(), norm_layer, use_dropout, use_bias, activation)]
        for i in range(n_downsampling):
            mult = 2 ** (n_downsampling - i)
            m += [convT(ngf * mult, int(ngf * mult / 2), kernel_size=3, stride=2, padding=updown_p,
                        output_padding=updown_p, bias=use_bias),
                  norm_layer(int(ngf * mult / 2)), activation()]
        m += list(_padder(padding_type, nd, 3)[0])
        m += [conv(ngf, ngf if nd == 2 else output_nc, kernel_size=7, padding=p)]
        self.model = nn.Sequential(*m)

    def forward(self, x):
        return self.model(x)


def ResNet(ndims, **kw):
    """ResNet.py:446-473."""
    return ResnetGenerator(ndims, **kw)


# ----------------------------------------------------------------------------- PatchGAN discriminator
class NLayerDiscriminator(nn.Module):
    """NLayerDiscriminator.py:104-180 (3D), :5-101 (2D)."""

    def __init__(self, ndims, input_nc=1, ngf=64, n_layers=3, norm_layer=None, kw=4, downsampling_kw=None):
        super().__init__()
        if ndims not in (2, 3):
            raise ValueError(ndims, "Only 2D or 3D currently implemented. Feel free to contribute more!")
        if norm_layer is None:
            norm_layer = _nd("BatchNorm", ndims)
        use_bias = _is_instance_norm(norm_layer, ndims)
        conv = _nd("Conv", ndims)
        ds_kw = kw if downsampling_kw is None else downsampling_kw
        seq = [conv(input_nc, ngf, kernel_size=ds_kw, stride=2, padding=1), nn.LeakyReLU(0.2, True)]
        mult = 1
        for n in range(1, n_layers):
            prev, mult = mult, min(2 ** n, 8)
            seq += [conv(ngf * prev, ngf * mult, kernel_size=ds_kw, stride=2, padding=1, bias=use_bias),
                    norm_layer(ngf * mult), nn.LeakyReLU(0.2, True)]
        prev, mult = mult, min(2 ** n_layers, 8)
        seq += [conv(ngf * prev, ngf * mult, kernel_size=kw, stride=1, padding=1, bias=use_bias),
                norm_layer(ngf * mult), nn.LeakyReLU(0.2, True)]
        seq += [conv(ngf * mult, 1, kernel_size=kw, stride=1, padding=1)]
        self.model = nn.Sequential(*seq)

    @property
    def FOV(self):
        # NLayerDiscriminator.py:79-97 (defined on the 2D class only): receptive field walked back
        # from one output pixel; the loop stops before layer 0 exactly as the reference's does.
        ks = [m.kernel_size[-1] for m in self.model if hasattr(m, "kernel_size")]
        st = [m.stride[-1] for m in self.model if hasattr(m, "kernel_size")]
        r = 1
        for l in range(len(ks) - 1, 0, -1):
            r = st[l] * r + (ks[l] - st[l])
        return r

    def forward(self, x):
        return self.model(x)


# ----------------------------------------------------------------------------- factory / init
def init_weights(net, init_type="normal", init_gain=0.02):
    """networks/utils.py:45-81."""
    def fn(m):
        cls = m.__class__.__name__
        if hasattr(m, "weight") and (cls.find("Conv") != -1 or cls.find("Linear") != -1):
            if init_type == "normal":
                nn.init.normal_(m.weight.data, 0.0, init_gain)
            elif init_type == "xavier":
                nn.init.xavier_normal_(m.weight.data, gain=init_gain)
            elif init_type == "kaiming":
                nn.init.kaiming_normal_(m.weight.data, a=0, mode="fan_in")
            elif init_type == "orthogonal":
                nn.init.orthogonal_(m.weight.data, gain=init_gain)
            else:
                raise NotImplementedError(init_type)
            if getattr(m, "bias", None) is not None:
                nn.init.constant_(m.bias.data, 0.0)
        elif cls.find("BatchNorm2d") != -1:
            nn.init.normal_(m.weight.data, 1.0, init_gain)
            nn.init.constant_(m.bias.data, 0.0)
    net.apply(fn)


def get_network(ndims, net_type, net_kwargs, init=True):
    """systems/BaseSystem.py:163-242 (the factory the drop-in must keep working)."""
    kw = dict(net_kwargs)
    final_activation = kw.pop("final_activation", None)
    output_nc = kw.pop("output_nc", kw["input_nc"])
    if isinstance(final_activation, str):
        final_activation = getattr(nn, final_activation)
    add_final = True
    if net_type == "unet":
        if "downsample_factors" not in kw:  # BaseSystem.py:145-161
            down = kw.pop("down_factor", 2)
            num_downs = kw.pop("num_downs", 3)
            kw["downsample_factors"] = [(down,) * ndims] * (num_downs - 1)
        net = UNet(**kw)
    elif net_type == "resnet":
        net = ResNet(ndims, **kw)
    elif net_type == "patchdiscriminator":
        kw["norm_layer"] = functools.partial(_nd("InstanceNorm", ndims), affine=False, track_running_stats=False)
        net = NLayerDiscriminator(ndims, **kw)
        add_final = False
    else:
        raise RuntimeError(f"Unknown network type requested: {net_type}")
    if add_final:
        layers = [net, _nd("Conv", ndims)(kw["ngf"], output_nc, (1,) * ndims,
                                          padding=kw.get("padding_type", "valid"))]
        if final_activation is not None:
            layers.append(final_activation())  # the reference appends the class (BaseSystem.py:221-222: bug)
        net = nn.Sequential(*layers)
    if init:
        activation = kw.get("activation", nn.ReLU)
        if activation is not None:
            init_weights(net, init_type="kaiming")
        else:
            init_weights(net, init_type="normal", init_gain=0.05)
    return net


# ----------------------------------------------------------------------------- models
class MTLSDModel(nn.Module):
    """models/MTLSDModel.py:12-63."""

    def __init__(self, unet_kwargs=None, num_affs=3):
        super().__init__()
        if unet_kwargs is None:
            unet_kwargs = {"input_nc": 1, "ngf": 12, "fmap_inc_factor": 6,
                           "downsample_factors": [(2, 2, 2)] * 3, "constant_upsample": True}
        self.unet = UNet(**unet_kwargs)
        self.aff_head = ConvPass(unet_kwargs["ngf"], num_affs, [[1, 1, 1]], activation="Sigmoid")
        self.lsd_head = ConvPass(unet_kwargs["ngf"], 10, [[1, 1, 1]], activation="Sigmoid")
        self.output_arrays = ["pred_affs", "pred_lsds"]

    def forward(self, raw):
        z = self.unet(raw)
        return self.aff_head(z), self.lsd_head(z)


class CycleModel(nn.Module):
    """models/CycleModel.py:8-89 (crop_pad / sampling bottleneck: off in every BASELINE config)."""

    def __init__(self, netG1, netG2, split=False):
        super().__init__()
        self.netG1, self.netG2, self.split = netG1, netG2, split
        self.output_arrays = ["fake_B", "cycled_B", "fake_A", "cycled_A"]

    def forward(self, real_A=None, real_B=None):
        assert real_A is not None or real_B is not None
        fake_B = cycled_A = fake_A = cycled_B = None
        if real_A is not None:
            fake_B = self.netG1(real_A)
            cycled_A = self.netG2(fake_B.detach() if self.split else fake_B)
        if real_B is not None:
            fake_A = self.netG2(real_B)
            cycled_B = self.netG1(fake_A.detach() if self.split else fake_A)
        return fake_B, cycled_B, fake_A, cycled_A


# ----------------------------------------------------------------------------- losses
class GANLoss(nn.Module):
    """losses/GANLoss.py:5-64."""

    def __init__(self, gan_mode, target_real_label=1.0, target_fake_label=0.0):
        super().__init__()
        self.register_buffer("real_label", torch.tensor(target_real_label))
        self.register_buffer("fake_label", torch.tensor(target_fake_label))
        self.gan_mode = gan_mode
        if gan_mode == "lsgan":
            self.loss = nn.MSELoss()
        elif gan_mode == "vanilla":
            self.loss = nn.BCEWithLogitsLoss()
        elif gan_mode == "wgangp":
            self.loss = None
        else:
            raise NotImplementedError("gan mode %s not implemented" % gan_mode)

    def __call__(self, prediction, target_is_real):
        if self.gan_mode in ("lsgan", "vanilla"):
            t = (self.real_label if target_is_real else self.fake_label).expand_as(prediction)
            return self.loss(prediction, t)
        return -prediction.mean() if target_is_real else prediction.mean()


def _set_requires_grad(nets, flag):
    for net in nets:
        for p in net.parameters():
            p.requires_grad = flag


_G_LAMBDA = {"l1_loss": {"cycled": 10, "identity": 0}, "gan_loss": {"fake": 1, "cycled": 0}}
_D_LAMBDA = {"real": 1, "fake": 1, "cycled": 0}


class CycleLoss(nn.Module):
    """losses/LinkCycleLoss.py:10-176 and losses/SplitCycleLoss.py:10-194 in one class.

    link : one optimiser over G1 u G2; both generator backwards (retain_graph) then one step.
    split: optimiser per generator; G1 is stepped before G2's backward; plain backward().
    """

    def __init__(self, netD1, netG1, netD2, netG2, optimizers_G, optimizer_D, dims, split=False,
                 l1_loss=None, g_lambda_dict=None, d_lambda_dict=None, gan_mode="lsgan"):
        super().__init__()
        self.netD1, self.netG1, self.netD2, self.netG2 = netD1, netG1, netD2, netG2
        self.opt_G = optimizers_G  # link: [opt_G] ; split: [opt_G1, opt_G2]
        self.opt_D = optimizer_D
        self.dims, self.split = dims, split
        self.l1_loss = l1_loss if l1_loss is not None else nn.SmoothL1Loss()
        self.g_lambda_dict = g_lambda_dict or {"A": _G_LAMBDA, "B": _G_LAMBDA}
        self.d_lambda_dict = d_lambda_dict or {"A": _D_LAMBDA, "B": _D_LAMBDA}
        self.gan_mode = gan_mode
        self.gan_loss = GANLoss(gan_mode)
        self.loss_dict = {}

    def backward_D(self, side, dnet, dd):  # LinkCycleLoss.py:43-59
        loss = 0
        for key, lam in self.d_lambda_dict[side].items():
            if lam != 0:
                this = self.gan_loss(dnet(dd[key].detach()), key == "real")
                self.loss_dict[f"Discriminator_{side}/{key}"] = this
                loss = loss + lam * this
        loss.backward()
        return loss

    def backward_Ds(self, data, n_loop=5):  # :61-85
        _set_requires_grad([self.netG1, self.netG2], False)
        self.opt_D.zero_grad(set_to_none=True)
        if self.gan_mode.lower() == "wgangp":
            for _ in range(n_loop):
                l1 = self.backward_D("B", self.netD1, data["B"])
                l2 = self.backward_D("A", self.netD2, data["A"])
                self.opt_D.step()
                for d in (self.netD1, self.netD2):  # BaseCompetentLoss.py:44-48
                    for m in d.model:
                        if hasattr(m, "weight") and hasattr(m.weight, "data"):
                            m.weight.data = m.weight.data.clamp(-0.01, 0.01)
        else:
            l1 = self.backward_D("B", self.netD1, data["B"])
            l2 = self.backward_D("A", self.netD2, data["A"])
            self.opt_D.step()
        _set_requires_grad([self.netG1, self.netG2], True)
        return l1, l2

    def backward_G(self, side, gnet, dnet, dd):  # :87-114
        loss = 0
        real = dd["real"]
        for fcn_name, lambdas in self.g_lambda_dict[side].items():
            for key, lam in lambdas.items():
                if lam == 0:
                    continue
                if key == "identity" and key not in dd:
                    dd["identity"] = gnet(real)
                pred = dd[key]
                if fcn_name == "l1_loss":
                    r = real
                    if real.shape[-self.dims:] != pred.shape[-self.dims:]:
                        r = center_crop(real, tuple(pred.shape[-self.dims:]))
                    this = self.l1_loss(r, pred)
                else:
                    this = self.gan_loss(dnet(pred), True)
                self.loss_dict[f"{fcn_name}/{key}_{side}"] = this
                loss = loss + lam * this
        if self.split:
            loss.backward()  # SplitCycleLoss.py:113
        else:
            loss.backward(retain_graph=True)  # LinkCycleLoss.py:113
        return loss

    def backward_Gs(self, data):
        _set_requires_grad([self.netD1, self.netD2], False)
        if not self.split:  # LinkCycleLoss.py:116-135
            self.opt_G[0].zero_grad(set_to_none=True)
            l1 = self.backward_G("B", self.netG1, self.netD1, data["B"])
            l2 = self.backward_G("A", self.netG2, self.netD2, data["A"])
            self.opt_G[0].step()
            _set_requires_grad([self.netD1, self.netD2], True)
        else:  # SplitCycleLoss.py:116-151
            _set_requires_grad([self.netG1], True)
            _set_requires_grad([self.netG2], False)
            self.opt_G[0].zero_grad(set_to_none=True)
            l1 = self.backward_G("B", self.netG1, self.netD1, data["B"])
            self.opt_G[0].step()
            _set_requires_grad([self.netG2], True)
            _set_requires_grad([self.netG1], False)
            self.opt_G[1].zero_grad(set_to_none=True)
            l2 = self.backward_G("A", self.netG2, self.netD2, data["A"])
            self.opt_G[1].step()
            _set_requires_grad([self.netG1, self.netD1, self.netD2], True)
        return l1, l2

    def forward(self, real_A, fake_A, cycled_A, real_B, fake_B, cycled_B):  # :137-176
        if real_A.shape[-self.dims:] != fake_B.shape[-self.dims:]:
            real_A = center_crop(real_A, tuple(fake_A.shape[-self.dims:]))
            real_B = center_crop(real_B, tuple(fake_B.shape[-self.dims:]))
        data = {"A": {"real": real_A, "fake": fake_A, "cycled": cycled_A},
                "B": {"real": real_B, "fake": fake_B, "cycled": cycled_B}}
        g1, g2 = self.backward_Gs(data)
        d1, d2 = self.backward_Ds(data)
        self.loss_dict.update({"Total_Loss/D1": float(d1), "Total_Loss/D2": float(d2),
                               "Total_Loss/G1": float(g1), "Total_Loss/G2": float(g2)})
        return g1 + g2 + d1 + d2


class WeightedMSELoss_MTLSD(nn.Module):
    """losses/WeightedMSELoss_MTLSD.py:6-85."""

    def __init__(self):
        super().__init__()
        self.loss_dict = {}

    @staticmethod
    def _calc_loss(pred, target, weights):  # :11-25 (data-dependent branch at :15)
        scaled = weights * (pred - target) ** 2
        if len(torch.nonzero(scaled)) != 0:
            return torch.mean(torch.masked_select(scaled, torch.gt(weights, 0)))
        return torch.mean(scaled)

    def forward(self, pred_lsds, gt_lsds, lsds_weights, pred_affs, gt_affs, affs_weights):
        lsd = self._calc_loss(pred_lsds, gt_lsds, lsds_weights)
        aff = self._calc_loss(pred_affs, gt_affs, affs_weights)
        self.loss_dict = {"LSDs": lsd.detach(), "Affinities": aff.detach()}
        return lsd + aff


def lambda_lr_rule(epoch, epoch_count, n_epochs, n_epochs_decay):
    """optimizers/BaseDummyOptimizer.py:24-31."""
    return 1.0 - max(0, epoch + epoch_count - n_epochs) / (n_epochs_decay + 1.0)


# ----------------------------------------------------------------------------- drivers (call order of §3.1/§3.2)
def build_cyclegan(ndims, gnet_kwargs, dnet_kwargs, split=False, g_lr=1e-5, d_lr=1e-5, betas=(0.95, 0.999),
                   loss_kwargs=None, gnet_type="resnet", dnet_type="patchdiscriminator", init=True):
    """systems/CycleGAN.py:133-233 restated for dict configs; returns (model, loss, opts)."""
    import itertools
    G1 = get_network(ndims, gnet_type, gnet_kwargs, init)
    G2 = get_network(ndims, gnet_type, gnet_kwargs, init)
    D1 = get_network(ndims, dnet_type, dnet_kwargs, init)
    D2 = get_network(ndims, dnet_type, dnet_kwargs, init)
    model = CycleModel(G1, G2, split=split)
    opt_D = torch.optim.Adam(itertools.chain(D1.parameters(), D2.parameters()), lr=d_lr, betas=betas)
    if split:
        opts_G = [torch.optim.Adam(G1.parameters(), lr=g_lr, betas=betas),
                  torch.optim.Adam(G2.parameters(), lr=g_lr, betas=betas)]
    else:
        opts_G = [torch.optim.Adam(itertools.chain(G1.parameters(), G2.parameters()), lr=g_lr, betas=betas)]
    loss = CycleLoss(D1, G1, D2, G2, opts_G, opt_D, ndims, split=split, **(loss_kwargs or {}))
    return model, loss, opts_G + [opt_D]


def cyclegan_step(model, loss, real_A, real_B):
    """One gp.torch.Train iteration as reconstructed in SURVEY.md §3.1."""
    fake_B, cycled_B, fake_A, cycled_A = model(real_A=real_A, real_B=real_B)
    total = loss(real_A, fake_A, cycled_A, real_B, fake_B, cycled_B)
    return total, (fake_B, cycled_B, fake_A, cycled_A)


def mtlsd_step(model, loss_fn, opt, raw, gt_lsds, w_lsds, gt_affs, w_affs):
    """zero_grad -> model -> loss -> backward -> Adam.step."""
    opt.zero_grad(set_to_none=True)
    affs, lsds = model(raw)
    loss = loss_fn(lsds, gt_lsds, w_lsds, affs, gt_affs, w_affs)
    loss.backward()
    opt.step()
    return loss, affs, lsds


# ----------------------------------------------------------------------------- bf16 storage emulation
class _RoundBF16(torch.autograd.Function):
    """Round-trip through bfloat16 in forward AND backward (what storing a tensor in bf16 does)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).float()

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).float()


def round_bf16(x):
    return _RoundBF16.apply(x)


def emulate_bf16(module):
    """Turn a restated network into a model of the HIP path's PRECISION POLICY (bf16 storage of every activation
    and weight operand, fp32 arithmetic): weights are rounded to bf16 in place and every leaf layer's output is
    rounded through bf16 in both directions.  ReLU/max-pool decisions then agree with the GPU's except where fp32
    accumulation order matters, which is what makes gradient parity testable at all: against a pure-fp32 run a
    bf16 network's gradient differs by ~sqrt(P[sign flip]) in relative L2 however correct the kernels are.
    Returns the hook handles (call .remove() on each to undo)."""
    handles = []
    with torch.no_grad():
        for p in module.parameters():
            if p.dim() > 1:
                p.copy_(p.to(torch.bfloat16).float())
    for m in module.modules():
        if len(list(m.children())) == 0 and not isinstance(m, nn.Identity):
            handles.append(m.register_forward_hook(lambda mod, inp, out: round_bf16(out)))
    return handles

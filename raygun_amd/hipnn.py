"""Drop-in layer classes: torch.nn subclasses (same constructors, parameters, state_dict keys,
class names — so init_weights / set_norm_mode / load_state_dict of the reference keep working)
whose forward runs the gfx950 kernels.

Scoping: inside a raygun_amd network, activations travel as channels-last bf16 tensors; the
OUTERMOST raygun_amd module of a call converts its result to the fp32 NCDHW tensor the reference
would have returned (``scope()`` tracks the nesting).
"""
import contextlib
import functools
import os
import threading

import torch
import torch.nn as nn

from . import ops

_tls = threading.local()
CL_WEIGHTS = os.environ.get("RAYGUN_AMD_CL_WEIGHTS", "1") != "0"


@contextlib.contextmanager
def scope():
    """Yields True for the outermost raygun_amd module of the current call stack."""
    depth = getattr(_tls, "depth", 0)
    if depth == 0:
        ops.lift_reset()
    _tls.depth = depth + 1
    try:
        yield depth == 0
    finally:
        _tls.depth = depth


def scoped(forward):
    """Decorator for Module.forward: run inside a scope; leave channels-last at the outermost level."""
    @functools.wraps(forward)
    def wrapper(self, *args, **kwargs):
        with scope() as outer:
            y = forward(self, *args, **kwargs)
        return ops.leave(y) if outer else y
    return wrapper


def _symmetric(pad6_or_int, what):
    if isinstance(pad6_or_int, int):
        return (pad6_or_int,) * 3
    p = tuple(pad6_or_int)
    if len(p) == 3:
        return p
    if len(p) == 6 and p[0] == p[1] and p[2] == p[3] and p[4] == p[5]:
        return (p[4], p[2], p[0])  # F.pad order is (W_l, W_r, H_l, H_r, D_l, D_r)
    raise RuntimeError(f"raygun_amd.{what}: only symmetric padding is supported, got {p}")


_ACT_OF = {}


def act_of(m):
    """(kind, slope) if module `m` is an activation the kernels can fuse, else None."""
    for cls, fn in _ACT_OF.items():
        if isinstance(m, cls):
            return fn(m)
    return None


class ReLU(nn.ReLU):
    @scoped
    def forward(self, x):
        return ops.activation(x, "relu")


class LeakyReLU(nn.LeakyReLU):
    @scoped
    def forward(self, x):
        return ops.activation(x, "leaky", self.negative_slope)


class Sigmoid(nn.Sigmoid):
    @scoped
    def forward(self, x):
        return ops.activation(x, "sigmoid")


class Tanh(nn.Tanh):
    @scoped
    def forward(self, x):
        return ops.activation(x, "tanh")


class Identity(nn.Identity):
    pass


_ACT_OF.update({nn.ReLU: lambda m: ("relu", 0.0), nn.LeakyReLU: lambda m: ("leaky", m.negative_slope),
                nn.Sigmoid: lambda m: ("sigmoid", 0.0), nn.Tanh: lambda m: ("tanh", 0.0)})


class Conv3d(nn.Conv3d):
    """torch.nn.Conv3d on the implicit-GEMM kernel.  padding int/tuple/'valid'/'same'; padding_mode zeros|reflect.

    The fp32 master weight keeps torch's logical shape (Cout,Cin,KD,KH,KW) and state_dict key but lives in
    torch.channels_last_3d memory, [Cout][KD][KH][KW][Cin]: that is the order the weight-gradient kernels accumulate in and
    the packed MFMA operand is stored in, so the gradient needs no transpose pass, Adam (elementwise on the storage) updates
    in place and the forward packing is a straight convert.  RAYGUN_AMD_CL_WEIGHTS=0 keeps contiguous masters."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if CL_WEIGHTS and self.weight.shape[2] * self.weight.shape[3] * self.weight.shape[4] > 1:
            self.weight.data = self.weight.data.contiguous(memory_format=torch.channels_last_3d)

    def _pads(self):
        if isinstance(self.padding, str):
            if self.padding == "valid":
                return (0, 0, 0)
            tot = [d * (k - 1) for d, k in zip(self.dilation, self.kernel_size)]
            if any(t % 2 for t in tot):
                raise RuntimeError("raygun_amd.Conv3d: padding='same' needs odd kernels")
            return tuple(t // 2 for t in tot)
        return tuple(self.padding)

    def _check(self):
        if self.groups != 1 or any(d != 1 for d in self.dilation):
            raise RuntimeError("raygun_amd.Conv3d: groups/dilation != 1 are not built (use torch.nn.Conv3d)")
        if self.padding_mode not in ("zeros", "reflect", "replicate"):
            raise RuntimeError(f"raygun_amd.Conv3d: padding_mode {self.padding_mode!r} is not built")

    def run(self, x, act="none", slope=0.0):
        self._check()
        pads = self._pads()
        if self.padding_mode in ("reflect", "replicate") and any(pads):
            x = ops.pad3d(x, pads, self.padding_mode)
            pads = (0, 0, 0)
        return ops.conv3d(x, self.weight, self.bias, self.stride, pads, act, slope)

    @scoped
    def forward(self, x):
        return self.run(x)


class ConvTranspose3d(nn.ConvTranspose3d):
    def run(self, x, act="none", slope=0.0):
        if self.groups != 1 or any(d != 1 for d in self.dilation):
            raise RuntimeError("raygun_amd.ConvTranspose3d: groups/dilation != 1 are not built")
        return ops.conv_transpose3d(x, self.weight, self.bias, self.stride, self.padding, self.output_padding, act, slope)

    @scoped
    def forward(self, x):
        return self.run(x)


def _instance_norm_run(self, x, act, slope, residual, base_forward, enter=lambda t: t):
    """Shared by InstanceNorm3d / InstanceNorm2d.  affine=False runs on the HIP kernels, with or without running statistics
    (torch.nn.modules.instancenorm semantics: instance statistics normalise in training mode and update
    running_mean / running_var — unbiased variance, `momentum`, averaged over the batch — and eval mode normalises with the
    running statistics); the affine form adds the per-channel (gamma, beta) table to the same kernels (ops.norm_affine)."""
    if self.affine:
        x = enter(x)
        if self.track_running_stats and not self.training:
            return ops.norm_affine(x, self.weight, self.bias, "fixed", self.eps, act, slope, residual, self.running_mean, self.running_var)[0]
        y, st = ops.norm_affine(x, self.weight, self.bias, "instance", self.eps, act, slope, residual)
        if self.track_running_stats:
            with torch.no_grad():
                cnt = x.shape[2] * x.shape[3] * x.shape[4]
                m = self.momentum if self.momentum is not None else 0.0
                self.running_mean.mul_(1 - m).add_(st[:, :, 0].mean(0).to(self.running_mean.dtype), alpha=m)
                self.running_var.mul_(1 - m).add_((st[:, :, 1] * (cnt / max(cnt - 1, 1))).mean(0).to(self.running_var.dtype), alpha=m)
        return y
    x = enter(x)
    if self.track_running_stats and not self.training:
        rstd = torch.rsqrt(self.running_var.float() + self.eps)
        return ops.norm_fixed(x, self.running_mean, rstd, act, slope, residual)
    if not self.track_running_stats:
        return ops.instance_norm(x, self.eps, act, slope, residual)
    y, mr = ops.instance_norm(x, self.eps, act, slope, residual, return_stats=True)
    with torch.no_grad():
        n, c = x.shape[0], x.shape[1]
        cnt = x.shape[2] * x.shape[3] * x.shape[4]
        mean = mr[:, 0].view(n, c).mean(0)
        var = ((1.0 / (mr[:, 1] * mr[:, 1]) - self.eps).clamp_(min=0) * (cnt / max(cnt - 1, 1))).view(n, c).mean(0)
        m = self.momentum if self.momentum is not None else 0.0   # F.instance_norm's rule; num_batches_tracked is not touched by torch either
        self.running_mean.mul_(1 - m).add_(mean.to(self.running_mean.dtype), alpha=m)
        self.running_var.mul_(1 - m).add_(var.to(self.running_var.dtype), alpha=m)
    return y


class InstanceNorm3d(nn.InstanceNorm3d):
    """affine=False runs on the HIP kernels — the plain form get_network builds for the hot path (systems/BaseSystem.py:193-199,
    networks/ResNet.py:231-262 default) and the running-statistics form of the experiment configs; affine uses torch's op."""

    def hip_ok(self):
        return True

    def run(self, x, act="none", slope=0.0, residual=None):
        return _instance_norm_run(self, x, act, slope, residual, super().forward)

    @scoped
    def forward(self, x):
        return self.run(x)


def _batch_norm_run(self, x, act, slope, residual, enter=lambda t: t):
    """torch.nn.modules.batchnorm semantics on the HIP norm kernels: training mode (or no running statistics) normalises with the mean /
    biased variance over batch x voxels and updates running_mean / running_var (unbiased variance; `momentum`, or the cumulative average
    when momentum is None) and num_batches_tracked; eval mode uses the running statistics."""
    x = enter(x)
    use_batch = self.training or not self.track_running_stats
    w, b = (self.weight, self.bias) if self.affine else (None, None)
    if not use_batch:
        return ops.norm_affine(x, w, b, "fixed", self.eps, act, slope, residual, self.running_mean, self.running_var)[0]
    y, st = ops.norm_affine(x, w, b, "batch", self.eps, act, slope, residual)
    if self.training and self.track_running_stats:
        with torch.no_grad():
            self.num_batches_tracked += 1
            m = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
            cnt = x.shape[0] * x.shape[2] * x.shape[3] * x.shape[4]
            self.running_mean.mul_(1 - m).add_(st[:, 0].to(self.running_mean.dtype), alpha=m)
            self.running_var.mul_(1 - m).add_((st[:, 1] * (cnt / max(cnt - 1, 1))).to(self.running_var.dtype), alpha=m)
    return y


class BatchNorm3d(nn.BatchNorm3d):
    """NLayerDiscriminator3D's default norm_layer (NLayerDiscriminator.py:107-115) on the HIP norm kernels (ops.norm_affine, batch statistics)."""

    def hip_ok(self):
        return True

    def run(self, x, act="none", slope=0.0, residual=None):
        return _batch_norm_run(self, x, act, slope, residual)

    @scoped
    def forward(self, x):
        return self.run(x)


class ReflectionPad3d(nn.ReflectionPad3d):
    @scoped
    def forward(self, x):
        return ops.pad3d(x, _symmetric(self.padding, "ReflectionPad3d"), "reflect")


class ReplicationPad3d(nn.ReplicationPad3d):
    """padding_type="replicate" of the generators (ResNet.py:272-273, 394-395): the pad kernel's clamp mode."""
    @scoped
    def forward(self, x):
        return ops.pad3d(x, _symmetric(self.padding, "ReplicationPad3d"), "replicate")


class MaxPool3d(nn.MaxPool3d):
    @scoped
    def forward(self, x):
        ks = (self.kernel_size,) * 3 if isinstance(self.kernel_size, int) else tuple(self.kernel_size)
        st = (self.stride,) * 3 if isinstance(self.stride, int) else tuple(self.stride)
        if ks != st or self.padding not in (0, (0, 0, 0)) or self.ceil_mode:
            raise RuntimeError("raygun_amd.MaxPool3d: only kernel == stride, no padding, floor mode is built")
        return ops.maxpool3d(x, ks)


def passthrough(fn, x, *args, **kwargs):
    """Run a torch op that has no HIP kernel here (dropout, noise blocks, affine norms) on a dense fp32 view
    (4-D again if the network is a lifted 2-D one; the next raygun_amd layer re-enters)."""
    dense = ops.leave(x) if ops.is_cl(x) else x
    return fn(dense, *args, **kwargs)


# ---- 2-D twins: the same kernels on a (N,C,1,H,W) view -------------------------------------------------------------
def _pair(v):
    return (int(v),) * 2 if isinstance(v, int) else tuple(int(a) for a in v)


class Conv2d(nn.Conv2d):
    """torch.nn.Conv2d as a 3-D convolution with a (1,KH,KW) kernel."""

    def _pads(self):
        if isinstance(self.padding, str):
            if self.padding == "valid":
                return (0, 0)
            tot = [d * (k - 1) for d, k in zip(self.dilation, self.kernel_size)]
            if any(t % 2 for t in tot):
                raise RuntimeError("raygun_amd.Conv2d: padding='same' needs odd kernels")
            return tuple(t // 2 for t in tot)
        return tuple(self.padding)

    def run(self, x, act="none", slope=0.0):
        if self.groups != 1 or any(d != 1 for d in self.dilation):
            raise RuntimeError("raygun_amd.Conv2d: groups/dilation != 1 are not built (use torch.nn.Conv2d)")
        if self.padding_mode not in ("zeros", "reflect", "replicate"):
            raise RuntimeError(f"raygun_amd.Conv2d: padding_mode {self.padding_mode!r} is not built")
        x = ops.enter(x)
        pads = self._pads()
        if self.padding_mode in ("reflect", "replicate") and any(pads):
            x = ops.pad3d(x, (0,) + pads, self.padding_mode)
            pads = (0, 0)
        return ops.conv3d(x, self.weight.unsqueeze(2), self.bias, (1,) + tuple(self.stride), (0,) + pads, act, slope)

    @scoped
    def forward(self, x):
        return self.run(x)


class ConvTranspose2d(nn.ConvTranspose2d):
    def run(self, x, act="none", slope=0.0):
        if self.groups != 1 or any(d != 1 for d in self.dilation):
            raise RuntimeError("raygun_amd.ConvTranspose2d: groups/dilation != 1 are not built")
        return ops.conv_transpose3d(ops.enter(x), self.weight.unsqueeze(2), self.bias, (1,) + tuple(self.stride),
                                    (0,) + tuple(self.padding), (0,) + tuple(self.output_padding), act, slope)

    @scoped
    def forward(self, x):
        return self.run(x)


class InstanceNorm2d(nn.InstanceNorm2d):
    """As InstanceNorm3d, on the (N,C,1,H,W) view: incl. `InstanceNorm2d(track_running_stats=True, momentum=0.01)` of the published
    2-D CycleGAN configs (reference tests/torch/systems/CycleGAN_conf.json:62)."""

    def hip_ok(self):
        return True

    def run(self, x, act="none", slope=0.0, residual=None):
        return _instance_norm_run(self, x, act, slope, residual, super().forward, ops.enter)

    @scoped
    def forward(self, x):
        return self.run(x)


class BatchNorm2d(nn.BatchNorm2d):
    """NLayerDiscriminator2D's default norm_layer (NLayerDiscriminator.py:8-16), on the (N,C,1,H,W) view."""

    def hip_ok(self):
        return True

    def run(self, x, act="none", slope=0.0, residual=None):
        return _batch_norm_run(self, x, act, slope, residual, ops.enter)

    @scoped
    def forward(self, x):
        return self.run(x)


class ReflectionPad2d(nn.ReflectionPad2d):
    @scoped
    def forward(self, x):
        p = self.padding
        p = (p,) * 4 if isinstance(p, int) else tuple(p)
        if not (p[0] == p[1] and p[2] == p[3]):
            raise RuntimeError(f"raygun_amd.ReflectionPad2d: only symmetric padding is supported, got {p}")
        return ops.pad3d(ops.enter(x), (0, p[2], p[0]), "reflect")


class ReplicationPad2d(nn.ReplicationPad2d):
    @scoped
    def forward(self, x):
        p = self.padding
        p = (p,) * 4 if isinstance(p, int) else tuple(p)
        if not (p[0] == p[1] and p[2] == p[3]):
            raise RuntimeError(f"raygun_amd.ReplicationPad2d: only symmetric padding is supported, got {p}")
        return ops.pad3d(ops.enter(x), (0, p[2], p[0]), "replicate")


class MaxPool2d(nn.MaxPool2d):
    @scoped
    def forward(self, x):
        ks, st = _pair(self.kernel_size), _pair(self.stride)
        if ks != st or self.padding not in (0, (0, 0)) or self.ceil_mode:
            raise RuntimeError("raygun_amd.MaxPool2d: only kernel == stride, no padding, floor mode is built")
        return ops.maxpool3d(ops.enter(x), (1,) + ks)


_HIP_LAYER = {"Conv3d": Conv3d, "ConvTranspose3d": ConvTranspose3d, "InstanceNorm3d": InstanceNorm3d,
              "ReflectionPad3d": ReflectionPad3d, "MaxPool3d": MaxPool3d, "ReLU": ReLU, "LeakyReLU": LeakyReLU,
              "Sigmoid": Sigmoid, "Tanh": Tanh, "Conv2d": Conv2d, "ConvTranspose2d": ConvTranspose2d,
              "InstanceNorm2d": InstanceNorm2d, "ReflectionPad2d": ReflectionPad2d, "MaxPool2d": MaxPool2d,
              "BatchNorm3d": BatchNorm3d, "BatchNorm2d": BatchNorm2d, "ReplicationPad3d": ReplicationPad3d, "ReplicationPad2d": ReplicationPad2d}
_CONVS = (Conv3d, ConvTranspose3d, Conv2d, ConvTranspose2d)
_NORMS = (InstanceNorm3d, InstanceNorm2d, BatchNorm3d, BatchNorm2d)


def layers(nd):
    """(Conv, ConvTranspose, InstanceNorm, ReflectionPad, MaxPool) twins for 2-D or 3-D networks."""
    return {3: (Conv3d, ConvTranspose3d, InstanceNorm3d, ReflectionPad3d, MaxPool3d),
            2: (Conv2d, ConvTranspose2d, InstanceNorm2d, ReflectionPad2d, MaxPool2d)}[nd]


def hip_layer(factory):
    """Map a torch.nn layer class / functools.partial / name to its raygun_amd twin (unknown ones unchanged)."""
    if factory is None:
        return None
    if isinstance(factory, str):
        return _HIP_LAYER.get(factory, getattr(nn, factory))
    if isinstance(factory, functools.partial):
        return functools.partial(hip_layer(factory.func), *factory.args, **factory.keywords)
    if isinstance(factory, type) and factory.__module__.startswith("torch.nn") and factory.__name__ in _HIP_LAYER:
        return _HIP_LAYER[factory.__name__]
    return factory


def is_instance_norm_factory(factory, nd=3):
    """`use_bias = norm_layer == InstanceNorm3d` of the reference (networks/ResNet.py:259-262), for either class."""
    base = factory.func if isinstance(factory, functools.partial) else factory
    return base in (getattr(nn, f"InstanceNorm{nd}d"), InstanceNorm3d, InstanceNorm2d)


class FusedSequential(nn.Sequential):
    """nn.Sequential (same indices / state_dict keys / len) whose forward fuses
    [pad] conv [InstanceNorm] [activation]  into conv-epilogue / norm-apply kernels."""

    def run(self, x, residual=None, upto=None, start=0):
        mods = list(self)[start:upto]
        i, n = 0, len(mods)
        while i < n:
            m = mods[i]
            last = lambda j: j >= n  # noqa: E731
            if isinstance(m, _CONVS):
                nxt = mods[i + 1] if i + 1 < n else None
                a = act_of(nxt) if nxt is not None else None
                if a is not None:
                    x = m.run(x, a[0], a[1]); i += 2
                elif isinstance(nxt, _NORMS):
                    x = m.run(x)
                    a2 = act_of(mods[i + 2]) if i + 2 < n else None
                    if a2 is not None:
                        x = nxt.run(x, a2[0], a2[1]); i += 3
                    else:
                        fuse_res = residual if (i + 2 >= n) else None
                        x = nxt.run(x, residual=fuse_res); i += 2
                        if fuse_res is not None:
                            residual = None
                else:
                    x = m.run(x); i += 1
            elif isinstance(m, _NORMS):
                a = act_of(mods[i + 1]) if i + 1 < n else None
                if a is not None:
                    x = m.run(x, a[0], a[1]); i += 2
                else:
                    x = m.run(x); i += 1
            elif act_of(m) is not None and not isinstance(m, (ReLU, LeakyReLU, Sigmoid, Tanh)):
                a = act_of(m)  # a plain torch.nn activation instance handed in by the user
                x = ops.activation(x, a[0], a[1]); i += 1
            elif isinstance(m, (ReflectionPad3d, ReplicationPad3d, MaxPool3d, ReflectionPad2d, ReplicationPad2d, MaxPool2d, ReLU, LeakyReLU, Sigmoid, Tanh, FusedSequential)) or \
                    getattr(m, "_raygun_amd", False):
                x = m(x); i += 1
            elif isinstance(m, nn.Identity):
                i += 1
            else:
                x = passthrough(m, x); i += 1
        if residual is not None:
            x = ops.add(x, residual)
        return x

    @scoped
    def forward(self, x):
        return self.run(x)

"""networks/utils.py of the reference: init_weights, set_norm_mode, running-stat collection, noise blocks.
Pure host-side logic on torch modules (reference: src/raygun/torch/networks/utils.py:11-105)."""
import torch
from torch.nn import init


def get_norm_layers(net):
    return [n for n in net.modules() if "norm" in type(n).__name__.lower()]


def get_running_norm_stats(net):
    """utils.py:11-28 — concatenated running means/vars of every norm layer, or (None, None)."""
    try:
        norms = get_norm_layers(net)
        means = torch.cat([n.running_mean for n in norms])
        vars_ = torch.cat([n.running_var for n in norms])
    except Exception:
        return None, None
    return means, vars_


def set_norm_mode(net, mode="train"):
    """utils.py:31-42 — 'fix_norms' keeps the net in train mode but freezes every *norm* layer."""
    if mode == "fix_norms":
        net.train()
        for m in net.modules():
            if "norm" in type(m).__name__.lower():
                m.eval()
    if mode == "train":
        net.train()
    if mode == "eval":
        net.eval()


# init_type -> in-place initialiser of a weight tensor, given the gain (the reference's four schemes, utils.py:51-63)
_WEIGHT_INIT = {
    "normal": lambda w, gain: init.normal_(w, 0.0, gain),
    "xavier": lambda w, gain: init.xavier_normal_(w, gain=gain),
    "kaiming": lambda w, gain: init.kaiming_normal_(w, a=0, mode="fan_in"),
    "orthogonal": lambda w, gain: init.orthogonal_(w, gain=gain),
}


def _is_linear_map(module):
    """The layers the reference re-initialises: any class whose NAME contains Conv or Linear (so subclasses such as hipnn.Conv3d and
    ConvTranspose3d count) and that owns a weight."""
    name = type(module).__name__
    return ("Conv" in name or "Linear" in name) and getattr(module, "weight", None) is not None


def init_weights(net, init_type="normal", init_gain=0.02, nonlinearity="relu"):
    """utils.py:45-81: weights of every Conv* / Linear layer drawn by `init_type`, their biases zeroed; BatchNorm2d gets weight ~ N(1, gain)
    and bias 0; an unknown `init_type` raises NotImplementedError at the first such layer, as the reference does.  Prints the reference's line."""
    def visit(module):
        if _is_linear_map(module):
            try:
                draw = _WEIGHT_INIT[init_type]
            except KeyError:
                raise NotImplementedError("initialization method [%s] is not implemented" % init_type) from None
            draw(module.weight.data, init_gain)
            if getattr(module, "bias", None) is not None:
                module.bias.data.zero_()
        elif "BatchNorm2d" in type(module).__name__:
            init.normal_(module.weight.data, 1.0, init_gain)
            module.bias.data.zero_()

    print("initialize network with %s" % init_type)
    net.apply(visit)
    from ... import ops
    ops.invalidate_packed(net)  # the writes above go through .data: they do not bump the version the packed-operand caches key on


class NoiseBlock(torch.nn.Module):
    """utils.py:84-94 — appends one N(0,1) feature map (torch op; not on the benchmark path)."""

    def forward(self, x):
        shape = list(x.shape)
        shape[1] = 1
        noise = torch.empty(shape, device=x.device, dtype=x.dtype).normal_()
        return torch.cat([x, noise.requires_grad_()], 1)


class ParameterizedNoiseBlock(torch.nn.Module):
    """utils.py:97-105 — noise ~ N(x[:,0], relu(x[:,1]))."""

    def forward(self, x):
        noise = torch.normal(x[:, 0, ...], torch.relu(x[:, 1, ...])).unsqueeze(1)
        return torch.cat([x, noise.requires_grad_()], 1)

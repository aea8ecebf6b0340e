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


def init_weights(net, init_type="normal", init_gain=0.02, nonlinearity="relu"):
    """utils.py:45-81 — every Conv*/Linear weight; bias <- 0; BatchNorm2d weight ~ N(1, gain)."""

    def init_func(m):
        classname = m.__class__.__name__
        if hasattr(m, "weight") and (classname.find("Conv") != -1 or classname.find("Linear") != -1):
            if init_type == "normal":
                init.normal_(m.weight.data, 0.0, init_gain)
            elif init_type == "xavier":
                init.xavier_normal_(m.weight.data, gain=init_gain)
            elif init_type == "kaiming":
                init.kaiming_normal_(m.weight.data, a=0, mode="fan_in")
            elif init_type == "orthogonal":
                init.orthogonal_(m.weight.data, gain=init_gain)
            else:
                raise NotImplementedError("initialization method [%s] is not implemented" % init_type)
            if hasattr(m, "bias") and m.bias is not None:
                init.constant_(m.bias.data, 0.0)
        elif classname.find("BatchNorm2d") != -1:
            init.normal_(m.weight.data, 1.0, init_gain)
            init.constant_(m.bias.data, 0.0)

    print("initialize network with %s" % init_type)
    net.apply(init_func)
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

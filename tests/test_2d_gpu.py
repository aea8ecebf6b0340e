"""2-D layer twins: every raygun_amd 2-D module is its 3-D kernel on a (N,C,1,H,W) view.  Parity
against torch.nn's own 2-D modules on bf16-exact operands (forward, input gradient, parameter gradients)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from raygun_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1.5e-2


def bf(t):
    return t.to(torch.bfloat16).float()


def relerr(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def check_module(mine, ref, x, tol=TOL):
    dev = torch.device("cuda")
    ref.load_state_dict({k: bf(v) for k, v in ref.state_dict().items()})
    mine.load_state_dict(ref.state_dict())
    mine.to(dev)
    xr = x.clone().requires_grad_(True)
    xm = x.clone().to(dev).requires_grad_(True)
    yr, ym = ref(xr), mine(xm)
    assert ym.dim() == 4 and ym.dtype == torch.float32 and tuple(ym.shape) == tuple(yr.shape)
    g = bf(torch.from_numpy(synth.normal(tuple(yr.shape), 7)))
    yr.backward(g); ym.backward(g.to(dev))
    errs = {"y": relerr(ym, yr), "gx": relerr(xm.grad, xr.grad)}
    for (k, a), (_, b) in zip(mine.named_parameters(), ref.named_parameters()):
        errs["g_" + k] = relerr(a.grad, b.grad)
    bad = {k: v for k, v in errs.items() if not v <= tol}
    assert not bad, (bad, errs)


def X(shape, seed=1):
    return bf(torch.from_numpy(synth.normal(shape, seed)))


@pytest.mark.parametrize("cin,cout,k,s,p,hw", [(1, 16, 7, 1, 3, (20, 22)), (16, 32, 3, 2, 1, (18, 20)), (72, 40, 3, 1, 0, (12, 13)),
                                              (64, 64, 3, 1, 1, (16, 16)), (24, 1, 4, 1, 1, (9, 10))])
def test_conv2d(cin, cout, k, s, p, hw):
    from raygun_amd import hipnn
    check_module(hipnn.Conv2d(cin, cout, k, s, p), nn.Conv2d(cin, cout, k, s, p), X((2, cin) + hw))


def test_conv_transpose2d():
    from raygun_amd import hipnn
    check_module(hipnn.ConvTranspose2d(32, 16, 3, 2, 1, 1), nn.ConvTranspose2d(32, 16, 3, 2, 1, 1), X((2, 32, 9, 10)))


def test_norm_pad_sequence():
    """Smooth stack (no ReLU / max-pool masks to flip under bf16 storage)."""
    from raygun_amd import hipnn
    mine = hipnn.FusedSequential(hipnn.ReflectionPad2d(2), hipnn.Conv2d(8, 16, 5, bias=False), hipnn.InstanceNorm2d(16), hipnn.Tanh(),
                                 hipnn.Conv2d(16, 8, 3, padding=1), hipnn.Tanh())
    ref = nn.Sequential(nn.ReflectionPad2d(2), nn.Conv2d(8, 16, 5, bias=False), nn.InstanceNorm2d(16), nn.Tanh(),
                        nn.Conv2d(16, 8, 3, padding=1), nn.Tanh())
    check_module(mine, ref, X((2, 8, 20, 24)), tol=3e-2)


def test_maxpool2d_exact():
    from raygun_amd import hipnn
    x = X((2, 12, 18, 20))
    ym = hipnn.MaxPool2d(2)(x.cuda())
    assert torch.equal(ym.cpu(), nn.MaxPool2d(2)(x))


def test_networks_2d_structure_and_forward():
    """ResNet(2, ...) / NLayerDiscriminator(2, ...): reference layer counts, FOV, state_dict keys, forward vs the oracle."""
    import functools
    from oracle import torch_ref as O
    from raygun_amd.torch import networks
    IN2 = functools.partial(nn.InstanceNorm2d)
    dev = torch.device("cuda")
    g = networks.ResNet(2, input_nc=1, ngf=8, n_blocks=3, n_downsampling=2, norm_layer=IN2, padding_type="reflect")
    go = O.ResNet(2, input_nc=1, ngf=8, n_blocks=3, n_downsampling=2, norm_layer=IN2, padding_type="reflect")
    d = networks.NLayerDiscriminator(2, input_nc=1, ngf=8, n_layers=3, norm_layer=IN2)
    do = O.NLayerDiscriminator(2, input_nc=1, ngf=8, n_layers=3, norm_layer=IN2)
    assert len(g.model) == len(go.model) and list(g.state_dict()) == list(go.state_dict())
    assert len(d.model) == len(do.model) and list(d.state_dict()) == list(do.state_dict()) and d.FOV == do.FOV
    for mine, ref, shape in ((g, go, (1, 1, 64, 64)), (d, do, (1, 1, 64, 64))):
        sd = synth.fill_state_dict(ref.state_dict(), 31)
        ref.load_state_dict(sd); mine.load_state_dict(sd); mine.to(dev)
        x = torch.from_numpy(synth.uniform(shape, 5))
        with torch.no_grad():
            yr = ref(x); ym = mine(x.to(dev))
        assert tuple(ym.shape) == tuple(yr.shape)
        rel = float((ym.cpu() - yr).norm() / yr.norm())
        assert rel <= 3e-2, rel


def test_generator_tail_2d_is_composed_like_the_3d_one(monkeypatch):
    """get_network's Sequential(ResnetGenerator2D[..., Conv2d(ngf, ngf, 7)], Conv2d(ngf, 1, 1)[, Tanh]) (BaseSystem.py:209-224, ResNet.py:6-223):
    the last two layers run as ONE composed convolution on the 2-D path too (ops.conv3d_linear_pair with a (1, 7, 7) kernel) — same parameters
    and state_dict; outputs and every parameter's gradient against the layer-by-layer evaluation (RAYGUN_AMD_FUSE_TAIL=0) and the CPU oracle."""
    import functools
    from oracle import torch_ref as O
    from raygun_amd.torch import networks
    from raygun_amd.torch.systems.BaseSystem import RaygunSequential
    from raygun_amd import hipnn
    IN2 = functools.partial(nn.InstanceNorm2d)
    dev = torch.device("cuda")
    kw = dict(input_nc=1, ngf=16, n_blocks=2, n_downsampling=1, norm_layer=IN2, padding_type="reflect")
    go = nn.Sequential(O.ResNet(2, **kw), nn.Conv2d(16, 1, 1), nn.Tanh())
    sd = synth.fill_state_dict(go.state_dict(), 77)
    go.load_state_dict(sd)
    x = torch.from_numpy(synth.uniform((1, 1, 40, 44), 6))
    gy = torch.from_numpy(synth.uniform((1, 1, 40, 44), 8))
    yo = go(x); yo.backward(gy)
    res = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("RAYGUN_AMD_FUSE_TAIL", fuse)
        g = RaygunSequential(networks.ResNet(2, **kw), hipnn.Conv2d(16, 1, 1), hipnn.Tanh())
        assert list(g.state_dict()) == list(go.state_dict())
        g.load_state_dict(sd); g.to(dev)
        assert (g._linear_tail() is not None) == (fuse == "1")
        y = g(x.to(dev)); y.backward(gy.to(dev))
        res[fuse] = (y.detach().cpu(), {k: p.grad.detach().cpu() for k, p in g.named_parameters()})
    names = [k for k, _ in go.named_parameters()]
    tail_params = [k for k in names if k.startswith("1.")] + names[-4:-2]          # the 1x1 conv and the generator's last Conv2d (weight, bias)
    for fuse in ("1", "0"):
        y, grads = res[fuse]
        assert tuple(y.shape) == tuple(yo.shape) and float((y.detach() - yo.detach()).norm() / yo.detach().norm()) <= 3e-2, fuse
        for k in tail_params:       # next to the output: comparable with the fp32 oracle (deeper gradients carry bf16 ReLU-mask flips)
            ref = dict(go.named_parameters())[k].grad
            assert float((grads[k] - ref).norm() / ref.norm()) <= 6e-2, (fuse, k)
    assert float((res["1"][0] - res["0"][0]).norm() / res["0"][0].norm()) <= 2e-2
    for k in names:                 # fused vs layer by layer: the activations in front of the tail — hence every ReLU mask — are identical
        a, b = res["1"][1][k], res["0"][1][k]
        if k.endswith("weight") or k in tail_params:   # (the bias of a conv in front of an InstanceNorm has a zero gradient: rounding noise only)
            # (a bias gradient is w_b[m] * sum_v g(v): a sum with cancellation, whose bf16 rounding noise is the same factor for every channel)
            assert float((a - b).norm() / b.norm()) <= (3e-2 if k.endswith("weight") else 8e-2), k


@pytest.mark.parametrize("const_up,padding,down", [(True, "valid", "convolve"), (False, "valid", "convolve"),
                                                   (True, "same", "convolve"), (True, "valid", "max"), (False, "valid", "max")])
def test_unet_2d_vs_oracle(const_up, padding, down):
    """2-D UNet (the published CycleGAN generators are 2-D U-Nets, experiments/.../01_cycleGAN/train_conf.json): kernel rank 2
    picks the Conv2d twins (UNet.py:58), MaxPool2d, nearest / transposed-conv upsampling with crop_to_factor."""
    from oracle import torch_ref as O
    from raygun_amd.torch import networks
    kw = dict(input_nc=1, ngf=4, fmap_inc_factor=3, downsample_factors=[(2, 2), (2, 2)], constant_upsample=const_up,
              padding_type=padding, activation="Tanh", downsample_method=down)
    tol = 2e-2 if down == "convolve" else 8e-2  # max-pool argmax flips under bf16 storage (tests/test_networks_gpu.py header)
    ref, mine = O.UNet(**kw), networks.UNet(**kw)
    assert list(ref.state_dict()) == list(mine.state_dict()) and mine.ndims == 2
    sd = synth.fill_state_dict(ref.state_dict(), 61)
    ref.load_state_dict(sd); mine.load_state_dict(sd); mine.to("cuda")
    x = bf(torch.from_numpy(synth.uniform((1, 1, 92, 100), 9)))
    with torch.no_grad():
        y32 = ref(x)
    O.emulate_bf16(ref)  # same rounding points, fp32 math: max-pool decisions agree (tests/test_networks_gpu.py header)
    xr = x.clone().requires_grad_(True); xm = x.clone().cuda().requires_grad_(True)
    yr, ym = ref(xr), mine(xm)
    assert ym.dim() == 4 and tuple(ym.shape) == tuple(yr.shape)
    assert float((ym.detach().cpu() - y32).norm() / y32.norm()) <= 3e-2   # vs the fp32 reference
    assert float((ym.detach().cpu() - yr.detach()).norm() / yr.detach().norm()) <= 1e-2
    g = bf(torch.from_numpy(synth.normal(tuple(yr.shape), 3)))
    yr.backward(g); ym.backward(g.cuda())
    assert float((xm.grad.cpu() - xr.grad).norm() / xr.grad.norm()) <= tol
    for (k, a), (_, b) in zip(mine.named_parameters(), ref.named_parameters()):
        if b.grad is not None and b.grad.abs().max() > 1e-6:
            assert float((a.grad.cpu() - b.grad).norm() / b.grad.norm()) <= (3 if k.endswith('bias') else 2) * tol, k  # bias grads: cancelling sums of bf16-rounded terms

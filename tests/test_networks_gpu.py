"""GPU parity of the drop-in networks against the golden fixtures generated from the REFERENCE (fp32, CPU).

Tolerances (relative L2 unless noted), and why they differ:
  * forward outputs: <= 3e-2 — bf16 storage of every activation/weight, fp32 accumulate (measured 0.3-1.7e-2).
  * gradients of SMOOTH networks (Tanh, strided-conv downsampling): <= 4e-2 — every op of the composite backward
    (dgrad, wgrad, transposed conv, instance-norm bwd, pads, upsample/crop/concat bwd) against the fp32 reference.
  * gradients of ReLU / max-pool networks vs the fp32 reference are NOT comparable tightly: a pre-activation within
    bf16 rounding of zero flips its mask, which moves the gradient by ~sqrt(P[flip]/P[active]) in relative L2
    (5-30 % measured) however exact the kernels are.  Those are checked against the oracle in bf16-emulation mode
    (oracle/torch_ref.emulate_bf16: same rounding points, fp32 math) for single blocks, where decisions agree
    (<= 2e-2), and operator by operator in tests/test_ops_gpu.py at 1.5e-2 max-norm.
"""
import functools
import json
import os

import numpy as np
import pytest
import torch

from raygun_amd import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
IN3 = functools.partial(torch.nn.InstanceNorm3d)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def l2rel(a, b):
    a = a.detach().float().cpu().numpy().astype(np.float64) if torch.is_tensor(a) else np.asarray(a, np.float64)
    b = b.detach().float().cpu().numpy().astype(np.float64) if torch.is_tensor(b) else np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.sqrt(((a - b) ** 2).sum()) / (np.sqrt((b ** 2).sum()) + 1e-30))


def run_fixture(mod, fx, dev):
    meta = json.loads(str(fx["meta"]))
    seed, shp = meta["seed"], tuple(meta["in_shape"])
    mod.load_state_dict(synth.fill_state_dict(mod.state_dict(), seed))
    mod = mod.to(dev)
    x = T(synth.uniform(shp, seed + 1)).to(dev).requires_grad_(True)
    y = mod(x)
    ys = list(y) if isinstance(y, (list, tuple)) else [y]
    gs = [T(synth.uniform(tuple(t.shape), seed + 2 + i)).to(dev) for i, t in enumerate(ys)]
    torch.autograd.backward(ys, gs)
    return mod, x, ys, gs


def unet_kwargs(fx):
    kw = json.loads(str(fx["meta"]))["unet_kwargs"]
    kw["downsample_factors"] = [tuple(f) for f in kw["downsample_factors"]]
    return kw


def build(name):
    from raygun_amd.torch import networks as N
    from raygun_amd.torch.networks.UNet import ConvPass
    from raygun_amd.torch.networks.ResNet import ResnetBlock3D
    fx = np.load(os.path.join(G, name + ".npz"))
    dn = functools.partial(torch.nn.InstanceNorm3d, affine=False, track_running_stats=False)
    if name.startswith("unet"):
        return N.UNet(**unet_kwargs(fx)), fx
    table = {
        "convpass_valid": lambda: ConvPass(3, 5, [(3, 3, 3), (3, 3, 3)], "ReLU"),
        "convpass_same": lambda: ConvPass(3, 5, [(3, 3, 3), (3, 3, 3)], "ReLU", padding="same"),
        "convpass_residual": lambda: ConvPass(2, 4, [(3, 3, 3), (3, 3, 3)], "ReLU", residual=True),
        "convpass_sigmoid_1x1": lambda: ConvPass(12, 3, [[1, 1, 1]], "Sigmoid"),
        "resblock3d_reflect": lambda: ResnetBlock3D(6, "reflect", IN3, False, True),
        "resblock3d_valid": lambda: ResnetBlock3D(6, "valid", IN3, False, True),
        "resblock3d_zeros": lambda: ResnetBlock3D(6, "zeros", IN3, False, True),
        "resnet3d_small": lambda: N.ResNet(3, input_nc=1, ngf=8, n_blocks=2, n_downsampling=2, norm_layer=IN3, padding_type="same"),
        "resnet3d_valid_small": lambda: N.ResNet(3, input_nc=1, ngf=4, n_blocks=1, n_downsampling=1, norm_layer=IN3, padding_type="valid"),
        "resnet3d_smooth": lambda: N.ResNet(3, input_nc=1, ngf=8, n_blocks=2, n_downsampling=2, norm_layer=IN3,
                                            padding_type="same", activation=torch.nn.Tanh),
        "patchD3d_small": lambda: N.NLayerDiscriminator(3, input_nc=1, ngf=8, n_layers=3, norm_layer=dn),
    }
    return table[name](), fx


FORWARD = ["convpass_valid", "convpass_same", "convpass_residual", "convpass_sigmoid_1x1", "unet_small", "unet_same_small",
           "unet_smooth", "unet_smooth_tconv", "resblock3d_reflect", "resblock3d_valid", "resblock3d_zeros", "resnet3d_small",
           "resnet3d_valid_small", "resnet3d_smooth", "patchD3d_small"]


@pytest.mark.parametrize("name", FORWARD)
def test_forward_matches_reference(name):
    mod, fx = build(name)
    mod, x, ys, gs = run_fixture(mod, fx, torch.device("cuda"))
    for i, y in enumerate(ys):
        assert y.dtype == torch.float32 and y.is_contiguous(), "outermost module must return fp32 NCDHW like the reference"
        assert l2rel(y, fx[f"y{i}"]) <= 3e-2


@pytest.mark.parametrize("name", ["unet_smooth", "unet_smooth_tconv", "resnet3d_smooth", "convpass_sigmoid_1x1"])
def test_smooth_network_gradients_match_reference(name):
    mod, fx = build(name)
    mod, x, ys, gs = run_fixture(mod, fx, torch.device("cuda"))
    assert l2rel(x.grad, fx["gx"]) <= 4e-2
    params = dict(mod.named_parameters())
    checked = 0
    for k in fx.files:
        if k.startswith("gw::"):
            ref = fx[k]
            if np.abs(ref).max() < 1e-3:  # conv bias in front of an InstanceNorm: zero in exact arithmetic (fp32 noise in the fixture)
                continue
            assert l2rel(params[k[4:]].grad, ref) <= 4e-2, k
            checked += 1
    assert checked > 0


@pytest.mark.parametrize("name,oracle_ctor", [
    ("convpass_valid", lambda O: O.ConvPass(3, 5, [(3, 3, 3), (3, 3, 3)], "ReLU")),
    ("convpass_same", lambda O: O.ConvPass(3, 5, [(3, 3, 3), (3, 3, 3)], "ReLU", padding="same")),
    ("resblock3d_reflect", lambda O: O.ResnetBlock(3, 6, "reflect", IN3, False, True)),
    ("resblock3d_valid", lambda O: O.ResnetBlock(3, 6, "valid", IN3, False, True)),
    ("resblock3d_zeros", lambda O: O.ResnetBlock(3, 6, "zeros", IN3, False, True)),
    ("unet_same_small", None),
])
def test_relu_blocks_match_bf16_emulating_oracle(name, oracle_ctor):
    from oracle import torch_ref as O
    mod, fx = build(name)
    mod, x, ys, gs = run_fixture(mod, fx, torch.device("cuda"))
    meta = json.loads(str(fx["meta"]))
    omod = O.UNet(**unet_kwargs(fx)) if oracle_ctor is None else oracle_ctor(O)
    omod.load_state_dict(synth.fill_state_dict(omod.state_dict(), meta["seed"]))
    O.emulate_bf16(omod)
    xo = T(synth.uniform(tuple(meta["in_shape"]), meta["seed"] + 1)).to(torch.bfloat16).float().requires_grad_(True)
    yo = omod(xo)
    yo.backward(gs[0].cpu().to(torch.bfloat16).float())
    assert l2rel(ys[0], yo) <= 1e-2
    assert l2rel(x.grad, xo.grad) <= 2e-2
    po = dict(omod.named_parameters())
    for k, p in mod.named_parameters():
        if po[k].grad is not None and float(po[k].grad.abs().max()) > 1e-6:
            assert l2rel(p.grad, po[k].grad) <= 2e-2, k


def test_replicate_padding_generator_matches_bf16_emulating_oracle():
    """padding_type="replicate" (ResNet.py:268-269 / :392-393) runs on the HIP pad kernel (mode 2, edge clamp; the backward folds the border
    gradients onto the edge voxel) — checked against the bf16-emulating torch restatement, forward, input gradient and weight gradients."""
    from oracle import torch_ref as O
    from raygun_amd.torch import networks as N
    from raygun_amd import hipnn
    dev = torch.device("cuda")
    # Tanh: a smooth network, so bf16 rounding cannot flip ReLU masks and the whole-network gradients are comparable
    kw = dict(input_nc=1, ngf=8, n_blocks=2, n_downsampling=1, padding_type="replicate", activation=torch.nn.Tanh)
    net = N.ResNet(3, norm_layer=IN3, **kw)
    assert any(isinstance(m, hipnn.ReplicationPad3d) for m in net.modules())
    omod = O.ResNet(3, norm_layer=IN3, **kw)
    sd = synth.fill_state_dict(net.state_dict(), 77)
    net.load_state_dict(sd)
    omod.load_state_dict(synth.fill_state_dict(omod.state_dict(), 77))
    O.emulate_bf16(omod)
    net = net.to(dev)
    x0 = T(synth.uniform((1, 1, 20, 16, 18), 78)).to(torch.bfloat16).float()
    x = x0.to(dev).requires_grad_(True)
    y = net(x)
    g = T(synth.uniform(tuple(y.shape), 79)).to(torch.bfloat16).float()
    y.backward(g.to(dev))
    xo = x0.clone().requires_grad_(True)
    yo = omod(xo)
    yo.backward(g)
    assert l2rel(y, yo) <= 1e-2
    assert l2rel(x.grad, xo.grad) <= 2e-2
    po = dict(omod.named_parameters())
    last_bias = [k for k in po if k.endswith(".bias")][-1]
    for k, p in net.named_parameters():
        if k.endswith(".bias") and k != last_bias:
            continue  # a conv bias in front of an InstanceNorm: its gradient is zero in exact arithmetic, rounding noise in both implementations
        assert l2rel(p.grad, po[k].grad) <= 2e-2, k


def test_fused_sequence_equals_layer_by_layer():
    """FusedSequential (conv-epilogue / norm-apply fusion) must equal running its layers one by one."""
    from raygun_amd.torch import networks as N
    from raygun_amd import ops
    dev = torch.device("cuda")
    net = N.ResNet(3, input_nc=1, ngf=8, n_blocks=1, n_downsampling=1, norm_layer=IN3, padding_type="same")
    net.load_state_dict(synth.fill_state_dict(net.state_dict(), 5))
    net = net.to(dev)
    x = T(synth.uniform((1, 1, 16, 16, 16), 6)).to(dev)
    fused = net(x)
    y = ops.enter(x)
    from raygun_amd import hipnn
    with hipnn.scope():
        with hipnn.scope():
            for m in net.model:
                y = m(y)
    assert l2rel(fused, ops.leave(y)) <= 5e-3


def test_mtlsd_100_affinities_and_loss():
    """Real config U (259 146 829 parameters) at 100^3 -> 8^3 against the reference's affinities/LSDs/loss."""
    from raygun_amd.torch.models import MTLSDModel
    from raygun_amd.torch.losses import WeightedMSELoss_MTLSD
    fx = np.load(os.path.join(G, "mtlsd_100.npz"))
    dev = torch.device("cuda")
    m = MTLSDModel()
    assert sum(p.numel() for p in m.parameters()) == 259146829
    m.load_state_dict(synth.fill_state_dict(m.state_dict(), 51))
    m = m.to(dev)
    raw = T(synth.uniform((1, 1, 100, 100, 100), 52)).to(dev)
    affs, lsds = m(raw)
    gt_l = T(synth.uniform(tuple(lsds.shape), 53, 0, 1)).to(dev)
    gt_a = T(synth.uniform(tuple(affs.shape), 54, 0, 1)).to(dev)
    loss = WeightedMSELoss_MTLSD()(lsds, gt_l, torch.ones_like(lsds), affs, gt_a, torch.ones_like(affs))
    loss.backward()
    assert tuple(affs.shape) == (1, 3, 8, 8, 8) and tuple(lsds.shape) == (1, 10, 8, 8, 8)
    assert l2rel(affs, fx["affs"]) <= 1.5e-2 and l2rel(lsds, fx["lsds"]) <= 1.5e-2
    # post-sigmoid affinities, absolute: bf16 policy (BASELINE north_star asks 1e-3 rel; measured and reported in DESIGN.md)
    assert float(np.abs(affs.detach().cpu().numpy() - fx["affs"]).max()) <= 4e-2
    assert abs(float(loss.detach()) - float(fx["loss"])) <= 2e-2 * float(fx["loss"])
    P = dict(m.named_parameters())
    # last layers: gradient does not pass through deep ReLU stacks -> tight
    assert l2rel(P["aff_head.conv_pass.0.weight"].grad, fx["g_aff_w"]) <= 3e-2
    assert l2rel(P["unet.r_conv.0.0.conv_pass.2.bias"].grad, fx["g_last_bias"]) <= 4e-2
    big = P["unet.l_conv.3.conv_pass.2.weight"].grad.reshape(-1)[torch.from_numpy(fx["g_big_idx"]).to(dev)]
    assert l2rel(big, fx["g_big_val"]) <= 0.15  # 181 M-element wgrad behind 8 ReLU layers (mask flips, see module docstring)


def test_mtlsd_196_sampled():
    """The HEADLINE config (config U at 196^3 -> 104^3, the shape bench.py times) against the reference-generated fixture
    tests/golden/mtlsd_196_sampled.npz (tests/golden/make_golden.py:fx_mtlsd): 4096 sampled affinities/LSDs, their sums, the
    loss and three gradients.  This is the only place the 196^3 host dispatch (FM tile choice, split-K round counts,
    flat-plane wgrad, 192-row tiles, sub-pixel upsample-conv at the real crop offsets) is parity-checked end to end."""
    from raygun_amd.torch.models import MTLSDModel
    from raygun_amd.torch.losses import WeightedMSELoss_MTLSD
    fx = np.load(os.path.join(G, "mtlsd_196_sampled.npz"))
    dev = torch.device("cuda")
    m = MTLSDModel()
    m.load_state_dict(synth.fill_state_dict(m.state_dict(), 51))
    m = m.to(dev)
    raw = T(synth.uniform((1, 1, 196, 196, 196), 52)).to(dev)
    affs, lsds = m(raw)
    assert tuple(affs.shape) == (1, 3, 104, 104, 104) and tuple(lsds.shape) == (1, 10, 104, 104, 104)
    gt_l = T(synth.uniform(tuple(lsds.shape), 53, 0, 1)).to(dev)
    gt_a = T(synth.uniform(tuple(affs.shape), 54, 0, 1)).to(dev)
    loss = WeightedMSELoss_MTLSD()(lsds, gt_l, torch.ones_like(lsds), affs, gt_a, torch.ones_like(affs))
    loss.backward()
    flat = torch.cat([affs.detach().reshape(-1), lsds.detach().reshape(-1)])
    got = flat[torch.from_numpy(fx["out_idx"]).to(dev)].cpu().numpy()
    assert l2rel(got, fx["out_val"]) <= 1.5e-2
    assert float(np.abs(got - fx["out_val"]).max()) <= 4e-2          # post-sigmoid values in (0, 1)
    sums = np.array([float(affs.double().sum()), float(lsds.double().sum())])
    assert np.all(np.abs(sums - fx["out_sum"]) <= 2e-3 * np.abs(fx["out_sum"]))
    assert abs(float(loss.detach()) - float(fx["loss"])) <= 2e-2 * float(fx["loss"])
    P = dict(m.named_parameters())
    assert l2rel(P["aff_head.conv_pass.0.weight"].grad, fx["g_aff_w"]) <= 3e-2
    assert l2rel(P["unet.r_conv.0.0.conv_pass.2.bias"].grad, fx["g_last_bias"]) <= 4e-2
    big = P["unet.l_conv.3.conv_pass.2.weight"].grad.reshape(-1)[torch.from_numpy(fx["g_big_idx"]).to(dev)]
    assert l2rel(big, fx["g_big_val"]) <= 0.15  # 181 M-element wgrad behind 8 ReLU layers (mask flips, see module docstring)
    # every gradient tensor: its sum against the reference's (grad_sums = [sum, sum|.|] per parameter): a dispatch bug in any
    # one layer (a dropped tile, a split counted twice) moves these by O(1); bf16 mask flips move them by a few percent of sum|.|
    gs = json.loads(str(fx["grad_sums"]))
    for k, (s1, sabs) in gs.items():
        g = P[k].grad.double()
        assert abs(float(g.sum()) - s1) <= 0.08 * sabs, (k, float(g.sum()), s1, sabs)
        assert abs(float(g.abs().sum()) - sabs) <= 0.15 * sabs, (k, float(g.abs().sum()), sabs)


@pytest.mark.parametrize("size,name", [(100, "mtlsd_100"), (196, "mtlsd_196_sampled")])
def test_mtlsd_f16_storage_within_1e3_of_reference(size, name):
    """north_star: "affinities within 1e-3 rel of reference".  With IEEE-half storage (ops.precision("f16"): the same kernels
    built with v_mfma_f32_16x16x32_f16, libraygun_hip_f16.so) config U's affinities + LSDs are within 1e-3 relative L2 of the
    reference's fp32 CPU outputs at both fixture sizes (measured 8.2e-4 / 8.8e-4; bf16 storage: 6.6e-3 — one bf16 rounding is
    already 2-4e-3).  This is the block runner's default precision (predict.predict_volume)."""
    from raygun_amd import ops
    from raygun_amd.torch.models import MTLSDModel
    fx = np.load(os.path.join(G, name + ".npz"))
    dev = torch.device("cuda")
    m = MTLSDModel()
    m.load_state_dict(synth.fill_state_dict(m.state_dict(), 51))
    m = m.to(dev).eval()
    raw = T(synth.uniform((1, 1, size, size, size), 52)).to(dev)
    with torch.no_grad(), ops.precision("f16"):
        affs, lsds = m(raw)
    assert affs.dtype == torch.float32 and lsds.dtype == torch.float32
    if size == 100:
        assert l2rel(affs, fx["affs"]) <= 1e-3 and l2rel(lsds, fx["lsds"]) <= 1e-3
        assert float(np.abs(affs.cpu().numpy() - fx["affs"]).max()) <= 5e-3
    else:
        flat = torch.cat([affs.reshape(-1), lsds.reshape(-1)])
        got = flat[torch.from_numpy(fx["out_idx"]).to(dev)].cpu().numpy()
        assert l2rel(got, fx["out_val"]) <= 1e-3 and float(np.abs(got - fx["out_val"]).max()) <= 5e-3
        sums = np.array([float(affs.double().sum()), float(lsds.double().sum())])
        assert np.all(np.abs(sums - fx["out_sum"]) <= 2e-4 * np.abs(fx["out_sum"]))


def test_f16_storage_training_step_small_unet():
    """The f16 build carries every kernel (dgrad, wgrad, pointwise), so a train step runs under ops.precision("f16") too:
    gradients of a smooth small U-Net against the reference fixture, tighter than the bf16 policy's 4e-2."""
    from raygun_amd import ops
    mod, fx = build("unet_smooth")
    with ops.precision("f16"):
        mod, x, ys, gs = run_fixture(mod, fx, torch.device("cuda"))
    assert l2rel(ys[0], fx["y0"]) <= 3e-3
    assert l2rel(x.grad, fx["gx"]) <= 8e-3
    params = dict(mod.named_parameters())
    for k in fx.files:
        if k.startswith("gw::") and np.abs(fx[k]).max() >= 1e-3:
            assert l2rel(params[k[4:]].grad, fx[k]) <= 8e-3, k


@pytest.mark.parametrize("name", ["residual_unet_relu", "residual_unet_smooth", "residual_unet_plain_same"])
def test_residual_unet_matches_reference(name):
    """raygun_amd.torch.networks.ResidualUNet (residual ConvPasses with their grouped 1x1 map, `final` output
    block, input added to the output) against fixtures generated from the reference class itself
    (tests/golden/make_golden.py:fx_residual_unet): state_dict keys, forward; gradients for the smooth (Tanh, strided-conv down) variant."""
    from raygun_amd.torch.networks import ResidualUNet
    fx = np.load(os.path.join(G, name + ".npz"))
    meta = json.loads(str(fx["meta"]))
    kw = meta["kwargs"]; kw["downsample_factors"] = [tuple(f) for f in kw["downsample_factors"]]
    net = ResidualUNet(**kw)
    assert list(net.state_dict().keys()) == meta["keys"]
    net, x, ys, gs = run_fixture(net, fx, torch.device("cuda"))
    assert ys[0].dtype == torch.float32 and l2rel(ys[0], fx["y0"]) <= 3e-2
    if name == "residual_unet_smooth":   # Tanh + strided-conv downsampling: no ReLU / max-pool decisions (module docstring)
        assert l2rel(x.grad, fx["gx"]) <= 4e-2
        params = dict(net.named_parameters())
        checked = 0
        for k in fx.files:
            if k.startswith("gw::") and np.abs(fx[k]).max() >= 1e-3:
                assert l2rel(params[k[4:]].grad, fx[k]) <= 4e-2, k
                checked += 1
        assert checked > 4

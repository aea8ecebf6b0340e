"""CPU: the oracle (oracle/torch_ref.py) reproduces the fixtures recorded from the REFERENCE (tests/golden/*.npz).
This is what pins the oracle on machines where /root/reference does not exist (the GPU box); in the build container
tests/golden/make_golden.py additionally asserts oracle == reference live while generating."""
import functools
import json
import os

import numpy as np
import pytest
import torch

from oracle import torch_ref as O
from raygun_amd import synth

G = os.path.join(os.path.dirname(__file__), "golden")
IN3 = functools.partial(torch.nn.InstanceNorm3d)
DN = functools.partial(torch.nn.InstanceNorm3d, affine=False, track_running_stats=False)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def close(a, b, tol=2e-4):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape
    assert np.abs(a - b).max() <= tol * (np.abs(b).max() + 1e-12)


def unet_kw(fx):
    kw = json.loads(str(fx["meta"]))["unet_kwargs"]
    kw["downsample_factors"] = [tuple(f) for f in kw["downsample_factors"]]
    return kw


CASES = {
    "convpass_valid": lambda fx: O.ConvPass(3, 5, [(3, 3, 3), (3, 3, 3)], "ReLU"),
    "convpass_same": lambda fx: O.ConvPass(3, 5, [(3, 3, 3), (3, 3, 3)], "ReLU", padding="same"),
    "convpass_residual": lambda fx: O.ConvPass(2, 4, [(3, 3, 3), (3, 3, 3)], "ReLU", residual=True),
    "convpass_sigmoid_1x1": lambda fx: O.ConvPass(12, 3, [[1, 1, 1]], "Sigmoid"),
    "unet_small": lambda fx: O.UNet(**unet_kw(fx)),
    "unet_same_small": lambda fx: O.UNet(**unet_kw(fx)),
    "unet_smooth": lambda fx: O.UNet(**unet_kw(fx)),
    "unet_smooth_tconv": lambda fx: O.UNet(**unet_kw(fx)),
    "resblock3d_reflect": lambda fx: O.ResnetBlock(3, 6, "reflect", IN3, False, True),
    "resblock3d_valid": lambda fx: O.ResnetBlock(3, 6, "valid", IN3, False, True),
    "resblock3d_zeros": lambda fx: O.ResnetBlock(3, 6, "zeros", IN3, False, True),
    "resnet3d_small": lambda fx: O.ResNet(3, input_nc=1, ngf=8, n_blocks=2, n_downsampling=2, norm_layer=IN3, padding_type="same"),
    "resnet3d_valid_small": lambda fx: O.ResNet(3, input_nc=1, ngf=4, n_blocks=1, n_downsampling=1, norm_layer=IN3, padding_type="valid"),
    "resnet3d_smooth": lambda fx: O.ResNet(3, input_nc=1, ngf=8, n_blocks=2, n_downsampling=2, norm_layer=IN3, padding_type="same",
                                           activation=torch.nn.Tanh),
    "patchD3d_small": lambda fx: O.NLayerDiscriminator(3, input_nc=1, ngf=8, n_layers=3, norm_layer=DN),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_module_fixture(name):
    fx = np.load(os.path.join(G, name + ".npz"))
    meta = json.loads(str(fx["meta"]))
    mod = CASES[name](fx)
    mod.load_state_dict(synth.fill_state_dict(mod.state_dict(), meta["seed"]))
    x = T(synth.uniform(tuple(meta["in_shape"]), meta["seed"] + 1)).requires_grad_(True)
    y = mod(x)
    ys = list(y) if isinstance(y, (list, tuple)) else [y]
    torch.autograd.backward(ys, [T(synth.uniform(tuple(t.shape), meta["seed"] + 2 + i)) for i, t in enumerate(ys)])
    for i, t in enumerate(ys):
        close(t.detach().numpy(), fx[f"y{i}"])
    close(x.grad.numpy(), fx["gx"])
    P = dict(mod.named_parameters())
    for k in fx.files:
        if k.startswith("gw::") and np.abs(fx[k]).max() > 1e-3:
            close(P[k[4:]].grad.numpy(), fx[k], 5e-4)


def test_maxdown_upsample_instnorm_fixtures():
    fx = np.load(os.path.join(G, "maxdown_flexible_odd.npz"))
    x = T(synth.uniform((1, 3, 9, 7, 8), 21)).requires_grad_(True)
    y = O.MaxDownsample((2, 2, 2))(x)
    y.backward(T(synth.uniform((1, 3, 4, 3, 4), 22)))
    close(y.detach().numpy(), fx["y"], 0); close(x.grad.numpy(), fx["gx"], 0)
    with pytest.raises(RuntimeError):
        O.MaxDownsample((2, 2, 2), flexible=False)(x)
    fx = np.load(os.path.join(G, "upsample_crop_concat.npz"))
    f = T(synth.uniform((1, 3, 20, 20, 20), 31)).requires_grad_(True)
    g = T(synth.uniform((1, 4, 7, 7, 7), 32)).requires_grad_(True)
    up = O.Upsample(scale_factor=(2, 2, 2), mode="nearest", crop_factor=(4, 4, 4), next_conv_kernel_sizes=[(3, 3, 3), (3, 3, 3)])
    y = up(f, g)
    y.backward(T(synth.uniform((1, 7, 12, 12, 12), 33)))
    close(y.detach().numpy(), fx["y"], 0); close(f.grad.numpy(), fx["gf"], 0); close(g.grad.numpy(), fx["gg"], 1e-6)


def test_misc_fixture():
    fx = np.load(os.path.join(G, "misc.npz"))
    kw = dict(epoch_count=2, n_epochs=5, n_epochs_decay=4)
    close([O.lambda_lr_rule(e, **kw) for e in range(12)], fx["lambda_lr"], 1e-12)
    pr = T(synth.uniform((1, 3, 5, 5, 5), 101, 0, 1)); tg = T(synth.uniform((1, 3, 5, 5, 5), 102, 0, 1))
    w1 = (T(synth.uniform((1, 3, 5, 5, 5), 103, 0, 1)) > 0.4).float() * 2.0
    vals = [float(O.WeightedMSELoss_MTLSD._calc_loss(pr, tg, w)) for w in (w1, torch.zeros_like(w1))]
    close(vals, fx["wmse"], 1e-6)
    pred = T(synth.uniform((1, 1, 4, 4, 4), 104))
    close([float(O.GANLoss("lsgan")(pred, True)), float(O.GANLoss("lsgan")(pred, False))], fx["lsgan"], 1e-6)


def _cyclegan(fixture, nd, shape, gk, dk, split, loss_kwargs=None, g_lr=5e-4, d_lr=5e-4, tol=2e-4):
    fx = np.load(os.path.join(G, fixture + ".npz"))
    ref = json.loads(str(fx["loss_dicts"]))
    model, loss, _ = O.build_cyclegan(nd, gk, dk, split=split, g_lr=g_lr, d_lr=d_lr, betas=(0.5, 0.999), loss_kwargs=loss_kwargs, init=False)
    for i, net in enumerate([model.netG1, model.netG2, loss.netD1, loss.netD2]):
        net.load_state_dict(synth.fill_state_dict(net.state_dict(), 80 + i))
    for it in range(len(ref)):
        A = T(synth.uniform(shape, 90 + 2 * it)); B = T(synth.uniform(shape, 91 + 2 * it))
        O.cyclegan_step(model, loss, A, B)
        for k, v in ref[it].items():
            assert abs(float(loss.loss_dict[k]) - v) <= tol * max(abs(v), 1e-2), (it, k)
    assert list(model.state_dict().keys()) == json.loads(str(fx["meta"]))["state_keys_model"]
    return model, fx


def test_cyclegan_3d_link_and_split_steps():
    g3 = dict(input_nc=1, ngf=8, n_blocks=2, n_downsampling=2, norm_layer=IN3, padding_type="same")
    d3 = dict(input_nc=1, ngf=8, n_layers=3, kw=4)
    _cyclegan("cyclegan_step_link_3d32", 3, (1, 1, 32, 32, 32), g3, d3, False)
    _cyclegan("cyclegan_step_split_3d32", 3, (1, 1, 32, 32, 32), g3, d3, True)


def test_cyclegan_3d_full_width_ngf64_first_iteration():
    """BASELINE config 2 at its real width (ngf 64, 6 blocks, PatchD ngf 64) on 64^3: the oracle's first link iteration against the
    reference-generated fixture (one iteration here, ~40 s on 8 cores; make_golden.py asserted both while writing it)."""
    fx = np.load(os.path.join(G, "cyclegan_step_link_3d64_ngf64.npz"))
    ref = json.loads(str(fx["loss_dicts"]))[0]
    g3 = dict(input_nc=1, ngf=64, n_blocks=6, n_downsampling=2, norm_layer=IN3, padding_type="same")
    d3 = dict(input_nc=1, ngf=64, n_layers=3, kw=4)
    model, loss, _ = O.build_cyclegan(3, g3, d3, split=False, g_lr=1e-5, d_lr=1e-5, betas=(0.5, 0.999), init=False)
    for i, net in enumerate([model.netG1, model.netG2, loss.netD1, loss.netD2]):
        net.load_state_dict(synth.fill_state_dict(net.state_dict(), 80 + i))
    A = T(synth.uniform((1, 1, 64, 64, 64), 90)); B = T(synth.uniform((1, 1, 64, 64, 64), 91))
    _, outs = O.cyclegan_step(model, loss, A, B)
    for k, v in ref.items():
        assert abs(float(loss.loss_dict[k]) - v) <= 2e-4 * max(abs(v), 1e-2), k
    idx = fx["sample_idx"]
    for i, nm in enumerate(["fake_B", "cycled_B", "fake_A", "cycled_A"]):
        a, b = outs[i].detach().reshape(-1)[idx].numpy(), fx["sample_" + nm][0]
        assert np.abs(a - b).max() <= 2e-4 * np.abs(b).max(), nm
    gs = json.loads(str(fx["grad_sums"]))[0]
    for net, name in ((model.netG1, "G1"), (loss.netD2, "D2")):
        for n, p in net.named_parameters():
            assert abs(float(p.grad.double().abs().sum()) - gs[name][n][1]) <= 2e-4 * gs[name][n][1], (name, n)


def test_cyclegan_2d_256_plumbing_config0():
    """BASELINE config[0]: 2-D CycleGAN 256x256 single train step on the PyTorch CPU path (split loss, 2 iterations)."""
    IN2 = functools.partial(torch.nn.InstanceNorm2d)
    g2 = dict(input_nc=1, ngf=16, n_blocks=9, n_downsampling=0, norm_layer=IN2, padding_type="same")
    d2 = dict(input_nc=1, ngf=16, n_layers=4)
    _cyclegan("cyclegan_step_split_2d256", 2, (1, 1, 256, 256), g2, d2, True)


def test_cyclegan_2d_256_config1_as_specified():
    """SURVEY 8d "Config 1" exactly as written (tests/torch/systems/CycleGAN_conf.json:17-91, side_length 256): ngf 64, 9 blocks,
    InstanceNorm2d(track_running_stats=True, momentum=0.01), 4-layer patch D, split loss, lambda cycled 3, Adam 4e-5 / 4e-6.
    Pass = every loss_dict value of 2 iterations within 1e-5 relative of the reference-generated fixture, plus the generators'
    running statistics."""
    INr = functools.partial(torch.nn.InstanceNorm2d, track_running_stats=True, momentum=0.01)
    g2 = dict(input_nc=1, output_nc=1, ngf=64, n_blocks=9, n_downsampling=0, norm_layer=INr, padding_type="same")
    d2 = dict(input_nc=1, ngf=64, n_layers=4)
    lk = {"g_lambda_dict": {s_: {"l1_loss": {"cycled": 3, "identity": 0}, "gan_loss": {"fake": 1, "cycled": 0}} for s_ in "AB"},
          "d_lambda_dict": {s_: {"real": 1, "fake": 1, "cycled": 0} for s_ in "AB"}}
    torch.set_num_threads(8)   # the fixture was generated with 8 threads: same oneDNN reduction order
    model, fx = _cyclegan("cyclegan_step_split_2d256_conf1", 2, (1, 1, 256, 256), g2, d2, True, lk, 4e-5, 4e-6, tol=1e-5)
    rm = torch.cat([m.running_mean for m in model.netG1.modules() if "norm" in type(m).__name__.lower()])
    rv = torch.cat([m.running_var for m in model.netG1.modules() if "norm" in type(m).__name__.lower()])
    close(rm.numpy(), fx["running_mean_G1"], 1e-5); close(rv.numpy(), fx["running_var_G1"], 1e-5)


def test_mtlsd_100_forward():
    fx = np.load(os.path.join(G, "mtlsd_100.npz"))
    m = O.MTLSDModel()
    m.load_state_dict(synth.fill_state_dict(m.state_dict(), 51))
    with torch.no_grad():
        affs, lsds = m(T(synth.uniform((1, 1, 100, 100, 100), 52)))
    close(affs.numpy(), fx["affs"]); close(lsds.numpy(), fx["lsds"])


def test_oracle_vs_reference_live():
    """Re-check a few classes against the reference itself whenever it is mounted (build container only)."""
    from oracle import ref_import
    if not ref_import.available():
        pytest.skip("/root/reference not present (GPU box)")
    ref_import.install()
    from raygun.torch import networks as RN
    kw = dict(input_nc=1, ngf=4, fmap_inc_factor=2, downsample_factors=[(2, 2, 2)] * 2, constant_upsample=True)
    r, o = RN.UNet(**kw), O.UNet(**kw)
    sd = synth.fill_state_dict(r.state_dict(), 3)
    r.load_state_dict(sd); o.load_state_dict(sd)
    x = T(synth.uniform((1, 1, 44, 44, 44), 4))
    close(o(x).detach().numpy(), r(x).detach().numpy(), 1e-6)
    d_r, d_o = RN.NLayerDiscriminator(3, ngf=4, norm_layer=DN), O.NLayerDiscriminator(3, ngf=4, norm_layer=DN)
    sd = synth.fill_state_dict(d_r.state_dict(), 5)
    d_r.load_state_dict(sd); d_o.load_state_dict(sd)
    x = T(synth.uniform((1, 1, 32, 32, 32), 6))
    close(d_o(x).detach().numpy(), d_r(x).detach().numpy(), 1e-6)

"""Generate tests/golden/*.npz from the REFERENCE's own modules (build container only).

    python tests/golden/make_golden.py [--only name,name] [--skip-big]

Inputs and weights come from raygun_amd.synth (counter-based hash), so fixtures hold only the
reference's OUTPUTS plus a little metadata; any machine can regenerate the inputs.  While
generating, every fixture is also recomputed with the CPU restatement (oracle/torch_ref.py,
and oracle/mws*.py) and asserted equal — that is what pins the oracle.
"""
import argparse
import functools
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402
from oracle import torch_ref as O  # noqa: E402
from raygun_amd import synth  # noqa: E402

ref_import.install()
from raygun.torch import networks as RN  # noqa: E402
from raygun.torch.networks.UNet import ConvPass as RConvPass, MaxDownsample as RMaxDown, Upsample as RUpsample  # noqa: E402
from raygun.torch.networks.ResNet import ResnetBlock3D as RResBlock3D  # noqa: E402
from raygun.torch.models import MTLSDModel as RMTLSD  # noqa: E402
from raygun.torch.losses.WeightedMSELoss_MTLSD import WeightedMSELoss_MTLSD as RWMSE  # noqa: E402
from raygun.torch.losses import GANLoss as RGANLoss  # noqa: E402
from raygun.torch.systems import CycleGAN as RCycleGAN  # noqa: E402
from raygun.torch.optimizers import BaseDummyOptimizer as RDummyOpt  # noqa: E402

torch.set_num_threads(8)
IN3 = functools.partial(torch.nn.InstanceNorm3d)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"  wrote {name}.npz ({os.path.getsize(path) / 1e3:.1f} kB)")


def load_w(mod, seed):
    sd = synth.fill_state_dict(mod.state_dict(), seed)
    mod.load_state_dict(sd)
    return sd


def same_weights(ref, ora, seed):
    load_w(ref, seed)
    assert list(ref.state_dict().keys()) == list(ora.state_dict().keys()), "state_dict keys differ"
    ora.load_state_dict(ref.state_dict())


def close(a, b, tol=1e-5, what=""):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b).max() / (np.abs(a).max() + 1e-30)
    assert err <= tol, f"oracle != reference for {what}: rel err {err:.3e}"
    return err


def fwd_bwd(mod, x, gseed):
    x = x.clone().requires_grad_(True)
    y = mod(x)
    ys = y if isinstance(y, (list, tuple)) else [y]
    gs = [T(synth.uniform(tuple(t.shape), gseed + i)) for i, t in enumerate(ys)]
    torch.autograd.backward(ys, gs)
    return [t.detach().numpy() for t in ys], x.grad.numpy(), {k: p.grad.numpy() for k, p in mod.named_parameters()
                                                              if p.grad is not None}


def module_fixture(name, ref, ora, in_shape, seed, keep_wgrads=None, extra=None):
    same_weights(ref, ora, seed)
    x = T(synth.uniform(in_shape, seed + 1))
    ry, rgx, rgw = fwd_bwd(ref, x, seed + 2)
    oy, ogx, ogw = fwd_bwd(ora, x, seed + 2)
    for i, (a, b) in enumerate(zip(ry, oy)):
        close(a, b, what=f"{name}.y{i}")
    close(rgx, ogx, what=name + ".gx")
    for k in rgw:
        close(rgw[k], ogw[k], 2e-5, what=f"{name}.gw[{k}]")
    keep = list(rgw) if keep_wgrads is None else keep_wgrads
    arrs = {f"y{i}": a for i, a in enumerate(ry)}
    arrs["gx"] = rgx
    for k in keep:
        arrs["gw::" + k] = rgw[k]
    arrs["meta"] = json.dumps(dict(in_shape=list(in_shape), seed=seed, **(extra or {})))
    save(name, **arrs)


# ------------------------------------------------------------------ fixtures
def fx_convpass():
    module_fixture("convpass_valid", RConvPass(3, 5, [(3, 3, 3), (3, 3, 3)], "ReLU"),
                   O.ConvPass(3, 5, [(3, 3, 3), (3, 3, 3)], "ReLU"), (1, 3, 10, 10, 10), 11)
    module_fixture("convpass_same", RConvPass(3, 5, [(3, 3, 3), (3, 3, 3)], "ReLU", padding="same"),
                   O.ConvPass(3, 5, [(3, 3, 3), (3, 3, 3)], "ReLU", padding="same"), (1, 3, 10, 10, 10), 12)
    module_fixture("convpass_residual", RConvPass(2, 4, [(3, 3, 3), (3, 3, 3)], "ReLU", residual=True),
                   O.ConvPass(2, 4, [(3, 3, 3), (3, 3, 3)], "ReLU", residual=True), (1, 2, 10, 10, 10), 13)
    module_fixture("convpass_sigmoid_1x1", RConvPass(12, 3, [[1, 1, 1]], "Sigmoid"),
                   O.ConvPass(12, 3, [[1, 1, 1]], "Sigmoid"), (1, 12, 6, 6, 6), 14)


def fx_maxdown():
    x = T(synth.uniform((1, 3, 9, 7, 8), 21))
    g = T(synth.uniform((1, 3, 4, 3, 4), 22))
    outs = []
    for M in (RMaxDown, O.MaxDownsample):
        xi = x.clone().requires_grad_(True)
        y = M((2, 2, 2))(xi)
        y.backward(g)
        outs.append((y.detach().numpy(), xi.grad.numpy()))
    close(outs[0][0], outs[1][0], 0, "maxdown.y")
    close(outs[0][1], outs[1][1], 0, "maxdown.gx")
    save("maxdown_flexible_odd", y=outs[0][0], gx=outs[0][1])


def fx_upsample():
    # 56 -> 112 -> crop_to_factor 108 (factor 8, two 3^3 convs) + skip 192->108 scaled down: use small analogue
    # g: (1,4,7,7,7) up x2 -> 14 ; crop factor 4, convs 2x(3,3,3): c=4, n=floor((14-4)/4)=2 -> 12 ; skip 20 -> 12
    kw = dict(scale_factor=(2, 2, 2), mode="nearest", crop_factor=(4, 4, 4), next_conv_kernel_sizes=[(3, 3, 3), (3, 3, 3)])
    f = T(synth.uniform((1, 3, 20, 20, 20), 31))
    g = T(synth.uniform((1, 4, 7, 7, 7), 32))
    go = T(synth.uniform((1, 7, 12, 12, 12), 33))
    res = []
    for M in (RUpsample, O.Upsample):
        fi, gi = f.clone().requires_grad_(True), g.clone().requires_grad_(True)
        y = M(**kw)(fi, gi)
        y.backward(go)
        res.append((y.detach().numpy(), fi.grad.numpy(), gi.grad.numpy()))
    for a, b, n in zip(res[0], res[1], ("y", "gf", "gg")):
        close(a, b, 0, "upsample." + n)
    save("upsample_crop_concat", y=res[0][0], gf=res[0][1], gg=res[0][2])


def fx_unet_small():
    kw = dict(input_nc=1, ngf=4, fmap_inc_factor=2, downsample_factors=[(2, 2, 2)] * 3, constant_upsample=True)
    module_fixture("unet_small", RN.UNet(**kw), O.UNet(**kw), (1, 1, 100, 100, 100), 41,
                   keep_wgrads=["l_conv.0.conv_pass.0.weight", "l_conv.3.conv_pass.2.weight", "r_conv.0.0.conv_pass.0.bias",
                                "r_conv.0.1.conv_pass.0.weight"], extra=dict(unet_kwargs=kw))
    kw2 = dict(input_nc=3, ngf=4, fmap_inc_factor=3, downsample_factors=[(2, 2, 2)] * 2, padding_type="same", output_nc=6)
    module_fixture("unet_same_small", RN.UNet(**kw2), O.UNet(**kw2), (1, 3, 16, 16, 16), 42,
                   keep_wgrads=["l_conv.0.conv_pass.0.weight"], extra=dict(unet_kwargs=kw2))


def fx_smooth():
    """Networks with smooth activations and strided-conv downsampling: no ReLU/max-pool decisions, so the bf16 GPU
    gradients can be compared with the fp32 reference at a tight tolerance (see tests/test_networks_gpu.py)."""
    kw = dict(input_nc=2, ngf=4, fmap_inc_factor=2, downsample_factors=[(2, 2, 2)] * 2, activation="Tanh",
              downsample_method="convolve", constant_upsample=True)
    module_fixture("unet_smooth", RN.UNet(**kw), O.UNet(**kw), (1, 2, 62, 62, 62), 111, extra=dict(unet_kwargs=kw))
    kw2 = dict(kw, constant_upsample=False, num_heads=2)
    module_fixture("unet_smooth_tconv", RN.UNet(**kw2), O.UNet(**kw2), (1, 2, 62, 62, 62), 112, extra=dict(unet_kwargs=kw2))
    kg = dict(input_nc=1, ngf=8, n_blocks=2, n_downsampling=2, norm_layer=IN3, padding_type="same", activation=torch.nn.Tanh)
    module_fixture("resnet3d_smooth", RN.ResNet(3, **kg), O.ResNet(3, **kg), (1, 1, 24, 24, 24), 113)


def _mtlsd_run(model, loss_fn, raw, seed, lr=None):
    affs, lsds = model(raw)
    gt_l = T(synth.uniform(tuple(lsds.shape), seed + 3, 0, 1))
    gt_a = T(synth.uniform(tuple(affs.shape), seed + 4, 0, 1))
    loss = loss_fn(lsds, gt_l, torch.ones_like(lsds), affs, gt_a, torch.ones_like(affs))
    loss.backward()
    return affs.detach().numpy(), lsds.detach().numpy(), float(loss)


def fx_mtlsd(size, name, sampled):
    t0 = time.time()
    ref, ora = RMTLSD(), O.MTLSDModel()
    same_weights(ref, ora, 51)
    raw = T(synth.uniform((1, 1, size, size, size), 52))
    ra, rl, rloss = _mtlsd_run(ref, RWMSE(), raw, 50)
    oa, ol, oloss = _mtlsd_run(ora, O.WeightedMSELoss_MTLSD(), raw, 50)
    close(ra, oa, what=name + ".affs"); close(rl, ol, what=name + ".lsds"); close(rloss, oloss, what=name + ".loss")
    rg = {k: p.grad for k, p in ref.named_parameters()}
    og = {k: p.grad for k, p in ora.named_parameters()}
    gsum = {}
    for k in rg:
        close(rg[k].numpy(), og[k].numpy(), 5e-5, what=f"{name}.g[{k}]")
        gsum[k] = [float(rg[k].double().sum()), float(rg[k].double().abs().sum())]
    arrs = dict(loss=rloss, grad_sums=json.dumps(gsum),
                g_first=rg["unet.l_conv.0.conv_pass.0.weight"].numpy(),
                g_last_bias=rg["unet.r_conv.0.0.conv_pass.2.bias"].numpy(),
                g_aff_w=rg["aff_head.conv_pass.0.weight"].numpy())
    # a strided sample of the biggest weight gradient (181 M elements)
    big = rg["unet.l_conv.3.conv_pass.2.weight"].reshape(-1)
    idx = np.arange(0, big.numel(), max(1, big.numel() // 4096))[:4096]
    arrs["g_big_idx"], arrs["g_big_val"] = idx, big[idx].numpy()
    if sampled:
        flat = np.concatenate([ra.reshape(-1), rl.reshape(-1)])
        idx = (np.arange(4096, dtype=np.int64) * 2654435761) % flat.size
        arrs["out_idx"], arrs["out_val"] = idx, flat[idx]
        arrs["out_sum"] = np.array([ra.astype(np.float64).sum(), rl.astype(np.float64).sum()])
    else:
        arrs["affs"], arrs["lsds"] = ra, rl
    arrs["meta"] = json.dumps(dict(size=size, seed=51, secs=time.time() - t0))
    save(name, **arrs)


def fx_resnet():
    for pt, shape, out in (("reflect", (1, 6, 8, 12, 12), None), ("valid", (1, 6, 8, 12, 12), None),
                           ("zeros", (1, 6, 8, 12, 12), None)):
        r = RResBlock3D(6, pt, IN3, False, True)
        o = O.ResnetBlock(3, 6, pt, IN3, False, True)
        module_fixture(f"resblock3d_{pt}", r, o, shape, 61)
    kw = dict(input_nc=1, ngf=8, n_blocks=2, n_downsampling=2, norm_layer=IN3, padding_type="same")
    module_fixture("resnet3d_small", RN.ResNet(3, **kw), O.ResNet(3, **kw), (1, 1, 32, 32, 32), 62,
                   keep_wgrads=["model.1.weight", "model.10.conv_block.1.weight", "model.12.weight", "model.19.weight",
                                "model.19.bias", "model.15.bias"])
    kwv = dict(input_nc=1, ngf=4, n_blocks=1, n_downsampling=1, norm_layer=IN3, padding_type="valid")
    module_fixture("resnet3d_valid_small", RN.ResNet(3, **kwv), O.ResNet(3, **kwv), (1, 1, 40, 40, 40), 63,
                   keep_wgrads=["model.0.weight"])
    dn = functools.partial(torch.nn.InstanceNorm3d, affine=False, track_running_stats=False)
    module_fixture("patchD3d_small", RN.NLayerDiscriminator(3, input_nc=1, ngf=8, n_layers=3, norm_layer=dn),
                   O.NLayerDiscriminator(3, input_nc=1, ngf=8, n_layers=3, norm_layer=dn), (1, 1, 48, 48, 48), 64)


def fx_instnorm():
    x = T(synth.normal((2, 5, 6, 7, 8), 71)) * 3 + 1
    g = T(synth.uniform((2, 5, 6, 7, 8), 72))
    xi = x.clone().requires_grad_(True)
    y = torch.nn.InstanceNorm3d(5)(xi)
    y.backward(g)
    save("instnorm3d", y=y.detach().numpy(), gx=xi.grad.numpy())


def _cyc_cfg(nd, gk, dk, loss_type, lr=5e-4, d_lr=None):
    return dict(common_voxel_size=[1] * 3, ndims=nd, gnet_type="resnet", gnet_kwargs=gk, dnet_kwargs=dk,
                loss_type=loss_type, loss_kwargs={},
                g_optim_kwargs=dict(betas=[0.5, 0.999], lr=lr), d_optim_kwargs=dict(betas=[0.5, 0.999], lr=lr if d_lr is None else d_lr))


def _grad_sums(nets):
    """{net: {param: [sum, sum|.|]}} of the gradients the step left in .grad (G nets: the generator phase's, D nets: the
    discriminator phase's — LinkCycleLoss.py:121-135,61-85 zero them per phase, nothing clears them afterwards)."""
    return {k: {n: [float(p.grad.double().sum()), float(p.grad.double().abs().sum())] for n, p in net.named_parameters() if p.grad is not None}
            for k, net in nets.items()}


def fx_cyclegan(tag, nd, shape, gk, dk, loss_type, loss_kwargs=None, iters=2, g_lr=5e-4, d_lr=5e-4, sampled=False):
    cfg = _cyc_cfg(nd, dict(gk), dict(dk), loss_type, g_lr, d_lr)
    if loss_kwargs:
        cfg["loss_kwargs"] = loss_kwargs
    rs = RCycleGAN(cfg)
    rs.setup_networks(); rs.setup_model(); rs.setup_optimization()
    model, loss, opts = O.build_cyclegan(nd, dict(gk), dict(dk), split=(loss_type == "split"),
                                         g_lr=g_lr, d_lr=d_lr, betas=(0.5, 0.999), loss_kwargs=loss_kwargs, init=False)
    nets_r = dict(G1=rs.netG1, G2=rs.netG2, D1=rs.netD1, D2=rs.netD2)
    nets_o = dict(G1=model.netG1, G2=model.netG2, D1=loss.netD1, D2=loss.netD2)
    for i, k in enumerate(nets_r):
        same_weights(nets_r[k], nets_o[k], 80 + i)
    recs, samp_recs = [], []
    for it in range(iters):
        A = T(synth.uniform(shape, 90 + 2 * it)); B = T(synth.uniform(shape, 91 + 2 * it))
        rs.optimizer.zero_grad()
        fB, cB, fA, cA = rs.model(real_A=A, real_B=B)
        tot = rs.loss(A, fA, cA, B, fB, cB); tot.backward(); rs.optimizer.step()
        rd = {k: float(v) for k, v in rs.loss.loss_dict.items()}
        ot, outs = O.cyclegan_step(model, loss, A, B)
        od = {k: float(v) for k, v in loss.loss_dict.items()}
        assert set(rd) == set(od), (set(rd) ^ set(od))
        for k in rd:
            close(rd[k], od[k], 2e-5, f"{tag}.it{it}.{k}")
        close(fB.detach().numpy(), outs[0].detach().numpy(), 2e-5, f"{tag}.it{it}.fake_B")
        recs.append(rd)
        if sampled:   # full-width configurations: sampled outputs + sum / sum|.| of every gradient tensor, per iteration
            gr, go = _grad_sums(nets_r), _grad_sums(nets_o)
            for k in gr:
                for n in gr[k]:
                    close(gr[k][n][1], go[k][n][1], 1e-4, f"{tag}.it{it}.gradabs.{k}.{n}")
            flat = {"fake_B": fB, "cycled_B": cB, "fake_A": fA, "cycled_A": cA}
            for nm, (t_ref, t_ora) in {nm: (flat[nm], outs[i]) for i, nm in enumerate(["fake_B", "cycled_B", "fake_A", "cycled_A"])}.items():
                close(t_ref.detach().numpy(), t_ora.detach().numpy(), 2e-5, f"{tag}.it{it}.{nm}")
            idx = (np.arange(4096, dtype=np.int64) * 2654435761) % fB.numel()
            samp_recs.append(dict(grad_sums=gr, idx=idx, **{nm: t.detach().reshape(-1)[idx].numpy() for nm, t in flat.items()},
                                  **{nm + "_sum": [float(t.double().sum()), float(t.double().abs().sum())] for nm, t in flat.items()}))
    wsum = {}
    for k in nets_r:
        for n, p in nets_r[k].named_parameters():
            q = dict(nets_o[k].named_parameters())[n]
            close(p.detach().numpy(), q.detach().numpy(), 5e-5, f"{tag}.w.{k}.{n}")
        wsum[k] = [float(sum(p.double().sum() for p in nets_r[k].parameters())),
                   float(sum(p.double().abs().sum() for p in nets_r[k].parameters()))]
    from raygun.torch.networks.utils import get_running_norm_stats
    rm, rv = get_running_norm_stats(rs.netG1)
    extra = {} if rm is None else dict(running_mean_G1=rm.numpy(), running_var_G1=rv.numpy())
    if sampled:
        extra["grad_sums"] = json.dumps([r["grad_sums"] for r in samp_recs])
        extra["out_sums"] = json.dumps([{k: v for k, v in r.items() if k.endswith("_sum")} for r in samp_recs])
        extra["sample_idx"] = samp_recs[0]["idx"]
        for nm in ("fake_B", "cycled_B", "fake_A", "cycled_A"):
            extra["sample_" + nm] = np.stack([r[nm] for r in samp_recs])
    save(tag, loss_dicts=json.dumps(recs), weight_sums=json.dumps(wsum), **({} if sampled else dict(fake_B=fB.detach().numpy())), **extra,
         meta=json.dumps(dict(shape=list(shape), loss_type=loss_type, lr=g_lr, d_lr=d_lr, betas=[0.5, 0.999], iters=iters,
                              gk={k: str(v) for k, v in gk.items()}, dk={k: str(v) for k, v in dk.items()},
                              state_keys_model=list(rs.model.state_dict().keys()))))


def fx_cyclegans(big):
    IN2 = functools.partial(torch.nn.InstanceNorm2d)
    g3 = dict(input_nc=1, ngf=8, n_blocks=2, n_downsampling=2, norm_layer=IN3, padding_type="same")
    d3 = dict(input_nc=1, ngf=8, n_layers=3, kw=4)
    fx_cyclegan("cyclegan_step_link_3d32", 3, (1, 1, 32, 32, 32), g3, d3, "link")
    fx_cyclegan("cyclegan_step_split_3d32", 3, (1, 1, 32, 32, 32), g3, d3, "split")
    ident = {k: {s: {"l1_loss": {"cycled": 10, "identity": 0.5}, "gan_loss": {"fake": 1, "cycled": 0}} for s in "AB"}
             for k in ["g_lambda_dict"]}
    fx_cyclegan("cyclegan_step_link_identity_3d32", 3, (1, 1, 32, 32, 32), g3, d3, "link", loss_kwargs=ident, iters=1)
    # BASELINE config[0]: 2D 256x256 plumbing case (tests/torch/systems/CycleGAN_conf.json shapes, ngf reduced unless --big)
    ngf = 64 if big else 16
    g2 = dict(input_nc=1, ngf=ngf, n_blocks=9, n_downsampling=0, norm_layer=IN2, padding_type="same")
    d2 = dict(input_nc=1, ngf=ngf, n_layers=4)
    fx_cyclegan("cyclegan_step_split_2d256", 2, (1, 1, 256, 256), g2, d2, "split")
    fx_cyclegan("cyclegan_step_link_2d256", 2, (1, 1, 256, 256), g2, d2, "link", iters=1)


def fx_cyclegan_conf1():
    """BASELINE config[0] / SURVEY 8d "Config 1" as specified: tests/torch/systems/CycleGAN_conf.json:17-91 with side_length 256 —
    resnet generators ngf 64 / 9 blocks / no downsampling / InstanceNorm2d(track_running_stats=True, momentum=0.01) / "same",
    4-layer patch discriminators ngf 64, split loss, lambda cycled 3 / identity 0, Adam G lr 4e-5, D lr 4e-6, betas (0.5, 0.999)."""
    INr = functools.partial(torch.nn.InstanceNorm2d, track_running_stats=True, momentum=0.01)
    g2 = dict(input_nc=1, output_nc=1, ngf=64, n_blocks=9, n_downsampling=0, norm_layer=INr, padding_type="same")
    d2 = dict(input_nc=1, ngf=64, n_layers=4)
    lk = {"g_lambda_dict": {s_: {"l1_loss": {"cycled": 3, "identity": 0}, "gan_loss": {"fake": 1, "cycled": 0}} for s_ in "AB"},
          "d_lambda_dict": {s_: {"real": 1, "fake": 1, "cycled": 0} for s_ in "AB"}}
    fx_cyclegan("cyclegan_step_split_2d256_conf1", 2, (1, 1, 256, 256), g2, d2, "split", loss_kwargs=lk, iters=2, g_lr=4e-5, d_lr=4e-6)


def fx_misc():
    # LambdaLR rule through the reference's BaseDummyOptimizer
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=1.0)
    kw = dict(epoch_count=2, n_epochs=5, n_epochs_decay=4)
    d = RDummyOpt(scheduler="LambdaLR", scheduler_kwargs=kw, optimizer_G=opt)
    lrs = []
    for _ in range(12):
        lrs.append(opt.param_groups[0]["lr"]); opt.step(); d.step()
    exp = [O.lambda_lr_rule(e, **kw) for e in range(12)]
    close(lrs, exp, 1e-12, "lambda_lr")
    # weighted MSE: both branches of WeightedMSELoss_MTLSD.py:15-23
    pr = T(synth.uniform((1, 3, 5, 5, 5), 101, 0, 1)); tg = T(synth.uniform((1, 3, 5, 5, 5), 102, 0, 1))
    w1 = (T(synth.uniform((1, 3, 5, 5, 5), 103, 0, 1)) > 0.4).float() * 2.0
    w0 = torch.zeros_like(w1)
    vals = []
    for w in (w1, w0):
        a = float(RWMSE()._calc_loss(pr, tg, w)); b = float(O.WeightedMSELoss_MTLSD._calc_loss(pr, tg, w))
        close(a, b, 1e-7, "wmse"); vals.append(a)
    # lsgan
    pred = T(synth.uniform((1, 1, 4, 4, 4), 104))
    gl = [float(RGANLoss("lsgan")(pred, True)), float(RGANLoss("lsgan")(pred, False))]
    close(gl, [float(O.GANLoss("lsgan")(pred, True)), float(O.GANLoss("lsgan")(pred, False))], 1e-7, "ganloss")
    save("misc", lambda_lr=np.array(lrs), wmse=np.array(vals), lsgan=np.array(gl))


def fx_structure():
    out = {}
    m = RMTLSD()
    out["mtlsd_keys"] = {k: list(v.shape) for k, v in m.state_dict().items()}
    out["mtlsd_params"] = sum(p.numel() for p in m.parameters())
    out["mtlsd20_params"] = sum(p.numel() for p in RMTLSD(num_affs=20).parameters())
    cfg = _cyc_cfg(3, dict(input_nc=1, ngf=64, n_blocks=6, n_downsampling=2, norm_layer=IN3, padding_type="same"),
                   dict(input_nc=1, ngf=64, n_layers=3, kw=4), "link")
    s = RCycleGAN(cfg)
    s.setup_networks(); s.setup_model()
    out["cyclemodel_keys"] = {k: list(v.shape) for k, v in s.model.state_dict().items()}
    out["G_params"] = sum(p.numel() for p in s.netG1.parameters())
    out["D_params"] = sum(p.numel() for p in s.netD1.parameters())
    out["D_keys"] = {k: list(v.shape) for k, v in s.netD1.state_dict().items()}
    out["D_bn_params"] = sum(p.numel() for p in RN.NLayerDiscriminator(3).parameters())
    out["len_G_model"] = len(s.netG1[0].model)
    out["len_resnet_down3_blocks4"] = len(RN.ResNet(3, n_downsampling=3, n_blocks=4).model)
    out["len_D_model"] = len(s.netD1.model)
    out["unet_io"] = {str(n): 8 * ((n - 92) // 8) for n in (100, 132, 196)}
    # same numbers from the restatement
    om = O.MTLSDModel()
    assert {k: list(v.shape) for k, v in om.state_dict().items()} == out["mtlsd_keys"]
    og = O.get_network(3, "resnet", cfg["gnet_kwargs"], init=False)
    od = O.get_network(3, "patchdiscriminator", cfg["dnet_kwargs"], init=False)
    assert sum(p.numel() for p in og.parameters()) == out["G_params"] == 24876225
    assert sum(p.numel() for p in od.parameters()) == out["D_params"] == 11047873
    assert out["mtlsd_params"] == 259146829 and out["D_bn_params"] == 11048769
    with open(os.path.join(HERE, "structure.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("  wrote structure.json")


def fx_residual_unet():
    """raygun.torch.networks.ResidualUNet (ResidualUNet.py:364-701) — no oracle restatement exists for it: the fixtures hold the
    REFERENCE's outputs and gradients, the GPU build is compared with them directly (tests/test_networks_gpu.py)."""
    from raygun.torch.networks.ResidualUNet import ResidualUNet as RResUNet
    for name, kw, shape, seed in (
            ("residual_unet_relu", dict(input_nc=1, ngf=4, fmap_inc_factor=2, downsample_factors=[(2, 2, 2)], constant_upsample=True,
                                        residual=True), (1, 1, 36, 36, 36), 91),
            # smooth variants (Tanh, strided-conv downsampling: no ReLU / max-pool decisions) carry the gradient comparison
            ("residual_unet_smooth", dict(input_nc=4, ngf=4, fmap_inc_factor=3, downsample_factors=[(2, 2, 2)], constant_upsample=True,
                                          residual=True, activation="Tanh", downsample_method="convolve"), (1, 4, 38, 38, 38), 92),
            ("residual_unet_plain_same", dict(input_nc=2, ngf=8, fmap_inc_factor=2, downsample_factors=[(2, 2, 2)], constant_upsample=True,
                                              residual=False, padding_type="same", output_nc=2, activation="Tanh"), (1, 2, 16, 16, 16), 93)):
        ref = RResUNet(**kw)
        load_w(ref, seed)
        x = T(synth.uniform(shape, seed + 1))
        ry, rgx, rgw = fwd_bwd(ref, x, seed + 2)
        arrs = {f"y{i}": a for i, a in enumerate(ry)}
        arrs["gx"] = rgx
        for k in rgw:
            arrs["gw::" + k] = rgw[k]
        kwj = dict(kw); kwj["downsample_factors"] = [list(f) for f in kw["downsample_factors"]]
        arrs["meta"] = json.dumps(dict(in_shape=list(shape), seed=seed, kwargs=kwj, keys=list(ref.state_dict().keys())))
        save(name, **arrs)


def fx_cyclegan_ngf64():
    """BASELINE config 2 at its REAL width (ResNet ngf 64 / 6 blocks / 2 down / InstanceNorm3d / "same" + PatchD ngf 64 /
    3 layers / kw 4, link loss, lsgan, lambda cycled 10) on 64^3 volumes (128^3 does not fit this container's 64 GB next to the
    oracle's copy): 2 iterations — loss_dict, 4096 sampled values + sums of all four generator outputs, sum / sum|.| of every
    gradient tensor of G1, G2, D1, D2.  The ngf-8 fixtures above never reach the 256-channel kernels, the split-K PatchGAN layers
    or the composed 64 -> 64 7^3 tail."""
    g3 = dict(input_nc=1, ngf=64, n_blocks=6, n_downsampling=2, norm_layer=IN3, padding_type="same")
    d3 = dict(input_nc=1, ngf=64, n_layers=3, kw=4)
    fx_cyclegan("cyclegan_step_link_3d64_ngf64", 3, (1, 1, 64, 64, 64), g3, d3, "link", iters=2, g_lr=1e-5, d_lr=1e-5, sampled=True)   # config 2's Adam lr (default_cycleGAN_conf.json:33-45)


FIXTURES = {
    "cyclegan_ngf64": fx_cyclegan_ngf64,
    "residual_unet": fx_residual_unet,
    "convpass": fx_convpass, "maxdown": fx_maxdown, "upsample": fx_upsample, "unet_small": fx_unet_small,
    "resnet": fx_resnet, "smooth": fx_smooth, "instnorm": fx_instnorm, "misc": fx_misc, "structure": fx_structure,
    "cyclegan": lambda: fx_cyclegans(False),
    "cyclegan_conf1": fx_cyclegan_conf1,
    "mtlsd_100": lambda: fx_mtlsd(100, "mtlsd_100", False),
    "mtlsd_196": lambda: fx_mtlsd(196, "mtlsd_196_sampled", True),
}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--skip-big", action="store_true")
    a = ap.parse_args()
    names = [n for n in a.only.split(",") if n] or list(FIXTURES)
    for n in names:
        if a.skip_big and n in ("mtlsd_196",):
            continue
        print(f"[{n}]"); t0 = time.time()
        with torch.enable_grad():
            FIXTURES[n]()
        print(f"  {time.time() - t0:.1f}s")

"""GPU parity of every operator (through the C-ABI) against plain torch fp32 on the CPU.

Inputs/weights are rounded to bf16 first so that the only differences are accumulation order and
the bf16 rounding of the stored result: tolerance 1.5e-2 of the tensor's max-abs (bf16 has 8
significant bits: one output rounding is <= 2^-9 relative; gradients pass through 2-3 roundings).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from raygun_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1.5e-2


def bf(t):
    return t.to(torch.bfloat16).float()


def rnd(shape, seed, lo=-1.0, hi=1.0):
    return bf(torch.from_numpy(synth.uniform(shape, seed, lo, hi)))


def relerr(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def run_pair(hip_fn, ref_fn, inputs, params, gseed=7):
    """inputs/params: lists of CPU fp32 tensors. Returns errors for output, input grads, param grads."""
    dev = torch.device("cuda")
    xi = [t.clone().to(dev).requires_grad_(True) for t in inputs]
    pi = [t.clone().to(dev).requires_grad_(True) for t in params]
    y = hip_fn(*xi, *pi)
    xr = [t.clone().requires_grad_(True) for t in inputs]
    pr = [t.clone().requires_grad_(True) for t in params]
    yr = ref_fn(*xr, *pr)
    g = rnd(tuple(yr.shape), gseed)
    y.backward(g.to(dev).to(y.dtype))
    yr.backward(g)
    errs = {"y": relerr(y, yr)}
    for i, (a, b) in enumerate(zip(xi, xr)):
        errs[f"gx{i}"] = relerr(a.grad, b.grad)
    for i, (a, b) in enumerate(zip(pi, pr)):
        errs[f"gp{i}"] = relerr(a.grad, b.grad)
    return errs


def check(errs, tol=TOL):
    bad = {k: v for k, v in errs.items() if not v <= tol}
    assert not bad, f"errors above {tol}: {bad} (all: {errs})"


CONV_CASES = [
    # cin, cout, k, stride, pad, spatial, act
    (3, 5, 3, 1, 0, (10, 9, 11), "relu"),
    (1, 12, 3, 1, 0, (12, 12, 12), "relu"),
    (12, 12, 3, 1, 0, (9, 10, 11), "none"),
    (72, 40, 3, 1, 0, (7, 8, 9), "relu"),
    (40, 144, 3, 1, 1, (6, 6, 6), "leaky"),
    (16, 8, 1, 1, 0, (5, 6, 7), "sigmoid"),
    (8, 16, 3, 2, 1, (9, 10, 12), "none"),
    (1, 8, 4, 2, 1, (12, 12, 12), "leaky"),
    (8, 8, 4, 1, 1, (7, 7, 7), "none"),
    (4, 6, 7, 1, 0, (9, 9, 10), "none"),
    (200, 24, 3, 1, 0, (5, 5, 6), "relu"),
    (6, 4, (3, 1, 2), (2, 1, 1), (1, 0, 0), (8, 6, 7), "none"),
    # wide-channel kernel (igemm2: LDS-DMA ring, BK=64): Cin >= 64 and Cout >= 48
    (96, 160, 3, 1, 1, (6, 7, 8), "relu"),
    (64, 64, 3, 2, 1, (9, 10, 12), "none"),
    (72, 80, 3, 1, 0, (7, 8, 9), "leaky"),
    (136, 48, 1, 1, 0, (5, 6, 7), "none"),
    (144, 432, 3, 1, 0, (6, 6, 7), "relu"),
    (64, 72, 7, 1, 0, (9, 9, 9), "none"),
    (128, 64, 4, 2, 1, (10, 10, 10), "leaky"),
    # line-aligned kw-reuse weight-gradient kernel (stride-1 W, output lines >= 16 voxels): 144x144/3, 64x64/7, 128x128/4 taps
    (144, 144, 3, 1, 0, (4, 5, 21), "relu"),
    (144, 288, 3, 1, 1, (3, 4, 40), "none"),
    (64, 64, 7, 1, 0, (8, 9, 26), "none"),
    (128, 128, 4, 1, 1, (4, 5, 33), "leaky"),
    (128, 128, 3, 1, 1, (3, 3, 18), "none"),
    # same kernel, flat row order (valid stride-1 conv over dense planes: the (h,w) plane walked as one line in the input pitch)
    (144, 144, 3, 1, 0, (4, 19, 19), "relu"),
    (64, 64, 7, 1, 0, (8, 30, 30), "none"),
    (80, 72, 3, 1, 0, (5, 23, 37), "none"),
    # CycleGAN generator stem/tail and PatchGAN layers (networks/ResNet.py:283-345, NLayerDiscriminator.py): folded
    # single-channel stems, one-channel tails of wide inputs, strided 4^3 convs
    (1, 64, 4, 2, 1, (12, 12, 12), "leaky"),
    (1, 32, 7, 1, 0, (10, 11, 12), "relu"),
    (64, 1, 7, 1, 0, (9, 9, 10), "tanh"),
    (256, 1, 4, 1, 1, (5, 6, 6), "none"),
    (72, 1, 3, 1, 1, (5, 6, 7), "none"),
    (24, 40, 3, 1, 1, (6, 6, 7), "relu"),
]


@pytest.mark.parametrize("cin,cout,k,s,p,sp,act", CONV_CASES)
def test_conv3d(cin, cout, k, s, p, sp, act):
    from raygun_amd import ops
    kk = (k,) * 3 if isinstance(k, int) else k
    x = rnd((2, cin) + sp, 1)
    w = rnd((cout, cin) + kk, 2) * (1.0 / np.sqrt(cin * np.prod(kk)))
    w = bf(w)
    b = rnd((cout,), 3)
    afn = {"none": lambda t: t, "relu": F.relu, "leaky": lambda t: F.leaky_relu(t, 0.2), "sigmoid": torch.sigmoid, "tanh": torch.tanh}[act]
    errs = run_pair(lambda x_, w_, b_: ops.conv3d(x_, w_, b_, s, p, act, 0.2),
                    lambda x_, w_, b_: afn(F.conv3d(x_, w_, b_, s, p)), [x], [w, b])
    check(errs)


CONVT_CASES = [
    (8, 6, 3, 2, 1, 1, (5, 6, 7), "none"),
    (16, 8, 3, 2, 1, 1, (4, 4, 5), "relu"),
    (4, 4, 2, 2, 0, 0, (5, 5, 5), "none"),
    (6, 10, 3, 1, 1, 0, (5, 6, 4), "none"),
    (4, 3, 3, 2, 0, 0, (4, 5, 6), "none"),
    (128, 64, 3, 2, 1, 1, (5, 6, 7), "relu"),
    (64, 80, 3, 2, 1, 1, (6, 5, 4), "none"),
]


@pytest.mark.parametrize("cin,cout,k,s,p,op,sp,act", CONVT_CASES)
def test_conv_transpose3d(cin, cout, k, s, p, op, sp, act):
    from raygun_amd import ops
    x = rnd((2, cin) + sp, 11)
    w = bf(rnd((cin, cout, k, k, k), 12) * (1.0 / np.sqrt(cin * k ** 3)))
    b = rnd((cout,), 13)
    afn = {"none": lambda t: t, "relu": F.relu}[act]
    errs = run_pair(lambda x_, w_, b_: ops.conv_transpose3d(x_, w_, b_, s, p, op, act),
                    lambda x_, w_, b_: afn(F.conv_transpose3d(x_, w_, b_, s, p, op)), [x], [w, b])
    check(errs)


def test_conv_no_bias_and_frozen_weight():
    from raygun_amd import ops
    x = rnd((1, 8, 6, 6, 6), 21)
    w = bf(rnd((8, 8, 3, 3, 3), 22) * 0.1)
    dev = torch.device("cuda")
    xi = x.to(dev).requires_grad_(True)
    wi = w.to(dev)  # no grad: discriminator frozen during the generator phase (LinkCycleLoss.py:117-119)
    y = ops.conv3d(xi, wi, None, 1, 1)
    y.float().sum().backward()
    xr = x.clone().requires_grad_(True)
    yr = F.conv3d(xr, w, None, 1, 1)
    yr.sum().backward()
    assert relerr(y, yr) < TOL and relerr(xi.grad, xr.grad) < TOL


@pytest.mark.parametrize("shape,factor", [((1, 3, 9, 7, 8), (2, 2, 2)), ((2, 12, 8, 8, 8), (2, 2, 2)),
                                          ((1, 20, 6, 9, 6), (1, 3, 2))])
def test_maxpool(shape, factor):
    from raygun_amd import ops
    x = rnd(shape, 31)
    errs = run_pair(lambda t: ops.maxpool3d(t, factor), lambda t: F.max_pool3d(t, factor, factor), [x], [])
    check(errs, 1e-6)


@pytest.mark.parametrize("cf,cg", [(3, 4), (12, 72), (8, 16), (72, 432)])
def test_upcat(cf, cg):
    from raygun_amd import ops
    f = rnd((1, cf, 20, 20, 20), 41)
    g = rnd((1, cg, 7, 7, 7), 42)

    def ref(f_, g_):
        up = F.interpolate(g_, scale_factor=(2, 2, 2), mode="nearest")[..., 1:13, 1:13, 1:13]
        return torch.cat([f_[..., 4:16, 4:16, 4:16], up], 1)

    errs = run_pair(lambda f_, g_: ops.upcat(f_, g_, (2, 2, 2), (12, 12, 12)), ref, [f, g], [])
    check(errs, 4e-3)  # gg sums 8 bf16 values then rounds once


@pytest.mark.parametrize("act", ["none", "relu", "leaky"])
@pytest.mark.parametrize("c", [5, 64])
def test_instance_norm(act, c):
    from raygun_amd import ops
    x = bf(rnd((2, c, 6, 7, 8), 51) * 3 + 1)
    afn = {"none": lambda t: t, "relu": F.relu, "leaky": lambda t: F.leaky_relu(t, 0.2)}[act]
    errs = run_pair(lambda t: ops.instance_norm(t, 1e-5, act, 0.2), lambda t: afn(F.instance_norm(t, eps=1e-5)), [x], [])
    check(errs)


def test_instance_norm_residual():
    from raygun_amd import ops
    x = bf(rnd((1, 16, 6, 6, 6), 52) * 2)
    r = rnd((1, 16, 6, 6, 6), 53)
    errs = run_pair(lambda t, r_: ops.instance_norm(t, 1e-5, "none", 0.0, r_), lambda t, r_: F.instance_norm(t) + r_, [x, r], [])
    check(errs)


@pytest.mark.parametrize("mode", ["reflect", "zeros"])
@pytest.mark.parametrize("pad", [1, 3])
def test_pad3d(mode, pad):
    from raygun_amd import ops
    x = rnd((2, 5, 6, 7, 8), 61)
    ref = (lambda t: F.pad(t, (pad,) * 6, mode="reflect")) if mode == "reflect" else (lambda t: F.pad(t, (pad,) * 6))
    errs = run_pair(lambda t: ops.pad3d(t, pad, mode), ref, [x], [])
    check(errs, 4e-3)


def test_add_and_activation():
    from raygun_amd import ops
    a, b = rnd((1, 12, 5, 5, 5), 71), rnd((1, 12, 5, 5, 5), 72)
    check(run_pair(ops.add, lambda p, q: p + q, [a, b], []), 4e-3)
    check(run_pair(lambda t: ops.activation(t, "leaky", 0.2), lambda t: F.leaky_relu(t, 0.2), [a], []), 4e-3)
    check(run_pair(lambda t: ops.activation(t, "sigmoid"), torch.sigmoid, [a], []), 8e-3)


def test_losses():
    from raygun_amd import ops
    p = torch.from_numpy(synth.uniform((1, 3, 9, 9, 9), 81))
    t = torch.from_numpy(synth.uniform((1, 3, 9, 9, 9), 82))
    w = (torch.from_numpy(synth.uniform((1, 3, 9, 9, 9), 83)) > 0.2).float() * 1.5

    def wmse(p_, t_, w_):
        sc = w_ * (p_ - t_) ** 2
        return sc[w_ > 0].mean() if len(torch.nonzero(sc)) else sc.mean()

    dev = torch.device("cuda")
    for hip, ref in [(lambda q: ops.mse_const(q, 1.0), lambda q: F.mse_loss(q, torch.ones_like(q))),
                     (lambda q: ops.smooth_l1(q * 3, t.to(q.device) * 3), lambda q: F.smooth_l1_loss(q * 3, t * 3)),
                     (lambda q: ops.weighted_mse(q, t.to(q.device), w.to(q.device)), lambda q: wmse(q, t, w)),
                     (lambda q: ops.weighted_mse(q, t.to(q.device), 0 * w.to(q.device)), lambda q: wmse(q, t, 0 * w))]:
        a = p.clone().to(dev).requires_grad_(True)
        b = p.clone().requires_grad_(True)
        la, lb = hip(a), ref(b)
        (la * 2).backward(); (lb * 2).backward()
        assert abs(float(la) - float(lb)) <= 1e-5 * max(1.0, abs(float(lb)))
        assert float((a.grad.cpu() - b.grad).abs().max()) <= 1e-6 + 1e-5 * float(b.grad.abs().max())


def test_adam_matches_torch():
    from raygun_amd import ops
    dev = torch.device("cuda")
    p0 = torch.from_numpy(synth.normal((1000,), 91))
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-2, betas=(0.5, 0.999), weight_decay=0.01)
    p = p0.clone().to(dev); m = torch.zeros_like(p); v = torch.zeros_like(p)
    for step in range(1, 4):
        g = torch.from_numpy(synth.normal((1000,), 92 + step))
        ref.grad = g.clone(); opt.step()
        ops.adam_step_(p, g.to(dev), m, v, 1e-2, 0.5, 0.999, 1e-8, 0.01, step)
    assert float((p.cpu() - ref.detach()).abs().max()) < 1e-6


def test_cpu_tensor_raises():
    from raygun_amd import ops
    with pytest.raises(RuntimeError):
        ops.conv3d(torch.zeros(1, 1, 4, 4, 4), torch.zeros(1, 1, 3, 3, 3))


def test_shape_errors_raise():
    from raygun_amd import ops
    dev = torch.device("cuda")
    with pytest.raises(RuntimeError):
        ops.conv3d(torch.zeros(1, 2, 4, 4, 4, device=dev), torch.zeros(1, 3, 3, 3, 3, device=dev))
    with pytest.raises(RuntimeError):
        ops.conv3d(torch.zeros(1, 1, 2, 2, 2, device=dev), torch.zeros(1, 1, 3, 3, 3, device=dev))


@pytest.mark.parametrize("cin,cout", [(72, 40), (432, 144), (12, 12)])
def test_fused_adam_refreshes_packed_forward_operand(cin, cout):
    """FusedAdam rewrites the bf16 forward operand of a channels-last conv weight in its own pass (rg_adam_step_pack): the
    refreshed buffer must equal a fresh rg_pack_weights of the updated master bit for bit, and the updated master must equal
    the plain fused step's."""
    from raygun_amd import hipnn, ops
    from raygun_amd.torch.optimizers import FusedAdam
    if not hipnn.CL_WEIGHTS:
        pytest.skip("channels-last master weights are disabled")
    dev = torch.device("cuda")
    conv = hipnn.Conv3d(cin, cout, 3).to(dev)
    x = rnd((1, cin, 5, 6, 7), 11).to(dev)
    opt = FusedAdam(conv.parameters(), lr=1e-2)
    for it in range(2):
        opt.zero_grad()
        conv(x).float().pow(2).mean().backward()      # forward packs the operand (iteration 0) / reuses the refreshed one
        before = conv.weight.detach().clone()
        opt.step()
        cache = conv.weight._rg_pack_cache
        assert cache["ver"] == (conv.weight._version, conv.weight.data_ptr()) and "fwd_pack" in cache
        fresh = ops.pack_weights(conv.weight.detach().clone(memory_format=torch.preserve_format), swap=0)
        assert torch.equal(cache[(0, (0, 0, 0), (1, 1, 1), None)].view(torch.int16), fresh.view(torch.int16))
        assert not torch.equal(before, conv.weight.detach())


@pytest.mark.gpu
@pytest.mark.parametrize("gs,fs,target", [((9, 9, 9), (20, 20, 20), (18, 18, 18)), ((9, 10, 11), (22, 21, 23), (16, 18, 20)),
                                           ((8, 9, 8), (16, 17, 15), (13, 15, 11)), ((10, 9, 9), (24, 18, 20), (15, 17, 13))])
def test_upcat_conv3d_equals_upsample_crop_concat_conv(gs, fs, target):
    """ops.upcat_conv3d (27 -> 8 taps on the upsampled channels, depth-to-space) against the plain composition it replaces
    (torch fp32 on bf16-exact operands): output and all four gradients, for even and odd crop offsets."""
    from raygun_amd import ops
    dev = torch.device("cuda")
    cf, cg, cout = 16, 64, 24
    torch.manual_seed(0)
    f = torch.randn(1, cf, *fs, device=dev).bfloat16().float().requires_grad_()
    g = torch.randn(1, cg, *gs, device=dev).bfloat16().float().requires_grad_()
    w = (torch.randn(cout, cf + cg, 3, 3, 3, device=dev) * 0.05).bfloat16().float().requires_grad_()
    b = torch.randn(cout, device=dev).requires_grad_()
    up = torch.nn.functional.interpolate(g, scale_factor=2, mode="nearest")
    x = torch.cat([ops.center_crop(f, target), ops.center_crop(up, target)], 1)
    ref = torch.tanh(torch.nn.functional.conv3d(x, w, b))
    go = torch.randn_like(ref).bfloat16().float()
    rg = torch.autograd.grad(ref, (f, g, w, b), go)
    got = ops.upcat_conv3d(f, g, target, w, b, "tanh")
    assert tuple(got.shape) == tuple(ref.shape)
    gg = torch.autograd.grad(got, (f, g, w, b), go.to(got.dtype) if got.dtype != go.dtype else go)
    def close(a, r, tol):
        a, r = a.float(), r.float()
        assert (a - r).abs().max().item() <= tol * r.abs().max().item() + 1e-6, ((a - r).abs().max().item(), r.abs().max().item())
    close(got, ref, 3e-2)  # pre-activations of magnitude ~10 rounded to bf16 twice
    for a, r in zip(gg, rg):
        close(a, r, 2e-2)


@pytest.mark.gpu
def test_fork_and_crop_gradients():
    """ops.fork / ops.crop (skip connections): forward are views, backward = one rg_add / rg_add_pad pass; against autograd on
    plain slicing."""
    from raygun_amd import ops
    dev = torch.device("cuda")
    torch.manual_seed(1)
    x0 = torch.randn(2, 12, 9, 10, 11, device=dev).bfloat16().float()
    xa = ops.to_cl(x0).requires_grad_()
    a, b = ops.fork(xa)
    c = ops.crop(b, (1, 2, 0), (6, 5, 9))
    assert c.data_ptr() != 0 and tuple(c.shape) == (2, 12, 6, 5, 9) and torch.equal(c.float(), x0[:, :, 1:7, 2:7, 0:9])
    ga = torch.randn(2, 12, 9, 10, 11, device=dev).bfloat16()
    gc = torch.randn(2, 12, 6, 5, 9, device=dev).bfloat16()
    (gx,) = torch.autograd.grad((a, c), xa, (ops.to_cl(ga), ops.to_cl(gc)))
    ref = ga.float().clone()
    ref[:, :, 1:7, 2:7, 0:9] += gc.float()
    assert (gx.float() - ref).abs().max().item() <= 2 ** -7 * ref.abs().max().item()   # one bf16 rounding of the sum
    # crop alone (its gradient is the zero-padded tensor, channels-last), centre_crop through the same op
    xb = ops.to_cl(x0).requires_grad_()
    (gz,) = torch.autograd.grad(ops.center_crop(xb, (5, 6, 7)), xb, ops.to_cl(torch.ones(2, 12, 5, 6, 7, device=dev)))
    want = torch.zeros_like(x0); want[:, :, 2:7, 2:8, 2:9] = 1
    assert ops.is_cl(gz) and torch.equal(gz.float(), want)


@pytest.mark.gpu
def test_wide_transposed_conv_subpixel_path():
    """Stride-2 / k3 / pad-1 transposed convolutions with >= 192 input channels run as one 8*Cout-channel convolution +
    depth-to-space (ops._scatter_conv fast path): ConvTranspose3d forward and the input gradient of a strided Conv3d, against torch."""
    import torch.nn.functional as F
    from raygun_amd import ops
    dev = torch.device("cuda")
    torch.manual_seed(3)
    x = torch.randn(1, 192, 5, 6, 7, device=dev).bfloat16().float().requires_grad_()
    w = (torch.randn(192, 24, 3, 3, 3, device=dev) * 0.05).bfloat16().float().requires_grad_()
    b = torch.randn(24, device=dev)
    ref = torch.tanh(F.conv_transpose3d(x, w, b, stride=2, padding=1, output_padding=1))
    got = ops.conv_transpose3d(x, w, b, stride=2, padding=1, output_padding=1, act="tanh")
    assert tuple(got.shape) == (1, 24, 10, 12, 14)
    assert (got.float() - ref).abs().max().item() <= 2e-2
    go = torch.randn_like(ref).bfloat16().float()
    for a, r in zip(torch.autograd.grad(got, (x, w), go), torch.autograd.grad(ref, (x, w), go)):
        assert (a.float() - r).abs().max().item() <= 2e-2 * r.abs().max().item() + 1e-6
    # strided conv: its backward-data is the same scatter with 192 source channels
    xs = torch.randn(1, 16, 10, 12, 8, device=dev).bfloat16().float().requires_grad_()
    ws = (torch.randn(192, 16, 3, 3, 3, device=dev) * 0.05).bfloat16().float()
    gs = torch.randn(1, 192, 5, 6, 4, device=dev).bfloat16().float()
    (rg,) = torch.autograd.grad(F.conv3d(xs, ws, None, stride=2, padding=1), xs, gs)
    (gg,) = torch.autograd.grad(ops.conv3d(xs, ws, None, stride=2, padding=1), xs, gs)
    assert (gg.float() - rg).abs().max().item() <= 2e-2 * rg.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("cout", [144, 128])
def test_igemm4_192_row_tiles(cout, monkeypatch):
    """The 192-row tile (FM = 3) of igemm4 is only picked for large grids; force it (RG_FM=3) on a small problem: forward and
    backward-data against torch."""
    import torch.nn.functional as F
    from raygun_amd import ops
    monkeypatch.setenv("RG_FM", "3")
    dev = torch.device("cuda")
    torch.manual_seed(5)
    x = torch.randn(1, 64, 9, 10, 11, device=dev).bfloat16().float().requires_grad_()
    w = (torch.randn(cout, 64, 3, 3, 3, device=dev) * 0.05).bfloat16().float()
    b = torch.randn(cout, device=dev)
    ref = F.conv3d(x, w, b)
    got = ops.conv3d(x, w, b)
    assert (got.float() - ref).abs().max().item() <= 1.5e-2 * ref.abs().max().item()
    go = torch.randn_like(ref).bfloat16().float()
    (rg,) = torch.autograd.grad(ref, x, go)
    (gg,) = torch.autograd.grad(got, x, go)
    assert (gg.float() - rg).abs().max().item() <= 1.5e-2 * rg.abs().max().item()
